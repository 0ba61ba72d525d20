"""Split SELL format (csrc/sell.cpp split_overflow; round 3): on octree meshes a slice stores its typical row length and the longer
rows continue in an overflow part.  The sum of a row keeps its order, so the operator is bit-identical to the single SELL matrix;
checked on the CPU double (same planner, same slice ranges as the HIP back end), multi-rank with gloo, and on the GPU."""
import copy

import numpy as np
import pytest

import pcg_mi355x as pm
from util import golden, relerr, run_dist, check_solution_against_golden


def _mesh_part():
    from pcg_mi355x.octree import GradedOctreeMesh, make_octree_parts
    return make_octree_parts(GradedOctreeMesh((4, 4, 4), 3, band=1.2), 1)[0]


def _apply_and_solve(P0, monkeypatch, split):
    if split is None: monkeypatch.delenv("PCG_SELL_SPLIT", raising=False)
    else: monkeypatch.setenv("PCG_SELL_SPLIT", split)
    P = copy.deepcopy(P0)
    pm.configure(comm=None, device=0, operator="sell")
    op = pm.get_operator(P)
    x = np.random.default_rng(2).standard_normal(op.n)
    y = np.array(op.apply(x))
    info = op.matrix_info()
    pm.update_bc(P); pm.update_preconditioner(P); pm.solve(P)
    return y, info, P["GlobData"]["TimeList_Flag"][1], P["GlobData"]["TimeList_Iter"][1], np.array(P["Un"]), np.array(P["InvDiagPreCondVector0"])


FORMS = ["win", "split"]      # PCG_SPMV_OVF: windowed overflow in ONE launch (k_spmv_win, round 4, the default) / k_spmv + k_spmv_ovf (round 3)


def _check(P0, monkeypatch, form):
    monkeypatch.setenv("PCG_SPMV_OVF", form)
    single = _apply_and_solve(P0, monkeypatch, "0")
    split = _apply_and_solve(P0, monkeypatch, "1")
    auto = _apply_and_solve(P0, monkeypatch, None)
    assert split[1]["nnzb"] == single[1]["nnzb"]
    assert split[1]["stored_blocks"] < 0.75 * single[1]["stored_blocks"]       # 1.6 x -> 1.06 x the true blocks on this mesh
    assert auto[1]["stored_blocks"] == single[1]["stored_blocks"]              # below 65 536 rows: not split unless forced
    for other in (split, auto):
        assert np.array_equal(other[0], single[0])                             # the mat-vec, bit for bit
        assert np.array_equal(other[5], single[5])                             # the preconditioner
        assert other[2] == single[2] == 0 and abs(other[3] - single[3]) <= 1   # (the fused p.Ap groups its terms differently)
        assert relerr(other[4], single[4]) < 1e-9


@pytest.mark.parametrize("form", FORMS)
def test_split_matrix_is_bit_identical_on_the_cpu_double(hostops, monkeypatch, form):
    _check(_mesh_part(), monkeypatch, form)


def test_brick_matrices_are_not_split(hostops, monkeypatch):
    """A brick's rows are 8 ... 27 blocks in runs of equal length: nothing to gain, the single matrix stays."""
    from pcg_mi355x.brick import Brick, make_parts
    P = make_parts(Brick(13))[0]
    monkeypatch.delenv("PCG_SELL_SPLIT", raising=False)
    pm.configure(comm=None, device=0, operator="sell")
    a = pm.get_operator(copy.deepcopy(P)).matrix_info()
    monkeypatch.setenv("PCG_SELL_SPLIT", "0")
    b = pm.get_operator(copy.deepcopy(P)).matrix_info()
    assert a == b


@pytest.mark.parametrize("case,nproc", [("goct_p4", 4), ("oct_p3", 3)])
def test_split_matrix_multi_rank(tmp_path, monkeypatch, case, nproc):        # (the windowed form: the default)
    """Interface rows first, interior rows behind the exchange: the overflow part is split at the same row, every rank runs
    base + overflow for each range (forced split: the fixtures are small)."""
    import conftest
    conftest.build_hostops()
    monkeypatch.setenv("PCG_SELL_SPLIT", "1")
    outs = run_dist(case, nproc, "gloo", "hostops", tmp_path)
    g = golden(case)
    n = len(g["Fext"])
    U = np.zeros(n); Y = np.zeros(n)
    for o in reversed(outs):
        U[o["dofs"]] = o["Un"]; Y[o["dofs"]] = o["y_probe"]
    assert relerr(Y, g["y_probe"]) < 1e-14
    o0 = outs[0]
    check_solution_against_golden(g, int(o0["flag"]), int(o0["iter"]), float(o0["relres"]), U, o0["history"], tol_u=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("form", FORMS)
def test_split_matrix_is_bit_identical_on_the_gpu(gpu_lib, monkeypatch, form):
    _check(_mesh_part(), monkeypatch, form)


@pytest.mark.gpu
@pytest.mark.parametrize("form", FORMS)
def test_large_octree_matrix_is_split_by_default(gpu_lib, monkeypatch, form):
    """At 1 M dof the automatic rule applies (>= 65 536 rows, a third of the stored blocks goes): 1.57 -> 1.05 x the true blocks
    (two-launch form; 1.11 x in the windowed form, whose overflow slices are padded per window of 4 base slices), and the mat-vec is
    still the single matrix's bit for bit."""
    from pcg_mi355x.octree import GradedOctreeMesh, make_octree_parts
    from pcg_mi355x.operator import from_refmeshpart
    P = make_octree_parts(GradedOctreeMesh((12, 12, 12), 4, band=1.2), 1)[0]
    monkeypatch.setenv("PCG_SPMV_OVF", form)
    res = {}
    for tag, env in (("single", "0"), ("auto", None)):
        if env is None: monkeypatch.delenv("PCG_SELL_SPLIT", raising=False)
        else: monkeypatch.setenv("PCG_SELL_SPLIT", env)
        op = from_refmeshpart(P, kind="sell")
        x = np.random.default_rng(8).standard_normal(op.n)
        res[tag] = (np.array(op.apply(x)), op.matrix_info())
        op.close()
    pad = {"split": 1.08, "win": 1.13}[form]
    assert res["auto"][1]["stored_blocks"] < pad * res["auto"][1]["nnzb"] < pad * res["single"][1]["stored_blocks"] / 1.38
    assert np.array_equal(res["auto"][0], res["single"][0])


@pytest.mark.gpu
@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("case", ["goct_p4", "oct_p3"])
def test_split_matrix_multi_part_on_one_gpu(gpu_lib, monkeypatch, case, form):
    """All parts of a fixture on one GPU through the thread communicator, split forced."""
    import test_gpu_parity as T
    monkeypatch.setenv("PCG_SPMV_OVF", form)
    monkeypatch.setenv("PCG_SELL_SPLIT", "1")
    T.test_multi_part_kernels_on_one_gpu(gpu_lib, case, "sell")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1])
def test_split_on_random_block_matrices_on_gpu(gpu_lib, monkeypatch, seed):
    """The same random matrices on the device, both forms of the overflow part (k_spmv_win / k_spmv + k_spmv_ovf)."""
    for form in FORMS:
        test_split_on_random_block_matrices(gpu_lib, monkeypatch, seed, form)


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_split_on_random_block_matrices(hostops, monkeypatch, seed, form):
    """Rows of 0 ... 200 blocks in random order (hub nodes, empty rows, explicit zero blocks at the end of a row) through
    pcg_create_csr: split and single matrix give the same bits, both agree with scipy."""
    import scipy.sparse as sp
    from pcg_mi355x.operator import Operator
    rng = np.random.default_rng(seed)
    nn = 64 * 37 + 11
    lens = rng.choice([0, 1, 3, 8, 27, 27, 27, 40, 99, 200], nn)
    rows, cols = [], []
    for i, ln in enumerate(lens):
        c = np.unique(rng.integers(0, nn, ln))
        rows.append(np.full(len(c), i)); cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    blocks = rng.standard_normal((len(rows), 3, 3))
    blocks[rng.random(len(rows)) < 0.05] = 0.0                               # explicit zero blocks (some end a row)
    A = sp.bsr_matrix((blocks, cols, np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=nn))])), shape=(3 * nn, 3 * nn)).tocsr()
    A.sort_indices()
    x = rng.standard_normal(3 * nn)
    monkeypatch.setenv("PCG_SPMV_OVF", form)
    ys = {}
    for tag, env in (("single", "0"), ("split", "1")):
        monkeypatch.setenv("PCG_SELL_SPLIT", env)
        op = Operator.from_csr(A.indptr, A.indices, A.data)
        ys[tag] = (np.array(op.apply(x)), op.matrix_info()["stored_blocks"])
        op.close()
    assert ys["split"][1] < 0.6 * ys["single"][1]
    assert np.array_equal(ys["split"][0], ys["single"][0])
    assert relerr(ys["split"][0], A @ x) < 1e-13
