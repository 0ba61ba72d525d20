"""Synthetic input generator + native host assembler (no GPU: runs on the CPU test double for the
SELL/SpMV part, the assembler itself is the product's host code)."""
import numpy as np
import pytest
import scipy.sparse as sp

import golden_cases
import pcg_oracle
from pcg_mi355x.brick import Brick, make_parts, block_partition, hex8_stiffness
from util import relerr


def rigid_modes(N):
    idx = np.arange(N ** 3)
    X = np.stack([idx % N, (idx // N) % N, idx // (N * N)], 1).astype(float)
    modes = []
    for d in range(3):
        t = np.zeros((N ** 3, 3)); t[:, d] = 1; modes.append(t.ravel())
    for (a, b) in ((0, 1), (1, 2), (0, 2)):
        r = np.zeros((N ** 3, 3)); r[:, a] = -X[:, b]; r[:, b] = X[:, a]; modes.append(r.ravel())
    return modes


def test_hex8_stiffness_properties():
    Ke = hex8_stiffness()
    assert np.allclose(Ke, Ke.T, atol=1e-15)
    w = np.linalg.eigvalsh(Ke)
    assert (w > -1e-12).all() and (np.abs(w) < 1e-12).sum() == 6          # 6 rigid-body modes
    assert Brick(70).nnz == 80990208 and Brick(150).nnz == 809238528
    assert Brick(70).n_dof == 1029000 and Brick(150).n_dof == 10125000


def bsr(rowptr, cols, vals, n_nodes):
    return sp.bsr_matrix((vals, cols, rowptr), shape=(3 * n_nodes, 3 * n_nodes)).tocsr()


@pytest.mark.parametrize("n_types", [1, 3])
def test_assembled_operator_equals_ebe_oracle(hostops, n_types):
    from pcg_mi355x.operator import assemble_bsr3
    b = Brick(7, n_types=n_types)
    P = make_parts(b)[0]
    rp, c, v = assemble_bsr3(P["SubDomainData"]["StrucDataList"], b.n_node, n_threads=3)
    assert rp[-1] == (3 * 7 - 2) ** 3
    A = bsr(rp, c, v, b.n_node)
    assert abs(A - A.T).max() < 1e-14
    x = np.random.default_rng(1).standard_normal(b.n_dof)
    assert relerr(A @ x, pcg_oracle.matvec_local(P, x)) < 1e-14
    for m in rigid_modes(7):                                               # A . rigid = 0 (no BC applied)
        assert np.abs(A @ m).max() < 1e-11 * np.abs(m).max()
    # sign patterns must not change the physical operator
    if n_types > 1:
        P1 = make_parts(Brick(7, n_types=1))[0]
        rp1, c1, v1 = assemble_bsr3(P1["SubDomainData"]["StrucDataList"], b.n_node)
        assert np.array_equal(rp, rp1) and np.array_equal(c, c1) and np.abs(v - v1).max() < 1e-14


def test_assembly_is_deterministic_and_thread_independent(hostops):
    from pcg_mi355x.operator import assemble_bsr3
    b = Brick(6, n_types=2)
    P = make_parts(b)[0]
    g = P["SubDomainData"]["StrucDataList"]
    a = assemble_bsr3(g, b.n_node, n_threads=1)
    c = assemble_bsr3(g, b.n_node, n_threads=5)
    for u, w in zip(a, c):
        assert np.array_equal(u, w)


def test_node_permutation(hostops):
    from pcg_mi355x.operator import assemble_bsr3
    b = Brick(5)
    P = make_parts(b)[0]
    g = P["SubDomainData"]["StrucDataList"]
    perm = np.random.default_rng(3).permutation(b.n_node)
    A = bsr(*assemble_bsr3(g, b.n_node), b.n_node)
    Ap = bsr(*assemble_bsr3(g, b.n_node, node_perm=perm), b.n_node)
    dmap = (3 * perm[:, None] + np.arange(3)).ravel()
    x = np.random.default_rng(4).standard_normal(b.n_dof)
    xp = np.empty_like(x); xp[dmap] = x
    assert relerr((Ap @ xp)[dmap], A @ x) < 1e-14


def test_bad_dof_index_is_an_error(hostops):
    from pcg_mi355x.operator import assemble_bsr3
    from pcg_mi355x import PcgError
    b = Brick(4)
    P = make_parts(b)[0]
    g = dict(P["SubDomainData"]["StrucDataList"][0])
    t = g["ElemList_LocDofVector"].copy(); t[0, 0] = 3 * b.n_node + 5
    g["ElemList_LocDofVector"] = t
    with pytest.raises(PcgError):
        assemble_bsr3([g], b.n_node)


@pytest.mark.parametrize("rpl", [1, 2])
@pytest.mark.parametrize("grid", [(1, 1, 1), (2, 1, 2)])
def test_sell_spmv_and_interface_lists_on_test_double(hostops, rpl, grid):
    """SELL conversion, boundary-first renumbering and the fix-up lists, via the engine API on the
    CPU test double, against the oracle's per-part local mat-vec."""
    from pcg_mi355x.operator import from_refmeshpart
    b = Brick(8, n_types=2)
    parts = make_parts(b, block_partition(b, *grid))
    x = np.random.default_rng(5).standard_normal(b.n_dof)

    class NoComm:          # local checks only: no exchange is triggered by pcg_k_spmv_local
        rank = 0
        def make_hooks(self, op):
            from pcg_mi355x import _lib
            return _lib.CommHooks()
        def reraise(self):
            pass
    for P in parts:
        op = from_refmeshpart(P, comm=NoComm() if len(parts) > 1 else None, rows_per_lane=rpl)
        info = op.matrix_info()
        assert info["slice_rows"] == 64 * rpl and info["stored_blocks"] >= info["nnzb"]
        xl = x[P["DofVector"]]
        xe = op.to_engine(xl)
        y = np.empty(op.n)
        import ctypes as C
        pxy = C.c_double()
        from pcg_mi355x._lib import check
        check(op._L.pcg_k_spmv_local(op._h, xe.ctypes.data, y.ctypes.data, C.byref(pxy)))
        ref = pcg_oracle.matvec_local(P, xl)
        assert relerr(op.from_engine(y), ref) < 1e-14
        w = np.zeros(P["NDOF"]); w[P["LocDofEff"]] = P["DofWeightVector_Eff"]
        assert abs(pxy.value - np.dot(xl, ref * w)) <= 1e-12 * abs(np.dot(np.abs(xl), np.abs(ref)))
        op.close()


def test_operator_from_scalar_csr(hostops):
    """pcg_create_csr: an already assembled scipy CSR matrix in, same engine (SELL-BSR3) behind it."""
    from pcg_mi355x.operator import assemble_bsr3, Operator
    b = Brick(6, n_types=2)
    P = make_parts(b)[0]
    A = bsr(*assemble_bsr3(P["SubDomainData"]["StrucDataList"], b.n_node), b.n_node)
    A.eliminate_zeros()                                     # ragged blocks: some 3x3 blocks lose entries
    op = Operator.from_csr(A.indptr, A.indices, A.data)
    x = np.random.default_rng(9).standard_normal(b.n_dof)
    assert relerr(op.apply(x), pcg_oracle.matvec_local(P, x)) < 1e-14
    assert relerr(op.diag(), A.diagonal()) < 1e-15
    free = np.zeros(b.n_dof, bool); free[P["LocDofEff"]] = True
    op.set_masks(np.ones(b.n_dof, bool), free)
    inv = op.build_jacobi()
    xs, res, _ = op.solve(P["RefLoadVector"], None, inv, 1e-7, 5000, int(free.sum()))
    assert res.flag == 0
    r = (P["RefLoadVector"] - A @ xs)[free]
    assert np.linalg.norm(r) / np.linalg.norm(P["RefLoadVector"][free]) < 1.01e-7
    with pytest.raises(Exception):
        Operator.from_csr(np.array([0, 1, 2]), np.array([0, 1]), np.array([1.0, 1.0]))     # n not a multiple of 3


def test_operator_from_scalar_csr_kept_scalar(hostops):
    """pcg_create_csr(block = 1): the scalar format is kept (12 B per non-zero), any n - here a system with
    n % 3 != 0 (a 2-D 5-point Laplacian, ragged rows) and the brick operator, both solved to Tol."""
    import scipy.sparse as sp
    from pcg_mi355x.operator import assemble_bsr3, Operator
    m = 19
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(m, m))
    A = (sp.kron(sp.eye(m), T) + sp.kron(T, sp.eye(m))).tocsr()          # n = 361, rows of 3..5 entries
    assert A.shape[0] % 3 != 0
    op = Operator.from_csr(A.indptr, A.indices, A.data, block=1)
    rng = np.random.default_rng(4)
    x = rng.standard_normal(A.shape[0])
    assert relerr(op.apply(x), A @ x) < 1e-14
    assert relerr(op.diag(), A.diagonal()) < 1e-15
    inv = op.build_jacobi()
    b = rng.standard_normal(A.shape[0])
    xs, res, _ = op.solve(b, None, inv, 1e-9, 5000, A.shape[0])
    assert res.flag == 0 and np.linalg.norm(b - A @ xs) / np.linalg.norm(b) < 1.01e-9
    op.close()
    # the brick operator, scalar vs blocked format: same operator, same iteration path
    bk = Brick(6, n_types=2)
    P = make_parts(bk)[0]
    A3 = bsr(*assemble_bsr3(P["SubDomainData"]["StrucDataList"], bk.n_node), bk.n_node)
    free = np.zeros(bk.n_dof, bool); free[P["LocDofEff"]] = True
    out = []
    for blk in (1, 3):
        o = Operator.from_csr(A3.indptr, A3.indices, A3.data, block=blk)
        xb = np.random.default_rng(11).standard_normal(bk.n_dof)
        assert relerr(o.apply(xb), A3 @ xb) < 1e-13
        o.set_masks(np.ones(bk.n_dof, bool), free)
        xs, res, _ = o.solve(P["RefLoadVector"], None, o.build_jacobi(), 1e-7, 5000, int(free.sum()))
        out.append((xs, res.flag, res.iter))
        o.close()
    assert out[0][1] == out[1][1] == 0 and abs(out[0][2] - out[1][2]) <= 1
    assert relerr(out[0][0], out[1][0]) < 1e-7
    with pytest.raises(Exception):
        Operator.from_csr(A.indptr, A.indices, A.data, block=2)


def check_scalar_copy(rows_per_lane=0):
    """pcg_create_scalar_copy (round 4): the literal CSR data volume of an assembled engine, expanded from the 3x3-block format by
    the back end - same product as the block operator and as the oracle, 12 B per stored non-zero, the fused dot."""
    import ctypes as C
    from pcg_mi355x.operator import from_refmeshpart
    from pcg_mi355x._lib import check, PcgError
    b = Brick(14, n_types=2)                       # 2 744 nodes: 43 block slices, the last one ragged; 129 scalar slices
    P = make_parts(b)[0]
    op = from_refmeshpart(P, kind="sell", rows_per_lane=rows_per_lane)
    sc = op.scalar_copy()
    info_b, info_s = op.matrix_info(), sc.matrix_info()
    assert info_s["nnzb"] == 9 * info_b["nnzb"] and info_s["slice_rows"] == 64 and info_s["n_slices"] == -(-b.n_dof // 64)
    by, fl = sc.operator_cost()
    assert by == 12.0 * info_s["stored_blocks"] + 16.0 * b.n_dof + 8.0 * (info_s["n_slices"] + 1) and fl == 2.0 * info_s["nnzb"]
    x = np.random.default_rng(3).standard_normal(b.n_dof)
    y = np.empty(b.n_dof); pxy = C.c_double()
    check(sc._L.pcg_k_spmv_local(sc._h, x.ctypes.data, y.ctypes.data, C.byref(pxy)))
    ref = pcg_oracle.matvec_local(P, x)
    assert relerr(y, ref) < 1e-13 and relerr(y, op.apply(x)) < 1e-13
    assert abs(pxy.value - np.dot(x, ref)) <= 1e-12 * np.dot(np.abs(x), np.abs(ref))      # default masks: every dof owned and free
    assert sc.bench_spmv(1, 2).shape == (2,)
    sc.close()
    x2 = np.random.default_rng(4).standard_normal(b.n_dof)
    assert relerr(op.apply(x2), pcg_oracle.matvec_local(P, x2)) < 1e-13                 # the source engine is untouched
    op.close()
    e = from_refmeshpart(P, kind="ebe")
    with pytest.raises(PcgError, match="plain, unsplit"):
        e.scalar_copy()
    e.close()


def test_scalar_copy_of_an_assembled_engine(hostops):
    check_scalar_copy()
