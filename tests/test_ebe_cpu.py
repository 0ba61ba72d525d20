"""Matrix-free (element-by-element) operator, SURVEY 8(f)-1: host set-up (colouring, phases, packing),
the driver's EBE apply path and the N>1 exchange, on the CPU test double, against the reference
fixtures and the oracle."""
import copy

import numpy as np
import pytest

import golden_cases
import pcg_oracle
import pcg_mi355x as pm
from util import golden, relerr, check_solution_against_golden, run_dist, make_super_part

SINGLE = ["n9_p1", "n17_p1", "n9_maxiter", "n9_raise", "n9_zero_rhs", "oct_p1", "goct_p1", "goct_sym_p1"]


@pytest.fixture(autouse=True)
def _reset_cfg():
    yield
    pm.configure(comm=None, operator="sell")


@pytest.mark.parametrize("name", SINGLE)
def test_ebe_solve_matches_reference(hostops, name):
    brick, parts = golden_cases.build_case(name)
    g = golden(name)
    P = parts[0]
    pm.configure(comm=None, operator="ebe")
    op = pm.get_operator(P)
    info = op.operator_info()
    assert info["kind"] == "ebe" and info["n_chunks"] >= 1
    if hasattr(brick, "n_elem"):
        assert info["n_elem"] == brick.n_elem and info["n_chunks"] >= -(-brick.n_elem // 512)
    x = golden_cases.probe_for(brick, parts)
    assert relerr(pm.calc_mpfint(x, P), g["y_probe"]) < 1e-14
    assert np.array_equal(pm.calc_matvec_prod(P, "Preconditioner"), g["diag"])      # same order as np.bincount (:300)
    pm.update_bc(P); pm.update_preconditioner(P)
    if str(g["raised"]):
        with pytest.raises(Warning, match="TooSmallTolerance"):
            pm.solve(P)
        return
    out = pm.solve(P, history=True)
    if int(g["early"]):
        assert out is not None and out[1] == int(g["early_flag"])
        return
    inf = P["_pcg_mi355x_info"]
    check_solution_against_golden(g, inf.flag, inf.iter, inf.relres, P["Un"], inf.history, tol_iter=1,
                                  tol_u=1e-8 if inf.flag == 0 else 1e-6)


@pytest.mark.parametrize("chunked", [True, False])
@pytest.mark.parametrize("ept", ["1", "2"])
def test_chunked_and_per_colour_forms_agree_with_oracle(hostops, chunked, ept, monkeypatch):
    """3 pattern types with sign masks: per-colour form and chunked form (1 and 2 elements per thread)."""
    monkeypatch.setenv("PCG_EBE_EPT", ept)
    from pcg_mi355x.brick import Brick, make_parts
    b = Brick(12, n_types=3)
    P = make_parts(b)[0]
    pm.configure(comm=None, operator="ebe", ebe_chunked=chunked)
    op = pm.get_operator(P)
    info = op.operator_info()
    assert (info["n_chunks"] > 0) == chunked
    if not chunked:
        assert info["n_colors"] == 8          # global spatial order keeps the 2x2x2 parity colouring across groups
    x = np.random.default_rng(2).standard_normal(b.n_dof)
    assert relerr(op.apply(x), pcg_oracle.matvec_local(P, x)) < 1e-14
    pm.configure(comm=None, operator="sell")


def mixed_chunk_cases():
    """(name, part) pairs for the mixed-type chunk tests (CPU double and GPU): several hex8 types with sign frames, hanging-node
    patterns of 9-20 nodes, a 13-node transition pattern, and a single-type mesh (mixed form forced)."""
    from pcg_mi355x.brick import Brick, make_parts
    from pcg_mi355x.octree import GradedOctreeMesh, TwoLevelMesh, make_octree_parts
    yield "brick_3_types", make_parts(Brick(12, n_types=3))[0]
    yield "graded_octree", make_octree_parts(GradedOctreeMesh((4, 4, 4), 3, band=1.2), 1)[0]
    yield "two_level_octree", make_octree_parts(TwoLevelMesh(8, 8, 4, 3), 1)[0]
    yield "brick_1_type", make_parts(Brick(11))[0]
    # one pattern type per symmetry class, every element with its own dof order and signs (tiles only; PCG_EBE_MIXED=0: colour launches)
    yield "oriented_octree", make_octree_parts(GradedOctreeMesh((4, 4, 4), 3, band=1.2, symmetry=True), 1, sign_seed=11)[0]
    yield "oriented_brick", orient_hex8_elements(make_parts(Brick(9))[0], seed=3)


def orient_hex8_elements(P, seed):
    """Give every hex8 element of a one-type brick part a random one of the cube's 48 orientations: its dof list is re-ordered
    (node order, component order) and its sign vector set so that the SAME isotropic element matrix describes it (K = Q^T K Q for
    every cube symmetry Q) - the operator is unchanged up to rounding, but no element lists its dofs in x, y, z order any more."""
    from pcg_mi355x.octree import cube_symmetries, _sym_point, _CORNERS
    (g,) = P["SubDomainData"]["StrucDataList"]
    tbl, ne = g["ElemList_LocDofVector"], g["N_Elem"]
    syms = cube_symmetries()
    pick = np.random.default_rng(seed).integers(1, 48, ne)
    new_tbl, new_sgn = tbl.copy(), np.zeros_like(g["ElemList_SignVector"])
    for k in np.unique(pick):
        perm, sg = syms[k]
        src = [_CORNERS.index(_sym_point(syms[k], pt)) for pt in _CORNERS]
        comp = [perm.index(c) for c in range(3)]
        w = np.flatnonzero(pick == k)
        for l in range(8):
            for c in range(3):
                new_tbl[3 * l + c, w] = 3 * (tbl[3 * src[l], w] // 3) + comp[c]
                new_sgn[3 * l + c, w] = sg[comp[c]] < 0
    g["ElemList_LocDofVector"] = new_tbl
    g["ElemList_LocDofVector_Flat"] = new_tbl.ravel()
    g["ElemList_SignVector"] = new_sgn ^ g["ElemList_SignVector"]
    g["ElemList_LocNodeIdVector"] = np.ascontiguousarray(new_tbl[0::3] // 3)
    P["Flat_ElemLocDof"] = new_tbl.ravel()
    return P


@pytest.mark.parametrize("ept", ["1", "2"])
def test_mixed_type_chunks_agree_with_oracle_and_with_per_type_chunks(hostops, ept, monkeypatch):
    """Round 4: chunks that hold the elements of EVERY pattern type of a run of the Morton order (hex section + 16-element
    matrix-core tiles; csrc/ebe.cpp mixed planner, k_ebe_mixed / its CPU double) against the oracle's mat-vec and against the
    per-type chunks of round 3 (PCG_EBE_MIXED=0); fewer boundary slots; the fused p.Ap; a whole solve."""
    import ctypes as C
    from pcg_mi355x._lib import check
    from pcg_mi355x.operator import from_refmeshpart
    monkeypatch.setenv("PCG_EBE_EPT", ept)
    for name, P in mixed_chunk_cases():
        ys = {}
        # mixed chunks with the hex section (k_ebe_mixed), mixed chunks with the 8-node type in colour-pure matrix-core tiles
        # (k_ebe_mtile, PCG_EBE_HEX_TILES=1: the default below 1.2 M elements), per-type chunks (round 3)
        for mixed in ("1", "1t", "0", "auto"):                    # auto: the planner's own choice (no switch set)
            if mixed == "auto":
                monkeypatch.delenv("PCG_EBE_MIXED"); monkeypatch.delenv("PCG_EBE_HEX_TILES")
            else:
                monkeypatch.setenv("PCG_EBE_MIXED", mixed[0])
                monkeypatch.setenv("PCG_EBE_HEX_TILES", "1" if mixed == "1t" else "0")
            op = from_refmeshpart(copy.deepcopy(P), kind="ebe")
            x = np.random.default_rng(5).standard_normal(op.n)
            y = np.empty(op.n); pxy = C.c_double()
            xe = op.to_engine(x)
            check(op._L.pcg_k_spmv_local(op._h, xe.ctypes.data, y.ctypes.data, C.byref(pxy)))
            ys[mixed] = op.from_engine(y)
            ref = pcg_oracle.matvec_local(P, x)
            assert relerr(ys[mixed], ref) < 1e-14, (name, mixed)
            w = np.zeros(op.n); w[P["LocDofEff"]] = 1.0
            assert abs(pxy.value - np.dot(x, ref * w)) <= 1e-12 * np.dot(np.abs(x), np.abs(ref)), (name, mixed)
            if mixed in ("1", "1t") or (mixed == "0" and not name.startswith("oriented")):
                assert op.operator_info()["n_colors"] <= (1 if mixed != "0" else 4)     # launches per phase
            op.close()
        assert relerr(ys["1"], ys["0"]) < 1e-14 and relerr(ys["1t"], ys["0"]) < 1e-14 and relerr(ys["auto"], ys["0"]) < 1e-14
    monkeypatch.setenv("PCG_EBE_MIXED", "1")
    P = dict(mixed_chunk_cases())["graded_octree"]
    R = copy.deepcopy(P)
    pm.configure(comm=None, operator="ebe")
    pm.update_bc(P); pm.update_preconditioner(P); pm.solve(P)
    out = pcg_oracle.solve_step([R])
    assert P["GlobData"]["TimeList_Flag"][1] == out["flag"] == 0 and abs(P["GlobData"]["TimeList_Iter"][1] - out["iter"]) <= 1
    assert relerr(P["Un"], R["Un"]) < 1e-8


@pytest.mark.parametrize("kind", ["ebe", "sell"])
@pytest.mark.parametrize("N", [7, 8])
def test_patterns_with_nd_36_and_mixed_groups(hostops, kind, N):
    b, P = make_super_part(N)
    R = copy.deepcopy(P)
    pm.configure(comm=None, operator=kind)
    x = np.random.default_rng(1).standard_normal(b.n_dof)
    assert relerr(pm.calc_mpfint(x, P), pcg_oracle.matvec_local(R, x)) < 1e-14
    assert relerr(pm.calc_matvec_prod(P, "Preconditioner"), pcg_oracle.matvec_local(R, None, "Preconditioner")) < 1e-14
    pm.update_bc(P); pm.update_preconditioner(P); pm.solve(P)
    out = pcg_oracle.solve_step([R])
    assert P["GlobData"]["TimeList_Flag"][1] == out["flag"] == 0
    assert abs(P["GlobData"]["TimeList_Iter"][1] - out["iter"]) <= 1
    assert relerr(P["Un"], R["Un"]) < 1e-8


@pytest.mark.parametrize("case,nproc", [("n9_p2", 2), ("n13_t3_p4_ud", 4), ("oct_p3", 3), ("goct_p4", 4), ("goct_p3_ud", 3), ("goct_sym_p3", 3)])
def test_ebe_multi_rank(tmp_path, case, nproc):
    import conftest
    conftest.build_hostops()
    outs = run_dist(case, nproc, "gloo", "hostops", tmp_path, extra=["ebe"])
    g = golden(case)
    n = len(g["Un"])
    U = np.zeros(n); Y = np.zeros(n)
    for o in reversed(outs):
        U[o["dofs"]] = o["Un"]; Y[o["dofs"]] = o["y_probe"]
    assert relerr(Y, g["y_probe"]) < 1e-14
    o0 = outs[0]
    check_solution_against_golden(g, int(o0["flag"]), int(o0["iter"]), float(o0["relres"]), U, o0["history"], tol_iter=1)
