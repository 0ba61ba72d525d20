"""Lock-step parity (SURVEY 7 "hard parts"): EVERY iteration of a full solve, on identical inputs.

A CG run amplifies rounding differences exponentially, so the residual histories of two correct implementations part
ways after ~100 iterations (measured on the reference against itself).  History gates therefore
only cover the start of a solve.  Here the oracle (the pinned restatement of pcg_solver.py:438-562) walks the whole
solve of the 1 M-dof brick (BASELINE configs[1]) and hands the vectors of every iteration to the engine's kernels:

    p_i  = k_update_p(r_i, p_{i-1}, beta_i)                      must EQUAL the oracle's P bit for bit   (:447,:472-479)
    q_i  = operator(p_i) with the fused p.Ap                     <= 1e-13 relative                       (:482-488)
    r', x', [|p|^2,|x|^2,|r'|^2, rho_{i+1}], p_{i+1} = k_vec     r', x' bit-equal; sums <= 1e-13;        (:501-516,:462,
                                                                 p_{i+1} == M^-1 r' + (rho_{i+1} / rho_i) p_i bit for bit   :475-479)

(k_vec = the single vector launch of the single-part loop, round 3: update, grid-wide reduction, beta, next search direction)
so the late-iteration behaviour of every kernel is pinned at the per-kernel tolerance, for both operators.  The host's
stagnation test (:512-513) is evaluated on the engine's sums and must take the oracle's decision at every iteration
(`test_stagnation_exit_in_lock_step`: the Flag-3 fixture, whose free-running exit iteration is rounding-chaotic).
"""
import copy
import os

import numpy as np
import pytest

import pcg_oracle
from pcg_mi355x._lib import check
from pcg_mi355x.brick import Brick, make_parts
from pcg_mi355x.operator import from_refmeshpart

N_FULL = int(os.environ.get("PCG_LOCKSTEP_N", "70"))     # 70 -> 1 029 000 dof, ~750 iterations


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sell", "ebe"])
def test_every_iteration_of_a_full_solve_in_lock_step(gpu_lib, oracle_c, kind):
    lock_step(kind, N_FULL)


@pytest.mark.parametrize("kind", ["sell", "ebe"])
def test_lock_step_harness_on_the_cpu_double(hostops, oracle_c, kind):
    """The same walk at 6 591 dof on the CPU test double: checks the harness (and the double) where there is no GPU."""
    lock_step(kind, 13)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sell", "ebe"])
def test_stagnation_exit_in_lock_step(gpu_lib, oracle_c, kind):
    """n9_stagnate (Tol 1e-15; the reference ends with Flag 3 at iteration 177 via :560-562): the oracle walks the solve, the
    engine's sums of EVERY iteration drive the host's stagnation test and it must decide like the oracle each time - so the
    stagnation counter, and with it the exit, are pinned on the HIP kernels with identical inputs."""
    st = lock_step(kind, 0, case="n9_stagnate", expect_flag=3)
    assert st["stag_hits"] >= 3 and st["count"] >= 100


@pytest.mark.parametrize("kind", ["sell", "ebe"])
def test_stagnation_exit_in_lock_step_on_the_cpu_double(hostops, oracle_c, kind):
    lock_step(kind, 0, case="n9_stagnate", expect_flag=3)


LOCKSTEP_10M = int(os.environ.get("PCG_LOCKSTEP_10M", "10"))      # iterations of the 10 M-dof window (0 = skip; round 3 ran 25 by hand)


@pytest.mark.gpu
@pytest.mark.skipif(LOCKSTEP_10M <= 0, reason="10 M-dof lock-step window switched off (PCG_LOCKSTEP_10M=0)")
@pytest.mark.parametrize("kind", ["sell", "ebe"])
def test_lock_step_window_at_10m_dof(gpu_lib, oracle_c, kind):
    """The metric's size (brick N = 150, 10 125 000 dof): the first PCG_LOCKSTEP_10M (default 10) iterations in lock-step - part of
    the default GPU suite since round 4 (the oracle's C mat-vec takes ~0.8 s per iteration at this size: under a minute per operator)."""
    lock_step(kind, 150, max_iter=LOCKSTEP_10M, expect_flag=1)


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["ebe_mtile", "ebe_mixed", "sell_win"])
def test_octree_solve_in_lock_step(gpu_lib, oracle_c, monkeypatch, form):
    """Round 5: the kernels round 4 added, walked like the brick's - the 1 M-dof graded octree mesh of BASELINE
    configs[1] (GradedOctreeMesh((12,12,12), 4, symmetry=True): 8 pattern types in 95 orientations, per-element dof order + sign
    word), every one of its ~570 iterations with the oracle's vectors: k_ebe_mtile + k_ebe_shared (every element on the matrix
    cores, the default at this size), k_ebe_mixed (hex section on the vector FMAs, PCG_EBE_HEX_TILES=0) and k_spmv_win (windowed
    split SELL format).  Late iterations reach these kernels with tiny p and denormal-adjacent residuals under identical inputs;
    gates as for the brick: q and p.Ap <= 1e-13, the vector phase bit-equal."""
    if form == "ebe_mixed":
        monkeypatch.setenv("PCG_EBE_HEX_TILES", "0")
    st = lock_step("ebe" if form.startswith("ebe") else "sell", 0, octree=(12, 12, 12))
    info = st["operator_info"]
    if form == "ebe_mtile":
        assert info.get("hex_tiles", 1) != 0, info
    if form == "sell_win":
        assert info["stored_blocks"] > 0
    assert st["count"] > 400


def lock_step(kind, N, case=None, max_iter=None, expect_flag=0, octree=None):
    if octree is not None:
        from pcg_mi355x.octree import GradedOctreeMesh, make_octree_parts
        b = GradedOctreeMesh(octree, 4, band=1.2, seed=0, symmetry=True)
        P = make_octree_parts(b, 1)[0]
    elif case is not None:
        import golden_cases
        b, parts = golden_cases.build_case(case)
        P = parts[0]
    else:
        b = Brick(N, seed=0)
        P = make_parts(b)[0] if max_iter is None else make_parts(b, max_iter=max_iter)[0]
    R = copy.deepcopy(P)
    op = from_refmeshpart(P, kind=kind)
    n = b.n_dof
    eff = np.asarray(P["LocDofEff"], np.int64)
    pcg_oracle.update_bc([R], use_c=True)
    pcg_oracle.update_preconditioner([R])
    minv = np.zeros(n); minv[eff] = R["InvDiagPreCondVector0"]
    minv_e = op.to_engine(minv)
    worst = {"q": 0.0, "pq": 0.0, "sums": 0.0, "rho_next": 0.0}
    state = {"p_prev": np.zeros(n), "rho_next_engine": None, "count": 0, "stag_hits": 0}
    EPS = np.finfo(float).eps
    fused = 1 if L_fused_available(op) else 0
    L, h = op._L, op._h

    def full(v):
        out = np.zeros(n); out[eff] = v
        return out

    def observer(o):
        i = o["i"]
        r0, x0, p_o, q_o, r1, x1 = (full(o[k][0]) for k in ("R_before", "X_before", "P", "Q", "R_after", "X_after"))
        # -- rho of THIS iteration was produced by the previous fused update (engine: st[RHO_NEXT]) -------------------
        if state["rho_next_engine"] is not None and np.array_equal(r0, state["r_after_prev"]):   # (not after :527-531 replaced R)
            d = abs(state["rho_next_engine"] - o["rho"]) / abs(o["rho"])
            worst["rho_next"] = max(worst["rho_next"], d)
            assert d < 1e-13, (i, d)
        # -- search direction ------------------------------------------------------------------------------------------
        r0e, x0e, pe, qe = (np.ascontiguousarray(op.to_engine(v)) for v in (r0, x0, p_o, q_o))     # keep alive across the calls
        pp = op.to_engine(state["p_prev"]).copy()
        check(L.pcg_k_update_p(h, pp.ctypes.data, r0e.ctypes.data, minv_e.ctypes.data,
                               0.0 if i == 0 else float(o["beta"]), 1 if i == 0 else 0))
        assert np.array_equal(op.from_engine(pp), p_o), i
        # -- operator + fused p.Ap -------------------------------------------------------------------------------------
        y = np.empty(n); pxy = np.zeros(1)
        check(L.pcg_k_spmv_local(h, pe.ctypes.data, y.ctypes.data, pxy.ctypes.data))
        q_e = op.from_engine(y)
        dq = np.linalg.norm(q_e[eff] - q_o[eff]) / np.linalg.norm(q_o[eff])
        dpq = abs(pxy[0] - o["pq"]) / np.dot(np.abs(p_o[eff]), np.abs(q_o[eff]))
        worst["q"], worst["pq"] = max(worst["q"], dq), max(worst["pq"], dpq)
        assert dq < 1e-13 and dpq < 1e-13, (i, dq, dpq)
        # -- residual / solution update and the five sums ----------------------------------------------------------------
        rr = r0e.copy(); xn = np.empty(n); pn = np.empty(n); sums = np.zeros(5)
        check(L.pcg_k_vec_iteration(h, float(o["alpha"]), float(o["rho"]), pe.ctypes.data, qe.ctypes.data, rr.ctypes.data,
                                    x0e.ctypes.data, xn.ctypes.data, minv_e.ctypes.data, pn.ctypes.data, sums.ctypes.data, fused))
        assert np.array_equal(op.from_engine(rr)[eff], r1[eff]) and np.array_equal(op.from_engine(xn)[eff], x1[eff]), i
        assert np.array_equal(pn, minv_e * rr + (sums[3] / float(o["rho"])) * pe), i               # :447, :475, :479
        ds = max(abs(a - c) / c if c > 0 else abs(a) for a, c in zip(sums[:3], o["sq"]))      # |x|^2 = 0 before the first update
        worst["sums"] = max(worst["sums"], ds)
        assert ds < 1e-13 and sums[4] == 0.0, (i, ds)
        # -- the host's stagnation test (:512-513) on the ENGINE's sums takes the oracle's decision ------------------------
        stag_e = np.sqrt(sums[0]) * abs(o["alpha"]) < EPS * np.sqrt(sums[1])
        stag_o = np.sqrt(o["sq"][0]) * abs(o["alpha"]) < EPS * np.sqrt(o["sq"][1])
        assert stag_e == stag_o, (i, sums[:2], o["sq"][:2])
        state["stag_hits"] += int(stag_o)
        state["rho_next_engine"] = sums[3]
        state["r_after_prev"] = r1.copy()
        state["p_prev"] = p_o
        state["count"] += 1

    out = pcg_oracle.pcg([R], use_c=True, record=False, observer=observer)
    state["operator_info"] = op.operator_info() if kind == "ebe" else op.matrix_info()
    op.close()
    assert out["flag"] == expect_flag, out["flag"]
    if expect_flag == 0:
        assert state["count"] == out["iter"]                           # every iteration of the converged solve was compared
        assert state["count"] > (600 if N >= 70 else 50)
    print(f"lock-step {kind}{' ' + case if case else ''}: {state['count']} iterations of {n} dof (fused vector launch: {bool(fused)}), "
          f"stagnation test true {state['stag_hits']} times, worst relative deviations {worst}")
    return state


def L_fused_available(op):
    """Whether the engine runs the single vector launch (one probe call on zero vectors)."""
    n = op.n
    z = np.zeros(n); o = np.ones(n); s5 = np.zeros(5); pn = np.empty(n); xn = np.empty(n); rr = np.zeros(n)
    rc = op._L.pcg_k_vec_iteration(op._h, 0.0, 1.0, z.ctypes.data, z.ctypes.data, rr.ctypes.data, z.ctypes.data, xn.ctypes.data,
                                   o.ctypes.data, pn.ctypes.data, s5.ctypes.data, 1)
    return rc == 0


@pytest.mark.parametrize("kind", ["sell", "ebe"])
def test_octree_lock_step_harness_on_the_cpu_double(hostops, oracle_c, kind):
    """The octree walk of test_octree_solve_in_lock_step on a small graded mesh (same generator, same pattern library) on the CPU double."""
    st = lock_step(kind, 0, octree=(3, 3, 3))
    assert st["count"] > 50
