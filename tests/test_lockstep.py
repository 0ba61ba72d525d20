"""Lock-step parity (SURVEY 7 "hard parts"): EVERY iteration of a full solve, on identical inputs.

A CG run amplifies rounding differences exponentially, so the residual histories of two correct implementations part
ways after ~100 iterations (measured on the reference against itself).  History gates therefore
only cover the start of a solve.  Here the oracle (the pinned restatement of pcg_solver.py:438-562) walks the whole
solve of the 1 M-dof brick (BASELINE configs[1]) and hands the vectors of every iteration to the engine's kernels:

    p_i  = k_update_p(r_i, p_{i-1}, beta_i)                      must EQUAL the oracle's P bit for bit   (:447,:472-479)
    q_i  = operator(p_i) with the fused p.Ap                     <= 1e-13 relative                       (:482-488)
    r', x', [|p|^2,|x|^2,|r'|^2, rho_{i+1}] = k_fused_update     r', x' bit-equal; sums <= 1e-13         (:501-516,:462)

so the late-iteration behaviour of every kernel is pinned at the per-kernel tolerance, for both operators.
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


def lock_step(kind, N):
    b = Brick(N, seed=0)
    P = make_parts(b)[0]
    R = copy.deepcopy(P)
    op = from_refmeshpart(P, kind=kind)
    n = b.n_dof
    eff = np.asarray(P["LocDofEff"], np.int64)
    pcg_oracle.update_bc([R], use_c=True)
    pcg_oracle.update_preconditioner([R])
    minv = np.zeros(n); minv[eff] = R["InvDiagPreCondVector0"]
    minv_e = op.to_engine(minv)
    worst = {"q": 0.0, "pq": 0.0, "sums": 0.0, "rho_next": 0.0}
    state = {"p_prev": np.zeros(n), "rho_next_engine": None, "count": 0}
    L, h = op._L, op._h

    def full(v):
        out = np.zeros(n); out[eff] = v
        return out

    def observer(o):
        i = o["i"]
        r0, x0, p_o, q_o, r1, x1 = (full(o[k][0]) for k in ("R_before", "X_before", "P", "Q", "R_after", "X_after"))
        # -- rho of THIS iteration was produced by the previous fused update (engine: st[RHO_NEXT]) -------------------
        if state["rho_next_engine"] is not None:
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
        rr = r0e.copy(); xn = np.empty(n); sums = np.zeros(5)
        check(L.pcg_k_fused_update(h, float(o["alpha"]), pe.ctypes.data, qe.ctypes.data, rr.ctypes.data,
                                   x0e.ctypes.data, xn.ctypes.data, minv_e.ctypes.data, sums.ctypes.data))
        assert np.array_equal(op.from_engine(rr)[eff], r1[eff]) and np.array_equal(op.from_engine(xn)[eff], x1[eff]), i
        ds = max(abs(a - c) / c if c > 0 else abs(a) for a, c in zip(sums[:3], o["sq"]))      # |x|^2 = 0 before the first update
        worst["sums"] = max(worst["sums"], ds)
        assert ds < 1e-13 and sums[4] == 0.0, (i, ds)
        state["rho_next_engine"] = sums[3]
        state["p_prev"] = p_o
        state["count"] += 1

    out = pcg_oracle.pcg([R], use_c=True, record=False, observer=observer)
    op.close()
    assert out["flag"] == 0 and state["count"] == out["iter"]          # every iteration of the converged solve was compared
    assert state["count"] > (600 if N >= 70 else 50)
    print(f"lock-step {kind}: {state['count']} iterations of {n} dof, worst relative deviations {worst}")
