"""PCG control flow (csrc/pcg_driver.cpp), drop-in key contract and error behaviour against the
reference fixtures, on the CPU TEST DOUBLE back end (tests/hostops).  The HIP kernels themselves are
covered by the -m gpu tests; this file pins everything around them without a GPU."""
import numpy as np
import pytest

import golden_cases
import pcg_mi355x as pm
from util import golden, relerr, check_solution_against_golden

SINGLE = [n for n, c in golden_cases.CASES.items() if c.get("grid") == (1, 1, 1) or c.get("parts") == 1]


@pytest.mark.parametrize("name", SINGLE)
def test_single_part_solve_matches_reference(hostops, name):
    brick, parts = golden_cases.build_case(name)
    g = golden(name)
    P = parts[0]
    pm.configure(comm=None)
    x = golden_cases.probe_for(brick, parts)
    assert relerr(pm.calc_mpfint(x, P), g["y_probe"]) < 1e-14
    assert relerr(pm.calc_matvec_prod(P, "Preconditioner"), g["diag"]) < 1e-14
    pm.update_bc(P)
    pm.update_preconditioner(P)
    assert relerr(P["Fext"], g["Fext"]) < 1e-14
    un_before = P["Un"].copy()
    if str(g["raised"]):
        with pytest.raises(Warning, match="TooSmallTolerance"):
            pm.solve(P)
        assert np.array_equal(P["Un"], un_before)                 # the reference leaves Un untouched when it raises
        return
    out = pm.solve(P, history=True)
    info = P["_pcg_mi355x_info"]
    if int(g["early"]):
        assert out is not None and np.array_equal(P["Un"], un_before)     # :387-395 / :421-426 return a tuple
        assert (out[1], out[3]) == (int(g["early_flag"]), int(g["early_iter"]))
        assert abs(out[2] - float(g["early_relres"])) <= 1e-6 * float(g["early_relres"]) + 1e-300
        assert relerr(out[0], g["early_x"]) < 1e-14 or np.abs(g["early_x"]).max() == 0
        return
    assert out is None
    gd = P["GlobData"]
    assert gd["TimeList_Flag"][1] == info.flag and gd["TimeList_Iter"][1] == info.iter
    tol_u = 1e-8 if info.flag == 0 else 1e-6
    check_solution_against_golden(g, info.flag, info.iter, info.relres, P["Un"], info.history, tol_u=tol_u)
    assert gd["MP_TimeRecData"]["dT_Calc"] > 0


@pytest.mark.parametrize("kind", ["sell", "ebe"])
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_randomised_load_steps_vs_oracle(hostops, oracle_c, seed, kind):
    """Seeded random models (size, pattern types, prescribed displacements, loads, tolerance, warm start) through the
    engine and through the oracle (pinned bit-exact to the reference, tests/test_oracle_golden.py): same Flag, same
    iteration count (+-1 for the matrix-free operator), same solution, same early residual history."""
    import copy
    import pcg_oracle
    from pcg_mi355x.brick import Brick, make_parts
    rng = np.random.default_rng(100 + seed)
    b = Brick(int(rng.integers(6, 11)), seed=seed, n_types=int(rng.integers(1, 4)))
    P = make_parts(b, tol=float(10.0 ** -rng.integers(5, 9)), max_iter=int(rng.integers(40, 400)))[0]
    fixed = P["LocFixedDof"]
    P["Ud"] = np.zeros(P["NDOF"]); P["Ud"][fixed] = 1e-3 * rng.standard_normal(len(fixed))          # :234
    P["RefLoadVector"] = P["RefLoadVector"] + 0.1 * rng.standard_normal(P["NDOF"])
    if seed % 2:
        P["Un"] = 1e-2 * rng.standard_normal(P["NDOF"])                                                # warm start (:378)
    Q = copy.deepcopy(P)
    ref = pcg_oracle.solve_step([Q], use_c=True)
    pm.configure(comm=None, operator=kind)
    try:
        pm.update_bc(P); pm.update_preconditioner(P)
        assert relerr(P["Fext"], Q["Fext"]) < 1e-13
        pm.solve(P, history=True)
    finally:
        pm.configure(comm=None)
    info = P["_pcg_mi355x_info"]
    tol_iter = 1 if kind == "ebe" else 0
    assert info.flag == ref["flag"] and abs(info.iter - ref["iter"]) <= tol_iter, (info.flag, info.iter, ref["flag"], ref["iter"])
    assert relerr(P["Un"], Q["Un"]) < (1e-6 if info.flag else 1e-8) * (1 if info.iter == ref["iter"] else 20)
    m = min(len(info.history), int(0.3 * len(ref["history"])))
    if m:
        assert np.abs(info.history[:m, 2] / ref["history"][:m, 2] - 1).max() < 1e-10


@pytest.mark.parametrize("kind", ["sell", "ebe"])
@pytest.mark.parametrize("N", [2, 3])
def test_smallest_meshes(hostops, N, kind):
    """One element (8 nodes, 4 of them fixed: one SELL slice, one chunk, one padded tile) and eight elements."""
    import copy
    import pcg_oracle
    from pcg_mi355x.brick import Brick, make_parts
    P = make_parts(Brick(N))[0]
    Q = copy.deepcopy(P)
    ref = pcg_oracle.solve_step([Q])
    pm.configure(comm=None, operator=kind)
    try:
        pm.update_bc(P); pm.update_preconditioner(P); pm.solve(P)
    finally:
        pm.configure(comm=None)
    info = P["_pcg_mi355x_info"]
    assert (info.flag, info.iter) == (ref["flag"], ref["iter"]) and relerr(P["Un"], Q["Un"]) < 1e-10


def test_functional_api_and_resume(hostops):
    """solve_system() + begin/run/end in several chunks gives the identical result."""
    brick, parts = golden_cases.build_case("n9_p1")
    P = parts[0]
    op = pm.get_operator(P)
    fext, _ = op.update_bc(P["RefLoadVector"], P["Ud"], 1.0)
    inv = op.build_jacobi()
    x1, info = pm.solve_system(op, fext, None, None, 1e-7, 10000, P["GlobData"]["GlobNDofEff"])
    op.solve_begin(fext, None, inv, 1e-7, 10000, P["GlobData"]["GlobNDofEff"])
    for _ in range(1000):
        r = op.solve_run(7)
        if r.status != 4 or r.iters_done >= 118:
            break
    op.solve_run(-1)
    x2, res = op.solve_end()
    assert (info.flag, info.iter) == (res.flag, res.iter) == (0, 118)
    assert np.array_equal(x1, x2)


def _solve_both_ways(monkeypatch, case, kind="sell"):
    """The same load step with the one-iteration look-ahead of the solve loop on and off."""
    from pcg_mi355x.operator import from_refmeshpart
    out = []
    for la in ("1", "0"):
        monkeypatch.setenv("PCG_LOOK_AHEAD", la)                  # read when the engine is created
        _, parts = golden_cases.build_case(case)
        P = parts[0]
        op = from_refmeshpart(P, kind=kind)
        fext, _ = op.update_bc(P["RefLoadVector"], P["Ud"], 1.0)
        x, res, hist = op.solve(fext, P["Un"], op.build_jacobi(), P["GlobData"]["Tol"], P["GlobData"]["MaxIter"],
                                P["GlobData"]["GlobNDofEff"], history=True)
        out.append((x, res.flag, res.iter, res.relres, res.iters_done, res.iters_enqueued, res.n_matvec, hist))
        op.close()
    return out


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("kind", ["sell", "ebe"])
@pytest.mark.parametrize("case", ["n9_p1", "n9_maxiter", "n9_stagnate", "n9_flag4", "n9_flag2", "oct_p1"])
def test_look_ahead_changes_nothing(hostops, monkeypatch, case, kind, fused):
    """Iteration i+1 is enqueued before the host has seen the sums of iteration i; every exit path (converged,
    MaxIter, stagnation, breakdown, inf) must give bit-identical results with the look-ahead on and off - with the
    single vector launch per iteration (fused: it already writes p of iteration i+1, so a dropped look-ahead must not
    have overwritten p of iteration i: ring of three) and with the split form of the multi-GPU loop."""
    monkeypatch.setenv("PCG_VEC_FUSED", fused)
    on, off = _solve_both_ways(monkeypatch, case, kind)
    assert np.array_equal(on[0], off[0])
    assert on[1:5] == off[1:5] and on[6] == off[6]
    assert np.array_equal(on[7], off[7])
    assert 0 <= off[5] - off[4] <= 1                              # off: the consumed iterations (+ the one a breakdown froze)
    assert 0 <= on[5] - off[5] <= 2                               # on: plus the dropped look-aheads of a break / the :527 branch


@pytest.mark.parametrize("la", ["1", "0"])
@pytest.mark.parametrize("case", ["n9_p1", "n9_maxiter", "n9_stagnate", "n9_flag4", "n9_flag2", "oct_p1", "n17_p1", "n9_raise"])
def test_fused_vector_launch_changes_nothing(hostops, monkeypatch, case, la):
    """The single-part loop forms p of iteration i+1 inside the vector launch of iteration i (Backend::vec_update with
    p_next) and skips update_p; the true-residual branch (:527-549) voids that p and falls back to update_p.  Same bits
    as the split form on every exit path."""
    from pcg_mi355x.operator import from_refmeshpart
    monkeypatch.setenv("PCG_LOOK_AHEAD", la)
    out = []
    for fused in ("1", "0"):
        monkeypatch.setenv("PCG_VEC_FUSED", fused)
        _, parts = golden_cases.build_case(case)
        P = parts[0]
        op = from_refmeshpart(P)
        fext, _ = op.update_bc(P["RefLoadVector"], P["Ud"], 1.0)
        x, res, hist = op.solve(fext, P["Un"], op.build_jacobi(), P["GlobData"]["Tol"], P["GlobData"]["MaxIter"],
                                P["GlobData"]["GlobNDofEff"], history=True)
        out.append((x, (res.flag, res.iter, res.relres, res.iters_done, res.n_matvec), hist))
        op.close()
    assert out[0][1] == out[1][1]
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][2], out[1][2])


@pytest.mark.parametrize("la", ["1", "0"])
@pytest.mark.parametrize("at", [1, 2, 17])
@pytest.mark.parametrize("case", ["n9_p1", "n9_stagnate", "oct_p1"])
def test_fused_launch_time_out_is_finished_in_the_split_form(hostops, monkeypatch, case, at, la):
    """a fused vector launch whose grid barrier times out (a workgroup of its grid not resident) no longer ends the
    solve.  The test double injects the failure into the `at`-th fused launch (r', x' written; no sums, no p', st[ERR] raised, p'
    filled with NaN): the driver finishes that iteration in the split form - the bits of an undisturbed solve - keeps to the
    split form afterwards and counts the event in pcg_result.fused_fallbacks."""
    from pcg_mi355x.operator import from_refmeshpart
    monkeypatch.setenv("PCG_LOOK_AHEAD", la)
    out = []
    for inject in (None, str(at)):
        if inject is None: monkeypatch.delenv("PCG_TEST_VEC_ERR_AT", raising=False)
        else: monkeypatch.setenv("PCG_TEST_VEC_ERR_AT", inject)
        _, parts = golden_cases.build_case(case)
        P = parts[0]
        op = from_refmeshpart(P)
        fext, _ = op.update_bc(P["RefLoadVector"], P["Ud"], 1.0)
        x, res, hist = op.solve(fext, P["Un"], op.build_jacobi(), P["GlobData"]["Tol"], P["GlobData"]["MaxIter"],
                                P["GlobData"]["GlobNDofEff"], history=True)
        out.append((x, (res.flag, res.iter, res.relres, res.iters_done), hist, res.fused_fallbacks))
        if inject is not None:                                     # the engine stays usable, in the split form
            x2, res2, _ = op.solve(fext, P["Un"], op.build_jacobi(), P["GlobData"]["Tol"], P["GlobData"]["MaxIter"], P["GlobData"]["GlobNDofEff"])
            assert res2.fused_fallbacks == 0 and np.array_equal(x2, x)
        op.close()
    assert out[0][3] == 0 and out[1][3] == 1
    assert out[0][1] == out[1][1]
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][2], out[1][2])


def test_dropped_look_ahead_leaves_no_stop_flag_behind(hostops, monkeypatch):
    """the device stop flag is sticky within a solve.  A look-ahead iteration that is DROPPED because its
    predecessor entered the true-residual branch (:527) and the loop goes on (:544-546) may have raised it (p.Ap <= 0
    formed from the recurrence residual); the iteration enqueued again from the true residual must not inherit it, or
    the solve ends with a spurious Flag 4.  The test double injects p.Ap = -1 into exactly that dropped iteration."""
    from pcg_mi355x.operator import from_refmeshpart

    def run():
        _, parts = golden_cases.build_case("n9_stagnate")          # Tol 1e-15: enters the :527 branch, not converged, continues
        P = parts[0]
        op = from_refmeshpart(P)
        fext, _ = op.update_bc(P["RefLoadVector"], P["Ud"], 1.0)
        x, res, hist = op.solve(fext, P["Un"], op.build_jacobi(), P["GlobData"]["Tol"], P["GlobData"]["MaxIter"],
                                P["GlobData"]["GlobNDofEff"], history=True)
        op.close()
        return x, res, hist
    monkeypatch.setenv("PCG_LOOK_AHEAD", "1")
    x0, r0, h0 = run()
    tolb = P_tol = 1e-15 * r0.norm_b
    first = int(np.flatnonzero(h0[:, 2] <= tolb)[0])               # loop index i of the first entry into the branch
    assert first + 1 < r0.iters_done                                # ... after which the loop went on
    monkeypatch.setenv("PCG_TEST_NEG_PQ_AT", str(first + 2))       # enqueue i+2 = the look-ahead behind iteration i, dropped
    x1, r1, h1 = run()
    assert (r1.flag, r1.iter, r1.relres, r1.iters_done) == (r0.flag, r0.iter, r0.relres, r0.iters_done)
    assert np.array_equal(x0, x1) and np.array_equal(h0, h1)
    monkeypatch.setenv("PCG_TEST_NEG_PQ_AT", str(first + 1))       # control: the same fault in the REAL iteration i is a Flag 4
    x2, r2, h2 = run()
    assert r2.flag == 4 and r2.iters_done == first


def test_look_ahead_windows_enqueue_exactly_k(hostops):
    """pcg_solve_run(K) must leave nothing in flight: a timed window of K passes is K iterations of device work."""
    _, parts = golden_cases.build_case("n9_p1")
    P = parts[0]
    op = pm.get_operator(P)
    fext, _ = op.update_bc(P["RefLoadVector"], P["Ud"], 1.0)
    op.solve_begin(fext, None, op.build_jacobi(), 1e-7, 10000, P["GlobData"]["GlobNDofEff"])
    r = op.solve_run(10)
    assert (r.iters_done, r.iters_enqueued) == (10, 10)
    r = op.solve_run(1)
    assert (r.iters_done, r.iters_enqueued) == (11, 11)
    op.solve_run(-1)
    x, res = op.solve_end()
    assert (res.flag, res.iter) == (0, 118) and res.iters_enqueued - res.iters_done == 1    # the one behind the converged iteration


def test_missing_preconditioner_is_an_error(hostops):
    _, parts = golden_cases.build_case("n9_p1")
    op = pm.get_operator(parts[0])
    with pytest.raises(pm.PcgError):
        op.solve(parts[0]["RefLoadVector"])


def test_neighbours_without_comm_is_an_error(hostops):
    _, parts = golden_cases.build_case("n9_p2")
    pm.configure(comm=None)
    with pytest.raises(pm.PcgError):
        pm.get_operator(parts[0])


@pytest.mark.parametrize("collective", [0, 1])
def test_lone_part_enters_the_exchange_only_when_the_hooks_are_collective(hostops, collective):
    """pcg_comm_hooks.collective_exchange.  A part without neighbours that has hooks set (world size 1, or an island
    part of a multi-part job) calls halo_begin(NULL, NULL, 0) / halo_end only for a communicator that implements the exchange
    as a group-wide collective; a point-to-point communicator (the reference's Isend / Recv over an empty NbrMPIdVector,
    pcg_solver.py:318-328) never sees a call."""
    from pcg_mi355x import _lib
    from pcg_mi355x._lib import check
    _, parts = golden_cases.build_case("n9_p1")
    P = parts[0]
    op = pm.get_operator(P)
    calls = {"begin": 0, "end": 0, "allreduce": 0, "bad": 0}

    def hb(ctx, send, recv, count, stream):
        calls["begin"] += 1
        calls["bad"] += int(count != 0 or bool(send) or bool(recv))
        return 0

    def he(ctx, stream):
        calls["end"] += 1
        return 0

    def ar(ctx, buf, count, stream):
        calls["allreduce"] += 1
        return 0
    keep = (_lib.HALO_BEGIN_T(hb), _lib.HALO_END_T(he), _lib.ALLREDUCE_T(ar))
    hooks = _lib.CommHooks(None, *keep, collective)
    check(op._L.pcg_set_comm(op._h, hooks), "pcg_set_comm")
    fext, _ = op.update_bc(P["RefLoadVector"], P["Ud"], 1.0)
    x, res, _ = op.solve(fext, None, op.build_jacobi(), 1e-7, 10000, P["GlobData"]["GlobNDofEff"])
    check(op._L.pcg_set_comm(op._h, None), "pcg_set_comm")
    assert (res.flag, res.iter) == (0, 118) and calls["allreduce"] > 2 * 118 and calls["bad"] == 0
    if collective:
        assert calls["begin"] == calls["end"] >= res.n_matvec
    else:
        assert calls["begin"] == calls["end"] == 0
