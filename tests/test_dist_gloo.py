"""N>1 path on CPU: one process per part, torch.distributed gloo, world sizes 2/4/8, the product's
C++ control flow + Python comm hooks (all_to_all_single interface exchange, f64 all_reduce) on the
CPU test double.  Checked against fixtures produced by the reference run with the same partition."""
import numpy as np
import pytest

import golden_cases
from util import golden, relerr, run_dist, check_solution_against_golden
from pcg_mi355x.brick import Brick


@pytest.fixture(scope="module", autouse=True)
def _built():
    import conftest
    conftest.build_hostops()


def collect(case, outs):
    n = len(golden(case)["Fext"])
    U = np.zeros(n); Y = np.zeros(n); F = np.zeros(n); D = np.zeros(n)
    for o in reversed(outs):
        U[o["dofs"]] = o["Un"]; Y[o["dofs"]] = o["y_probe"]; F[o["dofs"]] = o["Fext"]; D[o["dofs"]] = o["diag"]
    return U, Y, F, D


@pytest.mark.parametrize("case,nproc", [("n9_p2", 2), ("n13_t3_p4_ud", 4), ("n9_p8", 8), ("n9_p2_maxiter", 2), ("n9_p2_flag4", 2),
                                        ("oct_p3", 3), ("oct_p2_z", 2), ("goct_p4", 4), ("goct_p3_ud", 3), ("goct_sym_p3", 3)])
def test_multi_rank_solve_matches_reference(tmp_path, case, nproc):
    outs = run_dist(case, nproc, "gloo", "hostops", tmp_path)
    g = golden(case)
    U, Y, F, D = collect(case, outs)
    assert relerr(Y, g["y_probe"]) < 1e-14
    assert relerr(D, g["diag"]) < 1e-14
    assert relerr(F, g["Fext"]) < 1e-13
    o0 = outs[0]
    for o in outs:                                     # every rank took the same control-flow path
        assert (int(o["flag"]), int(o["iter"])) == (int(o0["flag"]), int(o0["iter"]))
        assert float(o["relres"]) == float(o0["relres"])
    assert o0["tl_flag"] == int(g["flag"]) and o0["tl_iter"] == int(g["iter"])      # rank 0 stores (:593-596)
    for o in outs[1:]:
        assert o["tl_iter"] == 0                                                       # other ranks do not
    tol_u = 1e-8 if int(g["flag"]) == 0 else 1e-6
    check_solution_against_golden(g, int(o0["flag"]), int(o0["iter"]), float(o0["relres"]), U, o0["history"], tol_u=tol_u)
    # multi-rank result file: every rank's owned dofs at its offset of ONE U_0.mpidat (file_operations.py:348-375)
    from pcg_mi355x.io import read_result_vector
    res = str(tmp_path / "ResVecData") + "/"
    dof, u = read_result_vector(res + "Dof"), read_result_vector(res + "U_0")
    assert len(dof) == len(U) and len(np.unique(dof)) == len(dof)        # each global dof exactly once (ownership mask)
    assert relerr(u, g["Un"][dof]) < tol_u
    # two all-reduces per iteration instead of the reference's three (merged, same arithmetic); a look-ahead
    # iteration that was dropped (at most one per break / entry into the true-residual branch) adds its two
    dropped = int(o0["iters_enqueued"]) - int(o0["iters_done"])
    assert 0 <= dropped <= 2
    assert int(o0["n_allreduce"]) <= int(g["n_allreduce"]) + 2 * dropped
    assert float(o0["t_comm"]) > 0


def test_multi_rank_raise(tmp_path):
    outs = run_dist("n9_p2_raise", 2, "gloo", "hostops", tmp_path)
    for o in outs:
        assert str(o["raised"]) == "PCG : TooSmallTolerance"


@pytest.mark.parametrize("kind", ["sell", "ebe"])
def test_eight_ranks_match_one_rank_at_mid_size(tmp_path, hostops, kind):
    """46 875 dof split 2x2x2 (face, edge and corner neighbours, every rank builds only ITS part) against the same
    system on one rank: same Flag and iteration count (+-1), same solution, same operator on a probe vector."""
    import pcg_mi355x as pm
    from pcg_mi355x.brick import Brick, make_parts
    b = Brick(25, n_types=2)
    P = make_parts(b)[0]
    pm.configure(comm=None, operator=kind)
    try:
        y1 = pm.calc_mpfint(np.cos(0.37 * P["DofVector"]), P)
        pm.update_bc(P); pm.update_preconditioner(P); pm.solve(P)
    finally:
        pm.configure(comm=None)
    i1 = P["_pcg_mi355x_info"]
    outs = run_dist("brick:25:2:2x2x2", 8, "gloo", "hostops", tmp_path, extra=(kind,))
    U, Y = np.zeros(b.n_dof), np.zeros(b.n_dof)
    for o in reversed(outs):
        U[o["DofVector"]] = o["Un"]; Y[o["DofVector"]] = o["y_probe"]
    assert relerr(Y, y1) < 1e-13
    assert int(outs[0]["flag"]) == i1.flag == 0 and abs(int(outs[0]["iter"]) - i1.iter) <= 1
    assert relerr(U, P["Un"]) < 2e-7
    assert all(int(o["iter"]) == int(outs[0]["iter"]) for o in outs)


@pytest.mark.parametrize("kind", ["sell", "ebe"])
@pytest.mark.parametrize("case", ["n9_p2", "n9_p8", "oct_p3"])
def test_parts_in_threads_on_the_test_double(hostops, case, kind):
    """tests/thread_comm.py (the in-process communicator the GPU suite uses to run several parts on one GPU) checked
    here on the CPU double against the same fixtures as the gloo runs."""
    from thread_comm import solve_parts_in_threads
    mesh, parts = golden_cases.build_case(case)
    g = golden(case)
    infos = solve_parts_in_threads(parts, kind, on_gpu=False)
    U = np.zeros(len(g["Un"]))
    for p in reversed(parts):
        U[p["DofVector"]] = p["Un"]
    i0 = infos[0]
    assert all((i.flag, i.iter) == (i0.flag, i0.iter) for i in infos)
    check_solution_against_golden(g, i0.flag, i0.iter, i0.relres, U, i0.history, tol_iter=1 if kind == "ebe" else 0)


def fused_and_unfused_iterations_agree(on_gpu, monkeypatch, cases=("n9_p8", "oct_p3", "goct_p4", "n9_p2_flag4", "n13_t3_p4_ud"), kinds=("sell", "ebe")):
    """Round 4: the multi-part iteration in five launches (pack in the interface rows' epilogue, p.Ap and the five sums reduced by the
    last workgroup of the fix-up / vector launch, the status copy on the next update_p; PCG_ITER_FUSED, default on) against the
    round-3 sequence of ten stream operations (=0): same arithmetic in the same order - histories and solutions bit for bit."""
    from thread_comm import solve_parts_in_threads
    for case in cases:
        for kind in kinds:
            res = {}
            for fused in ("1", "0"):
                monkeypatch.setenv("PCG_ITER_FUSED", fused)
                _, parts = golden_cases.build_case(case)
                infos = solve_parts_in_threads(parts, kind, on_gpu=on_gpu)
                res[fused] = (infos, parts)
            for a, b, pa, pb in zip(res["1"][0], res["0"][0], res["1"][1], res["0"][1]):
                assert (a.flag, a.iter, a.relres, a.iters_done) == (b.flag, b.iter, b.relres, b.iters_done), (case, kind)
                assert np.array_equal(a.history, b.history), (case, kind)
                assert np.array_equal(pa["Un"], pb["Un"]), (case, kind)


def test_fused_multi_part_iteration_is_bit_identical_on_the_test_double(hostops, monkeypatch):
    fused_and_unfused_iterations_agree(False, monkeypatch)
    monkeypatch.setenv("PCG_LOOK_AHEAD", "0")                  # no look-ahead: every status copy is flushed by a launch of its own
    fused_and_unfused_iterations_agree(False, monkeypatch, cases=("n9_p8", "oct_p3"))


def test_rank_without_neighbours_enters_the_collective_exchange(tmp_path):
    """the callback communicator's exchange is a group-wide all_to_all_single; a rank whose part has no
    neighbours must enter it too (empty splits) or the other ranks hang.  3 ranks, part 2 is an island; checked against
    the oracle run with the same three parts as virtual ranks."""
    import copy
    import pcg_oracle
    from util import island_parts
    ref = island_parts()
    out = pcg_oracle.solve_step(ref)
    outs = run_dist("island", 3, "gloo", "hostops", tmp_path, timeout=300)
    for o, R in zip(outs, ref):
        assert int(o["flag"]) == out["flag"] == 0 and abs(int(o["iter"]) - out["iter"]) <= 1
        assert relerr(o["Un"], R["Un"]) < 1e-8
    assert int(outs[2]["n_halo"]) > 0            # the island rank took part in every exchange


def test_big_multi_part_harness_on_the_test_double(oracle_c, tmp_path):
    """The harness of test_native_comm.test_eight_parts_of_the_10m_dof_brick_on_one_gpu (tests/native_comm_worker.py bigbrick: eight
    parts of a 2x2x2 split through the NATIVE communicator branch of the driver vs the oracle's calcMatVecProd + interface sum, the
    diagonal, Fext, 30 iterations of history vs one engine) at 27 783 dof on the CPU double, where there is no GPU."""
    import json
    import os
    import subprocess
    import sys
    import conftest
    from test_native_comm import WORKER, check_big_brick_report
    env = dict(os.environ, PCG_TEST_LIB=conftest.build_hostops())
    env.pop("PCG_RCCL_LIB", None)
    out = str(tmp_path / "bb.json")
    r = subprocess.run([sys.executable, WORKER, "bigbrick", "21", "sell,ebe", out, "30"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    check_big_brick_report(json.load(open(out)), ("sell", "ebe"), 30)

