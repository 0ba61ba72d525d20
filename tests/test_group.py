"""The device group (C ABI pcg_group_*, csrc/group.cpp): ONE process drives every part of a model, one
library-owned host thread per member, communication through the engine's native communicator.

CPU tier: the test double's in-process communicator (tests/hostops/local_comm.cpp) - this is also the only `not gpu`
coverage of the driver's native-communicator branch (comm->halo_begin / halo_end / allreduce; the gloo tests drive the
callback seam).  GPU tier: tests/fakenccl on the one-GPU box (members share device 0), real RCCL across GPUs when the
box has them.  Fixtures: what the reference produced with the same partitions (tests/golden)."""
import copy
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import golden_cases
from util import ROOT, golden, relerr, check_solution_against_golden

WORKER = os.path.join(ROOT, "tests", "native_comm_worker.py")


def _global(parts, key_or_list, n):
    out = np.zeros(n)
    for k in range(len(parts) - 1, -1, -1):
        out[parts[k]["DofVector"]] = parts[k][key_or_list] if isinstance(key_or_list, str) else key_or_list[k]
    return out


def _run_case(case, kind, devices=None, mailbox=False, direct=False):
    from pcg_mi355x.group import GroupSolver
    mesh, parts = golden_cases.build_case(case)
    g = golden(case)
    n = len(g["Fext"])
    probe = golden_cases.probe_for(mesh, parts)
    gs = GroupSolver(parts, devices=devices, operator=kind, timing=True)
    try:
        if mailbox:
            assert gs.group.enable_mailbox(), "the mailbox all-reduce did not come up"
        if direct:
            assert gs.group.enable_direct_exchange(), "the direct exchange did not come up"
        ys = gs.group.apply([probe[P["DofVector"]] for P in parts])
        ds = gs.group.diag()
        assert relerr(_global(parts, ys, n), g["y_probe"]) < 1e-13
        assert relerr(_global(parts, ds, n), g["diag"]) < 1e-14
        gs.updateBC()
        gs.updatePreconditioner()
        assert relerr(_global(parts, "Fext", n), g["Fext"]) < 1e-13
        # the weighted dot of the reference (:381, on the free dofs :377) through the group
        want = sum(float(np.dot(P["Fext"][P["LocDofEff"]], (P["Fext"] * P["DofWeightVector"])[P["LocDofEff"]])) for P in parts)
        assert abs(gs.group.dot_w([P["Fext"] for P in parts], [P["Fext"] for P in parts]) / want - 1) < 1e-13
        ret = gs.PCG(history=True)
        assert ret is None
        infos = [P["_pcg_mi355x_info"] for P in parts]
        i0 = infos[0]
        for i in infos:                                   # every member took the same decisions on the same sums
            assert (i.flag, i.iter, i.relres, i.iters_done) == (i0.flag, i0.iter, i0.relres, i0.iters_done)
            assert np.array_equal(i.history, i0.history)
        tol_u = 1e-8 if int(g["flag"]) == 0 else 1e-6
        check_solution_against_golden(g, i0.flag, i0.iter, i0.relres, _global(parts, "Un", n), i0.history,
                                      tol_iter=1 if kind == "ebe" else 0, tol_u=tol_u)
        gd = parts[0]["GlobData"]
        step = gd["TimeStepCount"]
        assert (gd["TimeList_Flag"][step], gd["TimeList_Iter"][step]) == (i0.flag, i0.iter)
        st = [c.stats() for c in gs.group.comms]
        assert all(s["n_allreduce"] >= 2 * i0.iters_done for s in st)
        assert all(s["n_halo"] >= i0.iters_done for s in st)
        return parts, infos
    finally:
        gs.close()


@pytest.mark.parametrize("kind", ["sell", "ebe"])
@pytest.mark.parametrize("case", ["n9_p2", "n9_p8", "oct_p3", "n13_t3_p4_ud", "n9_p2_flag4", "n9_p2_maxiter"])
def test_group_solve_matches_reference_fixture(hostops, case, kind):
    _run_case(case, kind)


@pytest.mark.parametrize("kind", ["sell", "ebe"])
def test_group_is_bit_identical_to_one_thread_per_part_over_the_callback_seam(hostops, kind):
    """Same parts, same sums in rank order: the library's own threads + native communicator branch must reproduce the
    Python threads + pcg_comm_hooks run bit for bit (iteration history and solution)."""
    from thread_comm import solve_parts_in_threads
    _, parts_a = golden_cases.build_case("n9_p8")
    infos_a = solve_parts_in_threads(parts_a, kind, on_gpu=False)
    parts_b, infos_b = _run_case("n9_p8", kind)
    assert np.array_equal(infos_a[0].history, infos_b[0].history)
    for A, B in zip(parts_a, parts_b):
        assert np.array_equal(A["Un"], B["Un"])


def mailbox_reduction_is_bit_identical(cases=("n9_p8", "oct_p3", "n13_t3_p4_ud", "n9_p2_flag4", "goct_p4"), kinds=("sell", "ebe")):
    """Round 5, opt-in pcg_comm_enable_mailbox: MPI_SUM (pcg_solver.py:622-628) through peer-mapped mailboxes - p.Ap inside
    the interface fix-up launch, the five sums inside the vector launch, every other all-reduce by a one-wave kernel - summed in rank
    order like the collective it replaces in these runs: histories, exits and solutions bit for bit, every fixture still reproduced."""
    for case in cases:
        for kind in kinds:
            parts_a, infos_a = _run_case(case, kind)
            parts_b, infos_b = _run_case(case, kind, mailbox=True)
            for a, b, pa, pb in zip(infos_a, infos_b, parts_a, parts_b):
                assert (a.flag, a.iter, a.relres, a.iters_done) == (b.flag, b.iter, b.relres, b.iters_done), (case, kind)
                assert np.array_equal(a.history, b.history), (case, kind)
                assert np.array_equal(pa["Un"], pb["Un"]), (case, kind)


@pytest.mark.parametrize("kind", ["sell", "ebe"])
@pytest.mark.parametrize("mailbox", [False, True])
def test_group_with_a_member_without_neighbours(hostops, kind, mailbox):
    """A disconnected component as its own part: the member takes part in every all-reduce but in no exchange - the native
    exchange is point-to-point like the reference's Isend/Recv loops over an empty NbrMPIdVector (pcg_solver.py:318-328).
    Against the oracle's run of the same three parts."""
    import pcg_oracle
    from pcg_mi355x.group import GroupSolver
    from util import island_parts
    ref = island_parts()
    out = pcg_oracle.solve_step(ref)
    parts = island_parts()
    gs = GroupSolver(parts, operator=kind)
    try:
        if mailbox:        # the island member has no fix-up launch to carry p.Ap: its all-reduce is the one-wave kernel, same sequence number
            assert gs.group.enable_mailbox()
        gs.updateBC(); gs.updatePreconditioner(); gs.PCG()
        i0 = parts[0]["_pcg_mi355x_info"]
        assert i0.flag == out["flag"] == 0 and abs(i0.iter - out["iter"]) <= 1
        for P, R in zip(parts, ref):
            assert relerr(P["Un"], R["Un"]) < 1e-8
        st = [c.stats() for c in gs.group.comms]
        assert st[2]["n_halo"] == 0 and st[0]["n_halo"] > 0 and st[2]["n_allreduce"] == st[0]["n_allreduce"] > 0
    finally:
        gs.close()


def test_group_of_one(hostops):
    """A single member: no exchange, all-reduces over one rank - the fixture of the one-part run."""
    parts, infos = _run_case_one()
    assert len(parts) == 1


def _run_case_one():
    from pcg_mi355x.group import GroupSolver
    _, parts = golden_cases.build_case("n9_p1")
    g = golden("n9_p1")
    gs = GroupSolver(parts)
    try:
        gs.updateBC(); gs.updatePreconditioner(); gs.PCG(history=True)
        info = parts[0]["_pcg_mi355x_info"]
        check_solution_against_golden(g, info.flag, info.iter, info.relres, parts[0]["Un"], info.history)
        return parts, [info]
    finally:
        gs.close()


def test_group_raises_where_the_reference_raises(hostops):
    """pcg_solver.py:549 raise Warning('PCG : TooSmallTolerance') - on every rank there, once for the group here."""
    from pcg_mi355x.group import GroupSolver
    _, parts = golden_cases.build_case("n9_p2_raise")
    gs = GroupSolver(parts)
    try:
        gs.updateBC(); gs.updatePreconditioner()
        with pytest.raises(Warning, match="TooSmallTolerance"):
            gs.PCG()
    finally:
        gs.close()


def test_group_load_steps_warm_start(hostops):
    """The load-step loop (:1002-1008) over a 4-part model with non-zero Dirichlet data: every step starts from the previous
    Un of every part; against the oracle's multi-part loop."""
    import pcg_oracle
    from pcg_mi355x.group import GroupSolver
    _, parts = golden_cases.build_case("n13_t3_p4_ud")
    for P in parts:
        P["GlobData"]["TimeStepDelta"] = [0, 0.4, 1.0]
        P["GlobData"]["RefMaxTimeStepCount"] = 3
        for k in ("TimeList_Flag", "TimeList_RelRes", "TimeList_Iter"):
            P["GlobData"][k] = np.zeros(3)
    ref = copy.deepcopy(parts)
    gs = GroupSolver(parts, operator="ebe")
    try:
        its = []
        for step in (1, 2):
            for P in parts + ref:
                P["GlobData"]["TimeStepCount"] = step
            gs.updateBC(); gs.updatePreconditioner(); gs.PCG()
            out = pcg_oracle.solve_step(ref)
            gd = parts[0]["GlobData"]
            assert gd["TimeList_Flag"][step] == out["flag"] == 0 and abs(int(gd["TimeList_Iter"][step]) - out["iter"]) <= 1
            its.append(int(gd["TimeList_Iter"][step]))
            for P, R in zip(parts, ref):
                assert relerr(P["Un"], R["Un"]) < 1e-8
        assert its[1] < its[0]                                     # the warm start pays off
    finally:
        gs.close()


def test_group_load_step_driver_writes_the_n_rank_files(hostops, tmp_path):
    """run_load_steps_group (`python -m pcg_mi355x.run --group`): partition files of 8 parts read like 8 ranks would, solved by
    ONE process, result files in the layout 8 ranks write (file_operations.py:348-375: one segment + one metadata row per part)."""
    from pcg_mi355x import io as pio, run as prun
    _, parts = golden_cases.build_case("n9_p8")
    g = golden("n9_p8")
    prefix = str(tmp_path / "part" / "MeshPart_")
    pio.write_partition(prefix, parts)
    gds = [prun.init_glob_data() for _ in range(8)]
    rd = [pio.read_partition(prefix, 8, k, gds[k]) for k in range(8)]
    settings = {"TimeHistoryParam": {"ExportFlag": True, "ExportFrmRate": 1, "ExportFrms": [], "PlotFlag": False,
                                     "TimeStepDelta": [0, 1], "ExportVars": "U"},
                "SolverParam": {"Tol": 1e-7, "MaxIter": 10000}}
    for gd in gds:
        prun.apply_settings(gd, settings)
    res = str(tmp_path / "Results_Run1" / "ResVecData") + os.sep
    flag, relres, it, gs = prun.run_load_steps_group(rd, res, operator="ebe")
    gs.close()
    assert flag[1] == int(g["flag"]) and abs(int(it[1]) - int(g["iter"])) <= 1
    dof = pio.read_result_vector(res + "Dof")
    u1 = pio.read_result_vector(res + "U_1")
    assert len(np.unique(dof)) == len(dof) == len(g["Un"]) and np.all(pio.read_result_vector(res + "U_0") == 0)
    assert relerr(u1, g["Un"][dof]) < 1e-8
    meta = np.load(res + "U_1_metadat.npy", allow_pickle=True).item()
    owned = [int(np.asarray(P["DofWeightVector"]).astype(bool).sum()) for P in rd]
    assert list(meta["NfData"]) == owned and list(meta["OffsetData"]) == list(np.cumsum([0] + owned[:-1]) * 8)
    assert list(np.load(res + "Time_T.npy")) == [0.0, 1.0]
    rec = gds[3]["MP_TimeRecData"]
    assert rec["dT_Calc"] > 0 and rec["dT_CommWait"] >= 0


def test_run_cli_in_group_mode(hostops, tmp_path, capsys):
    """`python -m pcg_mi355x.run --group --n-parts 3 --partition-prefix ...` (here in-process on the test double): the files
    the reference's solver stage leaves behind - ResVecData/U_<k>, Dof, NodeId, Time_T and PlotData/TimeData with the
    calc / comm-wait split averaged over the parts (file_operations.py:101-109)."""
    from pcg_mi355x import io as pio, run as prun
    _, parts = golden_cases.build_case("oct_p3")
    g = golden("oct_p3")
    prefix = str(tmp_path / "MPI" / "")
    pio.write_partition(prefix, parts)
    results = str(tmp_path / "Results_Run1")
    prun.main(["--group", "--n-parts", "3", "--partition-prefix", prefix, "--results", results, "--operator", "dict"])
    out = capsys.readouterr().out
    assert ">calculation time:" in out and ">communication time:" in out
    td = np.load(os.path.join(results, "PlotData", "TimeData.npz"))
    assert int(td["Flag"][1]) == int(g["flag"]) and abs(int(td["Iter"][1]) - int(g["iter"])) <= 1
    dof = pio.read_result_vector(os.path.join(results, "ResVecData", "Dof"))
    u1 = pio.read_result_vector(os.path.join(results, "ResVecData", "U_1"))
    assert len(np.unique(dof)) == len(dof) == len(g["Un"]) and relerr(u1, g["Un"][dof]) < 1e-7
    assert float(td["CalcTime"]) > 0 and float(td["TotalTime"]) > 0
    with pytest.raises(SystemExit, match="n-parts"):
        prun.main(["--group", "--partition-prefix", prefix, "--results", results])


def test_group_argument_errors(hostops):
    from pcg_mi355x.group import DeviceGroup
    from pcg_mi355x.operator import from_refmeshpart
    L = hostops.lib()
    h = C.c_void_p()
    assert L.pcg_group_create(0, (C.c_int32 * 1)(0), C.byref(h)) != 0 and b"bad argument" in L.pcg_last_error()
    assert L.pcg_group_create(2, (C.c_int32 * 2)(0, -1), C.byref(h)) != 0 and b"out of range" in L.pcg_last_error()
    _, parts = golden_cases.build_case("n9_p2")
    g = DeviceGroup([0, 1])                                       # the test double has no device table: any id >= 0 is accepted
    try:
        assert L.pcg_group_size(g._h) == 2 and L.pcg_group_device(g._h, 1) == 1 and L.pcg_group_device(g._h, 2) == -1
        assert L.pcg_comm_rank(L.pcg_group_comm(g._h, 1)) == 1 and L.pcg_comm_size(L.pcg_group_comm(g._h, 0)) == 2
        with pytest.raises(hostops.PcgError, match="every member needs an operator"):
            g.build_jacobi()
        x = np.zeros(3)
        assert L.pcg_group_apply(g._h, (C.c_void_p * 2)(x.ctypes.data, x.ctypes.data), (C.c_void_p * 2)(x.ctypes.data, x.ctypes.data)) != 0
        assert b"member 0 has no engine" in L.pcg_last_error()
        with pytest.raises(hostops.PcgError, match="communicator lives on device 1, the engine on device 0"):
            from_refmeshpart(parts[1], device=0, comm=g.comms[1])     # pcg_set_comm_native checks the device
        op = from_refmeshpart(parts[1], device=0, comm=g.comms[0])     # created on device 0, member 1 lives on device 1
        try:
            with pytest.raises(hostops.PcgError, match="was created on device 0"):
                g.attach(1, op)
            assert L.pcg_group_attach(g._h, 5, op._h) != 0 and b"bad member" in L.pcg_last_error()
        finally:
            op.close()
    finally:
        g.close()
    with pytest.raises(ValueError, match="complete list"):
        DeviceGroup.from_refmeshparts(parts[:1] + parts[:1])


def test_a_failing_member_is_named(hostops):
    """Errors of the members' calls come back as ONE message that names every failing member and its device."""
    from pcg_mi355x.group import DeviceGroup
    _, parts = golden_cases.build_case("n9_p2")
    g = DeviceGroup.from_refmeshparts(parts)
    try:
        L = hostops.lib()
        b = [np.ones(op.n) for op in g.ops]
        xs = [np.empty(op.n) for op in g.ops]
        ptr = lambda arrs: (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])          # noqa: E731
        rc = L.pcg_group_solve(g._h, ptr(b), None, None, 1e-7, 0, 100, ptr(xs), None, 0, None)     # max_iter 0: refused before any collective
        msg = L.pcg_last_error()
        assert rc != 0 and b"member 0 (device 0)" in msg and b"member 1 (device 0)" in msg and b"max_iter" in msg
    finally:
        g.close()


# ---- GPU ----------------------------------------------------------------------------------------------------------------
def _gpu_env(fake):
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")
    env.pop("PCG_RCCL_LIB", None)
    if fake:
        import conftest
        env["PCG_RCCL_LIB"] = conftest.build_fakenccl()
    return env


@pytest.mark.gpu
@pytest.mark.parametrize("cases", ["n9_p2,n9_p8", "oct_p3,n13_t3_p4_ud,n9_p2_flag4"])
def test_group_on_one_gpu(gpu_lib, tmp_path, cases):
    """The HIP engine under the group: every member on device 0, talking through the RCCL stand-in (real RCCL refuses two
    ranks on one device).  Library threads, comm streams, fences, both operators, against the reference fixtures."""
    from test_native_comm import _check
    r = subprocess.run([sys.executable, WORKER, "group", cases, "sell,ebe", str(tmp_path)], env=_gpu_env(True),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for case in cases.split(","):
        world = len([f for f in os.listdir(tmp_path) if f.startswith(case + "_sell_rank")])
        for kind in ("sell", "ebe"):
            _check(case, kind, tmp_path, world)


@pytest.mark.gpu
def test_group_of_one_on_real_rccl(gpu_lib, tmp_path):
    """librccl itself under the group API (the box has one GPU: a group of one): in-process unique id, ncclCommInitRank on the
    member's thread, the all-reduces of the look-ahead loop issued from that thread."""
    from test_native_comm import _check
    r = subprocess.run([sys.executable, WORKER, "group", "n9_p1", "sell,ebe", str(tmp_path)], env=_gpu_env(False),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for kind in ("sell", "ebe"):
        _check("n9_p1", kind, tmp_path, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("case,world", [("n9_p2", 2), ("n13_t3_p4_ud", 4), ("n9_p8", 8)])
def test_group_across_gpus_on_real_rccl(gpu_lib, tmp_path, case, world):
    """One process, `world` GPUs, RCCL over xGMI between them (auto-skipped on a smaller box)."""
    from test_native_comm import _check
    if gpu_lib.lib().pcg_device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    env = _gpu_env(False)
    env["PCG_TEST_GROUP_DEVICES"] = ",".join(str(k) for k in range(world))
    r = subprocess.run([sys.executable, WORKER, "group", case, "sell,ebe", str(tmp_path)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for kind in ("sell", "ebe"):
        _check(case, kind, tmp_path, world)
