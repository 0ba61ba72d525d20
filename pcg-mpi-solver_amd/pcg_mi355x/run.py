"""Load-step driver: the solver stage of the reference pipeline on the MI355X engine.

Restates the `__main__` block of src/solver/pcg_solver.py (:965-1031): read this rank's part of the
partition (:980), read the solver settings (:981, GlobSettings.zpkl written as in
examples/run_basic_script.bash:30-49), then for every load step updateBC -> updatePreconditioner ->
PCG -> export (:1002-1008), and store Flag / RelRes / Iter and the calc/comm time split (:1020).
One process per GPU (`python -m torch.distributed.run --nproc-per-node N -m pcg_mi355x.run ...`)
replaces `mpiexec -np N python3 src/solver/pcg_solver.py <Run> <SpeedTestFlag>`.

    python -m pcg_mi355x.run --partition-prefix <PyDataPath_Part> --n-parts N \\
           --settings __pycache__/GlobSettings.zpkl --results <ScratchPath>/Results_Run1 [--operator ebe]
    python -m pcg_mi355x.run --mdf <Scratch>/ModelData/MDF/ ...     (no partition files: each rank partitions in memory)
    python -m pcg_mi355x.run --group --n-parts N ...                (ONE process drives the N GPUs: device group)
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np

from . import solver
from .io import importz, read_partition, ResultExporter

__all__ = ["init_glob_data", "apply_settings", "run_load_steps", "run_load_steps_group", "main"]


def init_glob_data():
    """The timer / bookkeeping entries initGlobData creates (pcg_solver.py:45-52)."""
    return {"MP_TimeRecData": {"dT_FileRead": 0.0, "dT_Calc": 0.0, "dT_CommWait": 0.0, "dT_CalcList": [],
                               "dT_CommWaitList": [], "TimeStepCountList": [], "t0": time.time()}}


def apply_settings(glob_data, settings, speed_test=False):
    """readGlobalSettings (pcg_solver.py:113-139)."""
    th, sp = settings["TimeHistoryParam"], settings["SolverParam"]
    glob_data["FintCalcMode"] = "outbin"
    glob_data["ExportVars"] = th["ExportVars"]
    glob_data["PlotFlag"] = th["PlotFlag"]
    glob_data["ExportFlag"] = th["ExportFlag"]
    glob_data["ExportKeyFrm"] = th["ExportFrmRate"]
    glob_data["ExportFrms"] = th["ExportFrms"]
    glob_data["TimeStepDelta"] = th["TimeStepDelta"]
    glob_data["RefMaxTimeStepCount"] = len(th["TimeStepDelta"])
    glob_data["MaxIter"] = sp["MaxIter"]
    glob_data["Tol"] = sp["Tol"]
    if speed_test:                                                     # :138-139
        glob_data["PlotFlag"] = 0
        glob_data["ExportFlag"] = 0


def run_load_steps(part, res_vec_path=None, comm=None):
    """pcg_solver.py:996-1008 for one part.  Returns rank-0 style lists (Flag, RelRes, Iter per step)."""
    gd = part["GlobData"]
    n_steps = int(gd.get("RefMaxTimeStepCount", len(gd["TimeStepDelta"])))
    part["Un"] = np.zeros(part["NDOF"])                               # :996 (1e-200*rand there: numerically zero)
    part["DofWeightVector_Eff"] = np.asarray(part["DofWeightVector"])[np.asarray(part["LocDofEff"], np.int64)]   # :997
    gd["TimeList_Flag"] = np.zeros(n_steps)                           # initExportData :162-165
    gd["TimeList_RelRes"] = np.zeros(n_steps)
    gd["TimeList_Iter"] = np.zeros(n_steps)
    dt = gd.get("dt", 1.0)
    time_list = [i * dt for i in range(n_steps)]                      # :167-168
    gd["TimeStepCount"] = 0
    exporter = None
    key_frm = int(gd.get("ExportKeyFrm", 0) or 0)
    frames = np.array(gd.get("ExportFrms", []), dtype=int)
    frames = frames[0] - 1 if len(frames) > 0 else frames             # :155-157
    if gd.get("ExportFlag") and res_vec_path and "U" in str(gd.get("ExportVars", "U")):
        exporter = ResultExporter(part, res_vec_path, comm)
        exporter.export(time_list[0])                                  # initial frame (:209)
    for step in range(1, n_steps):                                     # :1002
        gd["TimeStepCount"] = step                                     # updateTimeStep :213-216
        solver.update_bc(part)                                         # :1004
        solver.update_preconditioner(part)                             # :1005
        solver.solve(part)                                             # :1006
        if exporter is not None:                                       # exportContourData :853-858
            now = (key_frm > 0 and step % key_frm == 0) or (step in np.atleast_1d(frames))
            if now:
                exporter.export(time_list[step])
    return gd["TimeList_Flag"], gd["TimeList_RelRes"], gd["TimeList_Iter"]


def run_load_steps_group(parts, res_vec_path=None, devices=None, operator="sell"):
    """pcg_solver.py:996-1008 for ALL parts of the model in ONE process (device group, pcg_mi355x.group: member k = part k
    on devices[k]); same per-step calls, same result files as N ranks.  -> (Flag, RelRes, Iter per step, GroupSolver)."""
    from .group import GroupSolver
    from .io import GroupResultExporter
    parts = sorted(parts, key=lambda p: int(p["Id"]))
    gd = parts[0]["GlobData"]
    n_steps = int(gd.get("RefMaxTimeStepCount", len(gd["TimeStepDelta"])))
    for P in parts:
        g = P["GlobData"]
        P["Un"] = np.zeros(P["NDOF"])                                                                  # :996
        P["DofWeightVector_Eff"] = np.asarray(P["DofWeightVector"])[np.asarray(P["LocDofEff"], np.int64)]   # :997
        for k in ("TimeList_Flag", "TimeList_RelRes", "TimeList_Iter"):                               # :162-165
            g[k] = np.zeros(n_steps)
        g["TimeStepCount"] = 0
    dt = gd.get("dt", 1.0)
    time_list = [i * dt for i in range(n_steps)]
    key_frm = int(gd.get("ExportKeyFrm", 0) or 0)
    frames = np.array(gd.get("ExportFrms", []), dtype=int)
    frames = frames[0] - 1 if len(frames) > 0 else frames
    gs = GroupSolver(parts, devices=devices, operator=operator, timing=True)   # the reference always keeps its calc / comm split
    exporter = None
    if gd.get("ExportFlag") and res_vec_path and "U" in str(gd.get("ExportVars", "U")):
        exporter = GroupResultExporter(parts, res_vec_path)
        exporter.export(time_list[0])
    for step in range(1, n_steps):                                                                    # :1002
        for P in parts:
            P["GlobData"]["TimeStepCount"] = step
        gs.updateBC()                                                                                 # :1004
        gs.updatePreconditioner()                                                                     # :1005
        gs.PCG()                                                                                      # :1006
        if exporter is not None and ((key_frm > 0 and step % key_frm == 0) or (step in np.atleast_1d(frames))):
            exporter.export(time_list[step])
    return gd["TimeList_Flag"], gd["TimeList_RelRes"], gd["TimeList_Iter"], gs


def _main_group(args, n_parts):
    """--group: this ONE process drives n_parts GPUs (member k on device k modulo the visible devices)."""
    from . import _lib
    t0 = time.time()
    gds = [init_glob_data() for _ in range(n_parts)]
    if args.mdf is not None:
        from . import mdf as mdf_mod
        from .partition import partition_model, geometric_partition
        model = mdf_mod.read_mdf(args.mdf)
        try:
            ele_part = mdf_mod.read_mesh_part(args.mdf, n_parts)
        except FileNotFoundError:
            ele_part = geometric_partition(model, n_parts)
        parts = partition_model(model, ele_part, device=0 if _lib.backend_name() == "hip-gfx950" else None)   # index passes on the GPU
        for P, gd in zip(parts, gds):
            gd.update(P["GlobData"])
            P["GlobData"] = gd
        del model
    else:
        parts = [read_partition(args.partition_prefix, n_parts, k, gds[k]) for k in range(n_parts)]
    settings = importz(args.settings) if args.settings else {
        "TimeHistoryParam": {"ExportFlag": True, "ExportFrmRate": 1, "ExportFrms": [], "PlotFlag": False,
                             "TimeStepDelta": [0, 1], "ExportVars": "U"},
        "SolverParam": {"Tol": args.tol, "MaxIter": args.max_iter}}
    for gd in gds:
        apply_settings(gd, settings, args.speed_test)
        gd["MP_TimeRecData"]["dT_FileRead"] += time.time() - t0
    if os.path.exists(args.results) and os.listdir(args.results):
        from datetime import datetime
        os.rename(args.results.rstrip(os.sep), args.results.rstrip(os.sep) + "_" + datetime.now().strftime("%d%m%Y_%H%M%S"))
    from .group import default_devices
    t_start = time.time()
    flag, relres, it, gs = run_load_steps_group(parts, os.path.join(args.results, "ResVecData") + os.sep,
                                                default_devices(n_parts), args.operator)
    total = time.time() - t_start
    gs.close()
    os.makedirs(os.path.join(args.results, "PlotData"), exist_ok=True)
    recs = [gd["MP_TimeRecData"] for gd in gds]                       # rank 0 reports the mean over ranks (file_operations.py:101-109)
    calc, wait = float(np.mean([r["dT_Calc"] for r in recs])), float(np.mean([r["dT_CommWait"] for r in recs]))
    np.savez_compressed(os.path.join(args.results, "PlotData", "TimeData"), Flag=flag, RelRes=relres, Iter=it,
                        FileReadTime=recs[0]["dT_FileRead"], CalcTime=calc, CommWaitTime=wait, TotalTime=total)
    print(f">file read time:     {recs[0]['dT_FileRead']:.1f} sec\n>calculation time:   {calc:.1f} sec\n"
          f">communication time: {wait:.1f} sec\n>total runtime:      {total:.1f} sec\n"
          f">flag {flag[1:]}, iterations {it[1:]}, relres {relres[1:]}")


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--partition-prefix", default=None, help="PyDataPath_Part: files <prefix><N>_<id>.mpidat")
    ap.add_argument("--mdf", default=None, help="instead of partition files: an MDF directory (with MeshPart_<N>.npy or none "
                                                "-> coordinate bisection); every rank builds ONLY its own part in memory")
    ap.add_argument("--n-parts", type=int, default=None, help="defaults to WORLD_SIZE")
    ap.add_argument("--settings", default=None, help="GlobSettings.zpkl (examples/run_basic_script.bash:30-49); default: "
                                                     "one load step, --tol, --max-iter, export U")
    ap.add_argument("--tol", type=float, default=1e-7)
    ap.add_argument("--max-iter", type=int, default=10000)
    ap.add_argument("--results", required=True, help="result directory (Results_Run<R>)")
    ap.add_argument("--operator", choices=["sell", "dict", "ebe"], default="sell")
    ap.add_argument("--comm", choices=["native", "torch"], default="native",
                    help="N > 1: native = RCCL calls issued by the engine (default); torch = torch.distributed callbacks")
    ap.add_argument("--speed-test", action="store_true")
    ap.add_argument("--engine-side", action="store_true",
                    help="N > 1, native communicator (round 5, opt-in): MPI_SUM through peer-mapped mailboxes inside the engine's own launches "
                         "(pcg_comm_enable_mailbox) and the interface exchange of the iteration as stores into the neighbours' mapped "
                         "receive buffers (pcg_enable_direct_exchange; matrix-free engines without the interface-first launch) - no collective "
                         "kernel in the PCG loop.  Falls back to RCCL, on every rank together, where a peer cannot be mapped")
    ap.add_argument("--group", action="store_true",
                    help="ONE process drives all --n-parts GPUs (device group, C ABI pcg_group_*) instead of one process per GPU")
    args = ap.parse_args(argv)
    if args.group:
        if (args.partition_prefix is None) == (args.mdf is None):
            raise SystemExit("give exactly one of --partition-prefix / --mdf")
        if not args.n_parts or int(os.environ.get("WORLD_SIZE", "1")) != 1:
            raise SystemExit("--group: one process, give --n-parts")
        return _main_group(args, args.n_parts)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_parts = args.n_parts or world
    if n_parts != world:
        raise SystemExit("one process (GPU) per mesh part: launch with --nproc-per-node <n_parts> (pcg_solver.py:91)")
    share = os.environ.get("PCG_RUN_SHARE_GPU") == "1"      # dry run on a 1-GPU box: all ranks on device 0 (needs PCG_RCCL_LIB)
    dev = 0 if share else local_rank
    torch.cuda.set_device(dev)
    comm = None
    if world > 1:
        # torch.distributed is the control plane only (unique-id broadcast, result-file offsets, barriers)
        if share:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        from .dist import RcclComm, TorchComm
        if args.comm == "native":
            comm = RcclComm.from_torch(dev)
            comm.set_timing(True)                          # the reference always keeps its calc / comm-wait split (:631-641)
            if args.engine_side:                           # collective: every rank runs these same lines
                got = comm.enable_mailbox(True)
                os.environ["PCG_DIRECT_EXCHANGE"] = "1"    # Operator.set_comm enables the direct exchange (and says why when it cannot)
                if rank == 0:
                    print(f">engine-side all-reduce: {'on' if got else 'declined - ' + str(comm.mailbox_reason)}; engine-side exchange: requested")
        else:
            comm = TorchComm(device=torch.device("cuda", dev))
    solver.configure(comm=comm, device=dev, operator=args.operator)

    gd = init_glob_data()
    t0 = time.time()
    if (args.partition_prefix is None) == (args.mdf is None):
        raise SystemExit("give exactly one of --partition-prefix / --mdf")
    if args.mdf is not None:
        from . import mdf as mdf_mod
        from .partition import partition_model, geometric_partition
        model = mdf_mod.read_mdf(args.mdf)
        try:
            ele_part = mdf_mod.read_mesh_part(args.mdf, n_parts)
        except FileNotFoundError:
            ele_part = geometric_partition(model, n_parts)            # deterministic: every rank computes the same vector
        from . import _lib as _l
        part = partition_model(model, ele_part, only=[rank], device=dev if _l.backend_name() == "hip-gfx950" else None)[0]   # index passes on the GPU (csrc/part_setup.hip); the CPU test double has none
        gd.update(part["GlobData"])                                   # readModelData :107-108
        part["GlobData"] = gd
        del model
    else:
        part = read_partition(args.partition_prefix, n_parts, rank, gd)   # :980
    settings = importz(args.settings) if args.settings else {                # examples/run_basic_script.bash:34-44
        "TimeHistoryParam": {"ExportFlag": True, "ExportFrmRate": 1, "ExportFrms": [], "PlotFlag": False,
                             "TimeStepDelta": [0, 1], "ExportVars": "U"},
        "SolverParam": {"Tol": args.tol, "MaxIter": args.max_iter}}
    apply_settings(gd, settings, args.speed_test)                            # :981
    gd["MP_TimeRecData"]["dT_FileRead"] += time.time() - t0
    if rank == 0 and os.path.exists(args.results) and os.listdir(args.results):      # pcg_solver.py:67-72: never overwrite a run
        from datetime import datetime
        os.rename(args.results.rstrip(os.sep), args.results.rstrip(os.sep) + "_" + datetime.now().strftime("%d%m%Y_%H%M%S"))
    if world > 1:
        dist.barrier()
    res_vec = os.path.join(args.results, "ResVecData") + os.sep
    t_start = time.time()
    flag, relres, it = run_load_steps(part, res_vec, comm)
    total = time.time() - t_start
    if rank == 0:
        os.makedirs(os.path.join(args.results, "PlotData"), exist_ok=True)
        rec = gd["MP_TimeRecData"]
        np.savez_compressed(os.path.join(args.results, "PlotData", "TimeData"), Flag=flag, RelRes=relres, Iter=it,
                            FileReadTime=rec["dT_FileRead"], CalcTime=rec["dT_Calc"], CommWaitTime=rec["dT_CommWait"],
                            TotalTime=total)
        print(f">file read time:     {rec['dT_FileRead']:.1f} sec\n>calculation time:   {rec['dT_Calc']:.1f} sec\n"
              f">communication time: {rec['dT_CommWait']:.1f} sec\n>total runtime:      {total:.1f} sec\n"
              f">flag {flag[1:]}, iterations {it[1:]}, relres {relres[1:]}")
    op = part.pop("_pcg_mi355x_operator", None)
    if op is not None:
        if args.engine_side and rank == 0 and world > 1:
            print(f">engine-side exchange: {'on' if op.direct_exchange else 'declined - ' + str(op.direct_exchange_reason)}")
        op.close()
    if world > 1:
        dist.barrier()
        if hasattr(comm, "close"):
            comm.close()                     # ncclCommDestroy while the HIP runtime is still up
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
