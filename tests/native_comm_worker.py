"""Worker for the tests of the engine's NATIVE communicator (csrc/rccl_comm.hip) on the GPU.

usage: python tests/native_comm_worker.py threads <case[,case..]> <kind[,kind..]> <outdir>
           every part of a golden case in THIS process, one thread per part, each thread with its own RcclComm
       python tests/native_comm_worker.py proc <case> <kind> <outdir> <rank> <world> <idfile> [device]
           this process is rank <rank> of <world> (one part per process, as in production)
       python tests/native_comm_worker.py group <case[,case..]> <kind[,kind..]> <outdir>
           every part of a case as a member of ONE device group (pcg_group_*: the library's own thread per member)
The RCCL library is whatever csrc/rccl_comm.hip resolves: the real librccl (one rank per GPU), or - with
PCG_RCCL_LIB=tests/fakenccl/_build/libfakenccl.so - the shared-GPU test double.  Results go to
<outdir>/<case>_<kind>_rank<r>.npz in the layout of tests/dist_worker.py.
"""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pcg-mpi-solver_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np


def run_rank(P, x_probe, comm, kind, device, timing):
    import pcg_mi355x as pm
    from pcg_mi355x.operator import from_refmeshpart
    out = {"rank": comm.rank, "dofs": P["DofVector"]}
    op = from_refmeshpart(P, device=device, comm=comm, kind=kind)
    try:
        out["y_probe"] = op.apply(x_probe)
        out["diag"] = op.diag()
        fext, udi = op.update_bc(P["RefLoadVector"], P["Ud"], 1.0)
        inv = op.build_jacobi()
        out["Fext"] = fext
        gd = P["GlobData"]
        s0 = comm.stats()
        if timing:
            comm.set_timing(True)
        x, res, hist = op.solve(fext, P["Un"], inv, gd["Tol"], gd["MaxIter"], gd["GlobNDofEff"], history=True)
        s1 = comm.stats()
        comm.set_timing(False)
        info = pm.solver.SolveInfo(res, hist)
        out["Un"] = x + udi
        out["history"] = info.history
        out["flag"], out["iter"], out["relres"], out["status"] = info.flag, info.iter, info.relres, info.status
        out["iters_done"], out["iters_enqueued"] = info.iters_done, info.iters_enqueued
        out["t_comm"], out["t_total"] = info.t_comm_s, info.t_total_s
        for k in s1:
            out["stat_" + k] = s1[k] - s0[k]
    finally:
        op.close()
    return out


def build(case):
    import golden_cases
    mesh, parts = golden_cases.build_case(case, os.path.join(ROOT, "tests", "golden"))
    probe = golden_cases.probe_for(mesh, parts)
    return parts, probe


def main():
    mode = sys.argv[1]
    from pcg_mi355x import _lib
    from pcg_mi355x.dist import RcclComm
    _lib.use_library(None)
    timing = os.environ.get("PCG_TEST_COMM_TIMING", "1") == "1"
    if mode == "threads":
        cases, kinds, outdir = sys.argv[2].split(","), sys.argv[3].split(","), sys.argv[4]
        for case in cases:
            parts, probe = build(case)
            world = len(parts)
            uid = RcclComm.new_unique_id()
            comms = [None] * world
            for kind in kinds:
                outs, errs = [None] * world, [None] * world

                def run(r):
                    try:
                        if comms[r] is None:
                            comms[r] = RcclComm(r, world, 0, uid)          # collective: all threads are in here together
                        P = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in parts[r].items()}
                        outs[r] = run_rank(P, probe[P["DofVector"]], comms[r], kind, 0, timing)
                    except BaseException as e:      # noqa: BLE001
                        errs[r] = e
                ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
                for e in errs:
                    if e is not None:
                        raise e
                for r, o in enumerate(outs):
                    np.savez(os.path.join(outdir, f"{case}_{kind}_rank{r}.npz"), **o)
            for c in comms:
                c.close()
    elif mode == "group":
        # ONE process, every part of the case as a member of a device group (pcg_group_*): the library's own threads
        # drive the members; devices from PCG_TEST_GROUP_DEVICES (default: every member on device 0)
        from pcg_mi355x.group import GroupSolver
        import pcg_mi355x as pm
        cases, kinds, outdir = sys.argv[2].split(","), sys.argv[3].split(","), sys.argv[4]
        for case in cases:
            for kind in kinds:
                parts, probe = build(case)
                world = len(parts)
                devs = os.environ.get("PCG_TEST_GROUP_DEVICES")
                devs = [int(d) for d in devs.split(",")] if devs else [0] * world
                gs = GroupSolver(parts, devices=devs, operator=kind, timing=timing)
                try:
                    ys = gs.group.apply([probe[P["DofVector"]] for P in parts])
                    ds = gs.group.diag()
                    s0 = [c.stats() for c in gs.group.comms]
                    gs.updateBC(); gs.updatePreconditioner()
                    assert gs.PCG(history=True) is None
                    s1 = [c.stats() for c in gs.group.comms]
                    for r, P in enumerate(parts):
                        info = P["_pcg_mi355x_info"]
                        o = {"rank": r, "dofs": P["DofVector"], "y_probe": ys[r], "diag": ds[r], "Fext": P["Fext"], "Un": P["Un"],
                             "history": info.history, "flag": info.flag, "iter": info.iter, "relres": info.relres, "status": info.status,
                             "iters_done": info.iters_done, "iters_enqueued": info.iters_enqueued, "t_comm": info.t_comm_s,
                             "t_total": info.t_total_s}
                        for k in s1[r]:
                            o["stat_" + k] = s1[r][k] - s0[r][k]
                        np.savez(os.path.join(outdir, f"{case}_{kind}_rank{r}.npz"), **o)
                finally:
                    gs.close()
    elif mode == "proc":
        case, kind, outdir = sys.argv[2:5]
        rank, world, idfile = int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
        device = int(sys.argv[8]) if len(sys.argv) > 8 else 0
        parts, probe = build(case)
        assert len(parts) == world
        comm = RcclComm.from_file(rank, world, device, idfile, launch_id=os.path.basename(idfile))    # (a fresh file name per test launch)
        P = parts[rank]
        o = run_rank(P, probe[P["DofVector"]], comm, kind, device, timing)
        np.savez(os.path.join(outdir, f"{case}_{kind}_rank{rank}.npz"), **o)
        comm.close()
    elif mode == "torchpg":
        # the launch shape of bench.py / pcg_mi355x.run at N > 1, at world size 1: torch.distributed with the NCCL (= RCCL)
        # backend as control plane AND the engine's own communicators on the same librccl in the same process
        case, kind, outdir, port = sys.argv[2:6]
        import datetime
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=120))
        dist.barrier()
        comm = RcclComm.from_torch(0)                         # unique id through broadcast_object_list on the NCCL group
        parts, probe = build(case)
        P = parts[0]
        o = run_rank(P, probe[P["DofVector"]], comm, kind, 0, timing)
        torch.cuda.synchronize()
        dist.barrier()                                         # torch's communicator still works next to the engine's two
        box = [None]
        dist.all_gather_object(box, float(o["relres"]))
        assert box[0] == float(o["relres"])
        np.savez(os.path.join(outdir, f"{case}_{kind}_rank0.npz"), **o)
        comm.close()
        dist.destroy_process_group()
    elif mode == "selfloop":
        # ONE rank on real librccl whose part lists ITSELF as its only neighbour (PCG_RCCL_ALLOW_SELF=1): the interface
        # exchange then delivers the part's own partial sums back to it, so y = A_local x with the interface dofs doubled.
        case, kind, outdir = sys.argv[2:5]
        parts, probe = build(case)
        P = parts[0]
        assert len(P["NbrMPIdVector"]) == 1
        P["NbrMPIdVector"] = [0]
        P["Id"] = 0
        comm = RcclComm.from_file(0, 1, 0, os.path.join(outdir, "id_self_" + kind))
        from pcg_mi355x.operator import from_refmeshpart
        x = probe[P["DofVector"]]
        op = from_refmeshpart(P, comm=comm, kind=kind)
        y = op.apply(x)
        d = op.diag()
        st = comm.stats()
        op.close()
        Q = {k: v for k, v in P.items()}
        Q["NbrMPIdVector"], Q["OvrlpLocalDofVecList"], Q["OvrlpLocalNodeIdVecList"] = [], [], []
        lop = from_refmeshpart(Q, kind=kind)                      # the same part without any exchange
        y0, d0 = lop.apply(x), lop.diag()
        lop.close()
        np.savez(os.path.join(outdir, f"selfloop_{kind}.npz"), y=y, y0=y0, d=d, d0=d0, ovl=np.asarray(P["OvrlpLocalDofVecList"][0]),
                 n_halo=st["n_halo"])
        comm.close()
    else:
        raise SystemExit("mode must be threads, group, proc, torchpg or selfloop")


if __name__ == "__main__":
    main()
