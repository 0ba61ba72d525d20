"""Worker for the tests of the engine's NATIVE communicator (csrc/rccl_comm.hip) on the GPU.

usage: python tests/native_comm_worker.py threads <case[,case..]> <kind[,kind..]> <outdir>
           every part of a golden case in THIS process, one thread per part, each thread with its own RcclComm
       python tests/native_comm_worker.py proc <case> <kind> <outdir> <rank> <world> <idfile> [device]
           this process is rank <rank> of <world> (one part per process, as in production)
       python tests/native_comm_worker.py group <case[,case..]> <kind[,kind..]> <outdir>
           every part of a case as a member of ONE device group (pcg_group_*: the library's own thread per member)
       python tests/native_comm_worker.py bigbrick <N> <kind[,kind..]> <out.json> [iterations]
           BASELINE configs[3] at its own size: the N-node brick (150 -> 10 125 000 dof) split 2x2x2, eight engines on ONE GPU, one
           thread + one native communicator per part; prints / writes the deviations from the oracle and from the one-part engine
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
    op = from_refmeshpart(P, device=device, comm=comm, kind=kind)     # (PCG_EBE_ONE_PHASE=1 in the environment: no interface-first launch)
    try:
        if os.environ.get("PCG_TEST_DIRECT", "0") == "1":          # opt-in engine-side exchange (pcg_enable_direct_exchange), collective
            assert op.enable_direct_exchange(), f"direct exchange refused: {op.direct_exchange_reason}"
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


def big_brick(N, kinds, out_json, iters):
    """Multi-part parity at a BASELINE size: the native multi-part path - interface lists of 139 KB faces, the real
    n_bnd_slices, the PACK epilogue of the interface rows' launch, k_fixup's and k_vec<false>'s last-workgroup reductions at
    1.27 M dof per part - against pcg_oracle.calc_matvec / halo_sum (reference pcg_solver.py:242-336) on a probe vector, the
    assembled diagonal (:346-352, 'Preconditioner' mode), Fext, and `iters` iterations of residual history against ONE engine
    holding the whole system."""
    import json
    import time
    import pcg_oracle
    import pcg_mi355x as pm
    from pcg_mi355x.brick import Brick, make_parts, block_partition
    from pcg_mi355x.dist import RcclComm
    from pcg_mi355x.operator import from_refmeshpart
    t_start = time.time()
    b = Brick(N, seed=0)
    parts = make_parts(b, block_partition(b, 2, 2, 2), max_iter=iters)
    world = len(parts)
    probe = np.random.default_rng(7).standard_normal(b.n_dof)
    xs = [probe[P["DofVector"]] for P in parts]
    t0 = time.time()
    y_ref = pcg_oracle.calc_matvec(parts, xs, "Strain", use_c=True)                 # the reference's operator + interface sum
    d_ref = pcg_oracle.calc_matvec(parts, None, "Preconditioner")
    ref_parts = [dict(P) for P in parts]
    pcg_oracle.update_bc(ref_parts, use_c=True)
    t_oracle = time.time() - t0
    report = {"N": N, "dofs": int(b.n_dof), "parts": world, "dofs_per_part": [int(P["NDOF"]) for P in parts],
              "interface_dofs_per_part": [int(P["N_NbrDof"]) for P in parts], "iterations": iters, "oracle_s": t_oracle, "kinds": {}}

    def rel(a, c):
        nc = float(np.linalg.norm(c))
        return float(np.linalg.norm(a - c)) / (nc if nc > 0 else 1.0)      # (the lower parts of the brick carry no load: Fext == 0)
    uid = RcclComm.new_unique_id()
    comms = [None] * world
    for kind in kinds:
        outs, errs = [None] * world, [None] * world

        def run(r):
            try:
                if comms[r] is None:
                    comms[r] = RcclComm(r, world, 0, uid)
                P = parts[r]
                op = from_refmeshpart(P, device=0, comm=comms[r], kind=kind)
                try:
                    o = {"y": op.apply(xs[r]), "diag": op.diag()}
                    fext, udi = op.update_bc(P["RefLoadVector"], P["Ud"], 1.0)
                    inv = op.build_jacobi()
                    gd = P["GlobData"]
                    x, res, hist = op.solve(fext, P["Un"], inv, gd["Tol"], gd["MaxIter"], gd["GlobNDofEff"], history=True)
                    info = pm.solver.SolveInfo(res, hist)
                    o.update(fext=fext, x=x, hist=info.history, flag=info.flag, iter=info.iter, relres=info.relres, iters_done=info.iters_done)
                    o["info"] = op.matrix_info() if kind != "ebe" else op.operator_info()
                finally:
                    op.close()
                outs[r] = o
            except BaseException as e:      # noqa: BLE001
                errs[r] = e
        t0 = time.time()
        ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        for e in errs:
            if e is not None:
                raise e
        t_multi = time.time() - t0
        # ---- the same system as ONE engine ------------------------------------------------------------------------------------
        t0 = time.time()
        one = make_parts(b, max_iter=iters)[0]
        op1 = from_refmeshpart(one, device=0, kind=kind)
        fext1, _ = op1.update_bc(one["RefLoadVector"], one["Ud"], 1.0)
        inv1 = op1.build_jacobi()
        gd = one["GlobData"]
        x1, res1, hist1 = op1.solve(fext1, one["Un"], inv1, gd["Tol"], gd["MaxIter"], gd["GlobNDofEff"], history=True)
        i1 = pm.solver.SolveInfo(res1, hist1)
        op1.close()
        t_one = time.time() - t0
        h1 = np.asarray(i1.history, float).reshape(-1, 3)
        rec = {"multi_part_s": t_multi, "one_part_s": t_one, "y": [], "diag": [], "fext": [], "x_vs_one_part": [], "flag": [], "iters_done": [],
               "relres": [], "hist": []}
        for r, (P, o) in enumerate(zip(parts, outs)):
            hm = np.asarray(o["hist"], float).reshape(-1, 3)
            k = min(len(hm), len(h1), iters)
            rec["y"].append(rel(o["y"], y_ref[r]))
            rec["diag"].append(rel(o["diag"], d_ref[r]))
            rec["fext"].append(rel(o["fext"], ref_parts[r]["Fext"]))
            rec["x_vs_one_part"].append(rel(o["x"], x1[P["DofVector"]]))
            rec["flag"].append(int(o["flag"])); rec["iters_done"].append(int(o["iters_done"])); rec["relres"].append(float(o["relres"]))
            rec["hist"].append(float(np.max(np.abs(hm[:k] - h1[:k]) / np.maximum(np.abs(h1[:k]), 1e-300))))
            rec["hist_rows"] = int(k)
        rec["hist_identical_across_ranks"] = bool(all(np.array_equal(np.asarray(o["hist"]), np.asarray(outs[0]["hist"])) for o in outs))
        rec["one_part"] = {"flag": i1.flag, "iters_done": i1.iters_done, "relres": i1.relres}
        rec["part0_info"] = {k: (int(v) if isinstance(v, (int, np.integer)) else v) for k, v in outs[0]["info"].items() if np.isscalar(v)}
        report["kinds"][kind] = rec
        print(f"[bigbrick {kind}] y {max(rec['y']):.2e} diag {max(rec['diag']):.2e} fext {max(rec['fext']):.2e} hist {max(rec['hist']):.2e} "
              f"x {max(rec['x_vs_one_part']):.2e} ({t_multi:.0f} s + {t_one:.0f} s)", file=sys.stderr, flush=True)
    for c in comms:
        if c is not None:
            c.close()
    report["total_s"] = time.time() - t_start
    with open(out_json, "w") as f:
        json.dump(report, f)
    print(json.dumps(report))


def main():
    mode = sys.argv[1]
    from pcg_mi355x import _lib
    from pcg_mi355x.dist import RcclComm
    _lib.use_library(os.environ.get("PCG_TEST_LIB") or None)      # (PCG_TEST_LIB: the CPU double, for the harness check of the no-GPU tier)
    timing = os.environ.get("PCG_TEST_COMM_TIMING", "1") == "1"
    mailbox = os.environ.get("PCG_TEST_MAILBOX", "0") == "1"       # opt-in engine-side reduction (pcg_comm_enable_mailbox), collective

    refusal_ok = os.environ.get("PCG_TEST_MAILBOX_REFUSAL_OK", "0") == "1"     # ranks of ONE process on ONE device: the engine must decline

    def with_mailbox(comm):
        if mailbox and not comm.mailbox:
            got = comm.enable_mailbox()
            if refusal_ok:
                assert not got and "share a device" in (comm.mailbox_reason or ""), (got, comm.mailbox_reason)
                if comm.rank == 0: print("MAILBOX REFUSED:", comm.mailbox_reason, flush=True)
            else:
                assert got, f"mailbox all-reduce refused: {comm.mailbox_reason}"
        return comm
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
                            comms[r] = with_mailbox(RcclComm(r, world, 0, uid))          # collective: all threads are in here together
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
    elif mode == "bigbrick":
        big_brick(int(sys.argv[2]), sys.argv[3].split(","), sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else 30)
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
                    if mailbox:
                        got = gs.group.enable_mailbox()
                        if refusal_ok and len(set(devs)) < len(devs):
                            assert not got, "members of one process on one device must decline the mailbox all-reduce"
                            assert not gs.group.enable_direct_exchange(), "... and the direct exchange (pcg_group_enable_direct_exchange)"
                            print("MAILBOX REFUSED: device group with members on one device", flush=True)
                        else:
                            assert got, "mailbox all-reduce refused in the device group"
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
        comm = with_mailbox(RcclComm.from_file(rank, world, device, idfile, launch_id=os.path.basename(idfile)))    # (a fresh file name per test launch)
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
