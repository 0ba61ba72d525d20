"""Worker for the multi-process tests (one process per part, as in production: one process per GPU).

usage: python -m torch.distributed.run --nproc-per-node P tests/dist_worker.py <case> <backend> <lib> <outdir>
  backend gloo + lib hostops : CPU test of the N>1 path (control flow, interface lists, comm hooks)
  backend nccl + lib product : the same on GPUs (world_size 1 on the 1-GPU test box)
Each rank builds ITS part of the golden case, runs updateBC -> updatePreconditioner -> PCG through the
drop-in functions and writes its results to <outdir>/rank<r>.npz.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pcg-mpi-solver_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist


def main():
    case, backend, libkind, outdir = sys.argv[1:5]
    opkind = sys.argv[5] if len(sys.argv) > 5 else "sell"
    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    from pcg_mi355x import _lib
    if libkind == "hostops":
        import conftest
        _lib.use_library(conftest.HOSTOPS_LIB)
    else:
        _lib.use_library(None)
    device = None
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    dist.init_process_group(backend)
    import pcg_mi355x as pm
    from pcg_mi355x.dist import TorchComm
    import golden_cases
    comm = TorchComm(device=device)
    pm.configure(comm=comm, device=int(os.environ.get("LOCAL_RANK", 0)), operator=opkind)
    out = {"rank": rank}
    if case.startswith("partition:"):           # MDF model -> this rank's part only (pcg_mi355x.partition, only=[rank])
        import partition_cases as pc
        from pcg_mi355x.partition import partition_model
        model, ele_part = pc.build_model(case.split(":", 1)[1])
        assert int(ele_part.max()) + 1 == world
        P = pc.prepare_for_solve(partition_model(model, ele_part, only=[rank]))[0]
        x = np.cos(0.37 * P["DofVector"])
        out["DofVector"] = P["DofVector"]
    elif case.startswith("files:"):             # files:<partition prefix>: this rank's <N>_<rank>.mpidat, as pcg_solver.py:88-110 reads it
        import partition_cases as pc
        from pcg_mi355x.io import read_partition
        P = pc.prepare_for_solve([read_partition(case.split(":", 1)[1], world, rank)])[0]
        x = np.cos(0.37 * P["DofVector"])
        out["DofVector"] = P["DofVector"]
    elif case.startswith("brick:"):             # brick:<N>:<n_types>:<px>x<py>x<pz> - this rank's part only (make_parts(only=))
        from pcg_mi355x.brick import Brick, make_parts, block_partition
        _, n, nt, grid = case.split(":")
        brick = Brick(int(n), n_types=int(nt))
        P = make_parts(brick, block_partition(brick, *[int(v) for v in grid.split("x")]), only=[rank])[0]
        x = np.cos(0.37 * P["DofVector"])
        out["DofVector"] = P["DofVector"]
    elif case == "island":                      # two neighbouring parts + one part WITHOUT neighbours (tests/util.island_parts)
        from util import island_parts
        parts = island_parts()
        assert len(parts) == world
        P = parts[rank]
        x = np.cos(0.37 * P["DofVector"])
    else:
        brick, parts = golden_cases.build_case(case, os.path.join(ROOT, "tests", "golden"))
        assert len(parts) == world, (len(parts), world)
        P = parts[rank]
        x = golden_cases.probe_for(brick, parts)[P["DofVector"]]
    out["y_probe"] = pm.calc_mpfint(x, P)
    out["diag"] = pm.calc_matvec_prod(P, "Preconditioner")
    pm.update_bc(P)
    pm.update_preconditioner(P)
    out["Fext"] = P["Fext"]
    try:
        ret = pm.solve(P, history=True)
        out["raised"] = ""
    except Warning as w:
        ret = None
        out["raised"] = str(w)
    info = P["_pcg_mi355x_info"]
    out["Un"] = P["Un"]
    out["history"] = info.history
    out["flag"], out["iter"], out["relres"] = info.flag, info.iter, info.relres
    out["iters_done"], out["iters_enqueued"] = info.iters_done, info.iters_enqueued
    gd = P["GlobData"]
    out["tl_flag"], out["tl_iter"], out["tl_relres"] = gd["TimeList_Flag"][1], gd["TimeList_Iter"][1], gd["TimeList_RelRes"][1]
    out["n_allreduce"], out["n_halo"] = comm.n_allreduce, comm.n_halo
    out["dofs"] = P["DofVector"]
    out["t_comm"] = gd["MP_TimeRecData"]["dT_CommWait"]
    out["t_calc"] = gd["MP_TimeRecData"]["dT_Calc"]
    if not out["raised"] and ret is None:      # result export in the reference's layout (exportContourData :866-868)
        from pcg_mi355x.io import ResultExporter
        ex = ResultExporter(P, os.path.join(outdir, "ResVecData") + os.sep, comm)
        ex.export(1.0)
    if backend == "nccl":
        # the halo hook's collective on real RCCL: all_to_all_single on raw-pointer tensor views, issued asynchronously
        # on the engine's HIP stream (wrapped as an ExternalStream) and waited for stream-side.  World size 1 sends to
        # itself; the split-size form and the tensor / stream plumbing are the ones halo_begin / halo_end use.
        op = P["_pcg_mi355x_operator"]
        n_a2a = 4099
        a = torch.arange(n_a2a, dtype=torch.float64, device=device) * 0.5
        b = torch.zeros_like(a)
        torch.cuda.synchronize()
        stream_ptr = op._L.pcg_stream(op._h)
        comm._use_stream(stream_ptr)
        send, recv = comm._tensor(a.data_ptr(), n_a2a), comm._tensor(b.data_ptr(), n_a2a)
        splits = [0] * world
        splits[rank] = n_a2a
        work = dist.all_to_all_single(recv, send, splits, splits, async_op=True)
        work.wait()
        torch.cuda.synchronize()
        out["a2a_ok"] = bool(torch.equal(a, b))
        del send, recv
        comm._views.clear()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
