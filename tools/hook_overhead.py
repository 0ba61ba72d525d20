#!/usr/bin/env python
"""What the multi-GPU plumbing costs per iteration, measured on ONE GPU (development tool).

The per-GPU share of the 10 M-dof system on 8 GPUs is ~1.27 M dof (N = 75).  This runs that part alone
(a) without a communicator, (b) with TorchComm on an nccl process group of world size 1, where the two
all-reduce hooks per iteration run for real (Python callback -> torch.distributed -> RCCL kernel) but there is
no neighbour, and (c) with the engine's NATIVE communicator (RcclComm, world size 1: ncclAllReduce issued from C++ on
the compute stream).  (b) - (a) and (c) - (a) are the host/launch overheads every rank of a multi-GPU solve pays per
iteration before any wire time.   usage: python tools/hook_overhead.py [N]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pcg-mpi-solver_amd")]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np
import torch
import torch.distributed as dist
import pcg_mi355x as pm
from pcg_mi355x.brick import Brick, make_parts
from pcg_mi355x.dist import TorchComm, RcclComm
from pcg_mi355x.operator import from_refmeshpart

N = int(sys.argv[1]) if len(sys.argv) > 1 else 75
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29711", world_size=1, rank=0, device_id=torch.device("cuda", 0))
out = {"N": N}
for kind in ("sell", "ebe"):
    for with_comm in (False, "hooks", "native"):
        b = Brick(N)
        P = make_parts(b)[0]
        comm = None
        if with_comm == "hooks":
            comm = TorchComm(device=torch.device("cuda", 0))
        elif with_comm == "native":
            comm = native = globals().get("native") or RcclComm(0, 1, 0, RcclComm.new_unique_id())
            globals()["native"] = native
        op = from_refmeshpart(P, comm=comm, kind=kind)
        pm.configure(comm=comm, operator=kind)
        P["_pcg_mi355x_operator"] = op
        pm.update_bc(P); pm.update_preconditioner(P)
        gd = P["GlobData"]
        gd["MaxIter"] = 300                                  # fixed window, exits with Flag 1
        import time
        pm.solve(P)                                          # warm-up
        P["Un"] = np.zeros(P["NDOF"])
        t0 = time.perf_counter(); pm.solve(P); dt = time.perf_counter() - t0
        info = P["_pcg_mi355x_info"]
        key = f"{kind}_{with_comm if with_comm else 'nocomm'}"
        out[key] = {"iters": int(info.iter), "ms_per_iter_engine": 1e3 * info.t_total_s / max(1, info.iters_done), "t_comm_s": info.t_comm_s,
                    "wall_s": dt, "allreduce_calls": getattr(comm, "n_allreduce", 0)}
        print(key, out[key], file=sys.stderr, flush=True)
        op.close()
print(json.dumps(out))
dist.destroy_process_group()
