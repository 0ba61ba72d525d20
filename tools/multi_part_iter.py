#!/usr/bin/env python
"""Per-rank iteration time of the MULTI-PART loop on one GPU (development tool, round 4).

One part of the 2x2x2 split of the N-node brick (N = 150: 1.27 M dof + halo = one GPU's share of the 10 M-dof system on 8) runs the
multi-part loop on the real RCCL at world size 1: its neighbours are folded into ONE neighbour - itself (PCG_RCCL_ALLOW_SELF=1) -
so the interface rows' launch, the pack, the grouped ncclSend / ncclRecv on the communication stream, the fix-up and both
ncclAllReduce run as they would on 8 GPUs, minus the wire.  (The exchange returns the part's own partial sums, i.e. the operator
is not the assembled one: timing only - the window is short and checked for an early exit.)
usage: python tools/multi_part_iter.py [N] [steps] [kinds] [modes]      modes: PCG_ITER_FUSED values switched per solve (default 1,0);
       "m" = the five-launch form with the engine-side reduction (pcg_comm_enable_mailbox: no ncclAllReduce kernel; round 5);
       "d" = the engine-side EXCHANGE (pcg_enable_direct_exchange: stores into the neighbour's mapped buffer, no ncclSend / ncclRecv kernel, one
       stream), "dm" = both: no collective kernel left in the iteration"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pcg-mpi-solver_amd")]
os.environ["PCG_RCCL_ALLOW_SELF"] = "1"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np
import torch
from pcg_mi355x.brick import Brick, make_parts, block_partition
from pcg_mi355x.dist import RcclComm
from pcg_mi355x.operator import from_refmeshpart

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
kinds = (sys.argv[3] if len(sys.argv) > 3 else "sell,ebe").split(",")
modes = (sys.argv[4] if len(sys.argv) > 4 else "1,0").split(",")
b = Brick(N)
P = make_parts(b, block_partition(b, 2, 2, 2), only=[0])[0]
ovl = np.unique(np.concatenate([np.asarray(v, np.int64) for v in P["OvrlpLocalDofVecList"]]))
P["NbrMPIdVector"], P["OvrlpLocalDofVecList"], P["Id"] = [0], [ovl], 0
P["DofWeightVector"] = np.ones(P["NDOF"])
comm = RcclComm(0, 1, 0, RcclComm.new_unique_id())
out = []
for kind in kinds:
    op = from_refmeshpart(P, comm=comm, kind=kind)
    fext = np.random.default_rng(1).standard_normal(op.n)       # (this corner part carries no load of the brick's own load case)
    inv = op.build_jacobi()
    for rep in range(2):
        for fused in modes:
            os.environ["PCG_ITER_FUSED"] = "1" if fused in ("m", "d", "dm") else fused
            assert comm.enable_mailbox(fused in ("m", "dm")) == (fused in ("m", "dm")), comm.mailbox_reason
            assert op.enable_direct_exchange(fused in ("d", "dm")) == (fused in ("d", "dm")), op.direct_exchange_reason
            op.solve_begin(fext, None, inv, 1e-30, 100000, P["GlobData"]["GlobNDofEff"])
            op.solve_run(5)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = op.solve_run(steps)
            torch.cuda.synchronize(); t = time.perf_counter() - t0
            op.solve_end()
            rec = {"N": N, "dof": op.n, "interface_dofs": int(len(ovl)), "kind": kind, "PCG_ITER_FUSED": fused, "rep": rep,
                   "us_per_iter": t / steps * 1e6, "iters_done": int(r.iters_done), "ended_early": bool(r.iters_done < 5 + steps)}
            out.append(rec); print(rec, file=sys.stderr, flush=True)
    op.close()
comm.close()
print(json.dumps(out))
