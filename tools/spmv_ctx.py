#!/usr/bin/env python
"""Stand-alone SpMV timing under different launch contexts (PCG_BENCH_SPMV_CTX / PCG_BENCH_SPMV_DOT; development)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pcg-mpi-solver_amd")]
import numpy as np
import pcg_mi355x as pm
from pcg_mi355x.brick import Brick, make_parts
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
part = make_parts(Brick(N))[0]
pm.configure(comm=None, device=0, operator="sell")
op = pm.get_operator(part)
pm.update_bc(part); pm.update_preconditioner(part)
for rnd in range(2):
    for ctx in [int(c) for c in os.environ.get("CTX_LIST", "0,4,1,5,2,6,7").split(",")]:
        os.environ["PCG_BENCH_SPMV_CTX"] = str(ctx)
        ms = op.bench_spmv(5, 60)
        print(f"ctx {ctx} (x rewritten {ctx & 1}{' nt-stores' if ctx & 16 else ''}{' sc1-stores' if ctx & 64 else ''}{' sc0sc1-stores' if ctx & 128 else ''}{' nt-sc0sc1-stores' if ctx & 256 else ''}, 1 GiB read before {(ctx >> 1) & 1}{' plain loads' if ctx & 32 else ''}, no host wait {(ctx >> 2) & 1}): median {float(np.median(ms)):.4f} min {float(ms.min()):.4f}", flush=True)
