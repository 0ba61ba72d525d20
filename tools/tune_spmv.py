#!/usr/bin/env python
"""SpMV tuning sweep on one GPU (development tool): SELL slice height (rows per lane) x blocks per CU,
next to a torch device-to-device copy as the streaming-bandwidth yardstick of the box.
usage: python tools/tune_spmv.py [N] > gpurun_out/tune.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pcg-mpi-solver_amd")]
import numpy as np
import torch
from pcg_mi355x.brick import Brick, make_parts
from pcg_mi355x.operator import assemble_bsr3, Operator

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
res = {"N": N}
# yardstick: 4 GiB copy
n = 1 << 29
a = torch.empty(n, dtype=torch.float64, device="cuda").normal_()
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    b.copy_(a)
e1.record(); torch.cuda.synchronize()
res["copy_GBps_rw"] = 2 * n * 8 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
e0.record()
for _ in range(10):
    s = a.sum()
e1.record(); torch.cuda.synchronize()
res["torch_sum_read_GBps"] = n * 8 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
del a, b
torch.cuda.empty_cache()
print("yardstick", res, file=sys.stderr, flush=True)

t0 = time.time()
brick = Brick(N)
P = make_parts(brick)[0]
rp, c, v = assemble_bsr3(P["SubDomainData"]["StrucDataList"], brick.n_node)
res["setup_s"] = time.time() - t0
alg = 12.0 * brick.nnz + 20.0 * brick.n_dof
res["alg_bytes"] = alg
res["runs"] = []
QUICK = os.environ.get("TUNE_QUICK", "0") == "1"          # one blocked configuration (the default) as the in-box yardstick
for rpl in ((1,) if QUICK else (1, 2)):
    for dot in ((1,) if QUICK else (0, 1)):
        for xcd in ((0,) if QUICK else (1, 0)):
            for bpc in (4,):
                os.environ["PCG_SPMV_BLOCKS_PER_CU"] = str(bpc)
                os.environ["PCG_SPMV_XCD"] = str(xcd)
                os.environ["PCG_BENCH_SPMV_DOT"] = str(dot)
                op = Operator(brick.n_node, rp, c, v, 0, None, 0, rpl)
                info = op.matrix_info()
                ms = op.bench_spmv(5, 30)
                impl = info["stored_blocks"] * 76.0 + 16.0 * brick.n_dof
                r = {"rpl": rpl, "dot": dot, "xcd_aware": xcd, "blocks_per_cu": bpc, "min_ms": float(ms.min()),
                     "med_ms": float(np.median(ms)), "alg_GBps": alg / (float(np.median(ms)) * 1e-3) / 1e9,
                     "impl_GBps": impl / (float(np.median(ms)) * 1e-3) / 1e9, "padding": info["stored_blocks"] / info["nnzb"] - 1}
                res["runs"].append(r)
                print(r, file=sys.stderr, flush=True)
                op.close()
# the literal CSR data volume: scalar rows, 8 B value + 4 B column per non-zero (pcg_create_csr block = 1)
if os.environ.get("TUNE_SCALAR", "1") == "1":
    import scipy.sparse as sp
    t0 = time.time()
    A = sp.bsr_matrix((v.reshape(-1, 3, 3), c, rp), shape=(brick.n_dof, brick.n_dof)).tocsr()
    del v, c, rp
    res["scalar_setup_s"] = time.time() - t0
    res["scalar_runs"] = []
    for dot in (0, 1):
        for bpc in (4, 8):
            os.environ["PCG_SPMV_BLOCKS_PER_CU"] = str(bpc)
            os.environ["PCG_BENCH_SPMV_DOT"] = str(dot)
            op = Operator.from_csr(A.indptr, A.indices, A.data, block=1)
            info = op.matrix_info()
            ms = op.bench_spmv(5, 30)
            impl = info["stored_blocks"] * 12.0 + 16.0 * brick.n_dof      # stored entries x (f64 + i32) + x + y
            r = {"format": "scalar", "dot": dot, "blocks_per_cu": bpc, "min_ms": float(ms.min()), "med_ms": float(np.median(ms)),
                 "alg_GBps": alg / (float(np.median(ms)) * 1e-3) / 1e9, "impl_GBps": impl / (float(np.median(ms)) * 1e-3) / 1e9,
                 "padding": info["stored_blocks"] / info["nnzb"] - 1}
            res["scalar_runs"].append(r)
            print(r, file=sys.stderr, flush=True)
            op.close()
print(json.dumps(res))
