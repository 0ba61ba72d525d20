#!/usr/bin/env python
"""Split SELL format: sweep of the planner's two parameters (PCG_SELL_SPLIT_F = price of an overflow block relative to a base
block when the base width of a slice is chosen; PCG_SELL_SPLIT_WINDOW = rows per sorting window of the overflow part) on the
graded octree mesh, one process.  usage: python tools/split_sweep.py oct1m|oct10m [steps] [f:window ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pcg-mpi-solver_amd")]
import numpy as np
import torch
from pcg_mi355x.octree import GradedOctreeMesh, make_octree_parts
from pcg_mi355x.operator import from_refmeshpart
which = sys.argv[1] if len(sys.argv) > 1 else "oct1m"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
combos = sys.argv[3:] or ["0", "1.5:512", "3:512", "2:128", "3:128", "3:64"]
P = make_octree_parts(GradedOctreeMesh({"oct1m": (12, 12, 12), "oct10m": (38, 38, 38)}[which], 4, band=1.2), 1)[0]
for cb in combos:
    if cb == "0":
        os.environ["PCG_SELL_SPLIT"] = "0"
    else:
        os.environ.pop("PCG_SELL_SPLIT", None)
        parts = cb.split(":")                       # f:window[:s] - s = lanes of an overflow slice stay sorted by excess length
        os.environ["PCG_SELL_SPLIT_F"], os.environ["PCG_SELL_SPLIT_WINDOW"] = parts[0], parts[1]
        if len(parts) > 2: os.environ["PCG_SELL_SPLIT_KEEP_SORTED"] = "1"
        else: os.environ.pop("PCG_SELL_SPLIT_KEEP_SORTED", None)
    op = from_refmeshpart(P, kind="sell")
    info = op.matrix_info()
    fext, _ = op.update_bc(P["RefLoadVector"], P["Ud"], 1.0)
    inv = op.build_jacobi()
    rec = {"mesh": which, "f:window": cb, "stored_over_true": round(info["stored_blocks"] / info["nnzb"], 4)}
    for prof in (False, True):
        op.solve_begin(fext, None, inv, 1e-30, 100000, P["GlobData"]["GlobNDofEff"])
        op.solve_run(10)
        op.set_profiling(prof)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = op.solve_run(steps)
        torch.cuda.synchronize(); t = time.perf_counter() - t0
        op.set_profiling(False)
        op.solve_end()
        if not prof: rec.update(us_per_iter=round(t / steps * 1e6, 1), it_per_s=round(steps / t, 1))
        else: rec.update(operator_us=round(r.spmv_ms_sum / max(1, r.spmv_count) * 1e3, 1))
    print(json.dumps(rec), flush=True)
    op.close()
