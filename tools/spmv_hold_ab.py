#!/usr/bin/env python
"""k_spmv with / without HOLD (y stores at the end of a launch), stand-alone (back-to-back launches, nothing else on the GPU) and inside the
PCG loop (between two k_vec launches that leave 243 MB of freshly written vectors behind), same process.  usage: spmv_hold_ab.py [N=150]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pcg-mpi-solver_amd")]
import numpy as np
import torch
from pcg_mi355x.brick import Brick, make_parts
from pcg_mi355x.operator import from_refmeshpart
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
P = make_parts(Brick(N))[0]
for rep in range(2):
    for hold, place in (("0", "0"), ("4", "0"), ("0", "1"), ("4", "1"), ("auto", "1")):
        if hold == "auto": os.environ.pop("PCG_SPMV_HOLD", None)
        else: os.environ["PCG_SPMV_HOLD"] = hold
        os.environ["PCG_VEC_PLACEMENT"] = place
        os.environ["PCG_VEC_PLACEMENT_LOG"] = "1"
        op = from_refmeshpart(P, kind="sell")
        ms = op.bench_spmv(10, 60)
        fext, _ = op.update_bc(P["RefLoadVector"], P["Ud"], 1.0)
        inv = op.build_jacobi()
        op.solve_begin(fext, None, inv, 1e-30, 100000, P["GlobData"]["GlobNDofEff"])
        op.solve_run(20)
        op.set_profiling(True)
        r = op.solve_run(100)
        op.set_profiling(False)
        op.solve_end()
        print({"N": N, "hold": hold, "placement": place, "rep": rep, "standalone_median_us": float(np.median(ms)) * 1e3, "standalone_min_us": float(ms.min()) * 1e3,
               "in_loop_us": r.spmv_ms_sum / max(1, r.spmv_count) * 1e3, "tuning": op.tuning_info()}, flush=True)
        op.close()
