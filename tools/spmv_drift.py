#!/usr/bin/env python
"""Does the SpMV launch time drift with sustained load (DVFS / power state)?  Back-to-back batches of the stand-alone
SpMV benchmark on one operator; prints the median of every batch and the time since the first launch.
usage: spmv_drift.py [N] [batches] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pcg-mpi-solver_amd")]
import numpy as np
from pcg_mi355x.brick import Brick, make_parts
from pcg_mi355x.operator import from_refmeshpart
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 40
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
op = from_refmeshpart(make_parts(Brick(N))[0])
t0 = time.perf_counter()
out = []
for b in range(batches):
    ms = op.bench_spmv(0, reps)
    out.append((time.perf_counter() - t0, float(np.median(ms)), float(ms.min()), float(ms.max())))
for t, med, lo, hi in out:
    print(f"t={t:7.2f}s median {med:.4f} min {lo:.4f} max {hi:.4f}")
time.sleep(float(os.environ.get("DRIFT_PAUSE_S", "5")))
ms = op.bench_spmv(0, reps)
print(f"after a pause: median {float(np.median(ms)):.4f} min {float(ms.min()):.4f}")
