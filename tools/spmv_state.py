#!/usr/bin/env python
"""Which step of bench.py's sequence changes the stand-alone SpMV time?  (development)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pcg-mpi-solver_amd")]
import numpy as np
import pcg_mi355x as pm
from pcg_mi355x.brick import Brick, make_parts
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
part = make_parts(Brick(N))[0]
pm.configure(comm=None, device=0, operator="sell")
op = pm.get_operator(part)
def t(tag):
    ms = op.bench_spmv(5, 60)
    print(f"{tag:40s} median {float(np.median(ms)):.4f} min {float(ms.min()):.4f}", flush=True)
t("after creation")
pm.update_bc(part); t("after update_bc")
pm.update_preconditioner(part); t("after update_preconditioner")
tz = np.zeros(op.n); tz[2::3] = 1.0
op.apply(tz); t("after apply")
gd = part["GlobData"]
eff = np.asarray(part["LocDofEff"], np.int64)
inv = np.zeros(op.n); inv[eff] = part["InvDiagPreCondVector0"]
op.solve_begin(part["Fext"], np.zeros(op.n), inv, float(gd["Tol"]), 5000, int(gd["GlobNDofEff"]))
t("after solve_begin")
r = op.solve_run(20); t("after 20 iterations")
op.set_profiling(True)
r = op.solve_run(200)
print("in-loop operator ms", r.spmv_ms_sum / max(1, r.spmv_count), flush=True)
t("after 220 iterations")
op.set_profiling(False); t("profiling off")
op.solve_end(); t("after solve_end")
if os.environ.get("STATE_STREAM"):
    print("read stream", round(op.bench_hbm(8 << 30, "read", 10))); t("after stream bench")
