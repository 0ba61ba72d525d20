#!/usr/bin/env python
"""Same-process A/B of PCG_VEC_NT (non-temporal stores in k_update_p / k_vec): PCG iterations/s and the
in-loop operator time for both operators.  usage: vec_nt_ab.py [N] [steps]   (development)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pcg-mpi-solver_amd")]
import numpy as np
import pcg_mi355x as pm
from pcg_mi355x.brick import Brick, make_parts
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for kind in os.environ.get("NT_KINDS", "sell,ebe").split(","):
    part = make_parts(Brick(N))[0]
    pm.configure(comm=None, device=0, operator=kind)
    op = pm.get_operator(part)
    pm.update_bc(part); pm.update_preconditioner(part)
    gd = part["GlobData"]
    eff = np.asarray(part["LocDofEff"], np.int64)
    inv = np.zeros(op.n); inv[eff] = part["InvDiagPreCondVector0"]
    xs = {}
    for rnd in range(2):
        for nt in os.environ.get("NT_LIST", "0,1").split(","):
            os.environ["PCG_VEC_NT"] = nt
            op.solve_begin(part["Fext"], np.zeros(op.n), inv, float(gd["Tol"]), 20 + steps + 5, int(gd["GlobNDofEff"]))
            op.solve_run(20)
            op.set_profiling(True)
            op.sync() if hasattr(op, "sync") else None
            t0 = time.perf_counter()
            r = op.solve_run(steps)
            dt = time.perf_counter() - t0
            op.set_profiling(False)
            x, res = op.solve_end()
            xs[nt] = x
            print(f"N={N} {kind:4s} PCG_VEC_NT={nt}: {steps / dt:8.1f} it/s  ({dt / steps * 1e3:.4f} ms/iter), operator in loop {r.spmv_ms_sum / max(1, r.spmv_count):.4f} ms", flush=True)
    print(f"N={N} {kind}: x identical between the two store flavours: {all(np.array_equal(xs[k], xs['0']) for k in xs)}", flush=True)
    op.close(); part.pop("_pcg_mi355x_operator", None)
