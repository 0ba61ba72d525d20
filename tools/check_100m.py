#!/usr/bin/env python
"""BASELINE configs[4] size on ONE GPU (development check, too slow for the test-suite): 100 158 744 dof, 8.06e9 nnz.
Operator parity against the oracle's C mat-vec on the CPU, size-independent properties, and a PCG window whose
recurrence residual must equal the TRUE residual recomputed by the oracle.   usage: python tools/check_100m.py [N=322]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pcg-mpi-solver_amd"), os.path.join(ROOT, "oracle")]
import subprocess
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
import numpy as np
import pcg_mi355x as pm
import pcg_oracle
from pcg_mi355x.brick import Brick, make_parts
from pcg_mi355x.operator import from_refmeshpart

N = int(sys.argv[1]) if len(sys.argv) > 1 else 322
t0 = time.time()
b = Brick(N)
P = make_parts(b)[0]
print(f"N={N}: {b.n_dof} dof, {b.nnz} nnz; RefMeshPart {time.time() - t0:.1f}s", flush=True)
rng = np.random.default_rng(5)
x = rng.standard_normal(b.n_dof)
t0 = time.time(); ref = pcg_oracle.matvec_local(P, x, use_c=True); t_cpu = time.time() - t0
relerr = lambda a, c: np.linalg.norm(a - c) / np.linalg.norm(c)
for kind in ("sell", "ebe"):
    t0 = time.time()
    op = from_refmeshpart(P, kind=kind)
    t_setup = time.time() - t0
    ax = op.apply(x)
    e_op = relerr(ax, ref)
    t = np.zeros(b.n_dof); t[2::3] = 1.0
    rigid = np.abs(op.apply(t)).max()
    fext, _ = op.update_bc(P["RefLoadVector"], P["Ud"], 1.0)
    inv = op.build_jacobi()
    eff = P["LocDofEff"]
    op.solve_begin(fext, None, inv, 1e-7, 60, P["GlobData"]["GlobNDofEff"])
    op.solve_run(-1)
    xk, res = op.solve_end()
    r_true = (fext - pcg_oracle.matvec_local(P, xk, use_c=True))[eff]
    nb = np.linalg.norm(fext[eff])
    e_res = abs(np.linalg.norm(r_true) / nb - res.relres) / res.relres
    ok = e_op < 1e-13 and rigid < 1e-9 and res.flag == 1 and res.iters_done == 60 and e_res < 1e-9
    print(f"[{kind}] set-up {t_setup:.1f}s | operator vs oracle {e_op:.2e} (oracle mat-vec on the CPU {t_cpu:.1f}s) | |A t_z|max {rigid:.1e} | "
          f"60 iterations: flag {res.flag}, relres {res.relres:.6e}, |true - recurrence| / relres {e_res:.1e} | {'OK' if ok else 'FAIL'}", flush=True)
    op.close()
    if not ok:
        sys.exit(1)
