"""A/B of the assembled operator's storage formats on one GPU: plain SELL-BSR3 values vs the value dictionary
(PCG_FORMAT_DICTIONARY, k_spmv_dict).  usage: python tools/dict_lab.py [N=150; 0 = the octree workload] [steps=200] [kinds=sell,dict]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pcg-mpi-solver_amd"))
import numpy as np

import pcg_mi355x as pm
from pcg_mi355x.brick import Brick, make_parts
from pcg_mi355x.operator import from_refmeshpart

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
kinds = sys.argv[3].split(",") if len(sys.argv) > 3 else ["sell", "dict"]
if N == 0:                                                  # the bench's octree workload: two levels, hanging-node patterns
    from pcg_mi355x.octree import TwoLevelMesh, make_octree_parts
    P = make_octree_parts(TwoLevelMesh(96, 96, 40, 8, seed=0), 1, axis=0)[0]
else:
    b = Brick(N, seed=0)
    P = make_parts(b)[0]
ys = {}
for kind in kinds:
    t0 = time.perf_counter()
    op = from_refmeshpart(P, kind=kind)
    t_setup = time.perf_counter() - t0
    by, fl = op.operator_cost()
    ms = op.bench_spmv(10, 50)
    x = np.random.default_rng(0).standard_normal(op.n)
    ys[kind] = op.apply(x)
    fext, udi = op.update_bc(P["RefLoadVector"], P["Ud"], 1.0)
    inv = op.build_jacobi()
    op.solve_begin(fext, np.zeros(op.n), inv, 1e-7, 10000, int(P["GlobData"]["GlobNDofEff"]))
    op.solve_run(20)
    op.set_profiling(True)
    t0 = time.perf_counter()
    r = op.solve_run(steps)
    dt = time.perf_counter() - t0
    op.set_profiling(False)
    t0 = time.perf_counter()
    op.solve_run(-1)
    xs, res = op.solve_end()
    t_rest = time.perf_counter() - t0
    print(f"{kind:5s} N={N} dof={op.n} dict={op.matrix_dictionary_info()} setup {t_setup:.1f}s bytes/apply {by/1e9:.3f} GB | spmv standalone "
          f"median {np.median(ms):.4f} min {ms.min():.4f} ms | in-loop {r.spmv_ms_sum / max(1, r.spmv_count):.4f} ms | "
          f"{steps / dt:.0f} it/s ({dt / steps * 1e3:.4f} ms/it) | solve flag {res.flag} iter {res.iter} relres {res.relres:.3e} "
          f"({dt + t_rest:.2f}s for the rest)", flush=True)
    op.close()
if len(ys) == 2:
    a, c = ys[kinds[0]], ys[kinds[1]]
    print("bit-identical apply:", bool(np.array_equal(a, c)))
