#!/usr/bin/env python
"""Per-iteration time of the PCG loop, same process / same operator A/B over environment switches the engine re-reads at
solve_begin (PCG_VEC_FUSED, PCG_VEC_NT, PCG_VEC_KREG, PCG_LOOK_AHEAD is read at creation).
usage: python tools/iter_ab.py N[,N..] kind[,kind..] [steps] [VAR=a|b ...]      e.g.  iter_ab.py 75,150 ebe,dict 200 PCG_VEC_FUSED=1|0
N = nodes per side of the brick, or oct1m / oct10m = the graded octree mesh (oct1ms / oct10ms: one pattern type per symmetry class,
GradedOctreeMesh(symmetry=True))."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pcg-mpi-solver_amd")]
import numpy as np
import torch
if os.environ.get("PCG_LIB"):               # A/B against another build of the engine (tools/build_variant.sh; development only)
    from pcg_mi355x import _lib
    _lib.use_library(os.environ["PCG_LIB"])
from pcg_mi355x.brick import Brick, make_parts
from pcg_mi355x.operator import from_refmeshpart
Ns = [a if a.startswith("oct") else int(a) for a in sys.argv[1].split(",")]      # brick nodes per side, or oct1m / oct10m
kinds = sys.argv[2].split(",")
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
switches = [a.split("=", 1) for a in sys.argv[4:]] or [["PCG_VEC_FUSED", "1|0"]]
out = []
CREATE = ("PCG_LOOK_AHEAD", "PCG_SPMV_COL16", "PCG_SPMV_DICT_LDS", "PCG_SPMV_DICT_BLOCK", "PCG_EBE_EPT", "PCG_EBE_GREEDY_CHUNKS", "PCG_EBE_STREAMS", "PCG_EBE_ROWS_LDS", "PCG_EBE_DIRECT", "PCG_EBE_XCD", "PCG_SPMV_XCD", "PCG_SELL_SPLIT", "PCG_EBE_MIXED", "PCG_NODE_ORDER", "PCG_SPMV_OVF", "PCG_EBE_MIX_FLAGS", "PCG_EBE_TILE_CAP", "PCG_EBE_HEX_CAP", "PCG_EBE_HEX_FLAGS", "PCG_SPMV_OVF_WINDOW", "PCG_EBE_NODE_CAP", "PCG_EBE_MIX_SHELLS", "PCG_EBE_MIX_MTM", "PCG_EBE_HEX_TILES", "PCG_EBE_TARGET_CHUNKS", "PCG_EBE_MTILE_WAVES", "PCG_SPMV_HOLD", "PCG_VEC_PLACEMENT", "PCG_SPMV_PLACEMENTS")   # read when the operator is built
create = [sw for sw in switches if sw[0].split("+")[0] in CREATE] or [["_", "-"]]      # A+B=a1+b1|a2+b2: several variables switched together
switches = [sw for sw in switches if sw[0].split("+")[0] not in CREATE] or [["_", "-"]]
for N in Ns:
    if isinstance(N, str):                       # the multi-level graded octree mesh (pcg_mi355x.octree.GradedOctreeMesh)
        from pcg_mi355x.octree import GradedOctreeMesh, make_octree_parts
        P = make_octree_parts(GradedOctreeMesh({"oct1m": (12, 12, 12), "oct2m": (15, 15, 15), "oct3m": (18, 18, 18), "oct5m": (22, 22, 22), "oct10m": (38, 38, 38)}[N.rstrip("s")], 4, band=1.2, symmetry=N.endswith("s")), 1)[0]
    else:
        P = make_parts(Brick(N))[0]
    for kind in kinds:
      for cvar, cvals in create:
        for cv in cvals.split("|"):
          for name, val in zip(cvar.split("+"), cv.split("+")):
              os.environ[name] = val
          op = from_refmeshpart(P, kind=kind)
          if kind != "ebe":
              print({"N": N, "kind": kind, cvar: cv, "matrix": op.matrix_info(), "bytes": op.operator_cost()[0]}, file=sys.stderr, flush=True)
          fext, _ = op.update_bc(P["RefLoadVector"], P["Ud"], 1.0)
          inv = op.build_jacobi()
          for var, vals in switches:
            for rep in range(2):
                for v in vals.split("|"):
                    os.environ[var] = v
                    rec = {"N": N, "dof": op.n, "kind": kind, cvar: cv, var: v, "rep": rep}
                    for prof in (False, True):                        # un-profiled window first (events around every launch perturb it)
                        op.solve_begin(fext, None, inv, 1e-30, 100000, P["GlobData"]["GlobNDofEff"])
                        op.solve_run(20)
                        op.set_profiling(prof)
                        torch.cuda.synchronize(); t0 = time.perf_counter()
                        r = op.solve_run(steps)
                        torch.cuda.synchronize(); t = time.perf_counter() - t0
                        op.set_profiling(False)
                        op.solve_end()
                        if not prof:
                            rec.update(us_per_iter=t / steps * 1e6, it_per_s=steps / t)
                        else:
                            rec.update(operator_us=r.spmv_ms_sum / max(1, r.spmv_count) * 1e3, vec_us=r.vec_ms_sum / max(1, r.vec_count) * 1e3)
                    rec.pop("_", None)
                    out.append(rec); print(rec, file=sys.stderr, flush=True)
            os.environ.pop(var, None)
          op.close()
        for name in cvar.split("+"):
            os.environ.pop(name, None)
print(json.dumps(out))
