#!/usr/bin/env python
"""Run a few stand-alone operator applies (for rocprofv3 PMC passes).  usage: prof_op.py [sell|dict|ebe[,...]] [N] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pcg-mpi-solver_amd")]
import numpy as np
if os.environ.get("PROF_IMPORT_TORCH"):     # does a process that also runs torch's HIP context time the same kernel differently?
    import torch
    torch.cuda.init()
    _t = torch.zeros(1 << 20, device="cuda")
    print("torch", torch.__version__, "context up")
if os.environ.get("PCG_LIB"):               # A/B against another build of the engine (development only)
    from pcg_mi355x import _lib
    _lib.use_library(os.environ["PCG_LIB"])
from pcg_mi355x.brick import Brick, make_parts
from pcg_mi355x.operator import from_refmeshpart
kind = sys.argv[1] if len(sys.argv) > 1 else "ebe"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 150
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
if os.environ.get("PROF_OCTREE"):          # PROF_OCTREE=1m|10m (1ms|10ms: pattern types by symmetry class): the multi-level graded octree mesh (N is ignored)
    from pcg_mi355x.octree import GradedOctreeMesh, make_octree_parts
    m = GradedOctreeMesh({"1m": (12, 12, 12), "10m": (38, 38, 38)}[os.environ["PROF_OCTREE"].rstrip("s")], 4, band=1.2,
                        symmetry=os.environ["PROF_OCTREE"].endswith("s"))
    P = make_octree_parts(m, 1)[0]
    print("octree mesh", m.summary())
else:
    P = make_parts(Brick(N))[0]
for kd in kind.split(","):                  # several operators in one process (one rocprofv3 pass covers them all)
    op = from_refmeshpart(P, kind=kd, ebe_chunked=os.environ.get("PROF_EBE_CHUNKED", "1") == "1")
    ms = op.bench_spmv(3, reps)
    print(kd, N, op.operator_info(), "median ms", float(np.median(ms)), "min", float(ms.min()), flush=True)
    op.close()
