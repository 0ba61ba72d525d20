#!/usr/bin/env python
"""SpMV grid size at small systems (development tool): blocks per CU sweep at N nodes per side.
usage: python tools/tune_small.py [N ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pcg-mpi-solver_amd")]
import numpy as np
from pcg_mi355x.brick import Brick, make_parts
from pcg_mi355x.operator import assemble_bsr3, Operator
out = []
for N in [int(a) for a in sys.argv[1:]] or [70]:
    b = Brick(N)
    P = make_parts(b)[0]
    rp, c, v = assemble_bsr3(P["SubDomainData"]["StrucDataList"], b.n_node)
    for bpc in (2, 4, 6, 8, 12, 16):
        os.environ["PCG_SPMV_BLOCKS_PER_CU"] = str(bpc)
        os.environ["PCG_BENCH_SPMV_DOT"] = "1"
        op = Operator(b.n_node, rp, c, v, 0, None, 0, 1)
        ms = op.bench_spmv(10, 60)
        r = {"N": N, "dof": b.n_dof, "slices": op.matrix_info()["n_slices"], "blocks_per_cu": bpc, "med_ms": float(np.median(ms)), "min_ms": float(ms.min())}
        out.append(r); print(r, file=sys.stderr, flush=True)
        op.close()
print(json.dumps(out))
