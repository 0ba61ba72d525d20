#!/usr/bin/env python
"""Development tool: A/B of the matrix-free hex8 kernel variants on ONE box, one process (same clocks, same data).

For every configuration (environment knobs read when an engine is created) the 10 M-dof brick operator is built, checked
against the assembled operator's product on a random vector (<= 1e-13) and timed with pcg_bench_spmv (HIP events around one
whole apply = element kernel + shared-node sums), with and without the fused p.Ap epilogue.
usage: python tools/ebe_lab.py [N] [config ...]      config = name:KEY=VAL,KEY=VAL"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pcg-mpi-solver_amd")]
import numpy as np
from pcg_mi355x import _lib
if os.environ.get("PCG_LAB_LIB"):            # another build of the engine (development)
    _lib.use_library(os.environ["PCG_LAB_LIB"])
ABLATION = bool(os.environ.get("PCG_LAB_LIB"))
from pcg_mi355x.brick import Brick, make_parts
from pcg_mi355x.operator import from_refmeshpart

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 150
# (round 3: the round-1 kernel, the read-add-write accumulation and the matrix-core variant were removed from the library; what is
#  still selectable is the chunk size of the hex8 class)
DEFAULT = ["hexs_512_two_pass:PCG_EBE_EPT=2", "hex_256:PCG_EBE_EPT=1"]
configs = [a for a in sys.argv[1:] if ":" in a] or DEFAULT
KNOBS = ("PCG_EBE_EPT", "PCG_BENCH_SPMV_DOT")

b = Brick(N)
P = make_parts(b)[0]
x = np.random.default_rng(1).standard_normal(b.n_dof)
for k in KNOBS:
    os.environ.pop(k, None)
ref_op = from_refmeshpart(P, kind="sell")
y_ref = ref_op.apply(x)
ref_op.close()
out = {"N": N, "dof": b.n_dof}
for cfg in configs:
    name, kv = cfg.split(":", 1)
    for k in KNOBS:
        os.environ.pop(k, None)
    for item in kv.split(","):
        k, v = item.split("=")
        os.environ[k] = v
    res = {}
    for dot in ("0", "1"):
        os.environ["PCG_BENCH_SPMV_DOT"] = dot
        op = from_refmeshpart(P, kind="ebe")
        if dot == "0":
            err = np.linalg.norm(op.apply(x) - y_ref) / np.linalg.norm(y_ref)
            res["rel_err_vs_assembled"] = float(err)
            res["chunks"] = op.operator_info()["n_chunks"]
        ms = op.bench_spmv(20, 200)
        res["dot" + dot] = {"median_ms": float(np.median(ms)), "min_ms": float(ms.min())}
        op.close()
    out[name] = res
    print(name, json.dumps(res), file=sys.stderr, flush=True)
    assert ABLATION or res["rel_err_vs_assembled"] < 1e-13, (name, res)
print(json.dumps(out))
