#!/usr/bin/env python
"""Hanging-node element kernel without a node tile (k_ebe_direct, the default) against the tile kernel (PCG_EBE_DIRECT=0) and the
assembled operator: one apply on the same vector, graded octree mesh.  usage: python tools/ebe_direct_check.py [oct1m|oct10m|small]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pcg-mpi-solver_amd")]
import numpy as np
from pcg_mi355x.octree import GradedOctreeMesh, make_octree_parts
from pcg_mi355x.operator import from_refmeshpart
which = sys.argv[1] if len(sys.argv) > 1 else "oct1m"
roots = {"small": (4, 4, 4), "oct1m": (12, 12, 12), "oct10m": (38, 38, 38)}[which]
P = make_octree_parts(GradedOctreeMesh(roots, 4 if which != "small" else 3, band=1.2), 1)[0]
rng = np.random.default_rng(5)
ys = {}
x = None
for tag, kind, env in (("sell", "sell", {}), ("tile", "ebe", {"PCG_EBE_DIRECT": "0"}), ("direct", "ebe", {"PCG_EBE_DIRECT": "1"})):
    os.environ.update(env)
    op = from_refmeshpart(P, kind=kind)
    if x is None:
        x = rng.standard_normal(op.n)
    ys[tag] = np.array(op.apply(x))
    op.close()
scale = np.abs(ys["sell"]).max()
for a, b in (("tile", "sell"), ("direct", "sell"), ("direct", "tile")):
    print(f"{which} n={x.size} max|{a}-{b}|/max|y| = {np.abs(ys[a] - ys[b]).max() / scale:.3e}", flush=True)
assert np.abs(ys["direct"] - ys["sell"]).max() / scale < 1e-13
