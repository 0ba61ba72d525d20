#!/usr/bin/env python
"""Set-up timing (development tool, build container): pcg_mi355x.partition vs the reference's partition_mesh.py
on the same synthetic MDF model.  usage: python tools/time_partition.py N n_parts [--ref]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pcg-mpi-solver_amd"), os.path.join(ROOT, "oracle")]
import numpy as np
from pcg_mi355x import mdf, partition
from pcg_mi355x.brick import Brick, block_partition, default_grid

N, n_parts = int(sys.argv[1]), int(sys.argv[2])
b = Brick(N)
t0 = time.time(); model = mdf.model_from_brick(b); ep = block_partition(b, *default_grid(n_parts)); t_model = time.time() - t0
t0 = time.time(); parts = partition.partition_model(model, ep); t_all = time.time() - t0
t0 = time.time(); one = partition.partition_model(model, ep, only=[n_parts - 1]); t_one = time.time() - t0
print(f"N={N} dof={b.n_dof} elements={b.n_elem} parts={n_parts}: model {t_model:.2f}s, partition_model all parts {t_all:.2f}s, "
      f"one part (only=[rank]) {t_one:.2f}s", flush=True)
if "--device" in sys.argv:                     # the index passes as HIP kernels (csrc/part_setup.hip); first call pays library load
    partition.partition_model(model, ep, only=[0], device=0)
    t0 = time.time(); dev = partition.partition_model(model, ep, only=[n_parts - 1], device=0); t_dev = time.time() - t0
    same = all(np.array_equal(one[0][k], dev[0][k]) for k in ("DofVector", "NodeIdVector", "DofWeightVector", "Flat_ElemLocDof"))
    print(f"one part with the index passes on the GPU: {t_dev:.2f}s (identical: {same})", flush=True)
if "--ref" in sys.argv:
    import ref_shim
    work = tempfile.mkdtemp()
    mp = os.path.join(work, "MDF", "")
    mdf.write_mdf(mp, model); mdf.write_mesh_part(mp, ep)
    t0 = time.time()
    ref_shim.ref_partition(work, mp, os.path.join(work, "MPI", ""), n_parts)
    print(f"reference partition_mesh.py (1 worker, incl. file read + export): {time.time() - t0:.2f}s", flush=True)
