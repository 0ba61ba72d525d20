#!/bin/bash
# round 3, session p (two calls): XCD-aware chunk order of the matrix-free kernels, PCG_EBE_XCD = 0 (chunk = workgroup index) | 1
# (contiguous eighths) | G (runs of G chunks per XCD): iteration A/B on the brick (1.27 M, 10 M dof) and the graded octree mesh
# (1 M, 10 M dof); ebe parity subset with the default
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r03p"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tools/iter_ab.py 75,150 ebe 200 "PCG_EBE_XCD=0|1" 2>&1 | grep us_per_iter | cut -c1-260 | tee "$OUT/ab_brick.log"
timeout 900 python tools/iter_ab.py oct1m,oct10m ebe 150 "PCG_EBE_XCD=0|1" 2>&1 | grep us_per_iter | cut -c1-260 | tee "$OUT/ab_octree.log"
timeout 900 python tools/iter_ab.py 150 ebe 200 "PCG_EBE_XCD=0|1|16|64" 2>&1 | grep us_per_iter | cut -c1-260 | tee "$OUT/ab_brick_groups.log"
timeout 900 python tools/iter_ab.py oct10m ebe 150 "PCG_EBE_XCD=0|1|16|64" 2>&1 | grep us_per_iter | cut -c1-260 | tee "$OUT/ab_octree_groups.log"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_irregular_meshes.py -m gpu -x -q -k "octree or graded or goct or fixture or irregular or mixed or ebe" 2>&1 | tail -3 | tee "$OUT/pytest.log"
