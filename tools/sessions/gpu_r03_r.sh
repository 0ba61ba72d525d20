#!/bin/bash
# round 3, session r2: hex8 chunk size on the graded octree mesh (512- vs 256-element chunks, PCG_EBE_EPT=2|1), 1 M and 10 M dof
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r03r"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
PCG_EBE_STATS=1 timeout 900 python tools/iter_ab.py oct1m,oct10m ebe 150 "PCG_EBE_EPT=2|1" 2>&1 | grep -E "us_per_iter|ebe plan" | grep -v "^\[{" | cut -c1-260 | tee "$OUT/ab_octree_ept.log"
