#!/bin/bash
# round 6, session r: on a box where every vector candidate is slow - does another placement of the 6.9 GB VALUE ARRAY help (round 3's probe
# PCG_SPMV_PLACEMENTS=k at upload, then the round-6 vector placement on top)?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r06r"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PCG_VEC_PLACEMENT_LOG=1 PCG_SPMV_PLACEMENTS_LOG=1
timeout 900 python tools/iter_ab.py 150 sell 100 "PCG_SPMV_PLACEMENTS=1|4" 2>&1 | grep "placement\|k_spmv:\|us_per_iter" | grep -v "^\[{" | cut -c1-230 | tee "$OUT/ab.log"
