#!/bin/bash
# round 3, session k: does re-allocating the value array inside ONE process reach the fast placement?  (PCG_SPMV_PLACEMENTS)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r03k"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PCG_SPMV_PLACEMENTS_LOG=1
for rep in 1 2 3 4; do
  PCG_SPMV_PLACEMENTS=4 timeout 300 python tools/prof_op.py sell 150 20 2>&1 | grep -E "placements|median" | sed "s/^/process $rep: /"
done | tee "$OUT/placements.txt"
echo "== in the loop, probe on (3) / off (1)"
for p in 3 1 3 1; do
  PCG_SPMV_PLACEMENTS=$p timeout 400 python tools/iter_ab.py 150 sell 200 "PCG_VEC_FUSED=1" 2>&1 | grep -E "placements|us_per_iter" | cut -c1-230 | sed "s/^/placements=$p: /"
done | tee "$OUT/placements_loop.txt"
