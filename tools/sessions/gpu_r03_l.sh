#!/bin/bash
# round 3, session l: in-loop SpMV time over separate processes - default allocation vs a physically contiguous value array
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r03l"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2 3 4; do
  for mode in 0 3; do
    PCG_ALLOC_CONTIG=$mode timeout 300 python tools/iter_ab.py 150 sell 150 "PCG_VEC_FUSED=1" 2>&1 | grep -E "contiguous|us_per_iter" | head -2 | cut -c1-220 | sed "s/^/contig=$mode process $rep: /"
  done
done | tee "$OUT/contig_vals.txt"
