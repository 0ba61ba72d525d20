#!/bin/bash
# round 6, session e: which kind of box is this (k_spmv 1.03 or 1.20 ms), and what does its HBM give the value stream of k_spmv depending on
# how the concurrent waves are laid over the array (tools/micro/stream_patterns); then the in-kernel pattern switches at 10 M dof.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r06e"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for B in 4 8 2; do tools/micro/stream_patterns $B 2>&1 | tee -a "$OUT/stream_patterns.log"; done
timeout 600 python tools/iter_ab.py 150 sell 200 "PCG_SPMV_PATTERN=0|8|4" > "$OUT/ab_spmv_pattern.json" 2> "$OUT/ab_spmv_pattern.log"; echo rc=$?
grep "us_per_iter" "$OUT/ab_spmv_pattern.log" | cut -c1-260
tools/micro/stream_patterns 4 2>&1 | tee -a "$OUT/stream_patterns.log"
