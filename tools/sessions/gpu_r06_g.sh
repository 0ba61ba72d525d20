#!/bin/bash
# round 6, session g: where k_spmv loses against its value stream (tools/micro/spmv_ablation), next to the kernel itself on the same box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r06g"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tools/micro/spmv_ablation 150 4 2>&1 | tee "$OUT/spmv_ablation.log"
tools/micro/spmv_ablation 150 8 2>&1 | tee -a "$OUT/spmv_ablation.log"
timeout 600 python tools/iter_ab.py 150 sell 100 "PCG_VEC_NT=5|7" > "$OUT/iter.json" 2> "$OUT/iter.log"; grep "us_per_iter" "$OUT/iter.log" | cut -c1-260
