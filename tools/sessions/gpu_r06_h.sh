#!/bin/bash
# round 6, session h: the ablation with random operands - store flavours - and the product kernel on the SAME box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r06h"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tools/micro/spmv_ablation 150 4 1 2>&1 | grep "rep \|CUs\|operands" | tee "$OUT/spmv_ablation.log"
timeout 900 python tools/iter_ab.py 150 sell 100 "_=-" > "$OUT/iter.json" 2> "$OUT/iter.log"; grep "us_per_iter" "$OUT/iter.log" | cut -c1-260 | tee -a "$OUT/spmv_ablation.log"
