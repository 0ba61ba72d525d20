#!/bin/bash
# round 4, session e: per-rank iteration time of the multi-part loop (fused / round-3 sequence) at 1.27 M dof, kernel trace.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04e"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tools/multi_part_iter.py 150 100 sell,ebe,dict > "$OUT/multi_part_iter.json" 2> "$OUT/multi_part_iter.log"; grep "^{" "$OUT/multi_part_iter.log" | cut -c1-240
cd /tmp
echo "== kernel trace of the same (fused only)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_multi" -o k -- python "$R/tools/multi_part_iter.py" 150 100 sell,ebe 1 > "$OUT/prof_multi.log" 2>&1
f=$(find "$OUT/prof_multi" -name "*kernel_stats.csv" | head -1); head -30 "$f" | cut -c1-150
