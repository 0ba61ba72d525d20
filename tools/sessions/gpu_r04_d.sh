#!/bin/bash
# round 4, session d: the multi-part iteration in five launches: parity (fused vs round-3 sequence, thread communicator, the RCCL
# stand-in, real RCCL at world size 1), per-rank iteration time at 1.27 M dof on real RCCL with a self-loop exchange, kernel trace
# of that run (launch count per iteration).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04d"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest: multi-part"
timeout 1200 python -X faulthandler -m pytest tests -m gpu -q -x -k "fused_multi_part or multi_part_kernels or native_comm or group_on_one_gpu or eight_parts or nccl_hooks or split_matrix_multi" > "$OUT/pytest_multi.log" 2>&1; tail -5 "$OUT/pytest_multi.log"
echo "== per-rank iteration, 1.27 M dof part of the 2x2x2 split, real RCCL world 1 + self-loop exchange"
timeout 900 python tools/multi_part_iter.py 150 100 sell,ebe,dict > "$OUT/multi_part_iter.json" 2> "$OUT/multi_part_iter.log"; grep "^{" "$OUT/multi_part_iter.log" | cut -c1-240
cd /tmp
echo "== kernel trace of the same (fused only)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_multi" -o k -- python "$R/tools/multi_part_iter.py" 150 100 sell,ebe 1 > "$OUT/prof_multi.log" 2>&1
f=$(find "$OUT/prof_multi" -name "*kernel_stats.csv" | head -1); head -24 "$f" | cut -c1-160
