#!/bin/bash
# round 6, session q: the driver's bench command, once, on whatever box this is (run on several boxes: the spread on the final code)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r06q"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PCG_VEC_PLACEMENT_LOG=1
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.log" ) 2>&1 | grep real
tail -1 "$OUT/bench.json" | cut -c1-900; echo; grep "placement:\|k_spmv:" "$OUT/bench.log" | head -2 | cut -c1-200
