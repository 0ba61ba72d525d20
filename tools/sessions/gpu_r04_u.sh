#!/bin/bash
# round 4, session u: the driver's bench command once more after the PMC segmenter learnt the name k_ebe_mtile (session t's line had
# fallen back to the stored traffic figures), and the planner statistics of the two octree sizes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04u"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.log" ) 2>&1 | grep real; cut -c1-300 "$OUT/bench_driver_cmd.json"; echo; grep -i "pmc\|fail" "$OUT/bench_driver_cmd.log" | cut -c1-250
