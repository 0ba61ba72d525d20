#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest -m gpu"; timeout 1200 python -X faulthandler -m pytest tests -m gpu -q -rA > "$OUT/pytest_gpu_full.log" 2>&1; tail -12 "$OUT/pytest_gpu_full.log"
echo "== bench"; timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"; cat "$OUT/bench.log" | tail -6; python -c "
import json; d=json.load(open('$OUT/bench.json')); print('sell', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']); print(json.dumps(d['matrix_free'], indent=1))"
cd /tmp
echo "== rocprof stats (both operators)"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats2" -o r1 -- python "$R/bench.py" --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/prof_stats2_bench.json" 2> "$OUT/prof_stats2.log"
head -12 "$OUT/prof_stats2/r1_kernel_stats.csv" | cut -c1-200
