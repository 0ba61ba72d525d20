#!/bin/bash
# round 2, final evidence on the final code (contiguous-VRAM request off by default): bench.py as the driver calls it + rocprofv3 kernel stats
set -x
R="$PWD"; OUT="$PWD/gpurun_out/r02final5"; mkdir -p "$OUT"
export TMPDIR=/tmp
rocm-smi --showclocks --showpower --showcomputepartition --showmemorypartition > "$OUT/rocm_smi.txt" 2>&1
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"; tail -2 "$OUT/bench.log"; cut -c1-300 "$OUT/bench.json"; echo
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -o k -- python "$R/bench.py" --no-cpu-baseline --steps 100 > "$OUT/prof_stats_bench.json" 2> "$OUT/prof_stats.log"
f=$(find "$OUT/prof_stats" -name "*kernel_stats.csv" | head -1); head -6 "$f" | cut -c1-120
rm -f $(find "$OUT/prof_stats" -name "*kernel_trace.csv")
