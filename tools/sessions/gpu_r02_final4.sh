#!/bin/bash
# round 2, final evidence on the final code, part 2: the corrected dictionary test, bench.py as the driver calls it, and
# rocprofv3 kernel stats of the same command
set -x
R="$PWD"; OUT="$PWD/gpurun_out/r02final4"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_dictionary_format.py -m gpu -q -k "solve_on_gpu or larger_than_lds" 2>&1 | tail -4 > "$OUT/pytest_dict.log"; cat "$OUT/pytest_dict.log"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"; tail -3 "$OUT/bench.log"; cut -c1-600 "$OUT/bench.json"; echo
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -o k -- python "$R/bench.py" --no-cpu-baseline --steps 100 > "$OUT/prof_stats_bench.json" 2> "$OUT/prof_stats.log"
f=$(find "$OUT/prof_stats" -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-160
rm -f $(find "$OUT/prof_stats" -name "*kernel_trace.csv")
