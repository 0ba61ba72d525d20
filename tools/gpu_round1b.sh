#!/bin/bash
# GPU session 2: full parity log, SpMV A/B sweep, bench for both slice heights, rocprof stats + PMC passes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest -m gpu"; timeout 900 python -X faulthandler -m pytest tests -m gpu -q -rA > "$OUT/pytest_gpu_full.log" 2>&1; tail -4 "$OUT/pytest_gpu_full.log"
echo "== tune"; timeout 900 python tools/tune_spmv.py 150 > "$OUT/tune2.json" 2> "$OUT/tune2.log"; grep rpl "$OUT/tune2.log"
echo "== bench rpl1"; timeout 900 python bench.py --rows-per-lane 1 > "$OUT/bench_rpl1.json" 2> "$OUT/bench_rpl1.log"; cut -c1-400 "$OUT/bench_rpl1.json"
echo "== bench rpl2"; timeout 900 python bench.py --rows-per-lane 2 --no-cpu-baseline > "$OUT/bench_rpl2.json" 2> "$OUT/bench_rpl2.log"; cut -c1-400 "$OUT/bench_rpl2.json"
cd /tmp
echo "== rocprof stats"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -o r1 -- python "$R/bench.py" --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/prof_stats_bench.json" 2> "$OUT/prof_stats.log"
ls -R "$OUT/prof_stats" | head; 
echo "== rocprof pmc FETCH_SIZE"
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/prof_fetch" -o r1 -- python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-finish > "$OUT/prof_fetch_bench.json" 2> "$OUT/prof_fetch.log"
echo "== rocprof pmc WRITE_SIZE"
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/prof_write" -o r1 -- python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-finish > "$OUT/prof_write_bench.json" 2> "$OUT/prof_write.log"
cd "$R"
for d in prof_fetch prof_write; do f=$(ls $OUT/$d/*.db 2>/dev/null | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" "$OUT/$d/summary.md" && tail -25 "$OUT/$d/summary.md"; done
du -sh "$OUT"/* | tail -20
