#!/bin/bash
# One GPU-box session (development): full parity suite, bench (both operators + CPU baseline), rocprofv3
# kernel stats and the PMC passes for the HBM traffic of the SpMV.  Outputs -> gpurun_out/ ; copy what
# should be kept into profiles/.   usage: gpurun --timeout 2400 -- 'bash tools/gpu_session.sh [tag]'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; TAG="${1:-r}"; OUT="$PWD/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
nproc > "$OUT/host.txt"; grep -m1 "model name" /proc/cpuinfo >> "$OUT/host.txt"; rocminfo 2>/dev/null | grep -E "gfx9|Compute Unit" | head -4 >> "$OUT/host.txt"
echo "== pytest -m gpu"; timeout 1200 python -X faulthandler -m pytest tests -m gpu -q -rA > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$OUT/smoke.log"
echo "== bench"; timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"; tail -3 "$OUT/bench.log"; cut -c1-600 "$OUT/bench.json"; echo
cd /tmp
echo "== rocprofv3 kernel stats"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -o k -- python "$R/bench.py" --no-cpu-baseline > "$OUT/prof_stats_bench.json" 2> "$OUT/prof_stats.log"
head -14 "$OUT/prof_stats/k_kernel_stats.csv" | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprofv3 --pmc $c"
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc_$c" -o k -- python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-finish > "$OUT/pmc_${c}_bench.json" 2> "$OUT/pmc_$c.log"
  f=$(ls "$OUT/pmc_$c"/*.db 2>/dev/null | head -1); [ -n "$f" ] && python "$R/tools/rocpd_summary.py" "$f" "$OUT/pmc_$c/summary.md" && grep -E "k_spmv|k_ebe|k_fused|k_update" "$OUT/pmc_$c/summary.md" | grep "$c"
done
cd "$R"
echo "== bench 1M dof (BASELINE configs[1])"; timeout 600 python bench.py --nodes-per-side 70 --steps 300 > "$OUT/bench_n70.json" 2> "$OUT/bench_n70.log"; cut -c1-300 "$OUT/bench_n70.json"; echo
