#!/bin/bash
# round-2 session B: new parity tests (lock-step 1 M, load-step driver, mixed layout), rocprofv3 stats + PMC of the bench command
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r02b"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{ nproc; cat /sys/fs/cgroup/cpu.max 2>&1; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>&1; lscpu | grep -E "Socket|Core|Thread|NUMA node\(s\)|Model name"; free -g | head -2; } > "$OUT/host.txt" 2>&1
echo "== new tests"; timeout 1500 python -X faulthandler -m pytest tests/test_lockstep.py tests/test_partition.py::test_load_step_driver_on_gpu tests/test_gpu_parity.py::test_mixed_chunked_and_colour_groups_with_neighbours_on_gpu -m gpu -q -rA -s > "$OUT/pytest_new.log" 2>&1; grep -E "lock-step|passed|failed|PASSED|FAILED|Error" "$OUT/pytest_new.log" | tail -20
cd /tmp
echo "== rocprofv3 kernel stats of the bench command"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -o k -- python "$R/bench.py" --no-cpu-baseline > "$OUT/prof_stats_bench.json" 2> "$OUT/prof_stats.log"
f=$(find "$OUT/prof_stats" -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-170
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprofv3 --pmc $c"
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc_$c" -o k -- python "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-finish > "$OUT/pmc_${c}_bench.json" 2> "$OUT/pmc_$c.log"
  f=$(find "$OUT/pmc_$c" -name "*.db" | head -1); [ -n "$f" ] && python "$R/tools/rocpd_summary.py" "$f" "$OUT/pmc_$c/summary.md" && grep -E "k_spmv|k_ebe|k_fused|k_update" "$OUT/pmc_$c/summary.md" | grep "$c"
done
cd "$R"
echo "== bench (driver command)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.log"; cut -c1-300 "$OUT/bench_driver_cmd.json"; echo
