#!/bin/bash
# round 4, session t: evidence on the code at the end of the round (after the fragment ring of k_ebe_mixed, the symmetry-class octree
# workload in bench.py, k_ebe_mtile below 1.2 M elements) - the full GPU suite, smoke, the driver's bench command, the same command
# under rocprofv3 --kernel-trace --stats, the octree bench lines (1 M / 10 M dof).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04t"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{ nproc; cat /sys/fs/cgroup/cpu.max 2>&1; grep -m1 "model name" /proc/cpuinfo; } > "$OUT/host.txt"
rocm-smi --showclocks --showmaxpower --showpower --showmemorypartition --showcomputepartition --showperflevel > "$OUT/rocm_smi.txt" 2>&1
echo "== pytest -m gpu"; ( time timeout 2400 python -X faulthandler -m pytest tests -m gpu -q -rA -s > "$OUT/pytest_gpu.log" 2>&1 ) 2>&1 | grep real; grep -E "^(PASSED|FAILED|ERROR)|passed|failed" "$OUT/pytest_gpu.log" | grep -v "^PASSED" | tail -8; grep -E "lock-step|graded octree" "$OUT/pytest_gpu.log" | cut -c1-200 | tail -8
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee "$OUT/smoke.log"
echo "== the driver's bench command"; ( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.log" ) 2>&1 | grep real; cut -c1-300 "$OUT/bench_driver_cmd.json"; echo
echo "== octree bench lines"
for SZ in 1m 10m; do timeout 900 python bench.py --workload octree --octree-size $SZ --no-cpu-baseline > "$OUT/bench_octree_$SZ.json" 2> "$OUT/bench_octree_$SZ.log"; cut -c1-200 "$OUT/bench_octree_$SZ.json"; echo; done
cd /tmp
echo "== rocprofv3 kernel stats of the bench command"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -o k -- python "$R/bench.py" --no-cpu-baseline > "$OUT/prof_stats_bench.json" 2> "$OUT/prof_stats.log"
f=$(find "$OUT/prof_stats" -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-170
