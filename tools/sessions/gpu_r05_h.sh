#!/bin/bash
# round 5, session h: the code at the END of the round (after the engine-side exchange went in) - (1) the full GPU suite, (2) smoke,
# (3) the driver's bench command, (4) the same command under rocprofv3 --kernel-trace --stats.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r05h"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{ nproc; cat /sys/fs/cgroup/cpu.max 2>&1; grep -m1 "model name" /proc/cpuinfo; } > "$OUT/host.txt"
echo "== pytest -m gpu"; ( time timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -rA -s > "$OUT/pytest_gpu.log" 2>&1 ) 2>&1 | grep real; grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest_gpu.log" | tail -8 | cut -c1-250
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke | tee "$OUT/smoke.log"
echo "== the driver's bench command"; ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.log" ) 2>&1 | grep real; cut -c1-260 "$OUT/bench_driver_cmd.json"; echo; grep -i "fail" "$OUT/bench_driver_cmd.log" | cut -c1-200
cd /tmp
echo "== rocprofv3 kernel stats of the bench command"
( time timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -o k -- python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc-traffic > "$OUT/bench_under_rocprof.json" 2> "$OUT/prof_stats.log" ) 2>&1 | grep real
f=$(find "$OUT/prof_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp "$f" "$OUT/bench_kernel_stats.csv"; head -12 "$f" | cut -c1-170; }
rm -rf "$OUT/prof_stats"
