#!/bin/bash
# One GPU-box session that regenerates the round-2 evidence (development): full parity suite incl. lock-step, smoke, bench
# (both operators, CPU baseline, live PMC traffic), rocprofv3 kernel stats of the bench command, the matrix-free kernel A/B.
# Outputs -> gpurun_out/<tag>/ ; copy what should be kept into profiles/.   usage: gpurun --timeout 3000 -- 'bash tools/gpu_session.sh [tag]'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; TAG="${1:-r02}"; OUT="$PWD/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{ nproc; cat /sys/fs/cgroup/cpu.max 2>&1; grep -m1 "model name" /proc/cpuinfo; } > "$OUT/host.txt"
rocm-smi --showclocks --showmaxpower --showpower --showmemorypartition --showcomputepartition --showperflevel > "$OUT/rocm_smi.txt" 2>&1
echo "== pytest -m gpu"; timeout 2400 python -X faulthandler -m pytest tests -m gpu -q -rA -s > "$OUT/pytest_gpu.log" 2>&1; grep -E "lock-step|passed|failed" "$OUT/pytest_gpu.log" | tail -4
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee "$OUT/smoke.log"
echo "== bench (+ live PMC traffic)"; timeout 1500 python bench.py --pmc-traffic > "$OUT/bench.json" 2> "$OUT/bench.log"; tail -2 "$OUT/bench.log"; cut -c1-400 "$OUT/bench.json"; echo
echo "== matrix-free kernel A/B"; timeout 900 python tools/ebe_lab.py 150 > "$OUT/ebe_lab.json" 2> "$OUT/ebe_lab.log"; grep -v "^/opt" "$OUT/ebe_lab.log" | cut -c1-220
cd /tmp
echo "== rocprofv3 kernel stats of the bench command"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -o k -- python "$R/bench.py" --no-cpu-baseline > "$OUT/prof_stats_bench.json" 2> "$OUT/prof_stats.log"
f=$(find "$OUT/prof_stats" -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-150
