#!/bin/bash
# round 6, FINAL session - after the last product commit (its hash is the first line of every log): (1) the driver's bench command on the
# fresh box, (2) the full GPU suite in the driver's form (-x -q), (3) smoke, (4) the driver's bench command again (warm box: same box,
# 15 minutes of kernels later), (5) the same command under rocprofv3 --kernel-trace --stats, (6) bench.py --full (every object of rounds
# 1 - 5 in the full record), (7) the N = 2 line on the shared GPU.   usage: gpurun --timeout 3300 -- 'bash tools/sessions/gpu_r06_final.sh <commit>'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; HEAD="${1:-unknown}"; OUT="$PWD/gpurun_out/r06final"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PCG_TEST_LOG_DIR="$OUT/failed"
{ echo "commit $HEAD"; nproc; cat /sys/fs/cgroup/cpu.max 2>&1; grep -m1 "model name" /proc/cpuinfo; rocm-smi --showclocks --showmaxpower --showpower --showmemorypartition --showcomputepartition --showperflevel 2>&1 | grep -v "^=\|^$"; } > "$OUT/host.txt" 2>&1
bench() {  # $1 = tag
  echo "commit $HEAD" > "$OUT/bench_$1.log"
  ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_$1.json" 2>> "$OUT/bench_$1.log" ) 2>&1 | grep real
  wc -c "$OUT/bench_$1.json"; python - "$OUT/bench_$1.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, c = d["roofline"], d["cpu_baseline"]
print(f"value {d['value']:.1f} it/s; k_spmv {r['avg_launch_ms']:.4f} ms frac {r['frac']:.3f} traffic/bytes {r.get('traffic_over_bytes')} scalar_csr {r.get('scalar_csr_frac')}; cpu {c['value']:.1f} it/s on {c['cores']}; also {d.get('also')}")
PY
  cp bench_extras.json "$OUT/bench_extras_$1.json" 2>/dev/null
}
echo "== (1) the driver's bench command, fresh box"; bench cold
echo "== (2) pytest tests/ -x -q -m gpu"
echo "commit $HEAD" > "$OUT/pytest_gpu_HEAD.log"
( time timeout 1800 python -m pytest tests/ -x -q -m gpu -rfEs >> "$OUT/pytest_gpu_HEAD.log" 2>&1 ) 2>&1 | grep real; tail -4 "$OUT/pytest_gpu_HEAD.log" | cut -c1-300
echo "== (3) smoke"; { echo "commit $HEAD"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke; } | tee "$OUT/smoke.log"
echo "== (4) the driver's bench command, warm box"; bench warm
echo "== (5) rocprofv3 --kernel-trace --stats of the bench command"
cd /tmp
( time timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -o k -- python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc-traffic > "$OUT/bench_under_rocprof.json" 2> "$OUT/prof_stats.log" ) 2>&1 | grep real
f=$(find "$OUT/prof_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp "$f" "$OUT/bench_kernel_stats.csv"; head -14 "$f" | cut -c1-170; }
rm -rf "$OUT/prof_stats"
cd "$R"
echo "== (6) bench.py --full"
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 --full > "$OUT/bench_full.json" 2> "$OUT/bench_full.log" ) 2>&1 | grep real; cp bench_extras.json "$OUT/bench_extras_full.json" 2>/dev/null; wc -c "$OUT/bench_full.json" "$OUT/bench_extras_full.json"
echo "== (7) bench.py --gpus 2, both ranks on this GPU (RCCL stand-in)"
FAKE=$(python -c "import sys; sys.path.insert(0,'tests'); import conftest; print(conftest.build_fakenccl())" 2>/dev/null | tail -1)
( time PCG_BENCH_SHARE_GPU=1 PCG_RCCL_LIB="$FAKE" timeout 800 python bench.py --gpus 2 --steps 20 --warmup 5 > "$OUT/bench_2ranks_shared.json" 2> "$OUT/bench_2ranks_shared.log" ) 2>&1 | grep real; tail -1 "$OUT/bench_2ranks_shared.json" | cut -c1-600; echo
ls gpurun_out/bench_extras_n2_* 2>/dev/null | tail -1
