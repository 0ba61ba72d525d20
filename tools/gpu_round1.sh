#!/bin/bash
# One GPU-box session: parity tests, tuning sweep, bench, rocprof.  Outputs -> gpurun_out/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 > "$OUT/rocminfo.txt"
nproc > "$OUT/host.txt"; grep -m1 "model name" /proc/cpuinfo >> "$OUT/host.txt"; free -g | head -2 >> "$OUT/host.txt"
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee "$OUT/pytest_gpu.log"
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee "$OUT/smoke.log"
echo "== tune"; timeout 900 python tools/tune_spmv.py 150 > "$OUT/tune.json" 2> "$OUT/tune.log"; tail -12 "$OUT/tune.log"
echo "== bench"; timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"; tail -5 "$OUT/bench.log"; cat "$OUT/bench.json"
echo "== rocprof"; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o r1 -- python "$OLDPWD/bench.py" --steps 50 --warmup 5 --no-cpu-baseline > "$OUT/prof_bench.json" 2> "$OUT/prof_bench.log"); ls -R "$OUT/prof" | head -20
