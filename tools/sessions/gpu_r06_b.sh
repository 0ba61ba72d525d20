#!/bin/bash
# round 6, session b: (1) the driver's bench command on the new bench.py - compact line, wall time, extras file; (2) the N = 2 line on the
# shared GPU; (3) the opt-in engine-side tests on the rebuilt library (wall-clock time-out, collective sync at solve begin).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r06b"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PCG_TEST_LOG_DIR="$OUT/failed"
echo "== the driver's bench command"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.log" ) 2>&1 | grep real
wc -c "$OUT/bench_driver_cmd.json"; tail -1 "$OUT/bench_driver_cmd.json"; echo; grep -i "fail\|skipping\|\`" "$OUT/bench_driver_cmd.log" | cut -c1-220
cp bench_extras.json "$OUT/bench_extras_driver_cmd.json" 2>/dev/null
echo "== bench N = 2 on the shared GPU + native comm tests"
( time timeout 900 python -m pytest tests/test_native_comm.py -x -q -m gpu -k "bench_launches or parts_as_processes" > "$OUT/pytest_bench_n2.log" 2>&1 ) 2>&1 | grep real; tail -3 "$OUT/pytest_bench_n2.log" | cut -c1-300
echo "== engine-side tests"
( time timeout 1200 python -m pytest tests/test_zzz_engine_side.py -q -m gpu -rA > "$OUT/pytest_engine_side.log" 2>&1 ) 2>&1 | grep real; grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest_engine_side.log" | tail -8 | cut -c1-250
for f in "$OUT"/failed/*.log; do [ -f "$f" ] && { echo "---- $f"; grep -iv "amdgpu.ids" "$f" | grep -i -B2 -A12 "error\|Traceback\|timed out\|never arrived" | head -60; }; done
