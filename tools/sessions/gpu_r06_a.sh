#!/bin/bash
# round 6, session a: the driver's red test of round 5, reproduced in the driver's environment (no PCG_MAIL_SPINS override) with the whole
# output of every rank kept (PCG_TEST_LOG_DIR); then the full GPU suite in the driver's form (-x -q) with the engine-side tests collected last.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r06a"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PCG_TEST_LOG_DIR="$OUT/failed"
{ nproc; grep -m1 "model name" /proc/cpuinfo; rocm-smi --showmemorypartition --showcomputepartition 2>&1 | grep -i partition; } > "$OUT/host.txt" 2>&1
T=tests/test_zzz_engine_side.py::test_load_step_driver_with_the_engine_side_forms_on_gpu
for k in 1 2 3; do
  echo "== engine-side load-step test, default environment, run $k"
  ( time timeout 400 python -m pytest "$T" -x -q -m gpu > "$OUT/engine_side_default_$k.log" 2>&1 ) 2>&1 | grep real; tail -1 "$OUT/engine_side_default_$k.log"
done
echo "== same, look-ahead off"
PCG_LOOK_AHEAD=0 timeout 400 python -m pytest "$T" -x -q -m gpu > "$OUT/engine_side_no_lookahead.log" 2>&1; tail -1 "$OUT/engine_side_no_lookahead.log"
echo "== same, PCG_MAIL_SPINS=300000 (what every round-5 session exported)"
PCG_MAIL_SPINS=300000 timeout 400 python -m pytest "$T" -x -q -m gpu > "$OUT/engine_side_spins300k.log" 2>&1; tail -1 "$OUT/engine_side_spins300k.log"
ls "$OUT/failed" 2>/dev/null | head
for f in "$OUT"/failed/*.log; do [ -f "$f" ] && { echo "---- $f"; grep -v "^\s*$" "$f" | grep -iv "amdgpu.ids" | grep -i -B2 -A12 "error\|Traceback\|timed out\|never arrived" | head -80; }; done
echo "== pytest -m gpu -x -q (the driver's form)"
( time timeout 1700 python -m pytest tests/ -x -q -m gpu > "$OUT/pytest_gpu_x.log" 2>&1 ) 2>&1 | grep real; tail -5 "$OUT/pytest_gpu_x.log" | cut -c1-300
