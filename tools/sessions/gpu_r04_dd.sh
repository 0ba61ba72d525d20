#!/bin/bash
# round 4, session dd: the full GPU suite, smoke and the driver's bench command on the FINAL code of the round (after k_ebe_mtile learnt
# oriented 8-node types and the tests the planner's own choice).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04dd"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest -m gpu"; ( time timeout 2400 python -X faulthandler -m pytest tests -m gpu -q -rA -s > "$OUT/pytest_gpu.log" 2>&1 ) 2>&1 | grep real; grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest_gpu.log" | tail -5
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee "$OUT/smoke.log"
echo "== the driver's bench command"; ( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.log" ) 2>&1 | grep real; cut -c1-260 "$OUT/bench_driver_cmd.json"; echo; grep -i "fail" "$OUT/bench_driver_cmd.log" | cut -c1-200
