#!/bin/bash
# round 5, session j: `python -m pcg_mi355x.run --engine-side` - the load-step driver with the mailbox all-reduce and the direct exchange
# on 3 ranks sharing the GPU (one new test).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r05j"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PCG_MAIL_SPINS=300000
( time timeout 200 python -X faulthandler -m pytest tests -m gpu -q -rA -x -k "engine_side_forms" > "$OUT/pytest_run_engine_side.log" 2>&1 ) 2>&1 | grep real
grep -E "^(FAILED|ERROR|PASSED|SKIPPED)|passed|failed|declined|Error" "$OUT/pytest_run_engine_side.log" | cut -c1-300 | tail -12
