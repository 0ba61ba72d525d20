#!/bin/bash
# round 5, session g: the engine-side neighbour exchange, second pass - parity between processes (incl. the one-phase matrix-free engine),
# then the per-rank iteration of the 1.32 M-dof part: two-phase engine (1 = RCCL, d, dm) and one-phase engine (1 = whole operator then
# RCCL exchange, d, dm).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r05g"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PCG_MAIL_SPINS=300000
echo "== parity"
( time timeout 420 python -X faulthandler -m pytest tests -m gpu -q -rA -x -k "direct_exchange" > "$OUT/pytest_direct.log" 2>&1 ) 2>&1 | grep real
grep -E "^(FAILED|ERROR|PASSED|SKIPPED)|passed|failed|refused|Error" "$OUT/pytest_direct.log" | cut -c1-300 | tail -20
echo "== per-rank iteration, two-phase engine: 1 / d / dm"
timeout 300 python tools/multi_part_iter.py 150 100 ebe 1,d,dm > "$OUT/mpi_direct_two_phase.json" 2> "$OUT/mpi_direct_two_phase.log"; echo rc=$?
grep -E "us_per_iter|Error|error" "$OUT/mpi_direct_two_phase.log" | cut -c1-220
echo "== per-rank iteration, one-phase engine: 1 / d / dm"
PCG_EBE_ONE_PHASE=1 timeout 300 python tools/multi_part_iter.py 150 100 ebe 1,d,dm > "$OUT/mpi_direct_one_phase.json" 2> "$OUT/mpi_direct_one_phase.log"; echo rc=$?
grep -E "us_per_iter|Error|error" "$OUT/mpi_direct_one_phase.log" | cut -c1-220
