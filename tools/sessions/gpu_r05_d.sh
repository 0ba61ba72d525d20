#!/bin/bash
# round 5, session d: (1) the interior phase of the split matrix-free apply on a side stream (PCG_EBE_PHASE_STREAMS, the default):
# multi-part parity subset, then the per-rank iteration of the 1.32 M-dof part with and without; (2) mailbox tests after the change
# that declines the mailboxes when ranks of one process share a device.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r05d"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== multi-part parity subset (phase streams on)"
( time timeout 500 python -X faulthandler -m pytest tests -m gpu -q -rA -x -k "multi_part or parts_as_processes or mailbox" > "$OUT/pytest_mp.log" 2>&1 ) 2>&1 | grep real
grep -E "^(FAILED|ERROR|PASSED|SKIPPED)|passed|failed" "$OUT/pytest_mp.log" | cut -c1-260 | tail -30
echo "== per-rank iteration, phase streams 1 / 0 / 1 / 0"
for PS in 1 0 1 0; do
PCG_EBE_PHASE_STREAMS=$PS timeout 200 python tools/multi_part_iter.py 150 100 ebe 1,m > "$OUT/mpi_ps$PS.json" 2>> "$OUT/mpi_ps.log"; echo "PS=$PS rc=$?"
python - "$OUT/mpi_ps$PS.json" <<'P'
import json,sys
for r in json.load(open(sys.argv[1])): print("   ", r["kind"], "mode", r["PCG_ITER_FUSED"], "rep", r["rep"], "%.1f us" % r["us_per_iter"])
P
done
