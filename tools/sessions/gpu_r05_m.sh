#!/bin/bash
# round 5, session m: after the per-iteration check of the engine-side waits went into the driver (host code only) - smoke and a multi-part subset.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r05m"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PCG_MAIL_SPINS=300000
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke | tee "$OUT/smoke.log"
( time timeout 150 python -X faulthandler -m pytest tests -m gpu -q -rA -x -k "(multi_part_kernels and n9_p8) or (direct_exchange and oct_p3) or (mailbox_reduction_between and oct_p3) or fused_multi_part" > "$OUT/pytest_subset.log" 2>&1 ) 2>&1 | grep real
grep -E "^(FAILED|ERROR|PASSED|SKIPPED)|passed|failed|Error" "$OUT/pytest_subset.log" | cut -c1-300 | tail -12
