#!/bin/bash
# round 5, session l: after pcg_group_enable_direct_exchange went into the library - smoke, the group / refusal tests, one direct-exchange case.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r05l"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PCG_MAIL_SPINS=300000
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke | tee "$OUT/smoke.log"
( time timeout 170 python -X faulthandler -m pytest tests -m gpu -q -rA -x -k "declined or (direct_exchange and oct_p3) or test_abi" > "$OUT/pytest_subset.log" 2>&1 ) 2>&1 | grep real
grep -E "^(FAILED|ERROR|PASSED|SKIPPED)|passed|failed|Error" "$OUT/pytest_subset.log" | cut -c1-300 | tail -12
