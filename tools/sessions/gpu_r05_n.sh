#!/bin/bash
# round 5, session n: the library exactly as committed at the end of the round - smoke and the plain-C ABI test.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r05n"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke | tee "$OUT/smoke.log"
timeout 100 python -m pytest tests/test_abi.py -m gpu -q -rA 2>&1 | grep -E "PASSED|FAILED|passed|failed" | tee "$OUT/abi.log"
