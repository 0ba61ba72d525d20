#!/bin/bash
# round 6, session j: HOLD at 10 M dof: off / 4 launches of 4 slots / ONE launch, a wave flushes whenever its 4 slots are full
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r06j"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/iter_ab.py 150 sell 200 "PCG_SPMV_HOLD+PCG_SPMV_HOLD_SPLIT=0+1|4+1|4+0" > "$OUT/ab_hold_150.json" 2> "$OUT/ab_hold_150.log"; echo rc=$?
grep "us_per_iter" "$OUT/ab_hold_150.log" | cut -c1-260
