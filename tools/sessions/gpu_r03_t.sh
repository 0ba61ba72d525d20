#!/bin/bash
# round 3, session t: split SELL format, sweep of the planner's parameters on the 10 M-dof octree mesh (tools/split_sweep.py)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r03t"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1000 python tools/split_sweep.py oct10m 60 0 1.5:512 3:512 2:128 3:128 3:64 2>&1 | grep stored_over_true | tee "$OUT/sweep_10m.log"
