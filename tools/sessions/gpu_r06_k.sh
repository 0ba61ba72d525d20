#!/bin/bash
# round 6, session k: HOLD on / off, stand-alone against in the PCG loop, same process
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r06k"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tools/spmv_hold_ab.py 150 2>&1 | grep -v amdgpu.ids | tee "$OUT/spmv_hold_ab.log"
