#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r02ab; mkdir -p $O
STATE_STREAM=1 timeout 600 python tools/spmv_state.py 150 2>&1 | tee $O/state.txt | tail -20
