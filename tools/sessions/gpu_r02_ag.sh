#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r02ag; mkdir -p $O
NT_LIST=0,1,3,5,7 timeout 900 python tools/vec_nt_ab.py 150 200 2>&1 | tee $O/ab150.txt | grep "N="
