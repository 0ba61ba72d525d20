#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r02ae; mkdir -p $O
CTX_LIST=0,1,17,65,129,257 timeout 600 python tools/spmv_ctx.py 150 2>&1 | tee $O/ctx.txt | tail -15
