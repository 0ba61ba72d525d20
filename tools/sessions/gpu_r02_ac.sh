#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r02ac; mkdir -p $O
timeout 600 python tools/spmv_ctx.py 150 2>&1 | tee $O/ctx_dot0.txt | tail -15
PCG_BENCH_SPMV_DOT=1 timeout 600 python tools/spmv_ctx.py 150 2>&1 | tee $O/ctx_dot1.txt | tail -15
