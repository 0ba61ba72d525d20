#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r02ad; mkdir -p $O
PCG_BENCH_SPMV_CTX=8 timeout 600 python tools/prof_op.py sell 150 5 2>&1 | tee $O/placement.txt | grep -E "placement|median" 
PCG_BENCH_SPMV_CTX=8 timeout 600 python tools/prof_op.py sell 150 5 2>&1 | tee $O/placement2.txt | grep -E "placement|median"
