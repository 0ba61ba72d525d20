#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r02ah; mkdir -p $O
for i in 1 2; do
  echo "== default allocation" | tee -a $O/log.txt
  timeout 600 python tools/prof_op.py sell 150 40 2>&1 | tail -1 | cut -c90- | tee -a $O/log.txt
  echo "== PCG_ALLOC_CONTIG=1" | tee -a $O/log.txt
  PCG_ALLOC_CONTIG=2 timeout 600 python tools/prof_op.py sell 150 40 2>&1 | grep -E "contiguous|median" | cut -c1-60,90-170 | tee -a $O/log.txt
done
