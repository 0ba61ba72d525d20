#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r02af; mkdir -p $O
timeout 600 python tools/vec_nt_ab.py 150 200 2>&1 | tee $O/ab150.txt | grep "N="
timeout 600 python tools/vec_nt_ab.py 70 300 2>&1 | tee $O/ab70.txt | grep "N="
timeout 600 python tools/vec_nt_ab.py 31 300 2>&1 | tee $O/ab31.txt | grep "N="
