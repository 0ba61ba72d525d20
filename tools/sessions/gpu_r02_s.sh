#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_native_comm.py tests/test_gpu_parity.py -m gpu -q -k "real_rccl or irregular" -rA 2>&1 | tail -12
