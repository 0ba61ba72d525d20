#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r02i"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== device partition tests"; timeout 900 python -m pytest tests/test_partition.py -m gpu -q -k "device" 2>&1 | tail -5
echo "== timing"; timeout 900 python tools/time_partition.py 150 8 --device 2>&1 | grep -v "^/opt" | tee "$OUT/time_partition.log"
echo "== load-step driver with device partition"; timeout 900 python -m pytest tests/test_partition.py -m gpu -q -k "load_step or mdf_to_solution" 2>&1 | tail -3
