#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r02aa; mkdir -p $O
(while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory)" | tr '\n' ' '; echo; sleep 2; done) > $O/smi_series.txt &
SMI=$!
timeout 600 python tools/spmv_drift.py 150 40 200 2>&1 | tee $O/drift.txt | tail -45
kill $SMI
tail -30 $O/smi_series.txt | cut -c1-400
