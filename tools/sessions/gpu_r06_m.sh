#!/bin/bash
# round 6, session m: the new bit-identity test, the SpMV-related parity subset and the driver's bench command on the tuner code
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r06m"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spmv_forms or full_size or scalar_copy or dictionary" > "$OUT/pytest_new.log" 2>&1 ) 2>&1 | grep real; tail -3 "$OUT/pytest_new.log" | cut -c1-300
( time PCG_VEC_PLACEMENT_LOG=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.log" ) 2>&1 | grep real
tail -1 "$OUT/bench.json" | cut -c1-1300; echo; grep "placement\|k_spmv:\|headline\|scalar" "$OUT/bench.log" | cut -c1-260
