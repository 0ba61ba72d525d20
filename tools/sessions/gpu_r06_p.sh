#!/bin/bash
# round 6, session p: vector placement (ranked once as y, every kind of operator) on / off: iteration, operator and vector-phase times
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r06p"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PCG_VEC_PLACEMENT_LOG=1
timeout 900 python tools/iter_ab.py 150 sell,ebe 200 "PCG_VEC_PLACEMENT=0|1" 2>&1 | grep "placement:\|k_spmv:\|us_per_iter" | grep -v "^\[{" | cut -c1-250 | tee "$OUT/ab_placement.log"
