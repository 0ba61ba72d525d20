#!/bin/bash
# round 6, session i: k_spmv with y written at the end of the launch from LDS (HOLD; the slice range in as many launches as a wave's share
# needs slots) against the single launch with stores as they come - same process, several sizes; parity subset with the new default.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r06i"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tools/micro/spmv_ablation 150 4 1 2>&1 | grep "rep 1 [ADEOP]\|operands" | tee "$OUT/spmv_ablation.log"
echo "== A/B 10 M dof"
timeout 600 python tools/iter_ab.py 150 sell 200 "PCG_SPMV_HOLD=0|1" > "$OUT/ab_hold_150.json" 2> "$OUT/ab_hold_150.log"; echo rc=$?
grep "us_per_iter" "$OUT/ab_hold_150.log" | cut -c1-260
echo "== A/B 1.27 M dof brick, octree 1 M / 10 M (base + overflow)"
timeout 600 python tools/iter_ab.py 75,oct1ms,oct10ms sell 200 "PCG_SPMV_HOLD=0|1" > "$OUT/ab_hold_small.json" 2> "$OUT/ab_hold_small.log"; echo rc=$?
grep "us_per_iter" "$OUT/ab_hold_small.log" | cut -c1-260
echo "== parity subset, new default"
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_sell_split.py tests/test_lockstep.py -x -q -m gpu -k "not ebe and not octree_solve_in_lock_step" > "$OUT/pytest_subset.log" 2>&1 ) 2>&1 | grep real; tail -3 "$OUT/pytest_subset.log" | cut -c1-300
