#!/bin/bash
# round 6, session f: the software-pipelined k_spmv_p (one exposed memory latency per group of three block columns instead of two) against
# k_spmv, same process, 10 M / 1.27 M dof brick and the octree meshes; parity subset with the new default.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r06f"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tools/micro/stream_patterns 4 2>&1 | head -5 | tee "$OUT/stream_patterns.log"
echo "== A/B 10 M dof"
timeout 600 python tools/iter_ab.py 150 sell 200 "PCG_SPMV_PIPE=0|1" > "$OUT/ab_spmv_pipe_150.json" 2> "$OUT/ab_spmv_pipe_150.log"; echo rc=$?
grep "us_per_iter" "$OUT/ab_spmv_pipe_150.log" | cut -c1-260
echo "== A/B 1.27 M dof brick, octree 1 M / 10 M (base + overflow)"
timeout 600 python tools/iter_ab.py 75,oct1ms,oct10ms sell 200 "PCG_SPMV_PIPE=0|1" > "$OUT/ab_spmv_pipe_small.json" 2> "$OUT/ab_spmv_pipe_small.log"; echo rc=$?
grep "us_per_iter" "$OUT/ab_spmv_pipe_small.log" | cut -c1-260
echo "== parity subset, new default"
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_sell_split.py tests/test_lockstep.py -x -q -m gpu -k "not ebe and not octree_solve_in_lock_step" > "$OUT/pytest_subset.log" 2>&1 ) 2>&1 | grep real; tail -3 "$OUT/pytest_subset.log" | cut -c1-300
