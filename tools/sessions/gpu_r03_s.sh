#!/bin/bash
# round 3, session s: split SELL format (base part + overflow part for the long rows): GPU tests, same-process A/B of the PCG
# iteration on the graded octree mesh (1 M, 10 M dof; PCG_SELL_SPLIT=0 = single matrix, 1 = split), the brick unchanged
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r03s"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_sell_split.py tests/test_gpu_parity.py -m gpu -x -q -k "split or spmv_kernel_vs_oracle or graded_octree_1m_dof or octree_mesh_with_hanging" 2>&1 | tail -4 | tee "$OUT/pytest.log"
timeout 900 python tools/iter_ab.py oct1m,oct10m sell 100 "PCG_SELL_SPLIT=0|1" 2>&1 | grep us_per_iter | grep -v "^\[{" | cut -c1-260 | tee "$OUT/ab_octree_sell.log"
timeout 600 python tools/iter_ab.py 75 sell 100 "PCG_SELL_SPLIT=0|1" 2>&1 | grep us_per_iter | grep -v "^\[{" | cut -c1-260 | tee "$OUT/ab_brick_sell.log"
