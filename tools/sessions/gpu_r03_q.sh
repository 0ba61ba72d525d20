#!/bin/bash
# round 3, session q: the new agreement test (hanging-node kernels with / without node tile); hex8 chunk size (256- vs 512-element
# chunks, PCG_EBE_EPT=1|2) at small sizes: 0.43 M, 1.27 M, 3.4 M dof
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r03q"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "node_tile_agree" 2>&1 | tail -5 | tee "$OUT/pytest.log"
timeout 900 python tools/iter_ab.py 52,75,104 ebe 300 "PCG_EBE_EPT=2|1" 2>&1 | grep us_per_iter | cut -c1-260 | tee "$OUT/ab_ept.log"
