#!/bin/bash
# round 4, session x: 1024-element chunks of the hex8 class (PCG_EBE_EPT=4: 16 x 8 x 8 cells, k_ebe_hexs with 512 threads, two
# workgroups of eight waves per CU) against the 512-element chunks - parity of the brick operator under the knob, same-process A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04x"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
( time PCG_EBE_EPT=4 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "ebe and not octree and not mixed" > "$OUT/pytest.log" 2>&1 ) 2>&1 | grep real; tail -3 "$OUT/pytest.log" | cut -c1-200
for N in 150 100 75; do
  PCG_EBE_STATS=1 timeout 600 python tools/iter_ab.py $N ebe 200 "PCG_EBE_EPT=2|4" > "$OUT/ab_$N.json" 2> "$OUT/ab_$N.log"; grep -E "us_per|ebe plan: [0-9]" "$OUT/ab_$N.log" | cut -c1-300
done
