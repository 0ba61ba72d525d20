#!/bin/bash
# round 4, session j: where a wave of k_ebe_mixed spends its lifetime - shader-clock stamps at the phase boundaries (PCG_EBE_STAMPS=1,
# the 40th fused-dot launch runs the STAMP instantiation) on the 10 M-dof octree mesh (symmetry classes) and on the brick (hex section only).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04j"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
PCG_EBE_STAMPS=1 timeout 600 python tools/iter_ab.py oct10ms ebe 100 "PCG_EBE_MIX_MTM=4" > "$OUT/st_oct10ms.json" 2> "$OUT/st_oct10ms.log"; grep -E "^\[pcg\]|us_per" "$OUT/st_oct10ms.log" | cut -c1-260
PCG_EBE_STAMPS=1 timeout 600 python tools/iter_ab.py oct1ms ebe 100 "PCG_EBE_MIX_MTM=4" > "$OUT/st_oct1ms.json" 2> "$OUT/st_oct1ms.log"; grep -E "^\[pcg\]|us_per" "$OUT/st_oct1ms.log" | cut -c1-260
PCG_EBE_STAMPS=1 PCG_EBE_MIXED=1 timeout 600 python tools/iter_ab.py 150 ebe 100 "PCG_EBE_MIX_MTM=4" > "$OUT/st_brick.json" 2> "$OUT/st_brick.log"; grep -E "^\[pcg\]|us_per" "$OUT/st_brick.log" | cut -c1-260
