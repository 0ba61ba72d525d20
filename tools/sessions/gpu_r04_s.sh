#!/bin/bash
# round 4, session s: where k_ebe_mtile (hex tiles) stops winning against the hex section of k_ebe_mixed - octree meshes of 1 / 1.5 / 2.2 /
# 3.3 M dof (symmetry classes), same process A/B; and the brick below 600 k elements (N = 75: 1.27 M dof - a GPU's share of 10 M dof on 8).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04s"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for M in oct1ms oct2ms oct3ms oct5ms; do
  timeout 600 python tools/iter_ab.py $M ebe 200 "PCG_EBE_HEX_TILES=0|1" > "$OUT/ab_$M.json" 2> "$OUT/ab_$M.log"; grep -E "us_per" "$OUT/ab_$M.log" | grep "'rep': 1" | cut -c1-300
done
for N in 75 100; do
  timeout 600 python tools/iter_ab.py $N ebe 200 "PCG_EBE_MIXED+PCG_EBE_HEX_TILES=0+0|1+1|1+0" > "$OUT/ab_$N.json" 2> "$OUT/ab_$N.log"; grep -E "us_per" "$OUT/ab_$N.log" | grep "'rep': 1" | cut -c1-300
done
