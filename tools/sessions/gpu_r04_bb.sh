#!/bin/bash
# round 4, session bb: node orders for the ASSEMBLED operator on the octree meshes (development knob PCG_SELL_NODE_ORDER): generator
# order / Morton curve / blocks of 4^3 and 8^3 lattice units - stored blocks, bytes, iteration and SpMV time, same process.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04bb"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for M in oct10ms oct1ms; do
  timeout 900 python tools/iter_ab.py $M sell 100 "PCG_SELL_NODE_ORDER=|morton|block4|block8" > "$OUT/ab_$M.json" 2> "$OUT/ab_$M.log"; grep -E "matrix|us_per" "$OUT/ab_$M.log" | grep -E "matrix|'rep': 1" | cut -c1-330
done
