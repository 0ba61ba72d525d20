#!/bin/bash
# round 4, session r: k_ebe_mtile - every element on the matrix cores, the 8-node type in colour-pure tiles with its matrix in registers
# (PCG_EBE_HEX_TILES=1) - parity subset under the knob, same-process A/B against the hex section of k_ebe_mixed, clock stamps.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04r"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== parity under the knob"
( time PCG_EBE_HEX_TILES=1 timeout 900 python -m pytest tests -m gpu -q -x -k "mixed_type_chunks or oriented_patterns or graded_octree_1m or goct" > "$OUT/pytest.log" 2>&1 ) 2>&1 | grep real; tail -2 "$OUT/pytest.log" | cut -c1-200
for M in oct10ms oct1ms; do
  PCG_EBE_STAMPS=1 timeout 600 python tools/iter_ab.py $M ebe 100 "PCG_EBE_HEX_TILES=0|1" > "$OUT/ab_$M.json" 2> "$OUT/ab_$M.log"; grep -E "^\[pcg\] k_ebe_mtile|^\[pcg\]   (hex|other)|us_per" "$OUT/ab_$M.log" | cut -c1-330
done
PCG_EBE_MIXED=1 PCG_EBE_MIX_MTM=4 PCG_EBE_STAMPS=1 timeout 600 python tools/iter_ab.py 150 ebe 100 "PCG_EBE_HEX_TILES=0|1" > "$OUT/ab_brick.json" 2> "$OUT/ab_brick.log"; grep -E "^\[pcg\] k_ebe_mtile|^\[pcg\]   (hex|other)|us_per" "$OUT/ab_brick.log" | cut -c1-330
PCG_EBE_MIXED=1 timeout 600 python tools/iter_ab.py 150 ebe 100 "PCG_EBE_HEX_TILES=1" > "$OUT/ab_brick2.json" 2> "$OUT/ab_brick2.log"; grep -E "us_per" "$OUT/ab_brick2.log" | cut -c1-330
timeout 600 python tools/iter_ab.py 150 ebe 100 "PCG_EBE_HEX_TILES=0" > "$OUT/ab_brick_hexs.json" 2> "$OUT/ab_brick_hexs.log"; grep -E "us_per" "$OUT/ab_brick_hexs.log" | cut -c1-330
