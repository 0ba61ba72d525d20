#!/bin/bash
# round 4, session q: the 8-node type as matrix-core tiles too (PCG_EBE_HEX_TILES=2: no hex section, no scalar loads of Ke - a
# dependent s_load_dwordx16 costs 157-214 cycles, tools/micro/smem_latency) against the hex section on the vector FMAs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04q"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== parity under the knob"
( time PCG_EBE_HEX_TILES=2 PCG_EBE_TILE_CAP=64 timeout 900 python -m pytest tests -m gpu -q -x -k "mixed_type_chunks or oriented_patterns or graded_octree_1m" > "$OUT/pytest.log" 2>&1 ) 2>&1 | grep real; tail -2 "$OUT/pytest.log" | cut -c1-200
for M in oct10ms oct1ms; do
  PCG_EBE_STAMPS=1 timeout 600 python tools/iter_ab.py $M ebe 100 "PCG_EBE_HEX_TILES+PCG_EBE_TILE_CAP=0+24|2+64" > "$OUT/ab_$M.json" 2> "$OUT/ab_$M.log"; grep -E "^\[pcg\]|us_per" "$OUT/ab_$M.log" | cut -c1-260
done
PCG_EBE_MIXED=1 PCG_EBE_MIX_MTM=4 PCG_EBE_STAMPS=1 timeout 600 python tools/iter_ab.py 150 ebe 100 "PCG_EBE_HEX_TILES+PCG_EBE_TILE_CAP=0+24|2+64" > "$OUT/ab_brick.json" 2> "$OUT/ab_brick.log"; grep -E "^\[pcg\]|us_per" "$OUT/ab_brick.log" | cut -c1-260
