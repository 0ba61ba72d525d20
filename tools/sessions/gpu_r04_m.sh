#!/bin/bash
# round 4, session m: k_ebe_mixed with a wave's next tile requested one tile ahead (the first before the ticket of the second hex pass)
# - parity subset, A/B of the early first request (flag 8 = late) and of raised priority while holding a ticket (flag 128), stamps.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04m"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest subset"
( time timeout 1200 python -X faulthandler -m pytest tests -m gpu -q -x -k "mixed_type_chunks or oriented_patterns or goct_sym or graded_octree_1m" > "$OUT/pytest.log" 2>&1 ) 2>&1 | grep real; tail -3 "$OUT/pytest.log" | cut -c1-300
for M in oct10ms oct1ms; do
  PCG_EBE_STAMPS=1 timeout 600 python tools/iter_ab.py $M ebe 100 "PCG_EBE_MIX_FLAGS=0|8|128" > "$OUT/ab_$M.json" 2> "$OUT/ab_$M.log"; grep -E "^\[pcg\]|us_per" "$OUT/ab_$M.log" | cut -c1-260
done
