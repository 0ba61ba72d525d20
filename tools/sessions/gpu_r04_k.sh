#!/bin/bash
# round 4, session k: k_ebe_mixed with the matrix fragments requested four k-steps ahead (explicit register ring; before: one load,
# vmcnt(0), one matrix instruction) and one sign word per lane group - parity subset, A/B on the octree meshes, clock stamps.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04k"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest subset"
( time timeout 1200 python -X faulthandler -m pytest tests -m gpu -q -x -k "mixed_type_chunks or oriented_patterns or goct_sym or graded_octree_1m or ebe" > "$OUT/pytest.log" 2>&1 ) 2>&1 | grep real; tail -3 "$OUT/pytest.log" | cut -c1-300
for M in oct10ms oct10m oct1ms; do
  PCG_EBE_STAMPS=1 timeout 600 python tools/iter_ab.py $M ebe 100 > "$OUT/ab_$M.json" 2> "$OUT/ab_$M.log"; grep -E "^\[pcg\]|us_per" "$OUT/ab_$M.log" | cut -c1-260
done
