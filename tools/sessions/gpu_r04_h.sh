#!/bin/bash
# round 4, session h: pattern types by symmetry class (GradedOctreeMesh(symmetry=True): 8 types for 95 orientations, per-element dof
# order in the tiles of k_ebe_mixed) - GPU parity tests, same-box A/B of the matrix-free operator at 1 M / 10 M dof against the
# 95-type mesh, the shells / node-cap planner knobs, kernel trace of the 10 M-dof run.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04h"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest subset"
( time timeout 1200 python -X faulthandler -m pytest tests -m gpu -q -x -s -k "mixed_type_chunks_on_gpu or oriented_patterns or goct_sym or large_octree_matrix_is_split" > "$OUT/pytest.log" 2>&1 ) 2>&1 | grep real; grep -E "oriented|passed|failed|error" "$OUT/pytest.log" | cut -c1-400 | tail -8
echo "== matrix-free operator, 95 orientation types vs 8 symmetry classes; shells knob"
for M in oct10m oct10ms; do
  timeout 600 python tools/iter_ab.py $M ebe 100 "PCG_EBE_MIX_SHELLS=0|1" > "$OUT/ab_$M.json" 2> "$OUT/ab_$M.log"; grep "^{" "$OUT/ab_$M.log" | grep us_per | cut -c1-260
done
timeout 600 python tools/iter_ab.py oct10ms ebe 100 "PCG_EBE_NODE_CAP=640|512" > "$OUT/ab_nodecap.json" 2> "$OUT/ab_nodecap.log"; grep "^{" "$OUT/ab_nodecap.log" | grep us_per | cut -c1-260
for M in oct1m oct1ms; do
  timeout 300 python tools/iter_ab.py $M ebe 300 "PCG_EBE_MIX_SHELLS=0|1" > "$OUT/ab_$M.json" 2> "$OUT/ab_$M.log"; grep "^{" "$OUT/ab_$M.log" | grep us_per | cut -c1-260
done
cd /tmp
echo "== rocprofv3 kernel stats, 10 M dof, symmetry classes"
PCG_EBE_STATS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o k -- python "$R/tools/iter_ab.py" oct10ms ebe 100 > "$OUT/prof_ab.json" 2> "$OUT/prof_ab.log"
grep "ebe plan" "$OUT/prof_ab.log"
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); head -6 "$f" | cut -c1-60,140-330
