#!/bin/bash
# round 4, session i: k_ebe_mixed after the register diet (component order in ONE register, two registers per tile node kept to the
# write-out: 13-15 spilled VGPRs -> 1-2) - parity subset, same-box A/B 95 types / 8 symmetry classes, the 3-workgroups-per-CU
# instantiation (PCG_EBE_MIX_MTM=5), chunk caps at 1 M dof, and three PMC passes of the 10 M-dof operator (symmetry classes).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04i"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest subset"
( time timeout 1200 python -X faulthandler -m pytest tests -m gpu -q -x -k "mixed_type_chunks or oriented_patterns or goct_sym or graded_octree_1m" > "$OUT/pytest.log" 2>&1 ) 2>&1 | grep real; tail -3 "$OUT/pytest.log" | cut -c1-300
echo "== 10 M dof: 95 types, 8 classes, 8 classes at 3 workgroups per CU"
timeout 600 python tools/iter_ab.py oct10m ebe 100 "PCG_EBE_MIX_MTM=4|5" > "$OUT/ab_oct10m.json" 2> "$OUT/ab_oct10m.log"; grep "^{" "$OUT/ab_oct10m.log" | grep us_per | cut -c1-260
timeout 600 python tools/iter_ab.py oct10ms ebe 100 "PCG_EBE_MIX_MTM=4|5" > "$OUT/ab_oct10ms.json" 2> "$OUT/ab_oct10ms.log"; grep "^{" "$OUT/ab_oct10ms.log" | grep us_per | cut -c1-260
echo "== 1 M dof, symmetry classes: chunk caps"
timeout 600 python tools/iter_ab.py oct1ms ebe 300 "PCG_EBE_NODE_CAP+PCG_EBE_HEX_CAP=768+512|576+384|448+256|320+192" > "$OUT/ab_oct1ms.json" 2> "$OUT/ab_oct1ms.log"; grep "^{" "$OUT/ab_oct1ms.log" | grep us_per | cut -c1-300
cd /tmp
echo "== PMC passes, 10 M dof, symmetry classes"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1))
  PROF_OCTREE=10ms timeout 400 rocprofv3 --kernel-trace --pmc $set -d "$OUT/pmc$i" -o k -- python "$R/tools/prof_op.py" ebe 0 8 > "$OUT/pmc$i.log" 2>&1
  echo "pass $i rc=$?"; grep -E "median|rror" "$OUT/pmc$i.log" | head -3 | cut -c1-200
  f=$(find "$OUT/pmc$i" -name "*.db" | head -1); [ -n "$f" ] && python "$R/tools/rocpd_summary.py" "$f" "$OUT/pmc$i.md" && grep -E "k_ebe" "$OUT/pmc$i.md" | cut -c1-200
  rm -rf "$OUT/pmc$i"
done
