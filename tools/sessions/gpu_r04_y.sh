#!/bin/bash
# round 4, session y: PMC passes AFTER the fragment ring - k_ebe_mixed on the 10 M-dof octree mesh, k_ebe_mtile on the 1 M-dof one
# (symmetry classes) - to stand beside the passes of session i (before).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04y"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp
for SZ in 10ms 1ms; do
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  PROF_OCTREE=$SZ timeout 400 rocprofv3 --kernel-trace --pmc $set -d "$OUT/pmc_${SZ}_$i" -o k -- python "$R/tools/prof_op.py" ebe 0 8 > "$OUT/pmc_${SZ}_$i.log" 2>&1
  echo "$SZ pass $i rc=$?"; grep -E "median|rror" "$OUT/pmc_${SZ}_$i.log" | head -2 | cut -c1-200
  f=$(find "$OUT/pmc_${SZ}_$i" -name "*.db" | head -1); [ -n "$f" ] && python "$R/tools/rocpd_summary.py" "$f" "$OUT/pmc_${SZ}_$i.md" && grep -E "k_ebe_m" "$OUT/pmc_${SZ}_$i.md" | cut -c1-200
  rm -rf "$OUT/pmc_${SZ}_$i"
done
done
