#!/bin/bash
# PMC passes for the matrix-free kernels (round-2 default): wave occupancy / wait share / VALU and LDS activity
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r02n"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d "$OUT/pmc$i" -o k -- python "$R/tools/prof_op.py" ebe 150 8 > "$OUT/pmc$i.log" 2>&1
  f=$(find "$OUT/pmc$i" -name "*.db" | head -1); [ -n "$f" ] && python "$R/tools/rocpd_summary.py" "$f" "$OUT/pmc$i.md" && grep -E "k_ebe" "$OUT/pmc$i.md"
done
