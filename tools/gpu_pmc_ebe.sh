#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS -d "$OUT/pmc_ebe_a" -o r1 -- python "$R/tools/prof_op.py" ebe 150 6 > "$OUT/pmc_ebe_a.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU -d "$OUT/pmc_ebe_b" -o r1 -- python "$R/tools/prof_op.py" ebe 150 6 > "$OUT/pmc_ebe_b.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_TA_BUSY -d "$OUT/pmc_ebe_c" -o r1 -- python "$R/tools/prof_op.py" ebe 150 6 > "$OUT/pmc_ebe_c.log" 2>&1
cd "$R"
for d in pmc_ebe_a pmc_ebe_b pmc_ebe_c; do f=$(ls $OUT/$d/*.db 2>/dev/null | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" "$OUT/$d/summary.md" && grep -E "k_ebe_chunk24" "$OUT/$d/summary.md"; tail -2 "$OUT/$d.log"; done
