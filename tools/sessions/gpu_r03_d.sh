#!/bin/bash
# round 3, session d: dictionary SpMV with four packed words per 16-byte load + 16/8-byte x gathers (node-major vectors again)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r03d"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== tests of the assembled formats"
timeout 1200 python -m pytest tests/test_dictionary_format.py tests/test_gpu_parity.py tests/test_irregular_meshes.py -m gpu -x -q -k "not graded and not full_size_10m" 2>&1 | tail -5 | tee "$OUT/pytest_formats.log"
echo "== per iteration, dict / sell"
timeout 900 python tools/iter_ab.py 150,75 dict,sell 200 "PCG_VEC_FUSED=1" > "$OUT/iter.json" 2> "$OUT/iter.log"; grep us_per_iter "$OUT/iter.log" | cut -c1-260
echo "== stand-alone"
timeout 300 python tools/prof_op.py dict,sell 150 30 2>&1 | grep median
cd /tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set -d "$OUT/pmc$i" -o k -- python "$R/tools/prof_op.py" dict,sell 150 8 > "$OUT/pmc$i.log" 2>&1
  f=$(find "$OUT/pmc$i" -name "*.db" | head -1); [ -n "$f" ] && python "$R/tools/rocpd_summary.py" "$f" "$OUT/pmc$i.md" && grep -E "k_spmv" "$OUT/pmc$i.md" | cut -c1-140
  rm -rf "$OUT/pmc$i"
done
