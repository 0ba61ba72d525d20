#!/bin/bash
# round 3, session c: direction-major vectors + packed dictionary words + cheaper grid barrier: suite, A/B, counters after, bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r03c"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== GPU suite (without the 1 M lock-step walks)"
timeout 1500 python -m pytest tests -m gpu -x -q -s --deselect tests/test_lockstep.py::test_every_iteration_of_a_full_solve_in_lock_step 2>&1 | grep -E "graded octree|passed|failed|Error|error" | tail -12 | tee "$OUT/pytest_gpu.log"
echo "== A/B layout (created per value) x fused"
timeout 900 python tools/iter_ab.py 150 dict,sell 200 "PCG_LAYOUT_SOA=1|0" > "$OUT/iter_ab_layout.json" 2> "$OUT/iter_ab_layout.log"; grep us_per_iter "$OUT/iter_ab_layout.log" | cut -c1-260
timeout 900 python tools/iter_ab.py 75,150 ebe 200 "PCG_VEC_FUSED=1|0" > "$OUT/iter_ab_fused.json" 2> "$OUT/iter_ab_fused.log"; grep us_per_iter "$OUT/iter_ab_fused.log" | cut -c1-260
timeout 600 python tools/iter_ab.py 75 dict,sell 300 "PCG_LAYOUT_SOA=1|0" > "$OUT/iter_ab_layout_small.json" 2> "$OUT/iter_ab_layout_small.log"; grep us_per_iter "$OUT/iter_ab_layout_small.log" | cut -c1-260
echo "== dictionary SpMV: unroll depth of the block-column loop (stand-alone launches)"
for lib in "" tools/_build/libpcg_du6.so tools/_build/libpcg_du9.so; do
  PCG_LIB=$lib timeout 300 python tools/prof_op.py dict 150 30 2>&1 | grep median | sed "s|^|lib=${lib:-product(3)} |"
done | tee "$OUT/dict_unroll.txt"
echo "== counters after (k_spmv_dict / k_spmv, direction-major)"
cd /tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set -d "$OUT/pmc$i" -o k -- python "$R/tools/prof_op.py" dict,sell 150 8 > "$OUT/pmc$i.log" 2>&1
  f=$(find "$OUT/pmc$i" -name "*.db" | head -1); [ -n "$f" ] && python "$R/tools/rocpd_summary.py" "$f" "$OUT/pmc$i.md" && grep -E "k_spmv" "$OUT/pmc$i.md" | cut -c1-160
  rm -rf "$OUT/pmc$i"
done
cd "$R"
echo "== bench (driver command)"
timeout 1700 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"; echo "rc=$?"; grep -E "octree|error|Error" "$OUT/bench.log" | tail -8
python - "$OUT/bench.json" <<'P'
import json,sys
b=json.load(open(sys.argv[1])); r=b['roofline']
print('sell', b['value'], b['ms_per_step'], 'frac', r['frac'], 'stream', r['frac_of_stream_read'], 'traffic/bytes', r.get('traffic_over_bytes'), 'vec', b['roofline_vector_phase']['avg_launch_ms'], b['roofline_vector_phase']['frac'])
d=b['assembled_dictionary']; print('dict', d['value'], d['ms_per_step'], d['operator_avg_ms'], d['standalone_spmv'])
m=b['matrix_free']; print('ebe', m['value'], m['ms_per_step'], m['operator_avg_ms'])
o=b.get('octree',{}); print('octree', {k:(v.get('value'), v.get('operator_avg_ms')) for k,v in o.items() if isinstance(v,dict) and 'value' in v}, o.get('assembled_dictionary',{}).get('table'), o.get('error'))
P
