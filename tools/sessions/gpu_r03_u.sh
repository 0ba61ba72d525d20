#!/bin/bash
# round 3, session u: the final code - the driver's bench command first, then smoke, then the full parity suite
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r03u"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rocm-smi --showclocks --showmaxpower --showpower --showmemorypartition --showcomputepartition --showperflevel > "$OUT/rocm_smi.txt" 2>&1
echo "== bench (driver command)"; timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"; echo "rc=$?"; grep -E "octree|rror" "$OUT/bench.log" | tail -6 | cut -c1-200
python - "$OUT/bench.json" <<'P'
import json,sys
b=json.load(open(sys.argv[1])); r=b['roofline']; v=b['roofline_vector_phase']
print('sell', b['value'], b['ms_per_step'], 'frac', r['frac'], 'stream', r['frac_of_stream_read'], 'traffic/bytes', r.get('traffic_over_bytes'))
print('vec', v['avg_launch_ms'], v['frac'], v.get('traffic'), v.get('traffic_over_bytes'))
d=b['assembled_dictionary']; print('dict', d['value'], d['ms_per_step'], d['operator_avg_ms'], d['standalone_spmv'])
m=b['matrix_free']; print('ebe', m['value'], m['ms_per_step'], m['operator_avg_ms'])
o=b.get('octree',{}); print('octree', {k:(round(x.get('value')), x.get('operator_avg_ms'), x.get('sell_padding')) for k,x in o.items() if isinstance(x,dict) and 'value' in x}, o.get('error'))
c=b.get('cpu_baseline',{}); print('cpu', c.get('value'), c.get('cores'), (c.get('numpy_reference_path') or {}).get('value'))
P
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke | tee "$OUT/smoke.log"
echo "== pytest -m gpu"; timeout 1200 python -X faulthandler -m pytest tests -m gpu -q -rA -s > "$OUT/pytest_gpu.log" 2>&1; grep -E "lock-step|graded octree|passed|failed" "$OUT/pytest_gpu.log" | cut -c1-200 | tail -8
