#!/bin/bash
# round 3, session h (= session e on the final code): evidence of the round's code - full parity suite incl. the 1 M lock-step walks, smoke, dictionary workgroup
# A/B, the driver's bench command (PMC traffic passes, octree object, CPU baselines), rocprofv3 kernel stats of the bench command
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; TAG="${1:-r03h}"; OUT="$PWD/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{ nproc; cat /sys/fs/cgroup/cpu.max 2>&1; grep -m1 "model name" /proc/cpuinfo; } > "$OUT/host.txt"
rocm-smi --showclocks --showmaxpower --showpower --showmemorypartition --showcomputepartition --showperflevel > "$OUT/rocm_smi.txt" 2>&1
echo "== pytest -m gpu"; timeout 2400 python -X faulthandler -m pytest tests -m gpu -q -rA -s > "$OUT/pytest_gpu.log" 2>&1; grep -E "lock-step|graded octree|passed|failed" "$OUT/pytest_gpu.log" | cut -c1-300 | tail -8
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke | tee "$OUT/smoke.log"
echo "== bench (driver command)"; timeout 1700 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"; echo "rc=$?"; grep -E "octree|rror|PMC|pmc" "$OUT/bench.log" | tail -8
python - "$OUT/bench.json" <<'P'
import json,sys
b=json.load(open(sys.argv[1])); r=b['roofline']
print('sell', b['value'], b['ms_per_step'], 'frac', r['frac'], 'stream', r['frac_of_stream_read'], 'traffic/bytes', r.get('traffic_over_bytes'), 'vec', b['roofline_vector_phase']['avg_launch_ms'], b['roofline_vector_phase']['frac'])
d=b['assembled_dictionary']; print('dict', d['value'], d['ms_per_step'], d['operator_avg_ms'], d['standalone_spmv'])
m=b['matrix_free']; print('ebe', m['value'], m['ms_per_step'], m['operator_avg_ms'])
o=b.get('octree',{}); print('octree', {k:(v.get('value'), v.get('operator_avg_ms')) for k,v in o.items() if isinstance(v,dict) and 'value' in v}, o.get('assembled_dictionary_single_material',{}).get('table'), o.get('error'))
c=b.get('cpu_baseline',{}); print('cpu', c.get('value'), c.get('cores'), (c.get('numpy_reference_path') or {}).get('value'))
P
cd /tmp
echo "== rocprofv3 kernel stats of the bench command"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -o k -- python "$R/bench.py" --no-cpu-baseline --no-octree --no-pmc-traffic > "$OUT/prof_stats_bench.json" 2> "$OUT/prof_stats.log"
f=$(find "$OUT/prof_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/bench_kernel_stats.csv" && head -12 "$f" | cut -c1-160
rm -rf "$OUT/prof_stats"
