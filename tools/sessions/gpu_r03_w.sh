#!/bin/bash
# round 3, session w: the octree bench lines on the final code (format string / traffic field of bench.py fixed for this workload)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r03w"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
SIZE="${1:-1m}"
timeout ${BENCH_TIMEOUT:-700} python bench.py --workload octree --octree-size $SIZE --no-cpu-baseline --no-pmc-traffic > "$OUT/bench_octree_$SIZE.json" 2> "$OUT/bench_octree_$SIZE.log"; echo "rc=$?"
python - "$OUT/bench_octree_$SIZE.json" <<'P'
import json,sys
b=json.load(open(sys.argv[1])); r=b.get('roofline',{})
print(b['value'], b['ms_per_step'], 'frac', r.get('frac'), r.get('avg_launch_ms'), r.get('traffic'), b['config'].get('format')[:60], b['config'].get('stored_over_true_blocks'))
for k in ('assembled_dictionary','matrix_free'):
    d=b.get(k) or {}; print(k, d.get('value'), d.get('operator_avg_ms'))
P
