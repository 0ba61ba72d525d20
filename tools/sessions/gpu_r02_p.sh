#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r02p"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== bench octree"; timeout 900 python bench.py --workload octree --steps 200 --no-cpu-baseline > "$OUT/bench_octree.json" 2> "$OUT/bench_octree.log"; tail -2 "$OUT/bench_octree.log"; python - "$OUT/bench_octree.json" <<'P'
import json,sys
b=json.load(open(sys.argv[1])); m=b['matrix_free']
print('sell', b['value'], b['ms_per_step'], b['roofline']['avg_launch_ms'], b['roofline']['frac'], b['solve'])
print('ebe', m['value'], m['ms_per_step'], m['operator_avg_ms'], m['n_chunks'], m['solve'])
P
cd /tmp
PROF_OCTREE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o k -- python "$R/tools/prof_op.py" ebe 96 20 > "$OUT/prof.log" 2>&1; tail -2 "$OUT/prof.log"
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); python - "$f" <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_ebe' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'])
P
