#!/bin/bash
# round 3, session v (two calls): split SELL format - lanes of an overflow slice in row order vs in sorted order, a larger sorting
# window; kernel trace of the two SpMV kernels on the 10 M-dof octree mesh; the octree bench line (1 M dof) with the stand-alone
# operator time covering both kernels
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r03v"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
if [ "${1:-all}" != "second" ]; then
  timeout 600 python tools/split_sweep.py oct10m 60 1.5:512:s 1.5:512 1.5:2048 2>&1 | grep stored_over_true | tee "$OUT/sweep_lane_order.log"
fi
echo "== bench.py --workload octree (1 M dof)"
timeout 600 python bench.py --workload octree --octree-size 1m --no-cpu-baseline --no-pmc-traffic > "$OUT/bench_octree_1m.json" 2> "$OUT/bench_octree_1m.log"; echo "rc=$?"
python - "$OUT/bench_octree_1m.json" <<'P'
import json,sys
b=json.load(open(sys.argv[1])); r=b.get('roofline',{})
print(b['config'].get('workload'), b['value'], b['ms_per_step'], 'frac', r.get('frac'), r.get('avg_launch_ms'), b['config'].get('format'))
for k in ('assembled_dictionary','matrix_free'):
    d=b.get(k) or {}; print(k, d.get('value'), d.get('operator_avg_ms'))
P
cd /tmp
PROF_OCTREE=10m timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t" -o k -- python "$R/tools/prof_op.py" sell 0 20 > "$OUT/trace.log" 2>&1
grep median "$OUT/trace.log" | cut -c1-200
f=$(find "$OUT/t" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/oct10m_sell_split_kernel_stats.csv" && head -5 "$f" | cut -d, -f1-5 | cut -c1-40,100-220
rm -rf "$OUT/t"
