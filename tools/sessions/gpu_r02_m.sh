#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r02m"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^/opt" | tail -3 | tee "$OUT/smoke.log"
echo "== lock-step 1M (new matrix-free kernel)"; timeout 1200 python -m pytest tests/test_lockstep.py -m gpu -q -s 2>&1 | grep -E "lock-step|passed|failed" | tee "$OUT/lockstep.log"
echo "== bench with live PMC traffic"; timeout 1500 python bench.py --pmc-traffic > "$OUT/bench_pmc.json" 2> "$OUT/bench_pmc.log"; tail -2 "$OUT/bench_pmc.log"; python - "$OUT/bench_pmc.json" <<'P'
import json,sys
b=json.load(open(sys.argv[1])); r=b['roofline']
print('value',b['value'],'frac',r['frac'],'traffic',r['traffic'],'bytes',r['bytes_per_launch'],r.get('traffic_over_bytes'),r['traffic_note'][:120])
print('cpu',b['cpu_baseline']['value'],b['cpu_baseline']['cores'])
P
echo "== bench 100 M dof on one GPU"; timeout 1500 python bench.py --nodes-per-side 322 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_n322.json" 2> "$OUT/bench_n322.log"; tail -3 "$OUT/bench_n322.log"; python - "$OUT/bench_n322.json" <<'P'
import json,sys
b=json.load(open(sys.argv[1])); m=b['matrix_free']
print('sell', b['value'], b['ms_per_step'], b['roofline']['frac'], b['solve'])
print('ebe', m['value'], m['ms_per_step'], m['operator_avg_ms'], m['roofline']['frac_flops'], m['solve'])
P
