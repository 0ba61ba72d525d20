#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r02t"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 21 31 45; do
timeout 600 python bench.py --nodes-per-side $n --steps 150 --warmup 20 --no-cpu-baseline --no-finish > "$OUT/bench_n$n.json" 2>/dev/null; python - "$OUT/bench_n$n.json" $n <<'P'
import json,sys
b=json.load(open(sys.argv[1])); m=b['matrix_free']
print('N',sys.argv[2],'dof',b['config']['dofs'],'sell us/iter',1e3*b['ms_per_step'],'spmv us',1e3*b['roofline']['avg_launch_ms'],'ebe us/iter',1e3*m['ms_per_step'],'op us',1e3*m['operator_avg_ms'])
P
done
