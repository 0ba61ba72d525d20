#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r02r"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== ebe tests"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_native_comm.py -m gpu -q -k "ebe or multi_part or mixed or octree or irregular or smallest or threads or processes" 2>&1 | tail -3
for st in 0 1; do
echo "== bench octree PCG_EBE_STREAMS=$st"; PCG_EBE_STREAMS=$st timeout 900 python bench.py --workload octree --steps 300 --operator ebe --no-cpu-baseline > "$OUT/bench_octree_s$st.json" 2> "$OUT/bench_octree.log"; python - "$OUT/bench_octree_s$st.json" <<'P'
import json,sys
b=json.load(open(sys.argv[1]))
print('ebe', b['value'], b['ms_per_step'], b['solve'])
P
done
PROF_OCTREE=1 PCG_EBE_STREAMS=0 python tools/prof_op.py ebe 96 200 2>&1 | tail -1
PROF_OCTREE=1 PCG_EBE_STREAMS=1 python tools/prof_op.py ebe 96 200 2>&1 | tail -1
