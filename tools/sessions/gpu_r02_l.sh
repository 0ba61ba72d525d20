#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r02l"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import sys; sys.path.insert(0,'tests'); import conftest; print(conftest.build_fakenccl())"
echo "== bench --gpus 2 dry run at 10 M dof (two ranks share the GPU, RCCL stand-in)"
PCG_BENCH_SHARE_GPU=1 PCG_RCCL_LIB=$PWD/tests/fakenccl/_build/libfakenccl.so timeout 1200 python bench.py --gpus 2 --steps 20 --warmup 5 > "$OUT/bench_2ranks_shared_gpu.json" 2> "$OUT/bench_2ranks.log"; tail -3 "$OUT/bench_2ranks.log"; cut -c1-700 "$OUT/bench_2ranks_shared_gpu.json"; echo
echo "== comm overhead at world size 1 (native vs callbacks)"; timeout 600 python tools/hook_overhead.py > "$OUT/hook_overhead.json" 2> "$OUT/hook_overhead.log"; cat "$OUT/hook_overhead.json"; echo
cd /tmp
echo "== non-kernel time per iteration at 1.27 M dof (N=75)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof75" -o k -- python "$R/bench.py" --nodes-per-side 75 --steps 400 --warmup 20 --no-cpu-baseline --no-finish > "$OUT/bench_n75_prof.json" 2> "$OUT/prof75.log"
python - "$OUT" <<'P'
import csv,json,sys,glob
out=sys.argv[1]
b=json.load(open(out+'/bench_n75_prof.json'))
print('sell ms/step', b['ms_per_step'], 'ebe ms/step', b['matrix_free']['ms_per_step'])
tr=glob.glob(out+'/prof75/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(tr)))
print(len(rows),'kernel records')
P
