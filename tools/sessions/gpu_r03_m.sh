#!/bin/bash
# round 3, session m: kernel trace at 1.27 M dof (one GPU's share of 10 M on 8: the strong-scaling floor) and the N > 1 path of
# bench.py at 10 M dof with two ranks sharing the one GPU through the RCCL stand-in (correctness of the path at size; rate meaningless)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r03m"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== 1.27 M dof, un-instrumented vs kernel trace"
timeout 600 python tools/iter_ab.py 75 ebe,dict,sell 300 "PCG_VEC_FUSED=1" 2>&1 | grep us_per_iter | cut -c1-230 | tee "$OUT/iter_n75.log"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/n75_stats" -o k -- python "$R/bench.py" --nodes-per-side 75 --no-cpu-baseline --no-octree --no-pmc-traffic > "$OUT/bench_n75_under_rocprofv3.json" 2> "$OUT/n75.log"
f=$(find "$OUT/n75_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/bench_N75_kernel_stats.csv" && head -9 "$f" | cut -d, -f1-5 | cut -c1-120; rm -rf "$OUT/n75_stats"
cd "$R"
echo "== bench.py --gpus 2 at 10 M dof, ranks share the GPU (RCCL stand-in)"
LIB=$(python -c "import sys; sys.path.insert(0,'tests'); import conftest; print(conftest.build_fakenccl())")
PCG_RCCL_LIB=$LIB PCG_BENCH_SHARE_GPU=1 timeout 1500 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_2ranks_shared_gpu.json" 2> "$OUT/bench_2ranks.log"; echo "rc=$?"; tail -3 "$OUT/bench_2ranks.log" | cut -c1-250
python - "$OUT/bench_2ranks_shared_gpu.json" <<'P'
import json,sys
b=json.load(open(sys.argv[1])); print(b['n_gpus'], b['value'], b['solve'], b['comm'].get('transport','')[:60], b['comm'].get('per_rank_setup_s'), b['comm'].get('native_error'))
print('dict', (b.get('assembled_dictionary') or {}).get('solve'), 'ebe', (b.get('matrix_free') or {}).get('solve'))
P
