#!/bin/bash
# round 6, session n: the tuner at 100 M dof (BASELINE configs[4], N = 1): parity + bench; the N = 2 bench test after the probe's fix
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r06n"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PCG_VEC_PLACEMENT_LOG=1
( time timeout 600 python -m pytest tests/test_native_comm.py -x -q -m gpu -k "bench_launches" > "$OUT/pytest_bench_n2.log" 2>&1 ) 2>&1 | grep real; tail -2 "$OUT/pytest_bench_n2.log" | cut -c1-200
( time timeout 1200 python tools/check_100m.py > "$OUT/check_100m.log" 2>&1 ) 2>&1 | grep real; grep -v amdgpu "$OUT/check_100m.log" | tail -6 | cut -c1-250
( time timeout 900 python bench.py --nodes-per-side 322 --operator sell --steps 20 --warmup 5 --no-cpu-baseline --no-pmc-traffic --no-octree > "$OUT/bench_N322_sell.json" 2> "$OUT/bench_N322_sell.log" ) 2>&1 | grep real; tail -1 "$OUT/bench_N322_sell.json" | cut -c1-1000; echo; grep "placement\|k_spmv:" "$OUT/bench_N322_sell.log" | cut -c1-200
cp bench_extras.json "$OUT/bench_extras_N322_sell.json" 2>/dev/null
