#!/bin/bash
# round 6, session d: (1) how k_spmv's concurrent streams are laid over the value array (PCG_SPMV_PATTERN: contiguous runs of slices per wave,
# rotated block-column start) - same process A/B at 10 M dof; (2) the N = 2 bench line with the first-contact probe; (3) BASELINE configs[4]
# at N = 1 on the current code: parity at 100 M dof (tools/check_100m.py) and the bench line at --nodes-per-side 322.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r06d"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PCG_TEST_LOG_DIR="$OUT/failed"
echo "== (1) k_spmv stream patterns, 10 M dof"
timeout 600 python tools/iter_ab.py 150 sell 200 "PCG_SPMV_PATTERN=0|8|4|12" > "$OUT/ab_spmv_pattern.json" 2> "$OUT/ab_spmv_pattern.log"; echo rc=$?
grep "us_per_iter" "$OUT/ab_spmv_pattern.log" | cut -c1-260
echo "== (2) N = 2 bench line (first-contact probe) + the multi-part tests"
( time timeout 900 python -m pytest tests/test_native_comm.py -x -q -m gpu -k "bench_launches or parts_as_processes or real_rccl_world_size_1" > "$OUT/pytest_bench_n2.log" 2>&1 ) 2>&1 | grep real; tail -3 "$OUT/pytest_bench_n2.log" | cut -c1-300
echo "== (3) 100 M dof on one GPU"
( time timeout 1200 python tools/check_100m.py > "$OUT/check_100m.log" 2>&1 ) 2>&1 | grep real; tail -12 "$OUT/check_100m.log" | cut -c1-250
for K in sell ebe; do
( time timeout 900 python bench.py --nodes-per-side 322 --operator $K --steps 20 --warmup 5 --no-cpu-baseline --no-pmc-traffic --no-octree > "$OUT/bench_N322_$K.json" 2> "$OUT/bench_N322_$K.log" ) 2>&1 | grep real; tail -1 "$OUT/bench_N322_$K.json" | cut -c1-900; echo
cp bench_extras.json "$OUT/bench_extras_N322_$K.json" 2>/dev/null
done
