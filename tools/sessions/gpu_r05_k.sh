#!/bin/bash
# round 5, session k: BASELINE configs[4] at N = 1 on the current code - bench.py at 100 158 744 dof (brick N = 322), assembled operator,
# then matrix-free; timed windows only (the parity of both operators at this size on this code: tools/check_100m.py, session a).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r05k"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for OP in sell ebe; do
( time timeout 190 python bench.py --nodes-per-side 322 --operator $OP --steps 10 --warmup 3 --no-finish --no-octree --no-cpu-baseline --no-pmc-traffic > "$OUT/bench_N322_$OP.json" 2> "$OUT/bench_N322_$OP.log" ) 2>&1 | grep real
cut -c1-330 "$OUT/bench_N322_$OP.json"; echo; grep -iE "fail|error|Traceback" "$OUT/bench_N322_$OP.log" | head -3 | cut -c1-200
done
