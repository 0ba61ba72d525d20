#!/bin/bash
# round 5, session a: (1) the new parity tests - 10 M-dof brick in 8 parts on one GPU through the native communicator, the octree lock-step walks,
# the N > 1 bench line's new objects, the scalar copy; (2) the dress rehearsal of BASELINE configs[3] / [4] on the native multi-part path:
# bench.py --gpus 8 with all ranks on this one GPU (RCCL stand-in) at 10 M and 100 M dof; (3) tools/check_100m.py on the current code.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r05a"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
{ nproc; cat /sys/fs/cgroup/cpu.max 2>&1; grep -m1 "model name" /proc/cpuinfo; free -g | head -2; } > "$OUT/host.txt"
python -c "import sys; sys.path.insert(0,'tests'); import conftest; print(conftest.build_fakenccl())" > "$OUT/fakenccl.txt" 2>&1
FAKE=$(tail -1 "$OUT/fakenccl.txt")
echo "== new tests"
( time timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -rA -s -k "eight_parts_of_the_10m or octree_solve_in_lock_step or bench_launches_its_own_ranks or scalar_copy or fused_multi_part" > "$OUT/pytest_new.log" 2>&1 ) 2>&1 | grep real
grep -E "^(FAILED|ERROR|PASSED)|passed|failed|bigbrick|lock-step" "$OUT/pytest_new.log" | cut -c1-260 | tail -24
echo "== bench --gpus 8, shared GPU, 10 M dof"
( time PCG_BENCH_SHARE_GPU=1 PCG_RCCL_LIB="$FAKE" PCG_BENCH_RANKS_TIMEOUT_S=880 timeout 900 python bench.py --gpus 8 --steps 40 --warmup 5 > "$OUT/bench_8ranks_shared_10M.json" 2> "$OUT/bench_8ranks_shared_10M.log" ) 2>&1 | grep real
cut -c1-300 "$OUT/bench_8ranks_shared_10M.json"; echo; grep -iE "fail|error|Traceback" "$OUT/bench_8ranks_shared_10M.log" | head -5 | cut -c1-220
echo "== bench --gpus 8, shared GPU, 100 M dof"
( time PCG_BENCH_SHARE_GPU=1 PCG_RCCL_LIB="$FAKE" PCG_BENCH_RANKS_TIMEOUT_S=880 timeout 900 python bench.py --gpus 8 --steps 20 --warmup 3 --nodes-per-side 322 --no-octree --no-cpu-baseline > "$OUT/bench_8ranks_shared_100M.json" 2> "$OUT/bench_8ranks_shared_100M.log" ) 2>&1 | grep real
cut -c1-300 "$OUT/bench_8ranks_shared_100M.json"; echo; grep -iE "fail|error|Traceback" "$OUT/bench_8ranks_shared_100M.log" | head -5 | cut -c1-220
echo "== check_100m"
( time timeout 600 python tools/check_100m.py > "$OUT/check_100m.log" 2>&1 ) 2>&1 | grep real; tail -3 "$OUT/check_100m.log" | cut -c1-300
