#!/bin/bash
# round 5, session i: bench.py --gpus 2 with both ranks on the one GPU, current code: the N > 1 line with the engine-side A/B objects
# (comm.mailbox, comm.mailbox.direct_exchange) at the metric's own size.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r05i"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import sys; sys.path.insert(0,'tests'); import conftest; print(conftest.build_fakenccl())" > "$OUT/fakenccl.txt" 2>&1
FAKE=$(tail -1 "$OUT/fakenccl.txt")
( time PCG_BENCH_SHARE_GPU=1 PCG_RCCL_LIB="$FAKE" PCG_BENCH_RANKS_TIMEOUT_S=230 timeout 240 python bench.py --gpus 2 --steps 40 --warmup 5 --no-cpu-baseline > "$OUT/bench_2ranks_shared_10M.json" 2> "$OUT/bench_2ranks_shared_10M.log" ) 2>&1 | grep real
cut -c1-300 "$OUT/bench_2ranks_shared_10M.json"; echo; grep -iE "fail|error|Traceback" "$OUT/bench_2ranks_shared_10M.log" | head -5 | cut -c1-220
