#!/bin/bash
# round 5, session b: (1) the engine-side all-reduce (mailboxes) on the GPU: threads, processes sharing the device through hipIpcMemHandle,
# real RCCL at world size 1; (2) per-rank iteration of the multi-part loop at 1.32 M dof: RCCL all-reduce against mailboxes;
# (3) k_ebe_mtile on the 1 - 3 M-dof octree meshes: chunks cut smaller so that the mesh fills the GPU once (PCG_EBE_TARGET_CHUNKS),
# with clock stamps; (4) bench.py --gpus 2 on the shared GPU: the N > 1 line with every key north_star asks for.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r05b"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import sys; sys.path.insert(0,'tests'); import conftest; print(conftest.build_fakenccl())" > "$OUT/fakenccl.txt" 2>&1
FAKE=$(tail -1 "$OUT/fakenccl.txt")
echo "== mailbox tests"
( time timeout 700 python -X faulthandler -m pytest tests -m gpu -q -rA -k "mailbox" > "$OUT/pytest_mailbox.log" 2>&1 ) 2>&1 | grep real
grep -E "^(FAILED|ERROR|PASSED|SKIPPED)|passed|failed" "$OUT/pytest_mailbox.log" | cut -c1-260 | tail -24
echo "== multi-part iteration: RCCL all-reduce vs mailboxes"
timeout 400 python tools/multi_part_iter.py 150 100 sell,ebe 1,m > "$OUT/multi_part_iter_mail.json" 2> "$OUT/multi_part_iter_mail.log"; echo rc=$?
grep "us_per_iter" "$OUT/multi_part_iter_mail.log" | cut -c1-250
echo "== target chunks, octree 1 / 1.5 / 2.2 M dof"
timeout 500 python tools/iter_ab.py oct1ms ebe 300 "PCG_EBE_TARGET_CHUNKS=0|800|1000|1400" > "$OUT/ab_target_chunks_1m.json" 2> "$OUT/ab_target_chunks_1m.log"; echo rc=$?
grep "us_per_iter" "$OUT/ab_target_chunks_1m.log" | cut -c1-250
timeout 500 python tools/iter_ab.py oct2ms,oct3ms ebe 300 "PCG_EBE_TARGET_CHUNKS=0|1000" > "$OUT/ab_target_chunks_2m3m.json" 2> "$OUT/ab_target_chunks_2m3m.log"; echo rc=$?
grep "us_per_iter" "$OUT/ab_target_chunks_2m3m.log" | cut -c1-250
echo "== stamps at 1 M dof"
for TC in 0 1000; do
PCG_EBE_STAMPS=1 PCG_EBE_TARGET_CHUNKS=$TC timeout 300 python tools/iter_ab.py oct1ms ebe 100 "_=-" > /dev/null 2> "$OUT/stamps_1m_tc$TC.log"; echo rc=$?
grep -iE "stamp|phase|cycles" "$OUT/stamps_1m_tc$TC.log" | head -30 | cut -c1-300
done
echo "== bench --gpus 2, shared GPU"
( time PCG_BENCH_SHARE_GPU=1 PCG_RCCL_LIB="$FAKE" PCG_BENCH_RANKS_TIMEOUT_S=780 timeout 800 python bench.py --gpus 2 --steps 40 --warmup 5 > "$OUT/bench_2ranks_shared_10M.json" 2> "$OUT/bench_2ranks_shared_10M.log" ) 2>&1 | grep real
cut -c1-300 "$OUT/bench_2ranks_shared_10M.json"; echo; grep -iE "fail|error|Traceback" "$OUT/bench_2ranks_shared_10M.log" | head -5 | cut -c1-220
