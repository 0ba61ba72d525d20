#!/bin/bash
# round 5, session f: the engine-side neighbour exchange (pcg_enable_direct_exchange: k_halo_put stores into the neighbours' mapped
# buffers, the fix-up waits for their arrival words): parity between processes sharing the GPU, then the per-rank iteration of the
# 1.32 M-dof part - RCCL exchange (1), direct exchange (d), direct exchange + mailbox all-reduce (dm).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r05f"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PCG_MAIL_SPINS=300000
echo "== parity"
( time timeout 420 python -X faulthandler -m pytest tests -m gpu -q -rA -x -k "direct_exchange" > "$OUT/pytest_direct.log" 2>&1 ) 2>&1 | grep real
grep -E "^(FAILED|ERROR|PASSED|SKIPPED)|passed|failed|refused|Error" "$OUT/pytest_direct.log" | cut -c1-300 | tail -20
echo "== per-rank iteration: 1 / d / dm"
timeout 300 python tools/multi_part_iter.py 150 100 sell,ebe 1,d,dm > "$OUT/mpi_direct.json" 2> "$OUT/mpi_direct.log"; echo rc=$?
grep -E "us_per_iter|Error|error" "$OUT/mpi_direct.log" | cut -c1-220
