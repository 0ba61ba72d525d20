#!/bin/bash
# round 6, session c: k_ebe_mtile with 5 / 6 waves per workgroup (PCG_EBE_MTILE_WAVES) - parity, A/B at 1 / 1.5 / 2.2 M dof, clock stamps.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r06c"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for W in 5 6; do
  echo "== parity, $W waves"
  PCG_EBE_MTILE_WAVES=$W timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "graded_octree_1m or mixed_type_chunks" > "$OUT/pytest_mtile_w$W.log" 2>&1; tail -2 "$OUT/pytest_mtile_w$W.log" | cut -c1-200
done
echo "== A/B"
timeout 600 python tools/iter_ab.py oct1ms,oct2ms,oct3ms ebe 300 "PCG_EBE_MTILE_WAVES=4|5|6" > "$OUT/ab_mtile_waves.json" 2> "$OUT/ab_mtile_waves.log"; echo rc=$?
grep "us_per_iter" "$OUT/ab_mtile_waves.log" | cut -c1-260
echo "== with more chunks"
timeout 600 python tools/iter_ab.py oct1ms ebe 300 "PCG_EBE_MTILE_WAVES+PCG_EBE_TARGET_CHUNKS=5+800|6+800|6+1000" > "$OUT/ab_mtile_waves_chunks.json" 2> "$OUT/ab_mtile_waves_chunks.log"; echo rc=$?
grep "us_per_iter" "$OUT/ab_mtile_waves_chunks.log" | cut -c1-260
echo "== stamps"
for W in 4 5 6; do
PCG_EBE_STAMPS=1 PCG_EBE_MTILE_WAVES=$W timeout 300 python tools/iter_ab.py oct1ms ebe 100 "_=-" > /dev/null 2> "$OUT/stamps_1m_w$W.log"; echo rc=$?
grep -iE "stamp|phase|cycles|per wave" "$OUT/stamps_1m_w$W.log" | head -6 | cut -c1-300
done
