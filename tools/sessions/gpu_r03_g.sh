#!/bin/bash
# round 3, session g: octree mesh - SELL row sorting and per-class side streams, A/B at 1 M and 10 M dof
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r03g"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== tests touching the changed paths"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_irregular_meshes.py tests/test_dictionary_format.py -m gpu -x -q -k "octree or graded or fixture or irregular or mixed or dict" 2>&1 | tail -4 | tee "$OUT/pytest.log"
echo "== octree 1 M"
timeout 600 python tools/iter_ab.py oct1m ebe 150 "PCG_EBE_STREAMS=1|0" > "$OUT/oct1m_streams.json" 2> "$OUT/oct1m_streams.log"; grep us_per_iter "$OUT/oct1m_streams.log" | cut -c1-250
timeout 600 python tools/iter_ab.py oct1m sell 150 "PCG_SELL_SORT=1|0" > "$OUT/oct1m_sort.json" 2> "$OUT/oct1m_sort.log"; grep us_per_iter "$OUT/oct1m_sort.log" | cut -c1-250
echo "== octree 10 M"
timeout 900 python tools/iter_ab.py oct10m ebe 150 "PCG_EBE_STREAMS=1|0" > "$OUT/oct10m_streams.json" 2> "$OUT/oct10m_streams.log"; grep us_per_iter "$OUT/oct10m_streams.log" | cut -c1-250
timeout 900 python tools/iter_ab.py oct10m sell 150 "PCG_SELL_SORT=1|0" > "$OUT/oct10m_sort.json" 2> "$OUT/oct10m_sort.log"; grep us_per_iter "$OUT/oct10m_sort.log" | cut -c1-250
echo "== brick unaffected"
timeout 600 python tools/iter_ab.py 150 ebe 200 "PCG_VEC_FUSED=1" 2>&1 | grep us_per_iter | cut -c1-250
