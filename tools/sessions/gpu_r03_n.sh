#!/bin/bash
# round 3, session n: hanging-node pattern types without a node tile (k_ebe_direct; PCG_EBE_DIRECT=0 = the tile kernel k_ebe_rows):
# agreement with the tile kernel and the assembled operator, parity subset, A/B of the iteration at 1 M / 10 M dof (octree),
# kernel trace at 10 M dof
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r03n"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== one apply, three operators"
timeout 300 python tools/ebe_direct_check.py small 2>&1 | tail -4 | tee "$OUT/check.log"
timeout 600 python tools/ebe_direct_check.py oct1m 2>&1 | tail -4 | tee -a "$OUT/check.log"
echo "== parity subset"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_irregular_meshes.py -m gpu -x -q -k "octree or graded or goct or fixture or irregular or mixed or ebe" 2>&1 | tail -5 | tee "$OUT/pytest_direct.log"
echo "== A/B"
timeout 900 python tools/iter_ab.py oct1m ebe 200 "PCG_EBE_DIRECT=0|1" 2>&1 | grep us_per_iter | cut -c1-260 | tee "$OUT/ab_1m.log"
timeout 1200 python tools/iter_ab.py oct10m ebe 100 "PCG_EBE_DIRECT=0|1" 2>&1 | grep us_per_iter | cut -c1-260 | tee "$OUT/ab_10m.log"
echo "== kernel trace, 10 M dof"
cd /tmp
PROF_OCTREE=10m timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t" -o k -- python "$R/tools/prof_op.py" ebe 0 20 > "$OUT/trace_direct.log" 2>&1
grep median "$OUT/trace_direct.log" | cut -c1-200
f=$(find "$OUT/t" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/oct10m_ebe_direct_kernel_stats.csv" && head -8 "$f" | cut -d, -f1-5 | cut -c1-60,140-220
rm -rf "$OUT/t"
