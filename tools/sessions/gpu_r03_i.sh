#!/bin/bash
# round 3, session i: hanging-node element kernel with the pattern matrix in LDS (k_ebe_rows KLDS) - parity, A/B on the octree mesh
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r03i"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== parity of the matrix-free operator on meshes with hanging-node / irregular pattern types"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_irregular_meshes.py tests/test_native_comm.py -m gpu -x -q -k "octree or graded or fixture or irregular or mixed or ebe or goct or oct" 2>&1 | tail -4 | tee "$OUT/pytest.log"
echo "== octree 1 M / 10 M: Ke through LDS vs scalar loads"
timeout 600 python tools/iter_ab.py oct1m ebe 150 "PCG_EBE_ROWS_LDS=1|0" > "$OUT/oct1m_klds.json" 2> "$OUT/oct1m_klds.log"; grep us_per_iter "$OUT/oct1m_klds.log" | cut -c1-250
timeout 900 python tools/iter_ab.py oct10m ebe 150 "PCG_EBE_ROWS_LDS=1|0" > "$OUT/oct10m_klds.json" 2> "$OUT/oct10m_klds.log"; grep us_per_iter "$OUT/oct10m_klds.log" | cut -c1-250
cd /tmp
PROF_OCTREE=1m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/oct_stats" -o k -- python "$R/tools/prof_op.py" ebe 0 20 > "$OUT/oct_stats.log" 2>&1
f=$(find "$OUT/oct_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/octree_1m_kernel_stats_klds.csv" && head -7 "$f" | cut -d, -f1-5 | cut -c1-150; rm -rf "$OUT/oct_stats"
