#!/bin/bash
# round 4, session a: first GPU run of the mixed-type chunks (k_ebe_mixed): parity subset, same-process A/B against the per-type
# chunks of round 3 on the graded octree mesh (1 M / 10 M dof) and on the brick (hex section only), kernel trace at 10 M dof;
# the fused-launch time-out recovery on the device.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04a"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest subset"
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x -k "mixed_type or time_out or hanging_node or graded_octree_1m_dof or fused_vector or native_library" > "$OUT/pytest_subset.log" 2>&1; tail -15 "$OUT/pytest_subset.log"
echo "== A/B octree 1 M"
timeout 600 python tools/iter_ab.py oct1m ebe 200 "PCG_EBE_MIXED=1|0" > "$OUT/ab_oct1m.json" 2> "$OUT/ab_oct1m.log"; grep "^{" "$OUT/ab_oct1m.log" | cut -c1-260
echo "== A/B octree 10 M"
timeout 900 python tools/iter_ab.py oct10m ebe 100 "PCG_EBE_MIXED=1|0" > "$OUT/ab_oct10m.json" 2> "$OUT/ab_oct10m.log"; grep "^{" "$OUT/ab_oct10m.log" | cut -c1-260
echo "== A/B brick 10 M (hex section only vs k_ebe_hexs)"
timeout 600 python tools/iter_ab.py 150 ebe 100 "PCG_EBE_MIXED=0|1" > "$OUT/ab_brick.json" 2> "$OUT/ab_brick.log"; grep "^{" "$OUT/ab_brick.log" | cut -c1-260
cd /tmp
echo "== kernel trace, octree 10 M, mixed chunks"
PROF_OCTREE=10m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_mixed" -o k -- python "$R/tools/prof_op.py" ebe 0 20 > "$OUT/prof_mixed.log" 2>&1
f=$(find "$OUT/prof_mixed" -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-200
tail -3 "$OUT/prof_mixed.log"
