#!/bin/bash
# round 4, session f: the 10 M-dof octree parity test; k_vec preloading form at 1 M / 1.27 M dof (A/B); windowed overflow with more
# workgroups per CU at 10 M dof; fused-vector parity subset.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04f"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest: fused vector phase + 10 M octree"
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -x -s -k "fused_vector or graded_octree_10m or time_out or lock_step_harness or test_solve_matches_reference_fixture" > "$OUT/pytest.log" 2>&1; grep -E "graded octree 10|passed|failed|error" "$OUT/pytest.log" | tail -6
echo "== k_vec: preloading form (PCG_VEC_NT=5) vs general form (13), 1 M octree ebe, brick N=75 ebe/dict, brick N=70 sell"
timeout 600 python tools/iter_ab.py oct1m ebe 300 "PCG_VEC_NT=5|13" > "$OUT/ab_vec_oct1m.json" 2> "$OUT/ab_vec_oct1m.log"; grep "^{" "$OUT/ab_vec_oct1m.log" | cut -c1-230
timeout 600 python tools/iter_ab.py 75 ebe,dict 300 "PCG_VEC_NT=5|13" > "$OUT/ab_vec_n75.json" 2> "$OUT/ab_vec_n75.log"; grep "^{" "$OUT/ab_vec_n75.log" | grep us_per | cut -c1-230
echo "== windowed overflow at 10 M with more workgroups per CU"
timeout 900 python tools/iter_ab.py oct10m sell 60 "PCG_SPMV_OVF+PCG_SPMV_OVF_WINDOW+PCG_SPMV_BLOCKS_PER_CU=split+0+4|win+12+8|win+6+8|win+12+6" > "$OUT/ab_win8_oct10m.json" 2> "$OUT/ab_win8_oct10m.log"; grep "^{" "$OUT/ab_win8_oct10m.log" | grep us_per | cut -c1-300
