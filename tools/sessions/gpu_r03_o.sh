#!/bin/bash
# round 3, session o: HBM traffic of the matrix-free kernels on the graded octree mesh at 10 M dof (FETCH_SIZE, WRITE_SIZE, L2 hits),
# one counter set per pass
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r03o"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  PROF_OCTREE=10m timeout 400 rocprofv3 --kernel-trace --pmc $set -d "$OUT/pmc$i" -o k -- python "$R/tools/prof_op.py" ebe 0 5 > "$OUT/pmc$i.log" 2>&1
  echo "pass $i ($set) rc=$?"; grep -E "median|rror" "$OUT/pmc$i.log" | head -4 | cut -c1-200
  f=$(find "$OUT/pmc$i" -name "*.db" | head -1); [ -n "$f" ] && python "$R/tools/rocpd_summary.py" "$f" "$OUT/oct10m_ebe_pmc$i.md" && grep -E "k_ebe" "$OUT/oct10m_ebe_pmc$i.md" | cut -c1-220
  rm -rf "$OUT/pmc$i"
done
