#!/bin/bash
# round 4, session aa: depth of the fragment ring of the tile contraction (k-steps requested ahead: 3 / 4 = product / 5) - three builds in
# alternating processes, octree meshes (symmetry classes) at 10 M and 1 M dof.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04aa"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do
  for V in d3 cur d5; do
    L=""; [ "$V" != cur ] && L="$R/pcg-mpi-solver_amd/lib/ab/libpcg_$V.so"
    for M in oct10ms oct1ms; do
      PCG_LIB="$L" timeout 600 python tools/iter_ab.py $M ebe 200 "PCG_EBE_XCD=64" > "$OUT/ab_${M}_${V}_$rep.json" 2> "$OUT/ab_${M}_${V}_$rep.log"
      echo "$V $M: $(grep us_per "$OUT/ab_${M}_${V}_$rep.log" | grep "'rep': 1" | cut -c60-260)"
    done
  done
done
