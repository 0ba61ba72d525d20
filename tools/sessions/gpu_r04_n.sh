#!/bin/bash
# round 4, session n: same-box A/B of two BUILDS of the engine (tools/build_variant.sh): "k" = fragments requested ahead (committed
# 0f..), "cur" = + a wave's next tile requested one tile ahead.  Alternating processes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04n"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do
  for V in k cur; do
    L=""; [ "$V" != cur ] && L="$R/pcg-mpi-solver_amd/lib/ab/libpcg_$V.so"
    for M in oct10ms oct1ms; do
      PCG_LIB="$L" timeout 600 python tools/iter_ab.py $M ebe 100 "PCG_EBE_MIX_FLAGS=0" > "$OUT/ab_${M}_${V}_$rep.json" 2> "$OUT/ab_${M}_${V}_$rep.log"
      echo "$V $M: $(grep us_per "$OUT/ab_${M}_${V}_$rep.log" | grep "'rep': 1" | cut -c60-260)"
    done
  done
done
