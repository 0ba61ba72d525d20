#!/bin/bash
# round 4, session z: k_ebe_shared with four (node, direction) items per thread (all slot ranges, then all addends requested before the
# sums) against one item per thread - two builds in alternating processes (tools/build_variant.sh), and a parity subset.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04z"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "ebe and not 10m" > "$OUT/pytest.log" 2>&1 ) 2>&1 | grep real; tail -2 "$OUT/pytest.log" | cut -c1-200
for rep in 1 2; do
  for V in head cur; do
    L=""; [ "$V" != cur ] && L="$R/pcg-mpi-solver_amd/lib/ab/libpcg_$V.so"
    for M in 150 oct10ms oct1ms 75; do
      PCG_LIB="$L" timeout 600 python tools/iter_ab.py $M ebe 200 "PCG_EBE_XCD=64" > "$OUT/ab_${M}_${V}_$rep.json" 2> "$OUT/ab_${M}_${V}_$rep.log"
      echo "$V $M: $(grep us_per "$OUT/ab_${M}_${V}_$rep.log" | grep "'rep': 1" | cut -c60-260)"
    done
  done
done
