#!/bin/bash
# round 4, session v: k_ebe_hexs with the scalar loads of Ke scheduled by hand (PCG_EBE_HEX_SCHED=1) - parity of the brick operator under
# the knob, same-process A/B at 10 M / 3 M / 1.27 M dof.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04v"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
( time PCG_EBE_HEX_SCHED=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ebe and not octree and not mixed" > "$OUT/pytest.log" 2>&1 ) 2>&1 | grep real; tail -2 "$OUT/pytest.log" | cut -c1-200
for N in 150 100 75; do
  timeout 600 python tools/iter_ab.py $N ebe 200 "PCG_EBE_HEX_SCHED=0|1" > "$OUT/ab_$N.json" 2> "$OUT/ab_$N.log"; grep -E "us_per" "$OUT/ab_$N.log" | cut -c1-300
done
