#!/bin/bash
# round 4, session w: do the workgroups of a CU march through their phases in lock-step?  First-round workgroups start staggered
# (flags bits 8..: units of 1024 cycles per step, bits 16..17: which workgroups share a CU) - k_ebe_hexs on the brick, k_ebe_mixed on the
# 10 M-dof octree mesh.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04w"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
# 256*k + 65536*pattern: k = 6 / 12 units (6 k / 12 k cycles per step), patterns 0..2
timeout 900 python tools/iter_ab.py 150 ebe 200 "PCG_EBE_HEX_FLAGS=0|1536|3072|67072|68608|132608|134144" > "$OUT/ab_150.json" 2> "$OUT/ab_150.log"; grep -E "us_per" "$OUT/ab_150.log" | grep "'rep': 1" | cut -c30-300
timeout 900 python tools/iter_ab.py oct10ms ebe 100 "PCG_EBE_MIX_FLAGS=0|3072|6144|68608|71680|134144|137216" > "$OUT/ab_oct10ms.json" 2> "$OUT/ab_oct10ms.log"; grep -E "us_per" "$OUT/ab_oct10ms.log" | grep "'rep': 1" | cut -c30-300
