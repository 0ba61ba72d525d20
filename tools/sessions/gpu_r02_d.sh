#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r02d"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python tools/ebe_lab.py 150 "chunk_ept2:PCG_EBE_HEX=0,PCG_EBE_EPT=2" \
  "hex_ept2_lb3_rmw:PCG_EBE_HEX=1,PCG_EBE_EPT=2,PCG_EBE_ACC=0" "hex_ept2_lb3_atomic:PCG_EBE_HEX=1,PCG_EBE_EPT=2,PCG_EBE_ACC=1" \
  "hex_ept1_lb4_rmw:PCG_EBE_HEX=1,PCG_EBE_EPT=1,PCG_EBE_ACC=0" "hex_ept1_lb4_atomic:PCG_EBE_HEX=1,PCG_EBE_EPT=1,PCG_EBE_ACC=1" \
  "hex_ept1_lb5_rmw:PCG_EBE_HEX=2,PCG_EBE_EPT=1,PCG_EBE_ACC=0" "hex_ept1_lb5_atomic:PCG_EBE_HEX=2,PCG_EBE_EPT=1,PCG_EBE_ACC=1" \
  > "$OUT/ebe_lab2.json" 2> "$OUT/ebe_lab2.log"; grep -v "^/opt" "$OUT/ebe_lab2.log" | tail -12 | cut -c1-330
