#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r02e"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
CFG1="hex_ept1_lb5_atomic:PCG_EBE_HEX=2,PCG_EBE_EPT=1,PCG_EBE_ACC=1"
CFG2="hex_ept2_lb3_atomic:PCG_EBE_HEX=1,PCG_EBE_EPT=2,PCG_EBE_ACC=1"
echo "== full"; timeout 600 python tools/ebe_lab.py 150 "$CFG1" "$CFG2" 2>&1 >/dev/null | grep -v "^/opt" | cut -c1-200
for m in 1 2 3 4 8 15; do
  echo "== ablation mask $m (1 no contraction, 2 no accumulate, 4 no stores, 8 no x gather)"
  PCG_LAB_LIB=$PWD/tools/_build/libpcg_abl$m.so timeout 600 python tools/ebe_lab.py 150 "$CFG1" "$CFG2" 2>&1 >/dev/null | grep -v "^/opt" | cut -c1-200
done > "$OUT/ablation.log" 2>&1
cat "$OUT/ablation.log"
cd /tmp
echo "== per-kernel split (rocprofv3) of the best configuration"
PCG_EBE_HEX=2 PCG_EBE_EPT=1 PCG_EBE_ACC=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o k -- python "$GRAFT_REPO_ROOT/tools/prof_op.py" ebe 150 20 > "$OUT/prof.log" 2>&1
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); grep -E "k_ebe" "$f" | cut -c1-60,200-330
