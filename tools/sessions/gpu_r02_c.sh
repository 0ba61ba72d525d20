#!/bin/bash
# round-2 session C: matrix-free kernel A/B (tools/ebe_lab.py) + its parity tests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r02c"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== ebe lab"; timeout 1200 python tools/ebe_lab.py 150 > "$OUT/ebe_lab.json" 2> "$OUT/ebe_lab.log"; grep -v "^/opt" "$OUT/ebe_lab.log" | tail -12
for m in 1 2; do for e in 1 2; do
echo "== ebe tests HEX=$m EPT=$e"; PCG_EBE_HEX=$m PCG_EBE_EPT=$e timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "ebe or multi_part or mixed or octree or irregular or smallest" 2>&1 | tail -2
done; done
