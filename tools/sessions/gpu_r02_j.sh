#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r02j"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python tools/ebe_lab.py 150 "chunk_ept2:PCG_EBE_HEX=0,PCG_EBE_EPT=2" "hex_ept1_lb5_atomic:PCG_EBE_HEX=2,PCG_EBE_EPT=1,PCG_EBE_ACC=1" \
  "hexs_2pass_lb4_atomic:PCG_EBE_HEX=4,PCG_EBE_EPT=2,PCG_EBE_ACC=1" "hexs_2pass_lb4_rmw:PCG_EBE_HEX=4,PCG_EBE_EPT=2,PCG_EBE_ACC=0" \
  > "$OUT/ebe_lab5.json" 2> "$OUT/ebe_lab5.log"; grep -v "^/opt" "$OUT/ebe_lab5.log" | tail -12 | cut -c1-230
echo "== ebe tests HEX=4 EPT=2 ACC=1"; PCG_EBE_HEX=4 PCG_EBE_EPT=2 PCG_EBE_ACC=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "ebe or multi_part or mixed or octree or irregular or smallest" 2>&1 | tail -2
cd /tmp
PCG_EBE_HEX=4 PCG_EBE_EPT=2 PCG_EBE_ACC=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o k -- python "$GRAFT_REPO_ROOT/tools/prof_op.py" ebe 150 20 > "$OUT/prof.log" 2>&1
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); python - "$f" <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_ebe' in r['Name']: print(r['Name'][:50], r['Calls'], r['AverageNs'])
P
