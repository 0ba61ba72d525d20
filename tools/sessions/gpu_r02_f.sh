#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r02g"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python tools/ebe_lab.py 150 "chunk_ept2:PCG_EBE_HEX=0,PCG_EBE_EPT=2" "hex_ept1_lb5_atomic:PCG_EBE_HEX=2,PCG_EBE_EPT=1,PCG_EBE_ACC=1" \
  "hex_ept1_lb4_atomic:PCG_EBE_HEX=1,PCG_EBE_EPT=1,PCG_EBE_ACC=1" "hex_ept2_lb3_atomic:PCG_EBE_HEX=1,PCG_EBE_EPT=2,PCG_EBE_ACC=1" \
  > "$OUT/ebe_lab4.json" 2> "$OUT/ebe_lab4.log"; grep -v "^/opt" "$OUT/ebe_lab4.log" | tail -12 | cut -c1-230
echo "== ebe tests HEX=2 EPT=1 ACC=1"; PCG_EBE_HEX=2 PCG_EBE_EPT=1 PCG_EBE_ACC=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "ebe or multi_part or mixed or octree or irregular or smallest" 2>&1 | tail -2
cd /tmp
for cfg in "2 1 1" "0 2 0"; do set -- $cfg
PCG_EBE_HEX=$1 PCG_EBE_EPT=$2 PCG_EBE_ACC=$3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$1$2$3" -o k -- python "$GRAFT_REPO_ROOT/tools/prof_op.py" ebe 150 20 > "$OUT/prof.log" 2>&1
f=$(find "$OUT/prof_$1$2$3" -name "*kernel_stats.csv" | head -1); python - "$f" <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_ebe' in r['Name']: print(r['Name'][:50], r['Calls'], r['AverageNs'])
P
done
