#!/bin/bash
# round 4, session c: k_ebe_mixed with wave-granular hex passes + prefetched tile headers; ticket-ordered adds in k_ebe_hexs (brick);
# the split SELL format's two launches at 10 M dof under the PMC passes of bench.py (traffic per kernel); defaults at 1 M dof.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04c"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest subset"
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x -k "mixed_type or split_matrix_is_bit or fused_vector or test_ebe" > "$OUT/pytest_subset.log" 2>&1; tail -4 "$OUT/pytest_subset.log"
echo "== mixed chunks, octree 10 M: default / no wave skip (4) / barriers (1)"
timeout 900 python tools/iter_ab.py oct10m ebe 100 "PCG_EBE_MIX_FLAGS=0|4|1" > "$OUT/ab_mix_oct10m.json" 2> "$OUT/ab_mix_oct10m.log"; grep "^{" "$OUT/ab_mix_oct10m.log" | cut -c1-240
echo "== brick 10 M, k_ebe_hexs: tickets (0) / barriers (1)"
timeout 600 python tools/iter_ab.py 150 ebe 100 "PCG_EBE_HEX_FLAGS=0|1" > "$OUT/ab_hexs_brick.json" 2> "$OUT/ab_hexs_brick.log"; grep "^{" "$OUT/ab_hexs_brick.log" | cut -c1-240
echo "== brick 1.27 M (N=75), k_ebe_hexs tickets / barriers"
timeout 600 python tools/iter_ab.py 75 ebe 300 "PCG_EBE_HEX_FLAGS=0|1" > "$OUT/ab_hexs_n75.json" 2> "$OUT/ab_hexs_n75.log"; grep "^{" "$OUT/ab_hexs_n75.log" | cut -c1-240
echo "== octree 1 M assembled: default (4-slice windows) vs split"
timeout 600 python tools/iter_ab.py oct1m sell 200 "PCG_SPMV_OVF=win|split" > "$OUT/ab_win_oct1m.json" 2> "$OUT/ab_win_oct1m.log"; grep "^{" "$OUT/ab_win_oct1m.log" | grep us_per | cut -c1-240
echo "== bench, octree 10 M, both operators, PMC traffic per kernel"
( time timeout 1200 python bench.py --workload octree --octree-size 10m --no-cpu-baseline --steps 50 --warmup 10 --operator both > "$OUT/bench_oct10m.json" 2> "$OUT/bench_oct10m.log" ) 2>&1 | grep real; grep -v "^/opt" "$OUT/bench_oct10m.log" | cut -c1-250 | tail -8
python - "$OUT/bench_oct10m.json" <<'P'
import json,sys
b=json.load(open(sys.argv[1])); r=b['roofline']
print('sell', b['value'], 'frac', r['frac'], 'ms', r['avg_launch_ms'], 'bytes', r['bytes_per_launch'], 'traffic', r.get('traffic'))
print(r.get('traffic_note'))
m=b['matrix_free']; print('ebe', m['value'], m['operator_avg_ms'], m['roofline'].get('traffic'), m['roofline'].get('traffic_note'))
d=b['assembled_dictionary']; print('dict', d.get('value'), d.get('distinct_blocks'))
P
