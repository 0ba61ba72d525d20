#!/bin/bash
# round 4, session b: (1) parity of the new pieces (k_spmv_win, ticket-ordered k_ebe_mixed, scalar copy); (2) mixed chunks: tickets
# vs barriers, ablations, tile cap; (3) windowed overflow of the split SELL format vs the two-launch form, window sizes; (4) the
# driver's bench command on the new bench.py (CPU baselines, PMC child passes, device-side scalar-CSR point); (5) kernel trace.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r04b"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest subset"
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x -k "mixed_type or split or scalar_copy or time_out" > "$OUT/pytest_subset.log" 2>&1; tail -6 "$OUT/pytest_subset.log"
echo "== mixed chunks: tickets / barriers / ablations (16 no tile adds, 32 no tile MFMA, 64 no tile phase), octree 10 M"
timeout 900 python tools/iter_ab.py oct10m ebe 100 "PCG_EBE_MIX_FLAGS=0|1|16|32|64" "PCG_EBE_TILE_CAP=12|48" > "$OUT/ab_mix_oct10m.json" 2> "$OUT/ab_mix_oct10m.log"; grep "^{" "$OUT/ab_mix_oct10m.log" | cut -c1-240
echo "== mixed chunks at 1 M"
timeout 600 python tools/iter_ab.py oct1m ebe 200 "PCG_EBE_MIX_FLAGS=0|1" "PCG_EBE_TILE_CAP=12|48" > "$OUT/ab_mix_oct1m.json" 2> "$OUT/ab_mix_oct1m.log"; grep "^{" "$OUT/ab_mix_oct1m.log" | cut -c1-240
echo "== windowed overflow, octree 10 M"
timeout 900 python tools/iter_ab.py oct10m sell 60 "PCG_SPMV_OVF+PCG_SPMV_OVF_WINDOW=split+0|win+12|win+24|win+6" > "$OUT/ab_win_oct10m.json" 2> "$OUT/ab_win_oct10m.log"; grep "^{" "$OUT/ab_win_oct10m.log" | cut -c1-300
echo "== windowed overflow, octree 1 M"
timeout 600 python tools/iter_ab.py oct1m sell 200 "PCG_SPMV_OVF+PCG_SPMV_OVF_WINDOW=split+0|win+12|win+4|win+2" > "$OUT/ab_win_oct1m.json" 2> "$OUT/ab_win_oct1m.log"; grep "^{" "$OUT/ab_win_oct1m.log" | cut -c1-300
echo "== the driver's bench command"
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.log" ) 2>&1 | grep real; grep -v "^/opt" "$OUT/bench.log" | cut -c1-250 | tail -25
python - "$OUT/bench.json" <<'P'
import json,sys
try:
    b=json.load(open(sys.argv[1]))
except Exception as ex:
    print("bench line unreadable:", ex); sys.exit(0)
r=b.get('roofline',{})
print('value', b['value'], 'frac', r.get('frac'), 'traffic', r.get('traffic'), 'scalar', {k: r.get('scalar_csr_same_run',{}).get(k) for k in ('n','median_launch_ms','frac_of_peak')})
print('cpu', {k: b.get('cpu_baseline',{}).get(k) for k in ('value','cores','kind')}, 'c_port', b.get('cpu_baseline',{}).get('c_port'))
print('mf', b['matrix_free']['value'], b['matrix_free']['roofline'].get('traffic'))
o=b.get('octree',{})
for k in ('assembled','matrix_free'):
    print('octree', k, o.get(k,{}).get('value'), o.get(k,{}).get('roofline'))
print('octree cpu', {k: o.get('cpu_baseline',{}).get(k) for k in ('value','cores')})
P
cd /tmp
echo "== kernel trace, octree 10 M, assembled (windowed) + matrix-free"
PROF_OCTREE=10m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_oct10m" -o k -- python "$R/tools/prof_op.py" sell,ebe 0 20 > "$OUT/prof_oct10m.log" 2>&1
f=$(find "$OUT/prof_oct10m" -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-200
