#!/bin/bash
# round 3, session f: the octree mesh on the operators (chunk packing A/B, kernel stats, 10 M dof), 100 M dof on one GPU with the
# fused vector launch (tail path beyond the register-resident chunks), 10 M-dof lock-step window
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r03f"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== octree 1 M: chunk packing of the hanging-node types"
timeout 600 python tools/iter_ab.py oct1m ebe 150 "PCG_EBE_GREEDY_CHUNKS=1|0" > "$OUT/oct_greedy.json" 2> "$OUT/oct_greedy.log"; grep us_per_iter "$OUT/oct_greedy.log" | cut -c1-260
echo "== octree 1 M: kernel stats of the operators"
cd /tmp
PROF_OCTREE=1m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/oct_stats" -o k -- python "$R/tools/prof_op.py" ebe,sell 0 20 > "$OUT/oct_stats.log" 2>&1
f=$(find "$OUT/oct_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/octree_1m_kernel_stats.csv" && head -9 "$f" | cut -c1-200; rm -rf "$OUT/oct_stats"
cd "$R"
echo "== dictionary format after the revert to 512-thread workgroups + fused"
timeout 600 python tools/iter_ab.py 150 dict,ebe 200 "PCG_VEC_FUSED=1" > "$OUT/iter_final.json" 2> "$OUT/iter_final.log"; grep us_per_iter "$OUT/iter_final.log" | cut -c1-260
echo "== octree 10 M dof"
timeout 1200 python bench.py --workload octree --octree-size 10m --no-cpu-baseline > "$OUT/bench_octree_10m.json" 2> "$OUT/bench_octree_10m.log"; echo "rc=$?"; tail -3 "$OUT/bench_octree_10m.log" | cut -c1-300
python - "$OUT/bench_octree_10m.json" <<'P'
import json,sys
b=json.load(open(sys.argv[1]))
print(b['config']['workload'][:160]); print('sell', b['value'], b['ms_per_step'], b['roofline']['avg_launch_ms'], b['roofline']['frac'], b['solve'])
d=b['assembled_dictionary']; print('dict', d.get('value'), d.get('operator_avg_ms'), d.get('table'))
m=b['matrix_free']; print('ebe', m['value'], m['ms_per_step'], m['operator_avg_ms'], m['n_chunks'], m['solve'])
P
echo "== 10 M-dof lock-step window"
PCG_LOCKSTEP_10M=25 timeout 1500 python -m pytest tests/test_lockstep.py -m gpu -q -s -k "window_at_10m" 2>&1 | grep -E "lock-step|passed|failed|Error" | cut -c1-400 | tee "$OUT/lockstep_10m.log"
echo "== 100 M dof on one GPU"
timeout 1500 python bench.py --nodes-per-side 322 --operator ebe --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_100m_ebe.json" 2> "$OUT/bench_100m_ebe.log"; echo "rc=$?"; tail -2 "$OUT/bench_100m_ebe.log" | cut -c1-300
timeout 1500 python bench.py --nodes-per-side 322 --operator dict --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_100m_dict.json" 2> "$OUT/bench_100m_dict.log"; echo "rc=$?"; tail -2 "$OUT/bench_100m_dict.log" | cut -c1-300
python - "$OUT" <<'P'
import json,sys,os
for k in ("ebe","dict"):
    try:
        b=json.load(open(os.path.join(sys.argv[1], f"bench_100m_{k}.json")))
        print(k, b['value'], b['ms_per_step'], b['solve'], (b.get('assembled_dictionary') or {}).get('operator_avg_ms'), (b.get('assembled_dictionary') or {}).get('vector_phase',{}) )
    except Exception as ex: print(k, 'failed', ex)
P
