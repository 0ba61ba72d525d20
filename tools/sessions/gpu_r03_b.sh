#!/bin/bash
# round 3, session b: first contact of the fused vector launch (k_vec) with the GPU: parity suite, A/B per iteration, bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r03b"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== new tests first"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "vector or fused or kernels_vs_numpy" 2>&1 | tail -8 | tee "$OUT/pytest_vec.log"
timeout 600 python -m pytest tests/test_lockstep.py -m gpu -x -q -s -k "stagnation" 2>&1 | grep -E "lock-step|passed|failed|Error" | tee "$OUT/pytest_stag.log"
echo "== A/B per iteration"
timeout 900 python tools/iter_ab.py 75,150 ebe,dict,sell 200 "PCG_VEC_FUSED=1|0" > "$OUT/iter_ab.json" 2> "$OUT/iter_ab.log"; grep -c us_per_iter "$OUT/iter_ab.log"; cut -c1-230 "$OUT/iter_ab.log" | tail -30
echo "== full GPU suite (without the 1 M lock-step walks)"
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_lockstep.py::test_every_iteration_of_a_full_solve_in_lock_step 2>&1 | tail -12 | tee "$OUT/pytest_gpu.log"
echo "== bench (driver command)"
timeout 1500 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"; echo "rc=$?"; tail -4 "$OUT/bench.log"
python - "$OUT/bench.json" <<'P'
import json,sys
b=json.load(open(sys.argv[1])); r=b['roofline']
print('sell', b['value'], b['ms_per_step'], 'frac', r['frac'], 'traffic', r.get('traffic'), r.get('traffic_over_bytes'), 'vec', b.get('roofline_vector_phase'))
d=b['assembled_dictionary']; print('dict', d['value'], d['ms_per_step'], d['operator_avg_ms'], d.get('vector_phase'))
m=b['matrix_free']; print('ebe', m['value'], m['ms_per_step'], m['operator_avg_ms'], m.get('vector_phase'))
print('scalar csr', r.get('scalar_csr_same_run'))
c=b.get('cpu_baseline',{}); print('cpu', c.get('value'), c.get('cores'), c.get('numpy_reference_path'))
P
