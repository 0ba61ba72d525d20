#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$PWD/gpurun_out/r02h"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest -m gpu (all but lock-step)"; timeout 1800 python -X faulthandler -m pytest tests -m gpu -q -rA --deselect tests/test_lockstep.py > "$OUT/pytest_gpu.log" 2>&1; tail -4 "$OUT/pytest_gpu.log"
echo "== lab"; timeout 600 python tools/ebe_lab.py 150 "default:PCG_EBE_EPT=1" "chunk_ept2:PCG_EBE_HEX=0,PCG_EBE_EPT=2" 2>&1 >/dev/null | grep -v "^/opt" | cut -c1-230
echo "== bench"; timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"; tail -3 "$OUT/bench.log"; python - "$OUT/bench.json" <<'P'
import json,sys
b=json.load(open(sys.argv[1])); m=b['matrix_free']
print('sell', b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline']['hbm_stream_this_box'])
print('ebe', m['value'], m['ms_per_step'], m['operator_avg_ms'], m['roofline']['frac_flops'], m['standalone_operator'])
print('cpu', b['cpu_baseline']['value'], b['cpu_baseline']['cores'])
P
