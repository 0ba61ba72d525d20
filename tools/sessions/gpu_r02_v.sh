#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r02v; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "16_bit or spmv or look_ahead" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
# same-box A/B of the two column widths at 10 M dof
for c in 0 1; do
  PCG_SPMV_COL16=$c timeout 900 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-finish --operator sell > $O/bench_col16_$c.json 2> $O/bench_col16_$c.log || tail -5 $O/bench_col16_$c.log
  python - <<P
import json
for l in open("$O/bench_col16_$c.json"):
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print("COL16=$c", "it/s", round(d["value"],1), "spmv ms", round(r["avg_launch_ms"],4), "GB", round(r["bytes_per_launch"]/1e9,3), "frac", round(r["frac"],3), d["config"]["format"])
P
done
