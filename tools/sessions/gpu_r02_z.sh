#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r02z; mkdir -p $O
for i in 1 2; do
echo "== prof_op without torch" | tee -a $O/log.txt
timeout 600 python tools/prof_op.py sell 150 40 2>&1 | tail -1 | tee -a $O/log.txt
echo "== prof_op with torch context" | tee -a $O/log.txt
PROF_IMPORT_TORCH=1 timeout 600 python tools/prof_op.py sell 150 40 2>&1 | tail -2 | tee -a $O/log.txt
done
echo "== bench.py" | tee -a $O/log.txt
timeout 900 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-finish --operator sell > $O/bench.json 2> $O/bench.log
python - <<P | tee -a $O/log.txt
import json
for l in open("$O/bench.json"):
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print("bench it/s", round(d["value"],1), "spmv in-loop ms", round(r["avg_launch_ms"],4), "standalone", r["standalone_spmv"], "stream", {k:round(v) for k,v in r["hbm_stream_this_box"].items() if k!='note'})
P
