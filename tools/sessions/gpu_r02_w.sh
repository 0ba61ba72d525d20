#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r02w; mkdir -p $O
for rpl in 1 2; do
  timeout 900 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-finish --operator sell --rows-per-lane $rpl > $O/bench_rpl$rpl.json 2> $O/bench_rpl$rpl.log || tail -5 $O/bench_rpl$rpl.log
  python - <<P
import json
for l in open("$O/bench_rpl$rpl.json"):
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]
        print("RPL=$rpl", "it/s", round(d["value"],1), "spmv ms", round(r["avg_launch_ms"],4), "frac", round(r["frac"],3), "standalone", r["standalone_spmv"], "stream", r["hbm_stream_this_box"])
P
done
