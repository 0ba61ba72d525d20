#!/bin/bash
# round-2 session A: native communicator tests, full GPU suite, bench N=1, comm overhead at world size 1
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r02a"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
nproc > "$OUT/host.txt"; grep -m1 "model name" /proc/cpuinfo >> "$OUT/host.txt"
rocm-smi --showclocks --showmaxpower --showpower --showmemorypartition --showcomputepartition --showperflevel > "$OUT/rocm_smi.txt" 2>&1
rocm-smi --showclocks --showmaxpower --showpower --showmemorypartition --showcomputepartition --showperflevel --json > "$OUT/rocm_smi.json" 2>&1
echo "== native comm tests"; timeout 1500 python -X faulthandler -m pytest tests/test_native_comm.py -m gpu -q -rA -x > "$OUT/pytest_native.log" 2>&1; tail -15 "$OUT/pytest_native.log"
echo "== pytest -m gpu (rest)"; timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -rA --deselect tests/test_native_comm.py > "$OUT/pytest_gpu.log" 2>&1; tail -5 "$OUT/pytest_gpu.log"
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$OUT/smoke.log"
echo "== bench"; timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"; tail -4 "$OUT/bench.log"; cut -c1-400 "$OUT/bench.json"; echo
echo "== comm overhead"; timeout 600 python tools/hook_overhead.py > "$OUT/hook_overhead.json" 2> "$OUT/hook_overhead.log"; cat "$OUT/hook_overhead.json"
