#!/bin/bash
# round 5, session c: (1) the in-band grid barrier of the fused vector launch (k_vec<true>: the published sums are their own arrival
# flags): parity subset, then an A/B against the counter form inside one process (PCG_VEC_INBAND); (2) the mailbox reduction with
# several ranks in ONE process (threads / device group) given enough hardware queues; (3) a kernel TIMELINE of the multi-part
# iteration at 1.32 M dof (rocprofv3 --kernel-trace, csv): where the 58 us between the single-part and the multi-part iteration go.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; OUT="$PWD/gpurun_out/r05c"; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== parity subset (in-band barrier is the default)"
( time timeout 700 python -X faulthandler -m pytest tests -m gpu -q -rA -k "fused_vector_phase or fused_vector_launch or time_out or stagnation_exit_in_lock_step or mailbox_reduction_on_one_gpu" > "$OUT/pytest_vec.log" 2>&1 ) 2>&1 | grep real
grep -E "^(FAILED|ERROR|PASSED|SKIPPED)|passed|failed" "$OUT/pytest_vec.log" | cut -c1-260 | tail -30
echo "== in-band vs counters"
timeout 500 python tools/iter_ab.py oct1ms,75,150 ebe 300 "PCG_VEC_INBAND=1|0" > "$OUT/ab_vec_inband.json" 2> "$OUT/ab_vec_inband.log"; echo rc=$?
grep "us_per_iter" "$OUT/ab_vec_inband.log" | cut -c1-250
echo "== timeline of the multi-part iteration"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_mp" -o mp -- python "$R/tools/multi_part_iter.py" 150 30 ebe 1,m > "$OUT/trace_mp.json" 2> "$OUT/trace_mp.log"; echo rc=$?
find "$OUT/trace_mp" -name "*kernel_trace.csv" | head -2
f=$(find "$OUT/trace_mp" -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && { cp "$f" "$OUT/mp_kernel_trace.csv"; wc -l "$OUT/mp_kernel_trace.csv"; }
rm -rf "$OUT/trace_mp"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_sp" -o sp -- python "$R/tools/iter_ab.py" 75 ebe 60 "_=-" > "$OUT/trace_sp.json" 2> "$OUT/trace_sp.log"; echo rc=$?
f=$(find "$OUT/trace_sp" -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && { cp "$f" "$OUT/sp_kernel_trace.csv"; wc -l "$OUT/sp_kernel_trace.csv"; }
rm -rf "$OUT/trace_sp"
