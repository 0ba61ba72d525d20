#!/bin/bash
# round 2, session AN: does the contiguous-VRAM request (PCG_ALLOC_CONTIG, 50c01a1) help or hurt each operator?  same box, alternating processes
set -x
OUT=gpurun_out/r02an; mkdir -p $OUT
for c in 1 0 1 0; do
  PCG_ALLOC_CONTIG=$c timeout 200 python bench.py --operator ebe --no-cpu-baseline --no-finish --steps 200 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('contig=$c ebe  it/s %.0f  operator %.4f ms' % (d['value'], d['roofline']['avg_launch_ms'] if 'roofline' in d else float('nan')) if False else 'contig=$c ebe it/s %.0f ms/it %.4f' % (d['value'], d['ms_per_step']))" | tee -a $OUT/ab.txt
done
for c in 1 0; do
  PCG_ALLOC_CONTIG=$c timeout 200 python bench.py --operator sell --no-cpu-baseline --no-finish --steps 200 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('contig=$c sell it/s %.0f ms/it %.4f spmv %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))" | tee -a $OUT/ab.txt
done
