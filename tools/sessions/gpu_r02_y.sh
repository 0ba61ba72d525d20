#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r02y; mkdir -p $O
ls tools/_build/ | head
for v in "" tools/_build/libpcg_sabl1.so tools/_build/libpcg_sabl2.so tools/_build/libpcg_sabl5.so; do
  echo "== lib ${v:-product}" | tee -a $O/abl.log
  PCG_LIB=$v timeout 600 python tools/prof_op.py sell 150 40 2>&1 | tail -1 | tee -a $O/abl.log
done
python - <<'P' | tee -a $O/abl.log
import os,sys
sys.path.insert(0,'pcg-mpi-solver_amd')
import numpy as np
from pcg_mi355x.operator import Operator
rp=np.arange(0,4,dtype=np.int64); c=np.arange(3,dtype=np.int32); v=np.ones(3)
op=Operator.from_csr(rp,c,v,block=1)
nb=6<<30
print('read(tuned)',round(op.bench_hbm(nb,'read',10)),'slices 8B',round(op.bench_hbm(nb,2,10)),'step-major',round(op.bench_hbm(nb,4,10)))
P
