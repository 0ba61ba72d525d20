#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python - <<'P'
import os,sys
sys.path.insert(0,'pcg-mpi-solver_amd')
import numpy as np
from pcg_mi355x.operator import Operator
rp=np.arange(0,4,dtype=np.int64); c=np.arange(3,dtype=np.int32); v=np.ones(3)
op=Operator.from_csr(rp,c,v,block=1)
nb=6<<30
for rep in range(2):
    print('read(tuned)',round(op.bench_hbm(nb,'read',10)))
    for b in (4,8):
        os.environ['PCG_STREAM_BLOCKS_PER_CU']=str(b)
        print(' blocks/CU',b,'slices 8B',round(op.bench_hbm(nb,2,10)),'slices 16B',round(op.bench_hbm(nb,3,10)),'step-major 8B',round(op.bench_hbm(nb,4,10)))
    del os.environ['PCG_STREAM_BLOCKS_PER_CU']
P
