#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for b in 4 8 16 32; do PCG_STREAM_BLOCKS_PER_CU=$b python - <<'P'
import os,sys
sys.path.insert(0,'pcg-mpi-solver_amd')
import numpy as np
from pcg_mi355x.operator import Operator
rp=np.arange(0,4,dtype=np.int64); c=np.arange(3,dtype=np.int32); v=np.ones(3)
op=Operator.from_csr(rp,c,v,block=1)
print('blocks/CU',os.environ['PCG_STREAM_BLOCKS_PER_CU'],'read',round(op.bench_hbm(2<<30,'read')), 'read 8GiB', round(op.bench_hbm(8<<30,'read',10)),'copy',round(op.bench_hbm(1<<30,'copy')))
P
done
