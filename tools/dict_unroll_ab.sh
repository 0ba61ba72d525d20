#!/bin/bash
# development builds of the engine with other unroll depths of the dictionary SpMV's block-column loop (tools/_build/libpcg_du<k>.so);
# A/B: PCG_LIB=tools/_build/libpcg_du9.so python tools/prof_op.py dict 150 20
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
for u in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -pthread -shared -Wl,-Bsymbolic -DPCG_DICT_UNROLL=$u \
      -Iinclude -Ipcg-mpi-solver_amd/csrc pcg-mpi-solver_amd/csrc/{hip_backend.hip,rccl_comm.hip,part_setup.hip,pcg_driver.cpp,group.cpp,assemble.cpp,sell.cpp,ebe.cpp} \
      -ldl -o tools/_build/libpcg_du$u.so &
done
wait
ls -la tools/_build/libpcg_du*.so
