#!/bin/bash
# Development: build ablated copies of the engine (k_spmv with parts of its access stream removed) into tools/_build/.
# The ablated kernels compute WRONG results by design; they only answer "what does each part cost on this box".
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"; C="$R/pcg-mpi-solver_amd/csrc"; O="$R/tools/_build"; mkdir -p "$O"
for m in 1 2 5; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -pthread -shared -Wl,-Bsymbolic -DPCG_SPMV_ABL=$m \
     -I"$R/include" -I"$C" "$C/hip_backend.hip" "$C/rccl_comm.hip" "$C/part_setup.hip" "$C/pcg_driver.cpp" "$C/assemble.cpp" "$C/sell.cpp" "$C/ebe.cpp" -ldl -o "$O/libpcg_sabl$m.so" &
done
wait
ls -la "$O"
