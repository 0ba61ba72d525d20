#!/bin/bash
# Development: build the engine of another git revision next to the product's library, for same-box A/B runs on the GPU
# (the box-to-box spread of the kernels is larger than most changes).  usage: tools/build_variant.sh <git-rev> <name>
#   -> pcg-mpi-solver_amd/lib/ab/libpcg_<name>.so   (git-ignored; travels with the gpurun snapshot; PCG_LIB=<path> python tools/iter_ab.py ...)
set -eu
REV="$1"; NAME="$2"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TMP="$(mktemp -d /tmp/pcg_variant.XXXXXX)"
git -C "$ROOT" archive "$REV" pcg-mpi-solver_amd/csrc include | tar -x -C "$TMP"
OUT="$ROOT/pcg-mpi-solver_amd/lib/ab"; mkdir -p "$OUT"
cd "$TMP/pcg-mpi-solver_amd/csrc"
for s in hip_backend.hip rccl_comm.hip part_setup.hip pcg_driver.cpp group.cpp assemble.cpp sell.cpp ebe.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -pthread -I"$TMP/include" -I. -c "$s" -o "$s.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -Wl,-Bsymbolic ./*.o -ldl -o "$OUT/libpcg_$NAME.so"
rm -rf "$TMP"
echo "$OUT/libpcg_$NAME.so"
