#!/bin/bash
# round 2, final evidence on the final code, part 1: the whole GPU suite (no -x: every test reports) + smoke()
set -x
mkdir -p gpurun_out/r02final3
rocm-smi --showclocks --showpower --showmemuse --showcomputepartition --showmemorypartition > gpurun_out/r02final3/rocm_smi.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -40 > gpurun_out/r02final3/pytest_gpu.log
tail -25 gpurun_out/r02final3/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02final3/smoke.log 2>&1; tail -4 gpurun_out/r02final3/smoke.log
