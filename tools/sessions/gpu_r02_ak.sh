#!/bin/bash
# round 2, session AK: k_spmv_dict workgroup size A/B (256 / 512 / 1024 threads per copy of the table) at 10 M dof; GPU parity tests
set -x
mkdir -p gpurun_out/r02ak
timeout 600 python -m pytest tests/test_dictionary_format.py tests/test_native_comm.py -m gpu -x -q -k "dictionary or torch_nccl" 2>&1 | tail -8 > gpurun_out/r02ak/pytest.log
cat gpurun_out/r02ak/pytest.log
for b in 256 512 1024; do
  PCG_SPMV_DICT_BLOCK=$b timeout 300 python tools/dict_lab.py 150 200 dict > gpurun_out/r02ak/lab_150_b$b.log 2>&1; tail -1 gpurun_out/r02ak/lab_150_b$b.log
done
PCG_SPMV_DICT_BLOCK=512 timeout 300 python tools/dict_lab.py 70 200 dict > gpurun_out/r02ak/lab_70_b512.log 2>&1; tail -1 gpurun_out/r02ak/lab_70_b512.log
