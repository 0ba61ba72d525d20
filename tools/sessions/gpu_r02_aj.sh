#!/bin/bash
# round 2, session AJ: value-dictionary format of the assembled operator (k_spmv_dict): parity on the GPU, A/B at 1 M and 10 M dof
set -x
mkdir -p gpurun_out/r02aj
timeout 600 python -m pytest tests/test_dictionary_format.py tests/test_native_comm.py -m gpu -x -q -k "dictionary or torch_nccl" 2>&1 | tail -15 > gpurun_out/r02aj/pytest.log
cat gpurun_out/r02aj/pytest.log
timeout 300 python tools/dict_lab.py 70 200 > gpurun_out/r02aj/lab_70.log 2>&1; tail -4 gpurun_out/r02aj/lab_70.log
timeout 400 python tools/dict_lab.py 150 200 > gpurun_out/r02aj/lab_150.log 2>&1; tail -4 gpurun_out/r02aj/lab_150.log
PCG_SPMV_DICT_LDS=0 timeout 300 python tools/dict_lab.py 150 100 dict > gpurun_out/r02aj/lab_150_nolds.log 2>&1; tail -2 gpurun_out/r02aj/lab_150_nolds.log
