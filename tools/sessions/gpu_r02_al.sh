#!/bin/bash
# round 2, session AL: k_spmv_dict with 80-byte table entries (ds_read_b128 instead of ds_read2_b64 pairs): parity + A/B at 10 M dof
set -x
mkdir -p gpurun_out/r02al
timeout 600 python -m pytest tests/test_dictionary_format.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r02al/pytest.log
cat gpurun_out/r02al/pytest.log
for b in 512 256 1024; do
  PCG_SPMV_DICT_BLOCK=$b timeout 300 python tools/dict_lab.py 150 200 dict > gpurun_out/r02al/lab_150_b$b.log 2>&1; tail -1 gpurun_out/r02al/lab_150_b$b.log
done
timeout 300 python tools/dict_lab.py 70 200 dict > gpurun_out/r02al/lab_70.log 2>&1; tail -1 gpurun_out/r02al/lab_70.log
