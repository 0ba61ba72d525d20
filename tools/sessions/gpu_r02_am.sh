#!/bin/bash
# round 2, session AM: value dictionary larger than LDS (frequency-ordered table, head in LDS, tail through the caches): parity + octree A/B
set -x
mkdir -p gpurun_out/r02am
timeout 600 python -m pytest tests/test_dictionary_format.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r02am/pytest.log
cat gpurun_out/r02am/pytest.log
timeout 200 python tools/dict_lab.py 0 200 > gpurun_out/r02am/lab_octree.log 2>&1; tail -3 gpurun_out/r02am/lab_octree.log
PCG_SPMV_DICT_LDS=0 timeout 200 python tools/dict_lab.py 0 200 dict > gpurun_out/r02am/lab_octree_nolds.log 2>&1; tail -1 gpurun_out/r02am/lab_octree_nolds.log
