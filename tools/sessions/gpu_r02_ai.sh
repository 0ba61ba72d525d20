#!/bin/bash
# round 2, session AI: the device group (pcg_group_*) on the HIP engine + the entry points that now bind their device
set -x
mkdir -p gpurun_out/r02ai
timeout 900 python -m pytest tests/test_group.py tests/test_partition.py tests/test_abi.py tests/test_native_comm.py -m gpu -x -q \
    -k "group or load_step or abi or world_size_1 or threads" 2>&1 | tail -15 > gpurun_out/r02ai/pytest_group.log
cat gpurun_out/r02ai/pytest_group.log
