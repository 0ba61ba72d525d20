#!/bin/bash
# round 2, final code (contiguous-VRAM request off, corrected dictionary test): the GPU suite without the two 90-second lock-step tests
mkdir -p gpurun_out/r02final6
timeout 135 python -m pytest tests -m gpu -q -k "not lock_step" 2>&1 | tail -6 > gpurun_out/r02final6/pytest_gpu_no_lockstep.log
cat gpurun_out/r02final6/pytest_gpu_no_lockstep.log
