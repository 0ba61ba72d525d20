"""The plain-C example of the device-group ABI on the GPU (runs last: it only adds a C front end to paths the suite has
already been through - tests/test_group.py - and must not stand in front of them under `pytest -x`)."""
import subprocess

import pytest

from test_abi import _build_c_example


@pytest.mark.gpu
def test_group_c_example_on_gpu(gpu_lib, tmp_path):
    """examples/solve_group.c with one part per visible GPU (a group of one on a 1-GPU box: real librccl at world size 1;
    RCCL over xGMI between the members on a multi-GPU box), solution checked against the closed form inside the program."""
    exe = _build_c_example(tmp_path, "solve_group")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "hip-gfx950" in r.stdout and "flag 0" in r.stdout
