import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "pcg-mpi-solver_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

# the reference pins BLAS to one thread (pcg_solver.py:10-15); the oracle restates that
for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(k, "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HOSTOPS_SRC = os.path.join(ROOT, "tests", "hostops", "host_backend.cpp")
HOSTOPS_LIB = os.path.join(ROOT, "tests", "hostops", "_build", "libpcg_hostops.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def build_hostops():
    """Compile the CPU test double (tests/hostops) + the product's back-end-agnostic C++ sources."""
    import subprocess
    csrc = os.path.join(ROOT, "pcg-mpi-solver_amd", "csrc")
    srcs = [HOSTOPS_SRC, os.path.join(os.path.dirname(HOSTOPS_SRC), "local_comm.cpp")] + \
           [os.path.join(csrc, f) for f in ("pcg_driver.cpp", "group.cpp", "assemble.cpp", "sell.cpp", "ebe.cpp")]
    deps = srcs + [os.path.join(csrc, "pcg_internal.hpp"), os.path.join(ROOT, "include", "pcg_mi355x.h")]
    if os.path.exists(HOSTOPS_LIB) and all(os.path.getmtime(HOSTOPS_LIB) >= os.path.getmtime(d) for d in deps):
        return HOSTOPS_LIB
    os.makedirs(os.path.dirname(HOSTOPS_LIB), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-Wl,-Bsymbolic",
           "-I" + os.path.join(ROOT, "include"), "-I" + csrc] + srcs + ["-o", HOSTOPS_LIB]
    subprocess.check_call(cmd)
    return HOSTOPS_LIB


FAKENCCL_SRC = os.path.join(ROOT, "tests", "fakenccl", "fakenccl.cpp")
FAKENCCL_LIB = os.path.join(ROOT, "tests", "fakenccl", "_build", "libfakenccl.so")


def build_fakenccl():
    """The shared-GPU stand-in for librccl (tests/fakenccl): several ranks on ONE device drive the engine's native
    communicator code.  Host-only code, compiled with hipcc for the HIP runtime headers."""
    import subprocess
    if os.path.exists(FAKENCCL_LIB) and os.path.getmtime(FAKENCCL_LIB) >= os.path.getmtime(FAKENCCL_SRC):
        return FAKENCCL_LIB
    os.makedirs(os.path.dirname(FAKENCCL_LIB), exist_ok=True)
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread",
                           FAKENCCL_SRC, "-o", FAKENCCL_LIB])
    return FAKENCCL_LIB


def build_oracle_c():
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def oracle_c():
    build_oracle_c()
    return True


@pytest.fixture()
def hostops():
    """Point the ctypes binding at the CPU TEST DOUBLE for the duration of one test."""
    from pcg_mi355x import _lib
    prev = _lib.library_path()
    _lib.use_library(build_hostops())
    yield _lib
    _lib._lib = None
    _lib._path = None
    if prev and os.path.exists(prev) and prev != HOSTOPS_LIB:
        _lib.use_library(prev)


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library on a real GPU; fails (does not skip) when it is missing on a GPU box."""
    import __graft_entry__
    __graft_entry__.build_engine()          # no-op when the in-tree build is current; a clone without build products compiles it
    from pcg_mi355x import _lib
    _lib.use_library(None)
    assert _lib.backend_name() == "hip-gfx950"
    assert _lib.lib().pcg_device_count() >= 1, "no gfx950 device visible"
    return _lib
