"""The product library: builds for gfx950 without a GPU, exports every symbol declared in
include/pcg_mi355x.h, has exactly one back end and refuses to run without a device."""
import ctypes
import os
import re
import subprocess

import pytest

from util import ROOT


@pytest.fixture(scope="module")
def product_lib():
    import __graft_entry__ as ge
    return ge.build_engine()


def test_header_symbols_exported(product_lib):
    hdr = open(os.path.join(ROOT, "include", "pcg_mi355x.h")).read()
    declared = set(re.findall(r"\b(pcg_[a-z0-9_]+)\s*\(", hdr)) - {"pcg_comm_hooks"}
    lib = ctypes.CDLL(product_lib)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    from pcg_mi355x import _lib
    assert set(_lib.EXPORTED_SYMBOLS) == declared


def test_clean_clone_builds_with_hipcc_for_gfx950(tmp_path):
    """Round 3 verdict: the in-tree build is digest-gated and its objects travel with the tree, so `build()` may link nothing.  Here
    the SOURCES alone (csrc/, include/, __graft_entry__.py - what a fresh clone holds) are copied to an empty directory and compiled
    from scratch with hipcc --offload-arch=gfx950; the result must export every symbol include/pcg_mi355x.h declares, carry gfx950
    code objects, and be a different file from the in-tree library."""
    import importlib.util
    import shutil
    pkg = tmp_path / "pcg-mpi-solver_amd"
    shutil.copytree(os.path.join(ROOT, "pcg-mpi-solver_amd", "csrc"), pkg / "csrc")
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    shutil.copy(os.path.join(ROOT, "__graft_entry__.py"), tmp_path / "__graft_entry__.py")
    assert not (pkg / "build").exists() and not (pkg / "lib").exists()
    spec = importlib.util.spec_from_file_location("graft_entry_clean_clone", tmp_path / "__graft_entry__.py")
    ge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ge)
    assert ge.ROOT == str(tmp_path)
    lib_path = ge.build_engine(force=True)
    assert lib_path.startswith(str(tmp_path)) and os.path.getsize(lib_path) > 1_000_000
    assert sorted(os.listdir(pkg / "build")) == sorted([s + ext for s in ge.SOURCES for ext in (".o", ".o.sha")])
    hdr = open(os.path.join(ROOT, "include", "pcg_mi355x.h")).read()
    declared = set(re.findall(r"\b(pcg_[a-z0-9_]+)\s*\(", hdr)) - {"pcg_comm_hooks"}
    lib = ctypes.CDLL(lib_path)
    assert not [s for s in sorted(declared) if not hasattr(lib, s)]
    lib.pcg_abi_version.restype = ctypes.c_int
    assert lib.pcg_abi_version() == int(re.search(r"#define PCG_ABI_VERSION (\d+)", hdr).group(1))
    assert "gfx950" in subprocess.run(["strings", lib_path], capture_output=True, text=True).stdout


def test_product_has_only_the_hip_backend(product_lib):
    lib = ctypes.CDLL(product_lib)
    lib.pcg_backend_name.restype = ctypes.c_char_p
    assert lib.pcg_backend_name() == b"hip-gfx950"
    syms = subprocess.run(["nm", "-DC", product_lib], capture_output=True, text=True).stdout
    assert "HostBackend" not in syms and "k_spmv" in subprocess.run(
        ["strings", product_lib], capture_output=True, text=True).stdout


def test_fails_loudly_without_gpu(product_lib):
    lib = ctypes.CDLL(product_lib)
    if lib.pcg_device_count() > 0:
        pytest.skip("a GPU is visible")
    import numpy as np
    h = ctypes.c_void_p()
    rp = np.array([0, 1], np.int64); c = np.zeros(1, np.int32); v = np.eye(3).ravel()
    lib.pcg_create.argtypes = [ctypes.c_int32, ctypes.c_int64] + [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int32,
                                                                                           ctypes.POINTER(ctypes.c_void_p)]
    rc = lib.pcg_create(0, 1, rp.ctypes.data, c.ctypes.data, v.ctypes.data, 0, 0, ctypes.byref(h))
    lib.pcg_last_error.restype = ctypes.c_char_p
    assert rc != 0 and b"no CPU fallback" in lib.pcg_last_error()


def test_package_does_not_import_oracle():
    pkg = os.path.join(ROOT, "pcg-mpi-solver_amd", "pcg_mi355x")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "pcg_oracle" not in src and "import oracle" not in src and "ref_shim" not in src, f


def _build_c_example(tmp_path, name="solve_csr", double=False):
    """gcc examples/<name>.c against the product library (or, double=True, the CPU test double: same C ABI)."""
    import subprocess
    from util import ROOT
    if double:
        import conftest
        libdir, lib = os.path.dirname(conftest.build_hostops()), "pcg_hostops"
    else:
        import __graft_entry__
        __graft_entry__.build_engine()
        libdir, lib = os.path.join(ROOT, "pcg-mpi-solver_amd", "lib"), "pcg_mi355x"
    exe = str(tmp_path / (name + ("_double" if double else "")))
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", name + ".c"),
                           "-L" + libdir, "-l" + lib, "-Wl,-rpath," + libdir, "-lm", "-o", exe])
    return exe


def test_group_c_example(tmp_path):
    """examples/solve_group.c - pcg_group_* from plain C: a spring chain cut into G parts (interface nodes first, ownership
    flags, neighbour lists as the reference's partitioner builds them), solved by ONE process, checked against its closed
    form.  Links against the product library (and stops at the device check without a GPU); its logic runs here against
    the CPU test double, which exports the same C ABI: 1, 2, 4 and 8 parts take the same number of iterations."""
    import subprocess
    import torch
    exe = _build_c_example(tmp_path, "solve_group", double=True)
    its = set()
    for g in ("1", "2", "4", "8"):
        r = subprocess.run([exe, g, "193"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        assert f"{g} part(s)" in r.stdout and "flag 0" in r.stdout
        its.add(r.stdout.split(" iterations")[0].split()[-1])
    assert len(its) == 1
    assert subprocess.run([exe, "5", "193"], capture_output=True, text=True).returncode == 1       # 192 springs do not split into 5
    prod = _build_c_example(tmp_path, "solve_group")
    if not torch.cuda.is_available():
        r = subprocess.run([prod, "2"], capture_output=True, text=True)
        assert r.returncode == 2 and "no HIP device" in r.stderr


def test_c_abi_from_plain_c_without_gpu(tmp_path):
    """examples/solve_csr.c: the header compiles as C, the library links without Python / torch, and the program
    stops at the device check when there is no GPU (no CPU fallback)."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe, "8"], capture_output=True, text=True)
    assert r.returncode == 2 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_c_abi_from_plain_c_on_gpu(tmp_path):
    """The same program on the GPU: a 110 592-row 7-point Laplacian (n % 3 = 0 not required: scalar format) to 1e-9."""
    import subprocess
    exe = _build_c_example(tmp_path)
    r = subprocess.run([exe, "48"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "hip-gfx950" in r.stdout and "flag 0" in r.stdout


def test_device_entry_points_fail_loudly_without_gpu(product_lib):
    """The round-2 entry points (native communicator, device-side partition set-up) have no CPU path either: without a
    device they return an error code and a message, they do not fall back or crash."""
    import numpy as np
    lib = ctypes.CDLL(product_lib)
    if lib.pcg_device_count() > 0:
        pytest.skip("a GPU is visible")
    lib.pcg_last_error.restype = ctypes.c_char_p
    ptr = np.array([0, 2, 4], np.int64); flat = np.array([0, 1, 1, 2], np.int32); part = np.array([0, 1], np.int32)
    pairs = np.zeros((8, 2), np.int64); n = ctypes.c_int64()
    lib.pcg_part_interface.argtypes = [ctypes.c_int32, ctypes.c_int64, ctypes.c_int64] + [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p,
                                                                                                                  ctypes.POINTER(ctypes.c_int64)]
    rc = lib.pcg_part_interface(0, 3, 2, ptr.ctypes.data, flat.ctypes.data, part.ctypes.data, 8, pairs.ctypes.data, ctypes.byref(n))
    assert rc != 0 and b"no HIP device" in lib.pcg_last_error()
    bad = np.array([0, 7, 1, 2], np.int32)                       # node id out of range: rejected before any device work
    rc = lib.pcg_part_interface(0, 3, 2, ptr.ctypes.data, bad.ctypes.data, part.ctypes.data, 8, pairs.ctypes.data, ctypes.byref(n))
    assert rc != 0 and b"out of range" in lib.pcg_last_error()
    uid = ctypes.create_string_buffer(256)
    comm = ctypes.c_void_p()
    lib.pcg_comm_create_rccl.argtypes = [ctypes.c_int32] * 3 + [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    rc = lib.pcg_comm_create_rccl(0, 0, 1, uid, ctypes.byref(comm))
    assert rc != 0 and lib.pcg_last_error()


def test_test_double_talks_through_process_memory_only(hostops):
    """tests/hostops is a CPU double of the KERNELS; its stand-in for the native communicator (local_comm.cpp, in-process
    mailboxes for the group tests) is not RCCL, and the device partition passes exist only in the product library."""
    import numpy as np
    uid = ctypes.create_string_buffer(256)
    assert hostops.lib().pcg_rccl_unique_id(uid) == 0 and uid.raw.startswith(b"pcg-hostops-local-comm")
    n = ctypes.c_int64()
    rc = hostops.lib().pcg_part_local_numbering(0, 4, 2, np.zeros(2, np.int32).ctypes.data, np.zeros(2, np.int32).ctypes.data,
                                                np.zeros(2, np.int32).ctypes.data, ctypes.byref(n))
    assert rc != 0 and b"no device-side partition set-up" in hostops.lib().pcg_last_error()
