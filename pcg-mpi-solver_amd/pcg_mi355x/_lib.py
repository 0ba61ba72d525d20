"""ctypes binding of libpcg_mi355x.so (the C ABI declared in include/pcg_mi355x.h).

This is the stub a maintainer of the reference would add to call the engine.
There is no fallback: if the shared library is missing, or it has no usable gfx950 device,
`lib()` / `pcg_create` raise.
"""
from __future__ import annotations

import ctypes as C
import os

# torch bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  Import it
# BEFORE dlopen-ing the engine so that both resolve to one runtime instance (streams and device
# pointers are shared with torch.distributed in dist.py); torch itself is only plumbing here.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libpcg_mi355x.so")

_lib = None
_path = None


class PcgError(RuntimeError):
    pass


class ElemGroup(C.Structure):
    _fields_ = [("nd", C.c_int32), ("ne", C.c_int64), ("dof", C.c_void_p), ("sign", C.c_void_p),
                ("ck", C.c_void_p), ("ke", C.c_void_p)]


HALO_BEGIN_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
HALO_END_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
ALLREDUCE_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p)


class CommHooks(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("halo_begin", HALO_BEGIN_T), ("halo_end", HALO_END_T),
                ("allreduce", ALLREDUCE_T), ("collective_exchange", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("flag", C.c_int32), ("status", C.c_int32), ("iter", C.c_int64), ("iters_done", C.c_int64),
                ("n_matvec", C.c_int64), ("relres", C.c_double), ("norm_b", C.c_double),
                ("normr_act", C.c_double), ("t_total_s", C.c_double), ("t_comm_s", C.c_double),
                ("spmv_ms_sum", C.c_double), ("spmv_count", C.c_int64), ("iters_enqueued", C.c_int64),
                ("vec_ms_sum", C.c_double), ("vec_count", C.c_int64), ("fused_fallbacks", C.c_int64)]


class CommStats(C.Structure):
    _fields_ = [("halo_wait_ms", C.c_double), ("allreduce_ms", C.c_double), ("n_halo", C.c_int64),
                ("n_allreduce", C.c_int64), ("n_halo_timed", C.c_int64), ("n_allreduce_timed", C.c_int64)]


ABI_VERSION = 7            # include/pcg_mi355x.h PCG_ABI_VERSION: the struct layouts above belong to this version
RCCL_ID_BYTES = 256
FORMAT_DICTIONARY = 0x100

STATUS_NORMAL, STATUS_ZERO_RHS, STATUS_GOOD_X0, STATUS_TOO_SMALL_TOL, STATUS_RUNNING = range(5)

_P = C.c_void_p
_SIGS = {
    "pcg_abi_version": (C.c_int, []),
    "pcg_last_error": (C.c_char_p, []),
    "pcg_backend_name": (C.c_char_p, []),
    "pcg_device_count": (C.c_int, []),
    "pcg_asm_create": (C.c_int, [C.c_int64, C.c_int32, C.POINTER(ElemGroup), _P, C.c_int32, C.POINTER(_P)]),
    "pcg_asm_nnzb": (C.c_int64, [_P]),
    "pcg_asm_rowptr": (C.c_int, [_P, _P]),
    "pcg_asm_fill": (C.c_int, [_P, _P, _P]),
    "pcg_asm_destroy": (None, [_P]),
    "pcg_create": (C.c_int, [C.c_int32, C.c_int64, _P, _P, _P, C.c_int64, C.c_int32, C.POINTER(_P)]),
    "pcg_create_asm": (C.c_int, [C.c_int32, _P, C.c_int64, C.c_int32, C.POINTER(_P)]),
    "pcg_create_csr": (C.c_int, [C.c_int32, C.c_int64, _P, _P, _P, C.c_int64, C.c_int32, C.POINTER(_P)]),
    "pcg_create_scalar_copy": (C.c_int, [_P, C.POINTER(_P)]),
    "pcg_create_ebe": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, C.POINTER(ElemGroup), _P, C.c_int64, _P, C.c_int32, C.POINTER(_P)]),
    "pcg_destroy": (None, [_P]),
    "pcg_set_masks": (C.c_int, [_P, _P]),
    "pcg_set_halo": (C.c_int, [_P, C.c_int32, _P, _P, _P]),
    "pcg_set_comm": (C.c_int, [_P, C.POINTER(CommHooks)]),
    "pcg_stream": (_P, [_P]),
    "pcg_part_interface": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, _P, _P, _P, C.c_int64, _P, C.POINTER(C.c_int64)]),
    "pcg_part_local_numbering": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, _P, _P, _P, C.POINTER(C.c_int64)]),
    "pcg_rccl_unique_id": (C.c_int, [_P]),
    "pcg_comm_create_rccl": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P, C.POINTER(_P)]),
    "pcg_comm_destroy": (None, [_P]),
    "pcg_comm_rank": (C.c_int, [_P]),
    "pcg_comm_size": (C.c_int, [_P]),
    "pcg_set_comm_native": (C.c_int, [_P, _P]),
    "pcg_comm_set_timing": (C.c_int, [_P, C.c_int32]),
    "pcg_comm_enable_mailbox": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32)]),
    "pcg_enable_direct_exchange": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32)]),
    "pcg_comm_get_stats": (C.c_int, [_P, C.POINTER(CommStats)]),
    "pcg_apply": (C.c_int, [_P, _P, _P]),
    "pcg_diag": (C.c_int, [_P, _P]),
    "pcg_build_jacobi": (C.c_int, [_P, _P]),
    "pcg_update_bc": (C.c_int, [_P, _P, _P, C.c_double, _P, _P]),
    "pcg_dot_w": (C.c_int, [_P, _P, _P, C.POINTER(C.c_double)]),
    "pcg_solve_begin": (C.c_int, [_P, _P, _P, _P, C.c_double, C.c_int64, C.c_int64]),
    "pcg_solve_run": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.POINTER(Result)]),
    "pcg_solve_end": (C.c_int, [_P, _P, C.POINTER(Result)]),
    "pcg_solve": (C.c_int, [_P, _P, _P, _P, C.c_double, C.c_int64, C.c_int64, _P, _P, C.c_int64, C.POINTER(Result)]),
    "pcg_set_profiling": (C.c_int, [_P, C.c_int32]),
    "pcg_engine_device": (C.c_int, [_P]),
    "pcg_group_create": (C.c_int, [C.c_int32, _P, C.POINTER(_P)]),
    "pcg_group_destroy": (None, [_P]),
    "pcg_group_size": (C.c_int, [_P]),
    "pcg_group_device": (C.c_int, [_P, C.c_int32]),
    "pcg_group_comm": (_P, [_P, C.c_int32]),
    "pcg_group_attach": (C.c_int, [_P, C.c_int32, _P]),
    "pcg_group_apply": (C.c_int, [_P, _P, _P]),
    "pcg_group_diag": (C.c_int, [_P, _P]),
    "pcg_group_build_jacobi": (C.c_int, [_P, _P]),
    "pcg_group_update_bc": (C.c_int, [_P, _P, _P, C.c_double, _P, _P]),
    "pcg_group_dot_w": (C.c_int, [_P, _P, _P, C.POINTER(C.c_double)]),
    "pcg_group_solve": (C.c_int, [_P, _P, _P, _P, C.c_double, C.c_int64, C.c_int64, _P, _P, C.c_int64, _P]),
    "pcg_group_set_timing": (C.c_int, [_P, C.c_int32]),
    "pcg_group_enable_mailbox": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32)]),
    "pcg_group_enable_direct_exchange": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32)]),
    "pcg_bench_spmv": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "pcg_bench_hbm": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, _P]),
    "pcg_operator_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "pcg_operator_cost": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pcg_matrix_info": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "pcg_matrix_fingerprint": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "pcg_tuning_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "pcg_matrix_dictionary": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "pcg_k_update_p": (C.c_int, [_P, _P, _P, _P, C.c_double, C.c_int32]),
    "pcg_k_fused_update": (C.c_int, [_P, C.c_double, _P, _P, _P, _P, _P, _P, _P]),
    "pcg_k_vec_iteration": (C.c_int, [_P, C.c_double, C.c_double, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int32]),
    "pcg_k_residual": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "pcg_k_spmv_local": (C.c_int, [_P, _P, _P, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)


def use_library(path: str | None):
    """Load the engine from an explicit path (None = the in-tree product build).  The CPU test-suite
    uses this to load its test double (tests/hostops); the package itself never calls it."""
    global _lib, _path
    p = path or DEFAULT_PATH
    if not os.path.exists(p):
        raise PcgError(f"{p} not found - build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950); "
                       "there is no CPU fallback")
    lib = C.CDLL(p)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    if lib.pcg_abi_version() != ABI_VERSION:         # the structs carry no size field: refuse a library of another header version
        raise PcgError(f"{p} has ABI version {lib.pcg_abi_version()}, this binding is written for {ABI_VERSION} (include/pcg_mi355x.h)")
    _lib, _path = lib, p
    return lib


def lib():
    if _lib is None:
        use_library(None)
    return _lib


def library_path():
    return _path


def backend_name() -> str:
    return lib().pcg_backend_name().decode()


def check(rc: int, what: str = ""):
    if rc != 0:
        raise PcgError(f"{what}: {lib().pcg_last_error().decode()} (rc={rc})")
