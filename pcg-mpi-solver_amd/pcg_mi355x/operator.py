"""RefMeshPart -> engine operator (host-side set-up of the drop-in).

`from_refmeshpart()` consumes exactly the arrays the reference's calcMatVecProd reads
(src/solver/pcg_solver.py:245-250,263-276: SubDomainData['StrucDataList'][j] tables,
OvrlpLocalDofVecList, NbrMPIdVector, NDOF) plus the ownership weights and the free-dof map
(partition_mesh.py:868-887, :350-351), assembles the part's un-exchanged sub-domain matrix
A = sum_e P_e^T S_e (Ck_e Ke_type) S_e P_e once (native host code, pcg_asm_*), renumbers the nodes
interface-first so the exchange can overlap the interior rows, and uploads it to the GPU.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import check, PcgError

__all__ = ["Operator", "from_refmeshpart", "assemble_bsr3"]


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _pack_groups(groups):
    """ctypes view of SubDomainData['StrucDataList'] (the arrays calcMatVecProd reads, :267-276)."""
    keep = []
    arr = (_lib.ElemGroup * len(groups))()
    for k, g in enumerate(groups):
        tbl = np.ascontiguousarray(g["ElemList_LocDofVector"], dtype=np.int64)
        nd, ne = tbl.shape
        sign = np.ascontiguousarray(g["ElemList_SignVector"], dtype=np.uint8)
        if sign.shape != tbl.shape:
            raise ValueError("ElemList_SignVector shape mismatch")
        ck = _f64(g["ElemList_Ck"])
        ke = _f64(g["ElemStiffMat"])
        if ke.shape != (nd, nd) or ck.shape != (ne,):
            raise ValueError("ElemStiffMat / ElemList_Ck shape mismatch")
        keep += [tbl, sign, ck, ke]
        arr[k] = _lib.ElemGroup(nd, ne, tbl.ctypes.data, sign.ctypes.data, ck.ctypes.data, ke.ctypes.data)
    return arr, keep


def assemble_bsr3(groups, n_nodes, node_perm=None, n_threads=0):
    """Run the native assembler on reference type groups -> (rowptr i64, cols i32, vals f64[nnzb,3,3])."""
    L = _lib.lib()
    arr, keep = _pack_groups(groups)
    perm = None
    if node_perm is not None:
        perm = np.ascontiguousarray(node_perm, dtype=np.int64)
    h = C.c_void_p()
    check(L.pcg_asm_create(n_nodes, len(groups), arr, perm.ctypes.data if perm is not None else None,
                           n_threads, C.byref(h)), "pcg_asm_create")
    try:
        nnzb = L.pcg_asm_nnzb(h)
        rowptr = np.empty(n_nodes + 1, np.int64)
        cols = np.empty(nnzb, np.int32)
        vals = np.empty((nnzb, 3, 3), np.float64)
        check(L.pcg_asm_rowptr(h, rowptr.ctypes.data), "pcg_asm_rowptr")
        check(L.pcg_asm_fill(h, cols.ctypes.data, vals.ctypes.data), "pcg_asm_fill")
    finally:
        L.pcg_asm_destroy(h)
    return rowptr, cols, vals


class Operator:
    """One part's operator on one GPU.  Vectors in/out are NumPy f64 of the part's local length
    (`NDOF`) in the REFERENCE's local numbering; the interface-first renumbering is internal."""

    def __init__(self, n_nodes, rowptr=None, cols=None, vals=None, n_boundary_nodes=0, dof_new_of_old=None, device=0,
                 rows_per_lane=0, ebe_groups=None, node_perm=None, node_coords=None, ebe_chunked=True):
        """Assembled operator from 3x3-block CSR (rowptr/cols/vals), or - with `ebe_groups` - the
        matrix-free operator straight from the reference's type-group tables."""
        L = _lib.lib()
        self._L = L
        self.n_nodes = int(n_nodes)
        self.n = 3 * self.n_nodes
        self._map = None if dof_new_of_old is None else np.ascontiguousarray(dof_new_of_old, dtype=np.int64)
        h = C.c_void_p()
        if ebe_groups is not None:
            self.kind = "ebe"
            arr, keep = _pack_groups(ebe_groups)
            perm = None if node_perm is None else np.ascontiguousarray(node_perm, np.int64)
            xyz = None if node_coords is None else _f64(np.asarray(node_coords).reshape(self.n_nodes, 3))
            # hex8 chunk size: 512 elements (an 8x8x8 cell, contracted in two passes of 256: k_ebe_hexs) unless that leaves fewer
            # chunks than a few per CU - then 256 (k_ebe_hex).  Measured: profiles/r02_ebe_lab_*.log; PCG_EBE_EPT overrides.
            n_elem = sum(int(np.asarray(g["ElemList_Ck"]).shape[0]) for g in ebe_groups)
            # (1 M dof / 328 k elements: 0.034 ms per apply with 717 chunks of 512 vs 0.040 with 1433 of 256; 59 k elements: 256 wins)
            ept = os.environ.get("PCG_EBE_EPT", "1" if n_elem < 256 * 512 else "2")
            # one phase (flags bit 2): no interface-first launch - for jobs that exchange AFTER the whole operator, i.e. with the direct
            # exchange (PCG_DIRECT_EXCHANGE=1 on every rank implies it; PCG_EBE_ONE_PHASE=0|1 overrides)
            one_phase = os.environ.get("PCG_EBE_ONE_PHASE", "1" if os.environ.get("PCG_DIRECT_EXCHANGE") == "1" else "0") == "1"
            check(L.pcg_create_ebe(device, self.n_nodes, len(ebe_groups), arr, perm.ctypes.data if perm is not None else None,
                                   int(n_boundary_nodes), xyz.ctypes.data if xyz is not None else None,
                                   (0 if ebe_chunked else 1) | (2 if ept == "1" else 0) | (4 if one_phase else 0), C.byref(h)), "pcg_create_ebe")
            self.nnzb = self.nnz = 0
        else:
            self.kind = "sell"
            self.nnzb = int(rowptr[-1])
            self.nnz = 9 * self.nnzb
            rowptr = np.ascontiguousarray(rowptr, np.int64)
            cols = np.ascontiguousarray(cols, np.int32)
            vals = _f64(vals)
            check(L.pcg_create(device, self.n_nodes, rowptr.ctypes.data, cols.ctypes.data, vals.ctypes.data,
                               int(n_boundary_nodes), int(rows_per_lane), C.byref(h)), "pcg_create")
        self._h = h
        self._comm = None
        self._hooks = None
        self.glob_n_eff = None
        self.last_result = None

    @classmethod
    def from_assembler(cls, groups, n_nodes, node_perm=None, n_boundary_nodes=0, dof_new_of_old=None, device=0, rows_per_lane=0,
                       n_threads=0):
        """Assembled operator straight from the reference's type-group tables through the native assembler (pcg_asm_create ->
        pcg_create_asm): no 3x3-block CSR arrays on the Python side, and with rows_per_lane | FORMAT_DICTIONARY the values
        are never materialised (rows are produced once, hashed and stored as table indices)."""
        self = cls.__new__(cls)
        L = _lib.lib()
        arr, keep = _pack_groups(groups)
        perm = None if node_perm is None else np.ascontiguousarray(node_perm, dtype=np.int64)
        a = C.c_void_p()
        check(L.pcg_asm_create(int(n_nodes), len(groups), arr, perm.ctypes.data if perm is not None else None, int(n_threads), C.byref(a)),
              "pcg_asm_create")
        h = C.c_void_p()
        try:
            nnzb = int(L.pcg_asm_nnzb(a))
            check(L.pcg_create_asm(device, a, int(n_boundary_nodes), int(rows_per_lane), C.byref(h)), "pcg_create_asm")
        finally:
            L.pcg_asm_destroy(a)
        del keep
        self._L, self._h, self.kind = L, h, "sell"
        self.n_nodes, self.n = int(n_nodes), 3 * int(n_nodes)
        self._map = None if dof_new_of_old is None else np.ascontiguousarray(dof_new_of_old, dtype=np.int64)
        self.nnzb, self.nnz = nnzb, 9 * nnzb
        self._comm = self._hooks = None
        self.glob_n_eff = None
        self.last_result = None
        return self

    def matrix_fingerprint(self):
        """FNV-1a of the host-side operator arrays (0 unless PCG_MATRIX_FINGERPRINT was set when the operator was built)."""
        out = C.c_uint64()
        check(self._L.pcg_matrix_fingerprint(self._h, C.byref(out)), "pcg_matrix_fingerprint")
        return out.value

    @classmethod
    def from_csr(cls, rowptr, cols, vals, device=0, block=0):
        """Engine operator from an assembled scalar CSR matrix (e.g. scipy.sparse.csr_matrix: indptr, indices,
        data), pcg_create_csr.  block = 0/3: n = 3*nodes rows regrouped into 3x3 node blocks (the fast format);
        block = 1: scalar rows kept (any n).  Single part; every dof owned and free until set_masks()."""
        self = cls.__new__(cls)
        L = _lib.lib()
        rowptr = np.ascontiguousarray(rowptr, np.int64)
        cols = np.ascontiguousarray(cols, np.int32)
        vals = _f64(vals)
        n = len(rowptr) - 1
        h = C.c_void_p()
        check(L.pcg_create_csr(device, n, rowptr.ctypes.data, cols.ctypes.data, vals.ctypes.data, 0, int(block),
                               C.byref(h)), "pcg_create_csr")          # block may carry _lib.FORMAT_DICTIONARY
        self._L, self._h, self.kind = L, h, "sell"
        self.n, self.n_nodes, self._map = n, n // 3, None
        info = self.matrix_info()
        self.nnzb, self.nnz = info["nnzb"], (1 if (block & 0xff) == 1 else 9) * info["nnzb"]
        self._comm = self._hooks = None
        self.glob_n_eff = None
        self.last_result = None
        return self

    def scalar_copy(self):
        """The same operator in the literal CSR data volume (one f64 + one i32 column per scalar non-zero, k_spmv_scalar), expanded
        from this engine's plain 3x3-block format on the device (pcg_create_scalar_copy) - the "CSR SpMV" measurement point at the
        metric's own size.  Engine numbering and masks are NOT carried over: a measurement / test object (apply, bench_spmv)."""
        other = Operator.__new__(Operator)
        h = C.c_void_p()
        check(self._L.pcg_create_scalar_copy(self._h, C.byref(h)), "pcg_create_scalar_copy")
        other._L, other._h, other.kind = self._L, h, "sell"
        other.n, other.n_nodes, other._map = self.n, self.n_nodes, self._map
        info = other.matrix_info()
        other.nnzb = other.nnz = info["nnzb"]
        other._comm = other._hooks = None
        other.glob_n_eff = None
        other.last_result = None
        return other

    # -- numbering ------------------------------------------------------------------------------
    def to_engine(self, v):
        v = _f64(v)
        if v.shape != (self.n,):
            raise ValueError(f"vector length {v.shape} != {self.n}")
        if self._map is None:
            return v
        out = np.empty_like(v)
        out[self._map] = v
        return out

    def from_engine(self, v):
        return v if self._map is None else v[self._map]

    # -- set-up ---------------------------------------------------------------------------------
    def set_masks(self, owned, free):
        flags = (np.asarray(owned, bool).astype(np.uint8) | (np.asarray(free, bool).astype(np.uint8) << 1))
        if self._map is not None:
            f2 = np.empty_like(flags)
            f2[self._map] = flags
            flags = f2
        flags = np.ascontiguousarray(flags)
        check(self._L.pcg_set_masks(self._h, flags.ctypes.data), "pcg_set_masks")

    def set_halo(self, peer_ids, dof_lists):
        peer = np.ascontiguousarray(peer_ids, np.int32)
        ptr = np.zeros(len(peer) + 1, np.int64)
        ptr[1:] = np.cumsum([len(d) for d in dof_lists])
        idx = np.concatenate([np.asarray(d, np.int64) for d in dof_lists]) if len(peer) else np.zeros(0, np.int64)
        if self._map is not None and len(idx):
            idx = self._map[idx]
        idx = np.ascontiguousarray(idx, np.int32)
        check(self._L.pcg_set_halo(self._h, len(peer), peer.ctypes.data, ptr.ctypes.data, idx.ctypes.data), "pcg_set_halo")
        self.peer_ids = [int(p) for p in peer]
        self.peer_counts = [int(c) for c in np.diff(ptr)]

    def set_comm(self, comm):
        """comm: pcg_mi355x.dist.RcclComm (native: the engine issues the RCCL calls itself), a callback communicator
        with make_hooks() (pcg_mi355x.dist.TorchComm, the tests' thread communicator), or None."""
        self._comm = comm
        check(self._L.pcg_set_comm_native(self._h, None), "pcg_set_comm_native")
        check(self._L.pcg_set_comm(self._h, None), "pcg_set_comm")
        self._hooks = None
        if comm is None:
            return
        if getattr(comm, "native", False):
            check(self._L.pcg_set_comm_native(self._h, comm.handle), "pcg_set_comm_native")
            if os.environ.get("PCG_DIRECT_EXCHANGE") == "1":       # opt-in for a whole job (every rank sets it): see enable_direct_exchange
                self.enable_direct_exchange()
            return
        self._hooks = comm.make_hooks(self)
        check(self._L.pcg_set_comm(self._h, C.byref(self._hooks)), "pcg_set_comm")

    direct_exchange = False
    direct_exchange_reason = None

    def enable_direct_exchange(self, on=True):
        """Opt-in, COLLECTIVE (every rank's engine of the job, between solves, after set_halo and set_comm with a native
        communicator): the interface exchange of the PCG iteration (pcg_solver.py:307-328) as stores into the neighbours'
        peer-mapped receive buffers instead of grouped ncclSend / ncclRecv (include/pcg_mi355x.h pcg_enable_direct_exchange).
        -> True when every rank mapped its neighbours; False (on every rank) = RCCL stays, `direct_exchange_reason` says why."""
        got = C.c_int32(0)
        check(self._L.pcg_enable_direct_exchange(self._h, 1 if on else 0, C.byref(got)), "pcg_enable_direct_exchange")
        self.direct_exchange = bool(got.value)
        self.direct_exchange_reason = None if (self.direct_exchange or not on) else (self._L.pcg_last_error() or b"").decode(errors="replace")
        return self.direct_exchange

    def stream_ptr(self):
        return self._L.pcg_stream(self._h)

    # -- operator-level calls -------------------------------------------------------------------
    def _raise_comm(self):
        if self._comm is not None:
            self._comm.reraise()

    def apply(self, x):
        """calcMatVecProd(.., 'Strain', x) (:242-336)."""
        xe = self.to_engine(x)
        y = np.empty(self.n)
        rc = self._L.pcg_apply(self._h, xe.ctypes.data, y.ctypes.data)
        self._raise_comm(); check(rc, "pcg_apply")
        return self.from_engine(y)

    def diag(self):
        """calcMatVecProd(.., 'Preconditioner') (:282-287 + exchange)."""
        d = np.empty(self.n)
        rc = self._L.pcg_diag(self._h, d.ctypes.data)
        self._raise_comm(); check(rc, "pcg_diag")
        return self.from_engine(d)

    def build_jacobi(self):
        """updatePreconditioner (:346-352); returns 1/diag on the free dofs (0 on fixed), full length."""
        d = np.empty(self.n)
        rc = self._L.pcg_build_jacobi(self._h, d.ctypes.data)
        self._raise_comm(); check(rc, "pcg_build_jacobi")
        return self.from_engine(d)

    def update_bc(self, ref_load, ud, delta):
        """updateBC (:226-238) -> (Fext, Udi)."""
        f = self.to_engine(ref_load); u = self.to_engine(ud)
        fo = np.empty(self.n); uo = np.empty(self.n)
        rc = self._L.pcg_update_bc(self._h, f.ctypes.data, u.ctypes.data, float(delta), fo.ctypes.data, uo.ctypes.data)
        self._raise_comm(); check(rc, "pcg_update_bc")
        return self.from_engine(fo), self.from_engine(uo)

    def dot_w(self, a, b):
        out = C.c_double()
        a = self.to_engine(a); b = self.to_engine(b)
        rc = self._L.pcg_dot_w(self._h, a.ctypes.data, b.ctypes.data, C.byref(out))
        self._raise_comm(); check(rc, "pcg_dot_w")
        return out.value

    # -- PCG ------------------------------------------------------------------------------------
    def solve_begin(self, b, x0=None, inv_diag=None, tol=1e-7, max_iter=10000, glob_n_eff=None):
        be = self.to_engine(b)
        x0e = self.to_engine(x0) if x0 is not None else None
        mde = self.to_engine(inv_diag) if inv_diag is not None else None
        gne = int(glob_n_eff if glob_n_eff is not None else (self.glob_n_eff or self.n))
        rc = self._L.pcg_solve_begin(self._h, be.ctypes.data, x0e.ctypes.data if x0e is not None else None,
                                     mde.ctypes.data if mde is not None else None, float(tol), int(max_iter), gne)
        self._raise_comm(); check(rc, "pcg_solve_begin")

    def solve_run(self, n_iters=-1, hist=None):
        res = _lib.Result()
        hp, hc = (hist.ctypes.data, hist.shape[0]) if hist is not None else (None, 0)
        rc = self._L.pcg_solve_run(self._h, int(n_iters), hp, hc, C.byref(res))
        self._raise_comm(); check(rc, "pcg_solve_run")
        return res

    def solve_end(self):
        res = _lib.Result()
        x = np.empty(self.n)
        rc = self._L.pcg_solve_end(self._h, x.ctypes.data, C.byref(res))
        self._raise_comm(); check(rc, "pcg_solve_end")
        self.last_result = res
        return self.from_engine(x), res

    def solve(self, b, x0=None, inv_diag=None, tol=1e-7, max_iter=10000, glob_n_eff=None, history=False):
        """-> (x, Result, hist or None).  hist rows = [NormP, NormX, NormR] per iteration (:507)."""
        self.solve_begin(b, x0, inv_diag, tol, max_iter, glob_n_eff)
        hist = np.zeros((int(max_iter), 3)) if history else None
        self.solve_run(-1, hist)
        x, res = self.solve_end()
        if hist is not None:
            hist = hist[:max(0, min(int(res.iters_done), int(max_iter)))]
        return x, res, hist

    def set_profiling(self, on=True, what=3):
        """HIP events around the operator launches (what & 1) and the vector-phase launches (what & 2) of the solve windows."""
        check(self._L.pcg_set_profiling(self._h, int(what) if on else 0), "pcg_set_profiling")

    def bench_spmv(self, warmup=10, reps=100):
        ms = np.zeros(reps, np.float32)
        check(self._L.pcg_bench_spmv(self._h, warmup, reps, ms.ctypes.data), "pcg_bench_spmv")
        return ms

    def bench_hbm(self, nbytes=2 << 30, mode="read", reps=20):
        """GB/s of a device stream over `nbytes`: mode "read" (read-only) or "copy" (traffic = 2 * nbytes)."""
        ms = np.zeros(reps, np.float32)
        m = {"read": 0, "copy": 1}.get(mode, mode)          # 2..4: access-pattern probes (hip_backend.hip k_stream_slices)
        check(self._L.pcg_bench_hbm(self._h, int(nbytes), int(m), reps, ms.ctypes.data), "pcg_bench_hbm")
        return (2 if m == 1 else 1) * nbytes / (float(np.median(ms)) * 1e-3) / 1e9

    def operator_info(self):
        k, a, b, c, d = C.c_int32(), C.c_int64(), C.c_int64(), C.c_int32(), C.c_int64()
        check(self._L.pcg_operator_info(self._h, C.byref(k), C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "pcg_operator_info")
        return {"kind": "ebe" if k.value == 1 else "sell", "n_elem": a.value, "n_slots": b.value, "n_colors": c.value,
                "n_chunks": d.value}

    def operator_cost(self):
        """(bytes, flops) one local operator apply has to move / compute, counted from the stored structures."""
        b, f = C.c_double(), C.c_double()
        check(self._L.pcg_operator_cost(self._h, C.byref(b), C.byref(f)), "pcg_operator_cost")
        return b.value, f.value

    def tuning_info(self):
        """{"spmv_launches_per_apply", "vectors_placed"}: what the engine decided by measurement at its first solve (pcg_tuning_info, ABI 7)."""
        a, b = C.c_int32(1), C.c_int32(0)
        check(self._L.pcg_tuning_info(self._h, C.byref(a), C.byref(b)), "pcg_tuning_info")
        return {"spmv_launches_per_apply": int(a.value), "vectors_placed": bool(b.value)}

    def matrix_dictionary(self):
        """Distinct 3x3 blocks of the value dictionary (PCG_FORMAT_DICTIONARY); 0 = plain values."""
        return self.matrix_dictionary_info()["distinct_blocks"]

    def matrix_dictionary_info(self):
        """{"distinct_blocks", "in_lds": the most frequent entries the SpMV kernel keeps in LDS, "lds_share": the share of the
        stored blocks those cover}."""
        n, h, sh = C.c_int64(), C.c_int64(), C.c_double()
        check(self._L.pcg_matrix_dictionary(self._h, C.byref(n), C.byref(h), C.byref(sh)), "pcg_matrix_dictionary")
        return {"distinct_blocks": n.value, "in_lds": h.value, "lds_share": sh.value}

    def matrix_info(self):
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        check(self._L.pcg_matrix_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "pcg_matrix_info")
        return {"nnzb": a.value, "stored_blocks": b.value, "n_slices": c.value, "slice_rows": d.value}

    def close(self):
        if getattr(self, "_h", None):
            rel = getattr(getattr(self, "_comm", None), "release_stream", None)
            if rel is not None:
                rel(self.stream_ptr())
            self._L.pcg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _blocked_node_order(xyz, is_b, block=8):
    """Old node ids in the order the matrix-free engine numbers them: interface nodes first (as always), then the
    interior nodes block by block of the node lattice (block = 8 lattice units = the edge of a 512-element chunk,
    csrc/ebe.cpp), inside a block the nodes strictly inside it before those on its low faces, x fastest.
    A chunk's own (exclusive) nodes - coordinates 8i+1 .. 8i+7 - then occupy ONE contiguous index range, so the chunk's
    x-tile load and y store are contiguous in memory instead of 9-node runs 1.2 kB apart."""
    c = np.asarray(xyz, float).reshape(-1, 3)
    lo = c.min(axis=0)
    h = []
    for d in range(3):                                     # lattice unit: smallest spacing between node planes
        u = np.unique(c[:, d])
        h.append(float(np.diff(u).min()) if len(u) > 1 else 1.0)
    ijk = np.floor((c - lo) / np.array(h) + 1e-6).astype(np.int64)
    if ijk.max() >= (1 << 20):
        return None
    blk = ijk // block
    loc = ijk - blk * block
    nb = blk.max(axis=0) + 1
    on_face = (loc == 0).any(axis=1)
    key = ((((blk[:, 2] * nb[1] + blk[:, 1]) * nb[0] + blk[:, 0]) * 2 + on_face) * block + loc[:, 2]) * block * block \
        + loc[:, 1] * block + loc[:, 0]
    interior = np.flatnonzero(~is_b)
    return np.concatenate([np.flatnonzero(is_b), interior[np.argsort(key[interior], kind="stable")]])


def from_refmeshpart(part, device=0, comm=None, rows_per_lane=0, n_threads=0, kind="sell", ebe_chunked=True):
    """Build the GPU operator of one RefMeshPart (see module docstring for the keys read).
    kind: "sell" = assembled SELL-BSR3 matrix (default), "dict" = the same matrix with its values replaced by a dictionary of
    its distinct 3x3 blocks (PCG_FORMAT_DICTIONARY; falls back to "sell" storage when there are too many),
    "ebe" = matrix-free element-by-element."""
    ndof = int(part["NDOF"])
    if ndof % 3:
        raise PcgError("NDOF must be a multiple of 3 (dof = 3*node + dir, partition_mesh.py:826)")
    n_nodes = ndof // 3
    groups = [g for g in part["SubDomainData"]["StrucDataList"] if g.get("ElemTypeId", 0) >= 0]   # :855-856
    nbr = list(part.get("NbrMPIdVector", []))
    ovl = [np.asarray(v, np.int64) for v in part.get("OvrlpLocalDofVecList", [])]
    node_perm = None
    dof_map = None
    n_bnd = 0
    xyz = part.get("NodeCoordVec")              # (3*NNode,) x,y,z per node (partition_mesh.py:357); optional
    blocked = kind == "ebe" and xyz is not None and os.environ.get("PCG_EBE_BLOCKED_ORDER", "1") == "1"
    # (Round 3 measured SELL-C-sigma here - rows sorted by length inside 2048-row windows, which takes the padding of the octree
    #  mesh from 56.6 % to 2.1 % - and removed it again: lanes of a slice then hold rows from anywhere in the window, the x gather
    #  loses its lane-to-lane locality and the SpMV gained 3 % for 35 % fewer bytes: profiles/r03_octree_ab_sessionG.log.  What the
    #  engine does instead: rows stay where they are, the part of a row beyond its slice's base width continues in an overflow
    #  matrix - csrc/sell.cpp split_overflow.)
    if len(nbr) or blocked:
        is_b = np.zeros(n_nodes, bool)
        for v in ovl:
            is_b[v // 3] = True
        order = _blocked_node_order(xyz, is_b) if blocked else None
        if order is None:
            order = np.concatenate([np.flatnonzero(is_b), np.flatnonzero(~is_b)])     # old ids, interface first
        node_perm = np.empty(n_nodes, np.int64)
        node_perm[order] = np.arange(n_nodes)
        n_bnd = int(is_b.sum())
        dof_map = (3 * node_perm[:, None] + np.arange(3)[None, :]).ravel()
    if kind == "ebe":
        op = Operator(n_nodes, None, None, None, n_bnd, dof_map, device, 0, ebe_groups=groups, node_perm=node_perm,
                      node_coords=xyz, ebe_chunked=ebe_chunked)
    elif kind in ("sell", "dict"):
        fmt = int(rows_per_lane) | (_lib.FORMAT_DICTIONARY if kind == "dict" else 0)
        # "dict": rows go from the assembler straight into the table format, the values are never materialised
        # (pcg_create_asm); PCG_ASM_STREAM=0 / =1 forces the array path / the assembler path for either kind
        if os.environ.get("PCG_ASM_STREAM", "1" if kind == "dict" else "0") == "1":
            op = Operator.from_assembler(groups, n_nodes, node_perm, n_bnd, dof_map, device, fmt, n_threads)
        else:
            rowptr, cols, vals = assemble_bsr3(groups, n_nodes, node_perm, n_threads)
            op = Operator(n_nodes, rowptr, cols, vals, n_bnd, dof_map, device, fmt)
    else:
        raise ValueError(kind)
    w = np.asarray(part["DofWeightVector"], float)
    if not np.all((w == 0) | (w == 1)):
        raise PcgError("DofWeightVector must be 0/1 (partition_mesh.py:870-887)")
    free = np.zeros(ndof, bool)
    free[np.asarray(part["LocDofEff"], np.int64)] = True
    op.set_masks(w == 1, free)
    op.set_halo(nbr, ovl)
    op.part_id = int(part.get("Id", 0))
    op.glob_n_eff = int(part["GlobData"]["GlobNDofEff"]) if "GlobData" in part else None
    if len(nbr) or comm is not None:
        if comm is None:
            raise PcgError("part has neighbours: pass comm=pcg_mi355x.dist.TorchComm(...)")
        op.set_comm(comm)
    return op
