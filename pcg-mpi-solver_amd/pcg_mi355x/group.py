"""One process, several GPUs: the device group (C ABI pcg_group_*, csrc/group.cpp).

The reference runs one MPI rank per part (`mpiexec -np N python pcg_solver.py`, src/solver/pcg_solver.py:91) and this
package's default launch mirrors it - one process per GPU (`pcg_mi355x.run`, `bench.py`).  A `DeviceGroup` is the
alternative for a host program that exists once (a notebook, an embedding application): member k is part k on device
`devices[k]`, and every method below makes the per-part call of the same name for ALL parts at once.  Inside the library
one persistent host thread per member drives that member's engine, so the interface exchange (grouped ncclSend/ncclRecv
on the member's communication stream) and the all-reduces proceed side by side exactly as they do between processes:
same kernels, same native communicator, same decisions on every member - the results are bit-identical to the N-process run.

`GroupSolver` puts the reference's names on it: `updateBC()`, `updatePreconditioner()`, `PCG()` do for every RefMeshPart of
the list what the reference's functions (:226-238, :346-352, :356-598) do on every rank, with the same keys read and written.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

from . import _lib
from ._lib import check, PcgError
from .operator import from_refmeshpart
from .solver import SolveInfo, _account

__all__ = ["DeviceGroup", "GroupSolver"]


def _ptrs(arrs):
    """ctypes array of host pointers, one per member (None -> NULL)."""
    return (C.c_void_p * len(arrs))(*[None if a is None else a.ctypes.data for a in arrs])


class _MemberComm:
    """What Operator.set_comm() needs of a native communicator; the pcg_comm itself is owned by the group."""
    native = True
    backend = "rccl-native (device group)"
    group = None

    def __init__(self, grp, k):
        self._g, self.rank, self.world = grp, k, grp.size
        self.device = grp.devices[k]

    @property
    def handle(self):
        return C.c_void_p(_lib.lib().pcg_group_comm(self._g._h, self.rank))

    def set_timing(self, on=True):
        check(_lib.lib().pcg_comm_set_timing(self.handle, 1 if on else 0), "pcg_comm_set_timing")

    def stats(self):
        st = _lib.CommStats()
        check(_lib.lib().pcg_comm_get_stats(self.handle, C.byref(st)), "pcg_comm_get_stats")
        return {k: getattr(st, k) for k, _ in _lib.CommStats._fields_}

    def reraise(self):
        pass

    def release_stream(self, stream_ptr):
        pass


def default_devices(n_parts):
    """Member k -> device k.  More parts than visible GPUs is an error (RCCL refuses two ranks of one communicator on one
    device) unless the tests' shared-GPU stand-in for librccl (PCG_RCCL_LIB) or the CPU test double is in use."""
    have = _lib.lib().pcg_device_count()
    if have > 0 and n_parts > have and not os.environ.get("PCG_RCCL_LIB"):
        raise PcgError(f"{n_parts} parts but only {have} GPU(s) visible: a device group needs one GPU per part "
                       f"(use torchrun with one rank per GPU on several nodes, or fewer parts)")
    have = max(1, have)
    return [k % have for k in range(n_parts)]


class DeviceGroup:
    def __init__(self, devices):
        self.devices = [int(d) for d in devices]
        self.size = len(self.devices)
        dev = (C.c_int32 * self.size)(*self.devices)
        h = C.c_void_p()
        check(_lib.lib().pcg_group_create(self.size, dev, C.byref(h)), "pcg_group_create")
        self._h = h
        self._L = _lib.lib()
        self.ops = [None] * self.size
        self.comms = [_MemberComm(self, k) for k in range(self.size)]

    # -- membership -----------------------------------------------------------------------------------
    def attach(self, k, op):
        """Member k's operator (created on devices[k]); the member's communicator is attached to it."""
        check(self._L.pcg_group_attach(self._h, int(k), op._h), "pcg_group_attach")
        op._comm = self.comms[k]
        self.ops[k] = op

    @classmethod
    def from_refmeshparts(cls, parts, devices=None, kind="sell", rows_per_lane=0, ebe_chunked=True):
        """Group over a COMPLETE list of RefMeshPart dicts (part Id k -> member k, :91); operators built one after the other
        (host assembly is multi-threaded itself), each on its member's device."""
        n = len(parts)
        if devices is None:
            devices = default_devices(n)
        if sorted(int(p.get("Id", k)) for k, p in enumerate(parts)) != list(range(n)):
            raise ValueError("parts must be the complete list with Id 0..N-1 (one part per member, pcg_solver.py:91)")
        parts = sorted(parts, key=lambda p: int(p.get("Id", 0)))
        g = cls(devices)
        try:
            for k, P in enumerate(parts):
                op = from_refmeshpart(P, device=g.devices[k], comm=g.comms[k], rows_per_lane=rows_per_lane, kind=kind,
                                      ebe_chunked=ebe_chunked)
                check(g._L.pcg_group_attach(g._h, k, op._h), "pcg_group_attach")
                g.ops[k] = op
        except BaseException:
            g.close()
            raise
        return g

    def _need_ops(self):
        if any(o is None for o in self.ops):
            raise _lib.PcgError("device group: every member needs an operator (attach)")

    # -- collective calls: one list entry per member ------------------------------------------------------------------
    def apply(self, xs):
        """calcMatVecProd(.., 'Strain', x) on every part (:242-336)."""
        self._need_ops()
        xe = [op.to_engine(x) for op, x in zip(self.ops, xs)]
        ys = [np.empty(op.n) for op in self.ops]
        check(self._L.pcg_group_apply(self._h, _ptrs(xe), _ptrs(ys)), "pcg_group_apply")
        return [op.from_engine(y) for op, y in zip(self.ops, ys)]

    def diag(self):
        self._need_ops()
        ds = [np.empty(op.n) for op in self.ops]
        check(self._L.pcg_group_diag(self._h, _ptrs(ds)), "pcg_group_diag")
        return [op.from_engine(d) for op, d in zip(self.ops, ds)]

    def build_jacobi(self):
        """updatePreconditioner on every part (:346-352) -> 1/diag on the free dofs (0 on fixed), full local length."""
        self._need_ops()
        ds = [np.empty(op.n) for op in self.ops]
        check(self._L.pcg_group_build_jacobi(self._h, _ptrs(ds)), "pcg_group_build_jacobi")
        return [op.from_engine(d) for op, d in zip(self.ops, ds)]

    def update_bc(self, ref_loads, uds, delta):
        """updateBC on every part (:226-238) -> ([Fext], [Udi])."""
        self._need_ops()
        f = [op.to_engine(v) for op, v in zip(self.ops, ref_loads)]
        u = [op.to_engine(v) for op, v in zip(self.ops, uds)]
        fo = [np.empty(op.n) for op in self.ops]
        uo = [np.empty(op.n) for op in self.ops]
        check(self._L.pcg_group_update_bc(self._h, _ptrs(f), _ptrs(u), float(delta), _ptrs(fo), _ptrs(uo)), "pcg_group_update_bc")
        return [op.from_engine(v) for op, v in zip(self.ops, fo)], [op.from_engine(v) for op, v in zip(self.ops, uo)]

    def dot_w(self, a_list, b_list):
        self._need_ops()
        a = [op.to_engine(v) for op, v in zip(self.ops, a_list)]
        b = [op.to_engine(v) for op, v in zip(self.ops, b_list)]
        out = C.c_double()
        check(self._L.pcg_group_dot_w(self._h, _ptrs(a), _ptrs(b), C.byref(out)), "pcg_group_dot_w")
        return out.value

    def solve(self, bs, x0s=None, inv_diags=None, tol=1e-7, max_iter=10000, glob_n_eff=None, history=False):
        """PCG on every part at once -> ([x], [Result], [hist] or None); the Results agree on flag / iter / relres."""
        self._need_ops()
        n = self.size
        be = [op.to_engine(v) for op, v in zip(self.ops, bs)]
        x0e = None if x0s is None else [None if v is None else op.to_engine(v) for op, v in zip(self.ops, x0s)]
        mde = None if inv_diags is None else [None if v is None else op.to_engine(v) for op, v in zip(self.ops, inv_diags)]
        gne = int(glob_n_eff if glob_n_eff is not None else (self.ops[0].glob_n_eff or sum(op.n for op in self.ops)))
        xs = [np.empty(op.n) for op in self.ops]
        hists = [np.zeros((int(max_iter), 3)) for _ in range(n)] if history else None
        res = (_lib.Result * n)()
        rc = self._L.pcg_group_solve(self._h, _ptrs(be), None if x0e is None else _ptrs(x0e), None if mde is None else _ptrs(mde),
                                     float(tol), int(max_iter), gne, _ptrs(xs), None if hists is None else _ptrs(hists),
                                     int(max_iter) if history else 0, res)
        check(rc, "pcg_group_solve")
        results = [res[k] for k in range(n)]
        for op, r in zip(self.ops, results):
            op.last_result = r
        if hists is not None:
            hists = [h[:max(0, min(int(r.iters_done), int(max_iter)))] for h, r in zip(hists, results)]
        return [op.from_engine(x) for op, x in zip(self.ops, xs)], results, hists

    def set_timing(self, on=True):
        check(self._L.pcg_group_set_timing(self._h, 1 if on else 0), "pcg_group_set_timing")

    def enable_mailbox(self, on=True):
        """pcg_comm_enable_mailbox on every member (one process: the members see each other's mailboxes through plain peer
        pointers) -> True when all of them switched."""
        got = C.c_int32(0)
        check(self._L.pcg_group_enable_mailbox(self._h, 1 if on else 0, C.byref(got)), "pcg_group_enable_mailbox")
        return bool(got.value)

    def enable_direct_exchange(self, on=True):
        """pcg_enable_direct_exchange on every member's engine (the members map each other's receive buffers through plain peer
        pointers; members on ONE device decline, like the mailboxes) -> True when all of them switched."""
        got = C.c_int32(0)
        check(self._L.pcg_group_enable_direct_exchange(self._h, 1 if on else 0, C.byref(got)), "pcg_group_enable_direct_exchange")
        return bool(got.value)

    def close(self):
        """Engines first, then the group (its communicators must outlive the engines they are attached to)."""
        for k, op in enumerate(self.ops):
            if op is not None:
                op.close()
                self.ops[k] = None
        if getattr(self, "_h", None):
            self._L.pcg_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GroupSolver:
    """The reference's load-step calls (pcg_solver.py:1002-1008) for ALL parts of a model in one process:

        gs = GroupSolver(parts)                       # parts = every RefMeshPart of the model, Id 0..N-1
        for step in ...:                              # for TimeStepCount in range(1, RefMaxTimeStepCount)
            gs.updateBC(); gs.updatePreconditioner(); gs.PCG()

    Each part's dict is read and written exactly as the reference's functions do on that part's rank."""

    def __init__(self, parts, devices=None, operator="sell", rows_per_lane=0, ebe_chunked=True, timing=False):
        self.parts = sorted(parts, key=lambda p: int(p.get("Id", 0)))
        self.group = DeviceGroup.from_refmeshparts(self.parts, devices, operator, rows_per_lane, ebe_chunked)
        if timing:
            self.group.set_timing(True)

    def _comm_ms(self):
        return [c["halo_wait_ms"] + c["allreduce_ms"] for c in (m.stats() for m in self.group.comms)]

    def update_bc(self):
        """updateBC (:226-238) on every part: Udi = Ud*delta ; Fext = F*delta - A.Udi."""
        P0 = self.parts[0]["GlobData"]
        delta = P0["TimeStepDelta"][P0["TimeStepCount"]]
        c0, t0 = self._comm_ms(), time.perf_counter()
        fext, udi = self.group.update_bc([P["RefLoadVector"] for P in self.parts], [P["Ud"] for P in self.parts], delta)
        t = time.perf_counter() - t0
        for P, f, u, a, b in zip(self.parts, fext, udi, c0, self._comm_ms()):
            P["Fext"], P["Udi"] = f, u
            _account(P["GlobData"], t, min(t, (b - a) * 1e-3))

    def update_preconditioner(self):
        """updatePreconditioner (:346-352) on every part."""
        c0, t0 = self._comm_ms(), time.perf_counter()
        inv = self.group.build_jacobi()
        t = time.perf_counter() - t0
        for P, d, a, b in zip(self.parts, inv, c0, self._comm_ms()):
            P["InvDiagPreCondVector0"] = d[np.asarray(P["LocDofEff"], np.int64)]
            _account(P.get("GlobData"), t, min(t, (b - a) * 1e-3))

    def solve(self, history=False):
        """PCG(RefMeshPart) (:356-598) on every part.  Returns None like the reference; on the two early exits (:387-395,
        :421-426) the list of per-part tuples (MP_X_Unq, Flag, RelRes, Iter) and no dict is touched; raises
        Warning('PCG : TooSmallTolerance') where the reference does (:549)."""
        parts = self.parts
        gd = parts[0]["GlobData"]
        invs = []
        for P in parts:
            inv = np.zeros(int(P["NDOF"]))
            inv[np.asarray(P["LocDofEff"], np.int64)] = P["InvDiagPreCondVector0"]
            invs.append(inv)
        xs, results, hists = self.group.solve([P["Fext"] for P in parts], [P["Un"] for P in parts], invs, float(gd["Tol"]),
                                              int(gd["MaxIter"]), int(gd["GlobNDofEff"]), history)
        infos = [SolveInfo(r, None if hists is None else hists[k]) for k, r in enumerate(results)]
        for P, r, info in zip(parts, results, infos):
            _account(P["GlobData"], r.t_total_s, r.t_comm_s)
            P["_pcg_mi355x_info"] = info
        st = int(results[0].status)
        if st == _lib.STATUS_ZERO_RHS:
            return [(x, 0, 0, 0) for x in xs]
        if st == _lib.STATUS_GOOD_X0:
            return [(x, 0, infos[0].relres, 0) for x in xs]
        if st == _lib.STATUS_TOO_SMALL_TOL:
            raise Warning("PCG : TooSmallTolerance")
        for k, (P, x, info) in enumerate(zip(parts, xs, infos)):
            g = P["GlobData"]
            if k == 0:                                  # only rank 0 records the outcome (:593-596)
                step = g["TimeStepCount"]
                g["TimeList_Flag"][step] = info.flag
                g["TimeList_RelRes"][step] = info.relres
                g["TimeList_Iter"][step] = info.iter
            P["Un"] = x + P["Udi"]
        return None

    # the reference's names
    updateBC = update_bc
    updatePreconditioner = update_preconditioner
    PCG = solve

    def close(self):
        self.group.close()
