"""Drop-in replacements for the hot-path functions of the reference's src/solver/pcg_solver.py.

Same names' meaning, same RefMeshPart keys read and written, same error behaviour:

    reference (pcg_solver.py)            here
    -----------------------------------  -----------------------------------------
    PCG(RefMeshPart)            :356     solve(RefMeshPart)        (alias PCG)
    updateBC(RefMeshPart)       :226     update_bc(RefMeshPart)    (alias updateBC)
    updatePreconditioner(..)    :346     update_preconditioner(..) (alias updatePreconditioner)
    calcMPFint(x, RefMeshPart)  :339     calc_mpfint(x, RefMeshPart)
    calcMatVecProd(..)          :242     calc_matvec_prod(..)

so a maintainer can write `ref.PCG = pcg_mi355x.solve` and keep the load-step
loop (:1002-1008), the export and the timing code.  The reference's module globals `Comm`/`Rank`
become the `comm` given to `configure()` (a pcg_mi355x.dist.TorchComm; None = one part).

All arithmetic happens in the HIP engine behind the C ABI; this module only moves NumPy arrays in
and out and mirrors the key contract.  There is no CPU fallback.
"""
from __future__ import annotations

import time

import numpy as np

from . import _lib
from .operator import Operator, from_refmeshpart

__all__ = ["configure", "get_operator", "solve", "PCG", "update_bc", "updateBC", "update_preconditioner",
           "updatePreconditioner", "calc_matvec_prod", "calc_mpfint", "solve_system", "SolveInfo"]

_OP_KEY = "_pcg_mi355x_operator"
_cfg = {"comm": None, "device": 0, "rows_per_lane": 0, "operator": "sell", "ebe_chunked": True}


def configure(comm=None, device=0, rows_per_lane=0, operator="sell", ebe_chunked=True):
    """Set the process-wide communicator / device (the reference's module globals Comm, Rank :968-970)
    and the operator kind ("sell" = assembled matrix, "dict" = assembled with a dictionary of its distinct 3x3 blocks,
    "ebe" = matrix-free like the reference)."""
    _cfg.update(comm=comm, device=device, rows_per_lane=rows_per_lane, operator=operator, ebe_chunked=ebe_chunked)


def get_operator(RefMeshPart) -> Operator:
    """The part's GPU operator, built on first use and cached on the dict (assembly is set-up work;
    the reference rebuilds nothing between load steps either, its element tables are static)."""
    op = RefMeshPart.get(_OP_KEY)
    if op is None:
        op = from_refmeshpart(RefMeshPart, device=_cfg["device"], comm=_cfg["comm"],
                              rows_per_lane=_cfg["rows_per_lane"], kind=_cfg["operator"], ebe_chunked=_cfg["ebe_chunked"])
        RefMeshPart[_OP_KEY] = op
    return op


def _rank():
    c = _cfg["comm"]
    return 0 if c is None else c.rank


def _account(GlobData, t_total, t_comm):
    """Two-bucket timers of the reference (updateTime :631-641): everything that is not waiting in
    a communication call is 'calculation'."""
    rec = GlobData.get("MP_TimeRecData") if isinstance(GlobData, dict) else None
    if rec is not None:
        rec["dT_Calc"] += max(0.0, t_total - t_comm)
        rec["dT_CommWait"] += t_comm
        rec["t0"] = time.time()


class _Timed:
    """updateTime around a set-up call (updateBC / updatePreconditioner contain one mat-vec + exchange each): wall time
    goes to 'calculation' except what the native communicator measured as blocked in communication."""

    def __init__(self, GlobData):
        self.gd = GlobData

    def __enter__(self):
        c = _cfg["comm"]
        self.c0 = c.stats() if getattr(c, "native", False) else None
        self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        t = time.perf_counter() - self.t0
        t_comm = 0.0
        if self.c0 is not None:
            c1 = _cfg["comm"].stats()
            t_comm = 1e-3 * ((c1["halo_wait_ms"] - self.c0["halo_wait_ms"]) + (c1["allreduce_ms"] - self.c0["allreduce_ms"]))
        _account(self.gd, t, min(t, t_comm))
        return False


def calc_matvec_prod(RefMeshPart, ComputeReference="Strain", MP_Xn=None):
    """calcMatVecProd (:242-336): interface-summed A.x ('Strain') or diag(A) ('Preconditioner')."""
    op = get_operator(RefMeshPart)
    if ComputeReference == "Strain":
        return op.apply(MP_Xn)
    if ComputeReference == "Preconditioner":
        return op.diag()
    raise ValueError(ComputeReference)


def calc_mpfint(MP_Un, RefMeshPart):
    """calcMPFint (:339-342)."""
    return calc_matvec_prod(RefMeshPart, "Strain", MP_Un)


def update_bc(RefMeshPart):
    """updateBC (:226-238): Udi = Ud*delta ; Fext = F*delta - A.Udi."""
    gd = RefMeshPart["GlobData"]
    delta = gd["TimeStepDelta"][gd["TimeStepCount"]]
    op = get_operator(RefMeshPart)
    with _Timed(gd):
        fext, udi = op.update_bc(RefMeshPart["RefLoadVector"], RefMeshPart["Ud"], delta)
    RefMeshPart["Fext"] = fext
    RefMeshPart["Udi"] = udi


def update_preconditioner(RefMeshPart):
    """updatePreconditioner (:346-352): InvDiagPreCondVector0 = (1/diag(A))[LocDofEff]."""
    op = get_operator(RefMeshPart)
    with _Timed(RefMeshPart.get("GlobData")):
        inv = op.build_jacobi()
    RefMeshPart["InvDiagPreCondVector0"] = inv[np.asarray(RefMeshPart["LocDofEff"], np.int64)]


class SolveInfo:
    def __init__(self, res, hist=None):
        self.flag = int(res.flag)
        self.status = int(res.status)
        self.iter = int(res.iter)
        self.iters_done = int(res.iters_done)
        self.n_matvec = int(res.n_matvec)
        self.relres = float(res.relres)
        self.norm_b = float(res.norm_b)
        self.t_total_s = float(res.t_total_s)
        self.t_comm_s = float(res.t_comm_s)
        self.spmv_ms_sum = float(res.spmv_ms_sum)
        self.spmv_count = int(res.spmv_count)
        self.iters_enqueued = int(res.iters_enqueued)
        self.history = hist

    def __repr__(self):
        return (f"SolveInfo(flag={self.flag}, iter={self.iter}, relres={self.relres:.3e}, "
                f"n_matvec={self.n_matvec}, status={self.status})")


def solve_system(op: Operator, b, x0=None, inv_diag=None, tol=1e-7, max_iter=10000, glob_n_eff=None,
                 history=False):
    """Functional core: Jacobi-PCG on the operator's free dofs.  Vectors have the part's full local
    length (fixed dofs are ignored / returned as 0).  -> (x, SolveInfo)."""
    x, res, hist = op.solve(b, x0, inv_diag, tol, max_iter, glob_n_eff, history)
    return x, SolveInfo(res, hist)


def solve(RefMeshPart, history=False):
    """PCG(RefMeshPart) (:356-598).

    Reads Un, Fext, DofWeightVector_Eff (through the operator's ownership mask), NDOF, LocDofEff,
    InvDiagPreCondVector0 and GlobData{GlobNDofEff, MaxIter, Tol, TimeStepCount}; writes
    RefMeshPart['Un'] = X_Unq + Udi (:598) and, on rank 0, GlobData['TimeList_Flag|RelRes|Iter']
    (:593-596).  Like the reference it returns None, except for the two early exits which return
    the tuple (MP_X_Unq, Flag, RelRes, Iter) WITHOUT touching RefMeshPart (:387-395, :421-426), and
    it raises Warning('PCG : TooSmallTolerance') where the reference does (:549).
    """
    op = get_operator(RefMeshPart)
    gd = RefMeshPart["GlobData"]
    eff = np.asarray(RefMeshPart["LocDofEff"], np.int64)
    n = int(RefMeshPart["NDOF"])
    inv = np.zeros(n)
    inv[eff] = RefMeshPart["InvDiagPreCondVector0"]
    max_iter = int(gd["MaxIter"])
    x, res, hist = op.solve(RefMeshPart["Fext"], RefMeshPart["Un"], inv, float(gd["Tol"]), max_iter,
                            int(gd["GlobNDofEff"]), history)
    _account(gd, res.t_total_s, res.t_comm_s)
    info = SolveInfo(res, hist)
    RefMeshPart["_pcg_mi355x_info"] = info
    if res.status == _lib.STATUS_ZERO_RHS:
        return x, 0, 0, 0
    if res.status == _lib.STATUS_GOOD_X0:
        return x, 0, info.relres, 0
    if res.status == _lib.STATUS_TOO_SMALL_TOL:
        raise Warning("PCG : TooSmallTolerance")
    if _rank() == 0:
        step = gd["TimeStepCount"]
        gd["TimeList_Flag"][step] = info.flag
        gd["TimeList_RelRes"][step] = info.relres
        gd["TimeList_Iter"][step] = info.iter
    RefMeshPart["Un"] = x + RefMeshPart["Udi"]
    return None


# the reference's names
PCG = solve
updateBC = update_bc
updatePreconditioner = update_preconditioner
calcMatVecProd = calc_matvec_prod
calcMPFint = calc_mpfint
