"""TEST INFRASTRUCTURE - CPU restatement (NumPy) of the reference PCG hot path.

This is the parity oracle for the MI355X engine.  It is NOT product code: only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg import it, and only as the checker /
the timed CPU baseline.  The product path (pcg_mi355x) never imports it and has no CPU fallback.

Every function restates one reference function with the SAME NumPy expressions in the SAME
order, so in the build container it is bit-identical to the reference run through
oracle/ref_shim.py (pinned by oracle/make_golden.py, which asserts exact equality and writes
tests/golden/*.npz; tests/test_oracle_golden.py re-checks the oracle against those fixtures
wherever the tests run).  Parity status: PINNED against the reference's own functions executed
live (the reference ships no tests/golden vectors of its own).

Ranks are "virtual": N parts advance in lock-step inside one process; an allreduce is the sum of
the per-part partials in rank order; the interface exchange copies the neighbour's partial sums
directly.  (reference: one part per MPI rank, pcg_solver.py:91.)

Reference map (all in /root/reference/src/solver/pcg_solver.py):
  matvec_local          :255-300   EBE gather / sign / Ke @ (Ck*U) / sign / bincount
  halo_sum              :303-334   Isend/Recv partial sums on OvrlpLocalDofVecList, then +=
  calc_matvec           :242-336   (= calcMatVecProd; calcMPFint :339-342)
  update_bc             :226-238
  update_preconditioner :346-352
  pcg                   :356-598   every branch, flags 0-4, MATLAB-style Iter+1
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

EPS = np.finfo(float).eps            # pcg_solver.py:972

_HERE = os.path.dirname(os.path.abspath(__file__))
_clib = None


def _load_c():
    """Optional C kernel for the EBE mat-vec (oracle/ebe_matvec.c), for sizes where NumPy temporaries hurt."""
    global _clib
    if _clib is None:
        path = os.path.join(_HERE, "_build", "libebe_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle C kernel not built (run __graft_entry__.build() or make -C oracle)")
        lib = ctypes.CDLL(path)
        lib.ebe_matvec_group.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.ebe_matvec_group.restype = None
        lib.scatter_add_seq.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.scatter_add_seq.restype = None
        _clib = lib
    return _clib


def matvec_local(part, x, mode="Strain", use_c=False):
    """Local (un-exchanged) operator apply.  pcg_solver.py:255-300, FintCalcMode 'outbin'."""
    groups = part["SubDomainData"]["StrucDataList"]
    flat_val = np.zeros(part["NCountDof"], dtype=float)                 # :257
    I = 0
    for g in groups:                                                     # :265
        tbl = g["ElemList_LocDofVector"]
        if mode == "Strain":                                             # :271-280
            Ke = g["ElemStiffMat"]
            sign = g["ElemList_SignVector"]
            Ck = g["ElemList_Ck"]
            if use_c:
                lib = _load_c()
                nd, ne = tbl.shape
                out = np.empty((nd, ne))
                tbl64 = np.ascontiguousarray(tbl, dtype=np.int64)
                sg = np.ascontiguousarray(sign, dtype=np.uint8)
                xx = np.ascontiguousarray(x, dtype=float)
                lib.ebe_matvec_group(nd, ne, tbl64.ctypes.data, sg.ctypes.data, Ck.ctypes.data,
                                     np.ascontiguousarray(Ke).ctypes.data, xx.ctypes.data, out.ctypes.data)
                ref_vec = out
            else:
                U = x[tbl]                                               # :277
                U[sign] *= -1.0                                          # :278
                ref_vec = np.dot(Ke, Ck * U)                             # :279
                ref_vec[sign] *= -1.0                                    # :280
        elif mode == "Preconditioner":                                   # :282-287
            ref_vec = g["ElemList_Ck"] * g["ElemDiagStiffMat"][np.newaxis].T
        else:
            raise ValueError(mode)
        n = tbl.size
        flat_val[I:I + n] = ref_vec.ravel()                              # :295-297
        I += n
    if use_c:
        y = np.zeros(part["NDOF"])
        idx = np.ascontiguousarray(part["Flat_ElemLocDof"], dtype=np.int64)
        _load_c().scatter_add_seq(len(idx), idx.ctypes.data, flat_val.ctypes.data, y.ctypes.data)
        return y
    return np.bincount(part["Flat_ElemLocDof"], weights=flat_val, minlength=part["NDOF"])   # :300


def halo_sum(parts, ys):
    """Interface sum-exchange.  pcg_solver.py:303-334: each part sends its partial sums on the
    overlap DOFs to every neighbour, then adds what it received in neighbour order."""
    by_id = {p["Id"]: k for k, p in enumerate(parts)}
    sent = []
    for p, y in zip(parts, ys):                                          # :307-309 (pack before any +=)
        sent.append([y[idx] for idx in p["OvrlpLocalDofVecList"]])
    for k, (p, y) in enumerate(zip(parts, ys)):
        for j, nbr in enumerate(p["NbrMPIdVector"]):                     # :333-334
            q = parts[by_id[nbr]]
            jq = q["NbrMPIdVector"].index(p["Id"])
            y[p["OvrlpLocalDofVecList"][j]] += sent[by_id[nbr]][jq]
    return ys


def calc_matvec(parts, xs, mode="Strain", use_c=False):
    """calcMatVecProd on all parts (pcg_solver.py:242-336)."""
    ys = [matvec_local(p, x, mode, use_c) for p, x in zip(parts, xs if xs is not None else [None] * len(parts))]
    return halo_sum(parts, ys)


def update_bc(parts, use_c=False):
    """pcg_solver.py:226-238."""
    udis = []
    for p in parts:
        gd = p["GlobData"]
        udis.append(p["Ud"] * gd["TimeStepDelta"][gd["TimeStepCount"]])            # :234
    fdis = calc_matvec(parts, udis, "Strain", use_c)                                # :235
    for p, udi, fdi in zip(parts, udis, fdis):
        gd = p["GlobData"]
        p["Fext"] = p["RefLoadVector"] * gd["TimeStepDelta"][gd["TimeStepCount"]] - fdi   # :236-237
        p["Udi"] = udi                                                              # :238


def update_preconditioner(parts):
    """pcg_solver.py:346-352."""
    diags = calc_matvec(parts, None, "Preconditioner")
    for p, d in zip(parts, diags):
        p["InvDiagPreCondVector0"] = (1.0 / d)[p["LocDofEff"]]


def _allreduce(vals):
    """MPI_SUM (pcg_solver.py:622-628): sum of per-rank partials, rank order."""
    tot = vals[0]
    for v in vals[1:]:
        tot = tot + v
    return tot


class TooSmallTolerance(Warning):
    """The reference does `raise Warning('PCG : TooSmallTolerance')` (pcg_solver.py:549)."""


def pcg(parts, use_c=False, record=True, observer=None):
    """PCG(RefMeshPart) on all parts in lock-step.  pcg_solver.py:356-598.

    observer (tests only, no effect on the arithmetic): called once per completed iteration with the vectors of that
    iteration (copies of R and X from before the updates, P, Q, the updated R and X, rho, beta, pq, alpha and the three
    sums of :504-507), so a lock-step test can feed the SAME inputs to the engine's kernels iteration by iteration.

    Mutates the part dicts like the reference (`Un`, rank-0 GlobData['TimeList_*']).  Returns a
    dict with flag/relres/iter (as stored), history (rows [NormP, NormX, NormR] per iteration,
    as logged from the 3-vector allreduce :507), n_matvec, and `early` (the tuple the reference
    returns on the two early exits, else None).
    """
    P = len(parts)
    gd0 = parts[0]["GlobData"]
    n_eff_glob = gd0["GlobNDofEff"]
    max_iter = gd0["MaxIter"]
    tol = gd0["Tol"]
    step = gd0["TimeStepCount"]
    eff = [p["LocDofEff"] for p in parts]
    W = [p["DofWeightVector_Eff"] for p in parts]
    Minv = [p["InvDiagPreCondVector0"] for p in parts]
    n_matvec = 0

    def A(x_unq):
        nonlocal n_matvec
        n_matvec += 1
        return calc_matvec(parts, x_unq, "Strain", use_c)

    Fext = [p["Fext"][e] for p, e in zip(parts, eff)]                   # :377
    X = [p["Un"][e] for p, e in zip(parts, eff)]                        # :378-379 (fancy index = copy)
    XMin = X                                                            # :380 (alias)
    n2b = np.sqrt(_allreduce([np.dot(f, f * w) for f, w in zip(Fext, W)]))   # :381-383
    tolb = tol * n2b                                                    # :384
    hist = []

    if n2b == 0:                                                        # :387-395
        out = []
        for p, e, x in zip(parts, eff, X):
            xu = np.zeros(p["NDOF"])
            xu[e] = x
            out.append((xu, 0, 0, 0))
        return {"flag": 0, "relres": 0, "iter": 0, "history": np.zeros((0, 3)), "n_matvec": 0, "early": out}

    flag = 1                                                            # :399-406
    rho = 1.0
    stag = 0
    more = 0
    max_stag = 3
    max_msteps = min([int(n_eff_glob / 50), 5, n_eff_glob - max_iter])
    i_min = 0
    it = 0
    X_unq = [np.zeros(p["NDOF"]) for p in parts]                        # :408-409
    P_unq = [np.zeros(p["NDOF"]) for p in parts]
    for xu, e, x in zip(X_unq, eff, X):
        xu[e] = x                                                       # :411
    ax = A(X_unq)                                                       # :412
    R = [f - a[e] for f, a, e in zip(Fext, ax, eff)]                    # :413-414
    normr = np.sqrt(_allreduce([np.dot(r, r * w) for r, w in zip(R, W)]))    # :415-416
    normr_min = normr
    normr_act = normr

    if normr <= tolb:                                                   # :421-426
        out = [(xu, 0, normr / n2b, 0) for xu in X_unq]
        return {"flag": 0, "relres": normr / n2b, "iter": 0, "history": np.zeros((0, 3)),
                "n_matvec": n_matvec, "early": out}

    Pv = [None] * P
    i = -1
    for i in range(max_iter):                                           # :438
        Y = [m * r for m, r in zip(Minv, R)]                            # :447
        if any(np.any(np.isinf(y)) for y in Y):                         # :448-450 (per-rank break; same for all here)
            flag = 2
            break
        Z = Y                                                           # :458
        rho_1 = rho                                                     # :461
        rho = _allreduce([np.dot(z, r * w) for z, r, w in zip(Z, R, W)])    # :462-463
        if rho == 0 or np.isinf(rho):                                   # :467-469
            flag = 4
            break
        if i == 0:                                                      # :472-479
            Pv = Z
        else:
            beta = rho / rho_1
            if beta == 0 or np.isinf(beta):
                flag = 4
                break
            Pv = [z + beta * pv for z, pv in zip(Z, Pv)]
        for pu, e, pv in zip(P_unq, eff, Pv):                           # :482
            pu[e] = pv
        q_unq = A(P_unq)                                                # :483
        Q = [q[e] for q, e in zip(q_unq, eff)]                          # :484
        pq = _allreduce([np.dot(pv, q * w) for pv, q, w in zip(Pv, Q, W)])   # :487-488
        if pq <= 0 or np.isinf(pq):                                     # :492-498
            flag = 4
            break
        alpha = rho / pq
        if np.isinf(alpha):
            flag = 4
            break
        if observer is not None:
            r_before, x_before = [r.copy() for r in R], [x.copy() for x in X]
        for r, q in zip(R, Q):                                          # :501
            r -= alpha * q
        sq = _allreduce([np.array([np.dot(pv, pv * w), np.dot(x, x * w), np.dot(r, r * w)])
                         for pv, x, r, w in zip(Pv, X, R, W)])           # :504-507
        normp, normx, normr = np.sqrt(sq)
        if record:
            hist.append([normp, normx, normr])
        if normp * abs(alpha) < EPS * normx:                            # :512-513
            stag += 1
        else:
            stag = 0
        for x, pv in zip(X, Pv):                                        # :516 (in place: XMin aliases X until :557)
            x += alpha * pv
        if observer is not None:
            observer(dict(i=i, beta=(None if i == 0 else beta), rho=rho, pq=pq, alpha=alpha, sq=np.array(sq), P=Pv, Q=Q,
                          R_before=r_before, X_before=x_before, R_after=R, X_after=X, Minv=Minv, W=W, eff=eff))
        normr_act = normr                                               # :518
        if normr <= tolb or stag >= max_stag or more > 0:               # :527
            for xu, e, x in zip(X_unq, eff, X):                         # :528
                xu[e] = x
            ax = A(X_unq)                                               # :529
            R = [f - a[e] for f, a, e in zip(Fext, ax, eff)]            # :530-531
            normr_act = np.sqrt(_allreduce([np.dot(r, r * w) for r, w in zip(R, W)]))   # :532-533
            if normr_act <= tolb:                                       # :540-543
                flag = 0
                it = i
                break
            else:
                if stag >= max_stag and more == 0:                      # :545
                    stag = 0
                more += 1                                               # :546
                if more >= max_msteps:                                  # :548-549
                    raise TooSmallTolerance("PCG : TooSmallTolerance")
        if normr_act < normr_min:                                       # :555-558
            normr_min = normr_act
            XMin = [np.array(x) for x in X]
            i_min = i
        if stag >= max_stag:                                            # :560-562
            flag = 3
            break

    if flag == 0:                                                       # :566-567
        relres = normr_act / n2b
    else:                                                               # :568-582
        for xu, e, xm in zip(X_unq, eff, XMin):
            xu[e] = xm
        ax = A(X_unq)
        R = [f - a[e] for f, a, e in zip(Fext, ax, eff)]
        normr = np.sqrt(_allreduce([np.dot(r, r * w) for r, w in zip(R, W)]))
        if normr < normr_act:
            X = XMin
            it = i_min
            relres = normr / n2b
        else:
            it = i
            relres = normr_act / n2b
    it += 1                                                             # :584
    gd0["TimeList_Flag"][step] = flag                                   # :593-596 (rank 0)
    gd0["TimeList_RelRes"][step] = relres
    gd0["TimeList_Iter"][step] = it
    for p, xu in zip(parts, X_unq):                                     # :598
        p["Un"] = xu + p["Udi"]
    return {"flag": flag, "relres": relres, "iter": it, "history": np.array(hist).reshape(-1, 3),
            "n_matvec": n_matvec, "early": None}


def solve_step(parts, use_c=False):
    """One load step of the reference's loop (pcg_solver.py:1004-1006)."""
    update_bc(parts, use_c)
    update_preconditioner(parts)
    return pcg(parts, use_c)
