"""TEST INFRASTRUCTURE (container-only): run the *unmodified* reference solver functions.

/root/reference has no tests and needs mpi4py (not installed).  This shim puts a minimal fake
`mpi4py.MPI` into sys.modules, imports `/root/reference/src/solver/pcg_solver.py` untouched and
drives `updateBC -> updatePreconditioner -> PCG` (reference loop: pcg_solver.py:1002-1008) on
hand-built RefMeshPart dicts.  N>1 ranks are "virtual ranks": one thread per part; the module
globals `Comm`/`Rank` that the reference reads (pcg_solver.py:321,326,593,625) become
thread-local proxies.  Isend/Recv move copies through per-(src,dst,tag) mailboxes; allreduce is
a barrier plus a sum in rank order.

Used only by oracle/make_golden.py (and optional local cross-checks) to PIN oracle/pcg_oracle.py
and to generate tests/golden/*.npz.  /root/reference does not exist on the GPU box, so nothing
that runs there imports this file.
"""
from __future__ import annotations

import os
import queue
import sys
import threading
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"

_tls = threading.local()


class _RankProxy:
    """Stands in for the module-global integer `Rank`."""

    def _v(self):
        return getattr(_tls, "rank", 0)

    def __eq__(self, o):
        return self._v() == o

    def __ne__(self, o):
        return self._v() != o

    def __hash__(self):
        return hash(self._v())

    def __int__(self):
        return self._v()

    def __index__(self):
        return self._v()

    def __mul__(self, o):
        return self._v() * o

    __rmul__ = __mul__

    def __repr__(self):
        return f"Rank({self._v()})"


class _Request:
    pass


class _World:
    def __init__(self):
        self.configure(1)

    def configure(self, n):
        self.n = n
        self.boxes = {}
        self.lock = threading.Lock()
        self.barrier_obj = threading.Barrier(n)
        self.slots = [None] * n
        self.n_allreduce = 0
        self.n_p2p = 0

    def _box(self, key):
        with self.lock:
            if key not in self.boxes:
                self.boxes[key] = queue.Queue()
            return self.boxes[key]

    # -- the subset of mpi4py.MPI.Comm the hot path uses --------------------------------------
    def Get_rank(self):
        return getattr(_tls, "rank", 0)

    def Get_size(self):
        return self.n

    def barrier(self):
        self.barrier_obj.wait()

    def Isend(self, buf, dest, tag=0):                  # pcg_solver.py:321
        me = self.Get_rank()
        self._box((me, int(dest), int(tag))).put(np.array(buf, copy=True))
        if me == 0:
            self.n_p2p += 1
        return _Request()

    def Recv(self, buf, source, tag=0):                 # pcg_solver.py:326
        me = self.Get_rank()
        data = self._box((int(source), me, int(tag))).get(timeout=600)
        buf[...] = data

    def allreduce(self, value, op=None):                # pcg_solver.py:625
        me = self.Get_rank()
        self.slots[me] = value
        self.barrier_obj.wait()
        total = self.slots[0]
        for r in range(1, self.n):                      # rank order, deterministic
            total = total + self.slots[r]
        if me == 0:
            self.n_allreduce += 1
        self.barrier_obj.wait()
        return total

    def gather(self, value, root=0):
        me = self.Get_rank()
        self.slots[me] = value
        self.barrier_obj.wait()
        out = list(self.slots) if me == root else None
        self.barrier_obj.wait()
        return out


WORLD = _World()
_ref = None


def load_reference():
    """Import the reference solver module with the fake MPI in place (idempotent)."""
    global _ref
    if _ref is not None:
        return _ref
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference checkout not present (expected in the build container only)")
    os.environ.setdefault("MPLBACKEND", "Agg")
    mpi4py = types.ModuleType("mpi4py")
    MPI = types.ModuleType("mpi4py.MPI")
    MPI.COMM_WORLD = WORLD
    MPI.SUM = "SUM"
    MPI.Request = types.SimpleNamespace(Waitall=lambda reqs: None)      # pcg_solver.py:328
    MPI.Comm = _World
    mpi4py.MPI = MPI
    sys.modules["mpi4py"] = mpi4py
    sys.modules["mpi4py.MPI"] = MPI
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import src.solver.pcg_solver as ref                                  # noqa: E402  (the real code)
    ref.Comm = WORLD
    ref.Rank = _RankProxy()
    ref.N_Workers = 1
    ref.eps = np.finfo(float).eps                                        # pcg_solver.py:972
    _ref = ref
    return ref


def _run_threads(n, fn):
    errs = [None] * n
    outs = [None] * n

    def tgt(r):
        _tls.rank = r
        try:
            outs[r] = fn(r)
        except BaseException as e:      # noqa: BLE001 - includes the reference's `raise Warning`
            errs[r] = e
            try:
                WORLD.barrier_obj.abort()
            except Exception:
                pass

    ths = [threading.Thread(target=tgt, args=(r,)) for r in range(n)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for e in errs:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in errs:
        if e is not None:
            raise e
    return outs


def ref_matvec(parts, xs, mode="Strain"):
    """Reference calcMatVecProd (pcg_solver.py:242-336) on every part; returns copies."""
    ref = load_reference()
    n = len(parts)
    WORLD.configure(n)
    ref.N_Workers = n

    def fn(r):
        if mode == "Strain":
            return np.array(ref.calcMatVecProd(parts[r], "Strain", xs[r]), copy=True)
        return np.array(ref.calcMatVecProd(parts[r], "Preconditioner"), copy=True)

    return _run_threads(n, fn)


def ref_solve(parts, record_history=True):
    """Reference load-step body updateBC -> updatePreconditioner -> PCG (pcg_solver.py:1004-1006).

    Mutates the part dicts exactly as the reference does.  Returns dict(history=[...NormR per
    iteration...], early=[per-part early-return tuple or None], n_allreduce, n_isend_rank0).
    """
    ref = load_reference()
    n = len(parts)
    WORLD.configure(n)
    ref.N_Workers = n
    hist = []
    orig_sum = ref.MPI_SUM

    def logged_sum(v, gd):
        out = orig_sum(v, gd)
        if record_history and getattr(_tls, "rank", 0) == 0 and isinstance(v, np.ndarray) and v.size == 3:
            hist.append(np.sqrt(np.asarray(out, float)).copy())          # [NormP, NormX, NormR]
        return out

    ref.MPI_SUM = logged_sum
    try:
        def fn(r):
            P = parts[r]
            ref.updateBC(P)
            ref.updatePreconditioner(P)
            return ref.PCG(P)

        early = _run_threads(n, fn)
    finally:
        ref.MPI_SUM = orig_sum
    return {"history": np.array(hist).reshape(-1, 3), "early": early,
            "n_allreduce": WORLD.n_allreduce, "n_isend_rank0": WORLD.n_p2p}
