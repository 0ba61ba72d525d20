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

    # -- what partition_mesh.py / file_operations.py additionally touch, single worker only ------------
    def Split_type(self, kind):                         # partition_mesh.py:92, file_operations.py:308
        assert self.n == 1
        return self

    def Allgather(self, send, recv):                    # partition_mesh.py:681
        assert self.n == 1
        recv[...] = np.asarray(send).reshape(recv.shape)

    def bcast(self, value, root=0):                     # partition_mesh.py:1352
        assert self.n == 1
        return value


class _Win:
    """MPI.Win.Allocate_shared stand-in: plain process memory (file_operations.py:320-327)."""

    def __init__(self, nbytes, itemsize):
        self.buf, self.itemsize = bytearray(int(nbytes)), itemsize

    @classmethod
    def Allocate_shared(cls, nbytes, itemsize, comm=None):
        return cls(nbytes, itemsize)

    def Shared_query(self, rank):
        return self.buf, self.itemsize


class _File:
    """MPI.File stand-in (partition_mesh.py:1362-1369): Open / Write / Close on an ordinary file."""

    def __init__(self, name):
        self.f = open(name, "wb")

    @classmethod
    def Open(cls, comm, name, amode=0):
        return cls(name)

    def Write(self, buf):
        self.f.write(np.asarray(buf).tobytes())

    def Close(self):
        self.f.close()


def _dtype_size(n):
    return types.SimpleNamespace(Get_size=lambda: n)


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
    MPI.COMM_SELF = WORLD
    MPI.COMM_TYPE_SHARED = 0
    MPI.LONG, MPI.DOUBLE, MPI.BOOL = _dtype_size(8), _dtype_size(8), _dtype_size(1)     # file_operations.py:311-313
    MPI.Win = _Win
    MPI.File = _File
    MPI.MODE_WRONLY, MPI.MODE_CREATE = 1, 2
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


_part = None


def load_partitioner():
    """Import the reference's partitioner + METIS front end (src/solver/partition_mesh.py, run_metis.py) with the
    fake MPI in place; module globals as their `__main__` sets them for ONE worker (partition_mesh.py:1374-1376)."""
    global _part
    if _part is not None:
        return _part
    load_reference()
    import src.solver.partition_mesh as pmesh                           # noqa: E402  (the real code)
    import src.solver.run_metis as rmetis                               # noqa: E402
    pmesh.Comm = WORLD
    pmesh.Rank = 0
    pmesh.N_Workers = 1
    _part = (pmesh, rmetis)
    return _part


def ref_partition(work_dir, mdf_path, out_prefix, n_parts):
    """Run the reference's pipeline stage 2+3 unmodified on the model in `mdf_path` (MeshPart_<n>.npy must exist):
    run_metis.config_GlobData (run_metis.py:19-43), then partition_mesh's `__main__` sequence (:1378-1418) with one
    worker owning all parts.  Writes <out_prefix><n>_<id>.mpidat + _metadat.npy through exportMP and returns the
    list of RefMeshPart dicts exactly as exported."""
    import pickle
    import zlib
    pmesh, rmetis = load_partitioner()
    WORLD.configure(1)
    mdf_path = os.path.join(mdf_path, "")
    rmetis.config_GlobData(mdf_path, mdf_path + "MeshData_Glob.zpkl")
    os.makedirs(os.path.join(work_dir, "__pycache__"), exist_ok=True)
    os.makedirs(os.path.dirname(out_prefix), exist_ok=True)
    paths = {"ScratchPath": work_dir, "MDF_Path": mdf_path, "PyDataPath_Part": out_prefix, "ModelName": "synthetic"}
    with open(os.path.join(work_dir, "__pycache__", "ModelDataPaths.zpkl"), "wb") as f:     # read_input_model.py:44-45
        f.write(zlib.compress(pickle.dumps(paths, pickle.HIGHEST_PROTOCOL)))
    cwd, argv = os.getcwd(), sys.argv
    try:
        os.chdir(work_dir)
        sys.argv = ["partition_mesh.py", str(n_parts), "0"]
        gd = pmesh.initModelData()
        mpg = {"GlobData": gd, "PotentialNbrDataFlag": False}
        pmesh.extract_Elepart(mpg)
        pmesh.extract_PlotSettings(mpg)
        pmesh.extract_ElemMeshData(mpg)
        pmesh.config_ElemVectors(mpg)
        pmesh.extract_NodalVectors(mpg)
        pmesh.config_TypeGroupList(mpg)
        pmesh.config_ElemMaterial(mpg)
        pmesh.config_ElemLib(mpg)
        pmesh.config_IntfcElem(mpg)
        pmesh.identify_PotentialNeighbours(mpg)
        pmesh.config_Neighbours(mpg)
        pmesh.config_NonlocalNeighbours(mpg)
        pmesh.exportMP(mpg)
    finally:
        os.chdir(cwd)
        sys.argv = argv
    meta = np.load(out_prefix + str(n_parts) + "_metadat.npy", allow_pickle=True).item()
    parts = []
    for k in range(n_parts):                                            # pcg_solver.py:100-106
        buf = np.fromfile(out_prefix + str(n_parts) + "_" + str(k) + ".mpidat", dtype=meta["DTypeData"][k], count=meta["NfData"][k])
        parts.append(pickle.loads(zlib.decompress(buf.tobytes())))
    return parts


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
