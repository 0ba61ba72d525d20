"""TEST INFRASTRUCTURE: several mesh parts in ONE process, one thread per part, exchanging through process memory.

The engine's multi-part code - interface-first ordering, k_halo_pack, the exchange hooks, k_fixup, the per-phase
element launches, the all-reduce hooks inside the look-ahead loop - normally needs one GPU per part.  The test box
has one.  This communicator implements the pcg_comm_hooks protocol (include/pcg_mi355x.h) between THREADS, so that
2..8 engines (one per part, all on the same device) run the real kernels and the real driver against each other:
  halo_begin : wait for the engine stream, hand a copy of every neighbour segment of the send buffer to that neighbour,
               meet at a barrier, copy what the neighbours handed over into the receive buffer
  halo_end   : nothing left to do
  allreduce  : wait for the engine stream, sum the values of all parts in rank order, write the sum back
Device buffers are viewed as CUDA tensors (HIP engine) or NumPy arrays (CPU test double); everything is synchronous -
this is about correctness of the multi-part kernels on the GPU, not about speed."""
import ctypes as C
import threading

import numpy as np

from pcg_mi355x import _lib


class ThreadWorld:
    def __init__(self, n, on_gpu):
        self.n, self.on_gpu = n, on_gpu
        self.barrier = threading.Barrier(n)
        self.slots = [None] * n
        self.mail = {}

    def comm(self, rank):
        return ThreadComm(self, rank)


class ThreadComm:
    def __init__(self, world, rank):
        self.w, self.rank, self.world = world, rank, world.n
        self._exc = None
        self.n_allreduce = self.n_halo = 0

    def reraise(self):
        if self._exc is not None:
            e, self._exc = self._exc, None
            raise e

    def _view(self, ptr, n):
        if not n:                   # a part without neighbours enters the exchange with nothing to send
            return np.zeros(0)
        if self.w.on_gpu:
            import torch
            from pcg_mi355x.dist import _DevView
            return torch.as_tensor(_DevView(ptr, n), device="cuda")
        return np.ctypeslib.as_array((C.c_double * n).from_address(ptr))

    def _sync(self, stream_p):
        if self.w.on_gpu:
            import torch
            torch.cuda.ExternalStream(int(stream_p or 0)).synchronize()

    def _host(self, v):
        return v.cpu().numpy().copy() if (self.w.on_gpu and not isinstance(v, np.ndarray)) else np.array(v, copy=True)

    def _store(self, v, values):
        if isinstance(v, np.ndarray) and not v.size:
            return
        if self.w.on_gpu:
            import torch
            v.copy_(torch.from_numpy(np.ascontiguousarray(values)))
            torch.cuda.synchronize()
        else:
            v[...] = values

    def make_hooks(self, op):
        peers, counts = list(getattr(op, "peer_ids", [])), list(getattr(op, "peer_counts", []))
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(int)

        def halo_begin(ctx, send_p, recv_p, count, stream_p):
            try:
                self._sync(stream_p)
                send = self._host(self._view(send_p, count))
                for k, q in enumerate(peers):
                    self.w.mail[(self.rank, q)] = send[offs[k]:offs[k + 1]]
                self.w.barrier.wait()
                got = np.concatenate([self.w.mail[(q, self.rank)] for q in peers]) if peers else np.zeros(0)
                assert len(got) == count
                self._store(self._view(recv_p, count), got)
                self.w.barrier.wait()
                self.n_halo += 1
                return 0
            except BaseException as e:      # noqa: BLE001 - must not propagate through the C frame
                self._exc = e
                self.w.barrier.abort()
                return -1

        def halo_end(ctx, stream_p):
            return 0

        def allreduce(ctx, buf_p, count, stream_p):
            try:
                self._sync(stream_p)
                v = self._view(buf_p, count)
                self.w.slots[self.rank] = self._host(v)
                self.w.barrier.wait()
                total = self.w.slots[0].copy()
                for r in range(1, self.world):
                    total = total + self.w.slots[r]
                self.w.barrier.wait()
                self._store(v, total)
                self.n_allreduce += 1
                return 0
            except BaseException as e:      # noqa: BLE001
                self._exc = e
                self.w.barrier.abort()
                return -1

        hooks = _lib.CommHooks(None, _lib.HALO_BEGIN_T(halo_begin), _lib.HALO_END_T(halo_end), _lib.ALLREDUCE_T(allreduce), 1)   # barrier-based: collective
        hooks._keep = (halo_begin, halo_end, allreduce)
        return hooks


def solve_parts_in_threads(parts, kind, on_gpu):
    """updateBC -> updatePreconditioner -> PCG on every part, one thread per part.  Returns the per-part SolveInfo."""
    import pcg_mi355x as pm
    from pcg_mi355x.operator import from_refmeshpart
    world = ThreadWorld(len(parts), on_gpu)
    ops, infos, errs = [None] * len(parts), [None] * len(parts), [None] * len(parts)

    def run(r):
        try:
            P = parts[r]
            comm = world.comm(r)
            ops[r] = op = from_refmeshpart(P, comm=comm, kind=kind)
            fext, udi = op.update_bc(P["RefLoadVector"], P["Ud"], 1.0)
            inv = op.build_jacobi()
            gd = P["GlobData"]
            x, res, hist = op.solve(fext, P["Un"], inv, gd["Tol"], gd["MaxIter"], gd["GlobNDofEff"], history=True)
            P["Un"] = x + udi
            infos[r] = pm.solver.SolveInfo(res, hist)
        except BaseException as e:          # noqa: BLE001
            errs[r] = e
            try:
                world.barrier.abort()
            except Exception:
                pass

    ths = [threading.Thread(target=run, args=(r,)) for r in range(len(parts))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for op in ops:
        if op is not None:
            op.close()
    for e in errs:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in errs:
        if e is not None:
            raise e
    return infos
