"""Inter-GPU communication of the PCG hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU node; "gloo" in the CPU test-suite).

Replaces the reference's mpi4py calls (src/solver/pcg_solver.py):
  * interface sum-exchange  Isend / Recv / Waitall  (:318-328)  -> ONE all_to_all_single with the
    per-neighbour counts as split sizes (RCCL lowers it to grouped ncclSend/ncclRecv, i.e. every
    neighbour pair uses its own direct xGMI link concurrently).  It is issued asynchronously
    right after the interface rows are packed, runs on RCCL's own stream, and the engine's compute
    stream only waits for it after the interior rows have been launched (halo_begin / halo_end).
  * MPI_SUM = pickled allreduce of 1, 1 and 3 doubles per iteration (:463,:488,:507) -> two f64
    all_reduce calls per iteration (p.Ap ; [|p|^2,|x|^2,|r|^2, next rho, #inf]) on device buffers.
Part id == rank (one part per rank, :91, dest=NbrMP_Id :320-321).

The engine calls these hooks through C function pointers (include/pcg_mi355x.h, pcg_comm_hooks);
buffers arrive as raw device pointers and are viewed as torch tensors without a copy.
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib

__all__ = ["TorchComm"]


class _DevView:
    """Minimal __cuda_array_interface__ carrier for a raw f64 device buffer."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class TorchComm:
    def __init__(self, group=None, device=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.on_gpu = self.backend == "nccl"
        self.device = device
        self._views = {}
        self._streams = {}
        self._exc = None
        self._work = None
        self.t_comm = 0.0
        self.n_allreduce = 0
        self.n_halo = 0

    # -- pointer -> tensor ------------------------------------------------------------------------
    def _tensor(self, ptr, n):
        key = (ptr, n)
        t = self._views.get(key)
        if t is None:
            if self.on_gpu:
                t = torch.as_tensor(_DevView(ptr, n), device=self.device if self.device is not None else "cuda")
            else:
                arr = np.ctypeslib.as_array((C.c_double * n).from_address(ptr))
                t = torch.from_numpy(arr)
            self._views[key] = t
        return t

    def _use_stream(self, stream_ptr):
        """Make the engine's HIP stream torch's current stream (sticky: one switch per solve, not one
        context manager per hook call - the hooks run 3-4 times per PCG iteration)."""
        if not self.on_gpu:
            return
        key = int(stream_ptr or 0)
        ext = self._streams.get(key)
        if ext is None:
            ext = self._streams[key] = torch.cuda.ExternalStream(key)
        if torch.cuda.current_stream().cuda_stream != key:
            torch.cuda.set_stream(ext)

    def release_stream(self, stream_ptr):
        """Called before an engine (and its HIP stream) is destroyed: never leave torch's current stream
        pointing at a dead handle, and drop the cached views of that engine's buffers."""
        key = int(stream_ptr or 0)
        if self.on_gpu and key in self._streams:
            if torch.cuda.current_stream().cuda_stream == key:
                torch.cuda.set_stream(torch.cuda.default_stream())
            del self._streams[key]
        self._views.clear()

    def reraise(self):
        if self._exc is not None:
            e, self._exc = self._exc, None
            raise e

    # -- hooks ----------------------------------------------------------------------------------------
    def make_hooks(self, op):
        """Build the pcg_comm_hooks struct for one Operator (its neighbour ids and counts)."""
        splits = [0] * self.world
        for pid, cnt in zip(op.peer_ids, op.peer_counts):
            if not (0 <= pid < self.world) or pid == self.rank:
                raise ValueError(f"neighbour part id {pid} is not a valid peer rank")
            splits[pid] = cnt
        # the send buffer is ordered by neighbour list position; all_to_all needs rank order
        if list(op.peer_ids) != sorted(op.peer_ids):
            raise ValueError("NbrMPIdVector must be ascending (partition_mesh.py builds it in part-id order)")

        def halo_begin(ctx, send_p, recv_p, count, stream_p):
            try:
                t0 = time.perf_counter()
                send = self._tensor(send_p, count)
                recv = self._tensor(recv_p, count)
                self._use_stream(stream_p)
                self._work = dist.all_to_all_single(recv, send, splits, splits, group=self.group, async_op=True)
                self.n_halo += 1
                self.t_comm += time.perf_counter() - t0
                return 0
            except BaseException as e:      # noqa: BLE001 - must not propagate through the C frame
                self._exc = e
                return -1

        def halo_end(ctx, stream_p):
            try:
                t0 = time.perf_counter()
                self._use_stream(stream_p)
                self._work.wait()
                self._work = None
                self.t_comm += time.perf_counter() - t0
                return 0
            except BaseException as e:      # noqa: BLE001
                self._exc = e
                return -1

        def allreduce(ctx, buf_p, count, stream_p):
            try:
                t0 = time.perf_counter()
                t = self._tensor(buf_p, count)
                self._use_stream(stream_p)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                self.n_allreduce += 1
                self.t_comm += time.perf_counter() - t0
                return 0
            except BaseException as e:      # noqa: BLE001
                self._exc = e
                return -1

        hooks = _lib.CommHooks(None, _lib.HALO_BEGIN_T(halo_begin), _lib.HALO_END_T(halo_end),
                               _lib.ALLREDUCE_T(allreduce))
        hooks._keep = (halo_begin, halo_end, allreduce)      # keep the closures alive
        return hooks
