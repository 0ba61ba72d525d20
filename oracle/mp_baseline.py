"""TEST / MEASUREMENT INFRASTRUCTURE - the oracle as R real processes, one mesh part each (the reference's own mode:
`mpiexec -np R python3 src/solver/pcg_solver.py`, one part per rank, one thread per rank, pcg_solver.py:10-15,91).

Used only by bench.py's `cpu_baseline` leg (and its test): it times the reference ALGORITHM on the host cores of the
GPU node, beside the GPU number.  It is never imported by the product package.

Each worker builds ITS part of the synthetic brick, then runs oracle/pcg_oracle.py's update_bc / update_preconditioner /
pcg on a one-element part list - i.e. exactly the per-rank NumPy arithmetic pinned against the reference - with the two
communication points of the oracle rebound from "virtual ranks in one process" to real inter-process exchange:
  pcg_oracle.halo_sum   <- neighbour sum-exchange through per-rank shared-memory outboxes (Isend/Recv/Waitall, :318-334)
  pcg_oracle._allreduce <- sum of the ranks' partials in rank order through a shared table     (MPI_SUM, :622-628)
Synchronisation is a spinning epoch barrier on shared counters (mpi4py / mpiexec are not in this image).  Time inside the
two functions is the 'communication wait' bucket, everything else 'calculation', as the reference's updateTime does
(:631-641); the report gives the mean over ranks like configTimeRecData (file_operations.py:101-109).

usage: python oracle/mp_baseline.py --nodes-per-side 150 --ranks 64 --iters 20 [--numpy] [--octree 1m]   -> one JSON line on stdout
--numpy: the reference's own NumPy expressions for the mat-vec (bit-identical to pcg_solver.py, oracle/make_golden.py) instead of
the C port - what bench.py quotes as cpu_baseline.value since round 4 (the C port is ~2x slower per dof and stays a secondary field).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[_k] = "1"                      # the reference pins its BLAS to one thread per rank (:10-15)

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, "pcg-mpi-solver_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

RED_W = 8                                     # doubles per all-reduce slot


def available_cores():
    """Cores this process may really use: the affinity mask, capped by a cgroup CPU quota when the container has one."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                   # cgroup v2
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())                # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def grid_for(r):
    """px*py*pz = r, as cubic as possible (the reference would get its parts from METIS, run_metis.py:88)."""
    best = (1, 1, r)
    for a in range(1, r + 1):
        if r % a:
            continue
        for b in range(a, r // a + 1):
            if (r // a) % b:
                continue
            c = r // a // b
            if c >= b and (c - a) < (best[2] - best[0]):
                best = (a, b, c)
    return best


class Shm:
    """Named shared memory as a NumPy array (created by one process, attached by the others)."""

    def __init__(self, name, nbytes=0, create=False):
        from multiprocessing import shared_memory
        self.m = shared_memory.SharedMemory(name=name, create=create, size=max(8, nbytes) if create else 0)
        self.create = create

    def array(self, dtype, count, offset=0):
        return np.ndarray((count,), dtype=dtype, buffer=self.m.buf, offset=offset)

    def close(self):
        try:
            self.m.close()
            if self.create:
                self.m.unlink()
        except Exception:
            pass


OCTREE_ROOTS = {"1m": (12, 12, 12), "10m": (38, 38, 38), "tiny": (3, 3, 3)}


def worker(rank, R, N, iters, prefix, use_c):
    """N: nodes per side of the brick (int), or "octree:<size>" = the graded octree mesh of bench.py's octree workload
    (pcg_mi355x.octree.GradedOctreeMesh, parts by recursive bisection of the element centroids - the METIS stand-in)."""
    import pcg_oracle
    from pcg_mi355x.brick import Brick, make_parts, block_partition
    ctl = Shm(prefix + "_ctl")
    arrive = ctl.array(np.int64, R, 0)                                    # epoch each rank has reached
    red = ctl.array(np.float64, 2 * R * RED_W, 8 * R).reshape(2, R, RED_W)  # double-buffered by epoch parity
    sizes = ctl.array(np.int64, R, 8 * R + 8 * 2 * R * RED_W)             # outbox doubles per rank
    epoch = [0]
    t_comm = [0.0]

    def barrier():
        epoch[0] += 1
        arrive[rank] = epoch[0]
        e = epoch[0]
        spins = 0
        while int(arrive.min()) < e:
            spins += 1
            if spins > 2000:                      # a late rank: stop burning the core it may need
                time.sleep(5e-5)

    if isinstance(N, str) and N.startswith("octree:"):
        from pcg_mi355x.octree import GradedOctreeMesh, make_octree_parts, bisect_elements
        mesh = GradedOctreeMesh(OCTREE_ROOTS[N.split(":", 1)[1]], 4 if N != "octree:tiny" else 3, band=1.2, seed=0, symmetry=True)
        P = make_octree_parts(mesh, R, elem_part=bisect_elements(mesh, R) if R > 1 else None, only=[rank])[0]
        del mesh
    else:
        brick = Brick(N, seed=0)
        P = make_parts(brick, block_partition(brick, *grid_for(R)) if R > 1 else None, only=[rank])[0]
    nbrs = [int(v) for v in P["NbrMPIdVector"]]
    ovl = [np.asarray(v, np.int64) for v in P["OvrlpLocalDofVecList"]]
    cnt = [len(v) for v in ovl]
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    # outbox: header (R int64 offsets, -1 = not a neighbour) + 2 x payload (double-buffered)
    n_out = int(off[-1])
    box = Shm(f"{prefix}_box{rank}", 8 * R + 16 * max(1, n_out), create=True)
    hdr = box.array(np.int64, R, 0)
    hdr[:] = -1
    for j, q in enumerate(nbrs):
        hdr[q] = off[j]
    pay = box.array(np.float64, 2 * max(1, n_out), 8 * R).reshape(2, max(1, n_out))
    sizes[rank] = n_out
    barrier()
    peers = {}
    for j, q in enumerate(nbrs):
        b = Shm(f"{prefix}_box{q}")
        h = b.array(np.int64, R, 0)
        nq = int(sizes[q])
        peers[q] = (b, int(h[rank]), b.array(np.float64, 2 * max(1, nq), 8 * R).reshape(2, max(1, nq)))
    halo_epoch = [0]

    def halo_sum(parts, ys):                                              # pcg_solver.py:303-334 for ONE local part
        t0 = time.perf_counter()
        y = ys[0]
        k = halo_epoch[0] & 1
        halo_epoch[0] += 1
        for j in range(len(nbrs)):                                        # :307-312 pack (before any +=)
            pay[k, off[j]:off[j + 1]] = y[ovl[j]]
        barrier()                                                         # :318-328 Isend / Recv / Waitall
        for j, q in enumerate(nbrs):                                      # :333-334 += in neighbour order
            _, o, pq = peers[q]
            y[ovl[j]] += pq[k, o:o + cnt[j]]
        t_comm[0] += time.perf_counter() - t0
        return ys

    def allreduce(vals):                                                  # MPI_SUM :622-628, rank order
        t0 = time.perf_counter()
        v = np.atleast_1d(np.asarray(vals[0], float))
        e = (epoch[0] + 1) & 1
        red[e, rank, :len(v)] = v
        barrier()
        tot = red[e, 0, :len(v)].copy()
        for q in range(1, R):
            tot = tot + red[e, q, :len(v)]
        t_comm[0] += time.perf_counter() - t0
        return tot if np.ndim(vals[0]) else float(tot[0])

    pcg_oracle.halo_sum = halo_sum
    pcg_oracle._allreduce = allreduce
    gd = P["GlobData"]
    barrier()
    t0 = time.perf_counter()
    pcg_oracle.update_bc([P], use_c=use_c)
    pcg_oracle.update_preconditioner([P])
    t_setup = time.perf_counter() - t0
    gd["MaxIter"] = iters
    barrier()
    t_comm[0] = 0.0
    t0 = time.perf_counter()
    out = pcg_oracle.pcg([P], use_c=use_c, record=False)
    t_solve = time.perf_counter() - t0
    barrier()
    res = {"rank": rank, "t_solve": t_solve, "t_comm": t_comm[0], "t_setup": t_setup, "n_matvec": out["n_matvec"],
           "flag": int(out["flag"]), "relres": float(out["relres"]), "ndof": int(P["NDOF"]), "n_nbr": len(nbrs)}
    for b, _, _ in peers.values():
        b.close()
    barrier()
    box.close()
    ctl.close()
    return res


def _entry(args):
    try:
        return worker(*args)
    except BaseException as e:      # noqa: BLE001
        import traceback
        return {"error": "".join(traceback.format_exception(type(e), e, e.__traceback__))}


def run(N, R, iters, use_c=True):
    import multiprocessing as mp
    import subprocess
    if use_c:
        subprocess.check_call(["make", "-s", "-C", HERE])
    prefix = f"pcgmp{os.getpid()}"
    ctl = Shm(prefix + "_ctl", 8 * R + 8 * 2 * R * RED_W + 8 * R, create=True)
    ctl.array(np.int64, R, 0)[:] = 0
    try:
        ctx = mp.get_context("spawn")                        # never fork a process that may hold GPU / thread state
        t0 = time.perf_counter()
        with ctx.Pool(R) as pool:
            res = pool.map(_entry, [(r, R, N, iters, prefix, use_c) for r in range(R)], chunksize=1)
        wall = time.perf_counter() - t0
    finally:
        ctl.close()
    bad = [r["error"] for r in res if "error" in r]
    if bad:
        raise RuntimeError(bad[0])
    t_solve = max(r["t_solve"] for r in res)
    comm = float(np.mean([r["t_comm"] for r in res]))
    return {"value": iters / t_solve, "unit": "iterations/s", "cores": R, "kind": "port" if use_c else "reference arithmetic (NumPy)",
            "grid": list(grid_for(R)) if not isinstance(N, str) else f"{R} parts by recursive bisection",
            "iterations": iters, "n_matvec": res[0]["n_matvec"], "t_solve_s": t_solve,
            "calc_s_mean": float(np.mean([r["t_solve"] - r["t_comm"] for r in res])), "comm_wait_s_mean": comm,
            "dofs_per_rank_max": max(r["ndof"] for r in res), "neighbours_max": max(r["n_nbr"] for r in res),
            "relres_after": res[0]["relres"], "wall_incl_setup_s": wall}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes-per-side", type=int, default=150)
    ap.add_argument("--octree", choices=sorted(OCTREE_ROOTS), default=None, help="the graded octree mesh of bench.py --workload octree instead of the brick")
    ap.add_argument("--ranks", type=int, default=0, help="0 = min(available cores, 64)")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--numpy", action="store_true", help="NumPy EBE mat-vec (the reference's expressions) instead of the C port")
    a = ap.parse_args()
    R = a.ranks or min(available_cores(), 64)
    print(json.dumps(run(f"octree:{a.octree}" if a.octree else a.nodes_per_side, R, a.iters, not a.numpy)), flush=True)


if __name__ == "__main__":
    main()
