#!/usr/bin/env python
"""bench.py - PCG iterations/sec + SpMV achieved HBM GB/s on the BASELINE.json workload.

  python bench.py --gpus N --steps K --warmup W
      N = 1 runs in this process.  N > 1: bench.py launches its own N ranks (python -m torch.distributed.run ...
      bench.py, one rank per GPU) unless it already runs inside such a launch (RANK / WORLD_SIZE in the environment),
      so both `python bench.py --gpus 8` and `python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8` work.

A "step" is ONE full PCG iteration of the reference algorithm (src/solver/pcg_solver.py:438-562: operator apply +
interface exchange, the weighted dots, the vector updates, the status read-back) on the synthetic elasticity brick of
SURVEY.md 8(d) (default N=150 nodes per side: n = 10 125 000, nnz = 809 238 528 = the metric's configuration;
--nodes-per-side 322 is the 100 M-dof system of BASELINE configs[4]), inputs resident in HBM.  W warm-up iterations,
then exactly K timed iterations between barrier + synchronize fences; the max over ranks is reported.  N > 1 splits the
SAME system into N parts, one per GPU (strong scaling); every rank builds only its own part and operator.  The
multi-GPU data path is the engine's native communicator (csrc/rccl_comm.hip): grouped ncclSend/ncclRecv on a
communication stream overlapped with the interior rows, two ncclAllReduce per iteration - no Python in the loop.

OUTPUT (round 6).  The LAST stdout line is a compact headline (< 4 KB, numbers and short names only - benchlib/line.py):
the contract's fields, `roofline` (dominant kernel k_spmv: stored bytes per launch / mean launch time from HIP events on the
engine's stream inside the timed window / 8 TB/s; `traffic` = HBM bytes per launch from two rocprofv3 --pmc passes of a short
child run; `scalar_csr_frac` = SURVEY 8(d)'s literal 12 nnz + 20 n kernel at the same size), `cpu_baseline` (the oracle - the
reference's NumPy arithmetic, bit-identical - as R processes x 1 thread on this box's host cores, + a one-core sample at the
same size), `roofline_iteration` (the whole iteration against N x 8 TB/s), and `also` (it/s of the matrix-free operator and of
the octree meshes).  The FULL record - every object with its notes - goes to bench_extras.json beside this file.
Every object after the headline window is optional: guarded by try/except and by a wall-clock budget (--extras-budget-s).
--full adds the frozen / informational objects (dictionary format, orientation variants, C-port CPU points, octree CPU runs).
The opt-in engine-side communication forms are measured only with --ab-engine-side (frozen).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "pcg-mpi-solver_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(k, "1")          # the reference's mode (pcg_solver.py:10-15); set before NumPy loads
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from benchlib import METRIC, HBM_PEAK_GBS, F64_PEAK_TFLOPS, log                                        # noqa: E402
from benchlib.launch import launch_ranks, _free_port                                                   # noqa: E402,F401
from benchlib.cpu import cpu_baseline, cpu_baseline_single, numpy_reference_point, scipy_csr_spmv_point  # noqa: E402,F401
from benchlib.points import scalar_csr_point, scalar_csr_point_device, box_identity                    # noqa: E402,F401
from benchlib.pmc import pmc_child, pmc_traffic_live                                                   # noqa: E402
from benchlib.octree import octree_object                                                              # noqa: E402
from benchlib.line import compact_line, write_extras, Budget                                           # noqa: E402


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nodes-per-side", type=int, default=int(os.environ.get("PCG_BENCH_N", "150")),
                    help="brick size N (150 -> 10M dof = the metric's configuration; 70 -> 1M; 322 -> 100M = BASELINE configs[4])")
    ap.add_argument("--workload", choices=["brick", "octree"], default="brick",
                    help="brick = SURVEY 8(d) uniform brick (the metric's configuration); octree = multi-level 2:1 graded mesh with "
                         "hanging-node transition patterns as the HEADLINE workload (BASELINE configs[1] names an octree mesh)")
    ap.add_argument("--octree-size", choices=["1m", "10m"], default="1m", help="--workload octree: 1 M dof (BASELINE configs[1]) or 10 M dof")
    ap.add_argument("--no-octree", action="store_true", help="skip the `octree` (1 M dof) and `octree_10m` objects")
    ap.add_argument("--rows-per-lane", type=int, default=int(os.environ.get("PCG_ROWS_PER_LANE", "0")))
    ap.add_argument("--operator", choices=["sell", "ebe", "dict", "both"], default="both",
                    help="sell = assembled SELL-BSR3 matrix only (the headline value / roofline); both = also the matrix-free operator on the "
                         "same system (and, with --full, the value-dictionary format)")
    ap.add_argument("--comm", choices=["native", "torch"], default=os.environ.get("PCG_BENCH_COMM", "native"),
                    help="N > 1: native = RCCL calls issued by the engine (default); torch = torch.distributed callbacks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-ranks", type=int, default=0, help="processes of the multi-core CPU baseline (0 = min(cores, 64))")
    ap.add_argument("--no-finish", action="store_true", help="do not run the solve to convergence after the timed window")
    ap.add_argument("--no-pmc-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc passes that measure the HBM traffic of the SpMV launch on this box (N = 1; they run "
                         "by default when rocprofv3 is on PATH and this process is not itself being profiled)")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)       # internal: the workload of one rocprofv3 --pmc pass (benchlib/pmc.py)
    ap.add_argument("--pmc-traffic", action="store_true", help="N = 1: insist on the PMC passes (even under a profiler)")
    ap.add_argument("--full", action="store_true",
                    help="every object of rounds 1 - 5 (dictionary format, orientation variants, C-port / scipy CPU points, CPU runs of the octree "
                         "meshes, PMC passes over every operator) and no wall-clock budget")
    ap.add_argument("--extras-budget-s", type=float, default=float(os.environ.get("PCG_BENCH_EXTRAS_BUDGET_S", "150")),
                    help="wall-clock seconds for the optional objects after the headline window; an object is skipped when its estimate no longer fits")
    ap.add_argument("--ab-engine-side", action="store_true",
                    help="N > 1: also measure the opt-in engine-side communication forms (mailbox all-reduce, direct exchange) beside the headline")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.pmc_child:
        return pmc_child(args)
    if args.full:
        args.extras_budget_s = 1e9
    t_wall0 = time.perf_counter()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    share = os.environ.get("PCG_BENCH_SHARE_GPU") == "1"      # dry run: all ranks on device 0 (1-GPU box, RCCL stand-in)
    dev = 0 if share else local_rank
    torch.cuda.set_device(dev)
    if world > 1:
        import datetime
        # control plane (barriers, the max over ranks, the unique-id broadcast); a rank that dies takes the job down in minutes
        if share:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev), timeout=datetime.timedelta(seconds=300))

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build_engine()      # no-op when the in-tree build is current (hipcc --offload-arch=gfx950 otherwise)
    if world > 1:
        dist.barrier()
    import pcg_mi355x as pm
    from pcg_mi355x import _lib
    from pcg_mi355x.brick import Brick, make_parts, block_partition, default_grid
    from pcg_mi355x.dist import RcclComm, TorchComm
    _lib.use_library(None)
    assert _lib.backend_name() == "hip-gfx950"
    comm, transport = None, "none (one part)"
    if world > 1:
        native_err = None
        if args.comm == "native":
            # The driver launches the ranks itself (torch.distributed.run), so the retry of launch_ranks() is not around them: a native
            # communicator that cannot be CREATED (library missing, a symbol, an RCCL error every rank sees) must not take the job down -
            # the ranks agree over the control plane and fall back to the torch callbacks together; the line says so (comm.native_error).
            try:
                comm = RcclComm.from_torch(dev)
                assert comm.world == world
            except Exception as ex:                    # noqa: BLE001
                native_err = repr(ex)[:300]
                comm = None
            if native_err is None and os.environ.get("PCG_BENCH_NO_PROBE") != "1":
                # First contact (round 6): before the 10 M-dof system is built, ONE small solve of each operator through the native
                # communicator - exchange, all-reduces, look-ahead, a 2 197-node brick in `world` parts - so that an RCCL call that
                # raises on this node (a refused send / recv group, a symbol, a stream mode) costs two seconds and ends in the
                # collective fall-back below instead of in the headline window with nothing printed.
                try:
                    pb = Brick(13, seed=0)
                    pp = make_parts(pb, block_partition(pb, *default_grid(world)), only=[rank])[0]
                    for kind in ("sell", "ebe"):
                        pp.pop("_pcg_mi355x_operator", None)
                        pp["Un"] = np.zeros(pp["NDOF"])                  # (not the warm start the first operator's solution would be)
                        pm.configure(comm=comm, device=dev, operator=kind)
                        pm.update_bc(pp); pm.update_preconditioner(pp); pm.solve(pp)
                        info = pp["_pcg_mi355x_info"]
                        pp.pop("_pcg_mi355x_operator").close()
                        if info.flag != 0:
                            raise RuntimeError(f"first-contact probe [{kind}]: flag {info.flag} after {info.iter} iterations")
                    if rank == 0:
                        log(f"first-contact probe of the native communicator: ok ({info.iter} iterations on {world} parts)")
                except Exception as ex:                # noqa: BLE001
                    native_err = "first-contact probe: " + repr(ex)[:260]
            bad = torch.tensor([0 if native_err is None else 1], dtype=torch.int32, device=torch.device("cpu") if share else torch.device("cuda", dev))
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if int(bad.item()):
                if comm is not None:
                    comm.close()
                    comm = None
                native_err = native_err or "the native communicator failed on another rank"
                log(f"[rank {rank}] native communicator not available ({native_err}); falling back to torch.distributed callbacks")
                os.environ["PCG_BENCH_NATIVE_FAILED"] = "1"
                os.environ.setdefault("PCG_BENCH_NATIVE_ERROR", native_err)
                args.comm = "torch"
        if comm is not None:
            transport = "native: engine-issued RCCL (grouped ncclSend/ncclRecv on a comm stream, ncclAllReduce on the compute stream)"
            if os.environ.get("PCG_RCCL_LIB"):
                transport += f" [PCG_RCCL_LIB={os.path.basename(os.environ['PCG_RCCL_LIB'])}]"
        else:
            comm = TorchComm(device=torch.device("cuda", dev))
            transport = "torch.distributed callbacks (all_to_all_single + all_reduce, backend %s)" % comm.backend
            if os.environ.get("PCG_BENCH_NATIVE_FAILED") == "1":
                transport += " - the native communicator failed on this node, see stderr"
    pm.configure(comm=comm, device=dev, rows_per_lane=args.rows_per_lane)

    N = args.nodes_per_side
    t0 = time.perf_counter()
    if args.workload == "octree":
        from pcg_mi355x.octree import GradedOctreeMesh, make_octree_parts, bisect_elements
        roots = {"1m": (12, 12, 12), "10m": (38, 38, 38)}[args.octree_size]
        brick = GradedOctreeMesh(roots, 4, band=1.2, seed=0, symmetry=True)        # .n_dof like a Brick; nnz filled in after assembly
        grid = (world, 1, 1)
        part = make_octree_parts(brick, world, elem_part=bisect_elements(brick, world) if world > 1 else None, only=[rank])[0]
        brick.nnz = None
        sm = brick.summary()
        wl_name = (f"multi-level 2:1-balanced octree mesh around a sphere, {sm['levels']} cell sizes, {sm['pattern_types']} pattern types in "
                   f"{sm['pattern_orientations']} orientations (up to {sm['nodes_per_element_max']} nodes per element), {brick.n_dof} dof, parts by recursive bisection")
    else:
        brick = Brick(N, seed=0)
        grid = default_grid(world)
        part = make_parts(brick, block_partition(brick, *grid) if world > 1 else None, only=[rank])[0]
        wl_name = f"synthetic 3D elasticity brick N={N} ({brick.n_dof} dof, {brick.nnz} nnz)"
    t_parts = time.perf_counter() - t0

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def gather_max(x):
        if world == 1:
            return float(x), [float(x)]
        box = [None] * world
        dist.all_gather_object(box, float(x))
        return max(box), box

    def gather_sum(x):
        if world == 1:
            return float(x)
        box = [None] * world
        dist.all_gather_object(box, float(x))
        return float(sum(box))

    def iteration_roofline(mm, steps):
        """The whole iteration against the HBM roofline (north_star: it/s "as fraction of HBM roofline"): the bytes an iteration has to
        move - the stored operator of every rank (pcg_operator_cost) + 73 B/dof of the vector phase (p, q, r, x, M^-1 read, 1 flag
        byte, r', x' written, p read and p' written) - over the measured time per step and the N GPUs' 8 TB/s each."""
        t = mm["elapsed"] / steps
        ach = mm["iter_bytes"] / t / 1e9
        return {"bound": "hbm", "bytes_per_iteration": mm["iter_bytes"], "ms_per_step": t * 1e3, "achieved": ach, "peak": HBM_PEAK_GBS * world,
                "unit": "GB/s", "frac": ach / (HBM_PEAK_GBS * world),
                "bytes_definition": "sum over the ranks of: stored operator bytes of one apply (pcg_operator_cost) + 73 B per local dof (vector phase)"}

    def measure(kind, part=part, steps=args.steps, warmup=args.warmup, standalone_reps=100):
        """Set up the operator of `kind`, run W warm-up + K timed PCG iterations, finish the solve."""
        part.pop("_pcg_mi355x_operator", None)
        pm.configure(comm=comm, device=dev, rows_per_lane=args.rows_per_lane, operator=kind)
        t0 = time.perf_counter()
        op = pm.get_operator(part)                   # native host set-up + upload (not timed)
        t_setup = time.perf_counter() - t0
        pm.update_bc(part)                           # Fext  (:226-238)
        pm.update_preconditioner(part)               # Jacobi (:346-352)
        if world == 1:                               # sanity (not timed): A . rigid translation == 0
            t = np.zeros(op.n); t[2::3] = 1.0
            rb = np.abs(op.apply(t)).max()
            log(f"[{kind}] set-up {t_setup:.1f}s; self-check |A.t_z|_max = {rb:.2e}")
            assert rb < 1e-9
        gd = part["GlobData"]
        eff = np.asarray(part["LocDofEff"], np.int64)
        inv = np.zeros(op.n); inv[eff] = part["InvDiagPreCondVector0"]
        extra = 2 * steps                       # the instrumented window (+ N > 1: a window with the communication timers on)
        max_iter = max(int(gd["MaxIter"]), warmup + steps + extra + 1)
        op.solve_begin(part["Fext"], np.zeros(op.n), inv, float(gd["Tol"]), max_iter, int(gd["GlobNDofEff"]))
        r = op.solve_run(warmup)
        assert r.status == 4 and r.iters_done == warmup, "solve ended inside the warm-up window"
        # Timed window: exactly K iterations.  The headline operator (sell) carries HIP events around its operator launches INSIDE
        # the window (the roofline's launch time is of the timed region; two event records against a 1.2 ms iteration); the
        # short-iteration operators are timed un-instrumented and their kernels in a second window of K iterations - event
        # records cost ~8 us per iteration, 3 % of a 0.3 ms iteration.
        op.set_profiling(kind == "sell", what=1)
        fence()
        t0 = time.perf_counter()
        r = op.solve_run(steps)                  # exactly K PCG iterations
        fence()
        elapsed_local = time.perf_counter() - t0
        assert r.iters_done == warmup + steps and r.status == 4, \
            f"solve ended inside the timed window (iters_done={r.iters_done}); use fewer steps"
        r1 = r
        op.set_profiling(True, what=3)                # second window: events around operator AND vector-phase launches
        fence()
        r = op.solve_run(steps)
        fence()
        op.set_profiling(False)
        if r.status != 4:
            raise RuntimeError("solve ended inside the instrumented window; use fewer steps")
        src = r1 if kind == "sell" else r
        op_ms = max(src.spmv_ms_sum / max(1, src.spmv_count), 1e-9)
        n_op = int(src.spmv_count)
        vec = None
        if r.vec_count > 0:                           # the vector phase (k_vec): HIP events around its launches (second window)
            fused = world == 1 and os.environ.get("PCG_VEC_FUSED", "1") != "0"
            vb = (73.0 if fused else 57.0) * op.n       # p, q, r, x, M^-1 in + flags + r', x' out (57 B/dof) [+ p in, p' out: 16 B/dof]
            vms = r.vec_ms_sum / r.vec_count
            vec = {"kernel": "k_vec<fused>: alpha, r/x update + five sums, grid-wide reduction, beta, p' - ONE launch per iteration" if fused else
                             "k_vec<split>: alpha, r/x update + partial sums (then k_reduce, the all-reduce and k_update_p)",
                   "avg_launch_ms": vms, "launches_timed": int(r.vec_count), "bytes_per_launch": vb,
                   "bytes_definition": ("73 B/dof: p, q, r, x, M^-1 read + 1 flag byte + r', x' written (57), then p read and p' written (16); "
                                        "z = M^-1 r' stays in registers across the grid barrier" if fused else
                                        "57 B/dof: p, q, r, x, M^-1 read + 1 flag byte + r', x' written"),
                   "bound": "hbm", "achieved": vb / (vms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": vb / (vms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                   "measured_in": "a second window of K iterations with events around every launch"}
        elapsed, per_rank = gather_max(elapsed_local)
        iter_bytes = gather_sum(op.operator_cost()[0] + 73.0 * op.n)
        comm_info = None
        if world > 1 and getattr(comm, "native", False):   # second window: HIP events around the exchange wait / all-reduces
            s0 = comm.stats()
            comm.set_timing(True)
            fence()
            t0 = time.perf_counter()
            r2 = op.solve_run(steps)
            fence()
            t_win = time.perf_counter() - t0
            comm.set_timing(False)
            s1 = comm.stats()
            if r2.status == 4:
                k = max(1, steps)
                comm_info = {"halo_wait_ms_per_iter": (s1["halo_wait_ms"] - s0["halo_wait_ms"]) / k,
                             "allreduce_ms_per_iter": (s1["allreduce_ms"] - s0["allreduce_ms"]) / k,
                             "exchanges_per_iter": (s1["n_halo"] - s0["n_halo"]) / k,
                             "allreduces_per_iter": (s1["n_allreduce"] - s0["n_allreduce"]) / k,
                             "ms_per_step_with_timers": t_win / k * 1e3}
        final = None
        if not args.no_finish:                        # not timed: convergence evidence
            t0 = time.perf_counter()
            op.solve_run(-1)
            x, res = op.solve_end()
            final = {"flag": int(res.flag), "iter": int(res.iter), "relres": float(res.relres),
                     "n_matvec": int(res.n_matvec), "solve_s": time.perf_counter() - t0 + elapsed}
        else:
            op.solve_end()
        standalone = None
        if world == 1:
            ms = op.bench_spmv(10, standalone_reps)
            standalone = {"min_ms": float(ms.min()), "median_ms": float(np.median(ms))}
        return {"op": op, "elapsed": elapsed, "per_rank_s": per_rank, "op_ms": op_ms, "n_op": n_op, "final": final,
                "standalone": standalone, "t_setup": t_setup, "comm": comm_info, "vec": vec, "iter_bytes": iter_bytes}


    def mf_roofline(mm):
        """The matrix-free operator's apply (all its launches) against both roofs."""
        eb, ef = mm["op"].operator_cost()
        t_op = mm["op_ms"] * 1e-3
        return {"kernel": "k_ebe_hexs (one 8-node pattern type) or k_ebe_mixed / k_ebe_mtile (several types) + k_ebe_shared = one operator apply",
                "avg_apply_ms": mm["op_ms"], "avg_launch_ms": mm["op_ms"], "launches_timed": mm["n_op"],
                "flops_per_apply": ef, "achieved_TFLOPs": ef / t_op / 1e12, "peak_TFLOPs": F64_PEAK_TFLOPS, "frac_flops": ef / t_op / 1e12 / F64_PEAK_TFLOPS,
                "bytes_per_apply": eb, "bytes_per_launch": eb, "achieved_GBps": eb / t_op / 1e9, "peak_GBps": HBM_PEAK_GBS, "frac_hbm": eb / t_op / 1e9 / HBM_PEAK_GBS,
                "bound": "hbm", "achieved": eb / t_op / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": eb / t_op / 1e9 / HBM_PEAK_GBS, "traffic": None,
                "bound_note": "neither roof saturated: latency / LDS-phase bound (DESIGN.md section 4)"}

    def mf_object(mm):
        oi = mm["op"].operator_info()
        return {"note": "SURVEY 8(f)-1: the reference's element-by-element operator kept matrix-free; same PCG driver, same inputs",
                "value": args.steps / mm["elapsed"], "unit": "iterations/s", "ms_per_step": mm["elapsed"] / args.steps * 1e3,
                "operator_avg_ms": mm["op_ms"], "operator_launches_timed": mm["n_op"], "n_elem": oi["n_elem"], "n_chunks": oi["n_chunks"],
                "standalone_operator": mm["standalone"], "solve": mm["final"], "comm": mm["comm"], "vector_phase": mm["vec"],
                "roofline_iteration": iteration_roofline(mm, args.steps), "roofline": mf_roofline(mm), "setup_s": mm["t_setup"]}

    # =================================================================================================================
    # 1. the headline window - the ONLY thing this run must deliver
    # =================================================================================================================
    box = box_identity(dev) if rank == 0 else None
    head_kind = "sell" if args.operator in ("both", "sell") else args.operator
    m = measure(head_kind)
    op = m["op"]
    n_loc, nnz_loc = op.n, op.nnz
    setup_s = {head_kind: round(m["t_setup"], 2)}
    stream = None
    if rank == 0:                                    # what THIS box's HBM delivers to a plain stream kernel on the engine's stream
        stream = {"read_GBps": op.bench_hbm(8 << 30, "read", 10), "copy_GBps": op.bench_hbm(1 << 30, "copy"),
                  "note": "pcg_bench_hbm: 16 B/lane non-temporal grid-stride kernels over 8 GiB (read, 8 loads in flight per lane) / 1 + 1 GiB (copy)"}
    out = {
        "metric": METRIC, "value": args.steps / m["elapsed"], "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": m["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{wl_name}, Jacobi-PCG Tol 1e-7, {world} part(s) {grid[0]}x{grid[1]}x{grid[2]}",
                   "dofs": brick.n_dof, "nnz": brick.nnz if brick.nnz is not None else (op.nnz if world == 1 else None), "parts": world,
                   "operator": {"sell": "assembled SELL-BSR3", "ebe": "matrix-free (EBE)", "dict": "assembled SELL-BSR3, value dictionary"}[head_kind]},
        "solve": m["final"], "roofline_iteration": iteration_roofline(m, args.steps), "roofline_vector_phase": m["vec"],
        "box": dict(box or {}, hbm_stream=stream), "skipped": [], "errors": {},
    }
    sell_bytes = None
    if head_kind in ("sell", "dict"):
        info = op.matrix_info()
        sell_bytes, sell_flops = op.operator_cost()
        t_k = m["op_ms"] * 1e-3
        achieved = sell_bytes / t_k / 1e9
        alg_bytes = 12.0 * nnz_loc + 20.0 * n_loc                 # what scalar CSR (f64 value + i32 column per nnz) would move
        col_bytes = 2 if (sell_bytes - 16.0 * n_loc) / info["stored_blocks"] - 72.0 < 3.0 else 4     # 16-bit column offsets, or i32 (+ the per-row bytes of an overflow part)
        out["config"]["format"] = f"SELL-{info['slice_rows']} over 3x3 blocks, {8 * col_bytes}-bit block columns"
        out["config"]["stored_over_true_blocks"] = info["stored_blocks"] / max(1, info["nnzb"])
        if args.workload == "octree":
            out["config"]["format"] += "; rows longer than their slice's base width continue in an overflow part (k_spmv_ovf; roofline.avg_launch_ms covers both launches)"
        out["config"]["spmv_achieved_GBps"] = achieved
        # round 6: one apply = L launches of k_spmv (y written at the end of each, kernels_spmv.hpp HOLD): the roofline is quoted per LAUNCH -
        # bytes / L over apply time / L - so that a profiler's per-kernel average is the same quantity
        tuning = op.tuning_info()                       # (decided by the engine at the first solve: measured, same bits either way)
        L = tuning["spmv_launches_per_apply"] if head_kind == "sell" and args.workload == "brick" else 1
        kname = "k_spmv_dict" if head_kind == "dict" else f"k_spmv<{info['slice_rows'] // 64},true,{'true' if col_bytes == 2 else 'false'},false,{'true' if L > 1 else 'false'}>"
        out["roofline"] = {
            "bound": "hbm", "kernel": kname + " SELL-BSR3 SpMV + fused p.Ap" + (", this rank's part" if world > 1 else ""),
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "bytes_per_launch": sell_bytes / L, "launches_per_apply": L, "bytes_per_apply": sell_bytes, "avg_apply_ms": m["op_ms"], "tuning": tuning,
            "bytes_definition": f"stored operator: {72 + col_bytes} B per stored 3x3 block (72 B values + one {8 * col_bytes}-bit column"
                                + (" offset from the slice's base column" if col_bytes == 2 else "") + ") + x in + y out (16 B/dof) + slice "
                                "pointers - the algorithmic traffic of the block format (pcg_operator_cost)",
            "avg_launch_ms": m["op_ms"] / L, "launches_timed": m["n_op"] * L, "traffic": None,
            "hbm_stream_this_box": stream, "frac_of_stream_read": achieved / stream["read_GBps"] if stream else None,
            "csr_equivalent_bytes": alg_bytes, "csr_equivalent_GBps": alg_bytes / t_k / 1e9,
            "csr_equivalent_note": "SURVEY 8(d) formula 12 nnz + 20 n: a scalar-CSR kernel's traffic for the same product; NOT what this "
                                   "kernel moves (it can exceed the HBM peak) - kept for comparison with CSR codes only; the literal CSR kernel is scalar_csr_same_run",
            "standalone_spmv": m["standalone"]}
        if rank == 0:
            log(f"{args.workload} N={N}: {brick.n_dof} dof, nnz {out['config']['nnz']}; parts {world} grid {grid}; local dof {op.n}, local nnz {op.nnz}; "
                f"RefMeshPart {t_parts:.1f}s, assemble+upload {m['t_setup']:.1f}s; SELL slices {info['n_slices']} x {info['slice_rows']} rows, "
                f"padding {info['stored_blocks'] / info['nnzb'] - 1:.2%}")
    else:
        out["roofline"] = mf_roofline(m)
        out["matrix_free_info"] = op.operator_info()
    if world > 1:
        out["comm"] = {"transport": transport, "ranks": comm.world, "per_rank_ms_per_step": [t / args.steps * 1e3 for t in m["per_rank_s"]]}
        if os.environ.get("PCG_BENCH_NATIVE_ERROR"):
            out["comm"]["native_error"] = os.environ["PCG_BENCH_NATIVE_ERROR"]
        if m["comm"]:
            out["comm"].update(m["comm"])
    if rank == 0:
        log(f"headline [{head_kind}]: {out['value']:.1f} it/s, {out['ms_per_step']:.4f} ms per step, operator {m['op_ms']:.4f} ms"
            + (f" = {out['roofline']['frac']:.3f} of {HBM_PEAK_GBS:.0f} GB/s" if out.get("roofline") else "") + f", solve {m['final']}")

    # =================================================================================================================
    # 2. everything below is OPTIONAL: each object guarded by try / except and by the wall-clock budget; at N > 1 every object is
    #    collective, rank 0 decides (Budget.go) and a watchdog on rank 0 prints the headline it has if the extras stall - the first
    #    contact with a real multi-GPU node must not lose the line.
    # =================================================================================================================
    budget = Budget(args.extras_budget_s, world, rank, dist if world > 1 else None)
    extras_guard = None
    if world > 1 and rank == 0:
        import threading
        deadline = float(os.environ.get("PCG_BENCH_EXTRAS_DEADLINE_S", "480"))     # (the extras budget is 150 s + the CPU leg; the launcher gives the ranks 900 s)

        def bail():
            out["extras"] = f"the optional objects did not finish within {deadline:.0f} s (PCG_BENCH_EXTRAS_DEADLINE_S): headline only, printed by the watchdog; the job was ended"
            out["extras_file"] = write_extras(out)
            print(compact_line(out), flush=True)
            os._exit(0)
        extras_guard = threading.Timer(deadline, bail)
        extras_guard.daemon = True
        extras_guard.start()
    want_cpu = not args.no_cpu_baseline
    cpu_reserve = lambda: 50.0 if (want_cpu and world > 1 and "cpu_baseline" not in out) else 0.0     # noqa: E731 - N > 1: the CPU leg runs last
    size_scale = max(1.0, brick.n_dof / 1.0e7)

    def optional(name, est_s, fn):
        if not budget.go(name, est_s + cpu_reserve()):
            out["skipped"].append(name)
            return None
        t0 = time.perf_counter()
        try:
            r = fn()
            if rank == 0:
                log(f"`{name}`: {time.perf_counter() - t0:.1f} s")
            return r
        except Exception as ex:          # noqa: BLE001 - the headline must survive (at N > 1 a failure here is the same on every rank)
            log(f"[rank {rank}] `{name}` failed: {ex!r}")
            out["errors"][name] = repr(ex)[:300]
            return None

    # ---- the literal "CSR SpMV" of the metric at THIS size: device-side scalar copy of the headline matrix
    if rank == 0 and world == 1 and head_kind == "sell" and args.workload == "brick" and not args.no_finish:
        sp = optional("scalar_csr", 6 * size_scale, lambda: scalar_csr_point_device(op))
        if sp:
            out["roofline"]["scalar_csr_same_run"] = sp
            log(f"scalar-CSR copy: {sp['nnz']} nnz, {sp['median_launch_ms']:.4f} ms = {sp['frac_of_peak']:.3f} of peak on 12 nnz + 20 n")
    op.close()

    # ---- N = 1: the CPU baseline right after the headline (N > 1: last, while the other ranks sleep on the store)
    def cpu_leg():
        log("timing the CPU baseline (the reference's NumPy arithmetic: one core at this size, then R processes x 1 thread) ...")
        return cpu_baseline(part, N if args.workload == "brick" else f"octree:{args.octree_size}", args.cpu_ranks, args.workload,
                            quick=not args.full, total_dofs=brick.n_dof)
    if want_cpu and world == 1:
        cb = optional("cpu_baseline", 45, cpu_leg)
        out["cpu_baseline"] = cb

    # ---- the matrix-free operator on the same system
    e = None
    if args.operator == "both":
        e = optional("matrix_free", 12 * size_scale, lambda: measure("ebe"))
        if e is not None:
            out["matrix_free"] = mf_object(e)
            setup_s["ebe"] = round(e["t_setup"], 2)
            e["op"].close()

    # ---- HBM traffic of THIS run's kernels: two rocprofv3 --pmc passes of a short child run (N = 1)
    import shutil
    profiled = any(k.startswith(("ROCPROF", "ROCP_", "ROCTX")) for k in os.environ)       # already under a profiler: no nested passes
    if world == 1 and not args.no_pmc_traffic and (args.pmc_traffic or (shutil.which("rocprofv3") and not profiled)):
        wl = args.workload
        segs = [f"{wl}:{head_kind}"] + ([f"{wl}:ebe"] if e is not None else [])
        if args.full and wl == "brick" and not args.no_octree:
            segs += ["octree:sell", "octree:ebe"]
        pmc_live = optional("pmc_traffic", (25 + 12 * len(segs)) * size_scale, lambda: pmc_traffic_live(args, segs))

        def pmc_note(r):
            return (f"measured on THIS box by two rocprofv3 --kernel-trace --pmc passes of a short run of this command (FETCH_SIZE {r['FETCH_SIZE_KB_raw']:.0f} KB "
                    f"x2 gfx950 correction + WRITE_SIZE {r['WRITE_SIZE_KB_raw']:.0f} KB per apply, mean of {r['applies']} applies; per kernel: "
                    + ", ".join(f"{k} {v / 1e6:.1f} MB" for k, v in r["kernels"].items()) + ")")
        if pmc_live:
            out["pmc_traffic"] = pmc_live
            live = pmc_live.get(f"{wl}:{head_kind}")
            if live and out.get("roofline"):
                Lr = out["roofline"].get("launches_per_apply", 1)          # (the PMC passes sum an apply's launches)
                out["roofline"].update(traffic=live["bytes"] / Lr, traffic_per_apply=live["bytes"], traffic_note=pmc_note(live),
                                       traffic_over_bytes=live["bytes"] / Lr / out["roofline"]["bytes_per_launch"])
                if live.get("vec") and out.get("roofline_vector_phase"):
                    v = out["roofline_vector_phase"]
                    v.update(traffic=live["vec"]["bytes"], traffic_over_bytes=live["vec"]["bytes"] / v["bytes_per_launch"])
            live = pmc_live.get(f"{wl}:ebe")
            if live and isinstance(out.get("matrix_free"), dict):
                r = out["matrix_free"]["roofline"]
                r.update(traffic=live["bytes"], traffic_over_bytes=live["bytes"] / r["bytes_per_apply"], traffic_note=pmc_note(live))
    if out.get("roofline") and out["roofline"].get("traffic") is None and head_kind == "sell" and world == 1 and args.workload == "brick":
        try:        # no live passes: the committed passes of an identical launch (profiles/pmc_traffic.json), marked as such in the full record
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(f"N{N}_rpl{info['slice_rows'] // 64}" + ("_col16" if col_bytes == 2 else ""))
            if pmc:
                out["roofline"].update(traffic=pmc["traffic_bytes_per_launch"] / L, traffic_over_bytes=pmc["traffic_bytes_per_launch"] / sell_bytes,
                                       traffic_note="from profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same launch "
                                                    "(gfx950-corrected), collected on another box in another session - NOT a measurement of this run")
        except OSError:
            pass

    # ---- north_star: "PCG-iterations/sec on a synthetic 3D elasticity octree mesh ... at 1/2/4/8 GPUs": the 10 M-dof graded octree mesh,
    # split into one part per rank by recursive bisection (the METIS stand-in), assembled and matrix-free, at EVERY N
    opart10 = None

    def octree10_object():
        nonlocal opart10
        from pcg_mi355x.octree import GradedOctreeMesh, make_octree_parts, bisect_elements
        t0 = time.perf_counter()
        roots10 = tuple(int(v) for v in os.environ.get("PCG_BENCH_OCTREE10_ROOTS", "38,38,38").split(","))     # (tests shrink the mesh)
        mesh10 = GradedOctreeMesh(roots10, 4, band=1.2, seed=0, symmetry=True)
        opart10 = make_octree_parts(mesh10, world, elem_part=bisect_elements(mesh10, world) if world > 1 else None, only=[rank])[0]
        o = {"workload": f"multi-level 2:1-balanced octree mesh around a sphere (GradedOctreeMesh({roots10}, levels=4, band=1.2, symmetry=True)), "
                         f"Jacobi-PCG Tol 1e-7, {world} part(s) by recursive bisection of the element centroids, one per GPU",
             "mesh": mesh10.summary(), "dofs": int(mesh10.n_dof), "parts": world, "mesh_setup_s": time.perf_counter() - t0,
             "steps": args.steps, "warmup": args.warmup, "local_dofs_this_rank": int(opart10["NDOF"]),
             "interface_dofs_this_rank": int(sum(len(v) for v in opart10["OvrlpLocalDofVecList"]))}
        del mesh10
        for kind, key in (("sell", "assembled"), ("ebe", "matrix_free")):
            mm = measure(kind, opart10, standalone_reps=20)
            by, fl = mm["op"].operator_cost()
            t_op = mm["op_ms"] * 1e-3
            ent = {"value": args.steps / mm["elapsed"], "unit": "iterations/s", "ms_per_step": mm["elapsed"] / args.steps * 1e3,
                   "per_rank_ms_per_step": [t / args.steps * 1e3 for t in mm["per_rank_s"]], "operator_avg_ms": mm["op_ms"],
                   "vector_phase_ms": mm["vec"]["avg_launch_ms"] if mm["vec"] else None, "solve": mm["final"], "setup_s": mm["t_setup"],
                   "comm": mm["comm"], "roofline_iteration": iteration_roofline(mm, args.steps),
                   "roofline": {"bound": "hbm", "scope": "this rank's operator apply (all its launches)", "bytes_per_apply": by, "avg_apply_ms": mm["op_ms"],
                                "achieved": by / t_op / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / t_op / 1e9 / HBM_PEAK_GBS,
                                "flops_per_apply": fl, "frac_flops": fl / t_op / 1e12 / F64_PEAK_TFLOPS, "traffic": None}}
            if kind == "ebe":
                ent["operator_info"] = mm["op"].operator_info()
            mm["op"].close()
            o[key] = ent
            if rank == 0:
                log(f"[octree 10 M dof, {world} part(s), {kind}] {ent['value']:.0f} it/s, operator {ent['operator_avg_ms']:.4f} ms, solve {ent['solve']}")
        opart10.pop("_pcg_mi355x_operator", None)
        return o
    do_octree = args.workload == "brick" and args.operator == "both" and not args.no_octree and not args.no_finish
    if do_octree:
        o10 = optional("octree_10m", 45, octree10_object)
        if o10 is not None:
            out["octree_10m"] = o10
    # ---- BASELINE configs[1]: the 1 M-dof octree mesh (N = 1)
    if do_octree and world == 1:
        o1 = optional("octree", 150 if args.full else 20,
                      lambda: octree_object(measure, log, with_cpu=want_cpu and args.full, cpu_ranks=args.cpu_ranks, iteration_roofline=iteration_roofline, full=args.full))
        if o1 is not None:
            out["octree"] = o1
            pl = out.get("pmc_traffic") or {}
            for key, seg in (("assembled", "octree:sell"), ("matrix_free", "octree:ebe")):
                if seg in pl and key in o1:
                    o1[key]["roofline"].update(traffic=pl[seg]["bytes"], traffic_over_bytes=pl[seg]["bytes"] / o1[key]["roofline"]["bytes_per_apply"])

    # ---- --full: the SAME assembled matrix with its values stored as 16-bit indices into the table of its distinct 3x3 blocks (frozen format)
    def dict_object():
        d = measure("dict")
        db, dfl = d["op"].operator_cost()
        nu = d["op"].matrix_dictionary()
        t_op = d["op_ms"] * 1e-3
        o = {"note": "the same assembled matrix, values replaced by a dictionary of its distinct 3x3 blocks held in LDS (lossless: the SpMV is "
                     "bit-identical to the plain format); applies when the matrix has <= 65535 distinct blocks" + ("" if nu else " - THIS matrix has too many: measured in the plain format"),
             "distinct_blocks": nu, "table": d["op"].matrix_dictionary_info(), "value": args.steps / d["elapsed"], "unit": "iterations/s",
             "ms_per_step": d["elapsed"] / args.steps * 1e3, "operator_avg_ms": d["op_ms"], "operator_launches_timed": d["n_op"],
             "standalone_spmv": d["standalone"], "solve": d["final"], "comm": d["comm"], "vector_phase": d["vec"],
             "roofline_iteration": iteration_roofline(d, args.steps),
             "roofline": {"kernel": "k_spmv_dict (SELL-64, 16-bit block index + column per stored block, table in LDS)",
                          "avg_launch_ms": d["op_ms"], "bytes_per_launch": db, "achieved_GBps": db / t_op / 1e9, "peak_GBps": HBM_PEAK_GBS,
                          "frac_hbm": db / t_op / 1e9 / HBM_PEAK_GBS, "flops_per_launch": dfl, "frac_flops": dfl / t_op / 1e12 / F64_PEAK_TFLOPS,
                          "bound": "LDS reads of the table (72 B per lane per block) + x gathers; neither HBM nor FMA saturated"}}
        setup_s["dict"] = round(d["t_setup"], 2)
        d["op"].close()
        return o
    if args.full and args.operator == "both":
        dd = optional("assembled_dictionary", 15 * size_scale, dict_object)
        if dd is not None:
            out["assembled_dictionary"] = dd
    if args.full and want_cpu and world == 1 and isinstance(out.get("octree_10m"), dict) and "PCG_BENCH_OCTREE10_ROOTS" not in os.environ:
        oc = optional("octree_10m_cpu", 60, lambda: cpu_baseline(opart10, "octree:10m", min(args.cpu_ranks or 16, 16), "octree", quick=True, total_dofs=out["octree_10m"]["dofs"]))
        if oc is not None:           # at most 16 processes: every one builds the 10 M-dof mesh for itself (3.6 GB at its peak)
            out["octree_10m"]["cpu_baseline"] = oc

    # ---- --ab-engine-side (N > 1, opt-in; FROZEN forms): the brick windows with the mailbox all-reduce, then with the
    # direct exchange on top.  Collective: every rank takes the same branch (enable_mailbox agrees on the outcome); the forms are
    # switched off again whatever happens (a failed window must not leave the mailboxes on for the later collectives).
    def engine_side_ab():
        ab = {"enabled": False}
        try:
            if not comm.enable_mailbox(True):
                return {"enabled": False, "reason": comm.mailbox_reason}
            ab = {"enabled": True, "note": "all-reduces through peer-mapped mailboxes (rank-order sums) inside k_fixup / k_vec; exchange unchanged"}
            for kind, key in (("sell", "assembled"), ("ebe", "matrix_free")):
                mm = measure(kind)
                ab[key] = {"value": args.steps / mm["elapsed"], "unit": "iterations/s", "ms_per_step": mm["elapsed"] / args.steps * 1e3,
                           "per_rank_ms_per_step": [t / args.steps * 1e3 for t in mm["per_rank_s"]], "solve": mm["final"], "comm": mm["comm"]}
                if isinstance(ab[key]["comm"], dict):
                    ab[key]["comm"]["allreduce_ms_per_iter"] = None      # fused into k_fixup / k_vec: not measurable by the events around Comm::allreduce
                mm["op"].close()
            os.environ["PCG_DIRECT_EXCHANGE"] = "1"       # every operator built from here on enables the direct exchange at set_comm
            direct = {"note": "interface exchange as stores into peer-mapped receive buffers (k_halo_put + arrival words) AND mailbox all-reduces; "
                              "matrix-free engine with one phase (one element launch)"}
            for kind, key in (("sell", "assembled"), ("ebe", "matrix_free")):
                mm = measure(kind)
                direct[key] = {"value": args.steps / mm["elapsed"], "unit": "iterations/s", "ms_per_step": mm["elapsed"] / args.steps * 1e3,
                               "per_rank_ms_per_step": [t / args.steps * 1e3 for t in mm["per_rank_s"]], "solve": mm["final"],
                               "enabled": bool(mm["op"].direct_exchange), "reason": mm["op"].direct_exchange_reason}
                mm["op"].close()
            ab["direct_exchange"] = direct
            return ab
        finally:
            os.environ.pop("PCG_DIRECT_EXCHANGE", None)
            try:
                comm.enable_mailbox(False)
            except Exception as ex:      # noqa: BLE001
                log(f"[rank {rank}] enable_mailbox(False): {ex!r}")
    if args.ab_engine_side and world > 1 and getattr(comm, "native", False) and args.operator == "both":
        ab = optional("engine_side_ab", 60 * size_scale, engine_side_ab)
        if ab is not None:
            out["comm"]["engine_side_ab"] = ab

    def shutdown():
        part.pop("_pcg_mi355x_operator", None)
        if world > 1:
            dist.barrier()
            if hasattr(comm, "close"):
                comm.close()                 # ncclCommDestroy while the HIP runtime is still up, not at interpreter exit
            dist.destroy_process_group()

    if world > 1:                                   # every rank's host set-up (its part + its operator), for the record
        import resource
        mine = {"rank": rank, "refmeshpart_s": round(t_parts, 2), "host_max_rss_GB": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 2),
                "operator_s": setup_s}
        setup_all = [None] * world
        dist.all_gather_object(setup_all, mine)
        out["comm"]["per_rank_setup_s"] = setup_all
    CPU_DONE_KEY = "pcg_bench_rank0_done"
    if rank != 0:
        # rank 0 now times the CPU baseline on the node's host cores: sleep on the rendezvous store (a socket wait) instead of
        # spinning in a barrier next to the processes being timed
        try:
            import datetime
            from torch.distributed.distributed_c10d import _get_default_store
            _get_default_store().wait([CPU_DONE_KEY], datetime.timedelta(seconds=1500))
        except Exception as ex:          # noqa: BLE001 - fall through to the barrier
            log(f"[rank {rank}] store wait: {ex!r}")
        shutdown()
        return
    if want_cpu and world > 1:
        budget.world = 1                    # (a local decision from here on: the other ranks are asleep)
        out["cpu_baseline"] = optional("cpu_baseline", 45, cpu_leg)
    out["skipped"] = sorted(set(out["skipped"]) | set(budget.skipped))
    out["wall_s"] = time.perf_counter() - t_wall0
    if extras_guard is not None:
        extras_guard.cancel()
    out["extras_file"] = write_extras(out)
    print(compact_line(out), flush=True)
    if world > 1:
        try:
            from torch.distributed.distributed_c10d import _get_default_store
            _get_default_store().set(CPU_DONE_KEY, "1")
        except Exception as ex:          # noqa: BLE001
            log(f"store set: {ex!r}")
    shutdown()


if __name__ == "__main__":
    main()
