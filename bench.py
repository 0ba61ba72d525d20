#!/usr/bin/env python
"""bench.py - PCG iterations/sec + SpMV achieved HBM GB/s on the BASELINE.json workload.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE full PCG iteration of the reference algorithm (src/solver/pcg_solver.py:438-562:
operator apply + interface exchange, the weighted dots, the vector updates, the status read-back)
on the synthetic 10M-DOF elasticity brick of SURVEY.md 8(d) (N=150 nodes per side, n=10 125 000,
nnz=809 238 528), with every input already resident in HBM.  W warm-up iterations, then exactly K
timed iterations between barrier+synchronize fences; the max over ranks is reported.  N>1 runs the
SAME 10M system split into N parts, one part per GPU (strong scaling, BASELINE configs[3]).

Extra objects on the JSON line:
  roofline     - the SpMV kernel (dominant): ALGORITHMIC bytes 12*nnz + 20*n (the stored
                 SELL-BSR3 format moves fewer bytes, reported as impl_*) / mean kernel time measured
                 with HIP events on the engine stream inside the timed region; peak 8 TB/s.
  cpu_baseline - the oracle (C port of the reference's EBE mat-vec + NumPy vector ops, 1 thread, as
                 the reference pins its BLAS) timed on a bounded sample of the same system.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "pcg-mpi-solver_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(k, "1")          # the reference's mode (pcg_solver.py:10-15); set before NumPy loads
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def cpu_baseline(part, budget_s=20.0):
    """Reference algorithm on the host: oracle (kind 'port'), 1 rank x 1 thread, bounded sample."""
    import copy
    import subprocess
    import pcg_oracle
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    P = {k: v for k, v in part.items() if not k.startswith("_pcg_mi355x")}
    P["GlobData"] = copy.deepcopy(part["GlobData"])
    P["Un"] = np.zeros(P["NDOF"])
    t0 = time.perf_counter()
    pcg_oracle.update_bc([P], use_c=True)                 # one mat-vec: calibrates the sample size
    t_mv = time.perf_counter() - t0
    pcg_oracle.update_preconditioner([P])
    m = int(max(3, min(50, budget_s / max(t_mv * 1.25, 1e-3))))
    P["GlobData"]["MaxIter"] = m
    t0 = time.perf_counter()
    out = pcg_oracle.pcg([P], use_c=True, record=False)
    t = time.perf_counter() - t0
    return {"value": m / t, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": f"first {m} PCG iterations of the same system (MaxIter={m}, {out['n_matvec']} EBE mat-vecs incl. "
                      f"initial and final residual), oracle/pcg_oracle.py + oracle/ebe_matvec.c, 1 thread",
            "matvec_ms": t_mv * 1e3, "host_cpu": _cpu_model(), "host_cores_available": os.cpu_count()}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nodes-per-side", type=int, default=int(os.environ.get("PCG_BENCH_N", "150")),
                    help="brick size N (150 -> 10M dof = the metric's configuration; 70 -> 1M; 322 -> 100M)")
    ap.add_argument("--workload", choices=["brick", "octree"], default="brick",
                    help="brick = SURVEY 8(d) uniform brick (the metric's configuration); octree = two-level 2:1 graded mesh with "
                         "hanging-node transition patterns, ~1.2 M dof (BASELINE configs[1] names an octree mesh)")
    ap.add_argument("--rows-per-lane", type=int, default=int(os.environ.get("PCG_ROWS_PER_LANE", "0")))
    ap.add_argument("--operator", choices=["sell", "ebe", "both"], default="both",
                    help="sell = assembled SELL-BSR3 matrix (the headline value/roofline); both = also time the matrix-free operator")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-finish", action="store_true", help="do not run the solve to convergence after the timed window")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    comm = None
    if world > 1:
        import datetime
        # a rank that dies must take the job down in minutes, not after the default 10-minute collective timeout
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=180))

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build_engine()      # no-op when the in-tree build is current (hipcc --offload-arch=gfx950 otherwise)
    if world > 1:
        dist.barrier()
    import pcg_mi355x as pm
    from pcg_mi355x import _lib
    from pcg_mi355x.brick import Brick, make_parts, block_partition, default_grid
    from pcg_mi355x.dist import TorchComm
    _lib.use_library(None)
    assert _lib.backend_name() == "hip-gfx950"
    if world > 1:
        comm = TorchComm(device=torch.device("cuda", local_rank))
    pm.configure(comm=comm, device=local_rank, rows_per_lane=args.rows_per_lane)

    N = args.nodes_per_side
    t0 = time.perf_counter()
    if args.workload == "octree":
        from pcg_mi355x.octree import TwoLevelMesh, make_octree_parts
        brick = TwoLevelMesh(96, 96, 40, 8, seed=0)                 # .n_dof like a Brick; nnz filled in after assembly
        grid = (world, 1, 1)
        part = make_octree_parts(brick, world, axis=0)[rank]
        brick.nnz = None
        wl_name = f"two-level octree mesh 96x96x(40 fine + 8 coarse), hanging-node transition patterns (nd=39), {brick.n_dof} dof"
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

    def measure(kind):
        """Set up the operator of `kind`, run W warm-up + K timed PCG iterations, finish the solve."""
        part.pop("_pcg_mi355x_operator", None)
        pm.configure(comm=comm, device=local_rank, rows_per_lane=args.rows_per_lane, operator=kind)
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
        max_iter = max(int(gd["MaxIter"]), args.warmup + args.steps + 1)
        op.solve_begin(part["Fext"], np.zeros(op.n), inv, float(gd["Tol"]), max_iter, int(gd["GlobNDofEff"]))
        r = op.solve_run(args.warmup)
        assert r.status == 4 and r.iters_done == args.warmup, "solve ended inside the warm-up window"
        op.set_profiling(True)                        # HIP events around every operator launch from here on
        fence()
        t0 = time.perf_counter()
        r = op.solve_run(args.steps)                  # exactly K PCG iterations
        fence()
        elapsed = time.perf_counter() - t0
        assert r.iters_done == args.warmup + args.steps and r.status == 4, \
            f"solve ended inside the timed window (iters_done={r.iters_done}); use fewer steps"
        op_ms = max(r.spmv_ms_sum / max(1, r.spmv_count), 1e-9)
        n_op = int(r.spmv_count)
        op.set_profiling(False)
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
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
            ms = op.bench_spmv(10, 100)
            standalone = {"min_ms": float(ms.min()), "median_ms": float(np.median(ms))}
        return {"op": op, "elapsed": elapsed, "op_ms": op_ms, "n_op": n_op, "final": final, "standalone": standalone,
                "t_setup": t_setup}

    m = measure("sell")
    op, elapsed, spmv_ms, n_spmv, final, standalone = m["op"], m["elapsed"], m["op_ms"], m["n_op"], m["final"], m["standalone"]
    info = op.matrix_info()
    if rank == 0:
        if brick.nnz is None:
            brick.nnz = op.nnz if world == 1 else None
        log(f"{args.workload} N={N}: {brick.n_dof} dof, nnz {brick.nnz}; parts {world} grid {grid}; local dof {op.n}, local nnz {op.nnz}; "
            f"RefMeshPart {t_parts:.1f}s, assemble+upload {m['t_setup']:.1f}s; SELL slices {info['n_slices']} x {info['slice_rows']} rows, "
            f"padding {info['stored_blocks'] / info['nnzb'] - 1:.2%}")
    n_loc, nnz_loc = op.n, op.nnz
    matrix_free = None
    if args.operator in ("both", "ebe"):
        op.close()
        try:
            e = measure("ebe")
        except Exception as ex:              # the headline line must survive a failure of the optional second measurement
            log(f"matrix-free measurement failed: {ex!r}")
            e = None
            matrix_free = {"error": repr(ex)}
    if args.operator in ("both", "ebe") and e is not None:
        oi = e["op"].operator_info()
        matrix_free = {"note": "SURVEY 8(f)-1: the reference's element-by-element operator kept matrix-free (k_ebe, colour-ordered, "
                               "deterministic); same PCG driver, same inputs", "value": args.steps / e["elapsed"],
                       "unit": "iterations/s", "ms_per_step": e["elapsed"] / args.steps * 1e3, "operator_avg_ms": e["op_ms"],
                       "operator_launches_timed": e["n_op"], "colors": oi["n_colors"], "n_elem": oi["n_elem"],
                       "standalone_operator": e["standalone"], "solve": e["final"],
                       "flops_per_apply": 2.0 * 24 * oi["n_slots"], "achieved_TFLOPs": 2.0 * 24 * oi["n_slots"] / (e["op_ms"] * 1e-3) / 1e12,
                       "csr_equivalent_GBps": (12.0 * nnz_loc + 20.0 * n_loc) / (e["op_ms"] * 1e-3) / 1e9}
        e["op"].close()

    if rank != 0:
        if world > 1:
            dist.barrier(); dist.destroy_process_group()
        return

    alg_bytes = 12.0 * nnz_loc + 20.0 * n_loc                     # f64 val + i32 col per nnz; x, y, i32 rowptr
    impl_bytes = info["stored_blocks"] * (72.0 + 4.0) + 16.0 * n_loc + 8.0 * (info["n_slices"] + 1)
    achieved = alg_bytes / (spmv_ms * 1e-3) / 1e9
    iters_per_s = args.steps / elapsed
    iter_bytes = alg_bytes + 176.0 * n_loc                         # SURVEY 8(d) B_iter
    traffic, traffic_src = None, None
    try:        # HBM bytes per SpMV launch from the PMC passes (separate rocprofv3 runs, see profiles/pmc_traffic.json)
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(f"N{N}_rpl{info['slice_rows'] // 64}")
        if pmc and world == 1:
            traffic, traffic_src = pmc["traffic_bytes_per_launch"], "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950-corrected)"
    except OSError:
        pass
    out = {
        "metric": "PCG iterations/sec + SpMV achieved HBM GB/s, 10M-DOF 3D elastostatic CSR",
        "value": iters_per_s, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{wl_name}, Jacobi-PCG Tol 1e-7, {world} part(s) {grid[0]}x{grid[1]}x{grid[2]}",
                   "dofs": brick.n_dof, "nnz": brick.nnz, "parts": world, "format": f"SELL-{info['slice_rows']} over 3x3 blocks",
                   "spmv_achieved_GBps": achieved, "iter_algorithmic_GBps": iter_bytes * world / (elapsed / args.steps) / 1e9},
        "roofline": {"bound": "hbm", "kernel": "k_spmv (SELL-BSR3 SpMV + fused p.Ap)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": spmv_ms, "launches_timed": n_spmv,
                     "impl_bytes_per_launch": impl_bytes, "impl_achieved": impl_bytes / (spmv_ms * 1e-3) / 1e9,
                     "impl_frac": impl_bytes / (spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "standalone_spmv": standalone,
                     "note": "achieved/frac use SURVEY 8(d)'s CSR-algorithmic bytes (12*nnz + 20*n); the stored SELL-BSR3 operator moves "
                             "fewer bytes (impl_*, one i32 column per 3x3 block), so frac can exceed 1; impl_frac and the PMC traffic are "
                             "the physical HBM utilisation"},
        "solve": final,
        "matrix_free": matrix_free,
    }
    if not args.no_cpu_baseline and world == 1:
        log("timing the CPU baseline (oracle port, 1 thread) ...")
        out["cpu_baseline"] = cpu_baseline(part)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
