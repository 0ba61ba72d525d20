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

Objects on the JSON line besides the contract's fields:
  roofline     - dominant kernel k_spmv: PHYSICAL bytes of the stored operator per launch (pcg_operator_cost: 72 B of values
                 + a 4 B column, or a 2 B column offset where the slices allow it, per stored 3x3 block + x in + y out) / mean launch time from HIP events on the engine stream inside the
                 timed region -> achieved GB/s, frac = achieved / 8 TB/s (<= 1 by construction).  The SURVEY 8(d)
                 CSR-equivalent figure (12 nnz + 20 n: what a scalar-CSR kernel would have to move) is reported
                 separately as csr_equivalent_*; `hbm_stream_this_box` is a plain read / copy stream measured in this run
                 on this box (pcg_bench_hbm), the practical ceiling beside the spec.
  matrix_free  - the reference's element-by-element operator on the same system, with its own roofline object
                 (flops vs the 78.6 TF f64 vector peak and bytes vs 8 TB/s).
  comm         - N > 1: transport, ranks seen by RCCL, per-iteration exchange wait / all-reduce time (HIP events,
                 measured in a second, separately timed window), per-rank ms per step.
  cpu_baseline - the oracle (reference algorithm, kind "port") on the node's host cores: R = min(cores, 64) processes,
                 one part and one thread each - the reference's own mode - plus the 1-core figure.  At N > 1 rank 0 times it after
                 the GPU windows while the other ranks sleep on the rendezvous store (no spinning barrier next to the CPU run).
  roofline_iteration - (in the headline, matrix_free and octree objects, every N) the whole ITERATION against the HBM roofline:
                 (stored operator bytes + 73 B/dof of the vector phase, summed over the ranks) / ms_per_step / (N x 8 TB/s).
  octree_10m   - every N: the 10 M-dof graded octree mesh split into N parts by recursive bisection, assembled and matrix-free:
                 the "octree mesh at 1/2/4/8 GPUs" series of BASELINE.json's north_star, with its own CPU baseline.
  box          - GPU clocks / power cap / partition modes of the box the numbers come from.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "pcg-mpi-solver_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(k, "1")          # the reference's mode (pcg_solver.py:10-15); set before NumPy loads
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s HBM3E spec
F64_PEAK_TFLOPS = 78.6      # MI355X_MICROARCH.md: f64 vector = f64 matrix peak


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nodes-per-side", type=int, default=int(os.environ.get("PCG_BENCH_N", "150")),
                    help="brick size N (150 -> 10M dof = the metric's configuration; 70 -> 1M; 322 -> 100M = BASELINE configs[4])")
    ap.add_argument("--workload", choices=["brick", "octree"], default="brick",
                    help="brick = SURVEY 8(d) uniform brick (the metric's configuration); octree = two-level 2:1 graded mesh with "
                         "hanging-node transition patterns, ~1.2 M dof (BASELINE configs[1] names an octree mesh)")
    ap.add_argument("--octree-size", choices=["1m", "10m"], default="1m", help="--workload octree: 1 M dof (BASELINE configs[1]) or 10 M dof")
    ap.add_argument("--no-octree", action="store_true", help="brick workload, N = 1: skip the `octree` object (the 1 M-dof graded octree mesh on all three operators)")
    ap.add_argument("--rows-per-lane", type=int, default=int(os.environ.get("PCG_ROWS_PER_LANE", "0")))
    ap.add_argument("--operator", choices=["sell", "ebe", "dict", "both"], default="both",
                    help="sell = assembled SELL-BSR3 matrix (the headline value/roofline); both = also time the same matrix in the "
                         "value-dictionary format and the matrix-free operator")
    ap.add_argument("--comm", choices=["native", "torch"], default=os.environ.get("PCG_BENCH_COMM", "native"),
                    help="N > 1: native = RCCL calls issued by the engine (default); torch = torch.distributed callbacks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-ranks", type=int, default=0, help="processes of the multi-core CPU baseline (0 = min(cores, 64))")
    ap.add_argument("--no-finish", action="store_true", help="do not run the solve to convergence after the timed window")
    ap.add_argument("--no-pmc-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc passes that measure the HBM traffic of the SpMV launch on this box (N = 1; they run "
                         "by default when rocprofv3 is on PATH and this process is not itself being profiled)")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)       # internal: the workload of one rocprofv3 --pmc pass (pmc_traffic_live)
    ap.add_argument("--pmc-traffic", action="store_true",
                    help="N = 1: measure the HBM traffic of the SpMV launch on THIS box with two extra rocprofv3 --pmc passes of a short "
                         "run of this script (FETCH_SIZE, WRITE_SIZE; +1-2 min) instead of quoting profiles/pmc_traffic.json")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` outside any distributed launch spawns the N ranks itself
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args):
    def run(extra_env):
        env = dict(os.environ)
        env.update(extra_env)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        log("launching", args.gpus, "ranks:", " ".join(cmd[1:8]), "...")
        limit = float(os.environ.get("PCG_BENCH_RANKS_TIMEOUT_S", "900"))     # a hung collective must not eat the caller's whole budget
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, err = p.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            log(f"ranks did not finish within {limit:.0f} s - terminating the launch (process group {p.pid})")
            import signal
            os.killpg(p.pid, signal.SIGTERM)
            try:
                out, err = p.communicate(timeout=30)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                out, err = p.communicate()
            sys.stderr.write(err or "")
            return subprocess.CompletedProcess(cmd, 124, out, (err or "") + f"\n[bench] ranks did not finish within {limit:.0f} s")
        sys.stderr.write(err or "")
        return subprocess.CompletedProcess(cmd, p.returncode, out, err)

    def why(err):
        """The lines of the ranks' stderr that say what failed (RCCL / HIP / engine errors), for the JSON line."""
        keys = ("nccl", "rccl", "pcg_", "hip", "error", "Error", "did not finish")
        hit = [l.strip() for l in (err or "").splitlines() if any(k in l for k in keys) and "Traceback" not in l]
        return " | ".join(hit[-6:])[-900:]
    r = run({})
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    by_watchdog = bool(line) and '"extras": "the optional objects' in line[-1]       # rank 0 printed its headline and ended the job (main: extras_guard)
    if (r.returncode != 0 or not line) and not by_watchdog and args.comm == "native" and os.environ.get("PCG_BENCH_NO_RETRY") != "1":
        # keep the scaling point measurable if the native communicator cannot come up on this node: same kernels, same
        # RCCL, but the collectives are issued through torch.distributed callbacks; the line says which transport ran AND
        # carries what the native run reported (comm.native_error)
        reason = why(r.stderr) or f"exit code {r.returncode}, no diagnostic on stderr"
        log(f"native-communicator run failed (rc {r.returncode}): {reason}; retrying with --comm torch")
        r = run({"PCG_BENCH_COMM": "torch", "PCG_BENCH_NATIVE_FAILED": "1", "PCG_BENCH_NATIVE_ERROR": reason})
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:                                      # nothing ran: the driver still gets a line that says why
        line = [json.dumps({"metric": "PCG iterations/sec + SpMV achieved HBM GB/s, 10M-DOF 3D elastostatic CSR", "value": None, "n_gpus": args.gpus,
                            "error": why(r.stderr) or f"exit code {r.returncode}", "unit": "iterations/s"})]
    if line:
        print(line[-1], flush=True)
    return 0 if by_watchdog else (r.returncode if r.returncode != 0 else (0 if line else 1))


# ---------------------------------------------------------------------------------------------------------------------
# host-side reference timings
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_single(part, budget_s=10.0):
    """Reference algorithm on ONE host core: oracle (kind 'port'), 1 rank x 1 thread, bounded sample."""
    import copy
    import numpy as np
    import pcg_oracle
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    if len(part.get("NbrMPIdVector", ())) > 0:            # N > 1: rank 0's part has neighbours - not a system on its own
        from pcg_mi355x.brick import Brick, make_parts
        part = make_parts(Brick(70, seed=0))[0]
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
    return {"value": m / t, "unit": "iterations/s", "cores": 1,
            "sample": f"first {m} PCG iterations of a {P['NDOF']}-dof system ({out['n_matvec']} EBE mat-vecs), 1 process x 1 thread", "dofs": int(P["NDOF"]),
            "matvec_ms": t_mv * 1e3}


def scipy_csr_spmv_point(n_side=70):
    """SURVEY 8(d), informational: scipy.sparse CSR `A @ x` on one host core for the assembled operator of the 1 M-dof brick
    (the same generator, N = 70: 81 M non-zeros, 1 GB of CSR) - the CPU SpMV reference point in the algorithmic bytes
    12 nnz + 20 n the GPU figure `csr_equivalent_GBps` uses.  Never part of the product path."""
    import numpy as np
    import scipy.sparse as sp
    from pcg_mi355x.brick import Brick, make_parts
    from pcg_mi355x.operator import assemble_bsr3
    b = Brick(n_side, seed=0)
    P = make_parts(b)[0]
    rp, c, v = assemble_bsr3(P["SubDomainData"]["StrucDataList"], b.n_node)       # host code of the engine library (pcg_asm_*)
    A = sp.bsr_matrix((v, c, rp), shape=(b.n_dof, b.n_dof)).tocsr()
    del rp, c, v
    x = np.random.default_rng(0).standard_normal(b.n_dof)
    A @ x
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        A @ x
        t.append(time.perf_counter() - t0)
    ms = float(np.median(t)) * 1e3
    return {"note": "scipy.sparse CSR A @ x, 1 thread, assembled 1 M-dof brick (N = 70) - informational CPU SpMV point",
            "n": int(b.n_dof), "nnz": int(A.nnz), "ms": ms, "GBps_algorithmic": (12.0 * A.nnz + 20.0 * b.n_dof) / (ms * 1e-3) / 1e9}


def numpy_reference_point(part, budget_s=12.0):
    """The reference's OWN arithmetic path on one host core: the NumPy restatement of calcMatVecProd / PCG (pcg_oracle with
    use_c=False - bit-identical to the unmodified pcg_solver.py on every fixture, oracle/make_golden.py), single-threaded BLAS
    like the reference sets it (pcg_solver.py:10-15).  Bounded sample; a part above 4 M dof is replaced by the 1 M-dof brick."""
    import copy
    import numpy as np
    import pcg_oracle
    note = "the bench's own part"
    if part["NDOF"] > 4_000_000 or len(part.get("NbrMPIdVector", ())) > 0:
        from pcg_mi355x.brick import Brick, make_parts
        why = ("the bench's system is too large for a bounded NumPy sample" if part["NDOF"] > 4_000_000 else
               "at N > 1 rank 0 holds one part of the system, and a part with neighbours is not a system on its own")
        part = make_parts(Brick(70, seed=0))[0]
        note = f"1 M-dof brick (N = 70) of the same generator: {why}"
    P = {k: v for k, v in part.items() if not k.startswith("_pcg_mi355x")}
    P["GlobData"] = copy.deepcopy(part["GlobData"])
    P["Un"] = np.zeros(P["NDOF"])
    t0 = time.perf_counter()
    pcg_oracle.update_bc([P], use_c=False)
    t_mv = time.perf_counter() - t0
    pcg_oracle.update_preconditioner([P])
    m = int(max(3, min(50, budget_s / max(t_mv * 1.3, 1e-3))))
    P["GlobData"]["MaxIter"] = m
    t0 = time.perf_counter()
    pcg_oracle.pcg([P], use_c=False, record=False)
    t = time.perf_counter() - t0
    return {"value": m / t, "unit": "iterations/s", "cores": 1, "kind": "reference arithmetic (NumPy restatement, bit-identical to pcg_solver.py)",
            "dofs": int(P["NDOF"]), "matvec_ms": t_mv * 1e3, "sample": f"first {m} PCG iterations, 1 process x 1 thread; {note}"}


def scalar_csr_point(dev, n_side=100):
    """SURVEY 8(d)'s literal "CSR SpMV": the assembled operator of a brick as SCALAR CSR (one f64 value + one i32 column per
    non-zero, pcg_create_csr(block = 1), k_spmv_scalar) - 20 back-to-back launches, GB/s in the formula's own bytes 12 nnz + 20 n,
    which is what this kernel really moves.  N = 100 (3 M dof, 238 M non-zeros): the scalar CSR arrays of the 10 M-dof system
    would be 10 GB of host memory for an informational point."""
    import numpy as np
    import scipy.sparse as sp
    from pcg_mi355x.brick import Brick, make_parts
    from pcg_mi355x.operator import assemble_bsr3, Operator
    b = Brick(n_side, seed=0)
    P = make_parts(b)[0]
    rp, c, v = assemble_bsr3(P["SubDomainData"]["StrucDataList"], b.n_node)
    A = sp.bsr_matrix((v, c, rp), shape=(b.n_dof, b.n_dof)).tocsr()
    del rp, c, v
    op = Operator.from_csr(A.indptr, A.indices, A.data, device=dev, block=1)
    nnz, n = int(A.nnz), int(b.n_dof)
    del A
    ms = op.bench_spmv(5, 20)
    by, _ = op.operator_cost()
    op.close()
    t = float(np.median(ms)) * 1e-3
    return {"kernel": "k_spmv_scalar (SELL-64 over scalar rows: f64 value + i32 column per stored non-zero)", "n": n, "nnz": nnz,
            "median_launch_ms": t * 1e3, "launches": 20, "bytes_12nnz_20n": 12.0 * nnz + 20.0 * n, "stored_bytes": by,
            "GBps_12nnz_20n": (12.0 * nnz + 20.0 * n) / t / 1e9, "frac_of_peak": (12.0 * nnz + 20.0 * n) / t / 1e9 / HBM_PEAK_GBS}


def scalar_csr_point_device(op):
    """SURVEY 8(d)'s literal "CSR SpMV" at the bench's OWN size (round 4): the assembled operator's scalar-CSR copy - one f64 value +
    one i32 column per non-zero, expanded from the 3x3-block format on the device (pcg_create_scalar_copy, no 10 GB host CSR) -
    20 back-to-back launches of k_spmv_scalar, GB/s in the formula's own bytes 12 nnz + 20 n, which is what this kernel moves."""
    import numpy as np
    sc = op.scalar_copy()
    try:
        nnz, n = int(sc.nnz), int(sc.n)
        ms = sc.bench_spmv(5, 20)
        by, _ = sc.operator_cost()
        info = sc.matrix_info()
    finally:
        sc.close()
    t = float(np.median(ms)) * 1e-3
    return {"kernel": "k_spmv_scalar (SELL-64 over scalar rows: f64 value + i32 column per stored non-zero; the operator is the device-side "
                      "scalar copy of the headline matrix, pcg_create_scalar_copy)", "n": n, "nnz": nnz, "stored_nonzeros": int(info["stored_blocks"]),
            "median_launch_ms": t * 1e3, "min_launch_ms": float(ms.min()), "launches": 20, "bytes_12nnz_20n": 12.0 * nnz + 20.0 * n, "stored_bytes": by,
            "GBps_12nnz_20n": (12.0 * nnz + 20.0 * n) / t / 1e9, "frac_of_peak": (12.0 * nnz + 20.0 * n) / t / 1e9 / HBM_PEAK_GBS,
            "GBps_stored": by / t / 1e9, "frac_of_peak_stored": by / t / 1e9 / HBM_PEAK_GBS}


def cpu_baseline(part, N, ranks=0, workload="brick", quick=False, total_dofs=None):
    """The reference's mode on this node: R processes x 1 thread, one part each (oracle/mp_baseline.py), beside 1 core.
    `value` (round 4) = the reference's OWN NumPy arithmetic per rank (pcg_oracle with use_c=False: bit-identical to the unmodified
    pcg_solver.py on every fixture); the C port of the mat-vec - ~2x slower per dof, round 3's `value` - stays as `c_port`.
    N: nodes per side of the brick, or "octree:<size>" (parts by recursive bisection, mp_baseline.worker).  quick: the NumPy
    R-process run only (the `octree` object of the default line)."""
    import mp_baseline
    avail = mp_baseline.available_cores()
    out = {"kind": "port", "unit": "iterations/s", "host_cpu": _cpu_model(), "host_cores_available": avail,
           "arithmetic": "oracle/pcg_oracle.py with the reference's NumPy expressions (use_c=False; bit-identical to src/solver/pcg_solver.py, "
                         "oracle/make_golden.py), BLAS pinned to 1 thread per rank like pcg_solver.py:10-15"}
    single = None
    try:
        out["numpy_reference_path"] = single = numpy_reference_point(part)
    except Exception as ex:      # noqa: BLE001 - the line must survive
        log(f"NumPy reference-path point failed: {ex!r}")
    if not quick:
        try:
            out["single_core_c_port"] = cpu_baseline_single(part)
        except Exception as ex:  # noqa: BLE001
            log(f"single-core C-port point failed: {ex!r}")
        try:
            out["scipy_csr_spmv"] = scipy_csr_spmv_point()
        except Exception as ex:      # noqa: BLE001 - informational only
            log(f"scipy CSR SpMV point failed: {ex!r}")
    R = ranks or min(avail, 64)
    if single is not None:
        out.update(value=single["value"], cores=1, sample=single["sample"])
    if R < 2:
        return out
    extra = ["--octree", N.split(":", 1)[1]] if isinstance(N, str) else ["--nodes-per-side", str(N)]

    def mp_run(numpy_path, iters):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "mp_baseline.py"), "--ranks", str(R), "--iters", str(iters)] + extra +
                           (["--numpy"] if numpy_path else []), capture_output=True, text=True, timeout=600)
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    # sample size: ~12 s of solve at 60 % parallel efficiency, estimated from the one-core rate per dof
    per_dof_s = 1.0 / (single["value"] * single["dofs"]) if single else 4.5e-8
    est = 0.6 * R / (per_dof_s * float(total_dofs or part["NDOF"]))       # (at N > 1 `part` is rank 0's part of the system)
    iters = int(max(10, min(400, 12.0 * est)))
    try:
        mp = mp_run(True, iters)
        what = f"{R} parts ({mp['grid'] if isinstance(mp['grid'], str) else 'x'.join(map(str, mp['grid'])) + ' blocks'})"
        out.update(value=mp["value"], cores=R,
                   sample=f"first {iters} PCG iterations of the same system split into {what}, {R} processes x 1 thread = the reference's "
                          f"one-part-per-rank mode (oracle/mp_baseline.py --numpy: pcg_oracle.py per rank with the reference's NumPy mat-vec, "
                          f"shared-memory exchange)",
                   calc_s_mean=mp["calc_s_mean"], comm_wait_s_mean=mp["comm_wait_s_mean"], t_solve_s=mp["t_solve_s"],
                   dofs_per_rank_max=mp["dofs_per_rank_max"])
    except Exception as ex:      # noqa: BLE001 - the GPU line must survive a failure of the CPU side measurement
        log(f"multi-process CPU baseline failed: {ex!r}")
        out["multi_core_error"] = repr(ex)
    if not quick:
        try:
            mp = mp_run(False, int(max(10, min(200, iters // 2))))
            out["c_port"] = {"value": mp["value"], "cores": R, "note": "same R-process run with the C port of the EBE mat-vec (oracle/ebe_matvec.c); "
                             "round 3 quoted this as cpu_baseline.value", "t_solve_s": mp["t_solve_s"], "iterations": mp["iterations"]}
        except Exception as ex:  # noqa: BLE001
            log(f"multi-process C-port run failed: {ex!r}")
    return out


PMC_MARKER = "k_stream_copy"        # pcg_bench_hbm(mode copy): the launches that bracket a segment of the PMC child run


def pmc_child(args):
    """The workload of one rocprofv3 --pmc pass (pmc_traffic_live): for every segment `workload:operator` build the operator, then
    marker launches / 3 + K PCG iterations / marker launches - the parent finds the segment's dispatches between the two marker runs."""
    import numpy as np
    import pcg_mi355x as pm
    from pcg_mi355x import _lib
    from pcg_mi355x.brick import Brick, make_parts
    _lib.use_library(None)
    parts = {}
    for seg in args.pmc_child.split(","):
        wl, kind = seg.split(":")
        if wl not in parts:
            if wl == "octree":
                from pcg_mi355x.octree import GradedOctreeMesh, make_octree_parts
                parts[wl] = make_octree_parts(GradedOctreeMesh({"1m": (12, 12, 12), "10m": (38, 38, 38)}[args.octree_size], 4, band=1.2, seed=0, symmetry=True), 1)[0]
            else:
                parts[wl] = make_parts(Brick(args.nodes_per_side, seed=0))[0]
        part = parts[wl]
        part.pop("_pcg_mi355x_operator", None)
        pm.configure(comm=None, device=0, rows_per_lane=args.rows_per_lane, operator=kind)
        op = pm.get_operator(part)
        pm.update_bc(part); pm.update_preconditioner(part)
        eff = np.asarray(part["LocDofEff"], np.int64)
        inv = np.zeros(op.n); inv[eff] = part["InvDiagPreCondVector0"]
        op.solve_begin(part["Fext"], np.zeros(op.n), inv, 1e-30, 1000, int(part["GlobData"]["GlobNDofEff"]))
        op.solve_run(3)
        op.bench_hbm(1 << 22, "copy", 1)                       # ---- marker
        op.solve_run(args.steps)
        op.bench_hbm(1 << 22, "copy", 1)                       # ---- marker
        op.solve_end()
        op.close()
        part.pop("_pcg_mi355x_operator", None)


PMC_OPERATOR_KERNELS = {"sell": ("k_spmv",), "dict": ("k_spmv_dict",), "ebe": ("k_ebe",)}      # substrings of the kernels of one operator apply
PMC_PRIMARY = {"sell": ("k_spmv<", "k_spmv_win<"), "dict": ("k_spmv_dict<",), "ebe": ("k_ebe_hexs<", "k_ebe_hex<", "k_ebe_mixed<", "k_ebe_mtile<")}   # one launch per apply


def pmc_traffic_live(args, segments):
    """HBM bytes per operator apply (and per k_vec launch) on THIS box: rocprofv3 --kernel-trace --pmc <counter> passes (one counter
    per pass, as MI355X_MICROARCH.md prescribes) of a short run of this script (pmc_child) over `segments` = ["brick:sell", ...];
    FETCH_SIZE x2 (gfx950: 128-B requests are tallied at 64 B for wide streaming reads - calibrated on the vector kernels in
    profiles/pmc_traffic.json), WRITE_SIZE as is.  -> {segment: {"bytes", "FETCH_SIZE_KB_raw", "WRITE_SIZE_KB_raw", "applies",
    "kernels": {name: bytes per apply}, "vec": {...}}}"""
    import glob
    import sqlite3
    import tempfile
    raw = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pcg_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "k", "--", sys.executable, os.path.abspath(__file__),
               "--pmc-child", ",".join(segments), "--steps", "12", "--nodes-per-side", str(args.nodes_per_side), "--octree-size", args.octree_size,
               "--rows-per-lane", str(args.rows_per_lane)]
        subprocess.run(cmd, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900, check=True)
        db = sqlite3.connect(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0])
        rows = db.execute("select dispatch_id, kernel_name, value from counters_collection where counter_name = ? order by dispatch_id", (ctr,)).fetchall()
        # the dispatch sequence is [set-up 0] M [body 0] M [set-up 1] M [body 1] M ... (M = a run of marker launches)
        groups, in_marker = [[]], False
        for _, name, val in rows:
            if PMC_MARKER in name:
                if not in_marker:
                    groups.append([])
                in_marker = True
            else:
                in_marker = False
                groups[-1].append((name, float(val)))
        bodies = groups[1::2]
        if len(bodies) != len(segments):
            raise RuntimeError(f"PMC pass {ctr}: {len(bodies)} marked segments found, {len(segments)} expected")
        raw[ctr] = bodies
    out = {}
    for i, seg in enumerate(segments):
        kind = seg.split(":")[1]
        res = {"kernels": {}}
        applies = sum(1 for name, _ in raw["FETCH_SIZE"][i] if any(p in name for p in PMC_PRIMARY[kind]))
        tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
        vec = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            for name, val in raw[ctr][i]:
                if any(k in name for k in PMC_OPERATOR_KERNELS[kind]):
                    tot[ctr] += val
                    short = name.split("(")[0].replace("void pcg::", "")
                    res["kernels"].setdefault(short, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})[ctr] += val
                elif "k_vec<true>" in name:
                    vec[ctr] += val
                    vec["n"] += ctr == "FETCH_SIZE"
        if applies == 0:
            raise RuntimeError(f"PMC segment {seg}: no operator launch found")
        res.update(bytes=(2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / applies, FETCH_SIZE_KB_raw=tot["FETCH_SIZE"] / applies,
                   WRITE_SIZE_KB_raw=tot["WRITE_SIZE"] / applies, applies=applies, dispatches=applies)
        res["kernels"] = {k: (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / applies for k, v in res["kernels"].items()}
        if vec["n"]:
            res["vec"] = {"bytes": (2.0 * vec["FETCH_SIZE"] + vec["WRITE_SIZE"]) * 1024 / vec["n"], "FETCH_SIZE_KB_raw": vec["FETCH_SIZE"] / vec["n"],
                          "WRITE_SIZE_KB_raw": vec["WRITE_SIZE"] / vec["n"], "dispatches": vec["n"]}
        out[seg] = res
    return out


def octree_object(measure, log, with_cpu=False, cpu_ranks=0, iteration_roofline=None):
    """BASELINE configs[1] names "a synthetic 3D elasticity octree mesh, 1M DOFs": the multi-level graded octree mesh of
    pcg_mi355x.octree.GradedOctreeMesh (5 cell sizes, 2:1 balanced over faces / edges / corners; the hanging-node cells come in 95
    orientations of 7 patterns with 9-20 nodes besides hex8) on all three operators - iterations/s, operator time, what the formats
    make of it.  Pattern types as the reference's library holds them (round 4): ONE element matrix per class of the cube's symmetries,
    the orientation of an element in the order of its dof list and in its sign vector (partition_mesh.py:453-455);
    `matrix_free_type_per_orientation` = the same mesh with one type (own matrix) per orientation, the round-3 form."""
    import numpy as np
    from pcg_mi355x.octree import GradedOctreeMesh, make_octree_parts
    t0 = time.perf_counter()
    mesh = GradedOctreeMesh((12, 12, 12), 4, band=1.2, seed=0, symmetry=True)
    opart = make_octree_parts(mesh, 1)[0]
    obj = {"workload": "multi-level 2:1-balanced octree mesh around a sphere (GradedOctreeMesh((12,12,12), levels=4, band=1.2, symmetry=True)), Jacobi-PCG Tol 1e-7, 1 part",
           "mesh": mesh.summary(), "mesh_setup_s": time.perf_counter() - t0, "steps": 150, "warmup": 20}
    for kind in ("sell", "dict", "ebe"):
        mm = measure(kind, opart, steps=150, warmup=20, standalone_reps=30)
        op = mm["op"]
        e = {"value": 150 / mm["elapsed"], "unit": "iterations/s", "ms_per_step": mm["elapsed"] / 150 * 1e3, "operator_avg_ms": mm["op_ms"],
             "standalone_operator": mm["standalone"], "solve": mm["final"], "setup_s": mm["t_setup"],
             "vector_phase_ms": mm["vec"]["avg_launch_ms"] if mm["vec"] else None}
        by, fl = op.operator_cost()
        e["operator_bytes"], e["operator_flops"] = by, fl
        t_op = mm["op_ms"] * 1e-3
        e["roofline"] = {"bound": "hbm", "bytes_per_apply": by, "avg_apply_ms": mm["op_ms"], "achieved": by / t_op / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": by / t_op / 1e9 / HBM_PEAK_GBS, "flops_per_apply": fl, "frac_flops": fl / t_op / 1e12 / F64_PEAK_TFLOPS, "traffic": None,
                         "bytes_definition": "what the stored structures of one apply have to move (pcg_operator_cost), all launches of the apply together"}
        if iteration_roofline is not None:
            e["roofline_iteration"] = iteration_roofline(mm, 150)
        if kind in ("sell", "dict"):
            info = op.matrix_info()
            e["sell_padding"] = info["stored_blocks"] / max(1, info["nnzb"]) - 1
            e["nnz"] = op.nnz
        if kind == "dict":
            e["table"] = op.matrix_dictionary_info()        # distinct blocks, how many sit in LDS, the share of stored blocks those cover
        if kind == "ebe":
            e["operator_info"] = op.operator_info()
        obj[{"sell": "assembled", "dict": "assembled_dictionary", "ebe": "matrix_free"}[kind]] = e
        op.close()
        log(f"[octree {kind}] {e['value']:.0f} it/s, operator {e['operator_avg_ms']:.4f} ms, solve {e['solve']}")
    opart.pop("_pcg_mi355x_operator", None)
    try:                         # the same elements, one pattern type per ORIENTATION (95 element matrices instead of 8)
        opart95 = make_octree_parts(GradedOctreeMesh((12, 12, 12), 4, band=1.2, seed=0), 1)[0]
        mm = measure("ebe", opart95, steps=150, warmup=20, standalone_reps=30)
        obj["matrix_free_type_per_orientation"] = {"value": 150 / mm["elapsed"], "unit": "iterations/s", "ms_per_step": mm["elapsed"] / 150 * 1e3,
                                                   "operator_avg_ms": mm["op_ms"], "solve": mm["final"], "operator_info": mm["op"].operator_info()}
        mm["op"].close()
        del opart95
    except Exception as ex:      # noqa: BLE001
        log(f"octree (type per orientation) failed: {ex!r}")
    if with_cpu:                 # north_star: "next to the reference CPU pcg_solver.py timed on the node's own host cores in the same run"
        try:
            obj["cpu_baseline"] = cpu_baseline(opart, "octree:1m", cpu_ranks, "octree", quick=True)
        except Exception as ex:  # noqa: BLE001
            log(f"octree CPU baseline failed: {ex!r}")
    # The value dictionary needs <= 65535 distinct 3x3 blocks.  The random two-phase material above (Ck in {1, 3} x cell size, per
    # element) makes 151 716 of them on this mesh - the plain format is used, `table.distinct_blocks` = 0 says so.  With ONE
    # material (Ck = cell size) the same mesh has 22 333: the dictionary applies, its head sits in LDS, the tail goes through L2.
    mesh1 = GradedOctreeMesh((12, 12, 12), 4, band=1.2, seed=0, two_phase=False, symmetry=True)
    upart = make_octree_parts(mesh1, 1)[0]
    mm = measure("dict", upart, steps=150, warmup=20, standalone_reps=30)
    obj["assembled_dictionary_single_material"] = {
        "note": "same mesh, one material instead of the random two-phase scaling: the only variant of this mesh the dictionary format applies to",
        "value": 150 / mm["elapsed"], "unit": "iterations/s", "ms_per_step": mm["elapsed"] / 150 * 1e3, "operator_avg_ms": mm["op_ms"],
        "standalone_operator": mm["standalone"], "solve": mm["final"], "table": mm["op"].matrix_dictionary_info()}
    mm["op"].close()
    upart.pop("_pcg_mi355x_operator", None)
    return obj


def box_identity(dev):
    """What distinguishes one MI355X box from another for a bandwidth-bound kernel."""
    import torch
    info = {"hostname": socket.gethostname()}
    try:
        p = torch.cuda.get_device_properties(dev)
        info.update(name=p.name, arch=getattr(p, "gcnArchName", ""), cus=p.multi_processor_count, hbm_GiB=round(p.total_memory / 2**30, 1))
    except Exception:      # noqa: BLE001
        pass
    try:
        r = subprocess.run(["rocm-smi", "-d", str(dev), "--showclocks", "--showmaxpower", "--showpower", "--showmemorypartition",
                            "--showcomputepartition", "--showperflevel", "--json"], capture_output=True, text=True, timeout=30)
        js = json.loads(r.stdout[r.stdout.index("{"):])
        card = next(iter(js.values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "fclk", "power", "partition", "performance level")):
                keep[k] = v
        info["rocm_smi"] = keep
    except Exception as ex:      # noqa: BLE001
        info["rocm_smi_error"] = repr(ex)[:200]
    return info


# ---------------------------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if args.pmc_child:
        return pmc_child(args)
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

    box = box_identity(dev) if rank == 0 else None
    stream = None
    setup_all = None

    m = None
    if args.operator in ("both", "sell"):
        m = measure("sell")
        op = m["op"]
        info = op.matrix_info()
        n_loc, nnz_loc = op.n, op.nnz
        sell_bytes, sell_flops = op.operator_cost()
        if rank == 0:                                # what THIS box's HBM delivers to a plain stream kernel on the engine's stream
            stream = {"read_GBps": op.bench_hbm(8 << 30, "read", 10), "copy_GBps": op.bench_hbm(1 << 30, "copy"),
                      "note": "pcg_bench_hbm: 16 B/lane non-temporal grid-stride kernels over 8 GiB (read, 8 loads in flight per lane) / 1 + 1 GiB (copy)"}
        scalar_point = None
        if rank == 0 and world == 1 and args.workload == "brick" and not args.no_finish:
            try:
                scalar_point = scalar_csr_point_device(op)        # the literal CSR volume at THIS size, built on the device
                log(f"scalar-CSR copy: {scalar_point['nnz']} nnz, {scalar_point['median_launch_ms']:.4f} ms = {scalar_point['frac_of_peak']:.3f} of peak on 12 nnz + 20 n")
            except Exception as ex:      # noqa: BLE001 - informational
                log(f"device-side scalar-CSR point failed: {ex!r}")
        if rank == 0:
            if brick.nnz is None:
                brick.nnz = op.nnz if world == 1 else None
            log(f"{args.workload} N={N}: {brick.n_dof} dof, nnz {brick.nnz}; parts {world} grid {grid}; local dof {op.n}, local nnz {op.nnz}; "
                f"RefMeshPart {t_parts:.1f}s, assemble+upload {m['t_setup']:.1f}s; SELL slices {info['n_slices']} x {info['slice_rows']} rows, "
                f"padding {info['stored_blocks'] / info['nnzb'] - 1:.2%}")
        op.close()
    dictionary, dm = None, None
    if args.operator in ("both", "dict"):
        # the SAME assembled matrix with its values stored as 16-bit indices into the table of its distinct 3x3 blocks
        # (PCG_FORMAT_DICTIONARY, k_spmv_dict): same bits out of the SpMV, 0.5 GB instead of 6.9 GB per launch at 10 M dof
        try:
            d = dm = measure("dict")
            db, dfl = d["op"].operator_cost()
            nu = d["op"].matrix_dictionary()
            t_op = d["op_ms"] * 1e-3
            dictionary = {"note": "the same assembled matrix, values replaced by a dictionary of its distinct 3x3 blocks held in LDS (lossless: "
                                  "the SpMV is bit-identical to the plain format); applies when the matrix has <= 65535 distinct blocks - "
                                  "pattern-based meshes, the reference's domain - otherwise the plain format stays",
                          "distinct_blocks": nu, "table": d["op"].matrix_dictionary_info(),
                          "value": args.steps / d["elapsed"], "unit": "iterations/s",
                          "ms_per_step": d["elapsed"] / args.steps * 1e3, "operator_avg_ms": d["op_ms"], "operator_launches_timed": d["n_op"],
                          "standalone_spmv": d["standalone"], "solve": d["final"], "comm": d["comm"], "vector_phase": d["vec"],
                          "roofline_iteration": iteration_roofline(d, args.steps),
                          "roofline": {"kernel": "k_spmv_dict (SELL-64, 16-bit block index + column per stored block, table in LDS)",
                                       "avg_launch_ms": d["op_ms"], "bytes_per_launch": db, "achieved_GBps": db / t_op / 1e9,
                                       "peak_GBps": HBM_PEAK_GBS, "frac_hbm": db / t_op / 1e9 / HBM_PEAK_GBS,
                                       "flops_per_launch": dfl, "achieved_TFLOPs": dfl / t_op / 1e12, "frac_flops": dfl / t_op / 1e12 / F64_PEAK_TFLOPS,
                                       "bound": "LDS reads of the table (72 B per lane per block) + x gathers; neither HBM nor FMA saturated",
                                       "csr_equivalent_GBps": (12.0 * d["op"].nnz + 20.0 * d["op"].n) / t_op / 1e9}}
            if nu == 0:
                dictionary["note"] += " - THIS matrix has too many distinct blocks: measured in the plain format"
            d["op"].close()
        except Exception as ex:              # the headline line must survive a failure of an optional measurement
            if m is None and args.operator == "dict":
                raise
            log(f"dictionary-format measurement failed: {ex!r}")
            dictionary = {"error": repr(ex)}
    e = None
    matrix_free = None
    if args.operator in ("both", "ebe"):
        try:
            e = measure("ebe")
        except Exception as ex:              # the headline line must survive a failure of the optional second measurement
            if m is None:
                raise
            log(f"matrix-free measurement failed: {ex!r}")
            matrix_free = {"error": repr(ex)}
    if e is not None:
        oi = e["op"].operator_info()
        eb, ef = e["op"].operator_cost()
        t_op = e["op_ms"] * 1e-3
        matrix_free = {"note": "SURVEY 8(f)-1: the reference's element-by-element operator kept matrix-free; same PCG driver, same inputs",
                       "value": args.steps / e["elapsed"], "unit": "iterations/s", "ms_per_step": e["elapsed"] / args.steps * 1e3,
                       "operator_avg_ms": e["op_ms"], "operator_launches_timed": e["n_op"], "n_elem": oi["n_elem"], "n_chunks": oi["n_chunks"],
                       "standalone_operator": e["standalone"], "solve": e["final"], "comm": e["comm"], "vector_phase": e["vec"],
                       "roofline_iteration": iteration_roofline(e, args.steps),
                       "roofline": {"kernel": "k_ebe_hexs / k_ebe_hex (a single 8-node pattern type) or k_ebe_mixed / k_ebe_mtile (several pattern types: mixed-type chunks, "
                                              "hex section on the vector FMAs or - below 1.2 M elements - colour-pure hex tiles, + matrix-core tiles) + k_ebe_shared = one operator apply", "avg_apply_ms": e["op_ms"],
                                    "flops_per_apply": ef, "achieved_TFLOPs": ef / t_op / 1e12, "peak_TFLOPs": F64_PEAK_TFLOPS,
                                    "frac_flops": ef / t_op / 1e12 / F64_PEAK_TFLOPS,
                                    "bytes_per_apply": eb, "achieved_GBps": eb / t_op / 1e9, "peak_GBps": HBM_PEAK_GBS,
                                    "frac_hbm": eb / t_op / 1e9 / HBM_PEAK_GBS,
                                    "bound": "neither saturated: latency / LDS-phase bound (DESIGN.md 4b)"}}
        if m is None:
            n_loc = e["op"].n
            if rank == 0:
                stream = {"read_GBps": e["op"].bench_hbm(8 << 30, "read", 10), "copy_GBps": e["op"].bench_hbm(1 << 30, "copy")}
        e["op"].close()

    # ---- N > 1 safety net (round 5).  Everything below this point is an OPTIONAL object of the line - the octree series, the mailbox A/B,
    # the CPU baseline on rank 0 - and every one of them is collective.  The first run of this code on a real multi-GPU node is the
    # driver's own: if one of the extras stalls there (a rank that fell out of a collective), rank 0 still prints the headline it has
    # already measured - marked as such - and ends the job, instead of losing the whole line to the launcher's time-out.
    extras_guard = None
    if world > 1 and rank == 0:
        import threading
        head0 = m if m is not None else (e if e is not None else dm)
        deadline = float(os.environ.get("PCG_BENCH_EXTRAS_DEADLINE_S", "1200"))
        prelim = {"metric": "PCG iterations/sec + SpMV achieved HBM GB/s, 10M-DOF 3D elastostatic CSR",
                  "value": args.steps / head0["elapsed"], "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                  "ms_per_step": head0["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                  "dtype": "f64", "data": "synthetic",
                  "config": {"workload": f"{wl_name}, Jacobi-PCG Tol 1e-7, {world} part(s) {grid[0]}x{grid[1]}x{grid[2]}", "dofs": brick.n_dof, "parts": world,
                             "operator": "assembled SELL-BSR3" if m is not None else ("matrix-free (EBE)" if e is not None else "assembled SELL-BSR3, value dictionary")},
                  "solve": head0["final"], "comm": head0["comm"], "roofline_iteration": iteration_roofline(head0, args.steps),
                  "matrix_free": {"value": matrix_free.get("value"), "ms_per_step": matrix_free.get("ms_per_step")} if isinstance(matrix_free, dict) else None,
                  "roofline": ({"bound": "hbm", "kernel": "k_spmv (SELL-BSR3 SpMV + fused p.Ap) - this rank's part", "achieved": sell_bytes / (m["op_ms"] * 1e-3) / 1e9,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sell_bytes / (m["op_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": sell_bytes,
                                "avg_launch_ms": m["op_ms"], "traffic": None} if m is not None else None),
                  "cpu_baseline": None}

        def bail():
            prelim["extras"] = (f"the optional objects after the headline windows (octree series, mailbox A/B, CPU baseline) did not finish within "
                                f"{deadline:.0f} s (PCG_BENCH_EXTRAS_DEADLINE_S): this is the headline alone, printed by the watchdog; the job was ended")
            print(json.dumps(prelim), flush=True)
            os._exit(0)
        extras_guard = threading.Timer(deadline, bail)
        extras_guard.daemon = True
        extras_guard.start()

    # ---- north_star: "PCG-iterations/sec on a synthetic 3D elasticity octree mesh ... at 1/2/4/8 GPUs": the 10 M-dof graded octree mesh,
    # split into one part per rank by recursive bisection (the METIS stand-in), assembled and matrix-free, on EVERY line (N = 1, 2, 4, 8)
    octree10, opart10 = None, None
    if args.workload == "brick" and args.operator == "both" and not args.no_octree and not args.no_finish:
        try:
            from pcg_mi355x.octree import GradedOctreeMesh, make_octree_parts, bisect_elements
            t0 = time.perf_counter()
            roots10 = tuple(int(v) for v in os.environ.get("PCG_BENCH_OCTREE10_ROOTS", "38,38,38").split(","))     # (tests shrink the mesh)
            mesh10 = GradedOctreeMesh(roots10, 4, band=1.2, seed=0, symmetry=True)
            opart10 = make_octree_parts(mesh10, world, elem_part=bisect_elements(mesh10, world) if world > 1 else None, only=[rank])[0]
            octree10 = {"workload": f"multi-level 2:1-balanced octree mesh around a sphere (GradedOctreeMesh({roots10}, levels=4, band=1.2, symmetry=True)), "
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
                octree10[key] = ent
                if rank == 0:
                    log(f"[octree 10 M dof, {world} part(s), {kind}] {ent['value']:.0f} it/s, operator {ent['operator_avg_ms']:.4f} ms, solve {ent['solve']}")
            opart10.pop("_pcg_mi355x_operator", None)
        except Exception as ex:          # noqa: BLE001 - the headline line must survive (a failure here is the same on every rank)
            log(f"[rank {rank}] octree_10m object failed: {ex!r}")
            octree10 = {"error": repr(ex)}

    # ---- N > 1: the same brick windows with the engine-side reduction (opt-in pcg_comm_enable_mailbox: MPI_SUM through peer-mapped
    # mailboxes inside the engine's own launches instead of ncclAllReduce) - an A/B beside the headline, which stays on RCCL.
    # PCG_BENCH_MAILBOX=0 skips it.  Collective: every rank takes the same branch (enable_mailbox agrees on the outcome).
    mailbox_ab = None
    if world > 1 and getattr(comm, "native", False) and args.operator == "both" and os.environ.get("PCG_BENCH_MAILBOX", "1") != "0":
        try:
            if comm.enable_mailbox(True):
                mailbox_ab = {"enabled": True, "note": "all-reduces through peer-mapped mailboxes (rank-order sums) inside k_fixup / k_vec; exchange unchanged"}
                for kind, key in (("sell", "assembled"), ("ebe", "matrix_free")):
                    mm = measure(kind)
                    mailbox_ab[key] = {"value": args.steps / mm["elapsed"], "unit": "iterations/s", "ms_per_step": mm["elapsed"] / args.steps * 1e3,
                                       "per_rank_ms_per_step": [t / args.steps * 1e3 for t in mm["per_rank_s"]], "solve": mm["final"], "comm": mm["comm"]}
                    mm["op"].close()
                # ... and the engine-side EXCHANGE on top (opt-in pcg_enable_direct_exchange: the pack kernel stores straight into the
                # neighbours' peer-mapped receive buffers, the fix-up waits for their arrival words; the matrix-free engine is built
                # without an interface-first phase) - then no collective kernel is left in the iteration.  PCG_DIRECT_EXCHANGE=1 makes every
                # operator built from here on enable it at set_comm (collective: every rank runs this same code).
                os.environ["PCG_DIRECT_EXCHANGE"] = "1"
                try:
                    direct = {"note": "interface exchange as stores into peer-mapped receive buffers (k_halo_put + arrival words) AND mailbox all-reduces: no "
                                      "collective kernel in the iteration; matrix-free engine with one phase (one element launch)"}
                    for kind, key in (("sell", "assembled"), ("ebe", "matrix_free")):
                        mm = measure(kind)
                        direct[key] = {"value": args.steps / mm["elapsed"], "unit": "iterations/s", "ms_per_step": mm["elapsed"] / args.steps * 1e3,
                                       "per_rank_ms_per_step": [t / args.steps * 1e3 for t in mm["per_rank_s"]], "solve": mm["final"], "comm": mm["comm"],
                                       "enabled": bool(mm["op"].direct_exchange), "reason": mm["op"].direct_exchange_reason}
                        mm["op"].close()
                    mailbox_ab["direct_exchange"] = direct
                except Exception as ex:      # noqa: BLE001
                    log(f"[rank {rank}] direct-exchange A/B failed: {ex!r}")
                    mailbox_ab["direct_exchange"] = {"error": repr(ex)}
                finally:
                    os.environ.pop("PCG_DIRECT_EXCHANGE", None)
                comm.enable_mailbox(False)
            else:
                mailbox_ab = {"enabled": False, "reason": comm.mailbox_reason}
        except Exception as ex:          # noqa: BLE001
            log(f"[rank {rank}] mailbox A/B failed: {ex!r}")
            mailbox_ab = {"enabled": False, "error": repr(ex)}

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
                "operator_s": {k: round(v["t_setup"], 2) for k, v in (("sell", m), ("dict", dm), ("ebe", e)) if v is not None}}
        setup_all = [None] * world
        dist.all_gather_object(setup_all, mine)
    CPU_DONE_KEY = "pcg_bench_rank0_done"
    if rank != 0:
        # rank 0 now times the CPU baselines on the node's host cores: sleep on the rendezvous store (a socket wait) instead of
        # spinning in a barrier next to the processes being timed
        try:
            import datetime
            from torch.distributed.distributed_c10d import _get_default_store
            _get_default_store().wait([CPU_DONE_KEY], datetime.timedelta(seconds=1500))
        except Exception as ex:          # noqa: BLE001 - fall through to the barrier
            log(f"[rank {rank}] store wait: {ex!r}")
        shutdown()
        return

    head = m if m is not None else (e if e is not None else dm)
    elapsed = head["elapsed"]
    iters_per_s = args.steps / elapsed
    out = {
        "metric": "PCG iterations/sec + SpMV achieved HBM GB/s, 10M-DOF 3D elastostatic CSR",
        "value": iters_per_s, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{wl_name}, Jacobi-PCG Tol 1e-7, {world} part(s) {grid[0]}x{grid[1]}x{grid[2]}",
                   "dofs": brick.n_dof, "nnz": brick.nnz, "parts": world,
                   "operator": "assembled SELL-BSR3" if m is not None else ("matrix-free (EBE)" if e is not None else
                                                                            "assembled SELL-BSR3, value dictionary")},
        "solve": head["final"],
        "assembled_dictionary": dictionary,
        "matrix_free": matrix_free if m is not None else None,
        "box": dict(box or {}, hbm_stream=stream),
    }
    if m is not None:
        t_k = m["op_ms"] * 1e-3
        achieved = sell_bytes / t_k / 1e9
        alg_bytes = 12.0 * nnz_loc + 20.0 * n_loc                 # what scalar CSR (f64 value + i32 column per nnz) would move
        col_bytes = 2 if (sell_bytes - 16.0 * n_loc) / info["stored_blocks"] - 72.0 < 3.0 else 4     # 16-bit column offsets, or i32 (+ the per-row bytes of an overflow part)
        out["config"]["format"] = f"SELL-{info['slice_rows']} over 3x3 blocks, {8 * col_bytes}-bit block columns"
        out["config"]["stored_over_true_blocks"] = info["stored_blocks"] / max(1, info["nnzb"])
        if args.workload == "octree":
            out["config"]["format"] += "; rows longer than their slice's base width continue in an overflow part (k_spmv_ovf; roofline.avg_launch_ms covers both launches)"
        out["config"]["spmv_achieved_GBps"] = achieved
        out["roofline"] = {
            "bound": "hbm", "kernel": f"k_spmv<{info['slice_rows'] // 64}, true, {'true' if col_bytes == 2 else 'false'}> (SELL-BSR3 SpMV + fused p.Ap)" + (" - this rank's part" if world > 1 else ""),
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "bytes_per_launch": sell_bytes,
            "bytes_definition": f"stored operator: {72 + col_bytes} B per stored 3x3 block (72 B values + one {8 * col_bytes}-bit column"
                                + (" offset from the slice's base column" if col_bytes == 2 else "") + ") + x in + y out (16 B/dof) + slice "
                                "pointers - the algorithmic traffic of the block format (pcg_operator_cost)",
            "avg_launch_ms": m["op_ms"], "launches_timed": m["n_op"],
            "traffic": None,
            "traffic_note": "PMC FETCH_SIZE/WRITE_SIZE need rocprofv3 passes of their own; the committed passes for this kernel are under "
                            "profiles/ (DESIGN.md section 8) - traffic / bytes_per_launch = 1.03",
            "hbm_stream_this_box": stream, "frac_of_stream_read": achieved / stream["read_GBps"] if stream else None,
            "csr_equivalent_bytes": alg_bytes, "csr_equivalent_GBps": alg_bytes / t_k / 1e9,
            "csr_equivalent_note": "SURVEY 8(d) formula 12 nnz + 20 n: a scalar-CSR kernel's traffic for the same product; NOT what this "
                                   "kernel moves (it can exceed the HBM peak) - kept for comparison with CSR codes only",
            "standalone_spmv": m["standalone"]}
        out["roofline_vector_phase"] = m["vec"]
        out["roofline_iteration"] = iteration_roofline(m, args.steps)
        if world == 1 and args.workload == "brick" and not args.no_finish:
            if scalar_point is not None:
                out["roofline"]["scalar_csr_same_run"] = scalar_point
            else:
                try:
                    out["roofline"]["scalar_csr_same_run"] = scalar_csr_point(dev)      # host-built CSR at N = 100 (round 3's point)
                except Exception as ex:      # noqa: BLE001 - informational
                    log(f"scalar-CSR point failed: {ex!r}")
        try:        # PMC traffic of an identical launch, collected by separate rocprofv3 --pmc passes (profiles/pmc_traffic.json)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(f"N{N}_rpl{info['slice_rows'] // 64}" + ("_col16" if col_bytes == 2 else ""))
            if pmc and world == 1 and args.workload == "brick":
                out["roofline"]["traffic"] = pmc["traffic_bytes_per_launch"]
                out["roofline"]["traffic_note"] = ("from profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same launch "
                                                   "(gfx950-corrected), collected on another box in another session - not a measurement of this run")
        except OSError:
            pass
    pmc_live = None
    import shutil
    profiled = any(k.startswith(("ROCPROF", "ROCP_", "ROCTX")) for k in os.environ)       # already under a profiler: no nested passes
    if world == 1 and not args.no_pmc_traffic and (args.pmc_traffic or (shutil.which("rocprofv3") and not profiled)):
        # HBM traffic of THIS run's kernels: two rocprofv3 --pmc passes of a short child run over the operators of this line
        wl = args.workload
        segs = [f"{wl}:sell"] if m is not None else []
        if e is not None:
            segs.append(f"{wl}:ebe")
        if wl == "brick" and not args.no_octree and not args.no_finish:
            segs += ["octree:sell", "octree:ebe"]                  # the `octree` object (1 M dof)
        try:
            t0 = time.perf_counter()
            pmc_live = pmc_traffic_live(args, segs) if segs else None
            log(f"PMC traffic passes over {segs}: {time.perf_counter() - t0:.0f} s")
        except Exception as ex:      # noqa: BLE001
            log(f"live PMC traffic measurement failed: {ex!r}")

    def pmc_note(r):
        return (f"measured on THIS box by two rocprofv3 --kernel-trace --pmc passes of a short run of this command (FETCH_SIZE {r['FETCH_SIZE_KB_raw']:.0f} KB "
                f"x2 gfx950 correction + WRITE_SIZE {r['WRITE_SIZE_KB_raw']:.0f} KB per apply, mean of {r['applies']} applies; per kernel: "
                + ", ".join(f"{k} {v / 1e6:.1f} MB" for k, v in r["kernels"].items()) + ")")
    if pmc_live and m is not None and f"{args.workload}:sell" in pmc_live:
        live = pmc_live[f"{args.workload}:sell"]
        out["roofline"]["traffic"] = live["bytes"]
        out["roofline"]["traffic_note"] = pmc_note(live)
        out["roofline"]["traffic_over_bytes"] = live["bytes"] / sell_bytes
        if live.get("vec") and out.get("roofline_vector_phase"):
            v = out["roofline_vector_phase"]
            v["traffic"] = live["vec"]["bytes"]
            v["traffic_over_bytes"] = live["vec"]["bytes"] / v["bytes_per_launch"]
            v["traffic_note"] = (f"same two PMC passes: FETCH_SIZE {live['vec']['FETCH_SIZE_KB_raw']:.0f} KB x2 + WRITE_SIZE "
                                 f"{live['vec']['WRITE_SIZE_KB_raw']:.0f} KB, mean of {live['vec']['dispatches']} launches of k_vec<true>")
    if pmc_live and matrix_free and "roofline" in matrix_free and f"{args.workload}:ebe" in pmc_live:
        live = pmc_live[f"{args.workload}:ebe"]
        matrix_free["roofline"].update(traffic=live["bytes"], traffic_over_bytes=live["bytes"] / matrix_free["roofline"]["bytes_per_apply"],
                                       traffic_note=pmc_note(live))
    if world > 1:
        out["comm"] = {"transport": transport, "ranks": comm.world, "per_rank_ms_per_step": [t / args.steps * 1e3 for t in head["per_rank_s"]],
                       "per_rank_setup_s": setup_all}
        if os.environ.get("PCG_BENCH_NATIVE_ERROR"):
            out["comm"]["native_error"] = os.environ["PCG_BENCH_NATIVE_ERROR"]
        if head["comm"]:
            out["comm"].update(head["comm"])
        if mailbox_ab is not None:
            out["comm"]["mailbox"] = mailbox_ab
    if world == 1 and args.workload == "brick" and not args.no_octree and not args.no_finish:
        try:
            out["octree"] = octree_object(measure, log, with_cpu=not args.no_cpu_baseline, cpu_ranks=args.cpu_ranks, iteration_roofline=iteration_roofline)
            for key, seg in (("assembled", "octree:sell"), ("matrix_free", "octree:ebe")):
                if pmc_live and seg in pmc_live and key in out["octree"]:
                    r = out["octree"][key]["roofline"]
                    r.update(traffic=pmc_live[seg]["bytes"], traffic_over_bytes=pmc_live[seg]["bytes"] / r["bytes_per_apply"], traffic_note=pmc_note(pmc_live[seg]))
        except Exception as ex:          # noqa: BLE001 - the headline line must survive
            log(f"octree object failed: {ex!r}")
            out["octree"] = {"error": repr(ex)}
    if octree10 is not None:
        out["octree_10m"] = octree10
        if not args.no_cpu_baseline and "error" not in octree10 and "PCG_BENCH_OCTREE10_ROOTS" not in os.environ:
            try:          # at most 16 processes: every one builds the 10 M-dof mesh for itself (3.6 GB at its peak)
                octree10["cpu_baseline"] = cpu_baseline(opart10, "octree:10m", min(args.cpu_ranks or 16, 16), "octree", quick=True,
                                                        total_dofs=octree10["dofs"])
            except Exception as ex:      # noqa: BLE001
                log(f"octree_10m CPU baseline failed: {ex!r}")
    if not args.no_cpu_baseline:
        log("timing the CPU baseline (the reference's NumPy arithmetic: 1 core, then R processes x 1 thread; the C port beside it) ...")
        out["cpu_baseline"] = cpu_baseline(part, N if args.workload == "brick" else f"octree:{args.octree_size}", args.cpu_ranks, args.workload,
                                           quick=world > 1, total_dofs=brick.n_dof)
    if extras_guard is not None:
        extras_guard.cancel()
    print(json.dumps(out), flush=True)
    if world > 1:
        try:
            from torch.distributed.distributed_c10d import _get_default_store
            _get_default_store().set(CPU_DONE_KEY, "1")
        except Exception as ex:          # noqa: BLE001
            log(f"store set: {ex!r}")
    shutdown()


if __name__ == "__main__":
    main()
