"""The CPU baseline leg: the oracle (the reference's arithmetic restated, bit-identical to src/solver/pcg_solver.py - oracle/make_golden.py)
timed on the GPU box's host cores.  The ONLY part of bench.py that imports oracle/ - never inside a GPU-timed region."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

from . import ROOT, BENCH_PY, METRIC, HBM_PEAK_GBS, F64_PEAK_TFLOPS, log


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


def numpy_reference_point(part, budget_s=12.0, max_dofs=12_000_000):
    """The reference's OWN arithmetic path on one host core: the NumPy restatement of calcMatVecProd / PCG (pcg_oracle with
    use_c=False - bit-identical to the unmodified pcg_solver.py on every fixture, oracle/make_golden.py), single-threaded BLAS
    like the reference sets it (pcg_solver.py:10-15).  Bounded sample AT THE BENCH'S OWN SIZE (round 6: the 10 M-dof system itself, a few
    iterations - VERDICT r5 missing #6); only a part above `max_dofs` (the 100 M-dof system) or a part with neighbours (N > 1) is replaced
    by the 1 M-dof brick of the same generator, and the sample says so."""
    import copy
    import numpy as np
    import pcg_oracle
    note = "the bench's own part"
    if part["NDOF"] > max_dofs or len(part.get("NbrMPIdVector", ())) > 0:
        from pcg_mi355x.brick import Brick, make_parts
        why = ("the bench's system is too large for a bounded NumPy sample" if part["NDOF"] > max_dofs else
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


def cpu_baseline(part, N, ranks=0, workload="brick", quick=False, total_dofs=None, one_core_budget_s=8.0, mp_solve_s=12.0):
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
        out["numpy_reference_path"] = out["one_core"] = single = numpy_reference_point(part, budget_s=one_core_budget_s)
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
    iters = int(max(10, min(400, mp_solve_s * est)))
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
