"""Informational GPU points beside the headline: SURVEY 8(d)'s literal scalar-CSR SpMV, and what box the numbers come from."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

from . import ROOT, BENCH_PY, METRIC, HBM_PEAK_GBS, F64_PEAK_TFLOPS, log

import socket


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
