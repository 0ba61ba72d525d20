"""HBM traffic of the run's own kernels from the PMC counters: separate rocprofv3 --pmc passes (one counter per pass, as
MI355X_MICROARCH.md prescribes) of a short child run of bench.py (--pmc-child)."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

from . import ROOT, BENCH_PY, METRIC, HBM_PEAK_GBS, F64_PEAK_TFLOPS, log


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
PMC_PRIMARY = {"sell": ("k_spmv<", "k_spmv_win<"), "dict": ("k_spmv_dict<",), "ebe": ("k_ebe_hexs<", "k_ebe_hex<", "k_ebe_mixed<", "k_ebe_mtile<")}   # the launch(es) that make an apply


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
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "k", "--", sys.executable, BENCH_PY,
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
        n_vec = sum(1 for name, _ in raw["FETCH_SIZE"][i] if "k_vec<true" in name)
        if n_vec > 0:
            applies = n_vec            # one fused vector launch per iteration = per apply (round 6: an apply of k_spmv may be several launches)
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
