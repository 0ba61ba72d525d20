"""The ONE stdout line of bench.py, and where everything else goes.

Round 5's line had grown to 23 KB (nine nested objects with prose in them) and the driver could not parse it (BENCH_r05.json:
`parsed: null`).  Since round 6 the last stdout line is a COMPACT headline - the contract's fields + `roofline` + `cpu_baseline`, numbers
and short names only, < 4 KB by construction (tests/test_bench_line.py) - and the full record of the run (every optional object, every
note) is written to `bench_extras.json` next to bench.py and summarised on stderr.
"""
from __future__ import annotations

import json
import os
import time

from . import ROOT, log

MAX_LINE = 4096


def _num(x, nd=6):
    """A float with `nd` significant digits (the line carries numbers, not 17-digit reprs); everything else unchanged."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{nd}g}")
    return x


def _pick(d, keys):
    return {k: _num(d.get(k)) for k in keys if isinstance(d, dict) and k in d}


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 3] + "..."


def compact(full: dict) -> dict:
    """The headline dict of the line from the full record `full` (bench.py main): numbers only, no prose."""
    out = {k: _num(full.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                          "vs_baseline", "dtype", "data")}
    cfg = full.get("config") or {}
    out["config"] = {"workload": _short(cfg.get("workload", ""), 160), **_pick(cfg, ("dofs", "nnz", "parts", "operator"))}
    rf = full.get("roofline")
    if isinstance(rf, dict):
        r = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch", "avg_launch_ms", "launches_timed", "traffic_over_bytes",
                       "launches_per_apply", "avg_apply_ms"))
        r["kernel"] = _short(rf.get("kernel", ""), 60)
        sc = rf.get("scalar_csr_same_run")
        if isinstance(sc, dict):                                # SURVEY 8(d)'s literal CSR kernel at the same size: 12 nnz + 20 n bytes over ITS time
            r["scalar_csr_frac"] = _num(sc.get("frac_of_peak"))
            r["scalar_csr_ms"] = _num(sc.get("median_launch_ms"))
        st = rf.get("hbm_stream_this_box")
        if isinstance(st, dict):
            r["box_read_stream_GBps"] = _num(st.get("read_GBps"))
        out["roofline"] = r
    else:
        out["roofline"] = None
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict) and cb.get("value") is not None:
        c = _pick(cb, ("kind", "value", "unit", "cores"))
        c["host_cpu"] = _short(cb.get("host_cpu", ""), 60)
        c["sample"] = _short(cb.get("sample", ""), 120)
        one = cb.get("one_core")
        if isinstance(one, dict):
            c["one_core_value"] = _num(one.get("value"))
            c["one_core_dofs"] = one.get("dofs")
        out["cpu_baseline"] = c
    else:
        out["cpu_baseline"] = None
    it = full.get("roofline_iteration")
    if isinstance(it, dict):
        out["roofline_iteration"] = _pick(it, ("frac", "achieved", "peak", "bytes_per_iteration"))
    sv = full.get("solve")
    if isinstance(sv, dict):
        out["solve"] = _pick(sv, ("flag", "iter", "relres"))
    also = {}
    mf = full.get("matrix_free")
    if isinstance(mf, dict) and mf.get("value") is not None:
        also["matrix_free_its"] = _num(mf["value"])
        also["matrix_free_op_ms"] = _num(mf.get("operator_avg_ms"))
    for key in ("octree", "octree_10m"):
        o = full.get(key)
        if isinstance(o, dict) and "error" not in o:
            e = {"dofs": o.get("dofs") or (o.get("mesh") or {}).get("dofs")}
            for sub, tag in (("assembled", "assembled_its"), ("matrix_free", "matrix_free_its")):
                if isinstance(o.get(sub), dict):
                    e[tag] = _num(o[sub].get("value"))
                    ri = o[sub].get("roofline_iteration")
                    if isinstance(ri, dict):
                        e[tag.replace("_its", "_iter_frac")] = _num(ri.get("frac"), 4)
            also[key] = e
    if also:
        out["also"] = also
    cm = full.get("comm")
    if isinstance(cm, dict):
        c = _pick(cm, ("ranks", "halo_wait_ms_per_iter", "allreduce_ms_per_iter", "exchanges_per_iter", "allreduces_per_iter"))
        c["transport"] = _short(cm.get("transport", ""), 48)
        if isinstance(cm.get("per_rank_ms_per_step"), list):
            c["per_rank_ms_per_step"] = [_num(v, 5) for v in cm["per_rank_ms_per_step"][:16]]
        if cm.get("native_error"):
            c["native_error"] = _short(cm["native_error"], 120)
        out["comm"] = c
    if full.get("skipped"):
        out["skipped"] = [_short(s, 40) for s in full["skipped"]][:12]
    if full.get("extras"):
        out["extras"] = _short(full["extras"], 200)
    out["extras_file"] = full.get("extras_file")
    return out


def compact_line(full: dict) -> str:
    """json of compact(full), guaranteed to fit MAX_LINE: optional members are dropped, least important first, until it does."""
    c = compact(full)
    line = json.dumps(c, separators=(",", ":"))
    for drop in ("skipped", "comm", "also", "solve", "roofline_iteration"):
        if len(line) < MAX_LINE:
            break
        c.pop(drop, None)
        line = json.dumps(c, separators=(",", ":"))
    assert len(line) < MAX_LINE, len(line)
    return line


def write_extras(full: dict, path: str | None = None) -> str | None:
    """The full record -> bench_extras.json (beside bench.py, and under gpurun_out/ when that exists so that it travels back from the
    GPU box); a few lines of it -> stderr.  Never raises: the headline must not die of its appendix."""
    paths = [path or os.environ.get("PCG_BENCH_EXTRAS", os.path.join(ROOT, "bench_extras.json"))]
    if path is None and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", f"bench_extras_n{full.get('n_gpus', 1)}_{int(time.time())}.json"))
    wrote = None
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(full, f, indent=1, default=repr)
            wrote = wrote or p
        except OSError as ex:
            log(f"could not write {p}: {ex!r}")
    try:
        rf, cb = full.get("roofline") or {}, full.get("cpu_baseline") or {}
        log(f"headline: {full.get('value')} {full.get('unit')} at N = {full.get('n_gpus')}; {str(rf.get('kernel'))[:40]} {rf.get('avg_launch_ms')} ms = "
            f"{rf.get('frac')} of {rf.get('peak')} {rf.get('unit')}, traffic {rf.get('traffic')}; cpu_baseline {cb.get('value')} it/s on {cb.get('cores')} cores; "
            f"full record: {wrote}")
    except Exception as ex:      # noqa: BLE001
        log(f"summary failed: {ex!r}")
    return wrote


class Budget:
    """Wall-clock budget for the OPTIONAL objects after the headline window: an object starts only while its estimated cost still fits
    (130 s of run for 26 ms of timed work).  At N > 1 rank 0 decides and tells the others (every object is collective)."""

    def __init__(self, seconds, world=1, rank=0, dist=None):
        self.t_end = time.perf_counter() + float(seconds)
        self.world, self.rank, self.dist = world, rank, dist
        self.skipped = []

    def left(self):
        return self.t_end - time.perf_counter()

    def go(self, name, est_s):
        ok = self.left() >= est_s
        if self.world > 1:
            box = [ok]
            self.dist.broadcast_object_list(box, src=0)
            ok = bool(box[0])
        if not ok:
            self.skipped.append(name)
            if self.rank == 0:
                log(f"skipping `{name}` (needs ~{est_s:.0f} s, {max(0.0, self.left()):.0f} s of the extras budget left; --extras-budget-s)")
        return ok
