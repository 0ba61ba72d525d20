"""bench.py's output contract (round 6; BENCH_r05.json was `parsed: null` because the line had grown to 23 KB): the last stdout line is
a compact headline - the driver's fields + `roofline` + `cpu_baseline`, numbers only, < 4 KB whatever the optional objects hold - and
the full record goes to a file.  Host logic only: no GPU, no oracle."""
import json
import os

import pytest

from util import ROOT


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _round5_record():
    """A real full record: the 23 KB line of round 5's last session on the driver's command (profiles/, committed)."""
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench_N150_sessionH_final.json")))


def test_compact_line_of_a_real_record_parses_and_is_small():
    from benchlib.line import compact_line, MAX_LINE
    full = _round5_record()
    assert len(json.dumps(full)) > 20000                          # what the driver could not parse
    full["extras_file"] = "/x/bench_extras.json"
    line = compact_line(full)
    assert len(line) < MAX_LINE == 4096 and "\n" not in line
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in out, k
    assert out["metric"] == full["metric"] and out["unit"] == "iterations/s" and out["n_gpus"] == 1 and out["dtype"] == "f64"
    assert abs(out["value"] / full["value"] - 1) < 1e-5 and out["vs_baseline"] is None
    assert set(out["config"]) >= {"workload", "dofs", "nnz", "parts", "operator"} and "model" not in out["config"]
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 and 0.5 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-3 * r["achieved"]
    assert r["traffic"] and 0.9 < r["traffic_over_bytes"] < 1.3 and 0.5 < r["scalar_csr_frac"] < 1.0
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "iterations/s" and c["sample"]
    assert out["also"]["matrix_free_its"] > out["value"] and out["also"]["octree_10m"]["matrix_free_its"] > 0
    # numbers and short names only: no string of the line is a paragraph
    def strings(o):
        if isinstance(o, str):
            yield o
        elif isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)
    assert max(len(s) for s in strings(out)) <= 200


def test_compact_line_survives_oversized_and_missing_objects():
    from benchlib.line import compact_line, MAX_LINE
    full = _round5_record()
    full["comm"] = {"transport": "native " * 50, "ranks": 8, "per_rank_ms_per_step": [1.23456789] * 64, "native_error": "x" * 5000}
    full["skipped"] = ["object_%d" % k * 20 for k in range(100)]
    full["config"]["workload"] = "w" * 3000
    line = compact_line(full)
    assert len(line) < MAX_LINE and json.loads(line)["value"] > 0
    for k in ("roofline", "cpu_baseline", "matrix_free", "octree", "octree_10m", "solve", "roofline_iteration"):      # a headline-only record
        full.pop(k, None)
    out = json.loads(compact_line(full))
    assert out["roofline"] is None and out["cpu_baseline"] is None and out["value"] > 0
    full["value"] = float("nan")                                   # never NaN / Infinity in the line: strict JSON
    assert json.loads(compact_line(full))["value"] is None


def test_budget_skips_what_no_longer_fits():
    from benchlib.line import Budget
    b = Budget(0.5)
    assert b.go("cheap", 0.1) and not b.go("expensive", 30.0) and b.skipped == ["expensive"]
    assert Budget(1e9).go("anything", 1e6)


def test_write_extras_never_raises(tmp_path):
    from benchlib.line import write_extras
    full = _round5_record()
    p = write_extras(full, str(tmp_path / "extras.json"))
    assert json.load(open(p))["value"] == full["value"]
    assert write_extras(full, str(tmp_path / "no" / "such" / "dir" / "x.json")) is None


def test_bench_defaults_are_the_driver_contract():
    b = _bench()
    a = b.parse_args([])
    assert a.gpus == 1 and a.steps > 0 and a.warmup > 0 and not a.full and not a.ab_engine_side and a.extras_budget_s <= 200
    a = b.parse_args(["--gpus", "8", "--steps", "20", "--warmup", "5"])
    assert (a.gpus, a.steps, a.warmup) == (8, 20, 5)
