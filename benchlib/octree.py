"""BASELINE configs[1]: "synthetic 3D elasticity octree mesh, 1M DOFs" - the `octree` object of the extras."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

from . import ROOT, BENCH_PY, METRIC, HBM_PEAK_GBS, F64_PEAK_TFLOPS, log

from .cpu import cpu_baseline


def octree_object(measure, log, with_cpu=False, cpu_ranks=0, iteration_roofline=None, full=False):
    """BASELINE configs[1] names "a synthetic 3D elasticity octree mesh, 1M DOFs": the multi-level graded octree mesh of
    pcg_mi355x.octree.GradedOctreeMesh (5 cell sizes, 2:1 balanced over faces / edges / corners; the hanging-node cells come in 95
    orientations of 7 patterns with 9-20 nodes besides hex8) on all three operators - iterations/s, operator time, what the formats
    make of it.  Pattern types as the reference's library holds them (round 4): ONE element matrix per class of the cube's symmetries,
    the orientation of an element in the order of its dof list and in its sign vector (partition_mesh.py:453-455);
    `matrix_free_type_per_orientation` = the same mesh with one type (own matrix) per orientation, the round-3 form.
    Default: the assembled and the matrix-free operator; full=True adds the dictionary format, the orientation variant and the CPU run."""
    import numpy as np
    from pcg_mi355x.octree import GradedOctreeMesh, make_octree_parts
    t0 = time.perf_counter()
    mesh = GradedOctreeMesh((12, 12, 12), 4, band=1.2, seed=0, symmetry=True)
    opart = make_octree_parts(mesh, 1)[0]
    obj = {"workload": "multi-level 2:1-balanced octree mesh around a sphere (GradedOctreeMesh((12,12,12), levels=4, band=1.2, symmetry=True)), Jacobi-PCG Tol 1e-7, 1 part",
           "mesh": mesh.summary(), "mesh_setup_s": time.perf_counter() - t0, "steps": 150, "warmup": 20}
    obj["dofs"] = int(mesh.n_dof)
    for kind in (("sell", "dict", "ebe") if full else ("sell", "ebe")):
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
    if not full:
        return obj
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
