"""TEST INFRASTRUCTURE: the synthetic models behind tests/golden/part_*.npz (partitioner + MDF parity).

Each case is a global model (pcg_mi355x.mdf) plus an element -> part vector.  oracle/make_partition_golden.py
feeds the MDF files to the UNMODIFIED reference pipeline (run_metis.config_GlobData -> partition_mesh ->
pcg_solver) and stores what it exported and solved; tests/test_partition.py rebuilds the same model here (no
reference needed) and requires pcg_mi355x.partition to reproduce every exported key exactly.
"""
from __future__ import annotations

import numpy as np

CASES = {
    # name: generator, partition
    "part_brick_p3": {"brick": dict(N=6, n_types=2), "grid": (1, 1, 3)},
    "part_brick_p4": {"brick": dict(N=7, n_types=3), "grid": (2, 2, 1)},
    "part_brick_p8": {"brick": dict(N=7, n_types=1), "grid": (2, 2, 2)},
    "part_brick_p1": {"brick": dict(N=5, n_types=2), "grid": (1, 1, 1)},
    "part_octree_p3": {"octree": (4, 4, 2, 2), "sign_seed": 7, "rcb": 3},       # hanging-node patterns, 2 pattern types
    "part_octree_p5": {"octree": (6, 4, 3, 1), "sign_seed": None, "rcb": 5},    # uneven recursive bisection
    "part_octree_p1": {"octree": (4, 6, 2, 2), "sign_seed": 3, "rcb": 1},
    # every element to a random part: disconnected parts, every part a neighbour of every other, nodes shared by up to 5
    "part_brick_rand5": {"brick": dict(N=6, n_types=2), "random": (5, 11)},
}
SOLVER = {"Tol": 1e-7, "MaxIter": 10000}


def build_model(name):
    """(model, ele_part) of a case."""
    from pcg_mi355x import mdf
    from pcg_mi355x.partition import geometric_partition
    c = CASES[name]
    if "brick" in c:
        from pcg_mi355x.brick import Brick, block_partition
        b = Brick(c["brick"]["N"], seed=0, n_types=c["brick"]["n_types"])
        if "random" in c:
            n, seed = c["random"]
            ep = np.random.default_rng(seed).integers(0, n, b.n_elem)
            ep[:n] = np.arange(n)                                     # no empty part
            return mdf.model_from_brick(b), ep.astype(np.int64)
        return mdf.model_from_brick(b), block_partition(b, *c["grid"]).astype(np.int64)
    from pcg_mi355x.octree import TwoLevelMesh
    mesh = TwoLevelMesh(*c["octree"], seed=0)
    model = mdf.model_from_octree(mesh, c["sign_seed"])
    return model, geometric_partition(model, c["rcb"])


def solver_glob_data(tol=SOLVER["Tol"], max_iter=SOLVER["MaxIter"]):
    """The GlobData entries pcg_solver.py adds after reading a part (initGlobData :45-52, readGlobalSettings :113-139)."""
    return {"MaxIter": int(max_iter), "Tol": float(tol), "TimeStepDelta": [0, 1], "TimeStepCount": 1, "FintCalcMode": "outbin",
            "MP_TimeRecData": {"dT_FileRead": 0.0, "dT_Calc": 0.0, "dT_CommWait": 0.0, "dT_CalcList": [], "dT_CommWaitList": [],
                               "TimeStepCountList": [], "t0": 0.0},
            "TimeList_Flag": np.zeros(2), "TimeList_RelRes": np.zeros(2), "TimeList_Iter": np.zeros(2)}


def prepare_for_solve(parts):
    """pcg_solver.py:996-997 + the solver's GlobData entries, in place."""
    for p in parts:
        gd = dict(p["GlobData"])
        gd.update(solver_glob_data())
        p["GlobData"] = gd
        p["Un"] = np.zeros(p["NDOF"])
        p["DofWeightVector_Eff"] = p["DofWeightVector"][p["LocDofEff"]]
    return parts


SKIP_GLOB = {"ScratchPath", "MDF_Path", "PyDataPath_Part"}          # absolute paths of the run that made the fixture


def flatten_part(part, prefix=""):
    """RefMeshPart (nested dicts / lists / arrays / scalars) -> {path: ndarray}; dtype and shape are kept, so an
    exact comparison of two flattened parts is an exact comparison of the parts."""
    out = {}

    def walk(v, path):
        if isinstance(v, dict):
            out[path + "/#keys"] = np.array(sorted(str(k) for k in v if k not in SKIP_GLOB))
            for k, x in v.items():
                if k not in SKIP_GLOB:
                    walk(x, f"{path}/{k}")
        elif isinstance(v, (list, tuple)):
            out[path + "/#len"] = np.array(len(v))
            for i, x in enumerate(v):
                walk(x, f"{path}/{i}")
        else:
            a = np.asarray(v)
            if a.dtype == object:
                raise TypeError(f"{path}: object array")
            out[path] = a

    walk(part, prefix)
    return out
