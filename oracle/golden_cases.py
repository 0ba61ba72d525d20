"""TEST INFRASTRUCTURE - the shared table of golden cases (inputs are re-generated from these
parameters by pcg_mi355x.brick; outputs live in tests/golden/<name>.npz, written by
oracle/make_golden.py from the *reference's own functions*)."""
from __future__ import annotations

import numpy as np

# name: dict(N, grid, n_types, tol, max_iter, ud (prescribed z-displacement on the fixed face), x0)
CASES = {
    "n9_p1":        dict(N=9,  grid=(1, 1, 1), n_types=1, tol=1e-7, max_iter=10000, ud=0.0),
    "n9_p2":        dict(N=9,  grid=(1, 1, 2), n_types=1, tol=1e-7, max_iter=10000, ud=0.0),
    "n9_p8":        dict(N=9,  grid=(2, 2, 2), n_types=1, tol=1e-7, max_iter=10000, ud=0.0),
    "n13_t3_p4_ud": dict(N=13, grid=(2, 2, 1), n_types=3, tol=1e-7, max_iter=10000, ud=0.01),
    "n17_p1":       dict(N=17, grid=(1, 1, 1), n_types=1, tol=1e-7, max_iter=10000, ud=0.0),
    "n17_t2_p8":    dict(N=17, grid=(2, 2, 2), n_types=2, tol=1e-7, max_iter=10000, ud=0.0),
    # failure / edge paths
    "n9_maxiter":   dict(N=9,  grid=(1, 1, 1), n_types=1, tol=1e-7, max_iter=30,    ud=0.0),      # Flag 1 -> min-residual iterate
    "n9_p2_maxiter": dict(N=9, grid=(1, 1, 2), n_types=1, tol=1e-7, max_iter=40,    ud=0.0),
    "n9_stagnate":  dict(N=9,  grid=(1, 1, 1), n_types=1, tol=1e-15, max_iter=1500, ud=0.0),      # Flag 3 via :560-562
    "n9_raise":     dict(N=9,  grid=(1, 1, 1), n_types=1, tol=1e-15, max_iter=10000, ud=0.0),     # MaxMSteps<0 -> raise Warning (:549)
    "n9_p2_raise":  dict(N=9,  grid=(1, 1, 2), n_types=1, tol=1e-15, max_iter=10000, ud=0.0),
    "n9_flag4":     dict(N=9,  grid=(1, 1, 1), n_types=1, tol=1e-7, max_iter=10000, ud=0.0, negate_ck=True),       # pq <= 0 (:492)
    "n9_p2_flag4":  dict(N=9,  grid=(1, 1, 2), n_types=1, tol=1e-7, max_iter=10000, ud=0.0, negate_ck=True),
    "n9_flag2":     dict(N=9,  grid=(1, 1, 1), n_types=1, tol=1e-7, max_iter=10000, ud=0.0, isolated_node=True),   # inf in M^-1 r (:448)
    # two-level octree mesh with hanging nodes: pattern types hex8 (two sizes) + 13-node transition (nd = 39)
    "oct_p1":       dict(octree=(8, 8, 4, 3), parts=1, axis=0, sign_seed=3, tol=1e-7, max_iter=10000),
    "oct_p3":       dict(octree=(8, 6, 3, 2), parts=3, axis=0, sign_seed=5, tol=1e-7, max_iter=10000),
    "oct_p2_z":     dict(octree=(6, 6, 4, 2), parts=2, axis=2, sign_seed=None, tol=1e-7, max_iter=10000),   # interface = the transition layer
    # multi-level 2:1-balanced octree mesh around a sphere (round 3): 3 cell sizes, dozens of pattern types with 9-20 nodes
    "goct_p1":      dict(graded=((3, 3, 3), 2, 1.2), parts=1, sign_seed=None, tol=1e-7, max_iter=10000),
    "goct_p4":      dict(graded=((3, 3, 3), 2, 1.2), parts=4, sign_seed=9, tol=1e-7, max_iter=10000),      # recursive bisection
    "goct_p3_ud":   dict(graded=((3, 3, 3), 2, 1.2), parts=3, sign_seed=5, tol=1e-7, max_iter=10000, ud=2e-3),   # + prescribed displacements (:234-237)
    # the same mesh with ONE pattern type per class of the cube's 48 symmetries (round 4): 4 types instead of 29, every element with its
    # own dof order and signs - the reference's pattern-library form (partition_mesh.py:453-455,1074)
    "goct_sym_p1":  dict(graded=((3, 3, 3), 2, 1.2), parts=1, sign_seed=None, tol=1e-7, max_iter=10000, symmetry=True),
    "goct_sym_p3":  dict(graded=((3, 3, 3), 2, 1.2), parts=3, sign_seed=7, tol=1e-7, max_iter=10000, symmetry=True, ud=2e-3),
    "n9_zero_rhs":  dict(N=9,  grid=(1, 1, 1), n_types=1, tol=1e-7, max_iter=10000, ud=0.0, zero_rhs=True),   # :387-395
    "n9_good_x0":   dict(N=9,  grid=(1, 1, 1), n_types=1, tol=1e-7, max_iter=10000, ud=0.0, good_x0="n9_p1"),  # :421-426
}


def build_case(name, golden_dir=None):
    """Re-create the RefMeshPart dicts of a case (before updateBC)."""
    import os
    from pcg_mi355x.brick import Brick, make_parts, block_partition
    c = CASES[name]
    if "octree" in c:
        from pcg_mi355x.octree import TwoLevelMesh, make_octree_parts
        mesh = TwoLevelMesh(*c["octree"], seed=0)
        return mesh, make_octree_parts(mesh, c["parts"], c["axis"], c["tol"], c["max_iter"], c["sign_seed"])
    if "graded" in c:
        from pcg_mi355x.octree import GradedOctreeMesh, make_octree_parts, bisect_elements
        roots, levels, band = c["graded"]
        mesh = GradedOctreeMesh(roots, levels, band=band, seed=0, symmetry=c.get("symmetry", False))
        ep = bisect_elements(mesh, c["parts"]) if c["parts"] > 1 else None
        parts = make_octree_parts(mesh, c["parts"], 0, c["tol"], c["max_iter"], c["sign_seed"], elem_part=ep)
        for p in parts:
            if c.get("ud", 0.0) != 0.0:                  # non-zero Dirichlet values on the fixed z-dofs, as for the brick cases
                fixed = p["LocFixedDof"]
                ud = np.zeros(p["NDOF"])
                zf = fixed[fixed % 3 == 2]
                ud[zf] = c["ud"] * (1.0 + 0.25 * np.sin(p["DofVector"][zf].astype(float)))
                p["Ud"] = ud
        return mesh, parts
    b = Brick(c["N"], seed=0, n_types=c["n_types"])
    parts = make_parts(b, block_partition(b, *c["grid"]), tol=c["tol"], max_iter=c["max_iter"])
    for p in parts:
        if c.get("ud", 0.0) != 0.0:
            fixed = p["LocFixedDof"]
            ud = np.zeros(p["NDOF"])
            zf = fixed[fixed % 3 == 2]
            ud[zf] = c["ud"] * (1.0 + 0.25 * np.sin(p["DofVector"][zf].astype(float)))
            p["Ud"] = ud
        if c.get("negate_ck"):                       # negative-definite operator -> p.Ap <= 0 -> Flag 4
            for g in p["SubDomainData"]["StrucDataList"]:
                g["ElemList_Ck"] = -g["ElemList_Ck"]
        if c.get("isolated_node"):                   # a loaded free node that no element touches: diag 0 -> 1/0 = inf -> Flag 2
            n0 = p["NDOF"]
            p["NDOF"] = n0 + 3
            p["NNode"] += 1
            for key in ("RefLoadVector", "Ud", "Vd", "Un", "DofWeightVector"):
                p[key] = np.concatenate([p[key], np.array([0.0, 0.0, -1.0]) if key == "RefLoadVector" else
                                         (np.ones(3) if key == "DofWeightVector" else np.zeros(3))])
            p["DofVector"] = np.concatenate([p["DofVector"], p["DofVector"][-1] + 1 + np.arange(3)])
            p["NodeIdVector"] = np.concatenate([p["NodeIdVector"], [p["NodeIdVector"][-1] + 1]])
            p["NodeCoordVec"] = np.concatenate([p["NodeCoordVec"], [0.0, 0.0, float(c["N"])]])
            p["LocDofEff"] = np.concatenate([p["LocDofEff"], n0 + np.arange(3)])
            p["DofWeightVector_Eff"] = p["DofWeightVector"][p["LocDofEff"]]
            p["GlobData"]["GlobNDof"] += 3
            p["GlobData"]["GlobNDofEff"] += 3
        if c.get("zero_rhs"):
            p["RefLoadVector"] = np.zeros(p["NDOF"])
        if c.get("good_x0"):
            gdir = golden_dir or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
            g = np.load(os.path.join(gdir, c["good_x0"] + ".npz"))
            p["Un"] = g["Un"][p["DofVector"]].copy()
    return b, parts


def probe_vector(brick, seed=7):
    """Seeded global vector used for the mat-vec probe of every case (brick or octree mesh: .n_dof)."""
    return np.random.default_rng(seed).standard_normal(brick.n_dof)


def probe_for(brick, parts, seed=7):
    """Probe vector covering every global dof id of the parts (the isolated-node case has 3 extra dofs)."""
    ng = max(brick.n_dof, max(int(p["DofVector"].max()) + 1 for p in parts))
    return np.concatenate([probe_vector(brick, seed), np.ones(ng - brick.n_dof)])
