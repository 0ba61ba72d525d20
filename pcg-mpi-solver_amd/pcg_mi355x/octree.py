"""Synthetic two-level (2:1 graded) octree mesh with hanging nodes, in the reference's data model.

BASELINE configs speak of "octree" meshes: the reference's elements are octree cells whose 2:1
transitions carry hanging nodes, stored as PATTERN TYPES with more than 8 nodes and their own
`Ke[type]` (partition_mesh.py:443-491,576-581; `NNodes = nd/3`, :581).  The reference's pattern
library (`Ke.mat`) comes from an external pre-processor and is not in the repository, so this module
builds a small, self-consistent stand-in that exercises the same code paths:

  * a fine region of unit hex8 cells (z < zf) under a coarse region of size-2 hex8 cells,
  * the coarse cells that sit on the fine region are TRANSITION cells: 8 corners + the 5 hanging nodes
    of their lower face (4 edge mid-points + face centre) = 13 nodes, nd = 39.

Pattern matrices: the size-2 hex8 matrix is 2 x the unit one (3-D elasticity: K ~ h); the transition
pattern is obtained by splitting the coarse cell into its 8 unit sub-cells (27 nodes), assembling them
and constraining every sub-cell node that is not one of the 13 kept nodes to the trilinear
interpolation of the 8 corners (K_pat = C^T K_27 C).  The result is conforming (fine face matches the
fine cells node by node, the other faces are bilinear like the neighbouring coarse cells), symmetric
positive semi-definite with exactly the 6 rigid-body modes.

Three pattern types result: 0 = hex8 (unit, Ck = material), 0 again for coarse cells with Ck doubled,
1 = transition (nd = 39).  Host-side set-up only.
"""
from __future__ import annotations

import numpy as np

from .brick import hex8_stiffness, glob_settings

__all__ = ["transition_stiffness", "TwoLevelMesh", "make_octree_parts"]


def _sub27():
    """27 lattice points of a size-2 cell, index = x + 3y + 9z, and the 8 unit sub-cells' node lists."""
    cells = []
    for cz in range(2):
        for cy in range(2):
            for cx in range(2):
                cells.append([(cx + (a & 1)) + 3 * (cy + ((a >> 1) & 1)) + 9 * (cz + ((a >> 2) & 1)) for a in range(8)])
    return cells


# kept nodes of the transition pattern, as (x, y, z) in {0,1,2}^3: 8 corners, then the lower-face hanging nodes
KEPT = [(0, 0, 0), (2, 0, 0), (0, 2, 0), (2, 2, 0), (0, 0, 2), (2, 0, 2), (0, 2, 2), (2, 2, 2),
        (1, 0, 0), (0, 1, 0), (2, 1, 0), (1, 2, 0), (1, 1, 0)]


def transition_stiffness():
    """39 x 39 matrix of the transition pattern (node-major dofs, node order = KEPT)."""
    Ke = hex8_stiffness()
    K27 = np.zeros((81, 81))
    for nodes in _sub27():
        idx = np.array([3 * n + d for n in nodes for d in range(3)])
        K27[np.ix_(idx, idx)] += Ke
    kept_idx = {p: k for k, p in enumerate(KEPT)}
    Cn = np.zeros((27, 13))                     # node-level constraint: u_27 = Cn u_13
    for z in range(3):
        for y in range(3):
            for x in range(3):
                n = x + 3 * y + 9 * z
                if (x, y, z) in kept_idx:
                    Cn[n, kept_idx[(x, y, z)]] = 1.0
                else:                           # trilinear interpolation of the 8 corners
                    for c in range(8):
                        cx, cy, cz = 2 * (c & 1), 2 * ((c >> 1) & 1), 2 * ((c >> 2) & 1)
                        w = (1 - abs(x - cx) / 2) * (1 - abs(y - cy) / 2) * (1 - abs(z - cz) / 2)
                        Cn[n, kept_idx[(cx, cy, cz)]] += w
    C = np.kron(Cn, np.eye(3))
    K = C.T @ K27 @ C
    return 0.5 * (K + K.T)


class TwoLevelMesh:
    """nx x ny x nzf unit cells below, (nx/2) x (ny/2) x nzc size-2 cells above (nx, ny even)."""

    def __init__(self, nx, ny, nzf, nzc, seed=0):
        assert nx % 2 == 0 and ny % 2 == 0 and nzf >= 1 and nzc >= 1
        self.nx, self.ny, self.nzf, self.nzc = nx, ny, nzf, nzc
        X, Y, Z = nx + 1, ny + 1, nzf + 2 * nzc + 1
        self.dims = (X, Y, Z)
        lat = lambda x, y, z: (z * Y + y) * X + x                                       # noqa: E731
        rng = np.random.default_rng(seed)

        def grid(ni, nj, nk):                       # cell origins, x fastest
            k, j, i = np.meshgrid(np.arange(nk), np.arange(nj), np.arange(ni), indexing="ij")
            return i.ravel(), j.ravel(), k.ravel()
        i, j, k = grid(nx, ny, nzf)
        fine = np.stack([lat(i + (a & 1), j + ((a >> 1) & 1), k + ((a >> 2) & 1)) for a in range(8)], 1)
        ic, jc, kc = grid(nx // 2, ny // 2, nzc)
        x0, y0, z0 = 2 * ic, 2 * jc, nzf + 2 * kc
        first = kc == 0
        trans = np.stack([lat(x0[first] + px, y0[first] + py, z0[first] + pz) for (px, py, pz) in KEPT], 1)
        rest = ~first
        coarse = np.stack([lat(x0[rest] + 2 * (a & 1), y0[rest] + 2 * ((a >> 1) & 1), z0[rest] + 2 * ((a >> 2) & 1))
                           for a in range(8)], 1)
        self.cells = {"fine": np.array(fine, np.int64).reshape(-1, 8), "coarse": np.array(coarse, np.int64).reshape(-1, 8),
                      "trans": np.array(trans, np.int64).reshape(-1, 13)}
        used = np.unique(np.concatenate([v.ravel() for v in self.cells.values()]))
        self.lattice_of_node = used                                                     # global node id -> lattice id
        self.n_node = len(used)
        self.n_dof = 3 * self.n_node
        remap = np.full(X * Y * Z, -1, np.int64)
        remap[used] = np.arange(self.n_node)
        self.cells = {k: remap[v] for k, v in self.cells.items()}
        self.coords = np.stack([used % X, (used // X) % Y, used // (X * Y)], 1).astype(float)
        # groups: type 0 = hex8 (fine cells Ck = material, coarse cells Ck = 2 * material), type 1 = transition
        mat = lambda n: np.where(rng.random(n) < 0.5, 1.0, 3.0)                          # noqa: E731
        self.group_nodes = [np.concatenate([self.cells["fine"], self.cells["coarse"]]), self.cells["trans"]]
        self.group_ck = [np.concatenate([mat(len(self.cells["fine"])), 2.0 * mat(len(self.cells["coarse"]))]),
                         mat(len(self.cells["trans"]))]
        self.group_ke = [hex8_stiffness(), transition_stiffness()]
        # element centroids (for geometric partitioning)
        self.group_centroid = [self.coords[g].mean(axis=1) for g in self.group_nodes]
        self.fixed_nodes = np.flatnonzero(self.coords[:, 2] == 0)
        self.top_nodes = np.flatnonzero(self.coords[:, 2] == self.coords[:, 2].max())

    def load_vector(self):
        F = np.zeros(self.n_dof)
        F[3 * self.top_nodes + 2] = -1.0
        return F


def make_octree_parts(mesh: TwoLevelMesh, n_parts=1, axis=0, tol=1e-7, max_iter=10000, sign_seed=None):
    """RefMeshPart dicts (same keys as brick.make_parts) for `n_parts` slabs along `axis` (by element centroid).
    sign_seed: give every pattern a random sign frame (Ke_t = D Ke D, mask undone per element, like brick.py)."""
    F = mesh.load_vector()
    fixed = np.zeros(mesh.n_dof, bool)
    fixed[(3 * mesh.fixed_nodes[:, None] + np.arange(3)).ravel()] = True
    ext = mesh.coords[:, axis].max()
    part_of = [np.minimum((c[:, axis] / ext * n_parts).astype(int), n_parts - 1) for c in mesh.group_centroid]
    flips = [np.zeros(3 * g.shape[1], bool) for g in mesh.group_nodes]
    if sign_seed is not None:
        r = np.random.default_rng(sign_seed)
        flips = [r.random(3 * g.shape[1]) < 0.4 for g in mesh.group_nodes]
    masks = []
    for pid in range(n_parts):
        m = np.zeros(mesh.n_node, bool)
        for g, po in zip(mesh.group_nodes, part_of):
            m[g[po == pid].ravel()] = True
        masks.append(m)
    parts = []
    for pid in range(n_parts):
        node_ids = np.flatnonzero(masks[pid])
        loc = np.full(mesh.n_node, -1, np.int64)
        loc[node_ids] = np.arange(len(node_ids))
        dof_ids = (3 * node_ids[:, None] + np.arange(3)).ravel()
        groups = []
        for t, (g, po, ck, ke, fl) in enumerate(zip(mesh.group_nodes, part_of, mesh.group_ck, mesh.group_ke, flips)):
            sel = np.flatnonzero(po == pid)
            if len(sel) == 0:
                continue
            ln = loc[g[sel]]                                                              # (ne, nn)
            tbl = np.ascontiguousarray((3 * ln[:, :, None] + np.arange(3)).reshape(len(sel), -1).T)
            d = np.where(fl, -1.0, 1.0)
            ke_t = ke * d[:, None] * d[None, :]
            groups.append({"ElemTypeId": t, "ElemList_LocDofVector": tbl, "ElemList_LocDofVector_Flat": tbl.ravel(),
                           "ElemList_LocNodeIdVector": np.ascontiguousarray(ln.T),
                           "ElemList_SignVector": np.ascontiguousarray(np.broadcast_to(fl[:, None], tbl.shape)),
                           "ElemList_Ck": ck[sel].copy(), "ElemStiffMat": ke_t, "ElemDiagStiffMat": np.diag(ke_t).copy(),
                           "N_Elem": len(sel), "NNodes": g.shape[1]})
        flat = np.concatenate([g["ElemList_LocDofVector_Flat"] for g in groups])
        n = 3 * len(node_ids)
        gd = {"GlobNDof": mesh.n_dof, "GlobNDofEff": int(mesh.n_dof - fixed.sum()), "GlobNFixedDof": int(fixed.sum()),
              "GlobNNode": mesh.n_node, "MaxIter": int(max_iter), "Tol": float(tol), "TimeStepDelta": [0, 1], "TimeStepCount": 1,
              "FintCalcMode": "outbin", "N_TotalMshPrt": n_parts,
              "MP_TimeRecData": {"dT_FileRead": 0.0, "dT_Calc": 0.0, "dT_CommWait": 0.0, "dT_CalcList": [], "dT_CommWaitList": [],
                                 "TimeStepCountList": [], "t0": 0.0},
              "TimeList_Flag": np.zeros(2), "TimeList_RelRes": np.zeros(2), "TimeList_Iter": np.zeros(2)}
        p = {"Id": pid, "SubDomainData": {"StrucDataList": groups, "MixedDataList": {}}, "NDOF": n, "NNode": len(node_ids),
             "DofVector": dof_ids, "NodeIdVector": node_ids, "RefLoadVector": F[dof_ids], "Ud": np.zeros(n), "Vd": np.zeros(n),
             "NodeCoordVec": mesh.coords[node_ids].ravel(), "LocDofEff": np.flatnonzero(~fixed[dof_ids]).astype(np.int64),
             "LocFixedDof": np.flatnonzero(fixed[dof_ids]).astype(np.int64), "DofEff": dof_ids[~fixed[dof_ids]],
             "Flat_ElemLocDof": flat, "NCountDof": len(flat), "NbrMPIdVector": [], "OvrlpLocalDofVecList": [],
             "OvrlpLocalNodeIdVecList": [], "DofWeightVector": np.ones(n), "NodeWeightVector": np.ones(len(node_ids)),
             "MPList_RefPlotDofIndicesList": [], "GlobData": gd}
        parts.append(p)
    ref_dir = np.arange(3)[:, None]
    for p in parts:                                                     # partition_mesh.py:817-887
        for qid in range(n_parts):
            if qid == p["Id"]:
                continue
            ov = np.flatnonzero(masks[p["Id"]] & masks[qid])
            if len(ov) == 0:
                continue
            l = np.searchsorted(p["NodeIdVector"], ov)
            p["OvrlpLocalNodeIdVecList"].append(l)
            p["OvrlpLocalDofVecList"].append((3 * l + ref_dir).T.ravel())
            p["NbrMPIdVector"].append(qid)
            if p["Id"] > qid:
                p["DofWeightVector"][p["OvrlpLocalDofVecList"][-1]] = 0
                p["NodeWeightVector"][l] = 0
    for p in parts:
        p["Un"] = np.zeros(p["NDOF"])
        p["DofWeightVector_Eff"] = p["DofWeightVector"][p["LocDofEff"]]
    return parts
