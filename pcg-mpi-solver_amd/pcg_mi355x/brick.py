"""Synthetic 3-D elastostatic input in the reference's data model.

The reference's demo input (data/concrete.zip) is absent from the checkout and its element
library `Ke.mat` comes from an external MATLAB pre-processor, so every workload here is the
synthetic brick of SURVEY 8(d): N x N x N nodes, (N-1)^3 unit hex8 elements, trilinear
stiffness (2x2x2 Gauss, E=1, nu=0.2), two-phase scaling Ck in {1,3} (seed 0), nodes on z=0
fixed, F_z=-1 on the top face.

`make_parts()` builds, for an element partition, the per-part `RefMeshPart` dicts with exactly
the keys and layouts the reference solver reads (reference: src/solver/partition_mesh.py
:257-268 local numbering, :350-398 nodal vectors, :443-491 type groups `(nd,Ne)` tables,
:576-581 element library, :817-887 overlap lists / Flat_ElemLocDof / ownership weights;
src/solver/pcg_solver.py:996-997 initial guess and DofWeightVector_Eff).  It is host-side set-up
(NumPy), not part of the GPU hot path.
"""
from __future__ import annotations

import numpy as np

__all__ = ["hex8_stiffness", "Brick", "make_parts", "block_partition", "glob_settings"]


def hex8_stiffness(E: float = 1.0, nu: float = 0.2, h: float = 1.0) -> np.ndarray:
    """24x24 stiffness of a trilinear cube, 2x2x2 Gauss.

    Local node a = dx + 2*dy + 4*dz (dx,dy,dz in {0,1}); DOF = 3*a + dir (node-major).
    """
    lam = E * nu / ((1 + nu) * (1 - 2 * nu))
    mu = E / (2 * (1 + nu))
    D = np.zeros((6, 6))
    D[:3, :3] = lam
    D[np.arange(3), np.arange(3)] += 2 * mu
    D[3:, 3:] = np.eye(3) * mu
    sgn = np.array([[2 * (a & 1) - 1, 2 * ((a >> 1) & 1) - 1, 2 * ((a >> 2) & 1) - 1] for a in range(8)], float)
    g = 1.0 / np.sqrt(3.0)
    Ke = np.zeros((24, 24))
    for gx in (-g, g):
        for gy in (-g, g):
            for gz in (-g, g):
                xi = np.array([gx, gy, gz])
                # dN_a/dxi_d on the reference cube [-1,1]^3, then physical (J = h/2 I)
                dN = np.empty((8, 3))
                for a in range(8):
                    f = 1 + sgn[a] * xi
                    dN[a, 0] = sgn[a, 0] * f[1] * f[2] / 8
                    dN[a, 1] = sgn[a, 1] * f[0] * f[2] / 8
                    dN[a, 2] = sgn[a, 2] * f[0] * f[1] / 8
                dN *= 2.0 / h
                B = np.zeros((6, 24))
                for a in range(8):
                    bx, by, bz = dN[a]
                    c = 3 * a
                    B[0, c] = bx
                    B[1, c + 1] = by
                    B[2, c + 2] = bz
                    B[3, c] = by
                    B[3, c + 1] = bx
                    B[4, c + 1] = bz
                    B[4, c + 2] = by
                    B[5, c] = bz
                    B[5, c + 2] = bx
                Ke += B.T @ D @ B * (h / 2) ** 3
    return 0.5 * (Ke + Ke.T)


class Brick:
    """Global description of the SURVEY 8(d) brick with N nodes per side."""

    def __init__(self, N: int, seed: int = 0, n_types: int = 1):
        assert N >= 2
        self.N = int(N)
        self.Ne1 = self.N - 1
        self.n_elem = self.Ne1 ** 3
        self.n_node = self.N ** 3
        self.n_dof = 3 * self.n_node
        rng = np.random.default_rng(seed)
        # element order: i fastest (x), then j, then k
        self.Ck = np.where(rng.random(self.n_elem) < 0.5, 1.0, 3.0)
        self.Ke = hex8_stiffness()
        # Octree-style pattern types: type t stores the element matrix in a sign-flipped
        # "pattern" frame, Ke_t = D_t Ke D_t, and the per-element sign mask undoes it (this is how
        # the reference's SignVector is used, pcg_solver.py:278-280).  n_types=1 -> mask all False.
        self.n_types = int(n_types)
        trng = np.random.default_rng(seed + 1000)
        self.type_flip = [np.zeros(24, bool)] + [trng.random(24) < 0.5 for _ in range(self.n_types - 1)]
        self.elem_type = (np.zeros(self.n_elem, np.int32) if n_types == 1
                          else trng.integers(0, n_types, self.n_elem).astype(np.int32))
        self.nnz = 9 * (3 * self.N - 2) ** 3

    # -- element -> global node table -------------------------------------------------------
    def elem_nodes(self, elem_ids: np.ndarray | None = None) -> np.ndarray:
        """(Ne, 8) global node ids, local node a = dx + 2 dy + 4 dz."""
        e = np.arange(self.n_elem, dtype=np.int64) if elem_ids is None else np.asarray(elem_ids, np.int64)
        n1, N = self.Ne1, self.N
        ei = e % n1
        ej = (e // n1) % n1
        ek = e // (n1 * n1)
        base = (ek * N + ej) * N + ei
        off = np.array([dx + N * dy + N * N * dz for dz in (0, 1) for dy in (0, 1) for dx in (0, 1)], np.int64)
        return base[:, None] + off[None, :]

    def node_elem_parts(self, node_ids: np.ndarray, elem_part: np.ndarray) -> np.ndarray:
        """(len(node_ids), 8) part ids of the up to eight elements around each node (-1 where the lattice ends): what a rank
        needs to find its neighbours from ITS OWN nodes only, without building any other part's node set."""
        n = np.asarray(node_ids, np.int64)
        N, n1 = self.N, self.Ne1
        i, j, k = n % N, (n // N) % N, n // (N * N)
        out = np.full((len(n), 8), -1, np.int32)
        for a in range(8):
            ei, ej, ek = i - (a & 1), j - ((a >> 1) & 1), k - ((a >> 2) & 1)
            ok = (ei >= 0) & (ei < n1) & (ej >= 0) & (ej < n1) & (ek >= 0) & (ek < n1)
            out[ok, a] = elem_part[((ek[ok] * n1) + ej[ok]) * n1 + ei[ok]]
        return out

    def load_vector(self) -> np.ndarray:
        F = np.zeros(self.n_dof)
        N = self.N
        top = np.arange((N - 1) * N * N, N * N * N, dtype=np.int64)
        F[3 * top + 2] = -1.0
        return F

    def fixed_dofs(self) -> np.ndarray:
        N = self.N
        bottom = np.arange(0, N * N, dtype=np.int64)
        return (3 * bottom[:, None] + np.arange(3)[None, :]).ravel()

    def type_Ke(self, t: int) -> np.ndarray:
        d = np.where(self.type_flip[t], -1.0, 1.0)
        return self.Ke * d[:, None] * d[None, :]


def block_partition(brick: Brick, px: int, py: int, pz: int) -> np.ndarray:
    """Element -> part id for a px x py x pz grid of element blocks (what METIS returns for a
    brick up to renumbering; mgmetis is not installed; reference: run_metis.py:88)."""
    n1 = brick.Ne1
    e = np.arange(brick.n_elem, dtype=np.int64)
    ei, ej, ek = e % n1, (e // n1) % n1, e // (n1 * n1)
    bx = np.minimum(ei * px // n1, px - 1)
    by = np.minimum(ej * py // n1, py - 1)
    bz = np.minimum(ek * pz // n1, pz - 1)
    return ((bz * py + by) * px + bx).astype(np.int32)


def default_grid(n_parts: int) -> tuple[int, int, int]:
    """1: 1x1x1, 2: slabs in z, 4: 2x2x1, 8: 2x2x2; else slabs."""
    return {1: (1, 1, 1), 2: (1, 1, 2), 4: (2, 2, 1), 8: (2, 2, 2)}.get(n_parts, (1, 1, n_parts))


def glob_settings(brick: Brick, tol: float = 1e-7, max_iter: int = 10000) -> dict:
    """The GlobData entries the solver reads (pcg_solver.py:46-52,121,127,131-132,368-374)."""
    n_fixed = len(brick.fixed_dofs())
    return {
        "GlobNDof": brick.n_dof,
        "GlobNDofEff": brick.n_dof - n_fixed,
        "GlobNFixedDof": n_fixed,
        "GlobNNode": brick.n_node,
        "GlobNElem": brick.n_elem,
        "MaxIter": int(max_iter),
        "Tol": float(tol),
        "TimeStepDelta": [0, 1],
        "TimeStepCount": 1,
        "FintCalcMode": "outbin",
        "MP_TimeRecData": {"dT_FileRead": 0.0, "dT_Calc": 0.0, "dT_CommWait": 0.0,
                           "dT_CalcList": [], "dT_CommWaitList": [], "TimeStepCountList": [], "t0": 0.0},
        "TimeList_Flag": np.zeros(2), "TimeList_RelRes": np.zeros(2), "TimeList_Iter": np.zeros(2),
    }


def make_parts(brick: Brick, elem_part: np.ndarray | None = None, tol: float = 1e-7,
               max_iter: int = 10000, index_dtype=np.int64, only=None) -> list[dict]:
    """RefMeshPart dicts, one per part (reference exports these per part: partition_mesh.py:1310-1317).

    only: iterable of part ids to build (default all) - a rank of a multi-GPU job builds just its own
    part; the neighbours' node sets it needs for the overlap lists are cheap boolean masks."""
    if elem_part is None:
        elem_part = np.zeros(brick.n_elem, np.int32)
    n_parts = int(elem_part.max()) + 1
    only = list(range(n_parts)) if only is None else [int(k) for k in only]
    F = brick.load_vector()
    fixed = np.zeros(brick.n_dof, bool)
    fixed[brick.fixed_dofs()] = True
    parts = []
    for pid in only:
        eids = np.flatnonzero(elem_part == pid)                      # ascending element ids
        gnodes = brick.elem_nodes(eids)                              # (Ne, 8)
        if n_parts == 1:
            node_ids = np.arange(brick.n_node, dtype=np.int64)
            lnodes = gnodes
        else:
            node_ids, inv = np.unique(gnodes, return_inverse=True)   # partition_mesh.py:257-268
            lnodes = inv.reshape(gnodes.shape).astype(np.int64)
        n_node = len(node_ids)
        dof_ids = (3 * node_ids[:, None] + np.arange(3)[None, :]).ravel()
        ldof = (3 * lnodes[:, :, None] + np.arange(3)[None, None, :]).reshape(len(eids), 24)
        etype = brick.elem_type[eids]
        groups = []
        for t in np.unique(etype):                                   # partition_mesh.py:443-491
            I = np.flatnonzero(etype == t)
            tbl = np.ascontiguousarray(ldof[I].T).astype(index_dtype)    # (nd, Ne), element-minor
            sign = np.ascontiguousarray(np.broadcast_to(brick.type_flip[t][:, None], tbl.shape))
            Ke_t = brick.type_Ke(int(t))
            groups.append({
                "ElemTypeId": int(t),
                "ElemList_LocDofVector": tbl,
                "ElemList_LocDofVector_Flat": tbl.ravel(),
                "ElemList_LocNodeIdVector": np.ascontiguousarray(lnodes[I].T),
                "ElemList_SignVector": sign,
                "ElemList_Ck": brick.Ck[eids[I]].copy(),
                "ElemStiffMat": Ke_t,                                # partition_mesh.py:577
                "ElemDiagStiffMat": np.diag(Ke_t).copy(),            # partition_mesh.py:578
                "ElemList_LocElemId": I,
                "N_Elem": len(I),
                "NNodes": 8,
            })
        flat = np.concatenate([g["ElemList_LocDofVector_Flat"] for g in groups])   # :849-860
        loc_fixed = np.flatnonzero(fixed[dof_ids])
        loc_eff = np.flatnonzero(~fixed[dof_ids])                    # :350-351
        gd = glob_settings(brick, tol, max_iter)
        gd["N_TotalMshPrt"] = n_parts
        part = {
            "Id": pid,
            "SubDomainData": {"StrucDataList": groups, "MixedDataList": {}},
            "NDOF": 3 * n_node, "NNode": n_node, "NElem": len(eids),
            "DofVector": dof_ids, "NodeIdVector": node_ids, "ElemIdVector": eids,
            "RefLoadVector": F[dof_ids], "Ud": np.zeros(3 * n_node), "Vd": np.zeros(3 * n_node),
            "NodeCoordVec": np.stack([node_ids % brick.N, (node_ids // brick.N) % brick.N, node_ids // (brick.N * brick.N)],
                                     1).astype(float).ravel(),                      # partition_mesh.py:357
            "DofEff": dof_ids[loc_eff], "LocDofEff": loc_eff.astype(np.int64),
            "LocFixedDof": loc_fixed.astype(np.int64),
            "Flat_ElemLocDof": flat, "NCountDof": len(flat),
            "NbrMPIdVector": [], "OvrlpLocalDofVecList": [], "OvrlpLocalNodeIdVecList": [],
            "DofWeightVector": np.ones(3 * n_node), "NodeWeightVector": np.ones(n_node),
            "MPList_RefPlotDofIndicesList": [],
            "GlobData": gd,
        }
        parts.append(part)
    # neighbours, overlap lists, ownership weights (partition_mesh.py:817-887); candidate order is
    # ascending part id (identify_PotentialNeighbours loops `for MP_Id_j in range(N_TotalMeshPart)`)
    # Every part finds its neighbours from ITS OWN nodes: the parts of the (up to eight) elements around each of them
    # (a rank of a multi-GPU job builds one part; no other part's node set is ever formed).
    ref_dir = np.arange(3)[:, None]
    for p in parts:
        adj = brick.node_elem_parts(p["NodeIdVector"], elem_part) if n_parts > 1 else None
        for qid in (np.unique(adj) if n_parts > 1 else []):
            if qid == p["Id"] or qid < 0:
                continue
            loc = np.flatnonzero((adj == qid).any(axis=1))                        # = intersect1d (:822) of the two node sets, ascending
            if len(loc) == 0:
                continue
            p["OvrlpLocalNodeIdVecList"].append(loc)
            p["OvrlpLocalDofVecList"].append((3 * loc + ref_dir).T.ravel())      # :826 node-major
            p["NbrMPIdVector"].append(int(qid))
            if p["Id"] > qid:                                                     # :885-887
                p["DofWeightVector"][p["OvrlpLocalDofVecList"][-1]] = 0
                p["NodeWeightVector"][loc] = 0
        p["N_NbrDof"] = int(sum(len(v) for v in p["OvrlpLocalDofVecList"]))
    for p in parts:                                                  # pcg_solver.py:996-997
        p["Un"] = np.zeros(p["NDOF"])
        p["DofWeightVector_Eff"] = p["DofWeightVector"][p["LocDofEff"]]
    return parts
