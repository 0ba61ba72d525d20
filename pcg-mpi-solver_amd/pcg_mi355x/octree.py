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

__all__ = ["transition_stiffness", "TwoLevelMesh", "make_octree_parts", "pattern_stiffness", "GradedOctreeMesh", "HANG_POS"]


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
        nf = len(self.cells["fine"])
        self.group_level = [np.concatenate([np.ones(nf), 2.0 * np.ones(len(self.cells["coarse"]))]), 2.0 * np.ones(len(self.cells["trans"]))]
        # element centroids (for geometric partitioning)
        self.group_centroid = [self.coords[g].mean(axis=1) for g in self.group_nodes]
        self.fixed_nodes = np.flatnonzero(self.coords[:, 2] == 0)
        self.top_nodes = np.flatnonzero(self.coords[:, 2] == self.coords[:, 2].max())

    def load_vector(self):
        F = np.zeros(self.n_dof)
        F[3 * self.top_nodes + 2] = -1.0
        return F


def bisect_elements(mesh, n_parts):
    """Element -> part id by recursive coordinate bisection of the element centroids (the stand-in for METIS,
    partition.geometric_partition; elements in group order), as a list of arrays per pattern group."""
    from .partition import geometric_partition
    part = geometric_partition({"sctrs": np.concatenate(mesh.group_centroid)}, n_parts)
    cuts = np.cumsum([len(c) for c in mesh.group_centroid])[:-1]
    return np.split(part, cuts)


def make_octree_parts(mesh, n_parts=1, axis=0, tol=1e-7, max_iter=10000, sign_seed=None, elem_part=None, only=None):
    """RefMeshPart dicts (same keys as brick.make_parts) for `n_parts` slabs along `axis` (by element centroid), or for the
    element -> part map `elem_part` (list of arrays per pattern group, e.g. bisect_elements()).  mesh: TwoLevelMesh or
    GradedOctreeMesh.  sign_seed: give every pattern a random sign frame (Ke_t = D Ke D, mask undone per element).
    only: part ids to build (default all; a rank of a multi-GPU job builds its own - the other parts' node masks, which the
    overlap lists need, are cheap)."""
    F = mesh.load_vector()
    fixed = np.zeros(mesh.n_dof, bool)
    fixed[(3 * mesh.fixed_nodes[:, None] + np.arange(3)).ravel()] = True
    ext = mesh.coords[:, axis].max()
    part_of = [np.minimum((c[:, axis] / ext * n_parts).astype(int), n_parts - 1) for c in mesh.group_centroid]
    if elem_part is not None:
        part_of = [np.asarray(v, np.int64) for v in elem_part]
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
    for pid in (range(n_parts) if only is None else [int(k) for k in only]):
        node_ids = np.flatnonzero(masks[pid])
        loc = np.full(mesh.n_node, -1, np.int64)
        loc[node_ids] = np.arange(len(node_ids))
        dof_ids = (3 * node_ids[:, None] + np.arange(3)).ravel()
        groups = []
        comps = getattr(mesh, "group_comp", None) or [None] * len(flips)
        oflips = getattr(mesh, "group_flip", None) or [None] * len(flips)
        for t, (g, po, ck, ke, fl) in enumerate(zip(mesh.group_nodes, part_of, mesh.group_ck, mesh.group_ke, flips)):
            sel = np.flatnonzero(po == pid)
            if len(sel) == 0:
                continue
            ln = loc[g[sel]]                                                              # (ne, nn)
            # slot 3 l + c of the type = component comp[e, c] of local node l (x, y, z order unless the mesh orients its types)
            cp = np.arange(3)[None, :] if comps[t] is None else comps[t][sel]
            tbl = np.ascontiguousarray((3 * ln[:, :, None] + cp[:, None, :]).reshape(len(sel), -1).T)
            sgn = np.broadcast_to(fl[:, None], tbl.shape)
            if oflips[t] is not None:
                sgn = sgn ^ np.tile(oflips[t][sel], (1, g.shape[1])).T
            d = np.where(fl, -1.0, 1.0)
            ke_t = ke * d[:, None] * d[None, :]
            groups.append({"ElemTypeId": t, "ElemList_LocDofVector": tbl, "ElemList_LocDofVector_Flat": tbl.ravel(),
                           "ElemList_LocNodeIdVector": np.ascontiguousarray(ln.T),
                           "ElemList_SignVector": np.ascontiguousarray(sgn),
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


# ---------------------------------------------------------------------------------------------------------------------
# Multi-level, 2:1-balanced octree mesh (round 3).  The reference's meshes are octree cells of many
# sizes whose hanging nodes make PATTERN TYPES (partition_mesh.py:443-491 groups the elements by `Type`, :576-581 reads one
# Ke per type, NNodes = nd / 3; up to 144 types, :1074).  This generator builds such a mesh around a spherical surface:
# cells are refined towards the surface over `levels` levels, the tree is balanced across faces, edges AND corners, and
# every coarse cell keeps exactly those of its 18 candidate hanging positions (12 edge mid-points, 6 face centres) at which
# a finer neighbour has a corner.  A pattern type = one 18-bit mask of kept positions: 9 to 26 nodes, nd = 27 ... 78.
# ---------------------------------------------------------------------------------------------------------------------
# the 18 candidate hanging positions of a cell in half-edge units, ascending position index a + 3 b + 9 c
HANG_POS = [(a, b, c) for c in range(3) for b in range(3) for a in range(3) if (a == 1) + (b == 1) + (c == 1) in (1, 2)]
_CORNERS = [(2 * (a & 1), 2 * ((a >> 1) & 1), 2 * ((a >> 2) & 1)) for a in range(8)]


def pattern_stiffness(mask):
    """Element matrix of the pattern with hanging positions `mask` (bit q = HANG_POS[q] is a node), for a cell of edge 2:
    the cell is split into its 8 unit sub-cells (27 lattice points, trilinear hex8 each) and every lattice point that is NOT a
    node follows hierarchically from the nodes - an edge mid-point is the mean of its two corners, a face centre the mean of
    its four edge mid-points, the cell centre the mean of the six face centres (K = C^T K27 C).  The rule for a face uses
    data of that face only, so two cells that share a face (or an edge) interpolate it identically: conforming across every
    2:1 transition; symmetric positive semi-definite with exactly the six rigid-body modes (linear fields are reproduced).
    Node order: the 8 corners (a = dx + 2 dy + 4 dz), then the kept positions in ascending HANG_POS order."""
    Ke = hex8_stiffness()
    K27 = np.zeros((81, 81))
    for nodes in _sub27():
        idx = np.array([3 * n + d for n in nodes for d in range(3)])
        K27[np.ix_(idx, idx)] += Ke
    kept = list(_CORNERS) + [HANG_POS[q] for q in range(18) if (mask >> q) & 1]
    col = {pt: k for k, pt in enumerate(kept)}
    rows = {}

    def row(pt):
        if pt in rows:
            return rows[pt]
        r = np.zeros(len(kept))
        if pt in col:
            r[col[pt]] = 1.0
        else:
            ones = [d for d in range(3) if pt[d] == 1]
            if len(ones) == 1:                                  # edge mid-point: its two corners
                d = ones[0]
                for v in (0, 2):
                    q = list(pt); q[d] = v
                    r += 0.5 * row(tuple(q))
            elif len(ones) == 2:                                # face centre: the four edge mid-points of the face
                for d in ones:
                    for v in (0, 2):
                        q = list(pt); q[d] = v
                        r += 0.25 * row(tuple(q))
            else:                                               # cell centre: the six face centres
                for d in range(3):
                    for v in (0, 2):
                        q = list(pt); q[d] = v
                        r += row(tuple(q)) / 6.0
        rows[pt] = r
        return r
    Cn = np.stack([row((x, y, z)) for z in range(3) for y in range(3) for x in range(3)])      # (27, n_kept), lattice index x + 3y + 9z
    C = np.kron(Cn, np.eye(3))
    K = C.T @ K27 @ C
    return 0.5 * (K + K.T)


# ---------------------------------------------------------------------------------------------------------------------
# Pattern library by SYMMETRY CLASS (round 4).  The reference's library holds one matrix per cell pattern up to the 48
# rotations / reflections of the cube (Type 0 ... 143, partition_mesh.py:1074); an element of another orientation refers to the
# class's matrix through the ORDER of its dof list (LocDofVector, partition_mesh.py:453) and its sign vector (:455): canonical
# dof (node l, component c) is the physical dof (node g(l), component sigma(c)) with sign s(c), g the symmetry that carries
# the canonical pattern onto the element.  `GradedOctreeMesh(symmetry=True)` builds its types that way: ONE matrix per class,
# every element with its own dof order and signs - the three dofs of a node are then NOT in x, y, z order in the dof list.
# ---------------------------------------------------------------------------------------------------------------------
def cube_symmetries():
    """The 48 signed permutation matrices Q (as (perm, sign): (Q v)[d] = sign[d] * v[perm[d]]), identity first."""
    import itertools
    out = []
    for perm in itertools.permutations(range(3)):
        for sg in itertools.product((1, -1), repeat=3):
            out.append((tuple(perm), tuple(sg)))
    out.sort(key=lambda q: (q != ((0, 1, 2), (1, 1, 1)), q))
    return out


def _sym_point(q, pt):
    """Image of the lattice point pt in {0,1,2}^3 of a cell of edge 2 under the symmetry q (about the cell centre)."""
    perm, sg = q
    return tuple(sg[d] * (pt[perm[d]] - 1) + 1 for d in range(3))


def _sym_mask(q, mask):
    out = 0
    for b, pt in enumerate(HANG_POS):
        if (mask >> b) & 1:
            out |= 1 << HANG_POS.index(_sym_point(q, pt))
    return out


def canonical_pattern(mask):
    """(canonical mask, q): the smallest image of `mask` under the 48 symmetries and a symmetry q that carries the canonical
    pattern ONTO `mask` (first in cube_symmetries() order)."""
    syms = cube_symmetries()
    canon = min(_sym_mask(q, mask) for q in syms)
    for q in syms:
        if _sym_mask(q, canon) == mask:
            return canon, q
    raise AssertionError("no symmetry found")


def pattern_frame(mask):
    """For an element with hanging mask `mask`: (canonical mask, node_src, comp, flip) - canonical local node l sits at the
    element's position node_src[l] (0..7 = corner a, 8 + b = HANG_POS[b]); canonical component c is the physical component
    comp[c], negated when flip[c]  (u_canon = Q^T u_phys)."""
    canon, q = canonical_pattern(mask)
    perm, sg = q
    kept = list(_CORNERS) + [HANG_POS[b] for b in range(18) if (canon >> b) & 1]
    src = []
    for pt in kept:
        im = _sym_point(q, pt)
        src.append(_CORNERS.index(im) if im in _CORNERS else 8 + HANG_POS.index(im))
    comp = [perm.index(c) for c in range(3)]                     # Q[d][c] != 0  <=>  perm[d] == c
    flip = [sg[comp[c]] < 0 for c in range(3)]
    return canon, np.array(src), np.array(comp), np.array(flip)


class GradedOctreeMesh:
    """`roots` = (Rx, Ry, Rz) root cells of edge 2**levels lattice units, refined over `levels` levels towards the sphere
    |x - centre| = radius: a cell of edge s is split when its centre lies within band * s of the surface; then balanced 2:1
    over the full 26-neighbourhood.  Same attribute set as TwoLevelMesh (group_nodes / group_ck / group_ke / group_centroid /
    group_level, coords, fixed_nodes, top_nodes, load_vector) - make_octree_parts, mdf.model_from_octree and
    partition.partition_model take either.  Host-side set-up only (whole-array NumPy)."""

    def __init__(self, roots=(4, 4, 4), levels=3, centre=None, radius=None, band=1.0, seed=0, two_phase=True, symmetry=False):
        L = int(levels)
        self.symmetry = bool(symmetry)
        R = np.array(roots, np.int64)
        S0 = 1 << L
        dims = R * S0                                            # lattice extent (cells of edge 1)
        self.roots, self.levels, self.band = tuple(int(r) for r in R), L, float(band)
        c = np.array(centre if centre is not None else dims / 2.0, float)
        rho = float(radius if radius is not None else 0.3 * dims.min())
        self.centre, self.radius = c, rho

        def nx(l):
            return R << l                                        # cells per axis at level l

        def key(l, i, j, k):
            n = nx(l)
            return (k * n[1] + j) * n[0] + i

        def unkey(l, q):
            n = nx(l)
            return q % n[0], (q // n[0]) % n[1], q // (n[0] * n[1])

        def children(l, q):                                      # keys at level l + 1
            i, j, k = unkey(l, q)
            out = [key(l + 1, 2 * i + (a & 1), 2 * j + ((a >> 1) & 1), 2 * k + ((a >> 2) & 1)) for a in range(8)]
            return np.stack(out, 1).ravel()

        def parents(l, q):                                       # keys at level l - 1
            i, j, k = unkey(l, q)
            return np.unique(key(l - 1, i // 2, j // 2, k // 2))

        def n26(l, q):
            i, j, k = unkey(l, q)
            n = nx(l)
            out = []
            for dk in (-1, 0, 1):
                for dj in (-1, 0, 1):
                    for di in (-1, 0, 1):
                        if di == dj == dk == 0:
                            continue
                        ii, jj, kk = i + di, j + dj, k + dk
                        ok = (ii >= 0) & (ii < n[0]) & (jj >= 0) & (jj < n[1]) & (kk >= 0) & (kk < n[2])
                        out.append(key(l, ii[ok], jj[ok], kk[ok]))
            return np.unique(np.concatenate(out)) if out else np.zeros(0, np.int64)

        # 1. refinement towards the surface, top down
        I = []                                                   # internal (split) cells per level 0 .. L-1
        cand = np.arange(int(np.prod(R)), dtype=np.int64)
        for l in range(L):
            s = S0 >> l
            i, j, k = unkey(l, cand)
            ctr = (np.stack([i, j, k], 1) + 0.5) * s
            d = np.abs(np.linalg.norm(ctr - c, axis=1) - rho)
            split = cand[d < band * s]
            I.append(np.unique(split))
            cand = children(l, I[l])
        # 2. 2:1 balance over faces, edges and corners, one sweep from the finest level down: every neighbour of a split
        #    cell exists at that cell's level, i.e. its parent is split too
        for l in range(L - 1, 0, -1):
            need = np.union1d(n26(l, I[l]), I[l])
            I[l - 1] = np.union1d(I[l - 1], parents(l, need))
        # 3. leaves
        leaves = []                                              # (level, keys)
        exist = np.arange(int(np.prod(R)), dtype=np.int64)
        for l in range(L + 1):
            if l < L:
                leaves.append(np.setdiff1d(exist, I[l], assume_unique=True))
                exist = np.sort(children(l, I[l]))
            else:
                leaves.append(exist)
        self.leaves_per_level = [len(v) for v in leaves]
        # 4. nodes = the corners of all leaves, on the finest lattice
        X, Y, Z = (int(v) + 1 for v in dims)
        self.dims = (X, Y, Z)

        def lat(x, y, z):
            return (z * Y + y) * X + x
        org, size = [], []
        for l, q in enumerate(leaves):
            s = S0 >> l
            i, j, k = unkey(l, q)
            org.append(np.stack([i, j, k], 1) * s)
            size.append(np.full(len(q), s, np.int64))
        org = np.concatenate(org)
        size = np.concatenate(size)
        corners = np.stack([lat(org[:, 0] + size * (a & 1), org[:, 1] + size * ((a >> 1) & 1), org[:, 2] + size * ((a >> 2) & 1))
                            for a in range(8)], 1)
        used = np.unique(corners)
        self.lattice_of_node = used
        self.n_node = len(used)
        self.n_dof = 3 * self.n_node
        self.n_elem = len(org)
        # lattice point -> node id (-1: not a node), dense over the finest lattice: one table look-up per (element, position)
        # instead of a binary search each (the 10 M-dof mesh of bench.py: 2.7 M elements x 26 positions)
        node_of_lat = np.full(X * Y * Z, -1, np.int32)
        node_of_lat[used] = np.arange(len(used), dtype=np.int32)
        # 5. pattern of every leaf: which of its 18 hanging positions are nodes
        h = size // 2
        mask = np.zeros(len(org), np.int64)
        hang_ids = np.zeros((len(org), 18), np.int64)            # (only the entries whose mask bit is set are ever read)
        big = np.flatnonzero(size >= 2)
        ob, hb = org[big], h[big]
        for q, (a, b, cc) in enumerate(HANG_POS):
            pos = node_of_lat[lat(ob[:, 0] + a * hb, ob[:, 1] + b * hb, ob[:, 2] + cc * hb)]
            hit = pos >= 0
            w = big[hit]
            mask[w] |= 1 << q
            hang_ids[w, q] = pos[hit]
        cn = node_of_lat[corners].astype(np.int64)               # (E, 8) node ids
        del node_of_lat
        rng = np.random.default_rng(seed)
        mat = np.where(rng.random(len(org)) < 0.5, 1.0, 3.0) if two_phase else np.ones(len(org))    # two-phase scaling like brick.py
        ctr = org + size[:, None] / 2.0
        self.pattern_masks = [0] + sorted(int(m) for m in np.unique(mask) if m != 0)
        self.n_orientations = len(self.pattern_masks)
        self.group_nodes, self.group_ck, self.group_ke, self.group_centroid, self.group_level = [], [], [], [], []
        # per group (ne, 3) arrays or None: canonical component c of an element is its physical component group_comp[e, c], negated
        # where group_flip[e, c] (None = x, y, z order, no flips: always so without `symmetry`, and for the hex8 type)
        self.group_comp, self.group_flip = [], []
        if symmetry:
            # ONE type per symmetry class: the elements of every orientation of the class, nodes in the canonical pattern's order
            frames = {m: pattern_frame(m) for m in self.pattern_masks if m != 0}
            types = [(0, [0])] + [(cm, [m for m in self.pattern_masks if m != 0 and frames[m][0] == cm])
                                  for cm in sorted({f[0] for f in frames.values()})]
            all_pos = np.concatenate([cn, hang_ids], 1)                      # (E, 26): corner a at a, HANG_POS[b] at 8 + b
        else:
            types = [(m, [m]) for m in self.pattern_masks]
        for cm, members in types:
            sel = np.flatnonzero(np.isin(mask, members))
            comp = flip = None
            if cm == 0:
                nodes = cn[sel]
                ck = size[sel] * mat[sel]                        # 3-D elasticity: K ~ edge length
                ke = hex8_stiffness()
            elif not symmetry:
                bits = [q for q in range(18) if (cm >> q) & 1]
                nodes = np.concatenate([cn[sel], hang_ids[sel][:, bits]], 1)
                ck = (size[sel] / 2.0) * mat[sel]                # pattern_stiffness is the matrix of a cell of edge 2
                ke = pattern_stiffness(cm)
            else:
                nodes = np.zeros((len(sel), 8 + bin(cm).count("1")), np.int64)
                comp = np.zeros((len(sel), 3), np.int64)
                flip = np.zeros((len(sel), 3), bool)
                for m in members:
                    _, src, cp, fl = frames[m]
                    w = np.flatnonzero(mask[sel] == m)
                    nodes[w] = all_pos[sel[w]][:, src]
                    comp[w] = cp
                    flip[w] = fl
                ck = (size[sel] / 2.0) * mat[sel]
                ke = pattern_stiffness(cm)
            self.group_nodes.append(np.ascontiguousarray(nodes))
            self.group_ck.append(ck)
            self.group_ke.append(ke)
            self.group_centroid.append(ctr[sel])
            self.group_level.append(size[sel].astype(float))
            self.group_comp.append(comp)
            self.group_flip.append(flip)
        self.pattern_masks = [cm for cm, _ in types]
        self.coords = np.stack([used % X, (used // X) % Y, used // (X * Y)], 1).astype(float)
        self.fixed_nodes = np.flatnonzero(self.coords[:, 2] == 0)
        self.top_nodes = np.flatnonzero(self.coords[:, 2] == self.coords[:, 2].max())
        self.nnz = None

    def load_vector(self):
        F = np.zeros(self.n_dof)
        F[3 * self.top_nodes + 2] = -1.0
        return F

    def summary(self):
        """What the mesh looks like to the operators: leaves per level, pattern types with node and element counts."""
        return {"dofs": int(self.n_dof), "elements": int(self.n_elem), "levels": self.levels + 1,
                "leaves_per_level_coarse_to_fine": [int(v) for v in self.leaves_per_level],
                "pattern_types": len(self.pattern_masks),
                **({"pattern_orientations": int(self.n_orientations),
                    "pattern_note": "one type per class of the cube's 48 symmetries; an element carries its orientation in the order "
                                    "of its dof list and in its sign vector (the reference's pattern library: partition_mesh.py:453-455,1074)"}
                   if self.symmetry else {}),
                "nodes_per_element_max": int(max(g.shape[1] for g in self.group_nodes)),
                "elements_hex8": int(len(self.group_nodes[0])),
                "elements_with_hanging_nodes": int(self.n_elem - len(self.group_nodes[0])),
                "elements_by_node_count": {str(nn): int(sum(len(g) for g in self.group_nodes if g.shape[1] == nn))
                                           for nn in sorted({g.shape[1] for g in self.group_nodes})}}
