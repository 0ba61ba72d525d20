"""The multi-level octree mesh generator (pcg_mi355x/octree.py GradedOctreeMesh, round 3): structural properties of the mesh
and of its pattern matrices, on the CPU.  Parity of the solver on such meshes: the reference-generated fixtures goct_p1 / goct_p4
(test_oracle_golden, test_driver_cpu, test_dist_gloo, ...) and, at 1 M dof on the GPU, test_gpu_parity.test_graded_octree_1m_dof."""
import numpy as np
import pytest

import pcg_oracle
from pcg_mi355x.brick import hex8_stiffness
from pcg_mi355x.octree import GradedOctreeMesh, HANG_POS, bisect_elements, make_octree_parts, pattern_stiffness


@pytest.fixture(scope="module")
def mesh():
    return GradedOctreeMesh((4, 4, 4), 3, band=1.0, two_phase=False)


def test_patterns_are_consistent_element_matrices():
    """mask 0 is the hex8 matrix of a cell of edge 2; every pattern is symmetric positive semi-definite with exactly the six
    rigid-body modes (translations AND rotations: the hierarchical constraints reproduce linear fields)."""
    assert np.abs(pattern_stiffness(0) - 2.0 * hex8_stiffness()).max() < 1e-14
    rng = np.random.default_rng(0)
    masks = [1 << q for q in range(18)] + [int(m) for m in rng.integers(1, 1 << 18, 12)] + [(1 << 18) - 1]
    for m in masks:
        K = pattern_stiffness(m)
        nn = 8 + bin(m).count("1")
        assert K.shape == (3 * nn, 3 * nn) and np.abs(K - K.T).max() == 0.0
        w = np.linalg.eigvalsh(K)
        assert w[0] > -1e-12 and np.sum(np.abs(w) < 1e-10) == 6, (m, w[:8])
        pts = np.array([(2 * (a & 1), 2 * ((a >> 1) & 1), 2 * ((a >> 2) & 1)) for a in range(8)] +
                       [HANG_POS[q] for q in range(18) if (m >> q) & 1], float)
        rot = np.stack([-pts[:, 1], pts[:, 0], np.zeros(nn)], 1).ravel()          # rigid rotation about z
        assert np.abs(K @ rot).max() < 1e-12


def test_mesh_is_two_to_one_balanced_and_patterns_complete(mesh):
    """Every mesh node that lies in the closed box of a leaf is one of its 8 corners or one of its kept hanging positions:
    no leaf has a neighbour more than one level finer across a face, an edge or a corner, and no hanging node is missed."""
    X, Y, Z = mesh.dims
    occ = np.zeros((X, Y, Z), bool)
    c = mesh.coords.astype(int)
    occ[c[:, 0], c[:, 1], c[:, 2]] = True
    levels = 0
    for nodes, size in zip(mesh.group_nodes, mesh.group_level):
        for e in range(0, len(nodes), max(1, len(nodes) // 400)):                  # a sample of every pattern type
            s = int(size[e])
            lo = mesh.coords[nodes[e, :8]].min(axis=0).astype(int)
            inside = int(occ[lo[0]:lo[0] + s + 1, lo[1]:lo[1] + s + 1, lo[2]:lo[2] + s + 1].sum())
            assert inside == nodes.shape[1], (inside, nodes.shape[1], s)
        levels = max(levels, int(np.log2(size.max())))
    assert 2 <= levels <= mesh.levels and len(mesh.pattern_masks) > 20      # (the coarsest cells may all have been split by the balance)
    assert sum(len(g) for g in mesh.group_nodes) == mesh.n_elem


def test_patch_test_on_the_assembled_mesh(mesh):
    """A linear displacement field produces zero internal forces at every interior node (conforming across all 2:1 transitions,
    faces and edges); a rigid translation produces none anywhere."""
    P = make_octree_parts(mesh, 1)[0]
    xyz = mesh.coords
    u = np.zeros(mesh.n_dof)
    u[0::3] = 0.01 * xyz[:, 0] + 0.02 * xyz[:, 1]; u[1::3] = -0.03 * xyz[:, 2]; u[2::3] = 0.005 * xyz[:, 0]
    f = pcg_oracle.matvec_local(P, u).reshape(-1, 3)
    interior = np.all((xyz > 0) & (xyz < np.array(mesh.dims) - 1), axis=1)
    assert np.abs(f[interior]).max() < 1e-12 and np.abs(f[~interior]).max() > 1e-3
    t = np.zeros(mesh.n_dof); t[1::3] = 1.0
    assert np.abs(pcg_oracle.matvec_local(P, t)).max() < 1e-12


def test_one_pattern_type_per_symmetry_class():
    """GradedOctreeMesh(symmetry=True): the pattern types are the classes of the hanging masks under the cube's 48 symmetries (the
    reference's library form, partition_mesh.py:1074: Type 0 ... 143); an element's orientation is carried by the order of its dof
    list and by its sign vector (:453-455).  The frame reproduces the pattern's own matrix (to rounding), the three dofs of a node
    stay together but not in x, y, z order, and the operator of the mesh is the one of the mesh with one type per ORIENTATION."""
    from pcg_mi355x.octree import cube_symmetries, pattern_frame
    assert len(cube_symmetries()) == 48 and cube_symmetries()[0] == ((0, 1, 2), (1, 1, 1))
    plain = GradedOctreeMesh((4, 4, 4), 3, band=1.2)
    sym = GradedOctreeMesh((4, 4, 4), 3, band=1.2, symmetry=True)
    assert len(plain.pattern_masks) == sym.n_orientations == 95 and len(sym.pattern_masks) == 8 and sym.n_elem == plain.n_elem
    for m in plain.pattern_masks[1::7]:
        canon, src, comp, flip = pattern_frame(m)
        bits = [b for b in range(18) if (m >> b) & 1]
        pos = {**{a: a for a in range(8)}, **{8 + b: 8 + i for i, b in enumerate(bits)}}
        idx = np.array([3 * pos[src[l]] + comp[c] for l in range(len(src)) for c in range(3)])
        sg = np.array([-1.0 if flip[c] else 1.0 for l in range(len(src)) for c in range(3)])
        Kp = pattern_stiffness(m)
        K2 = np.zeros_like(Kp)
        K2[np.ix_(idx, idx)] = pattern_stiffness(canon) * sg[:, None] * sg[None, :]
        assert np.abs(K2 - Kp).max() <= 4e-16 * np.abs(Kp).max()
    A, B = make_octree_parts(plain, 1)[0], make_octree_parts(sym, 1, sign_seed=4)[0]
    reordered = 0
    for g in B["SubDomainData"]["StrucDataList"]:
        t = g["ElemList_LocDofVector"]
        assert np.array_equal(t[0::3] // 3, t[1::3] // 3) and np.array_equal(t[0::3] // 3, t[2::3] // 3)      # node-blocked
        assert np.array_equal(np.sort(np.stack([t[0::3] % 3, t[1::3] % 3, t[2::3] % 3]), axis=0)[:, 0], np.tile(np.arange(3)[:, None], (1, t.shape[1])))
        reordered += int((t[:3] % 3 != np.arange(3)[:, None]).any(axis=0).sum())
    assert reordered > 100
    x = np.random.default_rng(3).standard_normal(plain.n_dof)
    ya, yb = pcg_oracle.matvec_local(A, x), pcg_oracle.matvec_local(B, x)
    assert np.abs(ya - yb).max() <= 1e-14 * np.abs(ya).max()
    assert np.abs(pcg_oracle.matvec_local(A, None, "Preconditioner") - pcg_oracle.matvec_local(B, None, "Preconditioner")).max() <= 1e-14 * np.abs(ya).max()


@pytest.mark.parametrize("n_parts", [2, 5, 8])
def test_bisection_parts_cover_the_mesh(mesh, n_parts):
    ep = bisect_elements(mesh, n_parts)
    sizes = np.bincount(np.concatenate(ep), minlength=n_parts)
    assert sizes.sum() == mesh.n_elem and sizes.min() > 0 and sizes.max() <= 1.3 * mesh.n_elem / n_parts + 1
    parts = make_octree_parts(mesh, n_parts, elem_part=ep)
    owned = np.zeros(mesh.n_dof, int)
    for p in parts:
        owned[p["DofVector"][p["DofWeightVector"] == 1]] += 1
        assert p["NbrMPIdVector"] == sorted(p["NbrMPIdVector"]) and p["Id"] not in p["NbrMPIdVector"]
    assert np.all(owned == 1)                                                      # every dof has exactly one owner (:885-887)
