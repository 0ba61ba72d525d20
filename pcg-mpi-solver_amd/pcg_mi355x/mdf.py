"""Model Definition Files (MDF): the on-disk model format the reference pipeline starts from.

The reference unpacks `<model>.zip` into `<Scratch>/ModelData/MDF/` (src/data/read_input_model.py:19-34) and
reads from there

  * `GlobN.mat`, `dt.mat`                      run_metis.py:21-38   (`Data` row vectors)
  * `Ke.mat`, `Me.mat`                         partition_mesh.py:541-546  (`Data` = 1 x n_types cell of matrices)
  * `MatProp.mat`                              partition_mesh.py:503-514  (`Data` = 1 x n_mat struct: E, Pos, Rho)
  * per-element `.bin` arrays, Fortran order   partition_mesh.py:172-175
        NodeGlbOffset / DofGlbOffset / SignOffset  int64 (E, 2)  INCLUSIVE [first, last] ranges into the flat arrays
        Type int32, Level / Ck / Cm / Ce float64, PolyMat int32, sctrs float64 (E, 3), StrsGlb / StrsSign int8 (E, 6)
  * flat element arrays                        partition_mesh.py:223-225
        SignFlat int8, NodeGlbFlat int32, DofGlbFlat int32      (zero-based ids)
  * nodal arrays                               partition_mesh.py:324-330
        DiagM, F, Ud, Vd, NodeCoordVec float64 (GlobNDof) ; DofEff, FixedDof int32
  * `MeshPart_<N>.npy`                         run_metis.py:91-93 (element -> part id, written by the METIS step)
  * `MeshData_Glob.zpkl`                       run_metis.py:24-43 (zlib(pickle(dict)) of the GlobN entries + dt)

`write_mdf` lets the synthetic generators (brick.py, octree.py) feed the UNMODIFIED reference pipeline
(run_metis.config_GlobData -> partition_mesh -> pcg_solver), which is how tests/golden/part_*.npz were made
(oracle/make_partition_golden.py); `read_mdf` lets this package consume a real model when one is supplied.
Host-side set-up only.
"""
from __future__ import annotations

import os

import numpy as np

from .io import exportz, importz

__all__ = ["ELEM_ARRAYS", "FLAT_ARRAYS", "NODAL_ARRAYS", "model_from_brick", "model_from_octree", "write_mdf", "read_mdf",
           "config_glob_data", "write_mesh_part", "read_mesh_part", "main"]

# name -> (dtype on disk, trailing shape); partition_mesh.py:172-175
ELEM_ARRAYS = {"NodeGlbOffset": (np.int64, (2,)), "DofGlbOffset": (np.int64, (2,)), "SignOffset": (np.int64, (2,)),
               "Type": (np.int32, ()), "Level": (np.float64, ()), "Ck": (np.float64, ()), "Cm": (np.float64, ()),
               "Ce": (np.float64, ()), "PolyMat": (np.int32, ()), "sctrs": (np.float64, (3,)),
               "StrsGlb": (np.int8, (6,)), "StrsSign": (np.int8, (6,))}
FLAT_ARRAYS = {"SignFlat": np.int8, "NodeGlbFlat": np.int32, "DofGlbFlat": np.int32}          # partition_mesh.py:223-225
NODAL_ARRAYS = {"DiagM": np.float64, "F": np.float64, "DofEff": np.int32, "FixedDof": np.int32, "Ud": np.float64,
                "Vd": np.float64, "NodeCoordVec": np.float64}                                   # partition_mesh.py:324-330
GLOBN_KEYS = ["GlobNElem", "GlobNDof", "GlobNDofGlbFlat", "GlobNNodeGlbFlat", "GlobNDofEff", "GlobNFacesFlat",
              "GlobNFaces", "GlobNPolysFlat", "GlobNFixedDof"]                                  # run_metis.py:26-36


def _ragged(rows_nodes):
    """[(ne_k, nn_k) int arrays] (element order = concatenation) -> NodeGlbFlat, NodeGlbOffset, DofGlbFlat, DofGlbOffset."""
    node_flat = np.concatenate([r.ravel() for r in rows_nodes]).astype(np.int64)
    nn = np.concatenate([np.full(len(r), r.shape[1], np.int64) for r in rows_nodes])
    end = np.cumsum(nn)
    node_off = np.stack([end - nn, end - 1], 1)                                 # inclusive ranges
    dof_flat = (3 * node_flat[:, None] + np.arange(3)[None, :]).ravel()        # dof = 3*node + dir (partition_mesh.py:826)
    dof_off = np.stack([3 * (end - nn), 3 * end - 1], 1)
    return node_flat, node_off, dof_flat, dof_off


def _finish(model, coords, fixed_dofs, F, ke_list, dt=1.0):
    n_dof = 3 * len(coords)
    E = len(model["Type"])
    fixed = np.zeros(n_dof, bool)
    fixed[fixed_dofs] = True
    model.update({
        "Cm": np.ones(E), "Ce": np.ones(E), "PolyMat": np.zeros(E, np.int32),
        "StrsGlb": np.tile(np.arange(6, dtype=np.int8), (E, 1)), "StrsSign": np.zeros((E, 6), np.int8),
        "DiagM": np.ones(n_dof), "F": np.asarray(F, float), "Ud": np.zeros(n_dof), "Vd": np.zeros(n_dof),
        "NodeCoordVec": np.asarray(coords, float).ravel(),                      # x, y, z of node n at 3n .. 3n+2 (:357,:693-695)
        "DofEff": np.flatnonzero(~fixed).astype(np.int32), "FixedDof": np.flatnonzero(fixed).astype(np.int32),
        "Ke": [np.asarray(k, float) for k in ke_list], "Me": [np.eye(len(k)) for k in ke_list],
        "MatProp": [{"E": 1.0, "Pos": 0.2, "Rho": 1.0}], "dt": float(dt),
    })
    model.update({"GlobNElem": E, "GlobNDof": n_dof, "GlobNNode": n_dof // 3, "GlobNDofGlbFlat": len(model["DofGlbFlat"]),
                  "GlobNNodeGlbFlat": len(model["NodeGlbFlat"]), "GlobNDofEff": int((~fixed).sum()), "GlobNFacesFlat": 0,
                  "GlobNFaces": 0, "GlobNPolysFlat": 0, "GlobNFixedDof": int(fixed.sum())})
    return model


def model_from_brick(brick):
    """The SURVEY 8(d) brick as a global MDF model (element order i fastest; types / sign frames as brick.py)."""
    en = brick.elem_nodes()
    node_flat, node_off, dof_flat, dof_off = _ragged([en])
    sign = np.stack(brick.type_flip)[brick.elem_type]                           # (E, 24) per-element sign mask
    N = brick.N
    ids = np.arange(brick.n_node)
    coords = np.stack([ids % N, (ids // N) % N, ids // (N * N)], 1)
    model = {"NodeGlbFlat": node_flat, "NodeGlbOffset": node_off, "DofGlbFlat": dof_flat, "DofGlbOffset": dof_off,
             "SignFlat": sign.ravel().astype(np.int8), "SignOffset": dof_off.copy(),
             "Type": brick.elem_type.astype(np.int32), "Level": np.ones(brick.n_elem), "Ck": brick.Ck.copy(),
             "sctrs": coords[en].mean(axis=1)}
    return _finish(model, coords, brick.fixed_dofs(), brick.load_vector(), [brick.type_Ke(t) for t in range(brick.n_types)])


def model_from_octree(mesh, sign_seed=None):
    """octree.TwoLevelMesh (hex8 cells of two sizes + 13-node transition patterns) or octree.GradedOctreeMesh (multi-level, many
    pattern types) as a global MDF model."""
    flips = [np.zeros(3 * g.shape[1], bool) for g in mesh.group_nodes]
    if sign_seed is not None:
        r = np.random.default_rng(sign_seed)
        flips = [r.random(3 * g.shape[1]) < 0.4 for g in mesh.group_nodes]
    node_flat, node_off, dof_flat, dof_off = _ragged(mesh.group_nodes)
    sign = np.concatenate([np.tile(f, len(g)) for f, g in zip(flips, mesh.group_nodes)])
    level = np.concatenate(mesh.group_level)                                    # cell edge length per element
    ke = []
    for k, f in zip(mesh.group_ke, flips):
        d = np.where(f, -1.0, 1.0)
        ke.append(k * d[:, None] * d[None, :])
    model = {"NodeGlbFlat": node_flat, "NodeGlbOffset": node_off, "DofGlbFlat": dof_flat, "DofGlbOffset": dof_off,
             "SignFlat": sign.astype(np.int8), "SignOffset": dof_off.copy(),
             "Type": np.concatenate([np.full(len(g), t, np.int32) for t, g in enumerate(mesh.group_nodes)]),
             "Level": level, "Ck": np.concatenate(mesh.group_ck), "sctrs": np.concatenate(mesh.group_centroid)}
    fixed = (3 * mesh.fixed_nodes[:, None] + np.arange(3)).ravel()
    return _finish(model, mesh.coords, fixed, mesh.load_vector(), ke)


# ---------------------------------------------------------------------------------------------------------------
def write_mdf(mdf_path, model):
    """Write `model` as the files listed in the module docstring (2-D arrays in Fortran order, zero-based ids)."""
    import scipy.io
    os.makedirs(mdf_path, exist_ok=True)
    p = lambda name: os.path.join(mdf_path, name)                               # noqa: E731
    scipy.io.savemat(p("GlobN.mat"), {"Data": np.array([[float(model[k]) for k in GLOBN_KEYS]])})
    scipy.io.savemat(p("dt.mat"), {"Data": np.array([[model["dt"]]])})
    for name in ("Ke", "Me"):
        cell = np.empty((1, len(model[name])), dtype=object)
        for k, m in enumerate(model[name]):
            cell[0, k] = np.asarray(m, float)
        scipy.io.savemat(p(name + ".mat"), {"Data": cell})
    mp = np.zeros((1, len(model["MatProp"])), dtype=[("E", "O"), ("Pos", "O"), ("Rho", "O")])
    for k, m in enumerate(model["MatProp"]):
        for f in ("E", "Pos", "Rho"):
            mp[0, k][f] = np.array([[float(m[f])]])
    scipy.io.savemat(p("MatProp.mat"), {"Data": mp})
    for name, (dt, _) in ELEM_ARRAYS.items():
        np.asarray(model[name]).astype(dt).ravel(order="F").tofile(p(name + ".bin"))
    for name, dt in {**FLAT_ARRAYS, **NODAL_ARRAYS}.items():
        np.asarray(model[name]).astype(dt).tofile(p(name + ".bin"))
    return mdf_path


def config_glob_data(mdf_path):
    """run_metis.config_GlobData (run_metis.py:19-43): GlobN.mat + dt.mat -> dict, also stored as MeshData_Glob.zpkl."""
    import scipy.io
    g = scipy.io.loadmat(os.path.join(mdf_path, "GlobN.mat"))["Data"][0]
    out = {"GlobNElem": int(g[0]), "GlobNDof": int(g[1]), "GlobNNode": int(g[1] / 3), "GlobNDofGlbFlat": int(g[2]),
           "GlobNNodeGlbFlat": int(g[3]), "GlobNDofEff": int(g[4]), "GlobNFacesFlat": int(g[5]), "GlobNFaces": int(g[6]),
           "GlobNPolysFlat": int(g[7]), "GlobNFixedDof": int(g[8])}
    out["dt"] = float(scipy.io.loadmat(os.path.join(mdf_path, "dt.mat"))["Data"][0][0])
    exportz(os.path.join(mdf_path, "MeshData_Glob.zpkl"), out)
    return out


def read_mdf(mdf_path):
    """Inverse of write_mdf / reader of a real model directory: arrays in the dtypes the reference converts to
    (int / float / bool, partition_mesh.py:174,223-225,324-330)."""
    import scipy.io
    p = lambda name: os.path.join(mdf_path, name)                               # noqa: E731
    zp = p("MeshData_Glob.zpkl")
    model = dict(importz(zp)) if os.path.exists(zp) else config_glob_data(mdf_path)
    E = model["GlobNElem"]
    for name, (dt, tail) in ELEM_ARRAYS.items():
        a = np.fromfile(p(name + ".bin"), dtype=dt)
        if a.size == 0:                                                         # optional arrays may be empty (:196-197)
            model[name] = a
            continue
        a = a.reshape((E,) + tail, order="F")
        model[name] = a.astype(np.int64) if np.issubdtype(dt, np.integer) else a
    sizes = {"SignFlat": model["GlobNDofGlbFlat"], "NodeGlbFlat": model["GlobNNodeGlbFlat"], "DofGlbFlat": model["GlobNDofGlbFlat"],
             "DofEff": model["GlobNDofEff"], "FixedDof": model["GlobNFixedDof"]}
    for name, dt in {**FLAT_ARRAYS, **NODAL_ARRAYS}.items():
        a = np.fromfile(p(name + ".bin"), dtype=dt)
        n = sizes.get(name, model["GlobNDof"])
        if a.size != n:
            raise ValueError(f"{name}.bin holds {a.size} items, GlobN.mat says {n}")
        model[name] = a.astype(np.int64) if np.issubdtype(dt, np.integer) else a
    for name in ("Ke", "Me"):
        model[name] = [np.array(m, dtype=float) for m in scipy.io.loadmat(p(name + ".mat"))["Data"][0]]
    raw = scipy.io.loadmat(p("MatProp.mat"), struct_as_record=False)["Data"][0]
    model["MatProp"] = [{"E": m.E[0][0], "Pos": m.Pos[0][0], "Rho": m.Rho[0][0]} for m in raw]    # partition_mesh.py:507-514
    return model


def write_mesh_part(mdf_path, ele_part):
    """MeshPart_<N>.npy (+ .mat with one-based ids), run_metis.py:91-93."""
    import scipy.io
    ele_part = np.asarray(ele_part).astype(int)
    n = int(ele_part.max()) + 1 if len(ele_part) else 1
    base = os.path.join(mdf_path, f"MeshPart_{n}")
    np.save(base + ".npy", ele_part)
    scipy.io.savemat(base + ".mat", {"RefPart": ele_part + 1})
    return base + ".npy"


def read_mesh_part(mdf_path, n_parts):
    return np.load(os.path.join(mdf_path, f"MeshPart_{int(n_parts)}.npy"))     # partition_mesh.py:104-105


def main(argv=None):
    """Write a synthetic model as a model archive (the `concrete.zip` role of examples/run_basic_script.bash:20):

        python -m pcg_mi355x.mdf --brick 70 --out brick70.zip          # SURVEY 8(d) brick, 1 029 000 dof
        python -m pcg_mi355x.mdf --octree 96 96 40 8 --out oct.zip     # two-level octree mesh with hanging nodes
    """
    import argparse
    import shutil
    import tempfile
    ap = argparse.ArgumentParser(description=main.__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    g = ap.add_mutually_exclusive_group(required=True)
    g.add_argument("--brick", type=int, metavar="N", help="N x N x N nodes")
    g.add_argument("--octree", type=int, nargs=4, metavar=("NX", "NY", "NZ_FINE", "NZ_COARSE"))
    ap.add_argument("--types", type=int, default=1, help="pattern types of the brick (sign frames)")
    ap.add_argument("--out", required=True, help="archive to write (.zip) or directory")
    args = ap.parse_args(argv)
    if args.brick:
        from .brick import Brick
        model = model_from_brick(Brick(args.brick, n_types=args.types))
    else:
        from .octree import TwoLevelMesh
        model = model_from_octree(TwoLevelMesh(*args.octree))
    if args.out.endswith(".zip"):
        tmp = tempfile.mkdtemp(prefix="mdf_")
        try:
            write_mdf(tmp, model)
            shutil.make_archive(args.out[:-4], "zip", tmp)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    else:
        write_mdf(args.out, model)
    print(f">elements:  {model['GlobNElem']}\n>nodes:     {model['GlobNNode']}\n>dofs:      {model['GlobNDof']}\n>written:   {args.out}")


if __name__ == "__main__":
    main()
