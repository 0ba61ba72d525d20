"""MDF model files + partition set-up against the UNMODIFIED reference pipeline.

tests/golden/part_*.npz hold what the reference's partition_mesh.py exported for the synthetic models of
oracle/partition_cases.py (every RefMeshPart key, flattened) and what the reference's PCG then solved;
tests/golden/refpart_* is one partition exactly as the reference wrote it.  Integer / index data: exact.
Gathered floats (coordinates, loads, Ke): exact.  Solutions: the solver gates of tests/util.py."""
import os

import numpy as np
import pytest

import partition_cases as pc
from pcg_mi355x import mdf, partition
from pcg_mi355x.io import read_partition, write_partition
from util import GOLDEN, golden, relerr, check_solution_against_golden, free_port, dump_failed_run


def assert_same_part(flat_ref, part, prefix):
    mine = pc.flatten_part(part, prefix)
    ref = {k: v for k, v in flat_ref.items() if k.startswith(prefix + "/")}
    assert mine.keys() == ref.keys(), sorted(set(mine) ^ set(ref))[:10]
    for k, v in ref.items():
        assert mine[k].dtype == v.dtype and mine[k].shape == v.shape and np.array_equal(mine[k], v), k


@pytest.mark.parametrize("name", list(pc.CASES))
def test_partition_model_reproduces_reference_export(name, tmp_path):
    g = golden(name)
    model, ele_part = pc.build_model(name)
    assert np.array_equal(ele_part, g["ele_part"])
    flat_ref = {k: g[k] for k in g.files}
    n_parts = int(g["n_parts"])
    # through the files (write_mdf -> read_mdf), as the pipeline runs
    mdf_path = mdf.write_mdf(str(tmp_path / "MDF"), model)
    mdf.write_mesh_part(mdf_path, ele_part)
    parts = partition.partition_model(mdf.read_mdf(mdf_path), mdf.read_mesh_part(mdf_path, n_parts))
    assert len(parts) == n_parts
    for k, p in enumerate(parts):
        assert_same_part(flat_ref, p, f"p{k}")
    # a rank building only its own part gets the identical dict
    k = n_parts - 1
    assert_same_part(flat_ref, partition.partition_model(model, ele_part, only=[k])[0], f"p{k}")


def test_mdf_round_trip_and_glob_data(tmp_path):
    model, _ = pc.build_model("part_octree_p3")
    path = mdf.write_mdf(str(tmp_path / "MDF"), model)
    gd = mdf.config_glob_data(path)                                   # run_metis.py:19-43
    assert gd["GlobNElem"] == model["GlobNElem"] and gd["GlobNNode"] == model["GlobNDof"] // 3 and gd["dt"] == model["dt"]
    assert os.path.exists(os.path.join(path, "MeshData_Glob.zpkl"))
    back = mdf.read_mdf(path)
    for name in list(mdf.ELEM_ARRAYS) + list(mdf.FLAT_ARRAYS) + list(mdf.NODAL_ARRAYS):
        assert np.array_equal(np.asarray(back[name]), np.asarray(model[name])), name
        assert back[name].dtype in (np.int64, np.float64), name        # the reference's int / float conversion
    for a, b in zip(back["Ke"], model["Ke"]):
        assert np.array_equal(a, b)
    assert back["MatProp"] == model["MatProp"]
    # Fortran-order (E, 2) offsets on disk (partition_mesh.py:335)
    raw = np.fromfile(os.path.join(path, "NodeGlbOffset.bin"), np.int64)
    assert np.array_equal(raw[:model["GlobNElem"]], model["NodeGlbOffset"][:, 0])
    with pytest.raises(ValueError):
        np.zeros(3).tofile(os.path.join(path, "F.bin"))
        mdf.read_mdf(path)


def test_reader_parses_reference_written_partition(tmp_path):
    """refpart_3_<id>.mpidat were written by the reference's exportMP (partition_mesh.py:1300-1369)."""
    g = golden("part_brick_p3")
    flat_ref = {k: g[k] for k in g.files}
    for k in range(3):
        p = read_partition(os.path.join(GOLDEN, "refpart_"), 3, k)
        assert_same_part(flat_ref, p, f"p{k}")
    # and our writer produces files our reader (= the reference's reader logic, pcg_solver.py:100-106) returns unchanged
    model, ele_part = pc.build_model("part_brick_p3")
    parts = partition.partition_model(model, ele_part)
    write_partition(str(tmp_path / "part"), parts)
    for k in range(3):
        assert_same_part(flat_ref, read_partition(str(tmp_path / "part"), 3, k), f"p{k}")


def test_partitioner_agrees_with_direct_generators():
    """brick.make_parts / octree.make_octree_parts (the bench generators) and MDF -> partition_model describe the
    same parts on every key the solver reads."""
    from pcg_mi355x.brick import Brick, make_parts, block_partition
    b = Brick(6, n_types=2)
    ep = block_partition(b, 1, 1, 3)
    direct = make_parts(b, ep)
    via = partition.partition_model(mdf.model_from_brick(b), ep)
    for d, v in zip(direct, via):
        for key in ("NDOF", "DofVector", "NodeIdVector", "RefLoadVector", "Ud", "LocDofEff", "LocFixedDof", "DofWeightVector",
                    "NbrMPIdVector", "Flat_ElemLocDof", "NodeCoordVec"):
            assert np.array_equal(np.asarray(d[key]), np.asarray(v[key])), key
        for a, c in zip(d["OvrlpLocalDofVecList"], v["OvrlpLocalDofVecList"]):
            assert np.array_equal(a, c)
        for gd, gv in zip(d["SubDomainData"]["StrucDataList"], v["SubDomainData"]["StrucDataList"]):
            for key in ("ElemList_LocDofVector", "ElemList_SignVector", "ElemList_Ck", "ElemStiffMat"):
                assert np.array_equal(gd[key], gv[key]), key


def test_geometric_partition_and_errors():
    model, _ = pc.build_model("part_octree_p5")
    for n in (1, 2, 3, 7):
        ep = partition.geometric_partition(model, n)
        cnt = np.bincount(ep, minlength=n)
        assert cnt.min() >= len(ep) // n - 1 and cnt.max() <= -(-len(ep) // n) + 1      # balanced
        parts = partition.partition_model(model, ep)
        # every dof owned exactly once: weights sum to the global dof count (partition_mesh.py:868-887)
        assert sum(int(p["DofWeightVector"].sum()) for p in parts) == model["GlobNDof"]
        for p in parts:                                                                     # symmetric neighbour lists
            for q, ov in zip(p["NbrMPIdVector"], p["OvrlpLocalNodeIdVecList"]):
                back = parts[q]["NbrMPIdVector"].index(p["Id"])
                assert np.array_equal(p["NodeIdVector"][ov], parts[q]["NodeIdVector"][parts[q]["OvrlpLocalNodeIdVecList"][back]])
    with pytest.raises(ValueError):
        partition.partition_model(model, np.zeros(3, int))
    bad = dict(model); bad["Ke"] = model["Ke"][:1]
    with pytest.raises(ValueError):
        partition.partition_model(bad, np.zeros(model["GlobNElem"], int))


@pytest.mark.parametrize("kind", ["sell", "ebe"])
@pytest.mark.parametrize("name", ["part_brick_p1", "part_octree_p1", "part_octree_p3", "part_brick_p4", "part_brick_rand5"])
def test_mdf_to_solution_matches_reference_pipeline(hostops, name, kind, tmp_path):
    """MDF files -> partition_model -> engine (CPU test double of the backend) vs the reference pipeline's solution."""
    import pcg_mi355x as pm
    from util import run_dist
    g = golden(name)
    model, ele_part = pc.build_model(name)
    n_parts = int(g["n_parts"])
    tol_iter = 1 if kind == "ebe" else 0          # borderline exits are recognised by check_solution_against_golden
    if n_parts == 1:
        parts = pc.prepare_for_solve(partition.partition_model(model, ele_part))
        pm.configure(operator=kind)
        try:
            pm.update_bc(parts[0]); pm.update_preconditioner(parts[0]); pm.solve(parts[0])
        finally:
            pm.configure()
        gd = parts[0]["GlobData"]
        un = np.zeros(model["GlobNDof"]); un[parts[0]["DofVector"]] = parts[0]["Un"]
        check_solution_against_golden(g, int(gd["TimeList_Flag"][1]), int(gd["TimeList_Iter"][1]), float(gd["TimeList_RelRes"][1]),
                                      un, None, tol_iter=tol_iter)
    else:
        res = run_dist("partition:" + name, n_parts, "gloo", "hostops", tmp_path, extra=(kind,))
        un = np.zeros(model["GlobNDof"])
        for r in reversed(res):
            un[r["DofVector"]] = r["Un"]
        check_solution_against_golden(g, int(res[0]["flag"]), int(res[0]["iter"]), float(res[0]["relres"]), un, None,
                                      tol_iter=tol_iter)


def test_partition_files_feed_a_multi_rank_solve(hostops, tmp_path):
    """Stage 3 -> stage 4 through the files: prepare() writes 3 partition files (the fixture's ElePart), three ranks
    read ONE file each like the reference's readModelData and solve; result = the reference pipeline's solution."""
    from pcg_mi355x import prepare
    from util import run_dist
    g = golden("part_octree_p3")
    model, ele_part = pc.build_model("part_octree_p3")
    mdf.write_mdf(str(tmp_path / "model"), model)
    prefix = prepare.prepare(str(tmp_path / "model"), str(tmp_path / "scratch"), 3, ele_part=ele_part, log=lambda m: None)
    assert sorted(f for f in os.listdir(prefix) if f.endswith(".mpidat")) == ["3_0.mpidat", "3_1.mpidat", "3_2.mpidat"]
    res = run_dist("files:" + prefix, 3, "gloo", "hostops", tmp_path)
    un = np.zeros(model["GlobNDof"])
    for r in reversed(res):
        un[r["DofVector"]] = r["Un"]
    check_solution_against_golden(g, int(res[0]["flag"]), int(res[0]["iter"]), float(res[0]["relres"]), un, None, tol_iter=1)


SETTINGS = {"TimeHistoryParam": {"ExportFlag": True, "ExportFrmRate": 1, "ExportFrms": [], "PlotFlag": False,
                                 "TimeStepDelta": [0, 1], "ExportVars": "U"},
            "SolverParam": {"Tol": 1e-7, "MaxIter": 10000}}                      # examples/run_basic_script.bash:34-44


def test_model_archive_cli(tmp_path):
    out = str(tmp_path / "oct.zip")
    mdf.main(["--octree", "4", "4", "2", "2", "--out", out])
    import shutil
    shutil.unpack_archive(out, str(tmp_path / "x"))
    m = mdf.read_mdf(str(tmp_path / "x"))
    ref, _ = pc.build_model("part_octree_p3")                       # same mesh, no sign frames
    assert m["GlobNElem"] == ref["GlobNElem"] and np.array_equal(m["NodeGlbFlat"], ref["NodeGlbFlat"])
    mdf.main(["--brick", "5", "--types", "2", "--out", str(tmp_path / "b")])
    assert mdf.read_mdf(str(tmp_path / "b"))["GlobNDof"] == 375


def test_pipeline_from_model_archive(hostops, tmp_path):
    """model.zip -> prepare (stages 1-3) -> partition files -> load-step driver (stage 4) -> result vectors."""
    import shutil
    import pcg_mi355x as pm
    from pcg_mi355x import prepare, run as prun, io as pio
    g = golden("part_brick_p1")
    model, _ = pc.build_model("part_brick_p1")
    mdf.write_mdf(str(tmp_path / "model"), model)
    archive = shutil.make_archive(str(tmp_path / "concrete"), "zip", str(tmp_path / "model"))
    msgs = []
    prefix = prepare.prepare(archive, str(tmp_path / "scratch"), 1, log=msgs.append)
    assert any("elements" in m for m in msgs)
    assert os.path.exists(str(tmp_path / "scratch" / "ModelData" / "MDF" / "MeshPart_1.npy"))      # run_metis.py:91
    gd = prun.init_glob_data()
    part = pio.read_partition(prefix, 1, 0, gd)
    prun.apply_settings(gd, SETTINGS)
    pm.configure(comm=None)
    res = str(tmp_path / "Results_Run1" / "ResVecData") + os.sep
    flag, relres, it = prun.run_load_steps(part, res)
    assert flag[1] == int(g["flag"]) and it[1] == int(g["iter"])
    assert relerr(pio.read_result_vector(res + "U_1"), g["Un"]) < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sell", "ebe"])
@pytest.mark.parametrize("name", ["part_brick_p1", "part_octree_p1"])
def test_mdf_to_solution_on_gpu(gpu_lib, name, kind, tmp_path):
    """MDF directory -> `python -m pcg_mi355x.run --mdf` (in-memory partition, HIP engine) vs the reference pipeline."""
    import subprocess
    import sys
    from pcg_mi355x import io as pio
    from util import ROOT
    g = golden(name)
    model, _ = pc.build_model(name)
    path = mdf.write_mdf(str(tmp_path / "MDF"), model)
    pio.exportz(str(tmp_path / "GlobSettings.zpkl"), SETTINGS)
    results = str(tmp_path / "Results_Run1")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "pcg-mpi-solver_amd"))
    r = subprocess.run([sys.executable, "-m", "pcg_mi355x.run", "--mdf", path, "--settings", str(tmp_path / "GlobSettings.zpkl"),
                        "--results", results, "--operator", kind], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    td = np.load(os.path.join(results, "PlotData", "TimeData.npz"))
    un = pio.read_result_vector(os.path.join(results, "ResVecData", "U_1"))
    check_solution_against_golden(g, int(td["Flag"][1]), int(td["Iter"][1]), float(td["RelRes"][1]), un, None,
                                  tol_iter=1 if kind == "ebe" else 0)


def _oracle_load_steps(model, ele_part, deltas):
    """The reference's load-step loop (pcg_solver.py:1002-1008) on the oracle: per step the global solution."""
    import pcg_oracle
    parts = pc.prepare_for_solve(partition.partition_model(model, ele_part))
    n = len(deltas)
    for p in parts:
        p["GlobData"]["TimeStepDelta"] = list(deltas)
        p["GlobData"]["TimeList_Flag"], p["GlobData"]["TimeList_RelRes"], p["GlobData"]["TimeList_Iter"] = np.zeros(n), np.zeros(n), np.zeros(n)
    sols, its = [], []
    for step in range(1, n):
        for p in parts:
            p["GlobData"]["TimeStepCount"] = step
        out = pcg_oracle.solve_step(parts)
        assert out["flag"] == 0
        un = np.zeros(model["GlobNDof"])
        for p in reversed(parts):
            un[p["DofVector"]] = p["Un"]
        sols.append(un); its.append(out["iter"])
    return sols, its


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sell", "ebe"])
@pytest.mark.parametrize("name,ranks", [("part_brick_p1", 1), ("part_octree_p3", 3), ("part_octree_p3", -3)])
def test_load_step_driver_on_gpu(gpu_lib, name, ranks, kind, tmp_path):
    run_load_step_driver(gpu_lib, name, ranks, kind, tmp_path)


def run_load_step_driver(gpu_lib, name, ranks, kind, tmp_path, extra=()):
    """SURVEY 8(f)-4 on the HIP engine: MDF -> `python -m pcg_mi355x.run` with TWO load steps (warm start from the
    previous Un, :358,:378) -> U_<k>.mpidat / TimeData in the layout export_vtk.py reads, vs the oracle's loop.
    3 ranks: one process per part with the engine's native communicator; on the 1-GPU box they share the device and talk
    through the RCCL stand-in (tests/fakenccl), with >= 3 GPUs through librccl.  The calc / comm-wait split (a7) comes
    from HIP events around the exchange wait and the all-reduces."""
    import subprocess
    import sys
    import conftest
    from pcg_mi355x import io as pio
    from util import ROOT
    deltas = [0, 0.5, 1.0]
    model, ele_part = pc.build_model(name)
    path = mdf.write_mdf(str(tmp_path / "MDF"), model)
    if ranks > 1:
        mdf.write_mesh_part(path, ele_part)
    settings = {"TimeHistoryParam": dict(SETTINGS["TimeHistoryParam"], TimeStepDelta=deltas), "SolverParam": SETTINGS["SolverParam"]}
    pio.exportz(str(tmp_path / "GlobSettings.zpkl"), settings)
    results = str(tmp_path / "Results_Run1")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "pcg-mpi-solver_amd"))
    cmd = [sys.executable]
    group = ranks < 0                          # -3: ONE process drives the three parts (--group, C ABI pcg_group_*)
    ranks = abs(ranks)
    if group:
        if gpu_lib.lib().pcg_device_count() < ranks:
            env.update(PCG_RCCL_LIB=conftest.build_fakenccl())
    elif ranks > 1:
        if gpu_lib.lib().pcg_device_count() < ranks:
            env.update(PCG_RUN_SHARE_GPU="1", PCG_RCCL_LIB=conftest.build_fakenccl())
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
                "--master-port", str(free_port())]
    cmd += ["-m", "pcg_mi355x.run", "--mdf", path, "--settings", str(tmp_path / "GlobSettings.zpkl"), "--results", results,
            "--operator", kind] + (["--group", "--n-parts", str(ranks)] if group else []) + list(extra)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    dump_failed_run(f"load_step_{name}_{ranks}_{kind}" + "_".join(extra).replace("-", ""), r)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    if extra:
        assert ">engine-side all-reduce: on" in r.stdout and ">engine-side exchange: on" in r.stdout, r.stdout[-2000:]
    sols, its = _oracle_load_steps(model, ele_part, deltas)
    td = np.load(os.path.join(results, "PlotData", "TimeData.npz"))
    dof = pio.read_result_vector(os.path.join(results, "ResVecData", "Dof"))
    assert len(np.unique(dof)) == len(dof) == model["GlobNDof"]
    for k in (1, 2):
        u = pio.read_result_vector(os.path.join(results, "ResVecData", f"U_{k}"))
        assert int(td["Flag"][k]) == 0 and abs(int(td["Iter"][k]) - its[k - 1]) <= 1
        assert relerr(u, sols[k - 1][dof]) < 2e-7
    assert list(np.load(os.path.join(results, "ResVecData", "Time_T.npy"))) == [0.0, 1.0, 2.0]
    assert float(td["CalcTime"]) > 0
    if ranks > 1:
        assert np.load(os.path.join(results, "ResVecData", "U_1_metadat.npy"), allow_pickle=True).item()["NfData"].shape == (ranks,)
        assert 0 < float(td["CommWaitTime"]) < float(td["TotalTime"])         # a7: time blocked in communication (GPU side)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(pc.CASES))
def test_partition_on_device_reproduces_reference_export(gpu_lib, name):
    """SURVEY 8(f)-3 on the GPU: interface discovery (scatter / mark / emit) and the local numbering (mark / scan / gather)
    as HIP kernels (csrc/part_setup.hip) - every exported key of every part identical (values, shapes, dtypes) to what the
    unmodified partition_mesh.py wrote for the same model and the same element -> part vector."""
    g = golden(name)
    model, ele_part = pc.build_model(name)
    flat_ref = {k: g[k] for k in g.files}
    n_parts = int(g["n_parts"])
    parts = partition.partition_model(model, ele_part, device=0)
    assert len(parts) == n_parts
    for k, p in enumerate(parts):
        assert_same_part(flat_ref, p, f"p{k}")
    k = n_parts - 1                                   # a rank that builds only its own part
    assert_same_part(flat_ref, partition.partition_model(model, ele_part, only=[k], device=0)[0], f"p{k}")


@pytest.mark.gpu
def test_partition_device_kernels_at_size(gpu_lib):
    """The device passes on a 1 M-dof brick split 2x2x2 (tile boundaries of the scan, multi-block launches, a few thousand
    interface pairs per part) against the whole-array host path."""
    from pcg_mi355x.brick import Brick, block_partition
    b = Brick(70, seed=0)
    model = mdf.model_from_brick(b)
    ele_part = block_partition(b, 2, 2, 2).astype(np.int64)
    for k in (0, 5):
        host = partition.partition_model(model, ele_part, only=[k])[0]
        dev = partition.partition_model(model, ele_part, only=[k], device=0)[0]
        fh, fd = pc.flatten_part(host, "p"), pc.flatten_part(dev, "p")
        assert fh.keys() == fd.keys()
        for key, v in fh.items():
            assert fd[key].dtype == v.dtype and np.array_equal(fd[key], v), key
