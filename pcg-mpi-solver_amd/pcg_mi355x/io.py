"""Partition-file and result-file I/O in the reference's formats.

Lets the engine consume the per-part files written by the reference's partitioner and write
results its `src/data/export_vtk.py` reads back, without MPI-IO.  Host-side plumbing (NumPy /
pickle / zlib); nothing here is on the GPU hot path.

  reference                                                   here
  ----------------------------------------------------------  ------------------------------
  exportz / importz            file_operations.py:32-42       exportz / importz
  exportMP (per-part .mpidat)  partition_mesh.py:1303-1385    write_partition
  readModelData                pcg_solver.py:88-110           read_partition
  writeMPIFile_parallel        file_operations.py:348-375     write_result_vector
  readMPIBinFile               file_operations.py:516-531     read_result_vector
  initExportData / exportContourData ('U' only)
                               pcg_solver.py:142-209,841-896  ResultExporter
"""
from __future__ import annotations

import os
import pickle
import zlib

import numpy as np

__all__ = ["exportz", "importz", "write_partition", "read_partition", "write_result_vector", "read_result_vector",
           "ResultExporter"]

# keys the reference's partitioner exports per part (partition_mesh.py:1310-1317)
REF_KEYS = ['Id', 'SubDomainData', 'NDOF', 'NNode', 'DofVector', 'NodeIdVector', 'InvDiagM', 'NodeWeightVector',
            'RefLoadVector', 'NbrMPIdVector', 'ElemIdVector', 'OvrlpLocalNodeIdVecList', 'OvrlpLocalDofVecList',
            'RefPlotData', 'MPList_RefPlotDofIndicesList', 'IntfcLocalNodeIdList', 'MPList_IntfcNodeIdVector',
            'MPList_IntfcNNode', 'DofWeightVector', 'LocFixedDof', 'Flat_ElemLocDof', 'NCountDof', 'N_NbrDof', 'Ud', 'Vd',
            'DofEff', 'LocDofEff', 'NElem', 'MatProp', 'NodeCoordVec']


def exportz(file_name, data):
    """zlib(pickle(data)) (file_operations.py:32-37)."""
    with open(file_name, "wb") as f:
        f.write(zlib.compress(pickle.dumps(data, pickle.HIGHEST_PROTOCOL)))


def importz(file_name):
    """file_operations.py:39-42."""
    with open(file_name, "rb") as f:
        return pickle.loads(zlib.decompress(f.read()))


def write_partition(prefix, parts):
    """Write `<prefix><N>_<id>.mpidat` + `<prefix><N>_metadat.npy` like exportMP (partition_mesh.py:1303-1385).
    Private engine entries (keys starting with '_') are not exported.  `parts` must be the COMPLETE partition (ids
    0..N-1, any order): the metadata arrays are indexed by part id, as readModelData reads them (pcg_solver.py:100-106)."""
    n = len(parts)
    ids = sorted(int(p["Id"]) for p in parts)
    if ids != list(range(n)):
        raise ValueError(f"write_partition needs every part of the partition exactly once (ids 0..{n - 1}), got {ids}; "
                         "a rank that built only its own part (partition_model(only=[rank])) cannot write the metadata")
    parts = sorted(parts, key=lambda p: int(p["Id"]))
    base = f"{prefix}{n}"
    os.makedirs(os.path.dirname(os.path.abspath(base)), exist_ok=True)
    meta = []
    for p in parts:
        ref = {"GlobData": {k: v for k, v in p["GlobData"].items()}}
        for key in REF_KEYS:
            if key in p:
                ref[key] = p[key]
        buf = np.frombuffer(zlib.compress(pickle.dumps(ref, pickle.HIGHEST_PROTOCOL)), "b")      # :1332
        with open(f"{base}_{p['Id']}.mpidat", "wb") as f:                                        # :1366-1369
            f.write(buf.tobytes())
        meta.append([buf.nbytes, len(buf), buf.dtype])
    meta = np.array(meta, dtype=object)
    offsets = np.cumsum(np.hstack([[0], meta[:-1, 0]]))
    np.save(base + "_metadat", np.array({"NfData": meta[:, 1], "DTypeData": meta[:, 2], "OffsetData": offsets},
                                        dtype=object))                                           # :1350-1353
    return base


def read_partition(prefix, n_parts, part_id, glob_data=None):
    """readModelData (pcg_solver.py:88-110): load part `part_id` of an `n_parts` partition; the part's
    GlobData is merged into `glob_data` (the solver's own dict) exactly as the reference does (:107-108)."""
    base = f"{prefix}{n_parts}"
    metadat = np.load(base + "_metadat.npy", allow_pickle=True).item()                           # :100
    nf, dtype = metadat["NfData"][part_id], metadat["DTypeData"][part_id]
    buf = np.fromfile(f"{base}_{part_id}.mpidat", dtype=dtype, count=int(nf))                    # readMPIFile :498-513
    part = pickle.loads(zlib.decompress(buf.tobytes()))                                          # :106
    gd = {} if glob_data is None else glob_data
    gd.update(part["GlobData"])                                                                  # :107
    part["GlobData"] = gd                                                                        # :108
    return part


def _gather_sizes(nbytes, n_items, dtype, comm):
    if comm is None or comm.world == 1:
        return [[nbytes, n_items, dtype]], 0
    import torch.distributed as dist
    out = [None] * comm.world
    dist.all_gather_object(out, [nbytes, n_items, str(dtype)], group=comm.group)
    return [[a, b, np.dtype(c)] for a, b, c in out], comm.rank


def write_result_vector(file_name, local_values, comm=None):
    """writeMPIFile_parallel (file_operations.py:348-375): every rank's array at its byte offset of ONE
    `<file_name>.mpidat`, plus `<file_name>_metadat.npy` {NfData, DTypeData, OffsetData} from rank 0."""
    data = np.ascontiguousarray(local_values)
    meta, rank = _gather_sizes(data.nbytes, len(data), data.dtype, comm)
    meta = np.array(meta, dtype=object)
    offsets = np.cumsum(np.hstack([[0], meta[:-1, 0]])).astype(np.int64)
    path = file_name + ".mpidat"
    if rank == 0:
        np.save(file_name + "_metadat", np.array({"NfData": meta[:, 1], "DTypeData": meta[:, 2], "OffsetData": offsets},
                                                 dtype=object))
        with open(path, "wb") as f:                       # create + size the file once
            f.truncate(int(meta[:, 0].sum()))
    if comm is not None and comm.world > 1:
        import torch.distributed as dist
        dist.barrier(group=comm.group)
    with open(path, "r+b") as f:
        f.seek(int(offsets[rank]))
        f.write(data.tobytes())
    if comm is not None and comm.world > 1:
        import torch.distributed as dist
        dist.barrier(group=comm.group)


def write_result_vector_parts(file_name, values_by_part):
    """The same two files as write_result_vector() written by N ranks, from ONE process that holds every part (device
    group, pcg_mi355x.group): segment k at part k's byte offset, one metadata row per part."""
    data = [np.ascontiguousarray(v) for v in values_by_part]
    meta = np.array([[d.nbytes, len(d), d.dtype] for d in data], dtype=object)
    offsets = np.cumsum(np.hstack([[0], meta[:-1, 0]])).astype(np.int64)
    np.save(file_name + "_metadat", np.array({"NfData": meta[:, 1], "DTypeData": meta[:, 2], "OffsetData": offsets}, dtype=object))
    with open(file_name + ".mpidat", "wb") as f:
        for d in data:
            f.write(d.tobytes())


def read_result_vector(file_name):
    """readMPIBinFile (file_operations.py:516-531): the concatenation of all ranks' segments."""
    metadat = np.load(file_name + "_metadat.npy", allow_pickle=True).item()
    return np.fromfile(file_name + ".mpidat", dtype=metadat["DTypeData"][0])


class ResultExporter:
    """initExportData + exportContourData for ExportVars 'U' (pcg_solver.py:142-209, :841-896): writes
    `Dof.mpidat`, `NodeId.mpidat`, `U_<k>.mpidat` and `Time_T.npy` into `res_vec_path`, owned dofs only
    (DofWeightVector as bool mask, :196-200), in the layout src/data/export_vtk.py:73-80 reads."""

    def __init__(self, part, res_vec_path, comm=None):
        self.part, self.path, self.comm = part, res_vec_path, comm
        self.rank = 0 if comm is None else comm.rank
        if self.rank == 0:
            os.makedirs(res_vec_path, exist_ok=True)
        if comm is not None and comm.world > 1:
            import torch.distributed as dist
            dist.barrier(group=comm.group)
        self.dof_mask = np.asarray(part["DofWeightVector"]).astype(bool)                           # :196
        self.node_mask = np.asarray(part["NodeWeightVector"]).astype(bool)                         # :197
        write_result_vector(os.path.join(res_vec_path, "Dof"), part["DofVector"][self.dof_mask], comm)       # :199
        write_result_vector(os.path.join(res_vec_path, "NodeId"), part["NodeIdVector"][self.node_mask], comm)  # :200
        self.count = 0
        self.times = []

    def export(self, time_value):
        write_result_vector(os.path.join(self.path, f"U_{self.count}"), self.part["Un"][self.dof_mask], self.comm)  # :866-868
        if self.rank == 0:                                                                          # :889-892
            self.times.append(time_value)
            np.save(os.path.join(self.path, "Time_T"), self.times)
        self.count += 1                                                                             # :894


class GroupResultExporter:
    """ResultExporter for a device group: one process writes what the N ranks of the reference write (same files, same
    per-part segments and metadata rows)."""

    def __init__(self, parts, res_vec_path):
        self.parts, self.path = parts, res_vec_path
        os.makedirs(res_vec_path, exist_ok=True)
        self.dof_masks = [np.asarray(P["DofWeightVector"]).astype(bool) for P in parts]             # :196
        node_masks = [np.asarray(P["NodeWeightVector"]).astype(bool) for P in parts]                # :197
        write_result_vector_parts(os.path.join(res_vec_path, "Dof"), [P["DofVector"][m] for P, m in zip(parts, self.dof_masks)])
        write_result_vector_parts(os.path.join(res_vec_path, "NodeId"), [P["NodeIdVector"][m] for P, m in zip(parts, node_masks)])
        self.count = 0
        self.times = []

    def export(self, time_value):
        write_result_vector_parts(os.path.join(self.path, f"U_{self.count}"), [P["Un"][m] for P, m in zip(self.parts, self.dof_masks)])
        self.times.append(time_value)
        np.save(os.path.join(self.path, "Time_T"), self.times)
        self.count += 1
