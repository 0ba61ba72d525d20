"""Pipeline stages 1-3 of the reference on the MI355X package: model archive -> MDF directory
-> element partition -> one `<N>_<id>.mpidat` per part, ready for `python -m pcg_mi355x.run`.

    reference stage (examples/run_basic_script.bash:20-28)                  here
    1  python src/data/read_input_model.py <Work> <Model> <Scratch> <zip>   --model <zip | MDF directory>
    2  python src/solver/run_metis.py <N>                                   MeshPart_<N>.npy if present, else recursive
                                                                            coordinate bisection (mgmetis is not installed)
    3  mpiexec -np <W> python src/solver/partition_mesh.py <N> 0            pcg_mi355x.partition.partition_model

    python -m pcg_mi355x.prepare --model concrete.zip --scratch <ScratchPath> --n-parts 8

Writes <Scratch>/ModelData/MDF/ (unpacked model, MeshData_Glob.zpkl, MeshPart_<N>.npy) and
<Scratch>/ModelData/MPI/<N>_<id>.mpidat + <N>_metadat.npy, the paths the reference uses (read_input_model.py:24-25).
Host-side set-up only.
"""
from __future__ import annotations

import argparse
import os
import shutil
import time

from . import mdf
from .io import write_partition
from .partition import geometric_partition, partition_model

__all__ = ["prepare", "main"]


def prepare(model_path, scratch_path, n_parts, ele_part=None, log=print):
    """Returns the partition-file prefix (`PyDataPath_Part`)."""
    t0 = time.time()
    mdf_path = os.path.join(scratch_path, "ModelData", "MDF", "")
    prefix = os.path.join(scratch_path, "ModelData", "MPI", "")
    os.makedirs(mdf_path, exist_ok=True)
    os.makedirs(prefix, exist_ok=True)
    if os.path.isdir(model_path):
        if os.path.abspath(model_path) != os.path.abspath(mdf_path):
            shutil.copytree(model_path, mdf_path, dirs_exist_ok=True)
    else:
        shutil.unpack_archive(model_path, mdf_path)                          # read_input_model.py:34
    gd = mdf.config_glob_data(mdf_path)                                       # run_metis.py:19-43
    log(f">elements:  {gd['GlobNElem']}\n>nodes:     {gd['GlobNNode']}\n>dofs:      {gd['GlobNDof']}")
    model = mdf.read_mdf(mdf_path)
    part_file = os.path.join(mdf_path, f"MeshPart_{n_parts}.npy")
    if ele_part is not None:
        mdf.write_mesh_part(mdf_path, ele_part)
    elif not os.path.exists(part_file):
        log(f">generating indices for {n_parts} mesh parts (recursive coordinate bisection)..")
        mdf.write_mesh_part(mdf_path, geometric_partition(model, n_parts))
    ele_part = mdf.read_mesh_part(mdf_path, n_parts)
    log(f">partitioning mesh into {n_parts} parts..")
    paths = {"ScratchPath": scratch_path, "MDF_Path": mdf_path, "PyDataPath_Part": prefix}
    parts = partition_model(model, ele_part, glob_data=paths)
    write_partition(prefix, parts)
    log(f">success!\n>total runtime: {time.time() - t0:.2f} sec")
    return prefix


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--model", required=True, help="model archive (.zip) or an unpacked MDF directory")
    ap.add_argument("--scratch", required=True, help="ScratchPath: ModelData/MDF and ModelData/MPI are created below it")
    ap.add_argument("--n-parts", type=int, required=True)
    args = ap.parse_args(argv)
    prepare(args.model, args.scratch, args.n_parts)


if __name__ == "__main__":
    main()
