"""pcg_mi355x - MI355X-native PCG iteration engine behind the reference solver's hot path.

Drop-in for `PCG(RefMeshPart)` and its helpers in ankitskr/PCG-MPI-solver
(src/solver/pcg_solver.py); see solver.py for the function map, include/pcg_mi355x.h for the C
ABI and DESIGN.md for the kernels.  Importing this package does not touch the GPU; the first
operator construction loads lib/libpcg_mi355x.so and fails loudly when it (or a gfx950 device) is
missing - there is no CPU path.
"""
from . import _lib
from ._lib import PcgError
from .operator import Operator, from_refmeshpart, assemble_bsr3
from . import io  # noqa: F401  (partition / result files; the CLI stages run / prepare / mdf are imported on demand)
from .group import DeviceGroup, GroupSolver
from .solver import (configure, get_operator, solve, PCG, update_bc, updateBC, update_preconditioner,
                     updatePreconditioner, calc_matvec_prod, calcMatVecProd, calc_mpfint, calcMPFint,
                     solve_system, SolveInfo)

__all__ = ["PcgError", "Operator", "from_refmeshpart", "assemble_bsr3", "DeviceGroup", "GroupSolver", "configure", "get_operator", "solve", "PCG",
           "update_bc", "updateBC", "update_preconditioner", "updatePreconditioner", "calc_matvec_prod",
           "calcMatVecProd", "calc_mpfint", "calcMPFint", "solve_system", "SolveInfo"]
