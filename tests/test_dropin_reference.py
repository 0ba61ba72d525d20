"""INTEGRATION.md section 1, executed: the five functions are rebound INSIDE the unmodified reference module and the
reference's own load-step sequence (pcg_solver.py:996-1008) runs on them.  Needs the reference checkout (build
container only; skipped on the GPU box, where /root/reference does not exist)."""
import os

import numpy as np
import pytest

import golden_cases
import pcg_mi355x as pm
from util import golden, relerr

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout not present")


@pytest.mark.parametrize("case", ["n9_p1", "oct_p1"])
def test_rebinding_the_five_functions_inside_the_reference_module(hostops, monkeypatch, case):
    import ref_shim
    ref = ref_shim.load_reference()
    ref_shim.WORLD.configure(1)
    for name in ("calcMatVecProd", "calcMPFint", "updateBC", "updatePreconditioner", "PCG"):
        assert callable(getattr(ref, name))
        monkeypatch.setattr(ref, name, getattr(pm, name))
    pm.configure(comm=None)
    _, parts = golden_cases.build_case(case)
    P = parts[0]
    g = golden(case)
    # the reference's own sequence, looked up in ITS namespace (pcg_solver.py:996-1008)
    P["Un"] = (1e-200) * np.random.rand(P["NDOF"])                      # :996
    P["DofWeightVector_Eff"] = P["DofWeightVector"][P["LocDofEff"]]     # :997
    for step in range(1, len(P["GlobData"]["TimeStepDelta"])):
        ref.updateTimeStep(P, step)                                     # the reference's own function
        ref.updateBC(P)
        ref.updatePreconditioner(P)
        assert ref.PCG(P) is None
    gd = P["GlobData"]
    assert int(gd["TimeList_Flag"][1]) == int(g["flag"]) and int(gd["TimeList_Iter"][1]) == int(g["iter"])
    assert relerr(P["Un"], g["Un"]) < 1e-8
    # the rebound mat-vec, called through the reference's namespace, against the oracle's restatement of :265-300
    import pcg_oracle
    x = np.cos(np.arange(P["NDOF"]) * 0.1)
    assert relerr(ref.calcMPFint(x, P), pcg_oracle.matvec_local(P, x)) < 1e-13
