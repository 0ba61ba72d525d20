"""Robustness on irregular inputs (CPU test double): random unstructured connectivity with several
pattern sizes, high-valence nodes, empty groups, parts whose nodes are all on the interface.  The
checker is the oracle's EBE mat-vec on the same tables."""
import copy

import numpy as np
import pytest

import pcg_oracle
import pcg_mi355x as pm
from pcg_mi355x.operator import from_refmeshpart
from util import relerr


def random_part(n_nodes, groups_spec, seed, hub=False):
    """groups_spec: list of (nodes_per_element, n_elements).  Elements connect random distinct nodes
    (plus, with hub=True, node 0 in every 3rd element: a valence far above any structured mesh)."""
    rng = np.random.default_rng(seed)
    groups = []
    for t, (k, ne) in enumerate(groups_spec):
        nd = 3 * k
        nodes = np.empty((ne, k), np.int64)
        for e in range(ne):
            base = rng.integers(0, n_nodes)
            cand = (base + rng.choice(min(n_nodes, 40), size=k, replace=False)) % n_nodes     # local-ish connectivity
            if hub and e % 3 == 0:
                cand[0] = 0
                cand = np.unique(cand)
                while len(cand) < k:
                    cand = np.unique(np.append(cand, rng.integers(0, n_nodes)))
            nodes[e] = cand[:k]
        # slot order: interleaved (node-major) for even types, direction-major for odd ones
        if t % 2 == 0:
            dof = (3 * nodes[:, :, None] + np.arange(3)[None, None, :]).reshape(ne, nd)
        else:
            dof = np.concatenate([3 * nodes + d for d in range(3)], axis=1)
        M = rng.standard_normal((nd, nd))
        Ke = M @ M.T + nd * np.eye(nd)
        tbl = np.ascontiguousarray(dof.T)
        groups.append({"ElemTypeId": t, "ElemList_LocDofVector": tbl, "ElemList_LocDofVector_Flat": tbl.ravel(),
                       "ElemList_SignVector": rng.random(tbl.shape) < 0.3, "ElemList_Ck": rng.random(ne) + 0.5,
                       "ElemStiffMat": Ke, "ElemDiagStiffMat": np.diag(Ke).copy(), "N_Elem": ne})
    n = 3 * n_nodes
    flat = np.concatenate([g["ElemList_LocDofVector_Flat"] for g in groups]) if groups else np.zeros(0, np.int64)
    fixed = np.zeros(n, bool); fixed[:6] = True
    return {"Id": 0, "SubDomainData": {"StrucDataList": groups, "MixedDataList": {}}, "NDOF": n, "NNode": n_nodes,
            "DofVector": np.arange(n), "NodeIdVector": np.arange(n_nodes), "RefLoadVector": rng.standard_normal(n),
            "Ud": np.zeros(n), "Un": np.zeros(n), "LocDofEff": np.flatnonzero(~fixed), "LocFixedDof": np.flatnonzero(fixed),
            "Flat_ElemLocDof": flat, "NCountDof": len(flat), "NbrMPIdVector": [], "OvrlpLocalDofVecList": [],
            "DofWeightVector": np.ones(n), "NodeWeightVector": np.ones(n_nodes), "MPList_RefPlotDofIndicesList": [],
            "NodeCoordVec": rng.random(n) * 10,
            "GlobData": {"GlobNDof": n, "GlobNDofEff": int((~fixed).sum()), "MaxIter": 5000, "Tol": 1e-9,
                         "TimeStepDelta": [0, 1], "TimeStepCount": 1, "FintCalcMode": "outbin",
                         "MP_TimeRecData": {"dT_FileRead": 0.0, "dT_Calc": 0.0, "dT_CommWait": 0.0, "t0": 0.0},
                         "TimeList_Flag": np.zeros(2), "TimeList_RelRes": np.zeros(2), "TimeList_Iter": np.zeros(2)}}


SPECS = [
    ([(8, 300)], False),                       # hex-like, irregular graph: chunked path with many sub-colours
    ([(8, 200), (4, 150), (9, 60)], False),    # mixed pattern sizes: chunked + generic paths together
    ([(8, 240), (6, 90)], True),               # hub node shared by ~110 elements
    ([(4, 500)], False),                       # only non-hex patterns
]


@pytest.mark.parametrize("kind", ["sell", "ebe"])
@pytest.mark.parametrize("spec,hub", SPECS)
def test_random_unstructured_operator(hostops, kind, spec, hub):
    P = random_part(97, spec, seed=len(spec) * 7 + hub, hub=hub)
    P["DofWeightVector_Eff"] = P["DofWeightVector"][P["LocDofEff"]]
    R = copy.deepcopy(P)
    try:
        op = from_refmeshpart(P, kind=kind)
    except pm.PcgError as e:
        # the only accepted refusal: a colouring limit, reported loudly (never a silent wrong answer)
        assert "colour" in str(e) and hub
        return
    x = np.random.default_rng(3).standard_normal(P["NDOF"])
    ref = pcg_oracle.matvec_local(R, x)
    assert relerr(op.apply(x), ref) < 1e-13
    assert relerr(op.diag(), pcg_oracle.matvec_local(R, None, "Preconditioner")) < 1e-13
    op.close()


def test_random_unstructured_solve(hostops):
    P = random_part(60, [(8, 150), (4, 80)], seed=11)
    P["DofWeightVector_Eff"] = P["DofWeightVector"][P["LocDofEff"]]
    R = copy.deepcopy(P)
    for kind in ("sell", "ebe"):
        Q = copy.deepcopy(P)
        pm.configure(comm=None, operator=kind)
        pm.update_bc(Q); pm.update_preconditioner(Q); pm.solve(Q)
        if kind == "sell":
            out = pcg_oracle.solve_step([R])
        assert Q["GlobData"]["TimeList_Flag"][1] == out["flag"] == 0
        assert abs(Q["GlobData"]["TimeList_Iter"][1] - out["iter"]) <= 1
        assert relerr(Q["Un"], R["Un"]) < 1e-8
    pm.configure(comm=None, operator="sell")


def test_empty_group_and_isolated_nodes(hostops):
    P = random_part(50, [(8, 40), (4, 0)], seed=5)       # second group has no element; many nodes untouched
    P["DofWeightVector_Eff"] = P["DofWeightVector"][P["LocDofEff"]]
    x = np.random.default_rng(1).standard_normal(P["NDOF"])
    ref = pcg_oracle.matvec_local(copy.deepcopy(P), x)
    for kind in ("sell", "ebe"):
        op = from_refmeshpart(P, kind=kind)
        y = op.apply(x)
        assert relerr(y, ref) < 1e-13 and np.all(y[ref == 0] == 0)
        op.close()
