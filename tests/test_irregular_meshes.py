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
    ([(20, 120), (5, 40), (30, 90), (7, 30), (13, 70)], False),   # node-blocked patterns of every node-count class: <= 24, <= 32, < 8, <= 16
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


def _mixed_layout_parts(N=9, grid=(1, 1, 2)):
    """Brick parts whose hex8 group is split in two: the first half keeps the node-major slot order (chunked kernels),
    the second half is rewritten direction-major (x dofs of the 8 nodes, then y, then z): the same elements for the
    oracle, but a pattern type the chunked form does not take (colour-by-colour launches that ADD into y)."""
    from pcg_mi355x.brick import Brick, make_parts, block_partition
    b = Brick(N, seed=0)
    parts = make_parts(b, block_partition(b, *grid))
    axis = int(np.argmax(grid))
    perm = np.array([3 * a + d for d in range(3) for a in range(8)])          # new slot -> old slot
    for P in parts:
        g = P["SubDomainData"]["StrucDataList"][0]
        # split by layer parity across the interface direction: the layer next to the interface is colour-launched
        # (interface phase) and shares its lower nodes with chunked elements that do NOT touch the interface
        # (interior phase) - the configuration in which a phase-0 `+=` was overwritten by a phase-1 store
        zlay = np.rint(np.asarray(P["NodeCoordVec"], float)[g["ElemList_LocDofVector"][0] // 3 * 3 + axis]).astype(int)
        top = zlay.max()
        out = []
        for t, (sl, pm_) in enumerate([(np.flatnonzero((top - zlay) % 2 == 1), None), (np.flatnonzero((top - zlay) % 2 == 0), perm)]):
            tbl = np.ascontiguousarray(g["ElemList_LocDofVector"][:, sl])
            sgn = np.ascontiguousarray(g["ElemList_SignVector"][:, sl])
            Ke = g["ElemStiffMat"]
            if pm_ is not None:
                tbl, sgn, Ke = np.ascontiguousarray(tbl[pm_]), np.ascontiguousarray(sgn[pm_]), np.ascontiguousarray(Ke[np.ix_(pm_, pm_)])
            out.append({"ElemTypeId": t, "ElemList_LocDofVector": tbl, "ElemList_LocDofVector_Flat": tbl.ravel(),
                        "ElemList_SignVector": sgn, "ElemList_Ck": g["ElemList_Ck"][sl].copy(), "ElemStiffMat": Ke,
                        "ElemDiagStiffMat": np.diag(Ke).copy(), "N_Elem": tbl.shape[1]})
        P["SubDomainData"] = {"StrucDataList": out, "MixedDataList": {}}
        P["Flat_ElemLocDof"] = np.concatenate([q["ElemList_LocDofVector_Flat"] for q in out])
        P["NCountDof"] = len(P["Flat_ElemLocDof"])
    return b, parts


def check_mixed_layout_multi_part(on_gpu):
    from thread_comm import solve_parts_in_threads
    for N, grid in [(9, (2, 1, 1)), (25, (1, 1, 2))]:
        b, parts = _mixed_layout_parts(N, grid)
        ref = copy.deepcopy(parts)
        out = pcg_oracle.solve_step(ref)
        infos = solve_parts_in_threads(parts, "ebe", on_gpu=on_gpu)
        assert all((i.flag, i.iter) == (infos[0].flag, infos[0].iter) for i in infos)
        assert infos[0].flag == out["flag"] == 0 and abs(infos[0].iter - out["iter"]) <= 1
        for P, R in zip(parts, ref):
            assert relerr(P["Un"], R["Un"]) < 1e-8


def test_mixed_chunked_and_colour_groups_with_neighbours(hostops):
    """a part WITH neighbours whose groups are partly chunkable, partly not.  The colour launches of the
    interface phase add into y, the chunk stores of the interior phase assign: interleaving the phases with the
    exchange would drop contributions, so such parts run the whole operator before the exchange."""
    check_mixed_layout_multi_part(False)
