"""The value dictionary of the assembled operator (PCG_FORMAT_DICTIONARY, csrc/sell.cpp compress_blocks, k_spmv_dict):
every stored 3x3 block becomes a 16-bit index into the table of the matrix's DISTINCT blocks.  Lossless by construction -
blocks are compared by bit pattern and the kernel multiplies the same values in the same order - so the contract is
BIT-IDENTITY with the plain format, on top of the usual parity with the reference fixtures (pcg_solver.py:242-336, :356-598).

CPU tier: the host code (dictionary builder, the driver's format switch, byte accounting) on the test double, whose SpMV
reads through the dictionary.  GPU tier: k_spmv_dict with the table in LDS and read through the caches."""
import ctypes as C

import numpy as np
import pytest

import golden_cases
import pcg_oracle
from pcg_mi355x.brick import Brick, make_parts, block_partition
from util import golden, relerr, check_solution_against_golden


def _spmv_local(op, xl):
    from pcg_mi355x._lib import check
    xe = op.to_engine(xl)
    y = np.empty(op.n)
    pxy = C.c_double()
    check(op._L.pcg_k_spmv_local(op._h, xe.ctypes.data, y.ctypes.data, C.byref(pxy)))
    return op.from_engine(y), pxy.value


def _pair(P, **kw):
    from pcg_mi355x.operator import from_refmeshpart
    return from_refmeshpart(P, kind="sell", **kw), from_refmeshpart(P, kind="dict", **kw)


def _check_identical(P, seed=3, with_diag=True, dot_exact=True, **kw):
    plain, dic = _pair(P, **kw)
    try:
        assert plain.matrix_dictionary() == 0 and dic.matrix_dictionary() > 0
        ip, idc = plain.matrix_info(), dic.matrix_info()
        assert (ip["nnzb"], ip["stored_blocks"], ip["n_slices"]) == (idc["nnzb"], idc["stored_blocks"], idc["n_slices"])
        x = np.random.default_rng(seed).standard_normal(plain.n)
        ya, da = _spmv_local(plain, x)
        yb, db = _spmv_local(dic, x)
        assert np.array_equal(ya, yb)                                    # same values, same order: the same bits
        # the fused p.Ap: identical when the partial sums are grouped alike (same workgroup shape), else a different tree
        if dot_exact and dic.matrix_dictionary() * 80 > 60 * 1024:      # a table beyond 60 KB runs 1024-thread workgroups whatever was asked
            dot_exact = False
        assert da == db if dot_exact else abs(da - db) <= 1e-13 * np.dot(np.abs(x), np.abs(ya))
        if with_diag:                                                    # (diag() of a part with neighbours exchanges)
            assert np.array_equal(plain.diag(), dic.diag())
        assert relerr(yb, pcg_oracle.matvec_local(P, x)) < 1e-14
        bp, fp = plain.operator_cost()
        bd, fd = dic.operator_cost()
        assert fp == fd and bd < 0.2 * bp                                # 4-6 B per stored block instead of 74-76 (+ the table)
        return dic.matrix_dictionary()
    finally:
        plain.close(); dic.close()


@pytest.mark.parametrize("n_types", [1, 3])
def test_dictionary_spmv_is_bit_identical_on_the_test_double(hostops, n_types):
    b = Brick(9, n_types=n_types)
    n_unique = _check_identical(make_parts(b)[0])
    # two material factors x a few stencil positions: a few hundred distinct blocks at most, whatever the mesh size
    assert n_unique < 1500 and n_unique < b.nnz // 9 // 4


@pytest.mark.parametrize("case,part", [("n9_p1", 0), ("n9_p8", 5), ("oct_p3", 1), ("n13_t3_p4_ud", 2), ("n17_p1", 0)])
@pytest.mark.parametrize("kind", ["sell", "dict"])
def test_assembler_path_builds_the_same_operator(hostops, monkeypatch, case, part, kind):
    """pcg_create_asm (rows straight from the assembler; for "dict" the values are never materialised) against
    pcg_asm_fill -> pcg_create -> compress_blocks: the same slice pointers, columns, values / indices + table (same order, same
    counts) and diagonal - compared through a fingerprint of the host arrays - and the same SpMV bits."""
    from pcg_mi355x.operator import from_refmeshpart
    monkeypatch.setenv("PCG_MATRIX_FINGERPRINT", "1")
    _, parts = golden_cases.build_case(case)
    P = parts[part]

    class NoComm:
        rank = 0
        native = False
        def make_hooks(self, op):
            from pcg_mi355x import _lib
            return _lib.CommHooks()
        def reraise(self):
            pass
        def release_stream(self, p):
            pass
    comm = NoComm() if len(parts) > 1 else None
    ops = []
    rpl = 2 if (kind == "sell" and case in ("n9_p8", "n17_p1")) else 0         # 128-row slices for the plain format too
    for stream in ("0", "1"):
        monkeypatch.setenv("PCG_ASM_STREAM", stream)
        ops.append(from_refmeshpart(P, kind=kind, comm=comm, rows_per_lane=rpl))
    try:
        fa, fb = ops[0].matrix_fingerprint(), ops[1].matrix_fingerprint()
        assert fa != 0 and fa == fb
        assert ops[0].matrix_dictionary_info() == ops[1].matrix_dictionary_info()
        assert (ops[0].nnzb, ops[0].matrix_info()) == (ops[1].nnzb, ops[1].matrix_info())
        x = np.random.default_rng(2).standard_normal(ops[0].n)
        ya, da = _spmv_local(ops[0], x)
        yb, db = _spmv_local(ops[1], x)
        assert np.array_equal(ya, yb) and da == db
    finally:
        for op in ops:
            op.close()


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_formats_and_paths_agree_on_random_unstructured_meshes(hostops, monkeypatch, seed):
    """Random connectivity, several pattern sizes and slot orders, random sign masks and random Ck (every block distinct: the
    table is as large as the matrix; capped so that some seeds overflow and keep the plain format), a hub node of high
    valence, rows of very different lengths (ragged slices, padding): the four ways to build the operator - arrays or
    assembler path, plain or dictionary - give the same SpMV bits, and the two paths the same host arrays."""
    from test_irregular_meshes import random_part
    from pcg_mi355x.operator import from_refmeshpart
    monkeypatch.setenv("PCG_MATRIX_FINGERPRINT", "1")
    if seed % 2:
        monkeypatch.setenv("PCG_SPMV_DICT_MAX", "500")                      # fewer entries than distinct blocks: fall back
    spec = [[(8, 200), (4, 150), (9, 60)], [(8, 240), (6, 90)], [(20, 120), (5, 40), (13, 70)], [(4, 500)]][seed]
    P = random_part(97, spec, seed=40 + seed, hub=seed == 1)
    x = np.random.default_rng(seed).standard_normal(P["NDOF"])
    want = pcg_oracle.matvec_local(P, x)
    ys, fps, dicts = {}, {}, {}
    for kind in ("sell", "dict"):
        for stream in ("0", "1"):
            monkeypatch.setenv("PCG_ASM_STREAM", stream)
            op = from_refmeshpart(P, kind=kind)
            try:
                ys[kind, stream] = _spmv_local(op, x)[0]
                fps[kind, stream] = op.matrix_fingerprint()
                dicts[kind, stream] = op.matrix_dictionary()
            finally:
                op.close()
    assert fps["sell", "0"] == fps["sell", "1"] != 0 and fps["dict", "0"] == fps["dict", "1"] != 0
    assert dicts["dict", "0"] == dicts["dict", "1"] and dicts["sell", "0"] == 0
    assert (dicts["dict", "1"] == 0) == bool(seed % 2)
    for k, y in ys.items():
        assert np.array_equal(y, ys["sell", "0"]), k
    assert relerr(ys["dict", "1"], want) < 1e-13


def test_assembler_path_falls_back_when_the_table_overflows(hostops, monkeypatch):
    from pcg_mi355x.operator import from_refmeshpart
    monkeypatch.setenv("PCG_SPMV_DICT_MAX", "20")
    monkeypatch.setenv("PCG_ASM_STREAM", "1")
    P = make_parts(Brick(7, seed=0))[0]
    op = from_refmeshpart(P, kind="dict")
    try:
        assert op.matrix_dictionary() == 0
        x = np.random.default_rng(1).standard_normal(op.n)
        assert relerr(_spmv_local(op, x)[0], pcg_oracle.matvec_local(P, x)) < 1e-14
    finally:
        op.close()


def test_dictionary_info_on_the_test_double(hostops):
    from pcg_mi355x.operator import from_refmeshpart
    op = from_refmeshpart(make_parts(Brick(7, seed=0))[0], kind="dict")
    try:
        info = op.matrix_dictionary_info()
        assert info["distinct_blocks"] > 0 and info["in_lds"] == 0 and info["lds_share"] == 0     # the double has no LDS
    finally:
        op.close()


def test_dictionary_size_does_not_grow_with_the_mesh(hostops):
    from pcg_mi355x.operator import from_refmeshpart
    sizes = []
    for N in (9, 13):
        op = from_refmeshpart(make_parts(Brick(N, seed=0))[0], kind="dict")
        sizes.append(op.matrix_dictionary())
        op.close()
    assert sizes[1] <= 1.25 * sizes[0]


def test_too_many_distinct_blocks_keep_the_plain_format(hostops, monkeypatch):
    from pcg_mi355x.operator import from_refmeshpart
    monkeypatch.setenv("PCG_SPMV_DICT_MAX", "20")
    P = make_parts(Brick(7, seed=0))[0]
    op = from_refmeshpart(P, kind="dict")
    try:
        assert op.matrix_dictionary() == 0
        x = np.random.default_rng(1).standard_normal(op.n)
        assert relerr(_spmv_local(op, x)[0], pcg_oracle.matvec_local(P, x)) < 1e-14
    finally:
        op.close()


def test_dictionary_from_a_scalar_csr_matrix(hostops):
    """pcg_create_csr(block = 3 | PCG_FORMAT_DICTIONARY): an assembled scipy matrix in, dictionary format behind it."""
    import scipy.sparse as sp
    from pcg_mi355x import _lib
    from pcg_mi355x.operator import assemble_bsr3, Operator
    b = Brick(6, n_types=2)
    P = make_parts(b)[0]
    rp, c, v = assemble_bsr3(P["SubDomainData"]["StrucDataList"], b.n_node)
    A = sp.bsr_matrix((v, c, rp), shape=(b.n_dof, b.n_dof)).tocsr()
    op = Operator.from_csr(A.indptr, A.indices, A.data, block=3 | _lib.FORMAT_DICTIONARY)
    ref = Operator.from_csr(A.indptr, A.indices, A.data)
    try:
        assert op.matrix_dictionary() > 0 and ref.matrix_dictionary() == 0
        x = np.random.default_rng(9).standard_normal(b.n_dof)
        assert np.array_equal(op.apply(x), ref.apply(x))
        with pytest.raises(_lib.PcgError, match="needs 3x3 node blocks"):
            Operator.from_csr(A.indptr, A.indices, A.data, block=1 | _lib.FORMAT_DICTIONARY)
    finally:
        op.close(); ref.close()


@pytest.mark.parametrize("case", ["n9_p1", "n9_flag4", "n9_maxiter", "oct_p1", "n17_p1", "goct_p1"])
def test_dictionary_solve_matches_reference_fixture(hostops, case):
    import pcg_mi355x as pm
    _, parts = golden_cases.build_case(case)
    g = golden(case)
    P = parts[0]
    pm.configure(comm=None, operator="dict")
    try:
        pm.update_bc(P); pm.update_preconditioner(P); pm.solve(P, history=True)
        assert pm.get_operator(P).matrix_dictionary() > 0
        info = P["_pcg_mi355x_info"]
        check_solution_against_golden(g, info.flag, info.iter, info.relres, P["Un"], info.history,
                                      tol_u=1e-8 if int(g["flag"]) == 0 else 1e-6)
    finally:
        P.pop("_pcg_mi355x_operator").close()
        pm.configure(comm=None, operator="sell")


@pytest.mark.parametrize("case", ["n9_p8", "oct_p3"])
def test_dictionary_multi_part(hostops, case):
    """Interface rows first, exchange, fix-up - unchanged by the storage format: the group run equals the plain one bit for bit."""
    from pcg_mi355x.group import GroupSolver
    outs = {}
    for kind in ("sell", "dict"):
        _, parts = golden_cases.build_case(case)
        gs = GroupSolver(parts, operator=kind)
        try:
            gs.updateBC(); gs.updatePreconditioner(); gs.PCG(history=True)
            assert all((op.matrix_dictionary() > 0) == (kind == "dict") for op in gs.group.ops)
            outs[kind] = (parts[0]["_pcg_mi355x_info"].history.copy(), [P["Un"].copy() for P in parts])
        finally:
            gs.close()
    assert np.array_equal(outs["sell"][0], outs["dict"][0])
    for a, b in zip(outs["sell"][1], outs["dict"][1]):
        assert np.array_equal(a, b)


# ---- GPU ----------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("lds,block", [("1", "256"), ("1", "512"), ("1", "1024"), ("0", "256")])
def test_dictionary_kernel_is_bit_identical_on_gpu(gpu_lib, monkeypatch, lds, block):
    """k_spmv_dict (table in LDS / read through the caches) against k_spmv on the same matrix: same bits, with and without
    the fused p.Ap, single part and a part with interface rows (two launches: interface slices, interior slices)."""
    monkeypatch.setenv("PCG_SPMV_DICT_LDS", lds)
    monkeypatch.setenv("PCG_SPMV_DICT_BLOCK", block)
    b = Brick(24, seed=0, n_types=2)
    _check_identical(make_parts(b)[0], dot_exact=block == "256")
    parts = make_parts(b, block_partition(b, 2, 1, 2))

    class NoComm:
        rank = 0
        native = False
        def make_hooks(self, op):
            from pcg_mi355x import _lib
            return _lib.CommHooks()
        def reraise(self):
            pass
        def release_stream(self, p):
            pass
    _check_identical(parts[3], with_diag=False, dot_exact=block == "256", comm=NoComm())


@pytest.mark.gpu
@pytest.mark.parametrize("head", ["auto", "100"])
def test_dictionary_larger_than_lds_on_gpu(gpu_lib, monkeypatch, head):
    """A two-level octree mesh with hanging-node patterns has ~2200 distinct blocks (176 KB of table): the 1945 most frequent
    stay in LDS (one 1024-thread workgroup per CU), the rest is read through the caches by the lanes that need it - and the
    same with a head of only 100 entries, so that most waves take both branches.  Bit-identical to the plain format."""
    from pcg_mi355x.octree import TwoLevelMesh, make_octree_parts
    from pcg_mi355x.operator import from_refmeshpart
    if head != "auto":
        monkeypatch.setenv("PCG_SPMV_DICT_LDS_ENTRIES", head)
    P = make_octree_parts(TwoLevelMesh(24, 24, 10, 4, seed=0), 1, axis=0)[0]
    plain, dic = _pair(P)
    try:
        info = dic.matrix_dictionary_info()
        assert info["distinct_blocks"] > 1945
        assert info["in_lds"] == (1945 if head == "auto" else 100) and 0.5 < info["lds_share"] < 1.0
        assert info["lds_share"] > (0.97 if head == "auto" else 0.5)        # frequency order: the head covers most stored blocks
        x = np.random.default_rng(5).standard_normal(plain.n)
        ya, da = _spmv_local(plain, x)
        yb, db = _spmv_local(dic, x)
        assert np.array_equal(ya, yb)
        assert abs(da - db) <= 1e-13 * np.dot(np.abs(x), np.abs(ya))
        assert relerr(yb, pcg_oracle.matvec_local(P, x)) < 1e-14
    finally:
        plain.close(); dic.close()
    if head == "100":                                                       # the brick's 693-entry table with a 100-entry head
        _check_identical(make_parts(Brick(24, seed=0, n_types=2))[0], dot_exact=False)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["n9_p1", "n9_flag4", "oct_p1", "n17_p1"])
def test_dictionary_solve_on_gpu(gpu_lib, case, monkeypatch):
    """Whole solves: against the reference fixture, and - with the workgroup shape of k_spmv (256 threads: the partial sums
    of the fused p.Ap are then grouped alike) - bit-identical to the plain format, iteration for iteration."""
    import pcg_mi355x as pm
    monkeypatch.setenv("PCG_SPMV_DICT_BLOCK", "256")
    outs = {}
    small_table = True
    g = golden(case)
    for kind in ("sell", "dict"):
        _, parts = golden_cases.build_case(case)
        P = parts[0]
        pm.configure(comm=None, device=0, operator=kind)
        try:
            pm.update_bc(P); pm.update_preconditioner(P); pm.solve(P, history=True)
            info = P["_pcg_mi355x_info"]
            outs[kind] = (info.history.copy(), P["Un"].copy())
            check_solution_against_golden(g, info.flag, info.iter, info.relres, P["Un"], info.history,
                                          tol_u=1e-8 if int(g["flag"]) == 0 else 1e-6)
            if kind == "dict":                 # tables beyond 60 KB run 1024-thread workgroups whatever PCG_SPMV_DICT_BLOCK says
                small_table = pm.get_operator(P).matrix_dictionary() * 80 <= 60 * 1024
        finally:
            P.pop("_pcg_mi355x_operator").close()
            pm.configure(comm=None, device=0, operator="sell")
    if small_table:
        assert np.array_equal(outs["sell"][0], outs["dict"][0]) and np.array_equal(outs["sell"][1], outs["dict"][1])
    else:                                    # 1024-thread workgroups: another summation tree for p.Ap - CG's usual sensitivity
        m = max(1, int(0.3 * len(outs["sell"][0])))
        assert np.abs(outs["dict"][0][:m, 2] / outs["sell"][0][:m, 2] - 1).max() < 1e-10
        assert relerr(outs["dict"][1], outs["sell"][1]) < (1e-8 if int(g["flag"]) == 0 else 1e-6)


@pytest.mark.gpu
def test_dictionary_at_1m_dof_on_gpu(gpu_lib):
    """BASELINE configs[1] size: 1 M dof, ~420 distinct blocks among 9.2 M stored ones; operator vs the oracle's CPU mat-vec,
    bit-identity with the plain format, and the dictionary's device footprint."""
    from pcg_mi355x.operator import from_refmeshpart
    b = Brick(70, seed=0)
    P = make_parts(b)[0]
    plain, dic = _pair(P)
    try:
        n_unique = dic.matrix_dictionary()
        assert 0 < n_unique < 1000
        x = np.random.default_rng(0).standard_normal(plain.n)
        ya = plain.apply(x)
        yb = dic.apply(x)
        assert np.array_equal(ya, yb)
        assert relerr(yb, pcg_oracle.matvec_local(P, x, use_c=True)) < 1e-13
        bp, _ = plain.operator_cost()
        bd, _ = dic.operator_cost()
        assert bd < 0.1 * bp
    finally:
        plain.close(); dic.close()
