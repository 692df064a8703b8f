"""-m gpu parity tests for CSR x dense SpMM (row-major B, C) vs the CPU oracle.
Cases mirror /root/reference/test/gtest/spmm_test.cpp:6-136 (n in {1,8,32,64,512} on
util::dims; scaled A / scaled B with alpha in {-10,1,5}; matrix_opt :138) -- the reference
has no device SpMM test, its CPU cases are the model."""
import os

import numpy as np
import pytest
import scipy.sparse as sps
import torch

import gpu_util as G
import spblas_reference_amd as sp
import util
from oracle import oracle
from spblas_reference_amd import generate

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def absprod(values, rowptr, colind, shape, B):
    A = sps.csr_matrix((np.abs(values).astype(np.float64), colind, rowptr), shape=shape)
    return A @ np.abs(B).astype(np.float64)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [1, 8, 32, 64, 512])
@pytest.mark.parametrize("dim", util.dims)
def test_spmm_reference_cases(gpu, dim, n, dtype):
    m, k, nnz = dim
    values, rowptr, colind, shape, _ = generate.generate_csr(m, k, nnz, dtype=dtype)
    B = generate.generate_dense(k, n, dtype=dtype)
    a = G.csr_on_device(values, rowptr, colind, shape, nnz)
    C = torch.full((m, n), float("nan"), dtype=G.dev(B).dtype, device="cuda")
    info = sp.multiply_inspect(a, G.dev(B), C)  # examples/spmm_csr.cpp:45-46 call shape
    sp.multiply(info, a, G.dev(B), C)
    C_ref = oracle.spmm(shape, rowptr, colind, values, B)
    lens = np.diff(rowptr)
    util.assert_parity(G.host(C), C_ref, absprod(values, rowptr, colind, shape, B), dtype, row_len=lens,
                       what=f"spmm {dim} n={n}")
    util.expect_eq_ref(C_ref, G.host(C))


@pytest.mark.parametrize("alpha", [-10, 1, 5])
def test_spmm_scaled_and_matrix_opt(gpu, alpha):
    values, rowptr, colind, shape, nnz = generate.generate_csr(100, 1000, 10000)
    B = generate.generate_dense(1000, 64)
    a = G.csr_on_device(values, rowptr, colind, shape, nnz)
    lens = np.diff(rowptr)
    ab = absprod(values, rowptr, colind, shape, B) * abs(alpha)
    for A_arg, B_arg, kw in ((sp.scaled(alpha, a), G.dev(B), {"scale_a": alpha}),
                             (a, sp.scaled(alpha, G.dev(B)), {"scale_b": alpha}),
                             (sp.scaled(alpha, sp.matrix_opt(a)), G.dev(B), {"scale_a": alpha})):
        C = torch.zeros((100, 64), device="cuda")
        sp.multiply(A_arg, B_arg, C)
        util.assert_parity(G.host(C), oracle.spmm(shape, rowptr, colind, values, B, **kw), ab, np.float32,
                           row_len=lens, what=f"spmm scaled {kw}")


@pytest.mark.parametrize("name", sorted(f for f in os.listdir(GOLDEN) if f.startswith("spmm_")))
def test_spmm_golden_bit_exact(gpu, name):
    g = np.load(os.path.join(GOLDEN, name))
    a = G.csr_on_device(g["values"], g["rowptr"], g["colind"], tuple(g["shape"]), len(g["values"]))
    C = torch.full(g["C"].shape, float("nan"), device="cuda")
    sp.multiply(a, G.dev(g["B"]), C)
    assert np.array_equal(G.host(C), g["C"])


def test_spmm_strided_operands_and_errors(gpu):
    values, rowptr, colind, shape, nnz = generate.generate_csr(64, 80, 900, seed=2)
    a = G.csr_on_device(values, rowptr, colind, shape, nnz)
    Bbig = G.dev(generate.generate_dense(80, 40))
    Cbig = torch.zeros((64, 48), device="cuda")
    Bv, Cv = Bbig[:, :24], Cbig[:, 8:32]  # row stride > n (mdspan with padding)
    sp.multiply(a, Bv, Cv)
    C_ref = oracle.spmm(shape, rowptr, colind, values, G.host(Bv.contiguous()))
    util.assert_parity(G.host(Cv.contiguous()), C_ref,
                       absprod(values, rowptr, colind, shape, G.host(Bv.contiguous())), np.float32,
                       row_len=np.diff(rowptr), what="strided spmm")
    assert float(Cbig[:, :8].abs().sum()) == 0 and float(Cbig[:, 32:].abs().sum()) == 0
    with pytest.raises(ValueError):  # multiply_impl.hpp:70-76
        sp.multiply(a, torch.zeros((79, 8), device="cuda"), torch.zeros((64, 8), device="cuda"))
    with pytest.raises(ValueError):  # neither layout_right nor layout_left: strided in both directions
        sp.multiply(a, torch.zeros((160, 16), device="cuda")[::2, ::2], torch.zeros((64, 8), device="cuda"))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [1, 8, 33, 130])
@pytest.mark.parametrize("layouts", ["left_left", "left_right", "right_left"])
def test_spmm_column_major_operands(gpu, layouts, n, dtype):
    """mdspan_col_major dense operands (detail/mdspan.hpp:31-36; the CPU path takes any layout through mdspan's
    operator(), backend/view_customizations.hpp:230-240; test/gtest/mdspan_overlays.cpp:38-45 builds such views): B and / or
    C column-major, with and without padding between the columns, plan-free and after multiply_inspect, scaled -- against
    the oracle on the same logical matrices."""
    m, k, nnz = 300, 700, 9000
    values, rowptr, colind, shape, _ = generate.generate_csr(m, k, nnz, dtype=dtype, seed=9)
    a = G.csr_on_device(values, rowptr, colind, shape, nnz)
    B = (generate.generate_dense(k, n, dtype=dtype) - 50).astype(dtype)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    lb, lc = layouts.split("_")

    def dense(rows, cols, layout, pad, init=None):
        if layout == "left":   # element (i, j) at i + j * (rows + pad)
            store = torch.full((cols, rows + pad), float("nan"), dtype=tdt, device="cuda")
            view = store[:, :rows].t()
        else:
            store = torch.full((rows, cols + pad), float("nan"), dtype=tdt, device="cuda")
            view = store[:, :cols]
        if init is not None:
            view.copy_(torch.from_numpy(init).cuda())
        return store, view

    C_ref = oracle.spmm(shape, rowptr, colind, values, B, scale_a=-2.5)
    ab = 2.5 * absprod(values, rowptr, colind, shape, B)
    for pad in (0, 5):
        _, Bv = dense(k, n, lb, pad, B)
        c_store, Cv = dense(m, n, lc, pad)
        assert (Bv.stride(1) == 1) == (lb == "right") or n == 1 or k == 1
        for inspect in (False, True):
            Cv.fill_(float("nan"))
            if inspect:
                info = sp.multiply_inspect(a, Bv, Cv)
                sp.multiply(info, sp.scaled(-2.5, a), Bv, Cv)
            else:
                sp.multiply(sp.scaled(-2.5, a), Bv, Cv)
            util.assert_parity(G.host(Cv.contiguous()), C_ref, ab, dtype, row_len=np.diff(rowptr),
                               what=f"spmm {layouts} n={n} pad={pad} inspect={inspect}")
            if pad:  # the padding between rows / columns of C is never written
                assert bool(torch.isnan(c_store[:, -pad:]).all())


def test_spmm_cfg3_shape_properties(gpu):
    """BASELINE cfg3 at reduced row count but full n = 128 panel: linearity in B and a
    sampled-row oracle check (the full 2M x 2M case is bench.py --workload spmm)."""
    m = k = 200_000
    values, rowptr, colind, shape, nnz = generate.uniform_csr_device(m, k, 32, seed=0)
    a = sp.csr_view(values, rowptr, colind, shape, nnz)
    g = torch.Generator(device="cuda").manual_seed(2)
    B1 = torch.rand((k, 128), device="cuda", generator=g)
    B2 = torch.rand((k, 128), device="cuda", generator=g)
    C1, C2, C3 = (torch.empty((m, 128), device="cuda") for _ in range(3))
    sp.multiply(a, B1, C1)
    sp.multiply(a, B2, C2)
    sp.multiply(a, B1 - 3.0 * B2, C3)
    lin = (C3 - (C1 - 3.0 * C2)).abs()
    assert bool((lin <= 8e-6 * (C1.abs() + 3.0 * C2.abs()) + 1e-30).all())
    rows = np.arange(0, m, 997)
    rp = rowptr.cpu().numpy()
    idx = torch.from_numpy(np.concatenate([np.arange(rp[r], rp[r + 1]) for r in rows])).cuda()
    sub_rp = np.concatenate([[0], np.cumsum(rp[rows + 1] - rp[rows])]).astype(np.int32)
    sub_c, sub_v = colind[idx].cpu().numpy(), values[idx].cpu().numpy()
    B_h = B1.cpu().numpy()
    C_ref = oracle.spmm((len(rows), k), sub_rp, sub_c, sub_v, B_h)
    util.assert_parity(C1[torch.from_numpy(rows).cuda()].cpu().numpy(), C_ref,
                       absprod(sub_v, sub_rp, sub_c, (len(rows), k), B_h), np.float32, what="cfg3 sampled rows")


def _banded(m, k, per_row, half_width, rng, clusters=(0,), dup=True):
    """Rows whose columns cluster around the diagonal (and around diagonal + offset for every further cluster):
    neighbouring rows share B rows -- the structure multiply_inspect hands to the LDS-staged matrix-core kernel."""
    rowptr = np.arange(m + 1, dtype=np.int64) * per_row
    centre = (np.arange(m) * (k / m)).astype(np.int64)
    off = rng.integers(-half_width, half_width + 1, (m, per_row))
    which = rng.integers(0, len(clusters), (m, per_row))
    cols = (centre[:, None] + off + np.asarray(clusters)[which]) % k
    if dup:
        cols[:, 1] = cols[:, 0]  # one repeated (row, column) pair per row
    return rowptr.astype(np.int32), cols.reshape(-1).astype(np.int32)


# which kernel takes the qualifying row blocks (csrc/spmm.hip): round 6's default -- the band kernel: B window staged in LDS,
# vector FMAs over the stored entries, single-window and tile-group modes --, its matrix-core sibling for blocks dense in
# their window (opt-in: every block that fits), and the tile kernel of rounds 3 - 5
PANEL_KERNELS = {"band": {}, "band_mfma": {"SPBLAS_GFX950_SPMM_BAND_DENSE": "0"}, "tiles": {"SPBLAS_GFX950_SPMM_BAND": "0"},
                 "band_4_waves_small_chunks": {"SPBLAS_GFX950_SPMM_BAND_WAVES": "4", "SPBLAS_GFX950_SPMM_BAND_CH": "64"}}


@pytest.mark.parametrize("kernel", list(PANEL_KERNELS))
@pytest.mark.parametrize("n", [32, 128, 200])
@pytest.mark.parametrize("clusters", [(0,), (0, 5000, 11000)])
def test_spmm_panel_path_matrix_cores(gpu, n, clusters, kernel, monkeypatch):
    """SpMM inspect is consumed: row blocks whose entries fall into a few aligned 64-column tiles are multiplied
    from LDS-staged B rows (band kernels / tile kernel, see PANEL_KERNELS); the result must equal the
    oracle like every other path, including repeated (row, column) pairs, unsorted columns, a ragged last block, rows of
    more than 64 entries, windows that wrap around or span three clusters (tile groups), n that is not a multiple of 32 and
    n > 128 (two passes)."""
    for k_, v_ in PANEL_KERNELS[kernel].items():
        monkeypatch.setenv(k_, v_)
    monkeypatch.setenv("SPBLAS_GFX950_SPMM_PANEL_MIN", "64")  # below the performance threshold: exercise the kernel
    rng = np.random.default_rng(17)
    m, k, per_row = 20011, 23000, 24 * len(clusters)
    rowptr, colind = _banded(m, k, per_row, 40, rng, clusters)
    nnz = len(colind)
    values = (rng.random(nnz) - 0.3).astype(np.float32)
    B = (rng.random((k, n)) - 0.5).astype(np.float32)
    a = G.csr_on_device(values, rowptr, colind, (m, k), nnz)
    Bd = G.dev(B)
    C = torch.full((m, n), float("nan"), device="cuda")
    info = sp.multiply_inspect(sp.matrix_opt(a), Bd, C)
    mi = info.state_.spmm_info()
    assert mi["inspected"] == 1 and mi["panel_blocks"] >= 0.9 * (m // 32), mi
    sp.multiply(info, sp.scaled(-1.5, a), Bd, C)
    C_ref = oracle.spmm((m, k), rowptr, colind, values, B, scale_a=-1.5)
    util.assert_parity(G.host(C), C_ref, 1.5 * absprod(values, rowptr, colind, (m, k), B), np.float32,
                       row_len=np.diff(rowptr), what=f"panel spmm n={n} clusters={clusters}")
    # the plan-free path (one lane group per row) gives the same answer to rounding
    C2 = torch.full((m, n), float("nan"), device="cuda")
    sp.multiply(sp.scaled(-1.5, a), Bd, C2)
    assert bool(((C - C2).abs() <= 1e-5 * (C.abs() + C2.abs()) + 1e-6).all())


def test_spmm_panel_threshold_default(gpu):
    """Default admission: >= 1/5 dense tiles go to the matrix cores, sparser banded blocks and uniform random
    columns stay with the row-group kernel (MFMA use 0 by construction for cfg3-like inputs)."""
    rng = np.random.default_rng(3)
    m = k = 8192
    x = torch.ones(k, 32, device="cuda")
    C = torch.empty(m, 32, device="cuda")
    for per_row, expect in ((64, True), (8, False)):
        rowptr, colind = _banded(m, k, per_row, 40, rng, dup=False)
        a = G.csr_on_device(np.ones(len(colind), np.float32), rowptr, colind, (m, k), len(colind))
        info = sp.multiply_inspect(a, x, C)
        assert (info.state_.spmm_info()["panel_blocks"] > 0.9 * (m // 32)) == expect, (per_row, info.state_.spmm_info())
    values, rowptr, colind, shape, nnz = generate.uniform_csr_device(200_000, 200_000, 32, seed=0)
    info = sp.multiply_inspect(sp.csr_view(values, rowptr, colind, shape, nnz), torch.ones(200_000, 32, device="cuda"),
                               torch.empty(200_000, 32, device="cuda"))
    assert info.state_.spmm_info() == {"inspected": 1, "panel_blocks": 0, "panel_nnz": 0, "long_rows": 0}


@pytest.mark.parametrize("kernel", list(PANEL_KERNELS))
def test_spmm_panel_path_nonfinite_b_rows(gpu, kernel, monkeypatch):
    for k_, v_ in PANEL_KERNELS[kernel].items():
        monkeypatch.setenv(k_, v_)
    monkeypatch.setenv("SPBLAS_GFX950_SPMM_PANEL_MIN", "64")
    """The dense tile holds zeros where A has no entry; 0 * inf must not leak into rows that do not reference the
    non-finite B row (the reference multiplies stored entries only, multiply_impl.hpp:85-91)."""
    rng = np.random.default_rng(5)
    m, k, n, per_row = 4096, 4096, 64, 24
    rowptr, colind = _banded(m, k, per_row, 30, rng, dup=False)
    nnz = len(colind)
    values = (rng.random(nnz) + 0.5).astype(np.float32)
    B = (rng.random((k, n)) + 0.5).astype(np.float32)
    B[1000, 3] = np.inf
    B[2500, :] = np.nan
    a = G.csr_on_device(values, rowptr, colind, (m, k), nnz)
    C = torch.full((m, n), 7.0, device="cuda")
    info = sp.multiply_inspect(a, G.dev(B), C)
    assert info.state_.spmm_info()["panel_blocks"] > 100
    sp.multiply(info, a, G.dev(B), C)
    C_ref = oracle.spmm((m, k), rowptr, colind, values, B)
    got = G.host(C)
    assert np.array_equal(np.isnan(got), np.isnan(C_ref)) and np.array_equal(np.isinf(got), np.isinf(C_ref))
    fin = np.isfinite(C_ref)
    assert np.isnan(C_ref).any() and np.isinf(C_ref).any()
    np.testing.assert_allclose(got[fin], C_ref[fin], rtol=2e-5)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spmm_long_rows_are_split(gpu, dtype):
    """Rows longer than the plan's nnz window (hub rows) are cut into parts by spmm_long_rows_kernel."""
    rng = np.random.default_rng(9)
    m, k, n = 3000, 5000, 72
    lens = rng.integers(0, 20, m)
    lens[17], lens[1500], lens[2999] = 30000, 9000, 2500
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(rowptr[-1])
    colind = rng.integers(0, k, nnz).astype(np.int32)
    values = (rng.random(nnz) - 0.5).astype(dtype)
    B = (rng.random((k, n)) - 0.5).astype(dtype)
    a = G.csr_on_device(values, rowptr, colind, (m, k), nnz)
    C = torch.full((m, n), float("nan"), dtype=G.dev(B).dtype, device="cuda")
    info = sp.multiply_inspect(a, G.dev(B), C)
    assert info.state_.spmm_info()["long_rows"] == (3 if dtype == np.float32 else 3)
    sp.multiply(info, sp.scaled(2.0, a), G.dev(B), C)
    C_ref = oracle.spmm((m, k), rowptr, colind, values, B, scale_a=2.0)
    util.assert_parity(G.host(C), C_ref, 2.0 * absprod(values, rowptr, colind, (m, k), B), dtype,
                       row_len=np.diff(rowptr), what="spmm long rows")


def test_row_sharded_spmm_and_spgemm_single_rank_hip_path(gpu):
    """sharded.ShardedSpMM / ShardedSpGEMM with their default (HIP) local operators, one rank: the multi-rank logic
    is covered on CPU by tests/test_sharded_cpu.py (gloo, world size 2) with the oracle injected."""
    from spblas_reference_amd import sharded
    m, k, n, nnz = 3000, 2000, 24, 40000
    values, rowptr, colind, shape, _ = generate.generate_csr(m, k, nnz, seed=7)
    a = G.csr_on_device(values, rowptr, colind, shape, nnz)
    b_h = np.random.default_rng(9).random((k, n)).astype(np.float32)
    op = sharded.ShardedSpMM(a, [0, m], n)
    c = op.local(G.dev(b_h))
    assert op.gather_c().data_ptr() == c.data_ptr()
    ref = np.asarray(oracle.spmm(shape, rowptr, colind, values, b_h)).reshape(m, n)
    absr = np.asarray(oracle.spmm(shape, rowptr, colind, np.abs(values), np.abs(b_h))).reshape(m, n)
    util.assert_parity(G.host(c).ravel(), ref.ravel(), absr.ravel(), np.float32, row_len=np.full(m * n, 64), what="sharded SpMM")
    bv, br, bc, bsh, _ = generate.generate_csr(k, m, nnz, seed=8)
    g = sharded.ShardedSpGEMM(a, G.csr_on_device(bv, br, bc, bsh, nnz), [0, m])
    (cr, cc, cv), (off, total) = g.compute()
    n_ref, _ = oracle.spgemm_symbolic(shape, rowptr, colind, bsh, br, bc)
    fr, fc, fv = oracle.spgemm_numeric(shape, rowptr, colind, values, bsh, br, bc, bv, capacity=n_ref)
    assert (off, total) == (0, n_ref) and np.array_equal(G.host(cr), fr) and np.array_equal(G.host(cc), fc[:n_ref])
    assert np.allclose(G.host(cv), fv[:n_ref], rtol=1e-5, atol=1e-3)
