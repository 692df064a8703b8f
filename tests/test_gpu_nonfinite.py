"""-m gpu: NaN and infinity travel exactly where the reference's loops would carry them.

/root/reference/include/spblas/algorithms/multiply_impl.hpp:33-53 computes y_i as the plain sum of a_ij * x_j over the
stored entries of row i: a non-finite x_j reaches the rows that store column j and no other row, and beta is never read
(the output is overwritten).  A re-tiled plan multiplies padding entries as well (value 0, column 0 of the slice): those
products must never reach y even when x holds NaN at the column a pad points to.
"""
import numpy as np
import pytest
import torch

import gpu_util as G
import spblas_reference_amd as sp
from oracle import oracle
from spblas_reference_amd import _capi, generate

pytestmark = pytest.mark.gpu

ALGS = {"noplan": None, "auto": _capi.SPMV_AUTO, "vector": _capi.SPMV_VECTOR, "rowblock": _capi.SPMV_ROWBLOCK,
        "sliced": _capi.SPMV_SLICED}


@pytest.mark.parametrize("alg,enc8", [(alg, "0") for alg in ALGS] + [("sliced", "2")])  # 2 = one-byte row codes forced
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spmv_nonfinite_x_reaches_exactly_the_rows_that_store_the_column(gpu, monkeypatch, alg, dtype, enc8):
    monkeypatch.setenv("SPBLAS_GFX950_PB_ENC8", enc8)
    m, n, nnz = 30000, 50000, 300000
    values, rowptr, colind, shape, _ = generate.generate_csr(m, n, nnz, dtype=dtype, seed=51)
    rng = np.random.default_rng(6)
    x_h = (rng.random(n) + 0.5).astype(dtype)
    # first column of several slices (where pads point), a few random ones, +/- infinity too
    bad = np.concatenate([[0, 1, 20480, 40960], rng.integers(0, n, 12)])
    x_h[bad[::3]] = np.nan
    x_h[bad[1::3]] = np.inf
    x_h[bad[2::3]] = -np.inf
    a = sp.csr_view(G.dev(values), G.dev(rowptr), G.dev(colind), shape, nnz)
    x = G.dev(x_h)
    y = torch.full((m,), float("nan"), dtype=x.dtype, device="cuda")
    if ALGS[alg] is None:
        sp.multiply(a, x, y)
    else:
        info = sp.multiply_inspect(a, x, y, alg=ALGS[alg])
        sp.multiply(info, a, x, y)
    torch.cuda.synchronize()
    with np.errstate(invalid="ignore", over="ignore"):
        y_ref = oracle.spmv(shape, rowptr, colind, values, x_h)
    got = G.host(y)
    touched = np.zeros(m, bool)
    touched[np.repeat(np.arange(m), np.diff(rowptr))[np.isin(colind, bad)]] = True
    assert touched.sum() > 20
    assert np.array_equal(np.isfinite(got), ~touched), "non-finite rows differ from the rows that store a bad column"
    assert np.array_equal(np.isfinite(y_ref), ~touched)
    # rows with +inf and -inf (or NaN) -> NaN; with one sign of infinity only -> that infinity: same as the oracle
    assert np.array_equal(np.isnan(got), np.isnan(y_ref))
    assert np.array_equal(got[np.isinf(y_ref)], y_ref[np.isinf(y_ref)])
    ok = ~touched
    tol = 1e-6 if dtype == np.float32 else 1e-12
    assert np.all(np.abs(got[ok] - y_ref[ok]) <= tol * np.abs(y_ref[ok]) + 1e-30)  # all terms positive


@pytest.mark.parametrize("ncols", [4, 64])
@pytest.mark.parametrize("inspect", [False, True])
def test_spmm_nonfinite_rows_of_b(gpu, ncols, inspect):
    m, k, nnz = 8000, 9000, 100000
    values, rowptr, colind, shape, _ = generate.generate_csr(m, k, nnz, seed=52)
    rng = np.random.default_rng(7)
    B_h = (rng.random((k, ncols)) + 0.5).astype(np.float32)
    bad_rows = rng.integers(0, k, 6)
    B_h[bad_rows[:3], 1] = np.nan        # one column of C only
    B_h[bad_rows[3:], ncols - 1] = np.inf
    a = sp.csr_view(G.dev(values), G.dev(rowptr), G.dev(colind), shape, nnz)
    B = G.dev(B_h)
    C = torch.full((m, ncols), float("nan"), device="cuda")
    if inspect:
        sp.multiply(sp.multiply_inspect(a, B, C), a, B, C)
    else:
        sp.multiply(a, B, C)
    torch.cuda.synchronize()
    with np.errstate(invalid="ignore", over="ignore"):
        ref = oracle.spmm(shape, rowptr, colind, values, B_h)
    got = G.host(C)
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref))
    fin = np.isfinite(ref)
    np.testing.assert_allclose(got[fin], ref[fin], rtol=2e-5)


def test_spgemm_nonfinite_values_stay_in_their_products(gpu):
    m, k, n = 2000, 1500, 1800
    av, ar, ac, ash, annz = generate.generate_csr(m, k, 20000, seed=53)
    bv, br, bc, bsh, bnnz = generate.generate_csr(k, n, 15000, seed=54)
    av, bv = av.copy() + 0.5, bv.copy() + 0.5
    av[[5, 700, 19999]] = [np.nan, np.inf, -np.inf]
    bv[[11, 9000]] = [np.inf, np.nan]
    a = sp.csr_view(G.dev(av), G.dev(ar), G.dev(ac), ash, annz)
    b = sp.csr_view(G.dev(bv), G.dev(br), G.dev(bc), bsh, bnnz)
    c_rp = torch.zeros(m + 1, dtype=torch.int32, device="cuda")
    c = sp.csr_view(None, c_rp, None, (m, n), 0)
    info = sp.multiply_compute(a, b, c)
    cn = info.result_nnz()
    c_val = torch.zeros(cn, device="cuda")
    c.update(c_val, c_rp, torch.zeros(cn, dtype=torch.int32, device="cuda"), (m, n), cn)
    for fill in range(3):  # one-shot fill, recording fill, fill by rank
        c_val.fill_(7.0)
        sp.multiply_fill(info, a, b, c)
        torch.cuda.synchronize()
        with np.errstate(invalid="ignore", over="ignore"):
            cr, cc, cv = oracle.spgemm_numeric(ash, ar, ac, av, bsh, br, bc, bv, capacity=cn)
        got = G.host(c_val)
        assert np.array_equal(G.host(c.colind()), cc)
        assert np.array_equal(np.isnan(got), np.isnan(cv)), f"fill {fill}"
        assert np.array_equal(np.isinf(got), np.isinf(cv)) and np.array_equal(got[np.isinf(cv)], cv[np.isinf(cv)])
        fin = np.isfinite(cv)
        np.testing.assert_allclose(got[fin], cv[fin], rtol=2e-5)
