"""-m gpu: y = alpha * A x + beta * y  and  C = alpha * A B + beta * C  through the C ABI.

The reference's multiply() overwrites its output (algorithms/multiply_impl.hpp:33-53), and the host layers pass
beta = 0; the C ABI (include/spblas_gfx950.h: spblas_gfx950_spmv / spblas_gfx950_spmm) takes the general form the vendor
calls it replaces take (rocsparse_spmv's alpha / beta, vendor/rocsparse/detail/spmv_impl.hpp:60-77).  Every kernel family
has its own epilogue -- lane groups, row blocks + long-row fix-up, sliced reduce, K-split combine, row map with pieces and
empty rows, hub rows, the transposed scatter, SpMM row groups / long rows / panels -- so each is driven here with
beta != 0 on a y that holds data, and with beta = 0 on a y full of NaN (which must not be read).
"""
import ctypes

import numpy as np
import pytest
import torch

import gpu_util as G
import spblas_reference_amd as sp
import util
from oracle import oracle
from spblas_reference_amd import _capi, generate

pytestmark = pytest.mark.gpu


def _ct(dtype):
    return ctypes.c_float if dtype == np.float32 else ctypes.c_double


def _vt(dtype):
    return _capi.F32 if dtype == np.float32 else _capi.F64


def _matrix(kind, dtype, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        m, n = 30000, 50000
        lens = rng.integers(0, 20, m)
    elif kind == "skewed":  # long rows, empty rows: row map, pieces, hub rows, variable bins
        m, n = 20000, 40000
        lens = np.minimum(rng.zipf(1.4, m), 30000)
        lens[rng.random(m) < 0.4] = 0
        lens[11] = 35000
    else:
        raise KeyError(kind)
    lens = lens.astype(np.int64)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(rowptr[-1])
    colind = rng.integers(0, n, nnz).astype(np.int32)
    values = (rng.random(nnz) - 0.5).astype(dtype)
    return (m, n), rowptr, colind, values


PLANS = {"noplan": None, "vector": _capi.SPMV_VECTOR, "rowblock": _capi.SPMV_ROWBLOCK, "sliced": _capi.SPMV_SLICED}


# (plan, matrix kind, test hooks -- the hooks shape the SLICED plan only)
CASES = [(p, k, {}) for p in PLANS for k in ("uniform", "skewed")] + [
    ("sliced", "uniform", {"SPBLAS_GFX950_PB_KSPLIT": "4"}),     # partial sums + combine kernel
    ("sliced", "uniform", {"SPBLAS_GFX950_PB_ENC8": "2"}),       # one-byte row codes
    ("sliced", "skewed", {"SPBLAS_GFX950_PB_HUB_LEN": "300"}),   # hub rows kept out of the tiles
    ("sliced", "skewed", {"SPBLAS_GFX950_PB_RITEMS": "2"}),      # reduce work items with a K split per group
]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("plan,kind,hooks", CASES)
def test_spmv_alpha_beta_every_epilogue(gpu, monkeypatch, plan, kind, hooks, dtype):
    for k, v in hooks.items():
        monkeypatch.setenv(k, v)
    shape, rowptr, colind, values = _matrix(kind, dtype, 81)
    m, n = shape
    nnz = int(rowptr[-1])
    rng = np.random.default_rng(5)
    x_h = (rng.random(n) - 0.5).astype(dtype)
    y0_h = (rng.random(m) - 0.5).astype(dtype)
    a = G.csr_on_device(values, rowptr, colind, shape, nnz)
    xd = G.dev(x_h)
    lib, hd = _capi.lib(), sp.api._Handle.current(xd.device)
    plan_ptr = None
    if PLANS[plan] is not None:
        y_probe = torch.empty(m, dtype=xd.dtype, device="cuda")
        info = sp.multiply_inspect(a, xd, y_probe, alg=PLANS[plan])
        assert info.state_.info()["alg"] == PLANS[plan]
        plan_ptr = info.state_.plan
    ax = oracle.spmv(shape, rowptr, colind, values, x_h).astype(np.float64)
    absrow = oracle.spmv_absrow(rowptr, colind, values, x_h)
    lens = np.diff(rowptr)
    for alpha, beta in ((1.0, 0.0), (-2.0, 0.0), (1.0, 1.0), (0.5, -3.0), (0.0, 2.0)):
        y = G.dev(y0_h.copy()) if beta != 0.0 else torch.full((m,), float("nan"), dtype=xd.dtype, device="cuda")
        al, be = _ct(dtype)(alpha), _ct(dtype)(beta)
        sp.api.check(lib.spblas_gfx950_spmv(hd.h, plan_ptr, _capi.OP_N, m, n, nnz, ctypes.byref(al), sp.api._ptr(a.rowptr()),
                                            sp.api._ptr(a.colind()), sp.api._ptr(a.values()), sp.api._ptr(xd),
                                            ctypes.byref(be), sp.api._ptr(y), _capi.I32, _vt(dtype)), "spmv")
        torch.cuda.synchronize()
        ref = alpha * ax + (beta * y0_h.astype(np.float64) if beta != 0.0 else 0.0)
        bound = abs(alpha) * absrow + (abs(beta) * np.abs(y0_h) if beta != 0.0 else 0.0)
        util.assert_parity(G.host(y), ref, bound, dtype, row_len=lens + 1, what=f"{plan} {kind} {hooks} alpha={alpha} beta={beta}")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spmv_transposed_alpha_beta(gpu, dtype):
    # OP_T: y (n entries) = alpha * A^T x + beta * y, the CSC / transposed(csr) slot
    shape, rowptr, colind, values = _matrix("uniform", dtype, 82)
    m, n = shape
    nnz = int(rowptr[-1])
    rng = np.random.default_rng(6)
    x_h, y0_h = (rng.random(m) - 0.5).astype(dtype), (rng.random(n) - 0.5).astype(dtype)
    a = G.csr_on_device(values, rowptr, colind, shape, nnz)
    xd = G.dev(x_h)
    lib, hd = _capi.lib(), sp.api._Handle.current(xd.device)
    tr, tc, tv = oracle.transpose(shape, rowptr, colind, values)
    atx = oracle.spmv((n, m), tr, tc, tv, x_h).astype(np.float64)
    absrow = oracle.spmv_absrow(tr, tc, tv, x_h)
    for alpha, beta in ((1.0, 0.0), (1.5, 1.0), (-0.5, 2.0)):
        y = G.dev(y0_h.copy()) if beta != 0.0 else torch.full((n,), float("nan"), dtype=xd.dtype, device="cuda")
        al, be = _ct(dtype)(alpha), _ct(dtype)(beta)
        sp.api.check(lib.spblas_gfx950_spmv(hd.h, None, _capi.OP_T, m, n, nnz, ctypes.byref(al), sp.api._ptr(a.rowptr()),
                                            sp.api._ptr(a.colind()), sp.api._ptr(a.values()), sp.api._ptr(xd),
                                            ctypes.byref(be), sp.api._ptr(y), _capi.I32, _vt(dtype)), "spmv op=T")
        torch.cuda.synchronize()
        ref = alpha * atx + (beta * y0_h.astype(np.float64) if beta != 0.0 else 0.0)
        bound = abs(alpha) * absrow + (abs(beta) * np.abs(y0_h) if beta != 0.0 else 0.0)
        util.assert_parity(G.host(y), ref, bound, dtype, row_len=np.diff(tr) + 1, what=f"op=T alpha={alpha} beta={beta}")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("kind", ["uniform", "long_rows", "banded"])
@pytest.mark.parametrize("ncols", [1, 8, 64, 130])
def test_spmm_alpha_beta_every_kernel(gpu, kind, ncols, dtype):
    rng = np.random.default_rng(83)
    if kind == "uniform":
        m, k = 6000, 7000
        lens = rng.integers(0, 30, m)
    elif kind == "long_rows":
        m, k = 3000, 60000
        lens = rng.integers(0, 10, m)
        lens[7], lens[2999] = 30000, 12000
    else:  # dense 32-row blocks inside a narrow band: the matrix-core panel kernel (fp32) takes them after inspect
        m, k = 4096, 4096
        lens = np.full(m, 48)
    lens = lens.astype(np.int64)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(rowptr[-1])
    rows = np.repeat(np.arange(m), lens)
    if kind == "banded":
        colind = np.clip(rows + rng.integers(-40, 40, nnz), 0, k - 1).astype(np.int32)
    else:
        colind = rng.integers(0, k, nnz).astype(np.int32)
    values = (rng.random(nnz) - 0.5).astype(dtype)
    B_h = (rng.random((k, ncols)) - 0.5).astype(dtype)
    C0_h = (rng.random((m, ncols)) - 0.5).astype(dtype)
    a = G.csr_on_device(values, rowptr, colind, (m, k), nnz)
    B = G.dev(B_h)
    lib, hd = _capi.lib(), sp.api._Handle.current(B.device)
    ab = oracle.spmm((m, k), rowptr, colind, values, B_h).astype(np.float64)
    ab_abs = oracle.spmm((m, k), rowptr, colind, np.abs(values), np.abs(B_h)).astype(np.float64)
    tol = np.maximum(util.TOL[np.dtype(dtype)], (lens + 1) * np.finfo(dtype).eps)[:, None]
    for inspect in (False, True):
        plan_ptr = None
        if inspect:
            info = sp.multiply_inspect(a, B, torch.empty((m, ncols), dtype=B.dtype, device="cuda"))
            plan_ptr = info.state_.plan
            if kind == "banded" and dtype == np.float32:
                assert info.state_.spmm_info()["panel_blocks"] > 0
            if kind == "long_rows":
                assert info.state_.spmm_info()["long_rows"] >= 1
        for alpha, beta in ((1.0, 0.0), (1.0, 1.0), (-0.5, 2.0), (0.0, 3.0)):
            C = G.dev(C0_h.copy()) if beta != 0.0 else torch.full((m, ncols), float("nan"), dtype=B.dtype, device="cuda")
            al, be = _ct(dtype)(alpha), _ct(dtype)(beta)
            sp.api.check(lib.spblas_gfx950_spmm(hd.h, plan_ptr, m, k, ncols, nnz, ctypes.byref(al), sp.api._ptr(a.rowptr()),
                                                sp.api._ptr(a.colind()), sp.api._ptr(a.values()), sp.api._ptr(B), ncols,
                                                ctypes.byref(be), sp.api._ptr(C), ncols, _capi.I32, _vt(dtype)), "spmm")
            torch.cuda.synchronize()
            ref = alpha * ab + (beta * C0_h.astype(np.float64) if beta != 0.0 else 0.0)
            bound = abs(alpha) * ab_abs + (abs(beta) * np.abs(C0_h) if beta != 0.0 else 0.0)
            err = np.abs(G.host(C).astype(np.float64) - ref)
            assert np.all(err <= tol * bound + 1e-300), \
                f"{kind} n={ncols} inspect={inspect} alpha={alpha} beta={beta}: worst {np.max(err - tol * bound)}"
