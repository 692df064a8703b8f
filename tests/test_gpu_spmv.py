"""-m gpu parity tests for CSR SpMV: the HIP path (through the C ABI) vs the CPU oracle.

Cases mirror /root/reference/test/gtest/device/spmv_test.cpp:11-146 (CsrView SpMV,
SpMV_Ascaled, SpMV_BScaled on util::dims with alpha in {-10,1,5}) and add the edge cases
SURVEY.md section 8c lists (empty/ragged rows, unsorted + duplicate columns, one very long
row, int64 offsets, fp64, inspect/execute, matrix_opt reuse, CSC/transposed operand).
"""
import ctypes
import os

import numpy as np
import pytest
import torch

import gpu_util as G
import spblas_reference_amd as sp
import util
from oracle import oracle
from spblas_reference_amd import _capi, generate

pytestmark = pytest.mark.gpu


def cdiv(a, b):
    return -(-a // b)
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
ALGS = {"noplan": None, "auto": _capi.SPMV_AUTO, "vector": _capi.SPMV_VECTOR, "rowblock": _capi.SPMV_ROWBLOCK,
        "sliced": _capi.SPMV_SLICED}


def run_spmv(a_view, x, m, alg, scale_a=None, scale_x=None):
    y = torch.full((m,), float("nan"), dtype=x.dtype, device=x.device)  # beta = 0 must not read y
    a = sp.scaled(scale_a, a_view) if scale_a is not None else a_view
    b = sp.scaled(scale_x, x) if scale_x is not None else x
    if alg is None:
        sp.multiply(a, b, y)
    else:
        info = sp.multiply_inspect(a, b, y, alg=alg)
        sp.multiply(info, a, b, y)
    return G.host(y)


def check(values, rowptr, colind, shape, x, y, scale=1.0, what="", ref_cmp=True):
    y_ref = oracle.spmv(shape, rowptr, colind, values, x, scale_a=None if scale == 1.0 else scale)
    exact, absrow = util.spmv_exact(rowptr, colind, values, x)
    absrow = absrow * abs(scale)
    lens = np.diff(rowptr)
    util.assert_parity(y, y_ref, absrow, values.dtype, row_len=lens, what=what + " vs oracle")
    util.assert_parity(y, exact * scale, absrow, values.dtype, row_len=lens, what=what + " vs float64")
    if ref_cmp:  # the reference's own comparator (test/gtest/util.hpp:7-23); it is relative to the
        util.expect_eq_ref(y_ref, y)  # RESULT, so it only applies to the sign-definite data its tests use


@pytest.mark.parametrize("alg", list(ALGS))
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("dim", util.dims)
def test_spmv_reference_device_test(gpu, dim, dtype, alg):
    # device/spmv_test.cpp:11-49: b = ones
    m, n, nnz = dim
    values, rowptr, colind, shape, _ = generate.generate_csr(m, n, nnz, dtype=dtype)
    a = G.csr_on_device(values, rowptr, colind, shape, nnz)
    x = np.ones(n, dtype=dtype)
    y = run_spmv(a, G.dev(x), m, ALGS[alg])
    check(values, rowptr, colind, shape, x, y, what=f"spmv {dim} {alg}")


@pytest.mark.parametrize("alpha", [-10, 1, 5])
@pytest.mark.parametrize("dim", util.dims)
def test_spmv_scaled_views(gpu, dim, alpha):
    # device/spmv_test.cpp:51-146: scaled(alpha, a) and scaled(alpha, b)
    m, n, nnz = dim
    values, rowptr, colind, shape, _ = generate.generate_csr(m, n, nnz)
    a = G.csr_on_device(values, rowptr, colind, shape, nnz)
    x = np.ones(n, dtype=np.float32)
    for kw in ({"scale_a": alpha}, {"scale_x": alpha}):
        for alg in (None, _capi.SPMV_ROWBLOCK):
            y = run_spmv(a, G.dev(x), m, alg, **kw)
            ref = util.naive_spmv(rowptr, colind, values, x, alpha_a=kw.get("scale_a"), alpha_b=kw.get("scale_x"))
            util.expect_eq_ref(ref, y)
            check(values, rowptr, colind, shape, x, y, scale=float(alpha), what=f"scaled {kw}")
    # both at once: alpha = product of all factors (detail/view_inspectors.hpp:55-77)
    y = run_spmv(a, G.dev(x), m, None, scale_a=alpha, scale_x=2.0)
    check(values, rowptr, colind, shape, x, y, scale=2.0 * alpha, what="scaled both")


@pytest.mark.parametrize("alg", list(ALGS))
@pytest.mark.parametrize("name", sorted(f for f in os.listdir(GOLDEN) if f.startswith("spmv_")))
def test_spmv_golden_bit_exact(gpu, name, alg):
    g = np.load(os.path.join(GOLDEN, name))
    a = G.csr_on_device(g["values"], g["rowptr"], g["colind"], tuple(g["shape"]), len(g["values"]))
    kw = {}
    if "scale_a" in g:
        kw["scale_a"] = float(g["scale_a"])
    if "scale_x" in g:
        kw["scale_x"] = float(g["scale_x"])
    y = run_spmv(a, G.dev(g["x"]), int(g["shape"][0]), ALGS[alg], **kw)
    assert np.array_equal(y, g["y"]), f"{name} {alg}: integer-valued fixture must be bit exact"


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("alg", list(ALGS))
def test_spmv_ragged_powerlaw_int64_offsets(gpu, dtype, alg):
    rng = np.random.default_rng(5)
    m, n = 3000, 4099
    lens = np.minimum(rng.zipf(1.6, m), 20000).astype(np.int64)
    lens[rng.random(m) < 0.3] = 0
    lens[17] = 9000
    lens[2998] = 5000
    rowptr = np.zeros(m + 1, np.int64)
    np.cumsum(lens, out=rowptr[1:])
    nnz = int(rowptr[-1])
    colind = rng.integers(0, n, nnz).astype(np.int32)
    values = (rng.random(nnz) - 0.5).astype(dtype)
    x = (rng.random(n) - 0.5).astype(dtype)
    for off64 in (False, True):
        a = G.csr_on_device(values, rowptr, colind, (m, n), nnz, offset64=off64)
        y = run_spmv(a, G.dev(x), m, ALGS[alg])
        check(values, rowptr.astype(np.int32), colind, (m, n), x, y, what=f"ragged {alg} off64={off64}", ref_cmp=False)


@pytest.mark.parametrize("ksplit", [0, 1, 4])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spmv_sliced_many_tiles_and_value_update(gpu, dtype, ksplit, monkeypatch):
    """The LDS-sliced re-tiling with tiny tiles (test hook env vars) so that a small matrix
    spans many (slice, bin) segments, incl. empty segments, ragged rows and a long row;
    then values change in place and the plan is refreshed with update_values."""
    monkeypatch.setenv("SPBLAS_GFX950_SLICE_COLS", "100")
    monkeypatch.setenv("SPBLAS_GFX950_SLICE_ROWS", "64")
    monkeypatch.setenv("SPBLAS_GFX950_PB_KSPLIT", str(ksplit))  # slice split of the reduce (0 = heuristic)
    rng = np.random.default_rng(11)
    m, n = 1500, 2111
    lens = rng.integers(0, 30, m)
    lens[rng.random(m) < 0.2] = 0
    lens[700] = 5000
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(rowptr[-1])
    colind = rng.integers(0, n, nnz).astype(np.int32)
    colind[rowptr[3]:rowptr[4]] = 2110  # a row living entirely in the last, narrower slice
    values = (rng.random(nnz) + 0.5).astype(dtype)
    x = (rng.random(n) + 0.5).astype(dtype)
    a = G.csr_on_device(values, rowptr, colind, (m, n), nnz)
    xd = G.dev(x)
    y = torch.full((m,), float("nan"), dtype=xd.dtype, device="cuda")
    info = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED)
    pi = info.state_.info()
    assert pi["alg"] == _capi.SPMV_SLICED and pi["n_slices"] == 22
    sp.multiply(info, sp.scaled(2.0, a), xd, y)
    check(values, rowptr, colind, (m, n), x, G.host(y), scale=2.0, what="sliced tiles")
    a.values().mul_(0.5)
    info.state_.update_values(a.values())
    sp.multiply(info, a, xd, y)
    check(values * dtype(0.5), rowptr, colind, (m, n), x, G.host(y), what="sliced after update_values")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spmv_sliced_empty_trailing_runs_on_allocation_boundary(gpu, dtype):
    """Regression: the reduce kernel's clamped loads touch entry `start` of an EMPTY run; for empty runs
    at the very end of the re-tiled arrays that is entry nnz.  R-MAT scale 18 (the sparsest rows and
    columns are the last ones, nnz * 8 B = exactly 32 MiB) faulted there before the streams got their
    slack.  Here: last rows empty, last columns unused, nnz sized so the arrays end on a 2 MiB boundary.
    The empty trailing bins (first entry == nnz) are also what an early form of the staged scatter of
    multiply_inspect read one element past the caller's arrays for (found by tools/fuzz_spmv.py)."""
    rng = np.random.default_rng(3)
    m, n = 40000, 50000
    nnz = (1 << 21) // np.dtype(dtype).itemsize * 2           # products array = exactly 4 MiB
    rows = np.sort(rng.integers(0, m - 6000, nnz))            # the last 6000 rows (>= 1 bin) stay empty
    rowptr = np.zeros(m + 1, np.int32)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr).astype(np.int32)
    colind = rng.integers(0, n - 25000, nnz).astype(np.int32)  # the last slice(s) stay empty
    values = (rng.random(nnz) + 0.5).astype(dtype)
    x = (rng.random(n) + 0.5).astype(dtype)
    a = G.csr_on_device(values, rowptr, colind, (m, n), nnz)
    xd = G.dev(x)
    y = torch.full((m,), float("nan"), dtype=xd.dtype, device="cuda")
    info = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED)
    assert info.state_.info()["alg"] == _capi.SPMV_SLICED and info.state_.info()["n_slices"] >= 3
    for _ in range(3):
        sp.multiply(info, a, xd, y)
    check(values, rowptr, colind, (m, n), x, G.host(y), what="sliced, empty trailing runs")


def _enc8_exceptions(rowptr, colind, W, H, blk):
    """Entries the one-byte row codes cannot reach, per wave-bin (the rule of spmv_sliced.hip, restated): runs sorted by
    row, cut into blocks of `blk`; inside a block the decoder follows D_j = min(r_j, D_{j-1} + 255) and entry j is an
    exception iff r_j - D_{j-1} >= 255."""
    m = len(rowptr) - 1
    rows = np.repeat(np.arange(m), np.diff(rowptr))
    out = {}
    for b in range((m + H - 1) // H):
        sel = (rows >= b * H) & (rows < (b + 1) * H)
        r, c = rows[sel] - b * H, colind[sel]
        n_exc = 0
        for s in np.unique(c // W):
            rr = r[c // W == s]                   # storage order = row order: already sorted
            for k0 in range(0, len(rr), blk):
                d = rr[k0]
                for rj in rr[k0 + 1:k0 + blk]:
                    if rj - d >= 255:
                        n_exc += 1
                        d += 255
                    else:
                        d = rj
        out[b] = n_exc
    return out


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("enc,per_row", [("0", 2), ("2", 2), ("2", 1), ("1", 1), ("fail", 2)])
def test_spmv_sliced_row_encodings(gpu, monkeypatch, dtype, enc, per_row):
    """Round 3: the reduce streams ONE byte of row code per entry (runs sorted by row, block bases, exception lists)
    instead of 16-bit rows.  Sparse tiles on purpose: 2 entries per row leave ~50 rows between the entries of a run, so
    some entries lie >= 255 rows beyond the decoder and go through the per-bin exception lists; with 1 entry per row the
    lists overflow and the build falls back to 16-bit rows by itself.  Every variant against the oracle; forced
    encodings through SPBLAS_GFX950_PB_ENC8 (0 = 16-bit rows, 2 = one-byte codes whatever the tile density, 1 = the
    density rule decides), the fallback also through SPBLAS_GFX950_PB_ENC8_FAIL."""
    W, H = 64, 4000
    monkeypatch.setenv("SPBLAS_GFX950_SLICE_COLS", str(W))
    monkeypatch.setenv("SPBLAS_GFX950_SLICE_ROWS", str(H))
    monkeypatch.setenv("SPBLAS_GFX950_PB_ENC8", "2" if enc == "fail" else enc)
    if enc == "fail":
        monkeypatch.setenv("SPBLAS_GFX950_PB_ENC8_FAIL", "1")
    rng = np.random.default_rng(23)
    m, n = 8000, 6400
    lens = np.full(m, per_row)
    lens[rng.random(m) < 0.1] = 0
    lens[4321] = 300                                  # a row that repeats inside runs (duplicate flags)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(rowptr[-1])
    colind = rng.integers(0, n, nnz).astype(np.int32)
    values = (rng.random(nnz) - 0.5).astype(dtype)
    x = (rng.random(n) - 0.5).astype(dtype)
    exc = _enc8_exceptions(rowptr, colind, W, H, 32 if dtype == np.float32 else 16)
    if per_row == 2:
        assert 0 < max(exc.values()) <= 128, exc      # the exception lists are used and suffice
    else:
        assert max(exc.values()) > 128, exc           # more than a list holds: the build must fall back
    a = G.csr_on_device(values, rowptr, colind, (m, n), nnz)
    xd = G.dev(x)
    y = torch.full((m,), float("nan"), dtype=xd.dtype, device="cuda")
    info = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED)
    assert info.state_.info()["alg"] == _capi.SPMV_SLICED and info.state_.info()["n_slices"] == 100
    want_u8 = enc == "2" and per_row == 2
    assert info.state_.sliced_info()["row_code_u8"] == int(want_u8)
    sp.multiply(info, sp.scaled(-1.5, a), xd, y)
    check(values, rowptr, colind, (m, n), x, G.host(y), scale=-1.5, what=f"rows enc={enc}", ref_cmp=False)
    y.fill_(1.0)                                      # alpha / beta form, values refreshed in place
    a.values().mul_(2.0)
    info.state_.update_values(a.values())
    alpha, beta = (ctypes.c_float if dtype == np.float32 else ctypes.c_double)(0.5), \
        (ctypes.c_float if dtype == np.float32 else ctypes.c_double)(2.0)
    hd = sp.api._Handle.current(xd.device)
    sp.api.check(_capi.lib().spblas_gfx950_spmv(hd.h, info.state_.plan, _capi.OP_N, m, n, nnz, ctypes.byref(alpha),
                                                sp.api._ptr(a.rowptr()), sp.api._ptr(a.colind()), sp.api._ptr(a.values()),
                                                sp.api._ptr(xd), ctypes.byref(beta), sp.api._ptr(y), _capi.I32,
                                                _capi.F32 if dtype == np.float32 else _capi.F64), "spmv")
    y_ref = oracle.spmv((m, n), rowptr, colind, values * dtype(2), x).astype(np.float64) * 0.5 + 2.0
    absrow = oracle.spmv_absrow(rowptr, colind, values * dtype(2), x) * 0.5 + 2.0
    util.assert_parity(G.host(y), y_ref.astype(dtype), absrow, dtype, row_len=np.diff(rowptr), what=f"alpha/beta enc={enc}")


def test_spmv_plan_introspection_and_long_rows(gpu):
    lens = np.full(500, 3, np.int64)
    lens[100] = 10000
    rowptr = np.concatenate([[0], np.cumsum(lens)])
    nnz = int(rowptr[-1])
    rng = np.random.default_rng(1)
    colind = rng.integers(0, 777, nnz).astype(np.int32)
    values = rng.random(nnz).astype(np.float32)
    a = G.csr_on_device(values, rowptr, colind, (500, 777), nnz)
    x = G.dev(rng.random(777).astype(np.float32))
    y = torch.zeros(500, device="cuda")
    info = sp.multiply_inspect(a, x, y)
    pi = info.state_.info()
    assert pi["alg"] == _capi.SPMV_ROWBLOCK and pi["n_long_rows"] == 1 and pi["max_row_len"] == 10000
    assert pi["n_windows"] == nnz // pi["window"] + 1 and pi["empty_rows"] == 0
    sp.multiply(info, a, x, y)
    check(values, rowptr.astype(np.int32), colind, (500, 777), G.host(x), G.host(y), what="long row")


def test_spmv_matrix_opt_caches_plan_and_values_may_change(gpu):
    values, rowptr, colind, shape, nnz = generate.generate_csr(1000, 1000, 20000, seed=9)
    a = G.csr_on_device(values, rowptr, colind, shape, nnz)
    a_opt = sp.matrix_opt(a)
    x = G.dev(np.random.default_rng(0).random(1000).astype(np.float32))
    y = torch.zeros(1000, device="cuda")
    sp.multiply_inspect(a_opt, x, y)
    assert a_opt._plan is not None
    sp.multiply(a_opt, x, y)  # no info: the plan comes from the matrix_opt
    check(values, rowptr, colind, shape, G.host(x), G.host(y), what="matrix_opt")
    # row-partition plans read the caller's arrays: changing values in place stays correct
    a.values().mul_(3.0)
    sp.multiply(a_opt, x, y)
    check(values * np.float32(3), rowptr, colind, shape, G.host(x), G.host(y), what="matrix_opt rescaled")


def test_spmv_values_changed_in_place(gpu):
    """The reference reads A's values on every multiply (algorithms/multiply_impl.hpp:48-52; the rocSPARSE slot
    passes the caller's pointer at execute time, vendor/rocsparse/detail/spmv_impl.hpp:72-77).  The sliced plan
    multiplies with a re-tiled copy, so: (1) AUTO must not choose it for a plain csr_view; (2) for a matrix_opt
    operand, or on explicit request, changes made in place through torch, through scale(), or by rebinding the
    view must be picked up WITHOUT an update_values call."""
    rng = np.random.default_rng(5)
    m, n, per = 260000, 1_000_000, 9          # x = 4 MB, 2.3 M entries: AUTO's sliced candidate
    rowptr = (np.arange(m + 1, dtype=np.int64) * per).astype(np.int32)
    nnz = m * per
    colind = rng.integers(0, n, nnz).astype(np.int32)
    values = (rng.random(nnz) + 0.5).astype(np.float32)
    x = (rng.random(n) + 0.5).astype(np.float32)
    xd = G.dev(x)
    for mode in ("plain", "matrix_opt", "explicit", "prepared"):
        a = G.csr_on_device(values, rowptr, colind, (m, n), nnz)
        y = torch.full((m,), float("nan"), device="cuda")
        operand = sp.matrix_opt(a) if mode in ("matrix_opt", "prepared") else a
        info = sp.multiply_inspect(operand, xd, y, alg=_capi.SPMV_SLICED if mode == "explicit" else _capi.SPMV_AUTO)
        alg = info.state_.info()["alg"]
        assert alg == (_capi.SPMV_ROWBLOCK if mode == "plain" else _capi.SPMV_SLICED), mode
        run = sp.prepared_multiply(info, operand, xd, y) if mode == "prepared" else (
            lambda: sp.multiply(info, operand, xd, y))
        run()
        check(values, rowptr, colind, (m, n), x, G.host(y), what=f"{mode}: as inspected", ref_cmp=False)
        a.values().mul_(2.0)                                   # in place through torch
        run()
        check(values * np.float32(2), rowptr, colind, (m, n), x, G.host(y), what=f"{mode}: mul_", ref_cmp=False)
        sp.scale(0.25, a)                                      # in place through the backend's own scale()
        run()
        check(values * np.float32(0.5), rowptr, colind, (m, n), x, G.host(y), what=f"{mode}: scale()", ref_cmp=False)
        if mode != "prepared":                                 # a prepared call is bound to its tensors
            fresh = G.dev(values * np.float32(3))
            a.update(fresh, a.rowptr(), a.colind())            # csr_view.hpp:36-49: rebind the value array
            run()
            check(values * np.float32(3), rowptr, colind, (m, n), x, G.host(y), what=f"{mode}: rebound",
                  ref_cmp=False)


def test_spmv_snapshot_sees_values_rewritten_by_the_librarys_own_kernels(gpu):
    """the sliced plan's value snapshot must also notice writes made through this library's own raw
    kernels -- multiply_fill refilling C's values (an AMG re-setup: inspect matrix_opt(C) once, refill C before every
    multiply), add() and transpose() writing into an inspected matrix's arrays -- none of which moves torch's version
    counter.  Reference behaviour: the values are read on every multiply (multiply_impl.hpp:48-52)."""
    from oracle import oracle
    rng = np.random.default_rng(11)
    m = k = 300000
    per = 3
    def rand_csr(seed_vals):
        rp = (np.arange(m + 1, dtype=np.int64) * per).astype(np.int32)
        ci = rng.integers(0, k, m * per).astype(np.int32)
        return rp, ci, (rng.random(m * per) + 0.5).astype(np.float32)
    arp, aci, av = rand_csr(0)
    brp, bci, bv = rand_csr(1)
    a = G.csr_on_device(av, arp, aci, (m, k), m * per)
    b = G.csr_on_device(bv, brp, bci, (k, k), k * per)
    c_rp = torch.zeros(m + 1, dtype=torch.int32, device="cuda")
    c = sp.csr_view(None, c_rp, None, (m, k), 0)
    info_c = sp.multiply_compute(a, b, c)
    cn = info_c.result_nnz()
    c.update(torch.zeros(cn, device="cuda"), c_rp, torch.zeros(cn, dtype=torch.int32, device="cuda"), (m, k), cn)
    sp.multiply_fill(info_c, a, b, c)
    x = (rng.random(k) + 0.5).astype(np.float32)
    xd = G.dev(x)
    y = torch.full((m,), float("nan"), device="cuda")
    info = sp.multiply_inspect(sp.matrix_opt(c), xd, y, alg=_capi.SPMV_SLICED)
    assert info.state_.info()["alg"] == _capi.SPMV_SLICED
    sp.multiply(info, sp.matrix_opt(c), xd, y)
    crp, cci = G.host(c.rowptr()), G.host(c.colind())
    check(G.host(c.values()), crp, cci, (m, k), x, G.host(y), what="C as inspected", ref_cmp=False)
    # refill C in place with new A values (same structure, same arrays, torch's counter does not move)
    a.values().mul_(3.0)
    ver = c.values()._version
    sp.multiply_fill(info_c, a, b, c)
    assert c.values()._version == ver  # the write is invisible to torch: only the module's write epoch sees it
    sp.multiply(info, sp.matrix_opt(c), xd, y)
    check(G.host(c.values()), crp, cci, (m, k), x, G.host(y), what="C refilled by multiply_fill", ref_cmp=False)
    y1 = G.host(y).copy()
    # transpose() writing into an inspected matrix's value array
    t_src = G.csr_on_device(*[arr for arr in (av * np.float32(0.5), arp, aci)], (m, k), m * per)
    t = sp.csr_view(torch.zeros(m * per, device="cuda"), torch.zeros(k + 1, dtype=torch.int32, device="cuda"),
                    torch.zeros(m * per, dtype=torch.int32, device="cuda"), (k, m), m * per)
    sp.transpose(t_src, t)
    yt = torch.full((k,), float("nan"), device="cuda")
    xm = G.dev((rng.random(m) + 0.5).astype(np.float32))
    info_t = sp.multiply_inspect(sp.matrix_opt(t), xm, yt, alg=_capi.SPMV_SLICED)
    sp.multiply(info_t, sp.matrix_opt(t), xm, yt)
    t_src.values().mul_(4.0)
    sp.transpose(t_src, t)       # same structure, new values, same output arrays
    sp.multiply(info_t, sp.matrix_opt(t), xm, yt)
    check(G.host(t.values()), G.host(t.rowptr()), G.host(t.colind()), (k, m), G.host(xm), G.host(yt),
          what="matrix rewritten by transpose()", ref_cmp=False)
    assert not np.array_equal(y1, np.zeros_like(y1))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spmv_sliced_compacts_empty_rows(gpu, monkeypatch, dtype):
    """Graph-like matrix: 70 % of the rows empty (in stretches, at both ends and scattered), a few hub rows, hot
    columns.  The SLICED plan tiles the non-empty rows only (variable-height bins over compact row numbers); y of an
    empty row must come out as beta * y, alpha / beta forms and row-range reduces must agree with the oracle."""
    rng = np.random.default_rng(41)
    m, n = 300000, 900000
    lens = np.where(rng.random(m) < 0.3, rng.integers(1, 40, m), 0)
    lens[:5000] = 0
    lens[-777:] = 0
    lens[120000:150000] = 0
    hubs = rng.choice(np.flatnonzero(lens), 6, replace=False)
    lens[hubs] = [30000, 20000, 2500, 1800, 1200, 70000]
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(rowptr[-1])
    colind = np.where(rng.random(nnz) < 0.4, rng.integers(0, 2000, nnz), rng.integers(0, n, nnz)).astype(np.int32)
    values = (rng.random(nnz) - 0.5).astype(dtype)
    x = (rng.random(n) - 0.5).astype(dtype)
    a = G.csr_on_device(values, rowptr, colind, (m, n), nnz)
    xd = G.dev(x)
    y = torch.full((m,), float("nan"), dtype=xd.dtype, device="cuda")
    # default: empty rows out AND every row longer than L entries cut into pieces of L entries (no hub rows left);
    # L = 32 for fp32, 2048 for fp64 (csrc/spmv_sliced.hip)
    info = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED)
    si = info.state_.sliced_info()
    assert si["variable_bins"] == 1 and si["hub_rows"] == 0
    L = 32 if dtype == np.float32 else 2048
    assert si["tiled_rows"] == int((lens > 0).sum()) + int(sum(-(-int(v) // L) - 1 for v in lens if v > L))
    sp.multiply(info, a, xd, y)
    yh = G.host(y)
    assert not np.isnan(yh).any() and not yh[lens == 0].any()
    check(values, rowptr, colind, (m, n), x, yh, what="compacted + split rows", ref_cmp=False)
    y.fill_(1.0)   # beta path of the split rows and of the empty rows, through the C ABI
    alpha_c, beta_c = (ctypes.c_float if dtype == np.float32 else ctypes.c_double)(0.5), \
        (ctypes.c_float if dtype == np.float32 else ctypes.c_double)(2.0)
    api_h = sp.api._Handle.current(y.device)
    assert _capi.lib().spblas_gfx950_spmv(api_h.h, info.state_.plan, _capi.OP_N, m, n, nnz, ctypes.byref(alpha_c),
                                         ctypes.c_void_p(a.rowptr().data_ptr()), ctypes.c_void_p(a.colind().data_ptr()),
                                         ctypes.c_void_p(a.values().data_ptr()), ctypes.c_void_p(xd.data_ptr()),
                                         ctypes.byref(beta_c), ctypes.c_void_p(y.data_ptr()), _capi.I32,
                                         _capi.F32 if dtype == np.float32 else _capi.F64) == 0
    want = 0.5 * np.asarray(oracle.spmv((m, n), rowptr, colind, values.astype(np.float64), x.astype(np.float64))) + 2.0
    absrow = 0.5 * oracle.spmv_absrow(rowptr, colind, values.astype(np.float64), x.astype(np.float64)) + 2.0
    util.assert_parity(G.host(y), want, absrow, dtype, row_len=lens, what="compacted + split rows, alpha and beta")
    expand, reduce_rows = info.state_.bind_stages(xd, y.data_ptr(), xd.dtype)
    expand()
    with pytest.raises(RuntimeError):      # pieces of one row lie in different bins: no partial row ranges
        reduce_rows(0, m // 2)
    del info
    # rows kept whole (hub rows go through their own kernel): row ranges work
    monkeypatch.setenv("SPBLAS_GFX950_PB_SPLIT_LEN", "0")
    info = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED)
    si = info.state_.sliced_info()
    assert si["variable_bins"] == 1 and si["hub_rows"] >= 2 and si["tiled_rows"] == int((lens > 0).sum())
    assert si["n_bins"] < cdiv(m, info.state_.info()["rows_per_bin"]) + 2100     # far fewer than a bin per H rows of y
    sp.multiply(info, a, xd, y)
    yh = G.host(y)
    assert not np.isnan(yh).any() and not yh[lens == 0].any()
    check(values, rowptr, colind, (m, n), x, yh, what="compacted rows", ref_cmp=False)
    sp.multiply(info, sp.scaled(-1.5, a), xd, y)
    check(values, rowptr, colind, (m, n), x, G.host(y), scale=-1.5, what="compacted rows, alpha", ref_cmp=False)
    # two-stage form with beta: y2 = 0.5 * A x + 2 * y2, the rows reduced in three ranges cut at arbitrary rows
    y2 = torch.full((m,), 3.0, dtype=xd.dtype, device="cuda")
    expand, reduce_rows = info.state_.bind_stages(xd, y2.data_ptr(), xd.dtype, alpha=0.5, beta=2.0)
    expand()
    for lo, hi in ((0, 100001), (100001, 234567), (234567, m)):
        reduce_rows(lo, hi)
    want = 0.5 * np.asarray(oracle.spmv((m, n), rowptr, colind, values.astype(np.float64), x.astype(np.float64))) + 6.0
    got = G.host(y2).astype(np.float64)
    assert np.array_equal(got[lens == 0], np.full(int((lens == 0).sum()), 6.0))
    absrow = 0.5 * oracle.spmv_absrow(rowptr, colind, values.astype(np.float64), x.astype(np.float64)) + 6.0
    util.assert_parity(got.astype(dtype), want, absrow, dtype, row_len=lens, what="compacted rows, beta + row ranges")


def test_prepared_calls_keep_their_stream_and_plans_die_in_stream_order(gpu):
    """prepared_multiply / bind_stages launch on the stream that was current when they were made, although every other
    API call re-binds the handle to the stream current at ITS call; a plan destroyed while the handle sits on another
    stream frees its workspaces behind the last launch that used them (spblas_gfx950_plan_destroy)."""
    rng = np.random.default_rng(17)
    m, n, per = 300000, 1_200_000, 8
    rowptr = (np.arange(m + 1, dtype=np.int64) * per).astype(np.int32)
    nnz = m * per
    colind = rng.integers(0, n, nnz).astype(np.int32)
    values = (rng.random(nnz) + 0.5).astype(np.float32)
    x = (rng.random(n) + 0.5).astype(np.float32)
    a = G.csr_on_device(values, rowptr, colind, (m, n), nnz)
    xd = G.dev(x)
    y = torch.full((m,), float("nan"), device="cuda")
    y2 = torch.full((m,), float("nan"), device="cuda")
    small = G.csr_on_device(values[:80], rowptr[:11], (colind[:80] % 50).astype(np.int32), (10, 50), 80)
    xs, ys = G.dev(x[:50]), torch.empty(10, device="cuda")
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        info = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED)
        run = sp.prepared_multiply(info, a, xd, y)
        expand, reduce_rows = info.state_.bind_stages(xd, y2.data_ptr(), torch.float32)
    for _ in range(3):
        sp.multiply(small, xs, ys)       # the handle now sits on the default stream
        run()                            # ... and goes back to `side` for the prepared call
        sp.multiply(small, xs, ys)
        expand()
        reduce_rows(0, m)
    sp.multiply(small, xs, ys)
    del run, expand, reduce_rows, info    # plan destroyed with the handle on the default stream, work pending on `side`
    side.synchronize()
    torch.cuda.synchronize()
    check(values, rowptr, colind, (m, n), x, G.host(y), what="prepared on a side stream", ref_cmp=False)
    check(values, rowptr, colind, (m, n), x, G.host(y2), what="bound stages on a side stream", ref_cmp=False)


def test_spmv_csc_and_transposed_operand(gpu):
    # y = A x with A given as csc_view, and y = A^T x via transposed(csr)
    # (vendor/rocsparse/detail/get_transpose.hpp:19-29; test/gtest/spmv_test.cpp:110-208)
    values, rowptr, colind, shape, nnz = generate.generate_csr(300, 200, 4000, seed=4)
    x = np.random.default_rng(3).random(300).astype(np.float32)
    a = G.csr_on_device(values, rowptr, colind, shape, nnz)
    y = torch.full((200,), float("nan"), device="cuda")
    sp.multiply(sp.transposed(a), G.dev(x), y)  # A^T (200x300) as csc_view
    y_ref = oracle.spmv_csc((200, 300), rowptr, colind, values, x)
    import scipy.sparse as sps
    absrow = sps.csr_matrix((np.abs(values).astype(np.float64), colind, rowptr), shape=shape).T @ np.abs(x)
    util.assert_parity(G.host(y), y_ref, absrow, np.float32, row_len=np.full(200, 64), what="csc spmv")
    sp.multiply(sp.scaled(2.0, sp.transposed(a)), G.dev(x), y)
    util.assert_parity(G.host(y), 2 * y_ref, 2 * absrow, np.float32, row_len=np.full(200, 64), what="csc scaled")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("offsets", [np.int32, np.int64])
@pytest.mark.parametrize("shape_kind", ["ragged", "hot_column", "many_empty_rows", "wide"])
def test_spmv_transposed_without_a_plan_two_pass_form(gpu, monkeypatch, dtype, offsets, shape_kind):
    """Round 6: un-inspected y = alpha A^T x + beta y on large matrices goes through a workspace instead of one global float
    atomic per entry (csrc/spmv.hip: t2_* kernels; cfg2-sized operand 4.75 -> see bench.py --workload csc_spmv): entries per
    column slice, cursors + work items (a slice holding far more than its share is cut into segments), products with 16-bit
    local columns scattered tile by tile -- the row of an entry by binary search in the tile's staged row offsets --, one LDS
    accumulation per slice.  Forced here for small shapes (SPBLAS_GFX950_SPMV_T2=1): ragged rows, a hot column that forces the
    segment path, a stretch of tens of thousands of empty rows inside one tile (row offsets not staged), a wide matrix with
    several slices; alpha / beta through the C ABI with NaN in the old y for beta = 0; compared with the oracle and with the
    scatter kernel (SPBLAS_GFX950_SPMV_T2=0)."""
    rng = np.random.default_rng(83)
    if shape_kind == "wide":
        m, n = 3000, 300_000
        lens = rng.integers(0, 60, m)
    elif shape_kind == "many_empty_rows":
        m, n = 120_000, 9000
        lens = np.zeros(m, np.int64)
        lens[rng.choice(m, 3000, replace=False)] = rng.integers(1, 40, 3000)
    else:
        m, n = 20_000, 70_000
        lens = rng.integers(0, 30, m)
        lens[777] = 5000
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(offsets)
    nnz = int(rowptr[-1])
    colind = rng.integers(0, n, nnz).astype(np.int32)
    if shape_kind == "hot_column":
        colind[rng.random(nnz) < 0.5] = 12345           # half of all entries in one column: its slice is cut into segments
    values = (rng.random(nnz) - 0.5).astype(dtype)
    x = (rng.random(m) - 0.5).astype(dtype)
    y0 = (rng.random(n) - 0.5).astype(dtype)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    t = lambda a_: torch.from_numpy(np.ascontiguousarray(a_)).cuda()
    dv, drp, dci, dx = t(values), t(rowptr), t(colind), t(x)
    rp32 = rowptr.astype(np.int32)
    ref = oracle.spmv_csc((n, m), rp32, colind, values, x).astype(np.float64)
    ab = oracle.spmv_csc((n, m), rp32, colind, np.abs(values), np.abs(x)).astype(np.float64)
    cnt = np.bincount(colind, minlength=n)
    hd = sp.api._Handle.current(dx.device)
    ct = ctypes.c_float if dtype == np.float32 else ctypes.c_double
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SPBLAS_GFX950_SPMV_T2", mode)
        for alpha, beta in ((1.0, 0.0), (-1.5, 0.25)):
            y = torch.full((n,), float("nan"), dtype=tdt, device="cuda") if beta == 0.0 else t(y0)
            al, be = ct(alpha), ct(beta)
            sp.api.check(_capi.lib().spblas_gfx950_spmv(hd.h, None, _capi.OP_T, m, n, nnz, ctypes.byref(al), sp.api._ptr(drp),
                                                        sp.api._ptr(dci), sp.api._ptr(dv), sp.api._ptr(dx), ctypes.byref(be),
                                                        sp.api._ptr(y), _capi.I32 if offsets == np.int32 else _capi.I64,
                                                        _capi.F32 if dtype == np.float32 else _capi.F64), "spmv")
            want = alpha * ref + (beta * y0 if beta else 0.0)
            scale = abs(alpha) * ab + (abs(beta) * np.abs(y0) if beta else 0.0)
            util.assert_parity(G.host(y), want, scale, dtype, row_len=cnt + 1, what=f"op = T, T2={mode}, {shape_kind}, beta={beta}")
            got[(mode, beta)] = G.host(y)
    # through the host layer: multiply(transposed(a), x, y)
    monkeypatch.setenv("SPBLAS_GFX950_SPMV_T2", "1")
    a = sp.csr_view(dv, drp, dci, (m, n), nnz)
    y = torch.full((n,), float("nan"), dtype=tdt, device="cuda")
    sp.multiply(sp.scaled(2.0, sp.transposed(a)), dx, y)
    util.assert_parity(G.host(y), 2.0 * ref, 2.0 * ab, dtype, row_len=cnt + 1, what="multiply(transposed(a), x, y), two-pass form")


def test_spmv_degenerate_shapes(gpu):
    # all rows empty; zero rows; single entry
    z = sp.csr_view(torch.zeros(0, device="cuda"), torch.zeros(6, dtype=torch.int32, device="cuda"),
                    torch.zeros(0, dtype=torch.int32, device="cuda"), (5, 7), 0)
    y = torch.full((5,), 3.0, device="cuda")
    sp.multiply(z, torch.ones(7, device="cuda"), y)
    assert np.array_equal(G.host(y), np.zeros(5, np.float32))
    info = sp.multiply_inspect(z, torch.ones(7, device="cuda"), y)
    sp.multiply(info, z, torch.ones(7, device="cuda"), y)
    assert np.array_equal(G.host(y), np.zeros(5, np.float32))
    one = sp.csr_view(torch.tensor([2.5], device="cuda"), torch.tensor([0, 0, 1], dtype=torch.int32, device="cuda"),
                      torch.tensor([1], dtype=torch.int32, device="cuda"), (2, 3), 1)
    y2 = torch.zeros(2, device="cuda")
    sp.multiply(one, torch.tensor([1., 4., 9.], device="cuda"), y2)
    assert np.array_equal(G.host(y2), np.array([0, 10], np.float32))


def test_spmv_errors_on_device(gpu):
    values, rowptr, colind, shape, nnz = generate.generate_csr(40, 40, 1000)
    a = G.csr_on_device(values, rowptr, colind, shape, nnz)
    with pytest.raises(ValueError):  # multiply_impl.hpp:37-41
        sp.multiply(a, torch.ones(39, device="cuda"), torch.zeros(40, device="cuda"))
    with pytest.raises(RuntimeError, match="conjugated"):  # spmv_impl.hpp:29-33
        sp.multiply(sp.conjugated(a), torch.ones(40, device="cuda"), torch.zeros(40, device="cuda"))
    # a plan built for another matrix is refused by the library
    other = G.csr_on_device(*generate.generate_csr(40, 40, 900, seed=2))
    info = sp.multiply_inspect(other, torch.ones(40, device="cuda"), torch.zeros(40, device="cuda"))
    sp.multiply(info, a, torch.ones(40, device="cuda"), torch.zeros(40, device="cuda"))  # key differs -> plan-free


@pytest.mark.parametrize("poisson", [False, True])
def test_spmv_full_size_properties_cfg2(gpu, poisson):
    """BASELINE cfg2 (10M x 10M, avg 10 nnz/row, fp32) through size-independent properties:
    (1) a seeded sample of rows against the oracle, (2) linearity A(ax+by) = aAx + bAy,
    (3) plan and plan-free kernels agree, (4) sum(y) == column-weighted checksum in fp64."""
    m = n = 10_000_000
    values, rowptr, colind, shape, nnz = generate.uniform_csr_device(m, n, 10, poisson=poisson, seed=0)
    a = sp.csr_view(values, rowptr, colind, shape, nnz)
    g = torch.Generator(device="cuda").manual_seed(1)
    x1 = torch.rand(n, device="cuda", generator=g)
    x2 = torch.rand(n, device="cuda", generator=g)
    y1, y2, y3, y1v = (torch.empty(m, device="cuda") for _ in range(4))
    # AUTO on a matrix_opt operand: x (40 MB) >> L2, random columns -> the LDS-sliced plan.  (A plain csr_view
    # keeps AUTO on plans that read the caller's values: test_spmv_values_changed_in_place.)
    info = sp.multiply_inspect(sp.matrix_opt(a), x1, y1)
    assert info.state_.info()["alg"] == _capi.SPMV_SLICED
    sp.multiply(info, a, x1, y1)
    sp.multiply(info, a, x2, y2)
    sp.multiply(info, a, 0.5 * x1 - 2.0 * x2, y3)
    info_rb = sp.multiply_inspect(a, x1, y1v, alg=_capi.SPMV_ROWBLOCK)
    sp.multiply(info_rb, a, x1, y1v)
    # (2) linearity, norm-wise: |y3 - (.5y1 - 2y2)| <= 4e-6 * (|.5 y1| + |2 y2|)
    lin = (y3 - (0.5 * y1 - 2.0 * y2)).abs()
    assert bool((lin <= 4e-6 * (0.5 * y1.abs() + 2.0 * y2.abs()) + 1e-30).all())
    # (3) the two kernels agree to rounding
    assert bool(((y1 - y1v).abs() <= 2e-6 * y1.abs() + 1e-30).all())
    # (4) checksum: sum_i y_i == sum_p v_p x_{c_p} in float64
    lhs = y1.double().sum().item()
    rhs = (values.double() * x1[colind.long()].double()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * abs(rhs)
    # (1) oracle on the first and last 2000 rows and 2000 random rows
    rp = rowptr.cpu().numpy().astype(np.int64)
    rows = np.unique(np.concatenate([np.arange(2000), np.arange(m - 2000, m),
                                     np.random.default_rng(0).integers(0, m, 2000)]))
    x_h = x1.cpu().numpy()
    y_h = y1.cpu().numpy()
    for chunk in (rows[:2000], rows[2000:]):
        lens = rp[chunk + 1] - rp[chunk]
        sub_rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        idx = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in chunk]) if lens.sum() else np.zeros(0, np.int64)
        idx_t = torch.from_numpy(idx).cuda()
        sub_c = colind[idx_t].cpu().numpy()
        sub_v = values[idx_t].cpu().numpy()
        y_ref = oracle.spmv((len(chunk), n), sub_rp, sub_c, sub_v, x_h)
        absrow = oracle.spmv_absrow(sub_rp, sub_c, sub_v, x_h)
        util.assert_parity(y_h[chunk], y_ref, absrow, np.float32, what="cfg2 sampled rows")


@pytest.mark.parametrize("poisson", [False, True])
def test_spmv_full_size_plain_csr_view_reads_the_values_of_the_call_cfg2(gpu, poisson):
    """Round 6 (review item): north_star's literal call shape at BASELINE cfg2's FULL size -- multiply_inspect on a plain
    csr_view (no matrix_opt), AUTO -> the value-free tiles -- with the values REWRITTEN IN PLACE after inspect, twice, through a
    raw view that moves no version counter.  Every multiply must read the caller's array of that call
    (multiply_impl.hpp:48-52; vendor/rocsparse/detail/spmv_impl.hpp:72-77).  Checked by (1) 6 000 sampled rows (first / last
    2 000 + random) against the oracle, (2) the fp64 checksum of all of y against sum(values * x[colind]) on the device,
    (3) agreement with the row-block kernel on the caller's arrays, all three after EACH rewrite, and (4) that the plan holds
    no values (value_free, no update call made anywhere)."""
    m = n = 10_000_000
    values, rowptr, colind, shape, nnz = generate.uniform_csr_device(m, n, 10, poisson=poisson, seed=0)
    a = sp.csr_view(values, rowptr, colind, shape, nnz)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(n, device="cuda", generator=g)
    y, yv = torch.empty(m, device="cuda"), torch.empty(m, device="cuda")
    info = sp.multiply_inspect(a, x, y)
    si = info.state_.sliced_info()
    assert info.state_.info()["alg"] == _capi.SPMV_SLICED and si["value_free"] == 1 and si["refresh_each_call"] == 1, si
    assert si["auto_trial"] == 0                                   # decided by rule: same plan on every box
    info_rb = sp.multiply_inspect(a, x, yv, alg=_capi.SPMV_ROWBLOCK)
    rp = rowptr.cpu().numpy().astype(np.int64)
    rows = np.unique(np.concatenate([np.arange(2000), np.arange(m - 2000, m), np.random.default_rng(0).integers(0, m, 2000)]))
    lens = rp[rows + 1] - rp[rows]
    sub_rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx_t = torch.from_numpy(np.concatenate([np.arange(rp[r], rp[r + 1]) for r in rows])).cuda()
    sub_c = colind[idx_t].cpu().numpy()
    x_h = x.cpu().numpy()
    rows_t = torch.from_numpy(rows).cuda()
    raw = torch.as_strided(values, values.shape, values.stride())
    for step, (mul, add) in enumerate(((1.0, 0.0), (-0.5, 0.125), (3.0, -1.0))):
        raw.data.mul_(mul).add_(add)                                # in place, after inspect; no update call anywhere
        y.fill_(float("nan"))
        sp.multiply(info, a, x, y)
        sp.multiply(info_rb, a, x, yv)
        sub_v = values[idx_t].cpu().numpy()
        y_ref = oracle.spmv((len(rows), n), sub_rp, sub_c, sub_v, x_h)
        absrow = oracle.spmv_absrow(sub_rp, sub_c, sub_v, x_h)
        util.assert_parity(y[rows_t].cpu().numpy(), y_ref, absrow, np.float32, row_len=lens,
                           what=f"cfg2 plain csr_view, sampled rows, rewrite {step}")
        prod = values.double() * x[colind.long()].double()
        lhs, rhs, scale = y.double().sum().item(), prod.sum().item(), prod.abs().sum().item()
        assert abs(lhs - rhs) <= 1e-6 * scale, (step, lhs, rhs)
        absy = torch.zeros(m, dtype=torch.float64, device="cuda").index_add_(
            0, torch.repeat_interleave(torch.arange(m, device="cuda"), (rowptr[1:] - rowptr[:-1]).long()), prod.abs())
        assert bool(((y - yv).abs().double() <= 2e-6 * absy + 1e-30).all()), step
        del prod, absy


@pytest.mark.parametrize("offsets", [np.int32, np.int64])
def test_spmv_and_spmm_take_64_bit_column_indices(gpu, offsets):
    """Round 6 (slot type surface): the rocSPARSE slot admits 64-bit indices (vendor/rocsparse/types.hpp:16-24); a csr_view /
    csc_view whose index array is int64 is narrowed once on the device (spblas_gfx950_narrow_indices, range-checked) and then
    takes every SpMV / SpMM path -- plan-free, inspected (the plan must be FOUND again by the multiplies: one narrowed array
    per index tensor), matrix_opt, op = T.  An index outside the matrix is an error, not a wrapped 32-bit value."""
    rng = np.random.default_rng(71)
    m, n, per = 30000, 50000, 12
    values, rowptr, colind, shape, nnz = generate.generate_csr(m, n, m * per, seed=5)
    x = (rng.random(n) - 0.5).astype(np.float32)
    t = lambda a_: torch.from_numpy(np.ascontiguousarray(a_)).cuda()
    ci64 = t(colind.astype(np.int64))
    a = sp.csr_view(t(values), t(rowptr.astype(offsets)), ci64, shape, nnz)
    xd = t(x)
    y = torch.full((m,), float("nan"), device="cuda")
    sp.multiply(a, xd, y)
    check(values, rowptr, colind, shape, x, G.host(y), what="int64 columns, plan-free", ref_cmp=False)
    for alg in (_capi.SPMV_ROWBLOCK, _capi.SPMV_SLICED):
        info = sp.multiply_inspect(a, xd, y, alg=alg)
        y.fill_(float("nan"))
        sp.multiply(info, sp.scaled(-2.0, a), xd, y)
        check(values, rowptr, colind, shape, x, G.host(y), scale=-2.0, what=f"int64 columns, inspected alg {alg}", ref_cmp=False)
        assert sp.api._find_plan(info, a, sp.api._int32_columns(a, "t")) is info.state_   # the multiply used THE plan
    a_opt = sp.matrix_opt(a)
    info = sp.multiply_inspect(a_opt, xd, y)
    y.fill_(float("nan"))
    sp.multiply(a_opt, xd, y)
    check(values, rowptr, colind, shape, x, G.host(y), what="int64 columns, matrix_opt", ref_cmp=False)
    # SpMM
    B = (rng.random((n, 24)) - 0.5).astype(np.float32)
    C = torch.full((m, 24), float("nan"), device="cuda")
    info = sp.multiply_inspect(a, t(B), C)
    sp.multiply(info, a, t(B), C)
    C_ref = oracle.spmm(shape, rowptr, colind, values, B)
    C_abs = oracle.spmm(shape, rowptr, colind, np.abs(values), np.abs(B))
    util.assert_parity(G.host(C), C_ref, C_abs, np.float32, row_len=np.diff(rowptr), what="int64 columns, SpMM")
    # the same arrays as a csc_view (64-bit row indices): y = A^T x
    a_csc = sp.csc_view(t(values), t(rowptr.astype(offsets)), ci64, (n, m), nnz)
    xt = (rng.random(m) - 0.5).astype(np.float32)
    yt = torch.full((n,), float("nan"), device="cuda")
    sp.multiply(a_csc, t(xt), yt)
    ref_t = oracle.spmv_csc((n, m), rowptr, colind, values, xt)
    abs_t = oracle.spmv_csc((n, m), rowptr, colind, np.abs(values), np.abs(xt))
    util.assert_parity(G.host(yt), ref_t, abs_t, np.float32, row_len=np.full(n, 64), what="int64 row indices, csc_view")
    # an index that does not fit the matrix
    bad = colind.astype(np.int64)
    bad[12345] = (1 << 32) + 7            # would wrap to column 7
    with pytest.raises(ValueError, match="outside the matrix"):
        sp.multiply(sp.csr_view(t(values), t(rowptr.astype(offsets)), t(bad), shape, nnz), xd, y)
    bad[12345] = n
    with pytest.raises(ValueError, match="outside the matrix"):
        sp.multiply_inspect(sp.csr_view(t(values), t(rowptr.astype(offsets)), t(bad), shape, nnz), xd, y)


def test_spmv_two_stage_and_overlapped_sharding_single_rank(gpu):
    """expand + reduce_rows (the stage API behind the overlapped multi-GPU step) equals one spmv;
    OverlappedShardedSpMV at world size 1 exercises the stripe-aligned plan and the y-base arithmetic."""
    from spblas_reference_amd import sharded
    m, n, chunks = 12000, 5000, 4
    L = m // chunks
    values, rowptr, colind, shape, nnz = generate.generate_csr(m, n, 150000, seed=21)
    a = G.csr_on_device(values, rowptr, colind, shape, nnz)
    x_h = np.random.default_rng(2).random(n).astype(np.float32)
    x = G.dev(x_h)
    ranges = sharded.striped_row_ranges(m, 1, chunks)
    op = sharded.OverlappedShardedSpMV(a, ranges, alg=_capi.SPMV_SLICED)
    pi = op.info.state_.info()
    assert pi["alg"] == _capi.SPMV_SLICED and pi["bin_aligned"] == 1 and L % pi["rows_per_bin"] == 0
    y = G.host(op.step(x))
    check(values, rowptr, colind, shape, x_h, y, what="overlapped single rank")
    # stage API directly, rows finished out of order
    y2 = torch.full((m,), float("nan"), device="cuda")
    expand, reduce_rows = op.info.state_.bind_stages(x, y2.data_ptr(), torch.float32, alpha=2.0)
    expand()
    for c in (2, 0, 3, 1):
        reduce_rows(c * L, (c + 1) * L)
    check(values, rowptr, colind, shape, x_h, G.host(y2), scale=2.0, what="two-stage")
    # a ROWBLOCK plan refuses the stage API
    info_rb = sp.multiply_inspect(a, x, y2, alg=_capi.SPMV_ROWBLOCK)
    with pytest.raises(sp.BackendError):
        info_rb.state_.bind_stages(x, y2.data_ptr(), torch.float32)[0]()


@pytest.mark.parametrize("split", ["pieces", "hub_rows"])
def test_spmv_sliced_with_a_few_dense_rows(gpu, monkeypatch, split):
    """A uniform matrix with a handful of very long rows: AUTO must still pick the LDS-sliced plan.  By default the long
    rows are cut into pieces that the tiles treat as rows ("pieces"; such a plan reduces all rows at once); with
    SPBLAS_GFX950_PB_SPLIT_LEN=0 they stay out of the tiles and are added by pb_hub_rows_kernel ("hub_rows"), and the
    two-stage entry points must cover them as well."""
    if split == "hub_rows":
        monkeypatch.setenv("SPBLAS_GFX950_PB_SPLIT_LEN", "0")
    rng = np.random.default_rng(21)
    m, n = 300000, 1500000
    lens = np.full(m, 8, np.int64)
    hubs = [5, 149999, m - 1]
    for r, ln in zip(hubs, (120000, 30000, 5000)):
        lens[r] = ln
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(rowptr[-1])
    colind = rng.integers(0, n, nnz).astype(np.int32)
    values = (rng.random(nnz) + 0.5).astype(np.float32)
    x = (rng.random(n) + 0.5).astype(np.float32)
    a = G.csr_on_device(values, rowptr, colind, (m, n), nnz)
    xd = G.dev(x)
    y = torch.full((m,), float("nan"), device="cuda")
    info = sp.multiply_inspect(sp.matrix_opt(a), xd, y)        # AUTO, value copy allowed
    pi = info.state_.info()
    assert pi["alg"] == _capi.SPMV_SLICED and pi["n_long_rows"] == 3
    sp.multiply(info, sp.scaled(-1.5, a), xd, y)
    check(values, rowptr, colind, (m, n), x, G.host(y), scale=-1.5, what="sliced + hub rows", ref_cmp=False)
    si = info.state_.sliced_info()
    # ("hub_rows": the 120 000- and the 30 000-entry row exceed the 16 384-entry hub threshold, the 5 000-entry row stays)
    assert (si["hub_rows"], si["tiled_rows"] > m) == ((0, True) if split == "pieces" else (2, False))
    # beta != 0 through the C ABI path of prepared_multiply is covered elsewhere; here: two-stage execution
    y2 = torch.full((m,), float("nan"), device="cuda")
    expand, reduce_rows = info.state_.bind_stages(xd, y2.data_ptr(), torch.float32, alpha=-1.5)
    expand()
    if split == "pieces":
        reduce_rows(0, m)
    else:
        H = pi["rows_per_bin"]
        cut = (m // 2 // H) * H
        reduce_rows(cut, m)
        reduce_rows(0, cut)
    assert torch.equal(y, y2)
    # values change in place: update_values refreshes the tiles, the hub rows read the caller's array
    a.values().mul_(2.0)
    info.state_.update_values(a.values())
    sp.multiply(info, a, xd, y)
    check(values * np.float32(2.0), rowptr, colind, (m, n), x, G.host(y), what="sliced + hub rows after update",
          ref_cmp=False)


@pytest.mark.parametrize("varbins", ["0", "1", "rule"])
def test_spmv_auto_declines_sliced_for_skewed_matrices(gpu, monkeypatch, varbins):
    """Hot columns (one slice carries most entries) or a heavy block of rows (one bin group does): the
    sliced plan would leave the chip waiting for a single workgroup, so AUTO must keep the row-block
    kernel; a forced SLICED plan must still be correct.  With variable-height bins (the default for row-skewed
    matrices; SPBLAS_GFX950_PB_VARBINS=0 keeps the arithmetic bins and their reduce work list under test) the
    heavy block of rows is spread over many bins and AUTO may take either plan."""
    # the static rules ("0": no variable bins, no timed trial -- AUTO must decline), the opt-in of rounds 2 - 5 ("1": the plan
    # is built and AUTO keeps whichever of the two was faster in a timed trial) and round 6's default ("rule": the plan is
    # built and kept by a RULE -- x of 8 MB here is far below the 40 MB from which a skewed matrix keeps its tiles -- so that
    # the same matrix gets the same plan on every box)
    if varbins == "rule":
        monkeypatch.delenv("SPBLAS_GFX950_AUTO_TRIAL", raising=False)
    else:
        monkeypatch.setenv("SPBLAS_GFX950_PB_VARBINS", varbins)
        monkeypatch.setenv("SPBLAS_GFX950_AUTO_TRIAL", varbins)
    # (rows stay whole here: a plan that cuts long rows into pieces reduces all rows at once, and this test also drives
    # the two-stage row-range form; test_spmv_sliced_compacts_empty_rows covers the pieces)
    monkeypatch.setenv("SPBLAS_GFX950_PB_SPLIT_LEN", "0")
    rng = np.random.default_rng(33)
    m, n, per = 400000, 2000000, 8
    rowptr = (np.arange(m + 1, dtype=np.int64) * per).astype(np.int32)
    nnz = m * per
    values = (rng.random(nnz) + 0.5).astype(np.float32)
    x = (rng.random(n) + 0.5).astype(np.float32)
    hot = rng.random(nnz) < 0.6
    colind = np.where(hot, rng.integers(0, 3000, nnz), rng.integers(0, n, nnz)).astype(np.int32)
    for cols, what in ((colind, "hot columns"),):
        a = G.csr_on_device(values, rowptr, cols, (m, n), nnz)
        xd = G.dev(x)
        y = torch.full((m,), float("nan"), device="cuda")
        info = sp.multiply_inspect(sp.matrix_opt(a), xd, y)
        if varbins in ("0", "rule"):
            assert info.state_.info()["alg"] == _capi.SPMV_ROWBLOCK, what
            assert info.state_.sliced_info()["auto_trial"] == 0
        else:
            assert info.state_.info()["alg"] in (_capi.SPMV_ROWBLOCK, _capi.SPMV_SLICED), what
        sp.multiply(info, a, xd, y)
        check(values, rowptr, cols, (m, n), x, G.host(y), what=what + " (auto)", ref_cmp=False)
        info = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED)
        assert info.state_.info()["expand_items"] > 0
        sp.multiply(info, a, xd, y)
        check(values, rowptr, cols, (m, n), x, G.host(y), what=what + " (forced sliced)", ref_cmp=False)
    # heavy rows below the hub threshold concentrated in one corner of the matrix
    lens = np.full(m, 4, np.int64)
    lens[:3000] = 1500
    rowptr2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz2 = int(rowptr2[-1])
    col2 = rng.integers(0, n, nnz2).astype(np.int32)
    val2 = (rng.random(nnz2) + 0.5).astype(np.float32)
    a = G.csr_on_device(val2, rowptr2, col2, (m, n), nnz2)
    xd = G.dev(x)
    y = torch.full((m,), float("nan"), device="cuda")
    info = sp.multiply_inspect(sp.matrix_opt(a), xd, y)
    if varbins == "0":
        assert info.state_.info()["alg"] == _capi.SPMV_ROWBLOCK
    sp.multiply(info, a, xd, y)
    check(val2, rowptr2, col2, (m, n), x, G.host(y), what="heavy row block (auto)", ref_cmp=False)
    # forced SLICED: the reduce runs from its work list (heavy bin groups split over several workgroups,
    # pb_combine_items_kernel sums their partial rows); a row-range call takes the uniform path instead
    info = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED)
    if varbins == "0":
        assert info.state_.info()["reduce_items"] > 0 and info.state_.sliced_info()["variable_bins"] == 0
    else:
        si = info.state_.sliced_info()
        assert si["variable_bins"] == 1 and si["n_bins"] > cdiv(m, info.state_.info()["rows_per_bin"])
    y.fill_(float("nan"))
    sp.multiply(info, sp.scaled(0.5, a), xd, y)
    check(val2, rowptr2, col2, (m, n), x, G.host(y), scale=0.5, what="heavy row block (forced sliced)",
          ref_cmp=False)
    y2 = torch.full((m,), float("nan"), device="cuda")
    expand, reduce_rows = info.state_.bind_stages(xd, y2.data_ptr(), torch.float32, alpha=0.5)
    expand()
    H = info.state_.info()["rows_per_bin"]
    cut = (m // 3 // H) * H
    reduce_rows(0, cut)
    reduce_rows(cut, m)
    check(val2, rowptr2, col2, (m, n), x, G.host(y2), scale=0.5, what="heavy row block (two-stage)", ref_cmp=False)
    a.values().mul_(-1.0)
    info.state_.update_values(a.values())
    sp.multiply(info, a, xd, y)
    check(-val2, rowptr2, col2, (m, n), x, G.host(y), what="heavy row block (after update)", ref_cmp=False)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("nt", ["0", "1"])
def test_spmv_sliced_product_store_flavours(gpu, monkeypatch, dtype, nt):
    """The expand kernel exists with plain and with non-temporal product stores (which is faster depends on the box;
    a handle can ask for a timed trial, csrc/spmv.hip: store_trial).  Forced either way here on a
    matrix with ragged rows, a long row and duplicates: same answers, and the plan reports what it runs."""
    monkeypatch.setenv("SPBLAS_GFX950_PB_NT", nt)
    rng = np.random.default_rng(17)
    m, n = 40000, 90000
    lens = rng.integers(0, 25, m)
    lens[123] = 20000
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(rowptr[-1])
    colind = rng.integers(0, n, nnz).astype(np.int32)
    colind[:50] = 7  # duplicates inside the first rows
    values = (rng.random(nnz) - 0.5).astype(dtype)
    x = (rng.random(n) - 0.5).astype(dtype)
    a = G.csr_on_device(values, rowptr, colind, (m, n), nnz)
    xd = G.dev(x)
    y = torch.full((m,), float("nan"), dtype=xd.dtype, device="cuda")
    info = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED)
    si = info.state_.sliced_info()
    assert si["nt_product_stores"] == int(nt) and si["store_trial"] == 0
    sp.multiply(info, a, xd, y)
    check(values, rowptr, colind, (m, n), x, G.host(y), what=f"product stores nt={nt}", ref_cmp=False)


def test_spmv_store_trial_runs_once_per_process_at_cfg2_size(gpu, monkeypatch):
    """Opt-in (SPBLAS_GFX950_PB_NT=-2, or SPBLAS_GFX950_OPT_STORE_TRIAL = 2 on the handle): the first plan of a handle with
    >= 32 M placed entries times the SpMV with plain and with non-temporal product stores (the faster flavour is a property
    of the box: tools/exp_r03o.sh) and the handle keeps the decision.  Run in a fresh interpreter: without the opt-in no
    trial runs and the stores are plain; with it the first cfg2-sized plan reports the trial and its two times, the second
    one only the decision; both forced flavours give the same y to rounding."""
    import subprocess
    import sys
    code = r'''
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
import spblas_reference_amd as sp
from spblas_reference_amd import generate, _capi
n = 10_000_000
v, rp, ci, shape, nnz = generate.uniform_csr_device(n, n, 10, seed=0)
a = sp.csr_view(v, rp, ci, shape, nnz)
x = torch.rand(n, device="cuda"); ys = []
out = {}
for tag, env in (("default", None), ("first", "-2"), ("second", "-2"), ("plain", "0"), ("nt", "1")):
    if env is None:
        os.environ.pop("SPBLAS_GFX950_PB_NT", None)
    else:
        os.environ["SPBLAS_GFX950_PB_NT"] = env
    y = torch.full((n,), float("nan"), device="cuda")
    info = sp.multiply_inspect(sp.matrix_opt(a), x, y)
    assert info.state_.info()["alg"] == _capi.SPMV_SLICED
    sp.multiply(info, a, x, y)
    torch.cuda.synchronize()
    out[tag] = info.state_.sliced_info()
    ys.append(y)
    del info
absrow = torch.zeros(n, device="cuda")
sp.multiply(sp.csr_view(v.abs(), rp, ci, shape, nnz), x, absrow)
out["max_diff"] = max(float(((ys[i] - ys[3]).abs() / absrow).max()) for i in (0, 1, 2, 4))
print("RESULT " + json.dumps(out))
'''
    env = dict(os.environ)
    env.pop("SPBLAS_GFX950_PB_NT", None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    first, second = out["first"], out["second"]
    assert out["default"]["store_trial"] == 0 and out["default"]["nt_product_stores"] == 0
    assert first["store_trial"] == 1 and second["store_trial"] == 0
    t = first["store_trial_ns"]
    assert 150_000 < t["plain"] < 600_000 and 150_000 < t["non_temporal"] < 600_000, t
    assert first["nt_product_stores"] == int(t["non_temporal"] < 0.995 * t["plain"])
    assert second["nt_product_stores"] == first["nt_product_stores"]
    assert out["plain"]["nt_product_stores"] == 0 and out["nt"]["nt_product_stores"] == 1
    assert out["plain"]["store_trial"] == 0 and out["nt"]["store_trial"] == 0
    assert out["max_diff"] <= 2e-6


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spmv_store_trial_on_a_small_plan(gpu, monkeypatch, dtype):
    """The store trial of large plans forced onto a small one (SPBLAS_GFX950_PB_TUNE_MIN=0): six SpMVs on a zero vector at
    inspect, two temporary vectors that are given back, the same answers afterwards whichever flavour won."""
    monkeypatch.setenv("SPBLAS_GFX950_PB_TUNE_MIN", "0")
    monkeypatch.setenv("SPBLAS_GFX950_PB_NT", "-2")
    values, rowptr, colind, shape, nnz = generate.generate_csr(30000, 50000, 600000, dtype=dtype, seed=23)
    x = (np.random.default_rng(2).random(50000) - 0.5).astype(dtype)
    a = G.csr_on_device(values, rowptr, colind, shape, nnz)
    xd = G.dev(x)
    free0 = torch.cuda.mem_get_info()[0]
    for rep in range(3):
        y = torch.full((30000,), float("nan"), dtype=xd.dtype, device="cuda")
        info = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED)
        assert info.state_.sliced_info()["nt_product_stores"] in (0, 1)
        sp.multiply(info, a, xd, y)
        check(values, rowptr, colind, shape, x, G.host(y), what=f"store trial on a small plan {rep}", ref_cmp=False)
        del info
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < 64 * 2 ** 20


@pytest.mark.parametrize("offsets", [np.int32, np.int64])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spmv_hot_column_split(gpu, monkeypatch, dtype, offsets):
    """Column-skewed matrix (csrc/spmv_hot.hip): the entries in the most referenced columns are taken out of the tiles and
    multiplied in row order with their x values in LDS, the rest keeps the tiled plan.  Forced here on a small matrix with
    everything the split has to get right: empty rows, rows with hot entries only / without any, rows with more hot
    entries than a window of 256 (per-window partials + fix-up), duplicates, alpha and beta, a rebound value array
    (both halves take their values again through the recorded source positions), in-place update, two-stage calls."""
    monkeypatch.setenv("SPBLAS_GFX950_PB_HOT", "1")
    rng = np.random.default_rng(91)
    m, n = 60000, 300000
    lens = rng.integers(0, 14, m)
    lens[rng.random(m) < 0.3] = 0
    lens[[5, 777, 31000, m - 1]] = [9000, 2500, 700, 40000]
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(offsets)
    nnz = int(rowptr[-1])
    hot_cols = rng.choice(n, 3000, replace=False)
    hot = rng.random(nnz) < 0.45
    colind = np.where(hot, hot_cols[rng.integers(0, 3000, nnz)], rng.integers(0, n, nnz)).astype(np.int32)
    # one row with hot entries only, one without any
    colind[rowptr[777]:rowptr[778]] = hot_cols[rng.integers(0, 3000, 2500)]
    cold = np.setdiff1d(np.arange(n), hot_cols)
    colind[rowptr[31000]:rowptr[31001]] = cold[rng.integers(0, len(cold), 700)]
    colind[rowptr[5]:rowptr[5] + 40] = hot_cols[7]  # duplicates
    values = (rng.random(nnz) - 0.5).astype(dtype)
    x = (rng.random(n) - 0.5).astype(dtype)
    a = G.csr_on_device(values, rowptr, colind, (m, n), nnz, offset64=offsets == np.int64)
    xd = G.dev(x)
    y = torch.full((m,), float("nan"), dtype=xd.dtype, device="cuda")
    info = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED)
    si = info.state_.sliced_info()
    hs = si["hot_split"]
    assert 2000 <= hs["hot_columns"] <= 32768 and hs["hot_entries"] + hs["tiled_entries"] == nnz
    assert hs["hot_entries"] >= 0.3 * nnz and hs["hot_long_rows"] >= 3  # (the 1-in-16 sample misses some hot columns of so small a matrix)
    sp.multiply(info, a, xd, y)
    check(values, rowptr, colind, (m, n), x, G.host(y), what="hot split", ref_cmp=False)
    # alpha and beta through the C ABI
    y0 = (rng.random(m) - 0.5).astype(dtype)
    y.copy_(G.dev(y0))
    import ctypes
    ct = ctypes.c_float if dtype == np.float32 else ctypes.c_double
    al, be = ct(-1.5), ct(0.25)
    hd = sp.api._Handle.current(xd.device)
    sp.api.check(_capi.lib().spblas_gfx950_spmv(hd.h, info.state_.plan, _capi.OP_N, m, n, nnz, ctypes.byref(al),
                                                sp.api._ptr(a.rowptr()), sp.api._ptr(a.colind()), sp.api._ptr(a.values()),
                                                sp.api._ptr(xd), ctypes.byref(be), sp.api._ptr(y),
                                                _capi.I32 if offsets == np.int32 else _capi.I64,
                                                _capi.F32 if dtype == np.float32 else _capi.F64), "spmv")
    _, absrow = util.spmv_exact(rowptr, colind, values, x)
    y_ref = -1.5 * oracle.spmv((m, n), rowptr, colind, values, x).astype(np.float64) + 0.25 * y0
    util.assert_parity(G.host(y), y_ref, 1.5 * absrow + 0.25 * np.abs(y0), dtype, row_len=np.diff(rowptr) + 1,
                       what="hot split alpha / beta")
    # rebound value array: the next multiply refreshes both halves
    v2 = (rng.random(nnz) - 0.5).astype(dtype)
    a.update(G.dev(v2), a.rowptr(), a.colind())
    y.fill_(float("nan"))
    sp.multiply(info, a, xd, y)
    check(v2, rowptr, colind, (m, n), x, G.host(y), what="hot split, rebound values", ref_cmp=False)
    # in place
    a.values().mul_(2.0)
    sp.multiply(info, a, xd, y)
    check((v2 * dtype(2)).astype(dtype), rowptr, colind, (m, n), x, G.host(y), what="hot split, values scaled in place",
          ref_cmp=False)
    # two-stage form: all rows in one reduce; a proper row range is refused (the hot part adds into y afterwards)
    expand, reduce_rows = info.state_.bind_stages(xd, y.data_ptr(), xd.dtype)
    y.fill_(float("nan"))
    expand()
    reduce_rows(0, m)
    check((v2 * dtype(2)).astype(dtype), rowptr, colind, (m, n), x, G.host(y), what="hot split, two-stage", ref_cmp=False)
    with pytest.raises(Exception):
        reduce_rows(0, m // 2)
    # the same matrix without the split gives the same answer to rounding
    monkeypatch.setenv("SPBLAS_GFX950_PB_HOT", "0")
    info2 = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED)
    assert "hot_split" not in info2.state_.sliced_info()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spmv_plain_inspected_csr_view_follows_values_written_in_place(gpu, dtype):
    """A plain csr_view (no matrix_opt) that is inspected: every multiply reads the caller's values OF THAT CALL
    (multiply_impl.hpp:48-52).  Large uniform matrices get the re-tiled plan here too since the end of round 4 -- in the
    form that takes the values again on every multiply (pb_refresh_bins_kernel), kept only when it beats the row-block
    kernel in the timed trial -- so values rewritten in place by a foreign kernel (no API call, no torch version bump that
    the library could see) must show in the next y.  SPBLAS_GFX950_PLAIN_SLICED=0 restores the row-block plan."""
    rng = np.random.default_rng(93)
    m = n = 1_800_000
    values, rowptr, colind, shape, nnz = generate.uniform_csr_device(m, n, 10, seed=4, dtype=torch.float32 if dtype == np.float32 else torch.float64)
    a = sp.csr_view(values, rowptr, colind, shape, nnz)
    x = torch.rand(n, dtype=values.dtype, device="cuda")
    y = torch.full((m,), float("nan"), dtype=values.dtype, device="cuda")
    info = sp.multiply_inspect(a, x, y)
    pi = info.state_.info()
    assert pi["alg"] in (_capi.SPMV_ROWBLOCK, _capi.SPMV_SLICED)
    if pi["alg"] == _capi.SPMV_SLICED:
        assert info.state_.sliced_info()["refresh_each_call"] == 1
        assert info.state_.sliced_info()["value_free"] == 1  # round 5: no copy of the values in the plan, no timed trial
        assert info.state_.sliced_info()["auto_trial"] == 0
    rows = np.unique(np.concatenate([np.arange(0, 800), np.arange(m - 800, m), rng.integers(0, m, 1500)]))
    rp_h = rowptr.cpu().numpy()
    idx = np.concatenate([np.arange(rp_h[r], rp_h[r + 1]) for r in rows])
    sub_rp = np.concatenate([[0], np.cumsum(rp_h[rows + 1] - rp_h[rows])]).astype(np.int32)
    sub_ci = colind.cpu().numpy()[idx]
    xh = x.cpu().numpy()
    for step in range(3):
        # rewrite the values through a raw view of the same memory: no version counter of `values` moves
        raw = torch.as_strided(values, values.shape, values.stride())
        raw.data.mul_(-1.25 if step else 1.0).add_(0.0625 * step)
        sp.multiply(info, a, x, y)
        sub_v = values.cpu().numpy()[idx]
        y_ref = oracle.spmv((len(rows), n), sub_rp, sub_ci, sub_v, xh)
        absrow = oracle.spmv_absrow(sub_rp, sub_ci, sub_v, xh)
        util.assert_parity(y[torch.from_numpy(rows).cuda()].cpu().numpy(), y_ref, absrow, dtype, row_len=np.diff(sub_rp),
                           what=f"plain inspected csr_view, values rewritten in place, step {step}")
    # ... also when the multiply is a recorded HIP graph: the replay reads the array as it is THEN
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        sp.multiply(info, a, x, y)
    torch.cuda.current_stream().wait_stream(s_)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        sp.multiply(info, a, x, y)
    values.mul_(0.5).sub_(0.03125)
    y.fill_(float("nan"))
    g.replay()
    torch.cuda.synchronize()
    sub_v = values.cpu().numpy()[idx]
    y_ref = oracle.spmv((len(rows), n), sub_rp, sub_ci, sub_v, xh)
    absrow = oracle.spmv_absrow(sub_rp, sub_ci, sub_v, xh)
    util.assert_parity(y[torch.from_numpy(rows).cuda()].cpu().numpy(), y_ref, absrow, dtype, row_len=np.diff(sub_rp),
                       what="plain inspected csr_view, graph replay after an in-place change")


@pytest.mark.parametrize("offsets", [np.int32, np.int64])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("waves,enc", [("4", "0"), ("4", "2"), ("8", "2"), ("8", "0")])
def test_spmv_value_free_tiles(gpu, monkeypatch, dtype, offsets, waves, enc):
    """Round 5: the tiled plan of a plain inspected csr_view holds NO copy of A's values.  The expand writes the gathered
    x[col]; the reduce of a bin stages the bin's window of the caller's values (rowptr[r0] .. rowptr[r1]) in LDS and
    multiplies through a 16-bit in-window position per entry (pb_reduce_vf_kernel: one bin per workgroup, accumulators per
    wavefront, partial rows added in wave order).  Small tiles through the test hooks (SPBLAS_GFX950_PB_VFREE=2 builds the
    value-free form on explicit request, PB_VF_ROWS caps the bin height), ragged rows so that windows start at every
    alignment, a value array that is itself only 4-byte aligned, both row encodings, values rewritten in place between
    multiplies (multiply_impl.hpp:48-52: the values of THAT call), alpha / beta."""
    monkeypatch.setenv("SPBLAS_GFX950_PB_VFREE", "2")
    monkeypatch.setenv("SPBLAS_GFX950_SLICE_COLS", "128")
    monkeypatch.setenv("SPBLAS_GFX950_PB_VF_ROWS", "300")
    monkeypatch.setenv("SPBLAS_GFX950_PB_VF_WAVES", waves)
    monkeypatch.setenv("SPBLAS_GFX950_PB_ENC8", enc)
    rng = np.random.default_rng(41)
    m, n = 7001, 2500
    lens = rng.integers(0, 30, m)
    lens[rng.random(m) < 0.1] = 0
    lens[1234] = 240                                   # a row that repeats inside runs (duplicate flags)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(offsets)
    nnz = int(rowptr[-1])
    colind = rng.integers(0, n, nnz).astype(np.int32)
    values = (rng.random(nnz) - 0.5).astype(dtype)
    x = (rng.random(n) - 0.5).astype(dtype)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    for misaligned in (False, True):
        backing = torch.zeros(nnz + 4, dtype=tdt, device="cuda")
        vt = backing[1:nnz + 1] if misaligned else backing[:nnz]
        vt.copy_(torch.from_numpy(values))
        assert (vt.data_ptr() % 16 != 0) == misaligned
        a = sp.csr_view(vt, G.dev(rowptr), G.dev(colind), (m, n), nnz)
        xd = G.dev(x)
        y = torch.full((m,), float("nan"), dtype=tdt, device="cuda")
        info = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED)
        si = info.state_.sliced_info()
        assert info.state_.info()["alg"] == _capi.SPMV_SLICED and si["value_free"] == 1 and si["refresh_each_call"] == 1, si
        assert si["row_code_u8"] == (1 if enc == "2" else 0)
        rp32 = rowptr.astype(np.int32)
        sp.multiply(info, a, xd, y)
        check(values, rp32, colind, (m, n), x, G.host(y), what=f"value-free tiles misaligned={misaligned}", ref_cmp=False)
        # values rewritten in place through a raw view: the next multiply reads them
        raw = torch.as_strided(vt, vt.shape, vt.stride())
        raw.data.mul_(-1.25).add_(0.0625)
        v2 = (values * dtype(-1.25) + dtype(0.0625)).astype(dtype)
        y.fill_(float("nan"))
        sp.multiply(info, sp.scaled(2.0, a), xd, y)
        check(v2, rp32, colind, (m, n), x, G.host(y), scale=2.0, what="value-free tiles, values rewritten in place", ref_cmp=False)
        # alpha / beta through the C ABI
        y0 = (rng.random(m) - 0.5).astype(dtype)
        y.copy_(torch.from_numpy(y0))
        ct = ctypes.c_float if dtype == np.float32 else ctypes.c_double
        al, be = ct(-1.5), ct(0.25)
        hd = sp.api._Handle.current(xd.device)
        sp.api.check(_capi.lib().spblas_gfx950_spmv(hd.h, info.state_.plan, _capi.OP_N, m, n, nnz, ctypes.byref(al),
                                                    sp.api._ptr(a.rowptr()), sp.api._ptr(a.colind()), sp.api._ptr(a.values()),
                                                    sp.api._ptr(xd), ctypes.byref(be), sp.api._ptr(y),
                                                    _capi.I32 if offsets == np.int32 else _capi.I64,
                                                    _capi.F32 if dtype == np.float32 else _capi.F64), "spmv")
        _, absrow = util.spmv_exact(rp32, colind, v2, x)
        y_ref = -1.5 * oracle.spmv((m, n), rp32, colind, v2, x).astype(np.float64) + 0.25 * y0
        util.assert_parity(G.host(y), y_ref, 1.5 * absrow + 0.25 * np.abs(y0), dtype, row_len=np.diff(rp32) + 1,
                           what="value-free tiles alpha / beta")
        # the multi-GPU step (round 6: value-free plans are accepted -- the reduce has the broadcast epilogue and reads the
        # value array registered with the plan as it is when the step runs): one "rank", two copies of y as the peers, rows
        # at an offset; values rewritten in place once more before the step
        raw.data.mul_(0.5).sub_(0.125)
        v3 = (v2 * dtype(0.5) - dtype(0.125)).astype(dtype)
        big = torch.full((2, m + 7), float("nan"), dtype=tdt, device="cuda")
        tab = torch.tensor([big[0].data_ptr(), big[1].data_ptr()], dtype=torch.int64, device="cuda")
        rc = _capi.lib().spblas_gfx950_spmv_step_bcast(hd.h, info.state_.plan, ctypes.byref(al), sp.api._ptr(xd),
                                                       ctypes.c_void_p(tab.data_ptr()), 2, 7, 1)
        assert rc == _capi.SUCCESS, rc
        torch.cuda.synchronize()
        assert bool(torch.isnan(big[:, :7]).all()) and torch.equal(big[0, 7:], big[1, 7:])
        _, absrow3 = util.spmv_exact(rp32, colind, v3, x)
        util.assert_parity(G.host(big[0, 7:]), -1.5 * oracle.spmv((m, n), rp32, colind, v3, x).astype(np.float64), 1.5 * absrow3,
                           dtype, row_len=np.diff(rp32), what="value-free tiles, fused broadcast step")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("waves", [8, 4])
@pytest.mark.parametrize("offset_mod", [1, 3])
def test_spmv_value_free_tiles_widest_window_misaligned(gpu, monkeypatch, dtype, waves, offset_mod):
    """Round-5 advisor (high): a bin whose span of the caller's array equals the accepted capacity EXACTLY, starting at an
    offset that is not a multiple of 16 bytes.  stage_window copies from the aligned-down start, i.e. span + shift elements;
    the window area therefore holds the capacity + 16 (the accumulators of wave 0 used to begin right at the capacity and
    received raw values of A in rows 0 .. 2 of such a bin).  Bin height 500 through the test hook; the capacity follows
    csrc/spmv_sliced.hip: (160 KiB - 64) / sizeof(T) - waves * (500 + 64) - 16."""
    H = 500
    monkeypatch.setenv("SPBLAS_GFX950_PB_VFREE", "2")
    monkeypatch.setenv("SPBLAS_GFX950_SLICE_COLS", "128")
    monkeypatch.setenv("SPBLAS_GFX950_PB_VF_ROWS", str(H))
    monkeypatch.setenv("SPBLAS_GFX950_PB_VF_WAVES", str(waves))
    size = np.dtype(dtype).itemsize
    cap = (160 * 1024 - 64) // size - waves * (H + 64) - 16
    assert cap < 65536
    rng = np.random.default_rng(53)
    m, n = 3 * H, 3000
    lens = np.empty(m, dtype=np.int64)
    lens[:H] = 10
    lens[7] += offset_mod                               # bin 1 starts at 5000 + offset_mod: never 16-byte aligned for fp32,
    lens[H:2 * H] = cap // H                            # odd for fp64
    lens[H:H + cap % H] += 1                            # bin 1 spans exactly `cap` entries
    lens[2 * H:] = rng.integers(0, 20, H)
    assert lens[H:2 * H].sum() == cap and lens[:H].sum() % (16 // size) != 0
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(rowptr[-1])
    colind = rng.integers(0, n, nnz).astype(np.int32)
    values = (1.0 + rng.random(nnz)).astype(dtype)      # all >= 1: a stray value in an accumulator cannot hide
    x = (rng.random(n) - 0.5).astype(dtype)
    a = G.csr_on_device(values, rowptr, colind, (m, n), nnz)
    assert a.values().data_ptr() % 16 == 0
    xd = G.dev(x)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    y = torch.full((m,), float("nan"), dtype=tdt, device="cuda")
    info = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED)
    si = info.state_.sliced_info()
    assert si["value_free"] == 1 and info.state_.info()["rows_per_bin"] == H, (si, info.state_.info())
    for _ in range(2):                                  # (persistent workgroups: the second pass reuses warm accumulators)
        y.fill_(float("nan"))
        sp.multiply(info, a, xd, y)
        check(values, rowptr, colind, (m, n), x, G.host(y), what="value-free tiles, widest window at a misaligned offset",
              ref_cmp=False)


def test_spmv_value_free_tiles_fall_back_when_the_matrix_does_not_fit(gpu, monkeypatch):
    """A matrix the value-free form does not take (a row far longer than the rest is cut into pieces: a row map) still
    gets the copying plan on explicit request, and the answer."""
    monkeypatch.setenv("SPBLAS_GFX950_PB_VFREE", "2")
    monkeypatch.setenv("SPBLAS_GFX950_SLICE_COLS", "128")
    rng = np.random.default_rng(43)
    m, n = 3000, 2500
    lens = rng.integers(1, 12, m)
    lens[77] = 6000
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(rowptr[-1])
    colind = rng.integers(0, n, nnz).astype(np.int32)
    values = (rng.random(nnz) - 0.5).astype(np.float32)
    x = (rng.random(n) - 0.5).astype(np.float32)
    a = G.csr_on_device(values, rowptr, colind, (m, n), nnz)
    xd = G.dev(x)
    y = torch.full((m,), float("nan"), device="cuda")
    info = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED)
    si = info.state_.sliced_info()
    assert si["value_free"] == 0 and si["refresh_each_call"] == 0, si
    sp.multiply(info, a, xd, y)
    check(values, rowptr, colind, (m, n), x, G.host(y), what="fallback from value-free tiles", ref_cmp=False)


@pytest.mark.parametrize("shape", ["uniform", "hot_split"])
def test_spmv_snapshot_plan_keeps_no_source_positions_until_the_values_change(gpu, monkeypatch, shape):
    """Round 5: a snapshot plan (matrix_opt / explicit SLICED) is built WITHOUT the 4 B per entry of source positions a value
    refresh gathers through -- and, for a hot-column split, without the CSR copy of A_rest its tiles were made from.  The
    first change of values builds the plan again from the caller's arrays, this time with them (device_bytes grows); later
    changes are the cheap gather.  Results follow the values at every step."""
    monkeypatch.setenv("SPBLAS_GFX950_SLICE_COLS", "128")
    rng = np.random.default_rng(47)
    m, n = 6000, 4000
    if shape == "hot_split":
        monkeypatch.setenv("SPBLAS_GFX950_PB_HOT", "1")
        lens = rng.integers(1, 40, m)
        nnz = int(lens.sum())
        colind = np.where(rng.random(nnz) < 0.4, rng.integers(0, 64, nnz), rng.integers(0, n, nnz)).astype(np.int32)
    else:
        lens = rng.integers(0, 24, m)
        nnz = int(lens.sum())
        colind = rng.integers(0, n, nnz).astype(np.int32)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    values = (rng.random(nnz) - 0.5).astype(np.float64)
    x = (rng.random(n) - 0.5).astype(np.float64)
    a = G.csr_on_device(values, rowptr, colind, (m, n), nnz)
    xd = G.dev(x)
    y = torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
    info = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED)
    if shape == "hot_split":
        assert "hot_split" in info.state_.sliced_info()
    bytes0 = info.state_.info()["device_bytes"]
    sp.multiply(info, a, xd, y)
    check(values, rowptr, colind, (m, n), x, G.host(y), what=f"{shape}: as inspected", ref_cmp=False)
    a.values().mul_(-2.0)                                  # first change: the plan is built again, with source positions
    info.state_.update_values(a.values())
    bytes1 = info.state_.info()["device_bytes"]
    assert bytes1 >= bytes0 + 4 * nnz, (bytes0, bytes1, nnz)
    sp.multiply(info, a, xd, y)
    check(values * -2.0, rowptr, colind, (m, n), x, G.host(y), what=f"{shape}: first change", ref_cmp=False)
    a.values().add_(0.125)                                 # second change: the gather through the source positions
    info.state_.update_values(a.values())
    assert info.state_.info()["device_bytes"] == bytes1
    sp.multiply(info, a, xd, y)
    check(values * -2.0 + 0.125, rowptr, colind, (m, n), x, G.host(y), what=f"{shape}: second change", ref_cmp=False)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["uniform", "hot_split"])
@pytest.mark.parametrize("via", ["update_values", "new_pointer"])
def test_spmv_snapshot_plan_whose_second_build_fails_falls_back_to_the_structure_only_plan(gpu, monkeypatch, shape, via):
    """Round-5 advisor (medium): the first value change of a snapshot plan builds the tiles again (with source positions).
    When that build fails -- out of memory with the pool still warm, or declined -- the plan must not stay `SLICED` on freed
    arrays: it goes back to the row-block plan on the caller's arrays (alg changes, device_bytes drops to the structure-only
    figure), the call succeeds, and every later multiply, whichever values pointer it passes, is right.  The failure is
    injected after a complete second build (SPBLAS_GFX950_TEST_FAIL_REBUILD), so the teardown frees a whole plan."""
    monkeypatch.setenv("SPBLAS_GFX950_SLICE_COLS", "128")
    rng = np.random.default_rng(59)
    m, n = 6000, 4000
    lens = rng.integers(1, 40, m)
    nnz = int(lens.sum())
    if shape == "hot_split":
        monkeypatch.setenv("SPBLAS_GFX950_PB_HOT", "1")
        colind = np.where(rng.random(nnz) < 0.4, rng.integers(0, 64, nnz), rng.integers(0, n, nnz)).astype(np.int32)
    else:
        colind = rng.integers(0, n, nnz).astype(np.int32)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    values = (rng.random(nnz) - 0.5).astype(np.float32)
    x = (rng.random(n) - 0.5).astype(np.float32)
    a = G.csr_on_device(values, rowptr, colind, (m, n), nnz)
    xd = G.dev(x)
    y = torch.full((m,), float("nan"), device="cuda")
    info = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED)
    assert info.state_.info()["alg"] == _capi.SPMV_SLICED
    bytes_tiles = info.state_.info()["device_bytes"]
    sp.multiply(info, a, xd, y)
    check(values, rowptr, colind, (m, n), x, G.host(y), what="as inspected", ref_cmp=False)
    monkeypatch.setenv("SPBLAS_GFX950_TEST_FAIL_REBUILD", "1")
    v2 = (values * np.float32(-2.0)).astype(np.float32)
    if via == "update_values":
        a.values().mul_(-2.0)
        info.state_.update_values(a.values())
        a2 = a
    else:  # a multiply that passes another array refreshes by itself
        a2 = sp.csr_view(G.dev(v2), a.rowptr(), a.colind(), (m, n), nnz)
    y.fill_(float("nan"))
    sp.multiply(info, a2, xd, y)
    monkeypatch.delenv("SPBLAS_GFX950_TEST_FAIL_REBUILD")
    pi = info.state_.info()
    assert pi["alg"] == _capi.SPMV_ROWBLOCK and pi["device_bytes"] < bytes_tiles, pi
    check(v2, rowptr, colind, (m, n), x, G.host(y), what="after the failed second build", ref_cmp=False)
    # the same pointer again (the call that used to skip the update and run on a torn-down plan), then a changed array
    y.fill_(float("nan"))
    sp.multiply(info, a2, xd, y)
    check(v2, rowptr, colind, (m, n), x, G.host(y), what="same pointer after the fallback", ref_cmp=False)
    a2.values().add_(0.25)
    y.fill_(float("nan"))
    sp.multiply(info, a2, xd, y)
    check(v2 + np.float32(0.25), rowptr, colind, (m, n), x, G.host(y), what="values changed in place after the fallback",
          ref_cmp=False)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["plain", "hot_split"])
def test_spmv_snapshot_plan_keeps_source_positions_from_the_start_on_request(gpu, monkeypatch, shape):
    """SPBLAS_GFX950_OPT_VALUE_SNAPSHOT = 2 (multiply_inspect(..., values_will_change=True)): the caller announces that the
    values will change, the plan holds its source positions from inspect on -- the first update_values is the gather, the
    plan is not built again (device_bytes does not move) -- and is 4 B per entry larger than the default plan."""
    monkeypatch.setenv("SPBLAS_GFX950_SLICE_COLS", "128")
    rng = np.random.default_rng(53)
    m, n = 6000, 4000
    if shape == "hot_split":
        monkeypatch.setenv("SPBLAS_GFX950_PB_HOT", "1")
        lens = rng.integers(1, 40, m)
        nnz = int(lens.sum())
        colind = np.where(rng.random(nnz) < 0.4, rng.integers(0, 64, nnz), rng.integers(0, n, nnz)).astype(np.int32)
    else:
        lens = rng.integers(0, 24, m)
        nnz = int(lens.sum())
        colind = rng.integers(0, n, nnz).astype(np.int32)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    values = (rng.random(nnz) - 0.5).astype(np.float32)
    x = (rng.random(n) - 0.5).astype(np.float32)
    a = G.csr_on_device(values, rowptr, colind, (m, n), nnz)
    xd = G.dev(x)
    y = torch.full((m,), float("nan"), dtype=torch.float32, device="cuda")
    lean = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED).state_.info()["device_bytes"]
    info = sp.multiply_inspect(a, xd, y, alg=_capi.SPMV_SLICED, values_will_change=True)
    bytes0 = info.state_.info()["device_bytes"]
    assert bytes0 >= lean + 4 * nnz, (lean, bytes0, nnz)
    sp.multiply(info, a, xd, y)
    check(values, rowptr, colind, (m, n), x, G.host(y), what=f"{shape}: as inspected", ref_cmp=False)
    for k, f in enumerate((-2.0, 0.5)):
        a.values().mul_(f)
        values = values * np.float32(f)
        info.state_.update_values(a.values())
        assert info.state_.info()["device_bytes"] == bytes0, "the plan was built again"
        sp.multiply(info, a, xd, y)
        check(values, rowptr, colind, (m, n), x, G.host(y), what=f"{shape}: change {k}", ref_cmp=False)
    # the option leaves a plain inspected view's semantics alone: AUTO without matrix_opt still reads the caller's values
    plain = sp.multiply_inspect(a, xd, y, values_will_change=True)
    assert plain.state_.info()["alg"] != _capi.SPMV_SLICED or plain.state_.sliced_info()["refresh_each_call"]
