"""-m gpu: every BASELINE.json config at its STATED size (cfg2 lives in test_gpu_spmv.py::test_spmv_full_size_properties_cfg2).

  cfg1  examples/simple_spmv.cpp plumbing case: fp32 CSR 10k x 10k, nnz = 1e6 (1 %), the reference generator's
        distribution (backend/generate.hpp:106-120) -- compared with the oracle in full, every algorithm
  cfg3  fp32 CSR x dense SpMM, A 2M x 2M 32 nnz/row, B 2M x 128 -- linearity + sampled rows vs the oracle,
        plus an R-MAT A of the same size class (hub rows) against the same checks
  cfg4  fp64 CSR SpMV, R-MAT scale 24 (268 M entries) -- checksum, linearity, the 2 000 heaviest rows and
        4 000 sampled rows vs the oracle at 1e-12 norm-wise, AUTO and forced SLICED
  cfg5  fp32 CSR x CSR SpGEMM 1M x 1M, 16 nnz/row -- nnz(C) and rowptr EXACT against a full oracle symbolic
        run, sorted columns, (AB)x = A(Bx), sampled rows exact

The oracle runs on the host only over what it finishes in seconds; everything at full
size is checked through size-independent properties on the device.
"""
import numpy as np
import pytest
import torch

import gpu_util as G
import spblas_reference_amd as sp
import util
from oracle import oracle
from spblas_reference_amd import _capi, generate

pytestmark = pytest.mark.gpu


def _rows_subproblem(rows, rowptr_d, colind_d, values_d):
    """CSR of the selected rows (host arrays), gathered on the device."""
    rows_d = torch.from_numpy(np.asarray(rows, dtype=np.int64)).cuda()
    rp = rowptr_d.long()
    lo, ln = rp[rows_d], rp[rows_d + 1] - rp[rows_d]
    sub_rp = torch.zeros(len(rows) + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(ln, 0, out=sub_rp[1:])
    total = int(sub_rp[-1])
    # position of every gathered entry: start of its row + offset inside the row
    owner = torch.repeat_interleave(torch.arange(len(rows), device="cuda"), ln)
    idx = lo[owner] + (torch.arange(total, device="cuda") - sub_rp[owner])
    return sub_rp.cpu().numpy().astype(np.int32), colind_d[idx].cpu().numpy(), values_d[idx].cpu().numpy()


# --------------------------------------------------------------------------------------------- cfg1
@pytest.mark.parametrize("alg", ["noplan", "auto", "vector", "rowblock", "sliced"])
def test_cfg1_simple_spmv_10k_1pct(gpu, alg):
    """BASELINE cfg1: the examples/simple_spmv.cpp call shape (multiply(scaled(a), x, y), :44-48) on
    10k x 10k with nnz = 1e6 distinct uniformly drawn entries, values U[0,100), columns unsorted inside a row --
    what spblas::generate_csr produces (backend/generate.hpp:49-120).  Every row against the oracle and fp64."""
    m = n = 10_000
    values, rowptr, colind, shape, nnz = generate.generate_csr(m, n, 1_000_000, seed=0)
    assert nnz == 1_000_000
    a = G.csr_on_device(values, rowptr, colind, shape, nnz)
    x = np.random.default_rng(0).random(n).astype(np.float32)
    y = torch.full((m,), float("nan"), device="cuda")
    algs = {"auto": _capi.SPMV_AUTO, "vector": _capi.SPMV_VECTOR, "rowblock": _capi.SPMV_ROWBLOCK,
            "sliced": _capi.SPMV_SLICED}
    if alg == "noplan":
        sp.multiply(sp.scaled(1.2, a), G.dev(x), y)
    else:
        info = sp.multiply_inspect(a, G.dev(x), y, alg=algs[alg])
        sp.multiply(info, sp.scaled(1.2, a), G.dev(x), y)
    y_ref = oracle.spmv(shape, rowptr, colind, values, x, scale_a=1.2)
    exact, absrow = util.spmv_exact(rowptr, colind, values, x)
    lens = np.diff(rowptr)
    util.assert_parity(G.host(y), y_ref, 1.2 * absrow, np.float32, row_len=lens, what=f"cfg1 {alg} vs oracle")
    util.assert_parity(G.host(y), 1.2 * exact, 1.2 * absrow, np.float32, row_len=lens, what=f"cfg1 {alg} vs fp64")
    util.expect_eq_ref(y_ref, G.host(y))  # the reference's own comparator (positive data)


# --------------------------------------------------------------------------------------------- cfg3
def _absprod_rows(sub_rp, sub_c, sub_v, B_sub):
    import scipy.sparse as sps
    A = sps.csr_matrix((np.abs(sub_v).astype(np.float64), sub_c, sub_rp), shape=(len(sub_rp) - 1, B_sub.shape[0]))
    return A @ np.abs(B_sub).astype(np.float64)


def _check_spmm_rows(rows, rowptr, colind, values, B, C, what):
    """Selected rows of C against oracle.spmm; the B rows they touch are compacted on the device first."""
    sub_rp, sub_c, sub_v = _rows_subproblem(rows, rowptr, colind, values)
    uniq, inv = np.unique(sub_c, return_inverse=True)
    B_sub = B[torch.from_numpy(uniq.astype(np.int64)).cuda()].cpu().numpy()
    C_ref = oracle.spmm((len(rows), len(uniq)), sub_rp, inv.astype(np.int32), sub_v, B_sub)
    got = C[torch.from_numpy(np.asarray(rows, dtype=np.int64)).cuda()].cpu().numpy()
    util.assert_parity(got, C_ref, _absprod_rows(sub_rp, inv.astype(np.int32), sub_v, B_sub), np.float32,
                       row_len=np.diff(sub_rp), what=what)


@pytest.mark.parametrize("inspect", [False, True])
def test_cfg3_spmm_full_size(gpu, inspect):
    """BASELINE cfg3 at its stated size: A 2M x 2M, 32 nnz/row uniform, B 2M x 128 row-major (1 GB), fp32."""
    m = k = 2_000_000
    values, rowptr, colind, shape, nnz = generate.uniform_csr_device(m, k, 32, seed=0)
    a = sp.csr_view(values, rowptr, colind, shape, nnz)
    g = torch.Generator(device="cuda").manual_seed(2)
    B1 = torch.rand((k, 128), device="cuda", generator=g)
    B2 = torch.rand((k, 128), device="cuda", generator=g)
    C1, C2, C3 = (torch.full((m, 128), float("nan"), device="cuda") for _ in range(3))
    if inspect:
        info = sp.multiply_inspect(sp.matrix_opt(a), B1, C1)  # examples/spmm_csr.cpp:45-46
        run = lambda B, C: sp.multiply(info, a, B, C)          # noqa: E731
    else:
        run = lambda B, C: sp.multiply(a, B, C)                # noqa: E731
    run(B1, C1)
    run(B2, C2)
    Bm = B1 - 3.0 * B2
    run(Bm, C3)
    del Bm
    lin = (C3 - (C1 - 3.0 * C2)).abs_()
    bound = (C1.abs() + 3.0 * C2.abs()).mul_(8e-6).add_(1e-30)
    assert bool((lin <= bound).all()), "cfg3 linearity in B"
    del lin, bound, C2, C3
    # column checksum: sum_i C[i, :] == sum_p v_p * B[c_p, :] in fp64, per output column
    colsum = C1.double().sum(0)
    w = torch.zeros(k, dtype=torch.float64, device="cuda").index_add_(0, colind.long(), values.double())
    ref = (w[:, None] * B1.double()).sum(0)
    assert bool(((colsum - ref).abs() <= 1e-6 * ref.abs()).all()), "cfg3 column checksum"
    rows = np.unique(np.concatenate([np.arange(1500), np.arange(m - 1500, m),
                                     np.random.default_rng(0).integers(0, m, 1500)]))
    _check_spmm_rows(rows, rowptr, colind, values, B1, C1, f"cfg3 sampled rows (inspect={inspect})")


def test_cfg3_spmm_rmat_hub_rows(gpu):
    """An R-MAT A of cfg3's size class (2M x 2M, 64 M entries; heaviest rows ~1e5 entries): the hub rows must not
    be walked by one lane group -- checked for correctness on the heaviest and on sampled rows, n = 128."""
    values, rowptr, colind, shape, nnz = generate.rmat_csr_device(21, 32, dtype=torch.float32, seed=1)
    m = k = shape[0]
    a = sp.csr_view(values, rowptr, colind, shape, nnz)
    g = torch.Generator(device="cuda").manual_seed(4)
    B = torch.rand((k, 128), device="cuda", generator=g)
    C = torch.full((m, 128), float("nan"), device="cuda")
    info = sp.multiply_inspect(sp.matrix_opt(a), B, C)
    assert info.state_.spmm_info()["long_rows"] > 100  # the hubs go to the split long-row kernel
    sp.multiply(info, sp.scaled(0.5, a), B, C)
    assert bool(torch.isfinite(C).all())
    lens = (rowptr[1:].long() - rowptr[:-1].long())
    heavy = torch.topk(lens, 300).indices.cpu().numpy()
    rows = np.unique(np.concatenate([heavy, np.random.default_rng(1).integers(0, m, 2000)]))
    sub_rp, sub_c, sub_v = _rows_subproblem(rows, rowptr, colind, values)
    uniq, inv = np.unique(sub_c, return_inverse=True)
    B_sub = B[torch.from_numpy(uniq.astype(np.int64)).cuda()].cpu().numpy()
    C_ref = oracle.spmm((len(rows), len(uniq)), sub_rp, inv.astype(np.int32), sub_v, B_sub, scale_a=0.5)
    got = C[torch.from_numpy(rows.astype(np.int64)).cuda()].cpu().numpy()
    util.assert_parity(got, C_ref, 0.5 * _absprod_rows(sub_rp, inv.astype(np.int32), sub_v, B_sub), np.float32,
                       row_len=np.diff(sub_rp), what="cfg3 R-MAT heavy + sampled rows")
    # the plan-free path gives the same answer to rounding
    C2 = torch.full((m, 128), float("nan"), device="cuda")
    sp.multiply(sp.scaled(0.5, a), B, C2)
    # (both sum positive terms; a k-entry row summed in sequence carries up to ~k*eps/2 of rounding, the split rows less)
    tol = torch.clamp(lens.float() * 6e-8, min=2e-5)[:, None]
    assert bool(((C - C2).abs() <= tol * C.abs() + 1e-30).all())


# --------------------------------------------------------------------------------------------- cfg4
@pytest.mark.parametrize("alg", ["auto", "rowblock", "sliced"])
def test_cfg4_spmv_rmat_scale24_f64(gpu, alg):
    """BASELINE cfg4 (single-GPU leg; the row-sharded leg is bench.py --gpus N --workload spmv_rmat and
    tests/test_sharded_cpu.py): fp64 CSR SpMV on an R-MAT graph of 2^24 vertices, edge factor 16, 268 M
    entries, duplicates kept.  Parity at 1e-12 norm-wise (BASELINE north_star) on the heaviest and on sampled
    rows; fp64 checksum and linearity over all 16.8 M rows."""
    values, rowptr, colind, shape, nnz = generate.rmat_csr_device(24, 16, dtype=torch.float64, seed=0)
    m = n = shape[0]
    assert nnz == 16 << 24
    a = sp.csr_view(values, rowptr, colind, shape, nnz)
    g = torch.Generator(device="cuda").manual_seed(5)
    x1 = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    x2 = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    y1, y2, y3 = (torch.full((m,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(3))
    if alg == "auto":
        # power-law rows and hot columns: no static rule -- AUTO builds the sliced plan (variable-height bins) and
        # keeps whichever of it and the row-block kernel was faster in its timed trial
        info = sp.multiply_inspect(sp.matrix_opt(a), x1, y1)
        pi = info.state_.info()
        assert pi["alg"] in (_capi.SPMV_ROWBLOCK, _capi.SPMV_SLICED) and pi["n_long_rows"] > 1000
        if pi["alg"] == _capi.SPMV_SLICED:
            assert info.state_.sliced_info()["variable_bins"] == 1
    elif alg == "rowblock":
        info = sp.multiply_inspect(a, x1, y1, alg=_capi.SPMV_ROWBLOCK)
        assert info.state_.info()["alg"] == _capi.SPMV_ROWBLOCK and info.state_.info()["n_long_rows"] > 1000
    else:
        info = sp.multiply_inspect(a, x1, y1, alg=_capi.SPMV_SLICED)
        assert info.state_.info()["alg"] == _capi.SPMV_SLICED
    sp.multiply(info, a, x1, y1)
    sp.multiply(info, a, x2, y2)
    sp.multiply(info, a, 0.5 * x1 - 2.0 * x2, y3)
    assert bool(torch.isfinite(y1).all())
    # linearity, norm-wise (the three products are sums of the same non-negative terms)
    lin = (y3 - (0.5 * y1 - 2.0 * y2)).abs()
    lens = (rowptr[1:].long() - rowptr[:-1].long()).double()
    tol = torch.clamp(lens * 2.3e-16, min=4e-12)
    assert bool((lin <= tol * (0.5 * y1.abs() + 2.0 * y2.abs()) + 1e-300).all()), "cfg4 linearity"
    # checksum: sum_i y_i == sum_p v_p x_{c_p}
    lhs = y1.sum().item()
    rhs = (values * x1[colind.long()]).sum().item()
    assert abs(lhs - rhs) <= 1e-11 * abs(rhs), "cfg4 checksum"
    # oracle: the 2 000 heaviest rows (hubs: up to ~3.7e5 entries) and 4 000 sampled rows
    heavy = torch.topk(lens, 2000).indices.cpu().numpy()
    sample = np.random.default_rng(0).integers(0, m, 4000)
    x_h, y_h = x1.cpu().numpy(), y1.cpu().numpy()
    for rows, what in ((np.unique(heavy), "heaviest rows"), (np.unique(sample), "sampled rows")):
        sub_rp, sub_c, sub_v = _rows_subproblem(rows, rowptr, colind, values)
        y_ref = oracle.spmv((len(rows), n), sub_rp, sub_c, sub_v, x_h)
        absrow = oracle.spmv_absrow(sub_rp, sub_c, sub_v, x_h)
        util.assert_parity(y_h[rows], y_ref, absrow, np.float64, row_len=np.diff(sub_rp), what=f"cfg4 {alg} {what}")


def test_cfg4_matrix_transposed_without_a_plan_f64(gpu):
    """cfg4's matrix as a csc_view operand, NOT inspected: y = A^T x on the R-MAT graph of 2^24 vertices goes through the
    two-pass form of csrc/spmv.hip (t2_* kernels) in its widest configuration -- 1 024 column slices of 16 384 columns (the
    LDS copy of a slice takes a whole CU), hot columns cut into segments that add into y.  Every element of y against
    oracle_spmv_csc (backend/algorithms.hpp:21-29 + multiply_impl.hpp:33-53) at 1e-12 norm-wise, alpha folded in, and the
    fp64 checksum sum(y) == sum_p v_p x[row_p]."""
    values, rowptr, colind, shape, nnz = generate.rmat_csr_device(24, 16, dtype=torch.float64, seed=0)
    m, n = shape
    a = sp.csr_view(values, rowptr, colind, shape, nnz)
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.rand(m, dtype=torch.float64, device="cuda", generator=g) - 0.25
    y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    sp.multiply(sp.scaled(-1.5, sp.transposed(a)), x, y)
    assert bool(torch.isfinite(y).all())
    lens = (rowptr[1:].long() - rowptr[:-1].long())
    rows_of = torch.repeat_interleave(torch.arange(m, device="cuda"), lens)
    rhs = (-1.5 * (values * x[rows_of]).sum()).item()
    mag = (1.5 * (values.abs() * x[rows_of].abs()).sum()).item()
    del rows_of
    assert abs(y.sum().item() - rhs) <= 1e-11 * mag, "checksum of the transposed product"
    rp_h, ci_h, v_h, x_h = rowptr.cpu().numpy().astype(np.int32), colind.cpu().numpy(), values.cpu().numpy(), x.cpu().numpy()
    ref = -1.5 * oracle.spmv_csc((n, m), rp_h, ci_h, v_h, x_h)
    ab = 1.5 * oracle.spmv_csc((n, m), rp_h, ci_h, np.abs(v_h), np.abs(x_h))
    cnt = np.bincount(ci_h, minlength=n) + 1
    util.assert_parity(y.cpu().numpy(), ref, ab, np.float64, row_len=cnt, what="cfg4 matrix, y = A^T x without a plan")


# --------------------------------------------------------------------------------------------- cfg5
def test_cfg5_spgemm_full_size(gpu):
    """BASELINE cfg5 at its stated size: fp32 CSR x CSR, 1M x 1M, 16 nnz/row each, multiply_compute +
    multiply_fill (examples/simple_spgemm.cpp:52-60 call shape).  nnz(C) ~ 2.56e8."""
    m = 1_000_000
    av, ar, ac, ash, annz = generate.uniform_csr_device(m, m, 16, seed=0)
    bv, br, bc, bsh, bnnz = generate.uniform_csr_device(m, m, 16, seed=1)
    d_a, d_b = sp.csr_view(av, ar, ac, ash, annz), sp.csr_view(bv, br, bc, bsh, bnnz)
    d_rp = torch.full((m + 1,), -1, dtype=torch.int32, device="cuda")
    d_c = sp.csr_view(None, d_rp, None, (m, m), 0)
    info = sp.multiply_compute(d_a, d_b, d_c)
    cn = info.result_nnz()
    assert tuple(info.result_shape()) == (m, m)
    # structural nnz and every row offset EXACT against the oracle's full symbolic pass (host, a few seconds)
    a_h = (ar.cpu().numpy(), ac.cpu().numpy())
    b_h = (br.cpu().numpy(), bc.cpu().numpy())
    n_ref, row_nnz = oracle.spgemm_symbolic((m, m), a_h[0], a_h[1], (m, m), b_h[0], b_h[1])
    assert cn == n_ref, f"nnz(C) {cn} != oracle {n_ref}"
    assert np.array_equal(d_rp.cpu().numpy().astype(np.int64), np.concatenate([[0], np.cumsum(row_nnz)]))
    d_c.update(torch.full((cn,), float("nan"), device="cuda"), d_rp,
               torch.full((cn,), -1, dtype=torch.int32, device="cuda"), (m, m), cn)
    sp.multiply_fill(info, d_a, d_b, d_c)
    assert d_c.size() == cn
    rp = d_rp.long()
    cols = d_c.colind().long()
    assert int(cols.min()) >= 0 and int(cols.max()) < m and bool(torch.isfinite(d_c.values()).all())
    # columns strictly ascending inside every row (spgemm_gustavsons.hpp:42 sorts them)
    same_row = torch.ones(cn - 1, dtype=torch.bool, device="cuda")
    ends = rp[1:-1] - 1
    same_row[ends[(ends >= 0) & (ends < cn - 1)]] = False
    assert bool(((cols[1:] > cols[:-1]) | ~same_row).all())
    del same_row, cols
    # (AB)x == A(Bx)
    x = torch.rand(m, device="cuda")
    t, y1, y2 = (torch.empty(m, device="cuda") for _ in range(3))
    sp.multiply(d_b, x, t)
    sp.multiply(d_a, t, y1)
    sp.multiply(d_c, x, y2)
    assert bool(((y1 - y2).abs() <= 2e-5 * y1.abs() + 1e-30).all())
    # sampled rows: indices exact, values to the parity bound
    rows = np.unique(np.concatenate([np.arange(0, m, 5003), [m - 1]]))
    sub_rp, sub_c, sub_v = _rows_subproblem(rows, ar, ac, av)
    bv_h = bv.cpu().numpy()
    n_sub, _ = oracle.spgemm_symbolic((len(rows), m), sub_rp, sub_c, (m, m), b_h[0], b_h[1])
    cr, cc, cv = oracle.spgemm_numeric((len(rows), m), sub_rp, sub_c, sub_v, (m, m), b_h[0], b_h[1], bv_h,
                                       capacity=n_sub)
    got_rp, got_c, got_v = _rows_subproblem(rows, d_rp, d_c.colind(), d_c.values())
    assert np.array_equal(got_rp, cr) and np.array_equal(got_c, cc)
    np.testing.assert_allclose(got_v, cv, rtol=2e-5)
    # numeric reuse (multiply_numeric after the first fill, vendor/rocsparse/multiply_spgemm.hpp:178-214): the second
    # pass accumulates by the product ranks recorded during the first one -- new values, fresh output arrays, same
    # structure; indices identical, values = 6 x the first result (A scaled by 2, B by 3: exact in binary)
    first_vals = d_c.values().clone()
    first_cols = d_c.colind().clone()
    av.mul_(2.0)
    bv.mul_(3.0)
    d_c.update(torch.full((cn,), float("nan"), device="cuda"), d_rp,
               torch.full((cn,), -1, dtype=torch.int32, device="cuda"), (m, m), cn)
    for attempt in range(3):  # 2nd pass: hash + recording; 3rd and 4th: by rank (4th into the array the 3rd filled)
        if attempt < 2:
            d_c.update(torch.full((cn,), float("nan"), device="cuda"), d_rp,
                       torch.full((cn,), -1, dtype=torch.int32, device="cuda"), (m, m), cn)
        else:
            d_c.values().fill_(float("nan"))
        sp.multiply_fill(info, d_a, d_b, d_c)
        assert torch.equal(d_c.colind(), first_cols), attempt
        assert bool(((d_c.values() - 6.0 * first_vals).abs() <= 1e-5 * (6.0 * first_vals).abs() + 1e-30).all()), attempt


def test_more_than_2_31_entries_need_64_bit_offsets(gpu):
    """The maximum-size edge of the CSR boundary: nnz > 2^31 - 1, which only 64-bit row offsets can describe (the
    reference's csr_view is templated on the offset type, /root/reference/include/spblas/views/csr_view.hpp; its vendor
    back ends pass 64-bit offsets straight through, vendor/rocsparse/spmv_impl.hpp).  33.5 M rows of 65 entries =
    2.18e9 entries (17.4 GB of values + columns -- nothing for a 288 GB device): SpMV plan-free and through every plan the
    inspect accepts, SpMM with two columns, all against a float64 evaluation on the device in row chunks (too large
    for the host oracle, which has pinned the same kernels on every smaller case).  The 32-bit position arrays of the
    SLICED re-tiling cannot hold this matrix: a forced SLICED inspect must refuse or fall back, never wrap around."""
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * 2 ** 30:
        pytest.skip("needs 60 GB of device memory")
    m, L, n = 1 << 25, 65, 1 << 20
    nnz = m * L
    assert nnz > 2 ** 31
    g = torch.Generator(device="cuda").manual_seed(7)
    colind = torch.empty(nnz, dtype=torch.int32, device="cuda")
    values = torch.empty(nnz, dtype=torch.float32, device="cuda")
    step = 1 << 28
    for o in range(0, nnz, step):  # chunked: the generators index with 32 bits in places
        e = min(nnz, o + step)
        colind[o:e] = torch.randint(0, n, (e - o,), dtype=torch.int32, device="cuda", generator=g)
        values[o:e] = torch.rand(e - o, device="cuda", generator=g) + 0.5
    # distinct tails: the last rows are where a wrapped 32-bit offset would read the wrong entries
    rowptr = torch.arange(m + 1, dtype=torch.int64, device="cuda") * L
    x = torch.rand(n, device="cuda", generator=g) + 0.5
    B = torch.rand((n, 2), device="cuda", generator=g) + 0.5

    def reference(rhs):  # float64, row chunks of 2^20 rows, one right-hand side at a time (a (rows, L, k) sum over
        # dim 1 came back with zeros from torch on this image; the (rows, L) form does not)
        cols = [rhs.double()] if rhs.dim() == 1 else [rhs[:, c].double().contiguous() for c in range(rhs.shape[1])]
        out = torch.empty((len(cols), m), dtype=torch.float64, device="cuda")
        rows = 1 << 20
        for r0 in range(0, m, rows):
            sl = slice(r0 * L, (r0 + rows) * L)
            idx, v = colind[sl].long(), values[sl].double()
            for c, rd in enumerate(cols):
                out[c, r0:r0 + rows] = (rd[idx] * v).view(rows, L).sum(1)
        assert bool((out > 0).all())
        return out[0] if rhs.dim() == 1 else out.t().contiguous()

    a = sp.csr_view(values, rowptr, colind, (m, n), nnz)
    y_ref = reference(x)  # all terms positive: the norm-wise bound is relative to the result
    tol = max(1e-6, L * np.finfo(np.float32).eps)

    def check_y(y, what):
        err = ((y.double() - y_ref).abs() / y_ref).max().item()
        assert err <= tol, f"{what}: max relative error {err}"
        # the rows past entry 2^31: a wrapped offset would have read other entries
        tail = slice((2 ** 31) // L + 1, m)
        assert ((y.double()[tail] - y_ref[tail]).abs() / y_ref[tail]).max().item() <= tol

    y = torch.full((m,), float("nan"), device="cuda")
    sp.multiply(a, x, y)
    check_y(y, "plan-free")
    for name, alg in (("auto", _capi.SPMV_AUTO), ("vector", _capi.SPMV_VECTOR), ("rowblock", _capi.SPMV_ROWBLOCK)):
        y.fill_(float("nan"))
        info = sp.multiply_inspect(a, x, y, alg=alg)
        assert info.state_.info()["alg"] != _capi.SPMV_SLICED
        sp.multiply(info, a, x, y)
        check_y(y, name)
        del info
    y.fill_(float("nan"))
    try:
        info = sp.multiply_inspect(a, x, y, alg=_capi.SPMV_SLICED)
    except Exception:
        info = None  # refused: fine
    if info is not None:
        assert info.state_.info()["alg"] != _capi.SPMV_SLICED, "32-bit positions cannot describe 2^31 entries"
        sp.multiply(info, a, x, y)
        check_y(y, "sliced -> fallback")
        del info

    C = torch.full((m, 2), float("nan"), device="cuda")
    sp.multiply(a, B, C)
    C_ref = reference(B)
    errC = ((C.double() - C_ref).abs() / C_ref).max().item()
    assert errC <= tol, f"SpMM: max relative error {errC}"
    infoC = sp.multiply_inspect(a, B, C)
    C.fill_(float("nan"))
    sp.multiply(infoC, a, B, C)
    errC = ((C.double() - C_ref).abs() / C_ref).max().item()
    assert errC <= tol, f"SpMM with inspect: max relative error {errC}"
