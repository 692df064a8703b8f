"""-m gpu parity tests for CSR x CSR SpGEMM vs the CPU oracle: indices EXACT (rowptr,
sorted colind, result_nnz, result_shape), values within the parity bound.
Cases mirror /root/reference/test/gtest/device/spgemm_test.cpp:12-97 (compute -> allocate ->
update -> fill, n in {m,k}), :99- (_AScaled/_BScaled), device/spgemm_reuse_test.cpp:12-114
(symbolic/numeric reuse with changed values) and test/gtest/spgemm_test.cpp:10-71."""
import os

import numpy as np
import pytest
import torch

import gpu_util as G
import spblas_reference_amd as sp
import util
from oracle import oracle
from spblas_reference_amd import generate

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def absprod_rows(a, b, c_rowptr, c_colind):
    """sum |a_v*b_v| per output entry, as the oracle orders C."""
    (av, ar, ac, ash), (bv, br, bc, bsh) = a, b
    cr, cc, cv = oracle.spgemm_numeric(ash, ar, ac, np.abs(av), bsh, br, bc, np.abs(bv), capacity=len(c_colind))
    assert np.array_equal(cr, c_rowptr) and np.array_equal(cc, c_colind)
    return cv.astype(np.float64)


def device_spgemm(a_h, b_h, use_state, scale_a=None, scale_b=None):
    (av, ar, ac, ash), (bv, br, bc, bsh) = a_h, b_h
    m, n = ash[0], bsh[1]
    d_a = G.csr_on_device(av, ar, ac, ash, len(av))
    d_b = G.csr_on_device(bv, br, bc, bsh, len(bv))
    A = sp.scaled(scale_a, d_a) if scale_a is not None else d_a
    B = sp.scaled(scale_b, d_b) if scale_b is not None else d_b
    d_c_rowptr = torch.full((m + 1,), -1, dtype=torch.int32, device="cuda")
    d_c = sp.csr_view(None, d_c_rowptr, None, (m, n), 0)       # device/spgemm_test.cpp:37-40
    if use_state:
        state = sp.spgemm_state_t()
        sp.multiply_compute(state, A, B, d_c)                   # :42-43
        nnz = state.result_nnz()
        assert state.result_shape() == (m, n)
    else:
        info = sp.multiply_compute(A, B, d_c)                   # examples/simple_spgemm.cpp:52
        nnz = info.result_nnz()
        assert info.result_shape() == (m, n)
    d_vals = torch.full((nnz,), float("nan"), dtype=G.dev(av).dtype, device="cuda")
    d_cols = torch.full((nnz,), -1, dtype=torch.int32, device="cuda")
    d_c.update(d_vals, d_c_rowptr, d_cols, (m, n), nnz)         # :49-50
    if use_state:
        sp.multiply_fill(state, A, B, d_c)                      # :52
    else:
        sp.multiply_fill(info, A, B, d_c)
    assert d_c.size() == nnz
    return nnz, G.host(d_c_rowptr), G.host(d_cols), G.host(d_vals)


def check_against_oracle(a_h, b_h, got, dtype, scale=1.0):
    (av, ar, ac, ash), (bv, br, bc, bsh) = a_h, b_h
    nnz, c_rowptr, c_colind, c_values = got
    ref_nnz, _ = oracle.spgemm_symbolic(ash, ar, ac, bsh, br, bc)
    assert nnz == ref_nnz                                        # result_nnz exact
    cr, cc, cv = oracle.spgemm_numeric(ash, ar, ac, av, bsh, br, bc, bv, capacity=ref_nnz,
                                       scale_a=None if scale == 1.0 else scale)
    assert np.array_equal(c_rowptr, cr)                          # indices exact
    assert np.array_equal(c_colind, cc)                          # sorted ascending per row, exact
    ab = absprod_rows(a_h, b_h, cr, cc) * abs(scale)
    util.assert_parity(c_values, cv, ab, dtype, row_len=np.full(len(cv), 64), what="spgemm values")
    util.expect_eq_ref(cv, c_values)


@pytest.mark.parametrize("use_state", [True, False])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("dim", util.dims)
def test_spgemm_reference_device_test(gpu, dim, dtype, use_state):
    m, k, nnz = dim
    for n in (m, k):
        a_h = generate.generate_csr(m, k, nnz, dtype=dtype)[:4]
        b_h = generate.generate_csr(k, n, nnz, seed=1, dtype=dtype)[:4]
        check_against_oracle(a_h, b_h, device_spgemm(a_h, b_h, use_state), dtype)


def test_spgemm_scaled(gpu):
    # device/spgemm_test.cpp:99-: alpha = 2 on A, then on B
    a_h = generate.generate_csr(100, 1000, 10000)[:4]
    b_h = generate.generate_csr(1000, 100, 10000, seed=1)[:4]
    check_against_oracle(a_h, b_h, device_spgemm(a_h, b_h, True, scale_a=2.0), np.float32, scale=2.0)
    check_against_oracle(a_h, b_h, device_spgemm(a_h, b_h, True, scale_b=2.0), np.float32, scale=2.0)


def test_spgemm_golden_bit_exact(gpu):
    g = np.load(os.path.join(GOLDEN, "spgemm_dups_cancel.npz"))
    a_h = (g["a_values"], g["a_rowptr"], g["a_colind"], tuple(g["a_shape"]))
    b_h = (g["b_values"], g["b_rowptr"], g["b_colind"], tuple(g["b_shape"]))
    nnz, cr, cc, cv = device_spgemm(a_h, b_h, True)
    assert nnz == int(g["c_nnz"])  # structural count: cancelled entries are stored as zeros
    assert np.array_equal(cr, g["c_rowptr"]) and np.array_equal(cc, g["c_colind"]) and np.array_equal(cv, g["c_values"])


def test_spgemm_all_bins_including_dense_rows(gpu):
    """Rows whose product counts land in every accumulator bin (<=64, <=512, <=4096, >4096)."""
    rng = np.random.default_rng(3)
    m, k, n = 300, 600, 20000
    a_lens = rng.integers(0, 6, m)
    a_lens[5] = 40   # x B rows of ~30 -> bin 3
    a_lens[6] = 300  # -> bin 4 (dense accumulator)
    a_lens[7] = 0
    ar = np.concatenate([[0], np.cumsum(a_lens)]).astype(np.int32)
    ac = np.concatenate([rng.choice(k, L, replace=False) for L in a_lens]).astype(np.int32)
    av = (rng.random(len(ac)) + 0.5).astype(np.float32)  # positive like the reference's U[0,100) data
    b_lens = rng.integers(0, 60, k)
    b_lens[ac[ar[6]]] = 5000
    br = np.concatenate([[0], np.cumsum(b_lens)]).astype(np.int32)
    bc = np.concatenate([rng.choice(n, L, replace=False) for L in b_lens]).astype(np.int32)
    bv = (rng.random(len(bc)) + 0.5).astype(np.float32)
    a_h, b_h = (av, ar, ac, (m, k)), (bv, br, bc, (k, n))
    check_against_oracle(a_h, b_h, device_spgemm(a_h, b_h, True), np.float32)


@pytest.mark.parametrize("record_at", ["second", "first"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spgemm_repeated_fills_every_bin_by_rank(gpu, monkeypatch, dtype, record_at):
    """Five numeric passes on one symbolic result whose rows sit in every accumulator bin: <= 64 and <= 256 products
    (one-byte ranks), 257 .. 1024 (two-byte ranks, e.g. a 27-point stencil times itself), and the larger hash / dense
    bins that keep the hash kernels.  New values and a new factor every pass, columns poisoned before each."""
    if record_at == "first":
        monkeypatch.setenv("SPBLAS_GFX950_SPGEMM_REUSE", "2")
    rng = np.random.default_rng(21)
    m, k, n = 400, 700, 20000
    a_lens = rng.integers(0, 6, m)
    a_lens[10:90] = rng.integers(12, 30, 80)     # x B rows of ~30: 360 .. 900 products -> bin 3
    a_lens[5] = 60                               # ~1800 products -> bin 4
    a_lens[6] = 300                              # dense accumulator
    a_lens[7] = 0
    ar = np.concatenate([[0], np.cumsum(a_lens)]).astype(np.int32)
    ac = np.concatenate([rng.choice(k, L, replace=False) for L in a_lens]).astype(np.int32)
    b_lens = rng.integers(20, 40, k)
    b_lens[ac[ar[6]]] = 5000
    br = np.concatenate([[0], np.cumsum(b_lens)]).astype(np.int32)
    bc = np.concatenate([rng.choice(n, L, replace=False) for L in b_lens]).astype(np.int32)
    bc[br[3] + 1] = bc[br[3]]                    # a repeated column inside a B row
    d_a = G.csr_on_device(np.zeros(len(ac), dtype), ar, ac, (m, k), len(ac))
    d_b = G.csr_on_device(np.zeros(len(bc), dtype), br, bc, (k, n), len(bc))
    d_rp = torch.full((m + 1,), -1, dtype=torch.int32, device="cuda")
    d_c = sp.csr_view(None, d_rp, None, (m, n), 0)
    state = sp.spgemm_state_t()
    sp.multiply_compute(state, d_a, d_b, d_c)
    cn = state.result_nnz()
    d_vals = torch.full((cn,), float("nan"), dtype=G.dev(np.zeros(1, dtype)).dtype, device="cuda")
    d_cols = torch.full((cn,), -1, dtype=torch.int32, device="cuda")
    d_c.update(d_vals, d_rp, d_cols, (m, n), cn)
    for it, scale in enumerate([None, 2.0, None, -0.5, None]):
        av = (rng.random(len(ac)) + 0.5).astype(dtype)
        bv = (rng.random(len(bc)) + 0.5).astype(dtype)
        d_a.values().copy_(G.dev(av))
        d_b.values().copy_(G.dev(bv))
        d_vals.fill_(float("nan"))
        d_cols.fill_(-1)
        sp.multiply_numeric(state, sp.scaled(scale, d_a) if scale is not None else d_a, d_b, d_c)
        got = (cn, G.host(d_rp), G.host(d_cols), G.host(d_vals))
        check_against_oracle((av, ar, ac, (m, k)), (bv, br, bc, (k, n)), got, dtype, scale or 1.0)


def test_spgemm_reuse_changed_values_and_pointers(gpu):
    # device/spgemm_reuse_test.cpp:12-114,326-399: symbolic once, numeric 3x with new values,
    # then with re-allocated (different pointer) value arrays
    m, k, nnz = 100, 1000, 10000
    a_h = list(generate.generate_csr(m, k, nnz)[:4])
    b_h = list(generate.generate_csr(k, m, nnz, seed=1)[:4])
    d_a = G.csr_on_device(*a_h, nnz)
    d_b = G.csr_on_device(*b_h, nnz)
    d_rp = torch.zeros(m + 1, dtype=torch.int32, device="cuda")
    d_c = sp.csr_view(None, d_rp, None, (m, m), 0)
    state = sp.spgemm_state_t()
    sp.multiply_symbolic_compute(state, d_a, d_b, d_c)
    cn = state.result_nnz()
    d_c.update(torch.zeros(cn, device="cuda"), d_rp, torch.zeros(cn, dtype=torch.int32, device="cuda"), (m, m), cn)
    sp.multiply_symbolic_fill(state, d_a, d_b, d_c)
    rng = np.random.default_rng(0)
    for it in range(4):
        if it:
            a_h[0] = (rng.random(nnz) * 100).astype(np.float32)
            b_h[0] = (rng.random(nnz) * 100).astype(np.float32)
            if it == 3:  # new allocations: pointers change, pattern does not
                d_a = G.csr_on_device(*a_h, nnz)
                d_b = G.csr_on_device(*b_h, nnz)
                new_rp = torch.zeros(m + 1, dtype=torch.int32, device="cuda")
                d_c.update(torch.zeros(cn, device="cuda"), new_rp, torch.zeros(cn, dtype=torch.int32, device="cuda"),
                           (m, m), cn)
            else:
                d_a.values().copy_(G.dev(a_h[0]))
                d_b.values().copy_(G.dev(b_h[0]))
        sp.multiply_numeric(state, d_a, d_b, d_c)
        got = (cn, G.host(d_c.rowptr()), G.host(d_c.colind()), G.host(d_c.values()))
        check_against_oracle(tuple(a_h), tuple(b_h), got, np.float32)


def test_spgemm_numeric_into_recycled_addresses(gpu):
    """device/spgemm_reuse_test.cpp:325-448 (SpGEMMReuseAndChangePointer): every numeric pass gets fresh copies of all
    nine arrays, and the allocator hands the addresses of the previous iteration's copies out again.  An equal
    c_colind address therefore proves nothing about its contents: numeric must (re)write the column indices unless
    it is provably the array the previous pass filled.  Six passes, so that the rank-reuse path (third pass on) is
    exercised; pass 4 poisons the recycled column array instead of copying the structure into it, pass 5 passes the
    SAME tensors again (the one case where the columns may stay untouched)."""
    m, k, nnz = 1000, 100, 10000
    a_h = list(generate.generate_csr(m, k, nnz)[:4])
    b_h = list(generate.generate_csr(k, m, nnz, seed=1)[:4])
    d_a = G.csr_on_device(*a_h, nnz)
    d_b = G.csr_on_device(*b_h, nnz)
    d_rp = torch.zeros(m + 1, dtype=torch.int32, device="cuda")
    d_c = sp.csr_view(None, d_rp, None, (m, m), 0)
    state = sp.spgemm_state_t()
    sp.multiply_symbolic_compute(state, d_a, d_b, d_c)
    cn = state.result_nnz()
    c_val, c_col = torch.zeros(cn, device="cuda"), torch.full((cn,), -7, dtype=torch.int32, device="cuda")
    d_c.update(c_val, d_rp, c_col, (m, m), cn)
    sp.multiply_symbolic_fill(state, d_a, d_b, d_c)
    # the symbolic stage leaves the structure in the caller's arrays (multiply_spgemm.hpp:147-176)
    ref_nnz, ref_rp, ref_ci, _ = util_spgemm_reference(a_h, b_h)
    assert np.array_equal(G.host(d_rp), ref_rp) and np.array_equal(G.host(c_col), ref_ci)
    rng = np.random.default_rng(0)
    seen = set()
    cur = None
    for it in range(6):
        a_h[0] = (rng.random(nnz) * 100).astype(np.float32)
        b_h[0] = (rng.random(nnz) * 100).astype(np.float32)
        if it != 5:
            cur = None                                          # drop the previous copies: their addresses are free again
            na = G.csr_on_device(*a_h, nnz)
            nb = G.csr_on_device(*b_h, nnz)
            n_col = torch.full((cn,), 123456, dtype=torch.int32, device="cuda") if it == 4 else c_col.clone()
            n_val, n_rp = torch.full((cn,), float("nan"), device="cuda"), d_rp.clone()
            nc = sp.csr_view(n_val, n_rp, n_col, (m, m), cn)
            cur = (na, nb, nc)
        else:
            cur[0].values().copy_(G.dev(a_h[0]))
            cur[1].values().copy_(G.dev(b_h[0]))
        seen.add(cur[2].colind().data_ptr())
        sp.multiply_numeric(state, cur[0], cur[1], cur[2])
        got = (cn, G.host(cur[2].rowptr()), G.host(cur[2].colind()), G.host(cur[2].values()))
        check_against_oracle(tuple(a_h), tuple(b_h), got, np.float32)
    assert len(seen) < 5, "the allocator was expected to recycle at least one address (the scenario under test)"


def util_spgemm_reference(a_h, b_h):
    """(nnz, rowptr, colind, values) of A * B from the oracle."""
    (av, ar, ac, ash), (bv, br, bc, bsh) = a_h, b_h
    n_ref, _ = oracle.spgemm_symbolic(ash, ar, ac, bsh, br, bc)
    cr, cc, cv = oracle.spgemm_numeric(ash, ar, ac, av, bsh, br, bc, bv, capacity=n_ref)
    return n_ref, cr, cc, cv


def test_spgemm_errors(gpu):
    a_h = generate.generate_csr(40, 40, 1000)[:4]
    d_a = G.csr_on_device(*a_h, 1000)
    d_rp = torch.zeros(41, dtype=torch.int32, device="cuda")
    d_c = sp.csr_view(None, d_rp, None, (40, 40), 0)
    info = sp.multiply_compute(d_a, d_a, d_c)
    cn = info.result_nnz()
    d_c.update(torch.zeros(cn - 1, device="cuda"), d_rp, torch.zeros(cn - 1, dtype=torch.int32, device="cuda"),
               (40, 40), cn - 1)
    with pytest.raises(RuntimeError, match="out of memory"):  # spgemm_gustavsons.hpp:44-48
        sp.multiply_fill(info, d_a, d_a, d_c)
    with pytest.raises(ValueError):  # shape mismatch :22-27
        sp.multiply_compute(d_a, d_a, sp.csr_view(None, d_rp, None, (40, 39), 0))


def test_spgemm_cfg5_shape_properties(gpu):
    """BASELINE cfg5 shape at 1/10 rows (100k x 100k, 16 nnz/row): structural nnz and sorted
    columns checked on device, sampled rows against the oracle, and (AB)x == A(Bx)."""
    m = 100_000
    av, ar, ac, ash, annz = generate.uniform_csr_device(m, m, 16, seed=0)
    bv, br, bc, bsh, bnnz = generate.uniform_csr_device(m, m, 16, seed=1)
    d_a, d_b = sp.csr_view(av, ar, ac, ash, annz), sp.csr_view(bv, br, bc, bsh, bnnz)
    d_rp = torch.zeros(m + 1, dtype=torch.int32, device="cuda")
    d_c = sp.csr_view(None, d_rp, None, (m, m), 0)
    state = sp.spgemm_state_t()
    sp.multiply_compute(state, d_a, d_b, d_c)
    cn = state.result_nnz()
    d_c.update(torch.empty(cn, device="cuda"), d_rp, torch.empty(cn, dtype=torch.int32, device="cuda"), (m, m), cn)
    sp.multiply_fill(state, d_a, d_b, d_c)
    rp = d_rp.long()
    assert int(rp[-1]) == cn and bool((rp[1:] >= rp[:-1]).all())
    # columns strictly ascending inside every row
    cols = d_c.colind().long()
    same_row = torch.ones(cn - 1, dtype=torch.bool, device="cuda")
    same_row[(rp[1:-1] - 1).clamp(min=0, max=cn - 2)] = False
    assert bool(((cols[1:] > cols[:-1]) | ~same_row).all())
    # (AB)x == A(Bx)
    x = torch.rand(m, device="cuda")
    t, y1, y2 = (torch.empty(m, device="cuda") for _ in range(3))
    sp.multiply(d_b, x, t)
    sp.multiply(d_a, t, y1)
    sp.multiply(d_c, x, y2)
    assert bool(((y1 - y2).abs() <= 1e-5 * y1.abs() + 1e-30).all())
    # sampled rows vs the oracle (exact indices)
    rows = np.arange(0, m, 5003)
    a_rp = ar.cpu().numpy()
    idx = np.concatenate([np.arange(a_rp[r], a_rp[r + 1]) for r in rows])
    sub = (av.cpu().numpy()[idx], np.concatenate([[0], np.cumsum(a_rp[rows + 1] - a_rp[rows])]).astype(np.int32),
           ac.cpu().numpy()[idx], (len(rows), m))
    b_h = (bv.cpu().numpy(), br.cpu().numpy(), bc.cpu().numpy(), (m, m))
    n_ref, _ = oracle.spgemm_symbolic(sub[3], sub[1], sub[2], b_h[3], b_h[1], b_h[2])
    cr, cc, cv = oracle.spgemm_numeric(sub[3], sub[1], sub[2], sub[0], b_h[3], b_h[1], b_h[2], b_h[0], capacity=n_ref)
    rp_h = d_rp.cpu().numpy()
    got_c = np.concatenate([d_c.colind().cpu().numpy()[rp_h[r]:rp_h[r + 1]] for r in rows])
    got_v = np.concatenate([d_c.values().cpu().numpy()[rp_h[r]:rp_h[r + 1]] for r in rows])
    assert np.array_equal(got_c, cc) and np.array_equal(np.diff(cr), rp_h[rows + 1] - rp_h[rows])
    np.testing.assert_allclose(got_v, cv, rtol=1e-5)


def _host_transpose(v, rp, ci, shape):
    """CSR arrays of X^T on the host (scipy is test infrastructure only)."""
    import scipy.sparse as sps
    t = sps.csr_matrix((v, ci, rp), shape=shape).T.tocsr()
    t.sort_indices()
    return t.data.astype(v.dtype), t.indptr.astype(np.int32), t.indices.astype(np.int32), (shape[1], shape[0])


@pytest.mark.parametrize("c_fmt", ["csr", "csc"])
@pytest.mark.parametrize("b_fmt", ["csr", "csc"])
@pytest.mark.parametrize("a_fmt", ["csr", "csc"])
def test_spgemm_mixed_csr_csc_operands(gpu, a_fmt, b_fmt, c_fmt):
    """The eight operand/result format combinations of the reference's CPU path
    (test/gtest/spgemm_csr_csc.cpp:10-350 MixedViews.*, spgemm_test.cpp:203-264 CscView.SpGEMM): a CSC
    operand is built from the transposed CSR arrays exactly as those tests do; the result must equal the
    oracle's CSR product (indices exact), read back through the requested format."""
    dtype = np.float32
    for (m, k, nnz) in [(100, 100, 100), (40, 1000, 1000), (1000, 100, 10000)]:
        n = m
        a_h = generate.generate_csr(m, k, nnz, dtype=dtype)[:4]
        b_h = generate.generate_csr(k, n, nnz, seed=1, dtype=dtype)[:4]

        def operand(h, fmt):
            v, rp, ci, sh = h
            if fmt == "csr":
                return G.csr_on_device(v, rp, ci, sh, len(v))
            tv, trp, tci, _ = _host_transpose(v, rp, ci, sh)     # CSR of X^T == CSC of X
            return sp.csc_view(G.dev(tv), G.dev(trp), G.dev(tci), sh, len(v))

        A, B = operand(a_h, a_fmt), operand(b_h, b_fmt)
        view = sp.csr_view if c_fmt == "csr" else sp.csc_view
        ptr = torch.full(((m if c_fmt == "csr" else n) + 1,), -1, dtype=torch.int32, device="cuda")
        C = view(None, ptr, None, (m, n), 0)
        info = sp.multiply_compute(A, B, C)
        cn = info.result_nnz()
        assert info.result_shape() == (m, n)
        vals = torch.full((cn,), float("nan"), device="cuda")
        inds = torch.full((cn,), -1, dtype=torch.int32, device="cuda")
        C.update(vals, ptr, inds, (m, n), cn)
        sp.multiply_fill(info, sp.scaled(2.0, A), B, C)
        assert C.size() == cn and tuple(C.shape()) == (m, n)
        got_v, got_p, got_i = G.host(vals), G.host(ptr), G.host(inds)
        # oracle: CSR product, transposed on the host when the result is CSC
        ref_nnz, _ = oracle.spgemm_symbolic(a_h[3], a_h[1], a_h[2], b_h[3], b_h[1], b_h[2])
        cr, cc, cv = oracle.spgemm_numeric(a_h[3], a_h[1], a_h[2], a_h[0], b_h[3], b_h[1], b_h[2], b_h[0],
                                           capacity=ref_nnz, scale_a=2.0)
        assert cn == ref_nnz
        ab = absprod_rows(a_h, b_h, cr, cc) * 2.0
        if c_fmt == "csc":
            abt, _, _, _ = _host_transpose(ab.astype(np.float64), cr, cc, (m, n))
            cv, cr, cc, _ = _host_transpose(cv, cr, cc, (m, n))
            ab = abt
        assert np.array_equal(got_p, cr) and np.array_equal(got_i, cc)
        util.assert_parity(got_v, cv, ab, dtype, row_len=np.full(len(cv), 64), what=f"spgemm {a_fmt}*{b_fmt}->{c_fmt}")


def _random_csr(rng, rows, cols, len_lo, len_hi, dtype, sorted_rows=True):
    lens = rng.integers(len_lo, len_hi + 1, rows)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    colind = np.empty(int(rowptr[-1]), np.int32)
    for r in range(rows):
        c = rng.choice(cols, int(lens[r]), replace=False)
        colind[rowptr[r]:rowptr[r + 1]] = np.sort(c) if sorted_rows else c
    values = (rng.random(len(colind)) - 0.5).astype(dtype)
    return values, rowptr, colind, (rows, cols)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [3_000_000, 300])
@pytest.mark.parametrize("b_len", [(12, 16), (5, 8), (1, 4), (0, 16)])
def test_spgemm_direct_rows(gpu, dtype, b_len, n):
    """Rows whose product count equals their structural length (csrc/spgemm.hip: spg_direct_kernel, a persistent kernel that
    sorts the products in registers and LDS -- no hash): very sparse operands with a wide column range, B rows of every
    admitted length class (16 / 8 / 4 lanes' worth, empty rows, unsorted rows), A rows of 0 .. the round's limit, a last
    B row that ends at the end of the arrays (its vector reads may not run over: those rows must take the hash kernel),
    alpha, and further fills with new values (by hash + recording, then by rank).  n = 300 columns: every row has products
    that share a column, many of them several times (the DUP instance of the kernel: ties in the sort, run sums, compaction).
    Exact structure, values in bound."""
    rng = np.random.default_rng(97 + b_len[0] + (n & 7))
    m, k = 6000, 9000
    sub = 16 if b_len[1] > 8 else 8 if b_len[1] > 4 else 4
    a_h = _random_csr(rng, m, k, 0, 256 // sub, dtype)
    b_h = _random_csr(rng, k, n, b_len[0], b_len[1], dtype, sorted_rows=False)
    if n < 10000:  # sums of many products per entry: positive values, as the reference's generators produce -- its comparator
        a_h = (np.abs(a_h[0]) + dtype(0.01),) + a_h[1:]  # (EXPECT_EQ_, relative to the RESULT) is not meant for cancellation
        b_h = (np.abs(b_h[0]) + dtype(0.01),) + b_h[1:]
    # make sure the last B row is used by many A rows
    av, ar, ac, ash = a_h
    first = ar[:-1][np.diff(ar) > 0]
    ac[first[::7]] = k - 1
    got = device_spgemm((av, ar, ac, ash), b_h, True, scale_a=-2.5)
    check_against_oracle((av, ar, ac, ash), b_h, got, dtype, scale=-2.5)
    # the kernel choice, and a second fill with other values through the same state
    d_a = G.csr_on_device(av, ar, ac, ash, len(av))
    d_b = G.csr_on_device(*b_h, len(b_h[0]))
    d_rp = torch.zeros(m + 1, dtype=torch.int32, device="cuda")
    d_c = sp.csr_view(None, d_rp, None, (m, n), 0)
    state = sp.spgemm_state_t()
    sp.multiply_compute(state, d_a, d_b, d_c)
    info = state.info()
    products = np.add.reduceat(np.diff(b_h[1])[ac], ar[:-1])[np.diff(ar) > 0] if len(ac) else np.zeros(0)
    wave_rows = int(((products > 64) & (products <= 256)).sum())
    assert info["nnz_c"] == got[0] and info["wave_per_row_rows"] == wave_rows
    assert 0.5 * wave_rows <= info["direct_rows"] <= wave_rows and wave_rows > 100
    if n < 10000:
        assert got[0] < 0.9 * products[products > 0].sum()  # (shared columns everywhere: C is much smaller than the product list)
    nnz = state.result_nnz()
    vals = torch.full((nnz,), float("nan"), dtype=G.dev(av).dtype, device="cuda")
    cols = torch.full((nnz,), -1, dtype=torch.int32, device="cuda")
    d_c.update(vals, d_rp, cols, (m, n), nnz)
    for rep in range(2):
        d_a.values().mul_(1.5)
        d_b.values().add_(0.25 if n < 10000 else -0.25)
        sp.multiply_fill(state, d_a, d_b, d_c)
        a2 = (G.host(d_a.values()), ar, ac, ash)
        b2 = (G.host(d_b.values()), b_h[1], b_h[2], b_h[3])
        check_against_oracle(a2, b2, (nnz, G.host(d_rp), G.host(cols), G.host(vals)), dtype)
        vals.fill_(float("nan"))
        cols.fill_(-1)
