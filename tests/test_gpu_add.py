"""-m gpu parity tests for the SURVEY 8f rows built on the SpGEMM accumulators:
  * add(a, b, c) / add_inspect / add_compute            (algorithms/add_impl.hpp:40-115;
    cases mirror /root/reference/test/gtest/add_test.cpp:9-60: inspect -> allocate -> update -> compute)
  * four-argument SpGEMM  C = alpha*A*B + beta*D         (vendor/rocsparse/multiply_spgemm.hpp:118-214;
    cases mirror test/gtest/device/rocsparse/spgemm_4args_test.cpp: plain, A/B/D scaled, reuse)
Indices (rowptr, sorted colind, result_nnz) are compared EXACTLY with the CPU oracle, values within
the parity bound."""
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


def _csr(m, n, nnz, seed, dtype):
    return generate.generate_csr(m, n, nnz, seed=seed, dtype=dtype)[:4]


def device_add(a_h, b_h, scale_a=None, scale_b=None, one_shot=False):
    (av, ar, ac, ash), (bv, br, bc, bsh) = a_h, b_h
    m, n = ash
    d_a = G.csr_on_device(av, ar, ac, ash, len(av))
    d_b = G.csr_on_device(bv, br, bc, bsh, len(bv))
    A = sp.scaled(scale_a, d_a) if scale_a is not None else d_a
    B = sp.scaled(scale_b, d_b) if scale_b is not None else d_b
    d_rowptr = torch.full((m + 1,), -1, dtype=torch.int32, device="cuda")
    if one_shot:  # add(a, b, c) with a pre-sized output (csr_builder semantics)
        cap = len(av) + len(bv)
        d_vals = torch.full((cap,), float("nan"), dtype=G.dev(av).dtype, device="cuda")
        d_cols = torch.full((cap,), -1, dtype=torch.int32, device="cuda")
        d_c = sp.csr_view(d_vals, d_rowptr, d_cols, (m, n), 0)
        sp.add(A, B, d_c)
        nnz = d_c.size()
        return nnz, G.host(d_rowptr), G.host(d_cols)[:nnz], G.host(d_vals)[:nnz]
    d_c = sp.csr_view(None, d_rowptr, None, (m, n), 0)           # add_test.cpp:24-26
    info = sp.add_inspect(A, B, d_c)                              # :28
    nnz = info.result_nnz()
    assert info.result_shape() == (m, n)
    d_vals = torch.full((nnz,), float("nan"), dtype=G.dev(av).dtype, device="cuda")
    d_cols = torch.full((nnz,), -1, dtype=torch.int32, device="cuda")
    d_c.update(d_vals, d_rowptr, d_cols)                          # :33
    sp.add_compute(info, A, B, d_c)                               # :35
    assert d_c.size() == nnz
    return nnz, G.host(d_rowptr), G.host(d_cols), G.host(d_vals)


def check_add(a_h, b_h, got, dtype, scale_a=None, scale_b=None):
    (av, ar, ac, ash), (bv, br, bc, bsh) = a_h, b_h
    nnz, c_rowptr, c_colind, c_values = got
    ref_nnz, ref_rowptr = oracle.add(ash, ar, ac, av, bsh, br, bc, bv, symbolic=True)
    assert nnz == ref_nnz
    cr, cc, cv = oracle.add(ash, ar, ac, av, bsh, br, bc, bv, scale_a=scale_a, scale_b=scale_b)
    assert np.array_equal(c_rowptr, cr) and np.array_equal(ref_rowptr, cr)
    assert np.array_equal(c_colind, cc)
    _, _, absv = oracle.add(ash, ar, ac, np.abs(av) * abs(scale_a or 1), bsh, br, bc, np.abs(bv) * abs(scale_b or 1))
    util.assert_parity(c_values, cv, absv.astype(np.float64), dtype, what="add values")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("dim", util.dims)
def test_add_reference_test(gpu, dim, dtype):
    m, n, nnz = dim
    a_h, b_h = _csr(m, n, nnz, 0, dtype), _csr(m, n, nnz, 1, dtype)
    check_add(a_h, b_h, device_add(a_h, b_h), dtype)
    check_add(a_h, b_h, device_add(a_h, b_h, one_shot=True), dtype)


@pytest.mark.parametrize("scales", [(2.0, None), (None, -0.5), (3.0, 0.25)])
def test_add_scaled_views(gpu, scales):
    sa, sb = scales
    a_h, b_h = _csr(300, 200, 5000, 2, np.float32), _csr(300, 200, 3000, 3, np.float32)
    check_add(a_h, b_h, device_add(a_h, b_h, sa, sb), np.float32, sa, sb)


def test_add_is_bit_exact_for_unique_columns(gpu):
    """With at most one entry per (row, column) in each operand every output value is a + b (or a
    copy): no reassociation, so the device result must equal the oracle bit for bit."""
    rng = np.random.default_rng(5)
    import scipy.sparse as sps
    A = sps.random(2000, 3000, density=0.004, format="csr", random_state=rng, dtype=np.float32)
    B = sps.random(2000, 3000, density=0.006, format="csr", random_state=rng, dtype=np.float32)
    a_h = (A.data, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.shape)
    b_h = (B.data, B.indptr.astype(np.int32), B.indices.astype(np.int32), B.shape)
    nnz, cr, cc, cv = device_add(a_h, b_h)
    rr, rc, rv = oracle.add(A.shape, a_h[1], a_h[2], a_h[0], B.shape, b_h[1], b_h[2], b_h[0])
    assert np.array_equal(cr, rr) and np.array_equal(cc, rc)
    assert np.array_equal(cv.view(np.uint32), rv.view(np.uint32))


def test_add_empty_and_disjoint_and_long_rows(gpu):
    # empty operands, empty rows, one very long row (dense accumulator bin), identical patterns
    m, n = 64, 20000
    rng = np.random.default_rng(7)
    a_rp = np.zeros(m + 1, np.int32)
    a_rp[1:] = 0
    a_h = (np.zeros(0, np.float32), a_rp, np.zeros(0, np.int32), (m, n))
    long_cols = rng.permutation(n)[:9000].astype(np.int32)
    b_len = np.zeros(m, np.int64)
    b_len[3] = 9000
    b_len[10] = 17
    b_rp = np.zeros(m + 1, np.int32)
    b_rp[1:] = np.cumsum(b_len)
    b_ci = np.concatenate([long_cols, rng.permutation(n)[:17].astype(np.int32)])
    b_h = (rng.random(len(b_ci), dtype=np.float32), b_rp, b_ci, (m, n))
    check_add(a_h, b_h, device_add(a_h, b_h), np.float32)
    check_add(b_h, a_h, device_add(b_h, a_h), np.float32)
    check_add(b_h, b_h, device_add(b_h, b_h), np.float32)
    check_add(a_h, a_h, device_add(a_h, a_h), np.float32)


def test_add_errors(gpu):
    a_h, b_h = _csr(50, 40, 100, 0, np.float32), _csr(50, 41, 100, 1, np.float32)
    d_a = G.csr_on_device(*a_h, len(a_h[0]))
    d_b = G.csr_on_device(*b_h, len(b_h[0]))
    rp = torch.zeros(51, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):                              # add_impl.hpp:44-47
        sp.add_inspect(d_a, d_b, sp.csr_view(None, rp, None, (50, 40), 0))
    b2 = _csr(50, 40, 100, 1, np.float32)
    d_b2 = G.csr_on_device(*b2, len(b2[0]))
    small = sp.csr_view(torch.zeros(3, device="cuda"), rp, torch.zeros(3, dtype=torch.int32, device="cuda"), (50, 40), 0)
    with pytest.raises(RuntimeError):                            # add_impl.hpp:67-72
        sp.add(d_a, d_b2, small)


# ------------------------------------------------------------------ C = alpha*A*B + beta*D
def device_spgemm4(a_h, b_h, d_h, sa=None, sb=None, sd=None, reuse_values=None):
    (av, ar, ac, ash), (bv, br, bc, bsh), (dv, dr, dc, dsh) = a_h, b_h, d_h
    m, n = ash[0], bsh[1]
    d_a, d_b, d_d = (G.csr_on_device(v, r, c, s, len(v)) for v, r, c, s in (a_h, b_h, d_h))
    A = sp.scaled(sa, d_a) if sa is not None else d_a
    B = sp.scaled(sb, d_b) if sb is not None else d_b
    D = sp.scaled(sd, d_d) if sd is not None else d_d
    d_rowptr = torch.full((m + 1,), -1, dtype=torch.int32, device="cuda")
    d_c = sp.csr_view(None, d_rowptr, None, (m, n), 0)            # spgemm_4args_test.cpp:49-52
    state = sp.spgemm_state_t()
    sp.multiply_compute(state, A, B, d_c, D)                       # :55
    nnz = state.result_nnz()
    d_vals = torch.full((nnz,), float("nan"), dtype=G.dev(av).dtype, device="cuda")
    d_cols = torch.full((nnz,), -1, dtype=torch.int32, device="cuda")
    d_c.update(d_vals, d_rowptr, d_cols, (m, n), nnz)              # :62-63
    sp.multiply_fill(state, A, B, d_c, D)                          # :65
    out = [(nnz, G.host(d_rowptr), G.host(d_cols), G.host(d_vals))]
    if reuse_values is not None:                                   # numeric again with new values, same pattern
        av2, bv2, dv2 = reuse_values
        d_a.values().copy_(G.dev(av2))
        d_b.values().copy_(G.dev(bv2))
        d_d.values().copy_(G.dev(dv2))
        d_vals.fill_(float("nan"))
        sp.multiply_numeric(state, A, B, d_c, D)
        out.append((nnz, G.host(d_rowptr), G.host(d_cols), G.host(d_vals)))
    return out


def check_spgemm4(a_h, b_h, d_h, got, dtype, alpha=1.0, beta=1.0):
    (av, ar, ac, ash), (bv, br, bc, bsh), (dv, dr, dc, dsh) = a_h, b_h, d_h
    nnz, c_rowptr, c_colind, c_values = got
    ref_nnz, _ = oracle.spgemm_symbolic_d(ash, ar, ac, bsh, br, bc, dsh, dr, dc)
    assert nnz == ref_nnz
    cr, cc, cv = oracle.spgemm_numeric_d(ash, ar, ac, av, bsh, br, bc, bv, dsh, dr, dc, dv, ref_nnz, alpha, beta)
    assert np.array_equal(c_rowptr, cr) and np.array_equal(c_colind, cc)
    _, _, ab = oracle.spgemm_numeric_d(ash, ar, ac, np.abs(av), bsh, br, bc, np.abs(bv), dsh, dr, dc, np.abs(dv),
                                       ref_nnz, abs(alpha), abs(beta))
    util.assert_parity(c_values, cv, ab.astype(np.float64), dtype, row_len=np.full(len(cv), 64),
                       what="spgemm4 values")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("dim", util.dims)
def test_spgemm_4args_reference_test(gpu, dim, dtype):
    m, k, nnz = dim
    for n in (m, k):
        a_h, b_h, d_h = _csr(m, k, nnz, 0, dtype), _csr(k, n, nnz, 1, dtype), _csr(m, n, nnz, 2, dtype)
        check_spgemm4(a_h, b_h, d_h, device_spgemm4(a_h, b_h, d_h)[0], dtype)


@pytest.mark.parametrize("scales", [(2.0, None, None), (None, 2.0, None), (None, None, 2.0), (2.0, 3.0, -0.5)])
def test_spgemm_4args_scaled(gpu, scales):
    sa, sb, sd = scales                                            # _AScaled / _BScaled / _DScaled variants
    a_h, b_h, d_h = _csr(100, 1000, 10000, 0, np.float32), _csr(1000, 100, 10000, 1, np.float32), \
        _csr(100, 100, 3000, 2, np.float32)
    alpha = (sa or 1.0) * (sb or 1.0)
    check_spgemm4(a_h, b_h, d_h, device_spgemm4(a_h, b_h, d_h, sa, sb, sd)[0], np.float32, alpha, sd or 1.0)


def test_spgemm_4args_numeric_reuse_and_mismatch(gpu):
    rng = np.random.default_rng(3)
    a_h, b_h, d_h = _csr(400, 300, 4000, 0, np.float32), _csr(300, 500, 6000, 1, np.float32), \
        _csr(400, 500, 5000, 2, np.float32)
    new = tuple(rng.random(len(x[0]), dtype=np.float32) for x in (a_h, b_h, d_h))
    first, second = device_spgemm4(a_h, b_h, d_h, reuse_values=new)
    check_spgemm4(a_h, b_h, d_h, first, np.float32)
    a2, b2, d2 = ((new[i],) + t[1:] for i, t in enumerate((a_h, b_h, d_h)))
    check_spgemm4(a2, b2, d2, second, np.float32)
    # the addend must be given to both phases
    d_a, d_b, d_d = (G.csr_on_device(v, r, c, s, len(v)) for v, r, c, s in (a_h, b_h, d_h))
    rp = torch.zeros(401, dtype=torch.int32, device="cuda")
    d_c = sp.csr_view(None, rp, None, (400, 500), 0)
    state = sp.spgemm_state_t()
    sp.multiply_compute(state, d_a, d_b, d_c, d_d)
    nnz = state.result_nnz()
    d_c.update(torch.zeros(nnz, device="cuda"), rp, torch.zeros(nnz, dtype=torch.int32, device="cuda"), (400, 500), nnz)
    with pytest.raises(RuntimeError):
        sp.multiply_fill(state, d_a, d_b, d_c)
    with pytest.raises(ValueError):                                # D must have C's shape
        sp.multiply_compute(state, d_a, d_b, d_c, d_a)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n_cols,d_max", [(40000, 16), (900, 64), (40000, 70)])
def test_spgemm_4args_sorted_rows_take_the_addend_as_a_fifth_element(gpu, dtype, n_cols, d_max, monkeypatch):
    """Round 6: rows whose products a wavefront sorts in registers (<= 256 products, A row in one round of loads: cfg5's shape)
    keep that path when an addend is present -- its row joins as a fifth element per lane, up to 320 entries per row -- instead
    of falling to the hash of the next bin (cfg5 + D: 6.3 -> 1.0 ms per fill).  16 x 16 products exactly fill the 256 slots;
    n_cols = 900 makes products and addend entries share columns (the duplicate-summing variant, enumeration order:
    products, then the addend); d_max = 70 puts some addend rows beyond one entry per lane (those rows must hash); ragged A
    rows and empty addend rows included.  Structure exact, values norm-wise, against the oracle; then a second fill with new
    values (which must NOT switch to the rank path: its per-row capacity is 256)."""
    rng = np.random.default_rng(61)
    m, k = 3000, 5000

    def csr(rows, cols, lens, seed):
        r = np.random.default_rng(seed)
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        ci = np.concatenate([r.choice(cols, l, replace=False) for l in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
        return (r.random(len(ci)) - 0.5).astype(dtype), rp, ci, (rows, cols)

    a_len = np.full(m, 16)
    a_len[::7] = rng.integers(1, 16, len(a_len[::7]))
    d_len = rng.integers(0, d_max + 1, m)
    d_len[::5] = 0
    a_h, b_h, d_h = csr(m, k, a_len, 1), csr(k, n_cols, np.full(k, 16), 2), csr(m, n_cols, d_len, 3)
    new = tuple((np.random.default_rng(9 + i).random(len(x[0])) - 0.5).astype(dtype) for i, x in enumerate((a_h, b_h, d_h)))
    first, second = device_spgemm4(a_h, b_h, d_h, sa=1.5, sd=-0.5, reuse_values=new)
    check_spgemm4(a_h, b_h, d_h, first, dtype, 1.5, -0.5)
    a2, b2, d2 = ((new[i],) + t[1:] for i, t in enumerate((a_h, b_h, d_h)))
    check_spgemm4(a2, b2, d2, second, dtype, 1.5, -0.5)
    # the state says which rows took the sort-based kernel
    d_a, d_b, d_d = (G.csr_on_device(v, r, c, s_, len(v)) for v, r, c, s_ in (a_h, b_h, d_h))
    rp = torch.zeros(m + 1, dtype=torch.int32, device="cuda")
    state = sp.spgemm_state_t()
    sp.multiply_compute(state, d_a, d_b, sp.csr_view(None, rp, None, (m, n_cols), 0), d_d)
    info = state.info()
    assert info["direct_rows"] >= (0.5 if d_max <= 64 else 0.3) * m, info
    monkeypatch.setenv("SPBLAS_GFX950_SPG_DIRECT_ADD", "0")      # the hash path of round 5 gives the same structure
    state0 = sp.spgemm_state_t()
    rp0 = torch.zeros(m + 1, dtype=torch.int32, device="cuda")
    sp.multiply_compute(state0, d_a, d_b, sp.csr_view(None, rp0, None, (m, n_cols), 0), d_d)
    assert state0.info()["direct_rows"] == 0 and torch.equal(rp, rp0)


def test_spgemm_4args_all_bins(gpu):
    """Rows of every accumulator size (empty, 128/512/2048/8192-slot hash, dense bitmap) with an addend
    that alone decides the bin for some rows."""
    rng = np.random.default_rng(11)
    m, k, n = 40, 3000, 30000
    a_len = rng.integers(0, 3, m)
    a_len[5], a_len[6], a_len[7] = 40, 200, 1500
    a_rp = np.zeros(m + 1, np.int32)
    a_rp[1:] = np.cumsum(a_len)
    a_ci = np.concatenate([rng.permutation(k)[:l] for l in a_len]).astype(np.int32)
    a_h = (rng.random(len(a_ci), dtype=np.float32), a_rp, a_ci, (m, k))
    b_h = _csr(k, n, 12000, 1, np.float32)
    d_len = rng.integers(0, 4, m)
    d_len[0], d_len[1], d_len[2], d_len[3] = 100, 600, 3000, 9000     # rows whose products are few
    d_rp = np.zeros(m + 1, np.int32)
    d_rp[1:] = np.cumsum(d_len)
    d_ci = np.concatenate([rng.permutation(n)[:l] for l in d_len]).astype(np.int32)
    d_h = (rng.random(len(d_ci), dtype=np.float32), d_rp, d_ci, (m, n))
    check_spgemm4(a_h, b_h, d_h, device_spgemm4(a_h, b_h, d_h, sd=0.5)[0], np.float32, 1.0, 0.5)


# ------------------------------------------------------------------ golden fixtures (integer data: bit exact)
@pytest.mark.parametrize("name", ["add_plain.npz", "add_scaled.npz"])
def test_add_golden_bit_exact(gpu, name):
    g = np.load(os.path.join(GOLDEN, name))
    shape = tuple(int(v) for v in g["shape"])
    a_h = (g["a_values"], g["a_rowptr"], g["a_colind"], shape)
    b_h = (g["b_values"], g["b_rowptr"], g["b_colind"], shape)
    sa, sb = float(g["scale_a"]), float(g["scale_b"])
    nnz, cr, cc, cv = device_add(a_h, b_h, None if sa == 1 else sa, None if sb == 1 else sb)
    assert nnz == int(g["c_nnz"])
    assert np.array_equal(cr, g["c_rowptr"]) and np.array_equal(cc, g["c_colind"])
    assert np.array_equal(cv, g["c_values"])                    # includes the row that cancels to stored zeros


def test_spgemm_4args_golden_bit_exact(gpu):
    g = np.load(os.path.join(GOLDEN, "spgemm4_scaled.npz"))
    sh = [tuple(int(v) for v in g[k]) for k in ("a_shape", "b_shape", "d_shape")]
    a_h = (g["a_values"], g["a_rowptr"], g["a_colind"], sh[0])
    b_h = (g["b_values"], g["b_rowptr"], g["b_colind"], sh[1])
    d_h = (g["d_values"], g["d_rowptr"], g["d_colind"], sh[2])
    nnz, cr, cc, cv = device_spgemm4(a_h, b_h, d_h, sa=float(g["alpha"]), sd=float(g["beta"]))[0]
    assert nnz == int(g["c_nnz"])
    assert np.array_equal(cr, g["c_rowptr"]) and np.array_equal(cc, g["c_colind"])
    assert np.array_equal(cv, g["c_values"])


def test_spgemm_with_empty_operands(gpu):
    """B (or A) without a single stored entry: no products, C = beta*D (or an all-empty C); the kernels'
    clamped B-row loads must not touch B's (zero-length) arrays."""
    m, k, n = 300, 200, 250
    a_h = _csr(m, k, 3000, 0, np.float32)
    e_kn = (np.zeros(0, np.float32), np.zeros(k + 1, np.int32), np.zeros(0, np.int32), (k, n))
    e_mk = (np.zeros(0, np.float32), np.zeros(m + 1, np.int32), np.zeros(0, np.int32), (m, k))
    b_h = _csr(k, n, 2000, 1, np.float32)
    d_h = _csr(m, n, 4000, 2, np.float32)
    check_spgemm4(a_h, e_kn, d_h, device_spgemm4(a_h, e_kn, d_h, sd=-2.0)[0], np.float32, 1.0, -2.0)
    check_spgemm4(e_mk, b_h, d_h, device_spgemm4(e_mk, b_h, d_h)[0], np.float32)


# ------------------------------------------------------------------ repeated fills: the rank path
def _with_duplicates(h, rng, every=7):
    """Repeat a column inside some rows (csr_view allows unsorted / repeated columns)."""
    v, rp, ci, sh = h
    ci = ci.copy()
    for r in range(0, len(rp) - 1, every):
        if rp[r + 1] - rp[r] >= 2:
            ci[rp[r] + 1] = ci[rp[r]]
    return (v, rp, ci, sh)


@pytest.mark.parametrize("record_at", ["second", "first"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_add_repeated_compute_by_rank(gpu, monkeypatch, dtype, record_at):
    """add_compute again and again on one add_inspect result (add_impl.hpp:110-113 allows it): the second call records
    the rank of every entry of A and B in its output row, later calls accumulate by rank.  Values and scale factors
    change between the calls; columns are poisoned before each call."""
    if record_at == "first":
        monkeypatch.setenv("SPBLAS_GFX950_SPGEMM_REUSE", "2")
    rng = np.random.default_rng(5)
    m, n = 700, 900
    a_h = _with_duplicates(_csr(m, n, 9000, 0, dtype), rng)
    b_h = _with_duplicates(_csr(m, n, 12000, 1, dtype), rng, every=5)
    # one long row on each side (falls outside the rank bins) and a few empty ones come with the generator
    (av, ar, ac, ash), (bv, br, bc, bsh) = a_h, b_h
    d_a, d_b = G.csr_on_device(av, ar, ac, ash, len(av)), G.csr_on_device(bv, br, bc, bsh, len(bv))
    d_rowptr = torch.full((m + 1,), -1, dtype=torch.int32, device="cuda")
    d_c = sp.csr_view(None, d_rowptr, None, (m, n), 0)
    info = sp.add_inspect(d_a, d_b, d_c)
    nnz = info.result_nnz()
    d_vals = torch.full((nnz,), float("nan"), dtype=G.dev(av).dtype, device="cuda")
    d_cols = torch.full((nnz,), -1, dtype=torch.int32, device="cuda")
    d_c.update(d_vals, d_rowptr, d_cols)
    for it, (sa, sb) in enumerate([(None, None), (2.0, None), (None, -0.5), (1.5, 3.0), (None, None)]):
        av2, bv2 = rng.random(len(av)).astype(dtype), rng.random(len(bv)).astype(dtype)
        d_a.values().copy_(G.dev(av2))
        d_b.values().copy_(G.dev(bv2))
        d_vals.fill_(float("nan"))
        d_cols.fill_(-1)
        A = sp.scaled(sa, d_a) if sa is not None else d_a
        B = sp.scaled(sb, d_b) if sb is not None else d_b
        sp.add_compute(info, A, B, d_c)
        got = (nnz, G.host(d_rowptr), G.host(d_cols), G.host(d_vals))
        check_add((av2, ar, ac, ash), (bv2, br, bc, bsh), got, dtype, sa, sb)


@pytest.mark.parametrize("record_at", ["second", "first"])
def test_spgemm_4args_repeated_fills_by_rank(gpu, monkeypatch, record_at):
    """multiply_numeric(state, A, B, C, D) five times with new values and factors: hash pass, recording pass, then the
    fills by rank with the addend's entries enumerated after the products (multiply_spgemm.hpp:178-214)."""
    if record_at == "first":
        monkeypatch.setenv("SPBLAS_GFX950_SPGEMM_REUSE", "2")
    rng = np.random.default_rng(9)
    dtype = np.float32
    a_h, b_h, d_h = _csr(500, 400, 5000, 0, dtype), _csr(400, 600, 7000, 1, dtype), _csr(500, 600, 6000, 2, dtype)
    d_h = _with_duplicates(d_h, rng)
    (av, ar, ac, ash), (bv, br, bc, bsh), (dv, dr, dc, dsh) = a_h, b_h, d_h
    d_a, d_b, d_d = (G.csr_on_device(v, r, c, s, len(v)) for v, r, c, s in (a_h, b_h, d_h))
    m, n = ash[0], bsh[1]
    d_rowptr = torch.full((m + 1,), -1, dtype=torch.int32, device="cuda")
    d_c = sp.csr_view(None, d_rowptr, None, (m, n), 0)
    state = sp.spgemm_state_t()
    sp.multiply_compute(state, d_a, d_b, d_c, d_d)
    nnz = state.result_nnz()
    d_vals = torch.full((nnz,), float("nan"), dtype=torch.float32, device="cuda")
    d_cols = torch.full((nnz,), -1, dtype=torch.int32, device="cuda")
    d_c.update(d_vals, d_rowptr, d_cols, (m, n), nnz)
    for it, (sa, sd) in enumerate([(None, None), (2.0, None), (None, -0.25), (0.5, 3.0), (None, None)]):
        new = [rng.random(len(x)).astype(dtype) for x in (av, bv, dv)]
        for t, x in zip((d_a, d_b, d_d), new):
            t.values().copy_(G.dev(x))
        d_vals.fill_(float("nan"))
        d_cols.fill_(-1)
        A = sp.scaled(sa, d_a) if sa is not None else d_a
        D = sp.scaled(sd, d_d) if sd is not None else d_d
        sp.multiply_numeric(state, A, d_b, d_c, D)
        got = (nnz, G.host(d_rowptr), G.host(d_cols), G.host(d_vals))
        check_spgemm4((new[0], ar, ac, ash), (new[1], br, bc, bsh), (new[2], dr, dc, dsh), got, dtype,
                      sa or 1.0, sd or 1.0)


def test_add_golden_bit_exact_by_rank(gpu, monkeypatch):
    """The golden add fixtures again with the ranks recorded in the first pass: the second pass IS the rank path."""
    monkeypatch.setenv("SPBLAS_GFX950_SPGEMM_REUSE", "2")
    for name in ("add_plain.npz", "add_scaled.npz"):
        g = np.load(os.path.join(GOLDEN, name))
        shape = tuple(int(v) for v in g["shape"])
        (av, ar, ac), (bv, br, bc) = (g["a_values"], g["a_rowptr"], g["a_colind"]), \
            (g["b_values"], g["b_rowptr"], g["b_colind"])
        sa, sb = float(g["scale_a"]), float(g["scale_b"])
        d_a, d_b = G.csr_on_device(av, ar, ac, shape, len(av)), G.csr_on_device(bv, br, bc, shape, len(bv))
        A = sp.scaled(sa, d_a) if sa != 1 else d_a
        B = sp.scaled(sb, d_b) if sb != 1 else d_b
        rp = torch.full((shape[0] + 1,), -1, dtype=torch.int32, device="cuda")
        d_c = sp.csr_view(None, rp, None, shape, 0)
        info = sp.add_inspect(A, B, d_c)
        nnz = info.result_nnz()
        vals = torch.full((nnz,), float("nan"), dtype=G.dev(av).dtype, device="cuda")
        cols = torch.full((nnz,), -1, dtype=torch.int32, device="cuda")
        d_c.update(vals, rp, cols)
        for _ in range(3):
            vals.fill_(float("nan"))
            cols.fill_(-1)
            sp.add_compute(info, A, B, d_c)
            assert np.array_equal(G.host(cols), g["c_colind"]) and np.array_equal(G.host(vals), g["c_values"])
