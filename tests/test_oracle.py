"""CPU tests (-m "not gpu"): pin the oracle.

The reference cannot be compiled in this image (see oracle/spblas_oracle.c header), and it
ships no stored vectors; its tests pin results with inline naive loops + EXPECT_EQ_.  Here
the oracle is checked against (1) those comparator loops restated on the reference's own
test shapes and scale factors, (2) an independent scipy.sparse product, (3) the exact
known-answer fixtures in tests/golden/ (integer-valued, so every summation order gives
the same bits).
"""
import os

import numpy as np
import pytest
import scipy.sparse as sps

import util
from oracle import oracle
from spblas_reference_amd import generate

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("dim", util.dims)
def test_spmv_matches_reference_test_loop(dim, dtype):
    # test/gtest/spmv_test.cpp:6-36  CsrView.SpMV (b = ones)
    m, n, nnz = dim
    values, rowptr, colind, shape, _ = generate.generate_csr(m, n, nnz, dtype=dtype)
    b = np.ones(n, dtype=dtype)
    c = oracle.spmv(shape, rowptr, colind, values, b)
    c_ref = util.naive_spmv(rowptr, colind, values, b)
    util.expect_eq_ref(c_ref, c)
    # same operation order => same bits up to FMA contraction of the compiled oracle
    assert np.allclose(c, c_ref, rtol=2 * np.finfo(dtype).eps * 128, atol=0)


@pytest.mark.parametrize("alpha", [-10, 1, 5])
@pytest.mark.parametrize("dim", util.dims)
def test_spmv_scaled_matches_reference_test_loop(dim, alpha):
    # spmv_test.cpp:38-72 SpMV_Ascaled, :74-108 SpMV_BScaled (alpha in {-10,1,5})
    m, n, nnz = dim
    values, rowptr, colind, shape, _ = generate.generate_csr(m, n, nnz)
    b = np.ones(n, dtype=np.float32)
    util.expect_eq_ref(util.naive_spmv(rowptr, colind, values, b, alpha_a=alpha),
                       oracle.spmv(shape, rowptr, colind, values, b, scale_a=alpha))
    util.expect_eq_ref(util.naive_spmv(rowptr, colind, values, b, alpha_b=alpha),
                       oracle.spmv(shape, rowptr, colind, values, b, scale_x=alpha))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spmv_vs_scipy_and_offsets64(dtype):
    values, rowptr, colind, shape, _ = generate.generate_csr(300, 257, 5000, seed=3, dtype=dtype)
    x = np.random.default_rng(1).random(257).astype(dtype)
    y = oracle.spmv(shape, rowptr, colind, values, x)
    exact, absrow = util.spmv_exact(rowptr, colind, values, x)
    lens = np.diff(rowptr)
    util.assert_parity(y, exact, absrow, dtype, row_len=lens, what="oracle vs float64")
    y64 = oracle.spmv(shape, rowptr.astype(np.int64), colind, values, x)
    assert np.array_equal(y, y64)
    assert np.array_equal(y, oracle.spmv_omp(rowptr, colind, values, x))


def test_spmv_shape_mismatch_raises_like_reference():
    # algorithms/multiply_impl.hpp:37-41 -> std::invalid_argument
    values, rowptr, colind, shape, _ = generate.generate_csr(10, 12, 30)
    with pytest.raises(ValueError):
        oracle.spmv(shape, rowptr, colind, values, np.ones(11, np.float32))
    with pytest.raises(ValueError):
        oracle.spmv(shape, rowptr, colind, values, np.ones(12, np.float32), y_len=9)


def test_spmv_csc_matches_csr_of_transpose():
    values, rowptr, colind, shape, _ = generate.generate_csr(50, 70, 600, seed=5)
    x = np.random.default_rng(2).random(50).astype(np.float32)
    # A (50x70) in CSR == A^T (70x50) in CSC; y = A^T x
    y = oracle.spmv_csc((70, 50), rowptr, colind, values, x)
    A = sps.csr_matrix((values.astype(np.float64), colind, rowptr), shape=shape)
    np.testing.assert_allclose(y, A.T @ x.astype(np.float64), rtol=1e-5)


@pytest.mark.parametrize("n", [1, 8, 32, 64, 512])
@pytest.mark.parametrize("dim", util.dims)
def test_spmm_matches_reference_test_loop(dim, n):
    # test/gtest/spmm_test.cpp:6-44 CsrView.SpMM: n in {1,8,32,64,512}, row-major B/C
    m, k, nnz = dim
    values, rowptr, colind, shape, _ = generate.generate_csr(m, k, nnz)
    B = generate.generate_dense(k, n)
    C = oracle.spmm(shape, rowptr, colind, values, B)
    c_ref = np.zeros((m, n), dtype=np.float32)
    for i in range(m):  # spmm_test.cpp:25-35
        for j_ptr in range(rowptr[i], rowptr[i + 1]):
            c_ref[i, :] += values[j_ptr] * B[colind[j_ptr], :]
    util.expect_eq_ref(c_ref, C)


@pytest.mark.parametrize("alpha", [-10, 1, 5])
def test_spmm_scaled(alpha):
    # spmm_test.cpp:46-136 SpMM_AScaled / SpMM_BScaled
    values, rowptr, colind, shape, _ = generate.generate_csr(100, 1000, 10000)
    B = generate.generate_dense(1000, 8)
    A = sps.csr_matrix((values.astype(np.float64), colind, rowptr), shape=shape)
    ref = alpha * (A @ B.astype(np.float64))
    for kw in ({"scale_a": alpha}, {"scale_b": alpha}):
        C = oracle.spmm(shape, rowptr, colind, values, B, **kw)
        np.testing.assert_allclose(C, ref, rtol=2e-5)
    with pytest.raises(ValueError):
        oracle.spmm(shape, rowptr, colind, values, B[:-1])


def _spa_reference_rows(a, b, n):
    """test/gtest/spgemm_test.cpp:38-67: per-row SPA accumulation of A*B."""
    (av, ar, ac), (bv, br, bc) = a, b
    rows = []
    for i in range(len(ar) - 1):
        acc = {}
        for p in range(ar[i], ar[i + 1]):
            k = ac[p]
            for q in range(br[k], br[k + 1]):
                j = int(bc[q])
                acc[j] = np.float32(acc.get(j, np.float32(0)) + av[p] * bv[q])
        rows.append(acc)
    return rows


@pytest.mark.parametrize("dim", util.dims)
def test_spgemm_matches_reference_test(dim):
    # spgemm_test.cpp:10-71 CsrView.SpGEMM: n in {m,k}; compute -> allocate -> fill;
    # per-row values via SPA, distinct-column count must match exactly (:67)
    m, k, nnz = dim
    for n in (m, k):
        av, ar, ac, ash, _ = generate.generate_csr(m, k, nnz)
        bv, br, bc, bsh, _ = generate.generate_csr(k, n, nnz, seed=1)
        c_nnz, row_nnz = oracle.spgemm_symbolic(ash, ar, ac, bsh, br, bc)
        cr, cc, cv = oracle.spgemm_numeric(ash, ar, ac, av, bsh, br, bc, bv, capacity=c_nnz)
        assert cr[-1] == c_nnz and np.array_equal(np.diff(cr), row_nnz)
        rows = _spa_reference_rows((av, ar, ac), (bv, br, bc), n)
        for i, acc in enumerate(rows):
            cols = cc[cr[i]:cr[i + 1]]
            assert len(acc) == len(cols)                      # spgemm_test.cpp:67
            assert np.all(np.diff(cols) > 0)                  # sorted ascending (gustavsons:42)
            util.expect_eq_ref(np.array([acc[int(j)] for j in cols], np.float32), cv[cr[i]:cr[i + 1]])
        # structural count == scipy's count on a pattern-only product
        P = (sps.csr_matrix((np.ones_like(av), ac, ar), shape=ash) @
             sps.csr_matrix((np.ones_like(bv), bc, br), shape=bsh))
        assert P.nnz == c_nnz


def test_spgemm_out_of_memory_and_scaling():
    av, ar, ac, ash, _ = generate.generate_csr(40, 40, 1000)
    bv, br, bc, bsh, _ = generate.generate_csr(40, 40, 1000, seed=1)
    c_nnz, _ = oracle.spgemm_symbolic(ash, ar, ac, bsh, br, bc)
    with pytest.raises(RuntimeError):  # spgemm_gustavsons.hpp:44-48
        oracle.spgemm_numeric(ash, ar, ac, av, bsh, br, bc, bv, capacity=c_nnz - 1)
    _, _, cv = oracle.spgemm_numeric(ash, ar, ac, av, bsh, br, bc, bv, capacity=c_nnz)
    _, _, cv2 = oracle.spgemm_numeric(ash, ar, ac, av, bsh, br, bc, bv, capacity=c_nnz, scale_a=2.0)
    np.testing.assert_allclose(cv2, 2 * cv, rtol=1e-6)  # spgemm_test.cpp:73-136 _AScaled
    with pytest.raises(ValueError):
        oracle.spgemm_symbolic(ash, ar, ac, (39, 40), br[:-1], bc)


@pytest.mark.parametrize("name", sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz")))
def test_oracle_reproduces_golden_exactly(name):
    g = np.load(os.path.join(GOLDEN, name))
    kind = str(g["kind"])
    if kind == "spmv":
        kw = {}
        if "scale_a" in g:
            kw["scale_a"] = float(g["scale_a"])
        if "scale_x" in g:
            kw["scale_x"] = float(g["scale_x"])
        y = oracle.spmv(tuple(g["shape"]), g["rowptr"], g["colind"], g["values"], g["x"], **kw)
        assert np.array_equal(y, g["y"])
    elif kind == "spmm":
        C = oracle.spmm(tuple(g["shape"]), g["rowptr"], g["colind"], g["values"], g["B"])
        assert np.array_equal(C, g["C"])
    elif kind == "spgemm":
        nnz, _ = oracle.spgemm_symbolic(tuple(g["a_shape"]), g["a_rowptr"], g["a_colind"], tuple(g["b_shape"]),
                                        g["b_rowptr"], g["b_colind"])
        assert nnz == int(g["c_nnz"])
        cr, cc, cv = oracle.spgemm_numeric(tuple(g["a_shape"]), g["a_rowptr"], g["a_colind"], g["a_values"],
                                           tuple(g["b_shape"]), g["b_rowptr"], g["b_colind"], g["b_values"],
                                           capacity=nnz)
        assert np.array_equal(cr, g["c_rowptr"]) and np.array_equal(cc, g["c_colind"])
        assert np.array_equal(cv, g["c_values"])
    elif kind == "add":
        sa, sb = float(g["scale_a"]), float(g["scale_b"])
        args = (tuple(g["shape"]), g["a_rowptr"], g["a_colind"], g["a_values"], tuple(g["shape"]), g["b_rowptr"],
                g["b_colind"], g["b_values"])
        nnz, rp0 = oracle.add(*args, symbolic=True)
        assert nnz == int(g["c_nnz"]) and np.array_equal(rp0, g["c_rowptr"])
        cr, cc, cv = oracle.add(*args, scale_a=None if sa == 1 else sa, scale_b=None if sb == 1 else sb)
        assert np.array_equal(cr, g["c_rowptr"]) and np.array_equal(cc, g["c_colind"])
        assert np.array_equal(cv, g["c_values"])
    elif kind == "spgemm4":
        sh = [tuple(g[k]) for k in ("a_shape", "b_shape", "d_shape")]
        nnz, _ = oracle.spgemm_symbolic_d(sh[0], g["a_rowptr"], g["a_colind"], sh[1], g["b_rowptr"], g["b_colind"],
                                          sh[2], g["d_rowptr"], g["d_colind"])
        assert nnz == int(g["c_nnz"])
        cr, cc, cv = oracle.spgemm_numeric_d(sh[0], g["a_rowptr"], g["a_colind"], g["a_values"], sh[1], g["b_rowptr"],
                                             g["b_colind"], g["b_values"], sh[2], g["d_rowptr"], g["d_colind"],
                                             g["d_values"], nnz, float(g["alpha"]), float(g["beta"]))
        assert np.array_equal(cr, g["c_rowptr"]) and np.array_equal(cc, g["c_colind"])
        assert np.array_equal(cv, g["c_values"])
    elif kind == "trsv":
        for upper in (False, True):
            for unit in (False, True):
                x = oracle.triangular_solve(tuple(g["shape"]), g["rowptr"], g["colind"], g["values"], g["b"],
                                            upper=upper, unit=unit)
                key = f"x_{'upper' if upper else 'lower'}_{'unit' if unit else 'explicit'}"
                assert np.array_equal(x, g[key]), key
    else:
        raise AssertionError(kind)


@pytest.mark.parametrize("dim", util.dims)
def test_transpose_matches_reference_test(dim):
    # test/gtest/transpose_test.cpp:9-71: B = A^T, compared as COO after sorting by (row, col)
    m, k, nnz = dim
    values, rowptr, colind, shape, _ = generate.generate_csr(m, k, nnz)
    tr, tc, tv = oracle.transpose(shape, rowptr, colind, values)
    A = sps.csr_matrix((values, colind, rowptr), shape=shape)
    AT = A.T.tocsr()
    AT.sort_indices()
    assert np.array_equal(tr, AT.indptr) and np.array_equal(tc, AT.indices) and np.array_equal(tv, AT.data)
    with pytest.raises(ValueError):
        oracle.transpose(shape, rowptr, colind, values, b_shape=(k + 1, m))
    with pytest.raises(RuntimeError):
        oracle.transpose(shape, rowptr, colind, values, capacity=nnz - 1)


# ---- add / four-argument SpGEMM ------------------------------------
@pytest.mark.parametrize("dim", util.dims)
def test_add_matches_reference_test_loop(dim):
    """test/gtest/add_test.cpp:37-58: per row, SPA over row i of A then of B; every stored output
    entry must equal the SPA value; add_inspect's nnz is the structural union."""
    m, n, nnz = dim
    av, ar, ac, ash, _ = generate.generate_csr(m, n, nnz, seed=0)
    bv, br, bc, bsh, _ = generate.generate_csr(m, n, nnz, seed=1)
    cnt, rp0 = oracle.add(ash, ar, ac, av, bsh, br, bc, bv, symbolic=True)
    rp, ci, cv = oracle.add(ash, ar, ac, av, bsh, br, bc, bv, capacity=cnt)
    assert np.array_equal(rp, rp0) and len(ci) == cnt == rp[-1]
    for i in range(m):
        spa = {}
        for p in range(ar[i], ar[i + 1]):
            spa[ac[p]] = np.float32(spa.get(ac[p], np.float32(0)) + av[p])
        for p in range(br[i], br[i + 1]):
            spa[bc[p]] = np.float32(spa.get(bc[p], np.float32(0)) + bv[p])
        cols = ci[rp[i]:rp[i + 1]]
        assert list(cols) == sorted(spa)                       # ascending, structural union
        for j, v in zip(cols, cv[rp[i]:rp[i + 1]]):
            assert v == spa[j]                                 # same order of additions: exact


def test_add_scaled_shape_and_capacity():
    A = sps.random(60, 50, density=0.1, format="csr", random_state=1, dtype=np.float64)
    B = sps.random(60, 50, density=0.2, format="csr", random_state=2, dtype=np.float64)
    rp, ci, cv = oracle.add(A.shape, A.indptr, A.indices, A.data, B.shape, B.indptr, B.indices, B.data,
                            scale_a=2.0, scale_b=-0.5)
    R = (2.0 * A - 0.5 * B).tocsr()
    assert abs(sps.csr_matrix((cv, ci, rp), shape=A.shape) - R).max() < 1e-15
    with pytest.raises(ValueError):                             # add_impl.hpp:44-47
        oracle.add(A.shape, A.indptr, A.indices, A.data, (60, 51), B.indptr, B.indices, B.data)
    with pytest.raises(RuntimeError):                           # add_impl.hpp:67-72
        oracle.add(A.shape, A.indptr, A.indices, A.data, B.shape, B.indptr, B.indices, B.data, capacity=3)


@pytest.mark.parametrize("dim", util.dims)
def test_spgemm_4args_matches_reference_test_loop(dim):
    """test/gtest/device/rocsparse/spgemm_4args_test.cpp:78-108: SPA += a_v*b_v over the products,
    then += d_v over row i of D; sizes must agree."""
    m, k, nnz = dim
    n = k
    av, ar, ac, ash, _ = generate.generate_csr(m, k, nnz, seed=0)
    bv, br, bc, bsh, _ = generate.generate_csr(k, n, nnz, seed=1)
    dv, dr, dc, dsh, _ = generate.generate_csr(m, n, nnz, seed=2)
    cnt, row_nnz = oracle.spgemm_symbolic_d(ash, ar, ac, bsh, br, bc, dsh, dr, dc)
    rp, ci, cv = oracle.spgemm_numeric_d(ash, ar, ac, av, bsh, br, bc, bv, dsh, dr, dc, dv, cnt)
    assert rp[-1] == cnt and np.array_equal(np.diff(rp), row_nnz)
    A = sps.csr_matrix((av.astype(np.float64), ac, ar), shape=ash)
    B = sps.csr_matrix((bv.astype(np.float64), bc, br), shape=bsh)
    D = sps.csr_matrix((dv.astype(np.float64), dc, dr), shape=dsh)
    R = (A @ B + D).toarray()
    C = sps.csr_matrix((cv.astype(np.float64), ci, rp), shape=(m, n)).toarray()
    util.expect_eq_ref(R.astype(np.float32), C.astype(np.float32))      # comparator in T = float
    for i in range(m):
        cols = ci[rp[i]:rp[i + 1]]
        assert np.all(np.diff(cols) > 0)
    with pytest.raises(ValueError):
        oracle.spgemm_symbolic_d(ash, ar, ac, bsh, br, bc, (m + 1, n), dr, dc)


# ---- triangular_solve ---------------------------------------------------
def _reference_triangular_solve(rowptr, colind, values, b, upper, unit):
    """The comparator of the reference test (test/gtest/triangular_solve_test.cpp:6-60): tmp = b[row];
    tmp -= a*x over the strict side; diag_val reset per row."""
    T = values.dtype.type
    m = len(rowptr) - 1
    x = np.zeros(m, dtype=values.dtype)
    rows = range(m - 1, -1, -1) if upper else range(m)
    for row in rows:
        tmp, diag = T(b[row]), T(0)
        for j in range(rowptr[row], rowptr[row + 1]):
            col = colind[j]
            if (col > row) if upper else (col < row):
                tmp = T(tmp - T(values[j] * x[col]))
            elif col == row:
                diag = values[j]
        x[row] = tmp if unit else T(tmp / diag)
    return x


@pytest.mark.parametrize("dim", util.square_dims)
@pytest.mark.parametrize("upper", [False, True])
def test_triangular_solve_matches_reference_test(dim, upper):
    """triangular_solve_test.cpp:63-86 (b = 0, values * 1e-3, unit diagonal) plus a non-trivial b."""
    m, n, nnz = dim
    v, rp, ci, shape, _ = generate.generate_csr(m, n, nnz)
    v = (v * np.float32(1e-3)).astype(np.float32)
    x = oracle.triangular_solve(shape, rp, ci, v, np.zeros(m, np.float32), upper=upper, unit=True)
    util.expect_eq_ref(_reference_triangular_solve(rp, ci, v, np.zeros(m, np.float32), upper, True), x)
    b = np.linspace(1, 2, m).astype(np.float32)
    x = oracle.triangular_solve(shape, rp, ci, v, b, upper=upper, unit=True)
    util.expect_eq_ref(_reference_triangular_solve(rp, ci, v, b, upper, True), x)


def test_triangular_solve_vs_scipy_and_shape_check():
    import scipy.sparse.linalg as spl
    rng = np.random.default_rng(0)
    n = 200
    A = sps.random(n, n, density=0.05, format="csr", random_state=rng, dtype=np.float64)
    b = rng.random(n)
    for upper in (False, True):
        T = ((sps.triu(A, 1) if upper else sps.tril(A, -1)) + sps.diags(rng.random(n) + 1.0)).tocsr()
        x = oracle.triangular_solve(T.shape, T.indptr, T.indices, T.data, b, upper=upper)
        assert np.allclose(x, spl.spsolve_triangular(T, b, lower=not upper), rtol=1e-10, atol=1e-12)
        x2 = oracle.triangular_solve(T.shape, T.indptr, T.indices, T.data, b, upper=upper, scale_a=2.0)
        assert np.allclose(2.0 * x2, x, rtol=1e-12)
    with pytest.raises(ValueError):
        oracle.triangular_solve((n, n), A.indptr, A.indices, A.data, b[:-1])
