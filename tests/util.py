"""Shared test helpers: the reference's comparator and shapes, restated.

EXPECT_EQ_  /root/reference/test/gtest/util.hpp:7-23 :
    |t-u| <= max(FLT_MIN, 64*eps*min(|t|+|u|, FLT_MAX))   (floating point), exact otherwise
dims        util.hpp:27-29, square_dims util.hpp:31-33
Parity bound of this build (BASELINE.json north_star): norm-wise,
|y_gpu - y_ref| <= TOL * sum_p |a_p * x_p| per row, TOL = 1e-6 (fp32) / 1e-12 (fp64).
"""
import numpy as np

dims = [(1000, 100, 100), (100, 1000, 10000), (40, 40, 1000)]
square_dims = [(1000, 1000, 100), (100, 100, 100), (40, 40, 1000)]

TOL = {np.dtype(np.float32): 1e-6, np.dtype(np.float64): 1e-12}


def expect_eq_ref(t, u):
    """Vectorised EXPECT_EQ_ (util.hpp:7-23)."""
    t = np.asarray(t)
    u = np.asarray(u)
    fi = np.finfo(t.dtype)
    norm = np.minimum(np.abs(t).astype(np.float64) + np.abs(u).astype(np.float64), float(fi.max))
    abs_error = np.maximum(float(fi.tiny), 64 * float(fi.eps) * norm)
    bad = np.abs(t.astype(np.float64) - u.astype(np.float64)) > abs_error
    assert not bad.any(), f"EXPECT_EQ_ failed at {np.flatnonzero(bad.ravel())[:8]}"


def naive_spmv(rowptr, colind, values, b, alpha_a=None, alpha_b=None):
    """The inline comparator loop of the reference tests
    (test/gtest/spmv_test.cpp:23-30,57-64,93-100; device/spmv_test.cpp:38-46)."""
    m = len(rowptr) - 1
    c_ref = np.zeros(m, dtype=values.dtype)
    T = values.dtype.type
    for i in range(m):
        acc = T(0)
        for j_ptr in range(rowptr[i], rowptr[i + 1]):
            j = colind[j_ptr]
            v = values[j_ptr]
            if alpha_a is not None:
                acc = T(acc + T(T(alpha_a) * v) * b[j])      # c_ref[i] += alpha * v * b[j]
            elif alpha_b is not None:
                acc = T(acc + T(v * T(alpha_b)) * b[j])      # c_ref[i] += v * alpha * b[j]
            else:
                acc = T(acc + v * b[j])
        c_ref[i] = acc
    return c_ref


def spmv_exact(rowptr, colind, values, x):
    """float64 product and the per-row norm sum |a*x| (the scale of the parity bound)."""
    import scipy.sparse as sps
    m = len(rowptr) - 1
    n = x.shape[0]
    A = sps.csr_matrix((values.astype(np.float64), colind, rowptr), shape=(m, n))
    Aabs = sps.csr_matrix((np.abs(values).astype(np.float64), colind, rowptr), shape=(m, n))
    return A @ x.astype(np.float64), Aabs @ np.abs(x).astype(np.float64)


def assert_parity(y, y_ref, absrow, dtype, row_len=None, what=""):
    """Norm-wise parity.  The reference accumulates sequentially in T, so its own result
    carries up to (k/2)*eps*sum|.| of rounding for a k-entry row; the bound is never
    tighter than that (only matters for rows far longer than BASELINE's 10-32)."""
    dt = np.dtype(dtype)
    tol = np.full(absrow.shape, TOL[dt])
    if row_len is not None:
        tol = np.maximum(tol, 0.5 * np.asarray(row_len, dtype=np.float64).reshape(
            (-1,) + (1,) * (absrow.ndim - 1)) * float(np.finfo(dt).eps))
    err = np.abs(y.astype(np.float64) - y_ref.astype(np.float64))
    bound = tol * absrow + float(np.finfo(dt).tiny)
    bad = ~(err <= bound)  # NaN (an output the kernel never wrote) must fail, not slip through
    assert not bad.any(), (f"{what}: {bad.sum()} entries exceed the parity bound; worst ratio "
                           f"{(err / np.maximum(bound, 1e-300)).max():.3g}")
