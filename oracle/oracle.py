"""ctypes/numpy binding of oracle/spblas_oracle.c.

TEST INFRASTRUCTURE ONLY (see the header of spblas_oracle.c): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product
package.  Error codes map to the exception types the reference throws
(include/spblas/algorithms/multiply_impl.hpp:37-41 -> ValueError for
std::invalid_argument; spgemm_gustavsons.hpp:44-48 -> RuntimeError).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_void_p = ctypes.c_void_p


def build(native=False):
    """Compile the oracle with gcc (seconds).  native=True uses the reference's own
    flags (-O3 -march=native) and must be built on the machine that runs it."""
    target = "liboracle.native.so" if native else "liboracle.so"
    subprocess.run(["make", "-s", "-C", _HERE, target], check=True)
    return os.path.join(_HERE, target)


def load(native=False):
    key = bool(native)
    if key in _LIBS:
        return _LIBS[key]
    path = os.path.join(_HERE, "liboracle.native.so" if native else "liboracle.so")
    src = os.path.join(_HERE, "spblas_oracle.c")
    if native or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        if native and os.path.exists(path):
            os.remove(path)  # a native build from another machine must not be reused
        path = build(native)
    _LIBS[key] = ctypes.CDLL(path)
    return _LIBS[key]


def _raise(rc):
    if rc == 0:
        return
    if rc == 1:
        raise ValueError("multiply: matrix dimensions are incompatible.")
    if rc == 2:
        raise RuntimeError("multiply: SpGEMM ran out of memory.")
    raise MemoryError("oracle allocation failed")


def _p(a):
    return a.ctypes.data_as(c_void_p)


def _ct(dtype):
    return ctypes.c_float if dtype == np.float32 else ctypes.c_double


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def _csr_args(rowptr, colind, values):
    rowptr = np.ascontiguousarray(rowptr)
    colind = np.ascontiguousarray(colind, dtype=np.int32)
    values = np.ascontiguousarray(values)
    return rowptr, colind, values


def spmv(shape, rowptr, colind, values, x, y_len=None, scale_a=None, scale_x=None, native=False):
    """Reference SpMV y = A x (multiply_impl.hpp:33-53).  `shape` is A's shape;
    y_len defaults to shape[0]; a mismatch raises ValueError like the reference."""
    lib = load(native)
    rowptr, colind, values = _csr_args(rowptr, colind, values)
    x = np.ascontiguousarray(x, dtype=values.dtype)
    m = shape[0] if y_len is None else y_len
    y = np.empty(m, dtype=values.dtype)
    sfx = _sfx(values.dtype)
    if rowptr.dtype == np.int64:
        sfx += "_o64"
    else:
        rowptr = rowptr.astype(np.int32, copy=False)
    T = _ct(values.dtype)
    fn = getattr(lib, "oracle_spmv_" + sfx)
    fn.restype = c_int
    rc = fn(c_i64(m), c_i64(x.shape[0]), c_i64(shape[0]), c_i64(shape[1]), _p(rowptr), _p(colind),
            _p(values), c_int(scale_a is not None), T(0 if scale_a is None else scale_a), _p(x),
            c_int(scale_x is not None), T(0 if scale_x is None else scale_x), _p(y))
    _raise(rc)
    return y


def spmv_omp(rowptr, colind, values, x, native=False):
    lib = load(native)
    rowptr, colind, values = _csr_args(rowptr, colind, values)
    rowptr = rowptr.astype(np.int32, copy=False)
    x = np.ascontiguousarray(x, dtype=values.dtype)
    y = np.empty(rowptr.shape[0] - 1, dtype=values.dtype)
    fn = getattr(lib, "oracle_spmv_omp_" + _sfx(values.dtype))
    fn.restype = c_int
    fn(c_i64(y.shape[0]), _p(rowptr), _p(colind), _p(values), _p(x), _p(y))
    return y


def spmv_csc(shape, colptr, rowind, values, x, scale_a=None, scale_x=None):
    """y = A x with A in CSC (backend/algorithms.hpp:21-29 traversal)."""
    lib = load()
    colptr, rowind, values = _csr_args(colptr, rowind, values)
    colptr = colptr.astype(np.int32, copy=False)
    x = np.ascontiguousarray(x, dtype=values.dtype)
    y = np.empty(shape[0], dtype=values.dtype)
    T = _ct(values.dtype)
    fn = getattr(lib, "oracle_spmv_csc_" + _sfx(values.dtype))
    fn.restype = c_int
    rc = fn(c_i64(shape[0]), c_i64(x.shape[0]), c_i64(shape[0]), c_i64(shape[1]), _p(colptr),
            _p(rowind), _p(values), c_int(scale_a is not None), T(0 if scale_a is None else scale_a),
            _p(x), c_int(scale_x is not None), T(0 if scale_x is None else scale_x), _p(y))
    _raise(rc)
    return y


def spmm(shape, rowptr, colind, values, B, c_shape=None, scale_a=None, scale_b=None):
    """Reference SpMM C = A B (multiply_impl.hpp:66-92), B/C row-major 2-D arrays."""
    lib = load()
    rowptr, colind, values = _csr_args(rowptr, colind, values)
    rowptr = rowptr.astype(np.int32, copy=False)
    B = np.ascontiguousarray(B, dtype=values.dtype)
    if c_shape is None:
        c_shape = (shape[0], B.shape[1])
    C = np.empty(c_shape, dtype=values.dtype)
    T = _ct(values.dtype)
    fn = getattr(lib, "oracle_spmm_" + _sfx(values.dtype))
    fn.restype = c_int
    rc = fn(c_i64(c_shape[0]), c_i64(B.shape[0]), c_i64(c_shape[1]), c_i64(shape[0]), c_i64(shape[1]),
            c_i64(B.shape[0]), c_i64(B.shape[1]), _p(rowptr), _p(colind), _p(values),
            c_int(scale_a is not None), T(0 if scale_a is None else scale_a), _p(B),
            c_i64(B.shape[1]), c_int(scale_b is not None), T(0 if scale_b is None else scale_b),
            _p(C), c_i64(c_shape[1]), c_int(1))
    _raise(rc)
    return C


def spgemm_symbolic(a_shape, a_rowptr, a_colind, b_shape, b_rowptr, b_colind, c_shape=None):
    """multiply_compute (spgemm_gustavsons.hpp:57-89): returns (nnz, row_nnz[m])."""
    lib = load()
    a_rowptr = np.ascontiguousarray(a_rowptr, dtype=np.int32)
    a_colind = np.ascontiguousarray(a_colind, dtype=np.int32)
    b_rowptr = np.ascontiguousarray(b_rowptr, dtype=np.int32)
    b_colind = np.ascontiguousarray(b_colind, dtype=np.int32)
    if c_shape is None:
        c_shape = (a_shape[0], b_shape[1])
    row_nnz = np.zeros(a_shape[0], dtype=np.int64)
    nnz = c_i64(0)
    lib.oracle_spgemm_symbolic.restype = c_int
    rc = lib.oracle_spgemm_symbolic(c_i64(a_shape[0]), c_i64(a_shape[1]), c_i64(b_shape[1]),
                                    c_i64(c_shape[0]), c_i64(c_shape[1]), c_i64(b_shape[0]),
                                    _p(a_rowptr), _p(a_colind), _p(b_rowptr), _p(b_colind),
                                    _p(row_nnz), ctypes.byref(nnz))
    _raise(rc)
    return nnz.value, row_nnz


def spgemm_numeric(a_shape, a_rowptr, a_colind, a_values, b_shape, b_rowptr, b_colind, b_values,
                   capacity, c_shape=None, scale_a=None, scale_b=None):
    """multiply_fill -> multiply (spgemm_gustavsons.hpp:17-52): returns
    (rowptr, colind, values) with columns sorted ascending within each row."""
    lib = load()
    a_rowptr = np.ascontiguousarray(a_rowptr, dtype=np.int32)
    a_colind = np.ascontiguousarray(a_colind, dtype=np.int32)
    b_rowptr = np.ascontiguousarray(b_rowptr, dtype=np.int32)
    b_colind = np.ascontiguousarray(b_colind, dtype=np.int32)
    a_values = np.ascontiguousarray(a_values)
    b_values = np.ascontiguousarray(b_values, dtype=a_values.dtype)
    if c_shape is None:
        c_shape = (a_shape[0], b_shape[1])
    c_rowptr = np.zeros(c_shape[0] + 1, dtype=np.int32)
    c_colind = np.zeros(max(capacity, 1), dtype=np.int32)
    c_values = np.zeros(max(capacity, 1), dtype=a_values.dtype)
    T = _ct(a_values.dtype)
    nnz = c_i64(0)
    fn = getattr(lib, "oracle_spgemm_numeric_" + _sfx(a_values.dtype))
    fn.restype = c_int
    rc = fn(c_i64(a_shape[0]), c_i64(a_shape[1]), c_i64(b_shape[1]), c_i64(c_shape[0]),
            c_i64(c_shape[1]), c_i64(b_shape[0]), _p(a_rowptr), _p(a_colind), _p(a_values),
            c_int(scale_a is not None), T(0 if scale_a is None else scale_a), _p(b_rowptr),
            _p(b_colind), _p(b_values), c_int(scale_b is not None),
            T(0 if scale_b is None else scale_b), _p(c_rowptr), _p(c_colind), _p(c_values),
            c_i64(capacity), ctypes.byref(nnz))
    _raise(rc)
    return c_rowptr, c_colind[:nnz.value], c_values[:nnz.value]


def spgemm_symbolic_d(a_shape, a_rowptr, a_colind, b_shape, b_rowptr, b_colind, d_shape, d_rowptr, d_colind,
                      c_shape=None):
    """Structural count of C = A*B + D (spgemm_4args_test.cpp:78-95,108): returns (nnz, row_nnz[m])."""
    lib = load()
    arrs = [np.ascontiguousarray(x, dtype=np.int32) for x in (a_rowptr, a_colind, b_rowptr, b_colind, d_rowptr,
                                                               d_colind)]
    if c_shape is None:
        c_shape = (a_shape[0], b_shape[1])
    row_nnz = np.zeros(a_shape[0], dtype=np.int64)
    nnz = c_i64(0)
    lib.oracle_spgemm_symbolic_d.restype = c_int
    rc = lib.oracle_spgemm_symbolic_d(c_i64(a_shape[0]), c_i64(a_shape[1]), c_i64(b_shape[1]), c_i64(c_shape[0]),
                                      c_i64(c_shape[1]), c_i64(b_shape[0]), c_i64(d_shape[0]), c_i64(d_shape[1]),
                                      *[_p(x) for x in arrs], _p(row_nnz), ctypes.byref(nnz))
    _raise(rc)
    return nnz.value, row_nnz


def spgemm_numeric_d(a_shape, a_rowptr, a_colind, a_values, b_shape, b_rowptr, b_colind, b_values, d_shape,
                     d_rowptr, d_colind, d_values, capacity, alpha=1.0, beta=1.0, c_shape=None):
    """C = alpha*A*B + beta*D (vendor/rocsparse/multiply_spgemm.hpp:118-214; expected values
    spgemm_4args_test.cpp:78-95): returns (rowptr, colind, values), columns ascending."""
    lib = load()
    a_rowptr, a_colind, b_rowptr, b_colind, d_rowptr, d_colind = [
        np.ascontiguousarray(x, dtype=np.int32) for x in (a_rowptr, a_colind, b_rowptr, b_colind, d_rowptr, d_colind)]
    a_values = np.ascontiguousarray(a_values)
    b_values = np.ascontiguousarray(b_values, dtype=a_values.dtype)
    d_values = np.ascontiguousarray(d_values, dtype=a_values.dtype)
    if c_shape is None:
        c_shape = (a_shape[0], b_shape[1])
    c_rowptr = np.zeros(c_shape[0] + 1, dtype=np.int32)
    c_colind = np.zeros(max(capacity, 1), dtype=np.int32)
    c_values = np.zeros(max(capacity, 1), dtype=a_values.dtype)
    T = _ct(a_values.dtype)
    nnz = c_i64(0)
    fn = getattr(lib, "oracle_spgemm_numeric_d_" + _sfx(a_values.dtype))
    fn.restype = c_int
    rc = fn(c_i64(a_shape[0]), c_i64(a_shape[1]), c_i64(b_shape[1]), c_i64(c_shape[0]), c_i64(c_shape[1]),
            c_i64(b_shape[0]), c_i64(d_shape[0]), c_i64(d_shape[1]), _p(a_rowptr), _p(a_colind), _p(a_values),
            T(alpha), _p(b_rowptr), _p(b_colind), _p(b_values), T(beta), _p(d_rowptr), _p(d_colind), _p(d_values),
            _p(c_rowptr), _p(c_colind), _p(c_values), c_i64(capacity), ctypes.byref(nnz))
    _raise(rc)
    return c_rowptr, c_colind[:nnz.value], c_values[:nnz.value]


def add(shape, a_rowptr, a_colind, a_values, b_shape, b_rowptr, b_colind, b_values, capacity=None, c_shape=None,
        scale_a=None, scale_b=None, symbolic=False):
    """Reference add(a, b, c) (algorithms/add_impl.hpp:40-77): returns (rowptr, colind, values) of A + B with
    ascending columns.  symbolic=True is add_inspect (:79-108): returns (nnz, rowptr)."""
    lib = load()
    a_rowptr, a_colind, b_rowptr, b_colind = [np.ascontiguousarray(x, dtype=np.int32)
                                              for x in (a_rowptr, a_colind, b_rowptr, b_colind)]
    a_values = np.ascontiguousarray(a_values)
    b_values = np.ascontiguousarray(b_values, dtype=a_values.dtype)
    if c_shape is None:
        c_shape = tuple(shape)
    if capacity is None:
        capacity = int(a_rowptr[-1]) + int(b_rowptr[-1])
    c_rowptr = np.zeros(c_shape[0] + 1, dtype=np.int32)
    c_colind = np.zeros(max(capacity, 1), dtype=np.int32)
    c_values = np.zeros(max(capacity, 1), dtype=a_values.dtype)
    T = _ct(a_values.dtype)
    nnz = c_i64(0)
    fn = getattr(lib, "oracle_add_" + _sfx(a_values.dtype))
    fn.restype = c_int
    rc = fn(c_i64(shape[0]), c_i64(shape[1]), c_i64(b_shape[0]), c_i64(b_shape[1]), c_i64(c_shape[0]),
            c_i64(c_shape[1]), _p(a_rowptr), _p(a_colind), _p(a_values), c_int(scale_a is not None),
            T(0 if scale_a is None else scale_a), _p(b_rowptr), _p(b_colind), _p(b_values),
            c_int(scale_b is not None), T(0 if scale_b is None else scale_b), _p(c_rowptr),
            None if symbolic else _p(c_colind), None if symbolic else _p(c_values), c_i64(capacity),
            ctypes.byref(nnz))
    if rc == 2:
        raise RuntimeError("add: ran out of memory.  CSR output view has insufficient memory.")
    _raise(rc)
    if symbolic:
        return nnz.value, c_rowptr
    return c_rowptr, c_colind[:nnz.value], c_values[:nnz.value]


def triangular_solve(shape, rowptr, colind, values, b, upper=False, unit=False, scale_a=None, x_len=None):
    """Reference triangular_solve(a, uplo, diag, b, x) (algorithms/triangular_solve_impl.hpp:41-94)."""
    lib = load()
    rowptr, colind, values = _csr_args(rowptr, colind, values)
    rowptr = rowptr.astype(np.int32, copy=False)
    b = np.ascontiguousarray(b, dtype=values.dtype)
    x = np.zeros(shape[1] if x_len is None else x_len, dtype=values.dtype)
    T = _ct(values.dtype)
    fn = getattr(lib, "oracle_trsv_" + _sfx(values.dtype))
    fn.restype = c_int
    rc = fn(c_i64(shape[0]), c_i64(shape[1]), c_i64(b.shape[0]), c_i64(x.shape[0]), _p(rowptr), _p(colind),
            _p(values), c_int(scale_a is not None), T(0 if scale_a is None else scale_a), c_int(bool(upper)),
            c_int(bool(unit)), _p(b), _p(x))
    _raise(rc)
    return x


def spmv_absrow(rowptr, colind, values, x):
    """Per-row sum |a_v * x_k| in float64: the scale of the norm-wise tolerance."""
    lib = load()
    rowptr, colind, values = _csr_args(rowptr, colind, values)
    rowptr = rowptr.astype(np.int32, copy=False)
    x = np.ascontiguousarray(x, dtype=values.dtype)
    out = np.empty(rowptr.shape[0] - 1, dtype=np.float64)
    fn = getattr(lib, "oracle_spmv_absrow_" + _sfx(values.dtype))
    fn.restype = None
    fn(c_i64(out.shape[0]), _p(rowptr), _p(colind), _p(values), _p(x), _p(out))
    return out


def transpose(shape, rowptr, colind, values, b_shape=None, capacity=None):
    """Reference transpose(a, b) (algorithms/transpose_impl.hpp:14-53): returns CSR arrays of A^T."""
    lib = load()
    rowptr, colind, values = _csr_args(rowptr, colind, values)
    rowptr = rowptr.astype(np.int32, copy=False)
    m, n = shape
    if b_shape is None:
        b_shape = (n, m)
    nnz = int(rowptr[-1])
    if capacity is None:
        capacity = nnz
    t_rowptr = np.zeros(b_shape[0] + 1, dtype=np.int32)
    t_colind = np.zeros(max(capacity, 1), dtype=np.int32)
    t_values = np.zeros(max(capacity, 1), dtype=values.dtype)
    fn = getattr(lib, "oracle_transpose_" + _sfx(values.dtype))
    fn.restype = c_int
    rc = fn(c_i64(m), c_i64(n), c_i64(b_shape[0]), c_i64(b_shape[1]), _p(rowptr), _p(colind), _p(values),
            c_i64(capacity), _p(t_rowptr), _p(t_colind), _p(t_values))
    if rc == 2:
        raise RuntimeError("transpose: Transpose ran out of memory.")
    _raise(rc)
    return t_rowptr, t_colind[:nnz], t_values[:nnz]
