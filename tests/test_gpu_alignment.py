"""-m gpu: operands that start at odd element offsets and dense operands with a leading dimension.

The reference's views are spans over caller memory (/root/reference/include/spblas/views/csr_view.hpp:33-58 takes
any contiguous range; mdspan operands carry their own extents, examples/simple_spmm.cpp): nothing promises more than
element alignment, and a caller may hand in the middle of a larger allocation.  The kernels use 8- and 16-byte
accesses where the layout allows it, so every array here starts one or three elements into its allocation, the
dense matrices are column windows of wider ones (leading dimension != columns), and the row count is odd.
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

ALGS = {"noplan": None, "auto": _capi.SPMV_AUTO, "vector": _capi.SPMV_VECTOR, "rowblock": _capi.SPMV_ROWBLOCK,
        "sliced": _capi.SPMV_SLICED}


def shifted(host, k):
    """host array copied to the device k elements into a larger allocation (so its address is only element-aligned)."""
    t = torch.from_numpy(np.ascontiguousarray(host))
    buf = torch.empty(t.numel() + k + 5, dtype=t.dtype, device="cuda")
    view = buf[k:k + t.numel()]
    view.copy_(t)
    assert view.data_ptr() % 16 != 0 or k == 0
    return view


@pytest.mark.parametrize("alg", list(ALGS))
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("off64", [False, True])
def test_spmv_with_every_array_off_alignment(gpu, alg, dtype, off64):
    m, n, nnz = 20001, 30011, 400000
    values, rowptr, colind, shape, _ = generate.generate_csr(m, n, nnz, dtype=dtype, seed=31)
    rp = rowptr.astype(np.int64) if off64 else rowptr
    x_h = np.random.default_rng(3).standard_normal(n).astype(dtype)
    a = sp.csr_view(shifted(values, 1), shifted(rp, 1), shifted(colind, 3), shape, nnz)
    x = shifted(x_h, 1)
    ybuf = torch.full((m + 8,), float("nan"), dtype=x.dtype, device="cuda")
    y = ybuf[3:3 + m]
    if ALGS[alg] is None:
        sp.multiply(sp.scaled(1.5, a), x, y)
    else:
        info = sp.multiply_inspect(a, x, y, alg=ALGS[alg])
        sp.multiply(info, sp.scaled(1.5, a), x, y)
    torch.cuda.synchronize()
    y_ref = oracle.spmv(shape, rowptr, colind, values, x_h, scale_a=1.5)
    absrow = 1.5 * oracle.spmv_absrow(rowptr, colind, values, x_h)
    util.assert_parity(G.host(y), y_ref, absrow, dtype, row_len=np.diff(rowptr), what=f"{alg} off-alignment")
    # nothing outside y was written
    assert torch.isnan(ybuf[:3]).all() and torch.isnan(ybuf[3 + m:]).all()


@pytest.mark.parametrize("ncols", [1, 3, 8, 33, 128])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("inspect", [False, True])
def test_spmm_on_column_windows_of_wider_matrices(gpu, ncols, dtype, inspect):
    m, k, nnz = 5001, 7003, 120000
    values, rowptr, colind, shape, _ = generate.generate_csr(m, k, nnz, dtype=dtype, seed=32)
    a = sp.csr_view(shifted(values, 3), shifted(rowptr, 1), shifted(colind, 1), shape, nnz)
    B_h = np.random.default_rng(4).standard_normal((k, ncols)).astype(dtype)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    Bw = torch.zeros((k, ncols + 7), dtype=tdt, device="cuda")
    B = Bw[:, 3:3 + ncols]
    B.copy_(torch.from_numpy(B_h))
    Cw = torch.full((m, ncols + 5), float("nan"), dtype=tdt, device="cuda")
    C = Cw[:, 1:1 + ncols]
    if inspect:
        info = sp.multiply_inspect(a, B, C)
        sp.multiply(info, a, B, C)
    else:
        sp.multiply(a, B, C)
    torch.cuda.synchronize()
    ref = oracle.spmm(shape, rowptr, colind, values, B_h)
    scale = oracle.spmm(shape, rowptr, colind, np.abs(values), np.abs(B_h))
    tol = util.TOL[np.dtype(dtype)]
    assert (np.abs(C.cpu().numpy() - ref) <= tol * scale + 1e-30).all()
    assert torch.isnan(Cw[:, :1]).all() and torch.isnan(Cw[:, 1 + ncols:]).all()  # the window's neighbours are untouched


def test_spgemm_and_transpose_with_arrays_off_alignment(gpu):
    m, k, n = 3001, 2503, 2005
    av, ar, ac, ash, annz = generate.generate_csr(m, k, 40000, seed=33)
    bv, br, bc, bsh, bnnz = generate.generate_csr(k, n, 30000, seed=34)
    a = sp.csr_view(shifted(av, 1), shifted(ar, 3), shifted(ac, 1), ash, annz)
    b = sp.csr_view(shifted(bv, 3), shifted(br, 1), shifted(bc, 3), bsh, bnnz)
    c_rp = shifted(np.zeros(m + 1, np.int32), 1)
    c = sp.csr_view(None, c_rp, None, (m, n), 0)
    info = sp.multiply_compute(a, b, c)
    cn = info.result_nnz()
    c_val, c_col = shifted(np.zeros(cn, np.float32), 1), shifted(np.zeros(cn, np.int32), 3)
    c.update(c_val, c_rp, c_col, (m, n), cn)
    sp.multiply_fill(info, a, b, c)
    torch.cuda.synchronize()
    ref_n, _ = oracle.spgemm_symbolic(ash, ar, ac, bsh, br, bc)
    cr, cc, cv = oracle.spgemm_numeric(ash, ar, ac, av, bsh, br, bc, bv, capacity=ref_n)
    assert cn == ref_n and np.array_equal(G.host(c_rp), cr) and np.array_equal(G.host(c_col), cc)
    np.testing.assert_allclose(G.host(c_val), cv, rtol=2e-5)
