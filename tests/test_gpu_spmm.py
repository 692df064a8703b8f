"""-m gpu parity tests for CSR x dense SpMM (row-major B, C) vs the CPU oracle.
Cases mirror /root/reference/test/gtest/spmm_test.cpp:6-136 (n in {1,8,32,64,512} on
util::dims; scaled A / scaled B with alpha in {-10,1,5}; matrix_opt :138) -- the reference
has no device SpMM test, its CPU cases are the model."""
import os

import numpy as np
import pytest
import scipy.sparse as sps
import torch

import gpu_util as G
import spblas_reference_amd as sp
import util
from oracle import oracle
from spblas_reference_amd import generate

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def absprod(values, rowptr, colind, shape, B):
    A = sps.csr_matrix((np.abs(values).astype(np.float64), colind, rowptr), shape=shape)
    return A @ np.abs(B).astype(np.float64)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n", [1, 8, 32, 64, 512])
@pytest.mark.parametrize("dim", util.dims)
def test_spmm_reference_cases(gpu, dim, n, dtype):
    m, k, nnz = dim
    values, rowptr, colind, shape, _ = generate.generate_csr(m, k, nnz, dtype=dtype)
    B = generate.generate_dense(k, n, dtype=dtype)
    a = G.csr_on_device(values, rowptr, colind, shape, nnz)
    C = torch.full((m, n), float("nan"), dtype=G.dev(B).dtype, device="cuda")
    info = sp.multiply_inspect(a, G.dev(B), C)  # examples/spmm_csr.cpp:45-46 call shape
    sp.multiply(info, a, G.dev(B), C)
    C_ref = oracle.spmm(shape, rowptr, colind, values, B)
    lens = np.diff(rowptr)
    util.assert_parity(G.host(C), C_ref, absprod(values, rowptr, colind, shape, B), dtype, row_len=lens,
                       what=f"spmm {dim} n={n}")
    util.expect_eq_ref(C_ref, G.host(C))


@pytest.mark.parametrize("alpha", [-10, 1, 5])
def test_spmm_scaled_and_matrix_opt(gpu, alpha):
    values, rowptr, colind, shape, nnz = generate.generate_csr(100, 1000, 10000)
    B = generate.generate_dense(1000, 64)
    a = G.csr_on_device(values, rowptr, colind, shape, nnz)
    lens = np.diff(rowptr)
    ab = absprod(values, rowptr, colind, shape, B) * abs(alpha)
    for A_arg, B_arg, kw in ((sp.scaled(alpha, a), G.dev(B), {"scale_a": alpha}),
                             (a, sp.scaled(alpha, G.dev(B)), {"scale_b": alpha}),
                             (sp.scaled(alpha, sp.matrix_opt(a)), G.dev(B), {"scale_a": alpha})):
        C = torch.zeros((100, 64), device="cuda")
        sp.multiply(A_arg, B_arg, C)
        util.assert_parity(G.host(C), oracle.spmm(shape, rowptr, colind, values, B, **kw), ab, np.float32,
                           row_len=lens, what=f"spmm scaled {kw}")


@pytest.mark.parametrize("name", sorted(f for f in os.listdir(GOLDEN) if f.startswith("spmm_")))
def test_spmm_golden_bit_exact(gpu, name):
    g = np.load(os.path.join(GOLDEN, name))
    a = G.csr_on_device(g["values"], g["rowptr"], g["colind"], tuple(g["shape"]), len(g["values"]))
    C = torch.full(g["C"].shape, float("nan"), device="cuda")
    sp.multiply(a, G.dev(g["B"]), C)
    assert np.array_equal(G.host(C), g["C"])


def test_spmm_strided_operands_and_errors(gpu):
    values, rowptr, colind, shape, nnz = generate.generate_csr(64, 80, 900, seed=2)
    a = G.csr_on_device(values, rowptr, colind, shape, nnz)
    Bbig = G.dev(generate.generate_dense(80, 40))
    Cbig = torch.zeros((64, 48), device="cuda")
    Bv, Cv = Bbig[:, :24], Cbig[:, 8:32]  # row stride > n (mdspan with padding)
    sp.multiply(a, Bv, Cv)
    C_ref = oracle.spmm(shape, rowptr, colind, values, G.host(Bv.contiguous()))
    util.assert_parity(G.host(Cv.contiguous()), C_ref,
                       absprod(values, rowptr, colind, shape, G.host(Bv.contiguous())), np.float32,
                       row_len=np.diff(rowptr), what="strided spmm")
    assert float(Cbig[:, :8].abs().sum()) == 0 and float(Cbig[:, 32:].abs().sum()) == 0
    with pytest.raises(ValueError):  # multiply_impl.hpp:70-76
        sp.multiply(a, torch.zeros((79, 8), device="cuda"), torch.zeros((64, 8), device="cuda"))
    with pytest.raises(ValueError):
        sp.multiply(a, torch.zeros((80, 8), device="cuda").t().contiguous().t(), torch.zeros((64, 8), device="cuda"))


def test_spmm_cfg3_shape_properties(gpu):
    """BASELINE cfg3 at reduced row count but full n = 128 panel: linearity in B and a
    sampled-row oracle check (the full 2M x 2M case is bench.py --workload spmm)."""
    m = k = 200_000
    values, rowptr, colind, shape, nnz = generate.uniform_csr_device(m, k, 32, seed=0)
    a = sp.csr_view(values, rowptr, colind, shape, nnz)
    g = torch.Generator(device="cuda").manual_seed(2)
    B1 = torch.rand((k, 128), device="cuda", generator=g)
    B2 = torch.rand((k, 128), device="cuda", generator=g)
    C1, C2, C3 = (torch.empty((m, 128), device="cuda") for _ in range(3))
    sp.multiply(a, B1, C1)
    sp.multiply(a, B2, C2)
    sp.multiply(a, B1 - 3.0 * B2, C3)
    lin = (C3 - (C1 - 3.0 * C2)).abs()
    assert bool((lin <= 8e-6 * (C1.abs() + 3.0 * C2.abs()) + 1e-30).all())
    rows = np.arange(0, m, 997)
    rp = rowptr.cpu().numpy()
    idx = torch.from_numpy(np.concatenate([np.arange(rp[r], rp[r + 1]) for r in rows])).cuda()
    sub_rp = np.concatenate([[0], np.cumsum(rp[rows + 1] - rp[rows])]).astype(np.int32)
    sub_c, sub_v = colind[idx].cpu().numpy(), values[idx].cpu().numpy()
    B_h = B1.cpu().numpy()
    C_ref = oracle.spmm((len(rows), k), sub_rp, sub_c, sub_v, B_h)
    util.assert_parity(C1[torch.from_numpy(rows).cuda()].cpu().numpy(), C_ref,
                       absprod(sub_v, sub_rp, sub_c, (len(rows), k), B_h), np.float32, what="cfg3 sampled rows")
