"""-m gpu: the execute half of the inspect / execute split inside a HIP graph.

An iterative solver calls multiply(info, A, x, y) thousands of times with one plan (the call shape of
/root/reference/examples/device/matrix_opt_example.cpp and of operation_info_t in
/root/reference/include/spblas/detail/operation_info_t.hpp); on this backend the natural way to run such a loop is a
captured graph.  That only works if execute is capturable: kernel launches and memsets on the handle's stream, no
allocation, no host synchronisation, no host readback.  These tests capture one call, replay it on new right-hand
sides and compare every replay with the oracle.
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

SPMV_ALGS = {"vector": _capi.SPMV_VECTOR, "rowblock": _capi.SPMV_ROWBLOCK, "sliced": _capi.SPMV_SLICED}


def capture(fn):
    """Warm up on a side stream (the usual torch recipe), then capture one call of fn."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


@pytest.mark.parametrize("alg", list(SPMV_ALGS))
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_spmv_execute_is_capturable(gpu, alg, dtype):
    m, n, nnz = 30000, 50000, 600000
    values, rowptr, colind, shape, _ = generate.generate_csr(m, n, nnz, dtype=dtype, seed=11)
    a = sp.csr_view(G.dev(values), G.dev(rowptr), G.dev(colind), shape, nnz)
    x = torch.zeros(n, dtype=G.dev(values).dtype, device="cuda")
    y = torch.full((m,), float("nan"), dtype=x.dtype, device="cuda")
    info = sp.multiply_inspect(a, x, y, alg=SPMV_ALGS[alg])
    assert info.state_.info()["alg"] == SPMV_ALGS[alg]
    g = capture(lambda: sp.multiply(info, sp.scaled(0.5, a), x, y))
    lens = np.diff(rowptr)
    for seed in range(3):
        x_h = np.random.default_rng(seed).standard_normal(n).astype(dtype)
        x.copy_(G.dev(x_h))
        y.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        y_ref = oracle.spmv(shape, rowptr, colind, values, x_h, scale_a=0.5)
        absrow = 0.5 * oracle.spmv_absrow(rowptr, colind, values, x_h)
        util.assert_parity(G.host(y), y_ref, absrow, dtype, row_len=lens, what=f"graph replay {seed} ({alg})")


def test_transposed_multiply_without_a_plan_is_capturable(gpu, monkeypatch):
    """y = A^T x on a csc_view that was never inspected: outside a capture large operands go through a workspace in the
    handle's scratch (two-pass form, csrc/spmv.hip t2_*), which a later call may re-allocate -- so a RECORDED call keeps the
    scatter kernel (scale + one float atomic per entry: launches only).  The two-pass form is forced for this small matrix
    (SPBLAS_GFX950_SPMV_T2=1) to show that the capture does not take it; a larger un-captured call between the replays grows
    the scratch, and the replays stay right."""
    monkeypatch.setenv("SPBLAS_GFX950_SPMV_T2", "1")
    m, n, nnz = 30000, 50000, 600000
    values, rowptr, colind, shape, _ = generate.generate_csr(m, n, nnz, dtype=np.float32, seed=12)
    a = sp.csr_view(G.dev(values), G.dev(rowptr), G.dev(colind), shape, nnz)
    x = torch.zeros(m, device="cuda")
    y = torch.full((n,), float("nan"), device="cuda")
    g = capture(lambda: sp.multiply(sp.scaled(0.5, sp.transposed(a)), x, y))
    cnt = np.bincount(colind, minlength=n) + 1
    # a bigger product outside the capture: the handle's scratch is released and allocated again
    v2, rp2, ci2, shape2, _ = generate.generate_csr(4 * m, n, 8 * nnz, dtype=np.float32, seed=13)
    a2 = sp.csr_view(G.dev(v2), G.dev(rp2), G.dev(ci2), shape2, 8 * nnz)
    for seed in range(3):
        x_h = np.random.default_rng(seed).standard_normal(m).astype(np.float32)
        x.copy_(G.dev(x_h))
        y.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        ref = 0.5 * oracle.spmv_csc((n, m), rowptr, colind, values, x_h)
        ab = 0.5 * oracle.spmv_csc((n, m), rowptr, colind, np.abs(values), np.abs(x_h))
        util.assert_parity(G.host(y), ref, ab, np.float32, row_len=cnt, what=f"graph replay {seed} of an un-inspected A^T x")
        if seed == 0:
            y2 = torch.empty(n, device="cuda")
            sp.multiply(sp.transposed(a2), torch.ones(4 * m, device="cuda"), y2)
            torch.cuda.synchronize()


def test_spmv_rmat_row_map_plan_is_capturable(gpu):
    # the cfg4 plan shape: row map, long rows in pieces, variable-height bins, empty-row fill -- five launches per call
    v, rp, ci, sh, nnz = generate.rmat_csr_device(14, 16, dtype=torch.float64, device="cuda", seed=9)
    a = sp.csr_view(v, rp, ci, sh, nnz)
    x = torch.zeros(sh[1], dtype=torch.float64, device="cuda")
    y = torch.full((sh[0],), float("nan"), dtype=torch.float64, device="cuda")
    info = sp.multiply_inspect(a, x, y, alg=_capi.SPMV_SLICED)
    assert info.state_.sliced_info()["variable_bins"] == 1
    g = capture(lambda: sp.multiply(info, a, x, y))
    rp_h, ci_h, v_h = G.host(rp), G.host(ci), G.host(v)
    for seed in range(2):
        x_h = np.random.default_rng(seed).random(sh[1])
        x.copy_(G.dev(x_h))
        y.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        util.assert_parity(G.host(y), oracle.spmv(sh, rp_h, ci_h, v_h, x_h), oracle.spmv_absrow(rp_h, ci_h, v_h, x_h),
                           np.float64, row_len=np.diff(rp_h), what=f"R-MAT graph replay {seed}")


@pytest.mark.parametrize("ncols", [8, 128])
def test_spmm_execute_is_capturable(gpu, ncols):
    m, k, nnz = 20000, 15000, 400000
    values, rowptr, colind, shape, _ = generate.generate_csr(m, k, nnz, seed=12)
    a = sp.csr_view(G.dev(values), G.dev(rowptr), G.dev(colind), shape, nnz)
    B = torch.zeros((k, ncols), device="cuda")
    C = torch.full((m, ncols), float("nan"), device="cuda")
    info = sp.multiply_inspect(a, B, C)
    g = capture(lambda: sp.multiply(info, a, B, C))
    for seed in range(2):
        B_h = np.random.default_rng(seed).random((k, ncols)).astype(np.float32)
        B.copy_(G.dev(B_h))
        C.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        np.testing.assert_allclose(G.host(C), oracle.spmm(shape, rowptr, colind, values, B_h), rtol=2e-5)


def test_spgemm_numeric_refill_is_capturable(gpu):
    # multiply_fill on a result whose structure is known (the reuse call shape of the reference's
    # test/gtest/device/spgemm_reuse_test.cpp): values of A and B change between replays, the structure does not
    m, k, n = 3000, 2500, 2000
    av, ar, ac, ash, annz = generate.generate_csr(m, k, 40000, seed=3)
    bv, br, bc, bsh, bnnz = generate.generate_csr(k, n, 30000, seed=4)
    a_val, b_val = G.dev(av), G.dev(bv)
    a = sp.csr_view(a_val, G.dev(ar), G.dev(ac), ash, annz)
    b = sp.csr_view(b_val, G.dev(br), G.dev(bc), bsh, bnnz)
    c_rp = torch.zeros(m + 1, dtype=torch.int32, device="cuda")
    c = sp.csr_view(None, c_rp, None, (m, n), 0)
    info = sp.multiply_compute(a, b, c)
    cn = info.result_nnz()
    c_val = torch.zeros(cn, device="cuda")
    c.update(c_val, c_rp, torch.zeros(cn, dtype=torch.int32, device="cuda"), (m, n), cn)
    sp.multiply_fill(info, a, b, c)  # first fill writes the structure
    sp.multiply_fill(info, a, b, c)
    torch.cuda.synchronize()
    try:
        g = capture(lambda: sp.multiply_fill(info, a, b, c))
    except RuntimeError as e:  # a fill that reads back or allocates cannot be captured: say so instead of guessing
        pytest.fail(f"multiply_fill on a known structure is not capturable: {e}")
    for seed in range(2):
        rng = np.random.default_rng(seed)
        av2, bv2 = rng.random(annz).astype(np.float32), rng.random(bnnz).astype(np.float32)
        a_val.copy_(G.dev(av2))
        b_val.copy_(G.dev(bv2))
        c_val.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        _, _, cv = oracle.spgemm_numeric(ash, ar, ac, av2, bsh, br, bc, bv2, capacity=cn)
        np.testing.assert_allclose(G.host(c_val), cv, rtol=2e-5)


@pytest.mark.parametrize("upper", [False, True])
def test_triangular_solve_with_inspect_is_capturable(gpu, monkeypatch, upper):
    """A recorded solve is replayed with the arguments it was recorded with, so inside a capture the library (a) does not
    use the cooperative kernel (its graph node carries no co-residency guarantee: replays left rows unsolved) but one
    launch per level group, and (b) allocates nothing (csrc/sptrsv.hip: trsv_solve_typed).  Replays and ordinary solves
    alternate here on purpose."""
    import scipy.sparse as sps
    rng = np.random.default_rng(21)
    n, k = 30000, 6
    rows = np.repeat(np.arange(n), k)
    cols = (rng.random(n * k) * rows).astype(np.int64)
    keep = cols < rows
    S = sps.csr_matrix(((rng.random(keep.sum()) - 0.5) * (0.5 / k), (rows[keep], cols[keep])), shape=(n, n))
    M = (S + sps.diags(1.0 + rng.random(n))).tocsr()
    if upper:
        M = M.T.tocsr()
    vals = M.data.astype(np.float32)
    rp, ci = M.indptr.astype(np.int32), M.indices.astype(np.int32)
    a = G.csr_on_device(vals, rp, ci, M.shape, M.nnz)
    b = torch.zeros(n, device="cuda")
    x = torch.full((n,), float("nan"), device="cuda")
    uplo = sp.upper_triangle if upper else sp.lower_triangle
    info = sp.triangular_solve_inspect(a, uplo, sp.explicit_diagonal, b, x)
    assert info.state_.info()["levels"] > 20
    solve = lambda: sp.triangular_solve(info, a, uplo, sp.explicit_diagonal, b, x)
    g = capture(solve)
    for seed in range(6):
        b_h = (np.random.default_rng(seed).random(n) + 0.5).astype(np.float32)
        b.copy_(G.dev(b_h))
        x.fill_(float("nan"))
        if seed % 3 == 2:
            solve()  # an ordinary solve between replays
        else:
            g.replay()
        torch.cuda.synchronize()
        ref = oracle.triangular_solve(M.shape, rp, ci, vals, b_h, upper=upper, unit=False)
        got = G.host(x)
        assert np.all(np.isfinite(got)), f"replay {seed}: {np.count_nonzero(~np.isfinite(got))} rows unsolved"
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=1e-6)


def test_first_triangular_solve_of_a_plan_cannot_be_recorded(gpu):
    # its control words are allocated by the first solve; inside a capture that is refused, not recorded
    import scipy.sparse as sps
    n = 2000
    M = (sps.tril(sps.random(n, n, density=0.01, format="csr", random_state=np.random.default_rng(2)), -1)
         + sps.diags(np.full(n, 2.0))).tocsr()
    a = G.csr_on_device(M.data.astype(np.float32), M.indptr.astype(np.int32), M.indices.astype(np.int32), M.shape, M.nnz)
    b = torch.ones(n, device="cuda")
    x = torch.zeros(n, device="cuda")
    info = sp.triangular_solve_inspect(a, sp.lower_triangle, sp.explicit_diagonal, b, x)
    g = torch.cuda.CUDAGraph()
    with pytest.raises(Exception):
        with torch.cuda.graph(g):
            sp.triangular_solve(info, a, sp.lower_triangle, sp.explicit_diagonal, b, x)
    torch.cuda.synchronize()
    sp.triangular_solve(info, a, sp.lower_triangle, sp.explicit_diagonal, b, x)  # the plan is still usable
    torch.cuda.synchronize()
    assert torch.isfinite(x).all()


def test_spmm_with_long_rows_needs_one_call_outside_the_capture(gpu):
    # the partial rows of the long-row kernel are sized by the column count of the first call; inside a capture that
    # growth is refused (a buffer allocated there would belong to the graph), after one ordinary call it records fine
    m, k, ncols = 3000, 40000, 16
    values, rowptr, colind, shape, nnz = generate.generate_csr(m, k, 60000, seed=5)
    long_row = np.arange(0, k, 2, dtype=np.int32)  # one row of 20 000 entries
    rowptr = np.concatenate([rowptr, [rowptr[-1] + long_row.size]]).astype(rowptr.dtype)
    colind = np.concatenate([colind, long_row]).astype(np.int32)
    values = np.concatenate([values, np.full(long_row.size, 0.25, dtype=np.float32)])
    shape, nnz = (m + 1, k), int(rowptr[-1])
    a = sp.csr_view(G.dev(values), G.dev(rowptr), G.dev(colind), shape, nnz)
    B = torch.rand((k, ncols), device="cuda")
    C = torch.full((m + 1, ncols), float("nan"), device="cuda")
    info = sp.multiply_inspect(a, B, C)
    assert info.state_.spmm_info()["long_rows"] >= 1
    g = torch.cuda.CUDAGraph()
    with pytest.raises(Exception):
        with torch.cuda.graph(g):
            sp.multiply(info, a, B, C)
    torch.cuda.synchronize()
    g = capture(lambda: sp.multiply(info, a, B, C))  # warms up outside, then records
    B_h = np.random.default_rng(1).random((k, ncols)).astype(np.float32)
    B.copy_(G.dev(B_h))
    C.fill_(float("nan"))
    g.replay()
    torch.cuda.synchronize()
    np.testing.assert_allclose(G.host(C), oracle.spmm(shape, rowptr, colind, values, B_h), rtol=5e-5)


def test_inspect_class_calls_are_refused_inside_a_capture(gpu):
    """multiply_inspect, multiply_compute, transpose and triangular_solve_inspect size their results on the host and
    allocate: on a capturing stream they return NOT_SUPPORTED at once (nothing is recorded, the capture stays valid) and
    work again right after it."""
    values, rowptr, colind, shape, nnz = generate.generate_csr(4000, 5000, 60000, seed=71)
    a = sp.csr_view(G.dev(values), G.dev(rowptr), G.dev(colind), shape, nnz)
    x = torch.rand(5000, device="cuda")
    y = torch.zeros(4000, device="cuda")
    t = sp.csr_view(torch.zeros(nnz, device="cuda"), torch.zeros(5001, dtype=torch.int32, device="cuda"),
                    torch.zeros(nnz, dtype=torch.int32, device="cuda"), (5000, 4000), nnz)
    c_rp = torch.zeros(4001, dtype=torch.int32, device="cuda")
    sp.multiply(a, x, y)  # the thread's handle exists before the capture
    torch.cuda.synchronize()
    refused = []
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for name, fn in (("multiply_inspect", lambda: sp.multiply_inspect(a, x, y, alg=_capi.SPMV_SLICED)),
                         ("transpose", lambda: sp.transpose(a, t)),
                         ("multiply_compute", lambda: sp.multiply_compute(a, sp.csr_view(t.values(), t.rowptr(), t.colind(),
                                                                                         (5000, 4000), nnz),
                                                                          sp.csr_view(None, c_rp, None, (4000, 4000), 0)))):
            try:
                fn()
            except Exception as e:  # noqa: BLE001
                refused.append((name, str(e)))
        sp.multiply(a, x, y)  # the plan-free execute IS recordable
    assert [n for n, _ in refused] == ["multiply_inspect", "transpose", "multiply_compute"], refused
    assert all("not supported" in msg.lower() for _, msg in refused), refused
    x.copy_(torch.rand(5000, device="cuda"))
    g.replay()
    torch.cuda.synchronize()
    util.assert_parity(G.host(y), oracle.spmv(shape, rowptr, colind, values, G.host(x)),
                       oracle.spmv_absrow(rowptr, colind, values, G.host(x)), np.float32, row_len=np.diff(rowptr),
                       what="plan-free SpMV recorded next to refused inspect calls")
    info = sp.multiply_inspect(a, x, y, alg=_capi.SPMV_SLICED)  # and outside the capture everything works as before
    sp.multiply(info, a, x, y)
    sp.transpose(a, t)
    torch.cuda.synchronize()


def test_cg_example_with_the_iteration_in_a_graph(gpu):
    # examples/cg_graph.py: SpMV + dot products + updates of one CG iteration recorded once, replayed 60 times
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "cg_graph.py")
    spec = importlib.util.spec_from_file_location("cg_graph_example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(100_000, 60) < 1e-8
