"""-m gpu parity tests for triangular_solve / triangular_solve_inspect against the
CPU oracle (restatement of algorithms/triangular_solve_impl.hpp:41-94).
Cases: the reference's own test shape (test/gtest/triangular_solve_test.cpp:63-104: general random
matrix, b = 0, values scaled by 1e-3, implicit unit diagonal), then real triangular systems with
explicit / unit diagonals, upper / lower, general matrices whose other triangle must be ignored,
scaled(alpha, a), deep chains (one level per row), wide levels, long rows, fp32 / fp64.
Parity: the device sums a row G lanes wide, the reference sequentially -> norm-wise bound on every
x_i, propagated through the solve by comparing residuals as well."""
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


def _tags(upper, unit):
    return (sp.upper_triangle if upper else sp.lower_triangle,
            sp.implicit_unit_diagonal if unit else sp.explicit_diagonal)


def device_solve(M, b, upper, unit, scale_a=None, inspect=True, dtype=np.float32):
    M = M.tocsr()
    vals = M.data.astype(dtype)
    d_a = G.csr_on_device(vals, M.indptr.astype(np.int32), M.indices.astype(np.int32), M.shape, M.nnz)
    A = sp.scaled(scale_a, d_a) if scale_a is not None else d_a
    d_b = G.dev(b.astype(dtype))
    d_x = torch.full((M.shape[1],), float("nan"), dtype=d_b.dtype, device="cuda")
    uplo, diag = _tags(upper, unit)
    if inspect:
        info = sp.triangular_solve_inspect(A, uplo, diag, d_b, d_x)
        sp.triangular_solve(info, A, uplo, diag, d_b, d_x)
        return G.host(d_x), info.state_.info()
    sp.triangular_solve(A, uplo, diag, d_b, d_x)
    return G.host(d_x), None


def check(M, b, upper, unit, dtype, scale_a=None, inspect=True):
    """(1) Norm-wise backward error of every row, the same bound the SpMV parity uses: the computed x_i
    must satisfy |b_i - (T x)_i| <= tol * (|b_i| + sum_k |t_ik| |x_k|), tol = 1e-6 (fp32) / 1e-12 (fp64)
    (never tighter than the k*eps/2 a k-term sum carries) -- T being the triangle the reference reads.
    (2) Forward error against the oracle; rounding differences are amplified by the conditioning of the
    solve, so this bound is 100x looser (the systems used here are diagonally dominant)."""
    M = M.tocsr()
    n = M.shape[0]
    x, info = device_solve(M, b, upper, unit, scale_a, inspect, dtype)
    Md = M.astype(dtype).astype(np.float64) * (1.0 if scale_a is None else float(dtype(scale_a)))
    ref = oracle.triangular_solve(M.shape, M.indptr, M.indices, M.data.astype(dtype), b.astype(dtype), upper=upper,
                                  unit=unit, scale_a=scale_a)
    assert np.all(np.isfinite(x))
    T = (sps.triu(Md, 1) if upper else sps.tril(Md, -1)) + (sps.eye(n) if unit else sps.diags(Md.diagonal()))
    T = T.tocsr()
    bd, xd = b.astype(dtype).astype(np.float64), x.astype(np.float64)
    resid = np.abs(T @ xd - bd)
    norm = np.abs(bd) + abs(T) @ np.abs(xd)
    k = np.diff(T.indptr) + 2
    tol = np.maximum(util.TOL[np.dtype(dtype)], 0.5 * k * np.finfo(dtype).eps)
    bad = ~(resid <= tol * norm)
    assert not bad.any(), f"row {np.flatnonzero(bad)[:5]}: resid {resid[bad][:5]} bound {(tol * norm)[bad][:5]}"
    # (never tighter than what the oracle's own sequential k-term row sums carry: k/2 * eps -- a 60 000-entry row in fp32)
    ftol = max(100 * util.TOL[np.dtype(dtype)], 0.5 * float(k.max()) * float(np.finfo(dtype).eps))
    scale = np.maximum(np.abs(ref), np.abs(ref).max() * 1e-3 + 1e-30)
    err = np.abs(xd - ref.astype(np.float64)) / scale
    assert err.max() <= ftol, f"max rel err vs oracle {err.max()} at {err.argmax()}"
    return x, info


def tri_system(n, density, upper, rng, dominant=True, dtype=np.float64):
    A = sps.random(n, n, density=density, format="csr", random_state=rng, dtype=np.float64)
    S = sps.triu(A, 1) if upper else sps.tril(A, -1)
    rowsum = np.asarray(abs(S).sum(axis=1)).ravel()
    d = rowsum + 1.0 + rng.random(n) if dominant else rng.random(n) + 1.0
    return (S + sps.diags(d)).tocsr()


@pytest.mark.parametrize("dim", util.square_dims)
def test_reference_test_shape(gpu, dim):
    """triangular_solve_test.cpp:63-86: general generate_csr matrix scaled by 1e-3, b = 0, x starts at
    1 -> the solve must return exactly 0 for both triangles with the implicit unit diagonal."""
    m, n, nnz = dim
    v, rp, ci, shape, _ = generate.generate_csr(m, n, nnz)
    M = sps.csr_matrix((v * np.float32(1e-3), ci, rp), shape=shape)
    for upper in (False, True):
        x, _ = device_solve(M, np.zeros(m), upper, True)
        ref = oracle.triangular_solve(shape, rp, ci, (v * np.float32(1e-3)), np.zeros(m, np.float32), upper=upper, unit=True)
        assert np.array_equal(x, ref) and not x.any()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("upper", [False, True])
@pytest.mark.parametrize("unit", [False, True])
def test_triangular_systems(gpu, dtype, upper, unit):
    rng = np.random.default_rng(1)
    for n, dens in ((1, 1.0), (17, 0.3), (500, 0.02), (3000, 0.004)):
        M = tri_system(n, dens, upper, rng)
        if unit:  # keep the iteration contractive without the division
            M = (sps.triu(M, 1) if upper else sps.tril(M, -1)) * 0.1 + sps.eye(n)
        b = rng.random(n) + 0.5
        x, info = check(M, b, upper, unit, dtype)
        assert info["levels"] >= 1 and info["max_level_width"] >= 1


def test_other_triangle_is_ignored_and_last_diagonal_wins(gpu):
    rng = np.random.default_rng(2)
    n = 400
    A = sps.random(n, n, density=0.03, format="csr", random_state=rng, dtype=np.float64)
    Gm = (A + sps.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 1.0)).tocsr()   # general matrix
    b = rng.random(n)
    for upper in (False, True):
        for unit in (False, True):
            Mx = Gm if not unit else (Gm * 0.01 + sps.eye(n)).tocsr()
            check(Mx, b, upper, unit, np.float64)
    # a row that stores the diagonal twice: the LAST stored entry is the divisor (triangular_solve_impl.hpp:64-66)
    rp = np.array([0, 2, 5], np.int32)
    ci = np.array([0, 0, 0, 1, 1], np.int32)
    v = np.array([2.0, 4.0, 1.0, 8.0, 5.0])
    M = sps.csr_matrix((v, ci, rp), shape=(2, 2))
    x, _ = device_solve(M, np.array([8.0, 12.0]), False, False, dtype=np.float64)
    ref = oracle.triangular_solve((2, 2), rp, ci, v, np.array([8.0, 12.0]), upper=False)
    assert np.array_equal(x, ref) and np.array_equal(ref, np.array([2.0, 2.0]))


def test_scaled_right_hand_side(gpu):
    """examples/simple_sptrsv.cpp:49-53 passes scaled(alpha, b): x = inv(T) (alpha b)."""
    rng = np.random.default_rng(13)
    n = 1500
    M = tri_system(n, 0.01, False, rng)
    b = rng.random(n) + 0.5
    d_a = G.csr_on_device(M.data.astype(np.float64), M.indptr.astype(np.int32), M.indices.astype(np.int32), M.shape, M.nnz)
    d_b, d_x = G.dev(b), torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    sp.triangular_solve(d_a, sp.lower_triangle, sp.explicit_diagonal, sp.scaled(-3.0, d_b), d_x)
    ref = oracle.triangular_solve(M.shape, M.indptr, M.indices, M.data, -3.0 * b, upper=False, unit=False)
    assert np.allclose(G.host(d_x), ref, rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("alpha", [2.0, -0.5])
def test_scaled_matrix_view(gpu, alpha):
    rng = np.random.default_rng(3)
    M = tri_system(800, 0.01, False, rng)
    b = rng.random(800)
    check(M, b, False, False, np.float64, scale_a=alpha)
    check(M, b, False, False, np.float32, scale_a=alpha, inspect=False)


def test_deep_chain_and_wide_levels(gpu):
    """Bidiagonal matrix: n levels of one row (single-workgroup chain kernel); diagonal matrix: one
    level of n rows (wide kernel); a banded block mixes both kinds of launch groups."""
    rng = np.random.default_rng(4)
    n = 5000
    Lb = sps.diags([np.full(n - 1, -0.5), np.full(n, 2.0)], [-1, 0]).tocsr()
    _, info = check(Lb, rng.random(n), False, False, np.float64)
    assert info["levels"] == n and info["max_level_width"] == 1 and info["launches_per_solve"] == 1
    Ub = Lb.T.tocsr()
    _, info = check(Ub, rng.random(n), True, False, np.float64)
    assert info["levels"] == n
    D = sps.diags(rng.random(20000) + 1.0).tocsr()
    _, info = check(D, rng.random(20000), False, False, np.float32)
    assert info["levels"] == 1 and info["max_level_width"] == 20000 and info["launches_per_solve"] == 1
    # block structure: 3 wide levels (rows depend on the previous block only) then a chain
    nb = 3000
    blocks = [[sps.diags(np.full(nb, 2.0)), None, None],
              [sps.random(nb, nb, density=0.001, random_state=rng) * 0.1, sps.diags(np.full(nb, 2.0)), None],
              [None, sps.random(nb, nb, density=0.001, random_state=rng) * 0.1, sps.diags(np.full(nb, 2.0))]]
    Lw = sps.bmat(blocks, format="csr")
    _, info = check(Lw, rng.random(3 * nb), False, False, np.float64)
    assert info["levels"] <= 3 and info["max_level_width"] >= nb


def test_long_rows_and_empty_rows(gpu):
    rng = np.random.default_rng(6)
    n = 1500
    dense_low = sps.tril(sps.random(n, n, density=0.3, random_state=rng, format="csr"), -1) * (0.5 / n)
    M = (dense_low + sps.eye(n)).tocsr()                       # rows up to ~450 entries -> 64 lanes per row
    _, info = check(M, rng.random(n), False, True, np.float64)
    assert info["lanes_per_row"] == 64
    # rows without any stored entry, unit diagonal: x = b
    E = sps.csr_matrix((n, n))
    x, _ = device_solve(E, np.arange(n, dtype=np.float64), False, True, dtype=np.float64)
    assert np.array_equal(x, np.arange(n, dtype=np.float64))


def test_errors_and_replan(gpu):
    rng = np.random.default_rng(7)
    M = tri_system(50, 0.1, False, rng)
    d_a = G.csr_on_device(M.data.astype(np.float32), M.indptr.astype(np.int32), M.indices.astype(np.int32), M.shape, M.nnz)
    b = torch.zeros(50, device="cuda")
    with pytest.raises(ValueError):
        sp.triangular_solve(d_a, sp.lower_triangle, sp.explicit_diagonal, b, torch.zeros(49, device="cuda"))
    with pytest.raises(TypeError):
        sp.triangular_solve(d_a, "lower", sp.explicit_diagonal, b, torch.zeros(50, device="cuda"))
    with pytest.raises(TypeError):
        sp.triangular_solve(d_a, sp.lower_triangle, sp.explicit_diagonal, b.double(), torch.zeros(50, device="cuda"))
    rect = G.csr_on_device(np.ones(1, np.float32), np.array([0, 1, 1], np.int32), np.zeros(1, np.int32), (2, 3), 1)
    with pytest.raises(ValueError):
        sp.triangular_solve(rect, sp.lower_triangle, sp.explicit_diagonal, torch.zeros(2, device="cuda"),
                            torch.zeros(3, device="cuda"))
    # an info inspected for the lower triangle is re-planned when used for the upper one
    x = torch.zeros(50, device="cuda")
    info = sp.triangular_solve_inspect(d_a, sp.lower_triangle, sp.explicit_diagonal, b, x)
    U = M.T.tocsr()
    d_u = G.csr_on_device(U.data.astype(np.float32), U.indptr.astype(np.int32), U.indices.astype(np.int32), U.shape, U.nnz)
    bb = G.dev(rng.random(50).astype(np.float32))
    sp.triangular_solve(info, d_u, sp.upper_triangle, sp.explicit_diagonal, bb, x)
    ref = oracle.triangular_solve(U.shape, U.indptr, U.indices, U.data.astype(np.float32), G.host(bb), upper=True)
    assert np.allclose(G.host(x), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("upper", [False, True])
@pytest.mark.parametrize("unit", [False, True])
def test_trsv_golden_bit_exact(gpu, upper, unit):
    """tests/golden/trsv_general_dyadic.npz: general matrix, +-1 entries, power-of-two diagonals -> every
    x_i is a dyadic rational computed exactly in any summation order; the other triangle is ignored."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "trsv_general_dyadic.npz"))
    n = int(g["shape"][0])
    M = sps.csr_matrix((g["values"], g["colind"], g["rowptr"]), shape=(n, n))
    M.has_canonical_format = True   # keep the stored (shuffled) order and duplicates untouched
    d_a = G.csr_on_device(g["values"], g["rowptr"], g["colind"], (n, n), len(g["values"]))
    d_b = G.dev(g["b"])
    d_x = torch.full((n,), float("nan"), device="cuda")
    uplo, diag = _tags(upper, unit)
    sp.triangular_solve(d_a, uplo, diag, d_b, d_x)
    key = f"x_{'upper' if upper else 'lower'}_{'unit' if unit else 'explicit'}"
    assert np.array_equal(G.host(d_x), g[key])


@pytest.mark.parametrize("mode", ["default", "kahn_inspect", "launch_per_level", "coop_small_grid", "coop_one_slot_pass"])
@pytest.mark.parametrize("upper", [False, True])
def test_alternative_inspect_and_solve_paths(gpu, monkeypatch, mode, upper):
    """The default is: levels by dependency polling (one self-scheduling kernel), solve = ONE cooperative launch with
    a grid barrier per level.  The other paths stay in the library -- Kahn's algorithm as the fallback of the polling
    inspect, one launch per wide level (what a stream capture or a device without cooperative launches gets) -- and must
    give the same answers (the barrier-free self-scheduling solve of rounds 2 - 4 was removed in round 5, sptrsv.hip):
    a random triangular system with a few hundred wide levels, fp32 and fp64.  The two coop_* modes push the
    cooperative kernel off its comfortable shape: 3 workgroups (every wide level needs the plain extra passes behind
    the pipelined one) and a `narrow` threshold above the widest level (workgroup 0 walks everything alone)."""
    env = {"kahn_inspect": ("SPBLAS_GFX950_TRSV_KAHN", "1"),
           "launch_per_level": ("SPBLAS_GFX950_TRSV_COOP", "0"), "coop_small_grid": ("SPBLAS_GFX950_TRSV_COOP_GRID", "3"),
           "coop_one_slot_pass": ("SPBLAS_GFX950_TRSV_NARROW", "100000")}.get(mode)
    if env:
        monkeypatch.setenv(*env)
    rng = np.random.default_rng(8)
    n, k = 60000, 6
    rows = np.repeat(np.arange(n), k)
    cols = (rng.random(n * k) * rows).astype(np.int64)
    keep = cols < rows
    S = sps.csr_matrix(((rng.random(keep.sum()) - 0.5) * (0.5 / k), (rows[keep], cols[keep])), shape=(n, n))
    M = (S + sps.diags(1.0 + rng.random(n))).tocsr()
    if upper:
        M = M.T.tocsr()
    for dtype in (np.float32, np.float64):
        x, info = check(M, rng.random(n) + 0.5, upper, False, dtype)
        assert info["levels"] > 20 and info["max_level_width"] > 128
        if mode == "launch_per_level":
            assert info["launches_per_solve"] > 20
        elif mode in ("default", "coop_small_grid"):
            # (under an HSA tool such as rocprofv3 the default falls back to one launch per level, sptrsv.hip)
            if not (os.environ.get("ROCP_TOOL_LIBRARIES") or os.environ.get("HSA_TOOLS_LIB")):
                assert info["launches_per_solve"] == 1


def test_a_solve_that_gave_up_waiting_is_reported(gpu, monkeypatch):
    """The grid barrier of the cooperative solve polls with a bound (SPBLAS_GFX950_TRSV_SPIN_LIMIT); a workgroup that runs
    into it raises the plan's status word and every workgroup leaves -- x is then incomplete.  spblas_gfx950_sptrsv_status
    (check_status) makes that visible.  A bound of zero polls forces it; the next solve with the default bound is whole again
    and reports success."""
    rng = np.random.default_rng(91)
    n, k = 200000, 6
    rows = np.repeat(np.arange(n), k)
    cols = (rng.random(n * k) * rows).astype(np.int64)
    keep = cols < rows
    S = sps.csr_matrix(((rng.random(keep.sum()) - 0.5) * (0.5 / k), (rows[keep], cols[keep])), shape=(n, n))
    M = (S + sps.diags(1.0 + rng.random(n))).tocsr()
    vals = M.data.astype(np.float32)
    d_a = G.csr_on_device(vals, M.indptr.astype(np.int32), M.indices.astype(np.int32), M.shape, M.nnz)
    b = (rng.random(n) + 0.5).astype(np.float32)
    d_b, d_x = G.dev(b), torch.zeros(n, dtype=torch.float32, device="cuda")
    info = sp.triangular_solve_inspect(d_a, sp.lower_triangle, sp.explicit_diagonal, d_b, d_x)
    if info.state_.info()["launches_per_solve"] != 1:
        pytest.skip("the cooperative solve is not in use here (an HSA tool is loaded)")
    info.state_.check_status()  # nothing solved yet: fine
    monkeypatch.setenv("SPBLAS_GFX950_TRSV_SPIN_LIMIT", "0")
    sp.triangular_solve(info, d_a, sp.lower_triangle, sp.explicit_diagonal, d_b, d_x)
    with pytest.raises(RuntimeError, match="ran into its bound"):
        info.state_.check_status()
    monkeypatch.delenv("SPBLAS_GFX950_TRSV_SPIN_LIMIT")
    sp.triangular_solve(info, d_a, sp.lower_triangle, sp.explicit_diagonal, d_b, d_x)
    info.state_.check_status()
    ref = oracle.triangular_solve(M.shape, M.indptr, M.indices, vals, b, upper=False, unit=False)
    assert np.allclose(G.host(d_x), ref, rtol=1e-4, atol=1e-6)


def test_a_solve_that_gave_up_waiting_fails_the_next_solve_by_itself(gpu, monkeypatch):
    """Round-5 advisor: no default path called check_status, so an incomplete x went unnoticed.  The kernel now also raises a
    pinned host word on that path and the NEXT solve on the plan reads it before it launches anything: it fails (once) with
    a HIP error status instead of running, without a stream synchronisation in any successful call; the solve after that is
    whole again."""
    rng = np.random.default_rng(93)
    n, k = 150000, 5
    rows = np.repeat(np.arange(n), k)
    cols = (rng.random(n * k) * rows).astype(np.int64)
    keep = cols < rows
    S = sps.csr_matrix(((rng.random(keep.sum()) - 0.5) * (0.5 / k), (rows[keep], cols[keep])), shape=(n, n))
    M = (S + sps.diags(1.0 + rng.random(n))).tocsr()
    vals = M.data.astype(np.float32)
    d_a = G.csr_on_device(vals, M.indptr.astype(np.int32), M.indices.astype(np.int32), M.shape, M.nnz)
    b = (rng.random(n) + 0.5).astype(np.float32)
    d_b, d_x = G.dev(b), torch.zeros(n, dtype=torch.float32, device="cuda")
    info = sp.triangular_solve_inspect(d_a, sp.lower_triangle, sp.explicit_diagonal, d_b, d_x)
    if info.state_.info()["launches_per_solve"] != 1:
        pytest.skip("the cooperative solve is not in use here (an HSA tool is loaded)")
    sp.triangular_solve(info, d_a, sp.lower_triangle, sp.explicit_diagonal, d_b, d_x)   # allocates the control words
    monkeypatch.setenv("SPBLAS_GFX950_TRSV_SPIN_LIMIT", "0")
    sp.triangular_solve(info, d_a, sp.lower_triangle, sp.explicit_diagonal, d_b, d_x)   # gives up; the call itself succeeds
    monkeypatch.delenv("SPBLAS_GFX950_TRSV_SPIN_LIMIT")
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError):
        sp.triangular_solve(info, d_a, sp.lower_triangle, sp.explicit_diagonal, d_b, d_x)
    sp.triangular_solve(info, d_a, sp.lower_triangle, sp.explicit_diagonal, d_b, d_x)   # reported once; this one runs
    info.state_.check_status()
    ref = oracle.triangular_solve(M.shape, M.indptr, M.indices, vals, b, upper=False, unit=False)
    assert np.allclose(G.host(d_x), ref, rtol=1e-4, atol=1e-6)
