"""Randomised stress of SpMM (CSR and CSC / transposed operands, with and without inspect) against the CPU oracle
(run on the GPU box):
    python tools/fuzz_spmm.py [iterations] [first_seed]
Random shapes, row-length distributions (uniform / power law with hub rows / banded = dense panels for the matrix-core
kernel / mostly empty / duplicate columns), column counts 1..260, value and offset types, dense operands that are
column windows of wider matrices (leading dimension != columns) at odd offsets, scaled views."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import spblas_reference_amd as sp
from oracle import oracle
import util

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
bad = 0
for it in range(iters):
    rng = np.random.default_rng(seed0 + it)
    m = int(rng.choice([1, 9, 300, 4000, 33000, 120000]))
    k = int(rng.choice([1, 17, 999, 20000, 90000]))
    ncols = int(rng.choice([1, 2, 3, 4, 7, 8, 16, 31, 32, 33, 64, 100, 128, 129, 260]))
    kind = rng.choice(["uniform", "powerlaw", "banded", "sparse_rows", "dups"])
    if kind == "uniform":
        lens = rng.integers(0, 40, m)
    elif kind == "powerlaw":
        lens = np.minimum(rng.zipf(1.4, m), 60000)
    elif kind == "banded":
        lens = np.full(m, min(k, int(rng.choice([8, 32, 48]))))
    elif kind == "sparse_rows":
        lens = np.where(rng.random(m) < 0.05, rng.integers(1, 300, m), 0)
    else:
        lens = rng.integers(0, 60, m)
    lens = lens.astype(np.int64)
    cap = 3_000_000 // max(1, ncols // 16)
    if lens.sum() > cap:
        lens = (lens * (cap / lens.sum())).astype(np.int64)
    rowptr = np.concatenate([[0], np.cumsum(lens)])
    nnz = int(rowptr[-1])
    rows = np.repeat(np.arange(m), lens)
    if kind == "banded":
        centre = (rows * (k / max(m, 1))).astype(np.int64)
        colind = np.clip(centre + rng.integers(-24, 24, nnz), 0, k - 1).astype(np.int32)
    elif kind == "dups":
        colind = rng.integers(0, max(1, min(k, 30)), nnz).astype(np.int32)
    else:
        colind = rng.integers(0, k, nnz).astype(np.int32)
    dtype = rng.choice([np.float32, np.float64])
    values = (rng.random(nnz) - 0.5).astype(dtype)
    B_h = (rng.random((k, ncols)) - 0.5).astype(dtype)
    off64 = bool(rng.random() < 0.3)
    inspect = bool(rng.random() < 0.6)
    transposed = bool(rng.random() < 0.25)   # C' = A^T B' through the CSC view of the same arrays
    windowed = bool(rng.random() < 0.5)
    alpha = float(rng.choice([1.0, 1.0, -1.5]))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    a = sp.csr_view(t(values), t(rowptr.astype(np.int64 if off64 else np.int32)), t(colind), (m, k), nnz)
    desc = (f"seed {seed0 + it}: {kind} {m}x{k} nnz={nnz} n={ncols} {np.dtype(dtype).name} off64={off64} inspect={inspect} "
            f"transposed={transposed} windowed={windowed} alpha={alpha}")
    try:
        if transposed:
            Bt_h = (rng.random((m, ncols)) - 0.5).astype(dtype)
            op, rhs_h, out_rows = sp.transposed(a), Bt_h, k
        else:
            op, rhs_h, out_rows = a, B_h, m
        if windowed:
            Bw = torch.zeros((rhs_h.shape[0], ncols + 6), dtype=tdt, device=dev)
            B = Bw[:, 3:3 + ncols]
            B.copy_(t(rhs_h))
            Cw = torch.full((out_rows, ncols + 3), float("nan"), dtype=tdt, device=dev)
            C = Cw[:, 1:1 + ncols]
        else:
            B = t(rhs_h)
            C = torch.full((out_rows, ncols), float("nan"), dtype=tdt, device=dev)
        A = sp.scaled(alpha, op) if alpha != 1.0 else op
        if inspect:
            info = sp.multiply_inspect(op, B, C)
            sp.multiply(info, A, B, C)
            sp.multiply(info, A, B, C)
        else:
            sp.multiply(A, B, C)
        torch.cuda.synchronize()
        got = C.cpu().numpy()
        rp32 = rowptr.astype(np.int32)
        if transposed:
            # row-major form of A^T by the oracle's stable transpose (duplicates stay separate entries, as on the device)
            tr, tc, tv = oracle.transpose((m, k), rp32, colind, values)
            ref = oracle.spmm((k, m), tr, tc, tv, rhs_h, scale_a=None if alpha == 1.0 else alpha)
            scale = abs(alpha) * oracle.spmm((k, m), tr, tc, np.abs(tv), np.abs(rhs_h)).astype(np.float64)
            lens_out = np.diff(tr)
        else:
            ref = oracle.spmm((m, k), rp32, colind, values, rhs_h, scale_a=None if alpha == 1.0 else alpha)
            scale = abs(alpha) * oracle.spmm((m, k), rp32, colind, np.abs(values), np.abs(rhs_h)).astype(np.float64)
            lens_out = lens
        tol = np.maximum(util.TOL[np.dtype(dtype)], lens_out * np.finfo(dtype).eps)[:, None]
        err = np.abs(got.astype(np.float64) - ref)
        if not np.all(err <= tol * scale + 1e-300):
            i = np.unravel_index(np.argmax(err - tol * scale), err.shape)
            raise AssertionError(f"entry {i}: got {got[i]} ref {ref[i]} bound {(tol * scale)[i]}")
        if windowed and not (torch.isnan(Cw[:, :1]).all() and torch.isnan(Cw[:, 1 + ncols:]).all()):
            raise AssertionError("wrote outside the C window")
        print("ok  ", desc)
    except AssertionError as e:
        bad += 1
        print("FAIL", desc, "->", str(e)[:200])
print("failures:", bad)
sys.exit(1 if bad else 0)
