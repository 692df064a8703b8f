"""Randomised stress of y = alpha A^T x + beta y (csc_view operands, multiply_impl.hpp:33-53 over a csc_view; the rocSPARSE
slot: vendor/rocsparse/detail/get_transpose.hpp:19-29) against the CPU oracle (run on the GPU box):
    python tools/fuzz_spmv_t.py [iterations] [first_seed]
Three forms per case: the un-inspected call through the C ABI in its two-pass form (forced for any size, with random slice
widths and segment lengths: SPBLAS_GFX950_SPMV_T2=1, _T2_W, _T2_SEG), the scatter kernel (=0), and the inspected csc_view
operand of the host layer (device transpose at inspect, the CSR plans on the copy, arrays handed back where the plan is
self-contained)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import spblas_reference_amd as sp
from spblas_reference_amd import _capi
from oracle import oracle
import util

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
HOOKS = ["SPBLAS_GFX950_SPMV_T2", "SPBLAS_GFX950_SPMV_T2_W", "SPBLAS_GFX950_SPMV_T2_SEG"]
dev = torch.device("cuda:0")
bad = 0
for it in range(iters):
    rng = np.random.default_rng(seed0 + it)
    for h in HOOKS:
        os.environ.pop(h, None)
    m = int(rng.choice([1, 7, 300, 5000, 40000, 150000]))
    n = int(rng.choice([1, 13, 999, 20000, 70000, 300000]))
    if os.environ.get("FUZZ_BIG"):  # sizes at which the two-pass form is the default (>= 4 M entries, >= 65 536 columns)
        m = int(rng.choice([200000, 1500000, 6000000]))
        n = int(rng.choice([70000, 900000, 12000000, 40000000]))
    kind = rng.choice(["uniform", "powerlaw", "banded", "sparse_rows", "dups", "hotcols", "empty_stretch"])
    if kind == "uniform":
        lens = rng.integers(0, 24, m)
    elif kind == "powerlaw":
        lens = np.minimum(rng.zipf(1.5, m), 30000)
    elif kind == "hotcols":
        lens = rng.integers(0, 40, m)
    elif kind == "banded":
        lens = np.full(m, min(n, 9))
    elif kind == "sparse_rows":
        lens = np.where(rng.random(m) < 0.05, rng.integers(1, 200, m), 0)
    elif kind == "empty_stretch":  # rows with entries at both ends, nothing between: one tile spans every empty row
        lens = np.zeros(m, np.int64)
        k = max(1, m // 50)
        lens[:k] = rng.integers(1, 30, k)
        lens[-k:] = rng.integers(1, 30, k)
    else:
        lens = rng.integers(0, 80, m)
    lens = lens.astype(np.int64)
    cap = 24_000_000 if os.environ.get("FUZZ_BIG") else 3_000_000
    if os.environ.get("FUZZ_BIG") and lens.sum() < 5_000_000:
        lens = lens * int(np.ceil(5_000_000 / max(1, lens.sum())))
    if lens.sum() > cap:
        lens = (lens * (cap / lens.sum())).astype(np.int64)
    rowptr = np.concatenate([[0], np.cumsum(lens)])
    nnz = int(rowptr[-1])
    if kind == "banded":
        rows = np.repeat(np.arange(m), lens)
        colind = ((rows * max(n // max(m, 1), 1) + rng.integers(0, min(n, 50), nnz)) % n).astype(np.int32)
    elif kind == "dups":
        colind = rng.integers(0, max(1, min(n, 40)), nnz).astype(np.int32)
    elif kind == "hotcols":
        nh = int(rng.choice([1, 5, 200]))
        hot_set = rng.integers(0, n, nh)
        is_hot = rng.random(nnz) < rng.choice([0.2, 0.6, 0.95])
        colind = np.where(is_hot, hot_set[rng.integers(0, nh, nnz)], rng.integers(0, n, nnz)).astype(np.int32)
    else:
        colind = rng.integers(0, n, nnz).astype(np.int32)
    dtype = rng.choice([np.float32, np.float64])
    values = (rng.random(nnz) - (0.5 if rng.random() < 0.5 else 0.0)).astype(dtype)
    x = (rng.random(m) - 0.5).astype(dtype)
    y0 = (rng.random(n) - 0.5).astype(dtype)
    off64 = bool(rng.random() < 0.3)
    alpha = float(rng.choice([1.0, -2.5]))
    beta = float(rng.choice([0.0, 0.0, 0.75]))
    hooks = {}
    if os.environ.get("FUZZ_BIG"):
        pass  # the sizes speak for themselves: default slice width and segment length
    elif rng.random() < 0.8:
        hooks["SPBLAS_GFX950_SPMV_T2_W"] = str(int(rng.choice([64, 320, 4096, 9984, 19392 if dtype == np.float32 else 9728])))
    if rng.random() < 0.6:
        hooks["SPBLAS_GFX950_SPMV_T2_SEG"] = str(int(rng.choice([100000, 1000000] if os.environ.get("FUZZ_BIG") else [64, 1000, 8192, 100000])))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    dv, drp, dci, dx = t(values), t(rowptr.astype(np.int64 if off64 else np.int32)), t(colind), t(x)
    desc = f"seed {seed0 + it}: {kind} A {m}x{n} nnz={nnz} {np.dtype(dtype).name} off64={off64} alpha={alpha} beta={beta} {hooks}"
    try:
        rp32 = rowptr.astype(np.int32)
        ref = oracle.spmv_csc((n, m), rp32, colind, values, x).astype(np.float64)
        ab = oracle.spmv_csc((n, m), rp32, colind, np.abs(values), np.abs(x)).astype(np.float64)
        want = alpha * ref + (beta * y0 if beta else 0.0)
        scale = abs(alpha) * ab + (abs(beta) * np.abs(y0) if beta else 0.0)
        cnt = np.bincount(colind, minlength=n) + 1
        hd = sp.api._Handle.current(dev)
        ct = ctypes.c_float if dtype == np.float32 else ctypes.c_double
        for mode in ("1", "0"):
            os.environ["SPBLAS_GFX950_SPMV_T2"] = mode
            if mode == "1":
                os.environ.update(hooks)
            y = t(y0) if beta else torch.full((n,), float("nan"), dtype=tdt, device=dev)
            al, be = ct(alpha), ct(beta)
            sp.api.check(_capi.lib().spblas_gfx950_spmv(hd.h, None, _capi.OP_T, m, n, nnz, ctypes.byref(al), sp.api._ptr(drp),
                                                        sp.api._ptr(dci), sp.api._ptr(dv), sp.api._ptr(dx), ctypes.byref(be),
                                                        sp.api._ptr(y), _capi.I64 if off64 else _capi.I32,
                                                        _capi.F32 if dtype == np.float32 else _capi.F64), "spmv")
            torch.cuda.synchronize()
            util.assert_parity(y.cpu().numpy(), want, scale, dtype, row_len=cnt, what=f"T2={mode} " + desc)
        for h in HOOKS:
            os.environ.pop(h, None)
        # the host layer: A^T as a csc_view, inspected (twice through the plan), then once without the plan
        a = sp.csr_view(dv, drp, dci, (m, n), nnz)
        at = sp.transposed(a)
        A = sp.scaled(alpha, at) if alpha != 1.0 else at
        y = torch.full((n,), float("nan"), dtype=tdt, device=dev)
        info = sp.multiply_inspect(at, dx, y)
        sp.multiply(info, A, dx, y)
        sp.multiply(info, A, dx, y)
        torch.cuda.synchronize()
        util.assert_parity(y.cpu().numpy(), alpha * ref, abs(alpha) * ab, dtype, row_len=cnt, what="inspected " + desc)
        y.fill_(float("nan"))
        sp.multiply(A, dx, y)
        torch.cuda.synchronize()
        util.assert_parity(y.cpu().numpy(), alpha * ref, abs(alpha) * ab, dtype, row_len=cnt, what="un-inspected " + desc)
        print("ok  ", desc)
    except AssertionError as e:
        bad += 1
        print("FAIL", desc, "->", str(e)[:200])
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("ERR ", desc, "->", type(e).__name__, str(e)[:200])
print("failures:", bad)
sys.exit(1 if bad else 0)
