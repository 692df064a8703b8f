"""Randomised stress of the SpMV plans against the CPU oracle (run on the GPU box):
    python tools/fuzz_spmv.py [iterations] [first_seed]
Random shapes, row-length distributions (uniform / power law / banded / mostly empty / duplicate-heavy),
value types, offset types, algorithms and the SLICED test hooks (tile sizes, slice split, reduce shape)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import spblas_reference_amd as sp
from spblas_reference_amd import _capi
from oracle import oracle
import util

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
HOOKS = ["SPBLAS_GFX950_SLICE_COLS", "SPBLAS_GFX950_SLICE_ROWS", "SPBLAS_GFX950_PB_KSPLIT", "SPBLAS_GFX950_PB_RWAVES",
         "SPBLAS_GFX950_PB_RBATCH", "SPBLAS_GFX950_PB_RLDS_KB", "SPBLAS_GFX950_PB_BINS", "SPBLAS_GFX950_PB_VARBINS",
         "SPBLAS_GFX950_PB_HUB_LEN", "SPBLAS_GFX950_PB_COMPACT", "SPBLAS_GFX950_PB_ENC8", "SPBLAS_GFX950_PB_ENC8_FAIL",
         "SPBLAS_GFX950_PB_LPT", "SPBLAS_GFX950_PB_XITEM_DIV", "SPBLAS_GFX950_PB_RITEMS", "SPBLAS_GFX950_PB_XLDS_KB",
         "SPBLAS_GFX950_PB_STAGED_SCATTER", "SPBLAS_GFX950_PB_SPLIT_LEN", "SPBLAS_GFX950_PB_RUN_MIN", "SPBLAS_GFX950_PB_NT", "SPBLAS_GFX950_PB_TUNE_MIN",
         "SPBLAS_GFX950_PB_HOT", "SPBLAS_GFX950_PB_HOT_MIN_PCT", "SPBLAS_GFX950_PB_VFREE", "SPBLAS_GFX950_PB_VF_ROWS",
         "SPBLAS_GFX950_PB_VF_WAVES", "SPBLAS_GFX950_PB_VF_GRID", "SPBLAS_GFX950_PB_KEEP_SRC", "SPBLAS_GFX950_PB_KEEP_REST",
         "SPBLAS_GFX950_PB_STAGE_Q16", "SPBLAS_GFX950_PB_WPART"]
dev = torch.device("cuda:0")
bad = 0
for it in range(iters):
    rng = np.random.default_rng(seed0 + it)
    for h in HOOKS:
        os.environ.pop(h, None)
    m = int(rng.choice([1, 7, 300, 5000, 40000, 150000]))
    n = int(rng.choice([1, 13, 999, 20000, 70000, 300000]))
    kind = rng.choice(["uniform", "powerlaw", "banded", "sparse_rows", "dups", "hotcols", "hotcols"])
    if kind == "uniform":
        lens = rng.integers(0, 24, m)
    elif kind == "powerlaw" or (kind == "hotcols" and rng.random() < 0.5):
        lens = np.minimum(rng.zipf(1.5, m), 30000)
    elif kind == "hotcols":
        lens = rng.integers(0, 40, m)
    elif kind == "banded":
        lens = np.full(m, min(n, 9))
    elif kind == "sparse_rows":
        lens = np.where(rng.random(m) < 0.05, rng.integers(1, 200, m), 0)
    else:
        lens = rng.integers(0, 80, m)
    lens = lens.astype(np.int64)
    if lens.sum() > 6_000_000:
        lens = (lens * (6_000_000 / lens.sum())).astype(np.int64)
    rowptr = np.concatenate([[0], np.cumsum(lens)])
    nnz = int(rowptr[-1])
    if kind == "banded":
        rows = np.repeat(np.arange(m), lens)
        colind = ((rows * max(n // max(m, 1), 1) + rng.integers(0, min(n, 50), nnz)) % n).astype(np.int32)
    elif kind == "dups":
        colind = rng.integers(0, max(1, min(n, 40)), nnz).astype(np.int32)   # few distinct columns: many duplicates
    elif kind == "hotcols":  # round 4: a few columns carry most of the entries (the hot-column split of the SLICED plan)
        nh = int(rng.choice([1, 5, 200, 5000]))
        hot_set = rng.integers(0, n, nh)
        is_hot = rng.random(nnz) < rng.choice([0.2, 0.6, 0.95])
        colind = np.where(is_hot, hot_set[rng.integers(0, nh, nnz)], rng.integers(0, n, nnz)).astype(np.int32)
    else:
        colind = rng.integers(0, n, nnz).astype(np.int32)
    dtype = rng.choice([np.float32, np.float64])
    values = (rng.random(nnz) - (0.5 if rng.random() < 0.5 else 0.0)).astype(dtype)
    x = (rng.random(n) - 0.5).astype(dtype)
    alg = rng.choice(["auto", "vector", "rowblock", "sliced", "none"])
    if os.environ.get("FUZZ_FORCE_HOT"):  # round 4: every case through the SLICED plan with the hot-column split forced
        alg = "sliced"
    off64 = bool(rng.random() < 0.3)
    hooks = {}
    if alg == "sliced":
        if rng.random() < 0.7:
            hooks["SPBLAS_GFX950_SLICE_COLS"] = str(int(rng.choice([4, 64, 1000, 20000])))
        if rng.random() < 0.7:
            hooks["SPBLAS_GFX950_SLICE_ROWS"] = str(int(rng.choice([1, 16, 64, 700, 5000])))
        hooks["SPBLAS_GFX950_PB_KSPLIT"] = str(int(rng.choice([0, 1, 2, 8, 32])))
        hooks["SPBLAS_GFX950_PB_RWAVES"] = str(int(rng.choice([4, 8])))
        hooks["SPBLAS_GFX950_PB_RBATCH"] = str(int(rng.choice([2, 4, 8])))
        hooks["SPBLAS_GFX950_PB_RLDS_KB"] = str(int(rng.choice([40, 80, 160])))
        hooks["SPBLAS_GFX950_PB_BINS"] = str(int(rng.choice([64, 512, 2048, 4096])))
        hooks["SPBLAS_GFX950_PB_VARBINS"] = str(int(rng.choice([-1, 0, 1])))   # variable-height bins: auto / off / forced
        hooks["SPBLAS_GFX950_PB_HUB_LEN"] = str(int(rng.choice([1, 300, 16384])))
        hooks["SPBLAS_GFX950_PB_COMPACT"] = str(int(rng.choice([-1, 0, 1])))    # tiles over the non-empty rows only
        # round 3: one-byte row codes (0 off / 1 rule / 2 forced; _FAIL = pretend the encoder overflowed -> 16-bit
        # fall-back), heaviest-first work lists, finer expand items, K split of heavy reduce groups, x slice size,
        # staged vs direct scatter, long rows in pieces, minimum run length
        hooks["SPBLAS_GFX950_PB_ENC8"] = str(int(rng.choice([0, 1, 2, 2])))
        if rng.random() < 0.1:
            hooks["SPBLAS_GFX950_PB_ENC8_FAIL"] = "1"
        hooks["SPBLAS_GFX950_PB_LPT"] = str(int(rng.choice([0, 1])))
        hooks["SPBLAS_GFX950_PB_XITEM_DIV"] = str(int(rng.choice([1, 2, 8])))
        hooks["SPBLAS_GFX950_PB_RITEMS"] = str(int(rng.choice([0, 1, 2, 4])))
        hooks["SPBLAS_GFX950_PB_XLDS_KB"] = str(int(rng.choice([40, 80, 160])))
        hooks["SPBLAS_GFX950_PB_STAGED_SCATTER"] = str(int(rng.choice([0, 1])))
        if rng.random() < 0.5:
            hooks["SPBLAS_GFX950_PB_SPLIT_LEN"] = str(int(rng.choice([64, 1000, 100000])))
        hooks["SPBLAS_GFX950_PB_NT"] = str(int(rng.choice([0, 1, -1])))  # product stores plain / non-temporal / by trial
        if rng.random() < 0.3:  # the store trial of large plans on a small one
            hooks["SPBLAS_GFX950_PB_TUNE_MIN"] = "0"
        if rng.random() < 0.3:
            hooks["SPBLAS_GFX950_PB_RUN_MIN"] = str(int(rng.choice([1, 8, 64])))
        # round 4: hot-column split off / by rule / forced, with the coverage threshold down to "whatever the sample shows"
        hooks["SPBLAS_GFX950_PB_HOT"] = str(int(rng.choice([-1, 0, 1, 1, 1])))
        hooks["SPBLAS_GFX950_PB_HOT_MIN_PCT"] = str(int(rng.choice([0, 1, 15])))
        if os.environ.get("FUZZ_FORCE_HOT"):
            hooks["SPBLAS_GFX950_PB_HOT"], hooks["SPBLAS_GFX950_PB_HOT_MIN_PCT"] = "1", "0"
        # round 5: value-free tiles on request (the build falls back by itself where they do not apply), bin height / waves /
        # grid of its reduce; source positions kept or not; 16-bit staging and per-part counts of the scatter on / off
        if rng.random() < 0.45 or os.environ.get("FUZZ_FORCE_VFREE"):
            hooks["SPBLAS_GFX950_PB_VFREE"] = "2"
            hooks["SPBLAS_GFX950_PB_VF_ROWS"] = str(int(rng.choice([0, 7, 64, 300, 3000])))
            hooks["SPBLAS_GFX950_PB_VF_WAVES"] = str(int(rng.choice([4, 8])))
            hooks["SPBLAS_GFX950_PB_VF_GRID"] = str(int(rng.choice([1, 3, 256])))
        if os.environ.get("FUZZ_FORCE_VFREE"):  # ... and nothing that would make the build fall back to the copying form
            hooks["SPBLAS_GFX950_PB_COMPACT"], hooks["SPBLAS_GFX950_PB_SPLIT_LEN"], hooks["SPBLAS_GFX950_PB_VARBINS"] = "0", "1000000", "0"
            hooks.pop("SPBLAS_GFX950_SLICE_ROWS", None)
            hooks["SPBLAS_GFX950_PB_HOT"] = "0"
        hooks["SPBLAS_GFX950_PB_KEEP_SRC"] = str(int(rng.choice([0, 1])))
        hooks["SPBLAS_GFX950_PB_KEEP_REST"] = str(int(rng.choice([0, 1])))
        hooks["SPBLAS_GFX950_PB_STAGE_Q16"] = str(int(rng.choice([0, 1])))
        hooks["SPBLAS_GFX950_PB_WPART"] = str(int(rng.choice([0, 1])))
    os.environ.update(hooks)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rp_dev = t(rowptr.astype(np.int64 if off64 else np.int32))
    a = sp.csr_view(t(values), rp_dev, t(colind), (m, n), nnz)
    xd = t(x)
    y = torch.full((m,), float("nan"), dtype=xd.dtype, device=dev)
    alpha = float(rng.choice([1.0, -2.5]))
    desc = f"seed {seed0 + it}: {kind} {m}x{n} nnz={nnz} {np.dtype(dtype).name} alg={alg} off64={off64} alpha={alpha} {hooks}"
    try:
        A = sp.scaled(alpha, a) if alpha != 1.0 else a
        if alg == "none":
            sp.multiply(A, xd, y)
        else:
            algs = {"auto": _capi.SPMV_AUTO, "vector": _capi.SPMV_VECTOR, "rowblock": _capi.SPMV_ROWBLOCK, "sliced": _capi.SPMV_SLICED}
            try:
                # (round 5: a caller that announces value changes gets the source positions at inspect)
                vwc = bool(alg == "sliced" and rng.random() < 0.4)
                info = sp.multiply_inspect(a, xd, y, alg=algs[alg], values_will_change=vwc)
                if vwc:
                    desc += " [values_will_change]"
            except Exception as e:  # noqa: BLE001 -- "not supported" for a forced algorithm is a legal answer
                print("SKIP", desc, "->", type(e).__name__, str(e)[:80])
                continue
            sp.multiply(info, A, xd, y)
            sp.multiply(info, A, xd, y)
            if alg == "sliced" and "hot_split" in info.state_.sliced_info():
                desc += " [hot split]"
            if rng.random() < 0.3:  # a rebound value array: snapshot plans (tiles, hot split) take their values again
                values = (values * dtype(-0.5) + dtype(0.25)).astype(dtype)
                a.update(t(values), a.rowptr(), a.colind())
                A = sp.scaled(alpha, a) if alpha != 1.0 else a
                sp.multiply(info, A, xd, y)
                desc += " +rebound"
                if rng.random() < 0.5:  # ... and again: the second change of a snapshot plan is the gather through the sources
                    values = (values * dtype(1.5) - dtype(0.0625)).astype(dtype)
                    a.update(t(values), a.rowptr(), a.colind())
                    A = sp.scaled(alpha, a) if alpha != 1.0 else a
                    sp.multiply(info, A, xd, y)
                    desc += " +rebound"
            if alg == "sliced" and info.state_.sliced_info().get("value_free"):
                desc += " [value-free]"
                if rng.random() < 0.5:  # written in place, behind the library's back: a value-free plan reads the array as it is
                    values = (values * dtype(2.0) - dtype(0.125)).astype(dtype)
                    raw = a.values()
                    torch.as_strided(raw, raw.shape, raw.stride()).data.copy_(t(values))
                    sp.multiply(info, A, xd, y)
                    desc += " +in place"
        torch.cuda.synchronize()
        yh = y.cpu().numpy()
        rp32 = rowptr.astype(np.int32)
        ref = oracle.spmv((m, n), rp32, colind, values, x, scale_a=None if alpha == 1.0 else alpha)
        absrow = abs(alpha) * oracle.spmv_absrow(rp32, colind, values, x)
        util.assert_parity(yh, ref, absrow, dtype, row_len=np.diff(rowptr), what=desc)
        print("ok  ", desc)
    except AssertionError as e:
        bad += 1
        print("FAIL", desc, "->", str(e)[:200])
print("failures:", bad)
sys.exit(1 if bad else 0)
