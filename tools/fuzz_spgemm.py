"""Randomised stress of SpGEMM (3- and 4-argument), add, transpose and SpTRSV against the CPU oracle (GPU box):
    python tools/fuzz_spgemm.py [iterations] [first_seed]"""
import os, sys
import numpy as np, scipy.sparse as sps, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_add as TA
import test_gpu_spgemm as TS
import test_gpu_sptrsv as TT
import gpu_util as G
import spblas_reference_amd as sp
from oracle import oracle

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0


def rcsr(rng, m, n, kind, dtype):
    if kind == "uniform":
        lens = rng.integers(0, 12, m)
    elif kind == "skew":
        lens = np.minimum(rng.zipf(1.7, m), 3000)
    else:
        lens = np.where(rng.random(m) < 0.1, rng.integers(1, 600, m), 0)
    lens = np.minimum(lens, n * 4).astype(np.int64)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(rp[-1])
    ci = rng.integers(0, n, nnz).astype(np.int32)
    v = (rng.random(nnz) + 0.25).astype(dtype)
    return (v, rp, ci, (m, n))


for it in range(iters):
    rng = np.random.default_rng(seed0 + it)
    dtype = rng.choice([np.float32, np.float64])
    # FUZZ_BIG=1: sizes at which the binned kernels (direct sort, hash tables by row size, dense bitmap) all get rows
    m, k, n = (int(rng.choice([3000, 40000, 200000] if os.environ.get("FUZZ_BIG") else [1, 30, 400, 3000])) for _ in range(3))
    kinds = [rng.choice(["uniform", "skew", "sparse"]) for _ in range(3)]
    desc = f"seed {seed0 + it}: {m}x{k}x{n} {kinds} {np.dtype(dtype).name}"
    try:
        a, b, d = rcsr(rng, m, k, kinds[0], dtype), rcsr(rng, k, n, kinds[1], dtype), rcsr(rng, m, n, kinds[2], dtype)
        TS.check_against_oracle(a, b, TS.device_spgemm(a, b, bool(it & 1)), dtype)
        al, be = float(rng.choice([1.0, 2.0])), float(rng.choice([1.0, -0.5]))
        TA.check_spgemm4(a, b, d, TA.device_spgemm4(a, b, d, sa=None if al == 1.0 else al, sd=None if be == 1.0 else be)[0],
                         dtype, al, be)
        # repeated numeric passes: with SPBLAS_GFX950_SPGEMM_REUSE=2 the first pass records the ranks and the second
        # one IS the rank path (bins 1-3, addend entries included); new values in between
        os.environ["SPBLAS_GFX950_SPGEMM_REUSE"] = "2" if it & 1 else "1"
        new_vals = tuple((rng.random(len(t[0])) + 0.25).astype(dtype) for t in (a, b, d))
        first, second = TA.device_spgemm4(a, b, d, sa=None if al == 1.0 else al, sd=None if be == 1.0 else be,
                                          reuse_values=new_vals)
        TA.check_spgemm4(a, b, d, first, dtype, al, be)
        a2, b2, d2 = ((new_vals[i],) + t[1:] for i, t in enumerate((a, b, d)))
        TA.check_spgemm4(a2, b2, d2, second, dtype, al, be)
        d_a3, d_b3 = G.csr_on_device(*a, len(a[0])), G.csr_on_device(*b, len(b[0]))
        rp3 = torch.full((m + 1,), -1, dtype=torch.int32, device="cuda")
        c3 = sp.csr_view(None, rp3, None, (m, n), 0)
        st3 = sp.spgemm_state_t()
        sp.multiply_compute(st3, d_a3, d_b3, c3)
        cn3 = st3.result_nnz()
        v3 = torch.full((cn3,), float("nan"), dtype=G.dev(a[0]).dtype, device="cuda")
        k3 = torch.full((cn3,), -1, dtype=torch.int32, device="cuda")
        c3.update(v3, rp3, k3, (m, n), cn3)
        for rep, (va, vb) in enumerate([(a[0], b[0]), (new_vals[0], new_vals[1]), (a[0], b[0])]):
            d_a3.values().copy_(G.dev(va))
            d_b3.values().copy_(G.dev(vb))
            v3.fill_(float("nan"))
            k3.fill_(-1)
            sp.multiply_fill(st3, d_a3, d_b3, c3)
            TS.check_against_oracle((va,) + a[1:], (vb,) + b[1:], (cn3, G.host(rp3), G.host(k3), G.host(v3)), dtype)
        os.environ.pop("SPBLAS_GFX950_SPGEMM_REUSE", None)
        e = rcsr(rng, m, n, kinds[0], dtype)
        TA.check_add(d, e, TA.device_add(d, e, None if al == 1.0 else al, None if be == 1.0 else be), dtype,
                     None if al == 1.0 else al, None if be == 1.0 else be)
        # transpose of a
        d_a = G.csr_on_device(*a, len(a[0]))
        nnz = len(a[0])
        tb = sp.csr_view(torch.empty(max(nnz, 1), dtype=G.dev(a[0]).dtype, device="cuda")[:nnz],
                         torch.empty(k + 1, dtype=torch.int32, device="cuda"),
                         torch.empty(max(nnz, 1), dtype=torch.int32, device="cuda")[:nnz], (k, m), nnz)
        sp.transpose(d_a, tb)
        rp, ci, vv = oracle.transpose((m, k), a[1], a[2], a[0])
        assert np.array_equal(G.host(tb.rowptr()), rp) and np.array_equal(G.host(tb.colind()), ci) and \
            np.array_equal(G.host(tb.values()), vv), "transpose mismatch"
        # triangular solve on a diagonally dominant square system
        q = int(rng.choice([1, 50, 2000]))
        M = TT.tri_system(q, min(1.0, 8.0 / q), bool(it & 2), rng)
        TT.check(M, rng.random(q) + 0.5, bool(it & 2), False, dtype)
        print("ok  ", desc)
    except AssertionError as ex:
        bad += 1
        print("FAIL", desc, "->", str(ex)[:300])
print("failures:", bad)
sys.exit(1 if bad else 0)
