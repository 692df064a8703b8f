"""Large-size sanity check of the sliced plan against the row-block kernel (same matrix, same x):
    python tools/scale_check.py [rows] [per_row]     default 40M x 40M, 10 per row (4e8 entries)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spblas_reference_amd as sp
from spblas_reference_amd import generate, _capi
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
per = int(sys.argv[2]) if len(sys.argv) > 2 else 10
v, rp, ci, shape, nnz = generate.uniform_csr_device(n, n, per, dtype=torch.float32, seed=0, device=dev)
a = sp.csr_view(v, rp, ci, shape, nnz)
x = torch.rand(n, device=dev) + 0.5
ys = {}
for name, alg in (("rowblock", _capi.SPMV_ROWBLOCK), ("sliced", _capi.SPMV_SLICED), ("auto", _capi.SPMV_AUTO)):
    y = torch.full((n,), float("nan"), device=dev)
    t0 = time.perf_counter()
    info = sp.multiply_inspect(a, x, y, alg=alg)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    f = sp.prepared_multiply(info, a, x, y)
    for _ in range(3): f()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t2) / 10
    pi = info.state_.info()
    print(f"{name:9s} alg {pi['alg']} S {pi['n_slices']} H {pi['rows_per_bin']}: inspect {(t1-t0)*1e3:7.1f} ms, {dt*1e3:7.3f} ms per SpMV, "
          f"{2*nnz/dt/1e9:7.1f} GFLOP/s", flush=True)
    ys[name] = y.clone()
    del info, f
ref = ys["rowblock"].double()
for k in ("sliced", "auto"):
    err = ((ys[k].double() - ref).abs().max() / ref.abs().max()).item()
    print(f"{k}: max |diff| / max |y| vs rowblock = {err:.3e}")
    assert err < 1e-5
print("ok")
