"""fp64 counterpart of cfg2 (10M x 10M, 10 per row): sliced vs row-block plan, us per SpMV."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spblas_reference_amd as sp
from spblas_reference_amd import generate, _capi
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
v, rp, ci, shape, nnz = generate.uniform_csr_device(n, n, 10, dtype=torch.float64, seed=0, device=dev)
a = sp.csr_view(v, rp, ci, shape, nnz)
x = torch.rand(n, dtype=torch.float64, device=dev); y = torch.empty(n, dtype=torch.float64, device=dev)
for name, alg in (("auto", _capi.SPMV_AUTO), ("rowblock", _capi.SPMV_ROWBLOCK), ("sliced", _capi.SPMV_SLICED)):
    info = sp.multiply_inspect(a, x, y, alg=alg)
    f = sp.prepared_multiply(info, a, x, y)
    for _ in range(10): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    pi = info.state_.info()
    print(f"{name:9s} plan alg {pi['alg']} H {pi['rows_per_bin']} S {pi['n_slices']}: {dt*1e6:8.1f} us  {2*nnz/dt/1e9:7.1f} GFLOP/s  "
          f"{(nnz*12+(n+1)*4+2*n*8)/dt/1e12:5.2f} TB/s algorithmic")
    del info, f
