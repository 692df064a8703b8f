"""What would numbering only the non-empty rows buy the tiled plan on cfg4?  Runs the SLICED plan on the R-MAT matrix
as it is and on the same matrix with its empty rows removed (same colind / values, shorter rowptr)."""
import sys, time, torch
sys.path.insert(0, ".")
import spblas_reference_amd as sp
from spblas_reference_amd import generate, _capi
v, rp, ci, shape, nnz = generate.rmat_csr_device(24, 16, dtype=torch.float64, device="cuda")
m, n = shape
keep = (rp[1:] > rp[:-1])
rpc = torch.cat([rp[:-1][keep], rp[-1:]]).contiguous()
mc = rpc.numel() - 1
x = torch.rand(n, dtype=torch.float64, device="cuda")
for name, r, rows in (("original", rp, m), ("compacted", rpc, mc)):
    a = sp.csr_view(v, r, ci, (rows, n), nnz)
    y = torch.empty(rows, dtype=torch.float64, device="cuda")
    info = sp.multiply_inspect(a, x, y, alg=_capi.SPMV_SLICED)
    f = sp.prepared_multiply(info, a, x, y)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    si = info.state_.sliced_info()
    print(f"{name:10s} rows {rows:9d}  {dt*1e3:.3f} ms  bins {si['n_bins']}  pad {si['expand_blocks']*16/si['placed_entries']:.3f}")
    del info, f
