"""ms per SpMV of the forced SLICED and ROWBLOCK plans on one R-MAT matrix: rmat_sliced_time.py <scale> <f32|f64>"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spblas_reference_amd as sp
from spblas_reference_amd import generate, _capi
scale = int(sys.argv[1]); dtype = torch.float32 if sys.argv[2] == "f32" else torch.float64
v, rp, ci, shape, nnz = generate.rmat_csr_device(scale, 16, dtype=dtype, device="cuda", seed=scale)
a = sp.csr_view(v, rp, ci, shape, nnz)
x = torch.rand(shape[1], dtype=dtype, device="cuda"); y = torch.empty(shape[0], dtype=dtype, device="cuda")
out = []
for name, alg in (("sliced", _capi.SPMV_SLICED), ("rowblock", _capi.SPMV_ROWBLOCK)):
    info = sp.multiply_inspect(a, x, y, alg=alg)
    f = sp.prepared_multiply(info, a, x, y)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); out.append(f"{name} {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
    if alg == _capi.SPMV_SLICED:
        si = info.state_.sliced_info(); out.append(f"hub {si['hub_rows']} bins {si['n_bins']}")
    del f, info
print(f"scale {scale} {sys.argv[2]}:", "; ".join(out), {k: v for k, v in os.environ.items() if k.startswith("SPBLAS_GFX950_PB")})
