"""A/B of the expand kernel's A' loads (non-temporal vs cached: -DPB_EXP_PLAIN_A) over matrix classes and value types:
ms per multiply of the SLICED snapshot plan, and the plan's padding (p_pad / a_pad).  Run under SPBLAS_GFX950_LIB=<variant>."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spblas_reference_amd as sp
from spblas_reference_amd import _capi, generate

def run(name, values, rowptr, colind, shape, nnz):
    a = sp.csr_view(values, rowptr, colind, shape, nnz)
    x = torch.rand(shape[1], dtype=values.dtype, device="cuda")
    y = torch.empty(shape[0], dtype=values.dtype, device="cuda")
    info = sp.multiply_inspect(sp.matrix_opt(a), x, y, alg=_capi.SPMV_SLICED)
    for _ in range(3):
        sp.multiply(info, a, x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        sp.multiply(info, a, x, y)
    e1.record()
    torch.cuda.synchronize()
    si = info.state_.sliced_info()
    print(f"{os.path.basename(os.environ.get('SPBLAS_GFX950_LIB', 'default'))} {name}: {e0.elapsed_time(e1) / 20:.4f} ms  "
          f"hot_split={'hot_split' in si} slices={si.get('n_slices')}", flush=True)

for dt, dn in ((torch.float32, "f32"), (torch.float64, "f64")):
    run(f"uniform 10M x 10M, 10/row {dn}", *generate.uniform_csr_device(10_000_000, 10_000_000, 10, dtype=dt, seed=0))
    run(f"uniform 4M x 4M, 32/row {dn}", *generate.uniform_csr_device(4_000_000, 4_000_000, 32, dtype=dt, seed=0))
    run(f"rmat 24 x16 {dn}", *generate.rmat_csr_device(24, 16, dtype=dt, seed=0))
    run(f"rmat 22 x32 {dn}", *generate.rmat_csr_device(22, 32, dtype=dt, seed=0))
