"""Forced SLICED plan on one R-MAT matrix, 12 SpMVs (for a kernel trace): rmat_sliced_once.py <scale> <f32|f64>"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spblas_reference_amd as sp
from spblas_reference_amd import generate, _capi
scale = int(sys.argv[1]); dtype = torch.float32 if sys.argv[2] == "f32" else torch.float64
v, rp, ci, shape, nnz = generate.rmat_csr_device(scale, 16, dtype=dtype, device="cuda", seed=scale)
a = sp.csr_view(v, rp, ci, shape, nnz)
x = torch.rand(shape[1], dtype=dtype, device="cuda"); y = torch.empty(shape[0], dtype=dtype, device="cuda")
info = sp.multiply_inspect(a, x, y, alg=_capi.SPMV_SLICED)
for _ in range(12):
    sp.multiply(info, a, x, y)
torch.cuda.synchronize()
print(info.state_.info(), info.state_.sliced_info())
