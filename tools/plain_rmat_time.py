import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import spblas_reference_amd as sp
from spblas_reference_amd import _capi, generate
values, rowptr, colind, shape, nnz = generate.rmat_csr_device(24, 16, dtype=torch.float64, seed=0)
a = sp.csr_view(values, rowptr, colind, shape, nnz)
n = shape[0]
x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.empty(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize(); t0=time.perf_counter(); info = sp.multiply_inspect(a, x, y); torch.cuda.synchronize(); print("inspect ms", (time.perf_counter()-t0)*1e3)
pi = info.state_.info(); print("alg", pi["alg"], {k: v for k, v in info.state_.sliced_info().items() if k in ("refresh_each_call","trial_rowblock_ns","trial_sliced_ns","auto_trial")})
for _ in range(3): sp.multiply(info, a, x, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): sp.multiply(info, a, x, y)
e1.record(); torch.cuda.synchronize(); print("multiply ms", e0.elapsed_time(e1)/10)
