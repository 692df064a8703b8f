"""multiply_inspect (sliced plan) at cfg2, four times: first call, second plan beside the first, and after
freeing (the stream-ordered pool then holds the plan memory): separates allocation cost from kernel time."""
import time, torch, ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import spblas_reference_amd as sp
from spblas_reference_amd import generate, _capi
dev = torch.device("cuda:0")
n = 10_000_000
v, rp, ci, shape, nnz = generate.uniform_csr_device(n, n, 10, dtype=torch.float32, seed=0, device=dev)
a = sp.csr_view(v, rp, ci, shape, nnz)
x = torch.rand(n, device=dev); y = torch.empty(n, device=dev)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    info = sp.multiply_inspect(sp.matrix_opt(a), x, y)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"inspect #{rep}: {(t1-t0)*1e3:.2f} ms", flush=True)
    if rep % 2 == 1:
        del info
        torch.cuda.synchronize()
