"""How long does a value refresh of the SLICED plan take at cfg2 (python tools/update_values_time.py [rows])?
The plan multiplies with its own re-tiled copy of A's values; spblas_gfx950_spmv_plan_update_values takes them again
from the caller's array through the recorded source positions."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spblas_reference_amd as sp
from spblas_reference_amd import _capi, generate

if len(sys.argv) > 1 and sys.argv[1] == "rmat":  # cfg4: fp64 R-MAT scale 24 (hot-column split + tiles over a row map)
    values, rowptr, colind, shape, nnz = generate.rmat_csr_device(24, 16, dtype=torch.float64, seed=0)
    m = shape[0]
else:
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    values, rowptr, colind, shape, nnz = generate.uniform_csr_device(m, m, 10, seed=0)
a = sp.csr_view(values, rowptr, colind, shape, nnz)
x = torch.rand(m, dtype=values.dtype, device="cuda")
y = torch.empty(m, dtype=values.dtype, device="cuda")
info = sp.multiply_inspect(sp.matrix_opt(a), x, y)
print("plan", info.state_.info()["alg"], info.state_.sliced_info().get("row_code_u8"))
hd = sp.api._Handle.current(x.device)
lib = _capi.lib()
def upd():
    sp.api.check(lib.spblas_gfx950_spmv_plan_update_values(hd.h, info.state_.plan, sp.api._ptr(values)), "update")
for name, fn in (("update_values", upd), ("multiply", lambda: sp.multiply(info, a, x, y))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 10:.3f} ms")
