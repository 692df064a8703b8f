"""cfg2 runs in a fast (~290 us) or a slow (~312 us) mode from process to process on the same box (round 3).  Is the mode
fixed per process or per allocation?  One process: build the plan eight times (dummy allocations of random sizes in between
shift where the arrays land) and time each; then re-create x / y as well."""
import os, sys, time, random
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import spblas_reference_amd as sp
from spblas_reference_amd import generate
os.environ["SPBLAS_GFX950_PB_NT"] = "0"
os.environ["SPBLAS_GFX950_TRACE_INSPECT"] = "1"  # prints where the plan's arrays live (stderr)
dev = torch.device("cuda:0")
n = 10_000_000
v, rp, ci, shape, nnz = generate.uniform_csr_device(n, n, 10, dtype=torch.float32, seed=0, device=dev)
a = sp.csr_view(v, rp, ci, shape, nnz)
x = torch.rand(n, device=dev); y = torch.empty(n, device=dev)
random.seed(int(time.time()))
keep = []
def timeit(info, x, y):
    f = sp.prepared_multiply(info, a, x, y)
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 40 * 1e3
for r in range(12):
    info = sp.multiply_inspect(sp.matrix_opt(a), x, y)
    t = timeit(info, x, y)
    print(f"plan {r}: {t:.1f} us", flush=True)
    del info
    torch.cuda.synchronize()
    keep.append(torch.empty(random.randint(1, 400) * (1 << 20), dtype=torch.uint8, device=dev))  # shift the allocators
    if r % 2 == 1:
        keep.pop(0)
info = sp.multiply_inspect(sp.matrix_opt(a), x, y)
for r in range(4):
    x2 = torch.rand(n, device=dev); y2 = torch.empty(n, device=dev)
    print(f"same plan, new x / y {r}: {timeit(info, x2, y2):.1f} us", flush=True)
    keep.append(x2); keep.append(y2)
