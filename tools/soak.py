"""Run-to-run stability of the cfg2 SpMV inside one process: 40 x 50 steps with idle gaps (prints GFLOP/s per
block).  Inside one box the rate is stable to 0.2 %; the 560-620 GFLOP/s spread of the round is box to box."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import spblas_reference_amd as sp
from spblas_reference_amd import generate
dev = torch.device("cuda:0")
n = 10_000_000
v, rp, ci, shape, nnz = generate.uniform_csr_device(n, n, 10, dtype=torch.float32, seed=0, device=dev)
a = sp.csr_view(v, rp, ci, shape, nnz)
x = torch.rand(n, device=dev); y = torch.empty(n, device=dev)
info = sp.multiply_inspect(sp.matrix_opt(a), x, y)
f = sp.prepared_multiply(info, a, x, y)
for _ in range(5): f()
torch.cuda.synchronize()
out = []
t_start = time.perf_counter()
for rep in range(int(os.environ.get('SOAK_BLOCKS', '40'))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    out.append(round(2 * nnz / ms / 1e6, 1))
    if rep in (9, 19, 29): time.sleep(2.0)   # idle gaps
print("elapsed %.1f s" % (time.perf_counter() - t_start))
print(out)
print("median", sorted(out)[len(out) // 2])
