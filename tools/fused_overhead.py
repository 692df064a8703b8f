"""Single-process measurement of what the fused step adds on top of the plain local SpMV of one row shard
(peer table with one entry, step_signal + step_wait kernels).  Usage: python tools/fused_overhead.py [rows]"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spblas_reference_amd as sp
from spblas_reference_amd import generate, sharded
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29733")
dist.init_process_group("gloo", rank=0, world_size=1)
dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
n = 10_000_000
v, rp, ci, shape, nnz = generate.uniform_csr_device(rows, n, 10, seed=0, device=dev)
a = sp.csr_view(v, rp, ci, shape, nnz)
x = torch.rand(n, device=dev); y = torch.empty(rows, device=dev)
info = sp.multiply_inspect(sp.matrix_opt(a), x, y)
plain = sp.prepared_multiply(info, a, x, y)
fused = sharded.FusedShardedSpMV(a, [0, rows], info=info)
striped = {st: sharded.FusedShardedSpMV(a, [0, rows], info=info, stripes=st) for st in (2, 4, 8)}
def timeit(fn, steps=300):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e6
print(f"rows {rows}: plain {timeit(plain):.1f} us, fused(P=1) {timeit(lambda: fused.step(x)):.1f} us, " +
      ", ".join(f"{st} stripes {timeit(lambda op=op: op.step(x)):.1f} us" for st, op in striped.items()))
def cpu_time(fn, steps=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    t = (time.perf_counter() - t0) / steps * 1e6
    torch.cuda.synchronize(); return t
print("host time per call while the GPU queue is never waited for: plain %.1f us, fused %.1f us, 4 stripes %.1f us"
      % (cpu_time(plain), cpu_time(lambda: fused.step(x)), cpu_time(lambda: striped[4].step(x))))
for op in striped.values():
    op.check_status(); op.close()
fused.check_status(); fused.close(); dist.destroy_process_group()
