"""cfg2-like matrix plus a few very long rows: AUTO plan and time per SpMV (does one dense row cost the
sliced plan?).  Usage: python tools/hub_check.py [rows] [hub_len]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spblas_reference_amd as sp
dev = torch.device("cuda:0")
m = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
hub = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
g = torch.Generator(device=dev).manual_seed(0)
lens = torch.full((m,), 10, dtype=torch.int64, device=dev)
for r in (7, m // 2, m - 3):
    lens[r] = hub
rp = torch.zeros(m + 1, dtype=torch.int64, device=dev); torch.cumsum(lens, 0, out=rp[1:])
nnz = int(rp[-1])
ci = torch.randint(0, m, (nnz,), dtype=torch.int32, device=dev, generator=g)
v = torch.rand(nnz, device=dev, generator=g)
a = sp.csr_view(v, rp.int(), ci, (m, m), nnz)
x = torch.rand(m, device=dev, generator=g); y = torch.empty(m, device=dev)
info = sp.multiply_inspect(sp.matrix_opt(a), x, y)
f = sp.prepared_multiply(info, a, x, y)
for _ in range(10): f()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): f()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"{m} rows + 3 rows of {hub}: plan {info.state_.info()['alg']}, {dt*1e6:.1f} us, {2*nnz/dt/1e9:.1f} GFLOP/s")
