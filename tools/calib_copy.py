"""Known-byte-count kernels for calibrating FETCH_SIZE / WRITE_SIZE on gfx950
(MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by 2x; other widths and
WRITE_SIZE are uncalibrated).  Runs (1) a 1 GiB float4-style device copy (torch clone:
read N, write N) and (2) a 4-byte random gather of 1e8 elements from a 40 MB table
(the SpMV x-gather pattern: index stream read + gathered reads + result write)."""
import torch

N = 1 << 28  # 268M floats = 1 GiB
a = torch.rand(N, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    b = a.clone()          # CALIB_COPY: reads 4N bytes, writes 4N bytes
torch.cuda.synchronize()
x = torch.rand(10_000_000, device="cuda")
idx = torch.randint(0, 10_000_000, (100_000_000,), device="cuda", dtype=torch.int64)
torch.cuda.synchronize()
for _ in range(3):
    g = x[idx]             # CALIB_GATHER: reads 8e8 (idx) + 4e8 useful gathered, writes 4e8
torch.cuda.synchronize()
print("calib done", float(b[0]), float(g[0]))
