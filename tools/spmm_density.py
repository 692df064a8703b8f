"""Crossover of the SpMM kernels on banded matrices: entries per row (all within +-48 columns of the diagonal, i.e. a
block of 32 rows touches 2-3 tiles of 64 columns) vs time of (a) the LDS-staged matrix-core kernel forced on every
block, (b) the row-group gather kernel.  n = 128, fp32, 1 M rows.  Run in ONE gpurun call."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, time, torch
sys.path.insert(0, %r)
import spblas_reference_amd as sp
m, per = 1_000_000, int(sys.argv[1])
g = torch.Generator(device="cuda").manual_seed(1)
off = torch.randint(-48, 49, (m, per), device="cuda", generator=g)
colind = ((torch.arange(m, device="cuda")[:, None] + off) %% m).to(torch.int32).reshape(-1)
rowptr = (torch.arange(m + 1, device="cuda", dtype=torch.int64) * per).to(torch.int32)
values = torch.rand(m * per, device="cuda", generator=g)
a = sp.csr_view(values, rowptr, colind, (m, m), m * per)
B = torch.rand((m, 128), device="cuda", generator=g); C = torch.empty((m, 128), device="cuda")
info = sp.multiply_inspect(a, B, C)
for _ in range(3): sp.multiply(info, a, B, C)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): sp.multiply(info, a, B, C)
torch.cuda.synchronize()
print((time.perf_counter() - t0) / 10 * 1e3, info.state_.spmm_info()["panel_blocks"])
''' % ROOT
print(f"{'nnz/row':>8} {'density':>8} {'panel ms':>9} {'gather ms':>10}")
for per in (4, 8, 12, 16, 24, 32, 48, 64, 96):
    res = []
    for env in ({"SPBLAS_GFX950_SPMM_PANEL_MIN": "1"}, {"SPBLAS_GFX950_SPMM_NO_PANEL": "1"}):
        r = subprocess.run([sys.executable, "-c", code, str(per)], capture_output=True, text=True, env=dict(os.environ, **env))
        res.append(r.stdout.strip().split() if r.returncode == 0 else ["nan", r.stderr[-200:]])
    print(f"{per:8d} {per / (2.5 * 64):8.3f} {float(res[0][0]):9.3f} {float(res[1][0]):10.3f}   (panel blocks {res[0][1]})")
