"""cfg2's shape in fp64 (10 M x 10 M, 1e8 entries): the SLICED plan the rules pick (one-byte row codes, 8-byte products)
against the plan-free kernel, norm-wise 1e-12, plus timing.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import spblas_reference_amd as sp
from spblas_reference_amd import generate
dev = torch.device("cuda:0")
n = 10_000_000
for poisson in (False, True):
    v, rp, ci, shape, nnz = generate.uniform_csr_device(n, n, 10, dtype=torch.float64, seed=3, device=dev, poisson=poisson)
    a = sp.csr_view(v, rp, ci, shape, nnz)
    x = torch.rand(n, dtype=torch.float64, device=dev) - 0.5
    y0 = torch.empty(n, dtype=torch.float64, device=dev); y1 = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
    sp.multiply(a, x, y0)
    info = sp.multiply_inspect(sp.matrix_opt(a), x, y1)
    sp.multiply(info, a, x, y1); torch.cuda.synchronize()
    absrow = torch.zeros(n, dtype=torch.float64, device=dev)
    sp.multiply(sp.csr_view(v.abs(), rp, ci, shape, nnz), x.abs(), absrow)
    err = ((y1 - y0).abs() / absrow.clamp_min(1e-300)).max().item()
    i = info.state_.info(); s = info.state_.sliced_info() if i["alg"] == 3 else {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): sp.multiply(info, a, x, y1)
    e1.record(); torch.cuda.synchronize()
    print(f"poisson={poisson} alg={i['alg']} row_code_u8={s.get('row_code_u8')} n_slices={i.get('n_slices')} max norm-wise diff vs plan-free {err:.3e}  {e0.elapsed_time(e1)/20:.3f} ms/SpMV")
    assert err <= 1e-12
    del info
