"""The workspace search of a large plan under memory pressure: with only ~2.4 GB free the plan (1.6 GB) fits but not all
four candidates for its product workspace -- the search must stop at what it gets and the plan must work."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import spblas_reference_amd as sp
from spblas_reference_amd import generate, _capi
dev = torch.device("cuda:0")
n = 10_000_000
v, rp, ci, shape, nnz = generate.uniform_csr_device(n, n, 10, seed=0, device=dev)
a = sp.csr_view(v, rp, ci, shape, nnz)
x = torch.rand(n, device=dev); y0 = torch.empty(n, device=dev); y1 = torch.full((n,), float("nan"), device=dev)
sp.multiply(a, x, y0)
torch.cuda.synchronize(); torch.cuda.empty_cache()
free, total = torch.cuda.mem_get_info()
hog = torch.empty(free - int(2.4 * 2**30), dtype=torch.uint8, device=dev)
print("free before inspect: %.2f GB" % (torch.cuda.mem_get_info()[0] / 2**30))
info = sp.multiply_inspect(sp.matrix_opt(a), x, y1)
i, s = info.state_.info(), info.state_.sliced_info()
print("alg", i["alg"], "candidates tested", s["workspace_candidates"], "free after: %.2f GB" % (torch.cuda.mem_get_info()[0] / 2**30))
sp.multiply(info, a, x, y1); torch.cuda.synchronize()
print("max rel diff vs plan-free:", float(((y1 - y0).abs() / y0.abs().clamp_min(1e-30)).max()))
