"""Debug aid: triangular solve captured in a graph, replayed; prints where the replays differ from the direct call."""
import os, sys
import numpy as np, scipy.sparse as sps, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import spblas_reference_amd as sp
import gpu_util as G
rng = np.random.default_rng(21); n = 4000
A = sps.random(n, n, density=0.002, format="csr", random_state=rng, dtype=np.float64)
S = sps.tril(A, -1); d = np.asarray(abs(S).sum(axis=1)).ravel() + 1.0 + rng.random(n)
M = (S + sps.diags(d)).tocsr()
vals = M.data.astype(np.float32); rp, ci = M.indptr.astype(np.int32), M.indices.astype(np.int32)
a = G.csr_on_device(vals, rp, ci, M.shape, M.nnz)
b = torch.zeros(n, device="cuda"); x = torch.full((n,), float("nan"), device="cuda")
info = sp.triangular_solve_inspect(a, sp.lower_triangle, sp.explicit_diagonal, b, x)
print("info", info.state_.info())
def solve(): sp.triangular_solve(info, a, sp.lower_triangle, sp.explicit_diagonal, b, x)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): solve()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): solve()
for seed in range(4):
    b.copy_(torch.rand(n, device="cuda")); x.fill_(float("nan")); g.replay(); torch.cuda.synchronize()
    xg = x.clone(); x.fill_(float("nan")); solve(); torch.cuda.synchronize()
    nan = torch.isnan(xg).nonzero().flatten()
    print(seed, "nan rows in replay:", nan.numel(), nan[:5].tolist(), "max diff", (xg - x).abs().nan_to_num(0).max().item())
