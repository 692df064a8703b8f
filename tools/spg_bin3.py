"""Repeated SpGEMM fills when every row has 257..1024 products (bin 3: two-byte ranks), e.g. 27 entries per row:
python tools/spg_bin3.py [rows] [per_row].  Prints the hash fill, the recording fill and the fills by rank."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import spblas_reference_amd as sp
from spblas_reference_amd import generate

m = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
per = int(sys.argv[2]) if len(sys.argv) > 2 else 27
dev = torch.device("cuda:0")
av, ar, ac, ash, annz = generate.uniform_csr_device(m, m, per, seed=0, device=dev)
bv, br, bc, bsh, bnnz = generate.uniform_csr_device(m, m, per, seed=1, device=dev)
a, b = sp.csr_view(av, ar, ac, ash, annz), sp.csr_view(bv, br, bc, bsh, bnnz)
c_rp = torch.zeros(m + 1, dtype=torch.int32, device=dev)
c = sp.csr_view(None, c_rp, None, (m, m), 0)
state = sp.spgemm_state_t()
sp.multiply_compute(state, a, b, c)
cn = state.result_nnz()
c.update(torch.empty(cn, device=dev), c_rp, torch.empty(cn, dtype=torch.int32, device=dev), (m, m), cn)
times = []
for it in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sp.multiply_fill(state, a, b, c)
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
ref = c.values().clone()
os.environ["SPBLAS_GFX950_SPGEMM_REUSE"] = "0"
state2 = sp.spgemm_state_t()
c2_rp = torch.zeros(m + 1, dtype=torch.int32, device=dev)
c2 = sp.csr_view(None, c2_rp, None, (m, m), 0)
sp.multiply_compute(state2, a, b, c2)
c2.update(torch.empty(cn, device=dev), c2_rp, torch.empty(cn, dtype=torch.int32, device=dev), (m, m), cn)
sp.multiply_fill(state2, a, b, c2)
torch.cuda.synchronize()
err = ((c2.values() - ref).abs() / (c2.values().abs() + 1e-30)).max().item()
same_cols = bool((c2.colind() == c.colind()).all().item())
print(f"rows {m} x {per}/row, products/row {per * per}, nnz(C) {cn}: fills ms " + " ".join(f"{t:.3f}" for t in times) +
      f" | hash vs rank: max rel diff {err:.2e}, columns equal {same_cols}")
