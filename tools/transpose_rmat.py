"""Transpose of an R-MAT matrix (skewed columns, many empty rows / columns): time and involution check.
python tools/transpose_rmat.py [scale] [f32|f64]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import spblas_reference_amd as sp
from spblas_reference_amd import generate

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
dt = torch.float64 if len(sys.argv) > 2 and sys.argv[2] == "f64" else torch.float32
dev = torch.device("cuda:0")
v, rp, ci, shape, nnz = generate.rmat_csr_device(scale, 16, dtype=dt, device=dev)
a = sp.csr_view(v, rp, ci, shape, nnz)
def empty(n_rows):
    return sp.csr_view(torch.empty(nnz, dtype=dt, device=dev), torch.empty(n_rows + 1, dtype=torch.int32, device=dev),
                       torch.empty(nnz, dtype=torch.int32, device=dev), (shape[1], shape[0]), nnz)
t, tt = empty(shape[1]), empty(shape[0])
sp.transpose(a, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    sp.transpose(a, t)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
sp.transpose(t, tt)
torch.cuda.synchronize()
# (A^T)^T has A's rows with the entries of each row sorted by column (stable): compare as sorted rows
ok_rp = bool((tt.rowptr() == rp).all().item())
key = lambda r, c: r.long() * shape[1] + c.long()
rows = torch.repeat_interleave(torch.arange(shape[0], device=dev), (rp[1:] - rp[:-1]).long())
k0, o0 = torch.sort(key(rows, ci), stable=True)
k1 = key(rows, tt.colind())
same = bool((k0 == k1).all().item()) and bool((v[o0] == tt.values()).all().item())
print(f"R-MAT scale {scale} {dt}: nnz {nnz}, transpose {ms:.3f} ms = {nnz / ms / 1e6:.1f} G entries/s; "
      f"(A^T)^T rowptr equal {ok_rp}, entries equal {same}")
