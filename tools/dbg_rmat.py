import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spblas_reference_amd as sp
from spblas_reference_amd import generate
scale = int(sys.argv[1]); dt = torch.float64 if sys.argv[2] == "f64" else torch.float32
dev = torch.device("cuda:0")
values, rowptr, colind, shape, nnz = generate.rmat_csr_device(scale, 16, dtype=dt, seed=0, device=dev)
print("generated", shape, nnz, rowptr.dtype, int((rowptr[1:] - rowptr[:-1]).max()), flush=True)
a = sp.csr_view(values, rowptr, colind, shape, nnz)
x = torch.rand(shape[1], dtype=dt, device=dev); y = torch.empty(shape[0], dtype=dt, device=dev)
info = sp.multiply_inspect(a, x, y, alg=sp._capi.SPMV_SLICED)
torch.cuda.synchronize(); print("inspect ok", info.state_.info(), flush=True)
if os.environ.get("DBG_BIN"):
    inf = info.state_.info(); H = inf["rows_per_bin"]; S = inf["n_slices"]
    W = -(-shape[1] // S); W = (W + 3) // 4 * 4
    for b in [int(v) for v in os.environ["DBG_BIN"].split(",")]:
        r0, r1 = b * H, min((b + 1) * H, shape[0])
        p0, p1 = int(rowptr[r0]), int(rowptr[r1])
        cols = colind[p0:p1].long()
        rows = torch.repeat_interleave(torch.arange(r0, r1, device=dev), (rowptr[r0 + 1:r1 + 1] - rowptr[r0:r1]).long())
        sl = cols // W
        cnt = torch.bincount(sl, minlength=S)
        key = rows * shape[1] + cols
        uniq, c = torch.unique(key, return_counts=True)
        print("bin", b, "rows", r0, r1, "entries", p1 - p0, "W", W, "per-slice", cnt.tolist(), "max dup", int(c.max()) if len(c) else 0,
              "max row len", int((rowptr[r0 + 1:r1 + 1] - rowptr[r0:r1]).max()), flush=True)
if len(sys.argv) > 3:
    ex, rd = info.state_.bind_stages(x, y.data_ptr(), dt)
    ex(); torch.cuda.synchronize(); print("expand ok", flush=True)
    H = info.state_.info()["rows_per_bin"]
    step = H * int(os.environ.get('DBG_BINS', '8'))
    for lo in range(int(os.environ.get('DBG_LO', '0')), shape[0], step):
        rd(lo, min(shape[0], lo + step)); torch.cuda.synchronize(); print("reduce ok", lo // H, flush=True)
else:
    sp.multiply(info, a, x, y); torch.cuda.synchronize(); print("exec ok", flush=True)
from oracle import oracle
ref = oracle.spmv(shape, rowptr.cpu().numpy(), colind.cpu().numpy(), values.cpu().numpy(), x.cpu().numpy())
err = np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max(); print("rel err", err)
