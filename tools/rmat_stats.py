"""Row / column skew of the cfg4 matrix (R-MAT scale 24, edge factor 16): what a tiled plan has to cope with."""
import sys, torch
sys.path.insert(0, ".")
import spblas_reference_amd as sp
from spblas_reference_amd import generate
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
v, rp, ci, shape, nnz = generate.rmat_csr_device(scale, 16, dtype=torch.float64, device="cuda")
m = shape[0]
rl = (rp[1:] - rp[:-1]).long()
print("m", m, "nnz", nnz, "max row", int(rl.max()), "empty rows", int((rl == 0).sum()))
for t in (256, 1024, 4096, 16384, 27000, 65536):
    sel = rl > t
    print(f"rows > {t:6d}: {int(sel.sum()):8d} rows, {int(rl[sel].sum()) / nnz * 100:6.2f} % of nnz")
W = 20480
sl = torch.bincount((ci.long() // W), minlength=(m + W - 1) // W)
print("slices", sl.numel(), "max", int(sl.max()), "avg", float(sl.float().mean()), "top-8 slices share %.2f %%" % (float(sl.topk(8).values.sum()) / nnz * 100))
cl = torch.bincount(ci.long(), minlength=m)
top = cl.topk(20480).values.sum()
print("hottest 20480 columns: %.2f %% of nnz; hottest 163840: %.2f %%" % (float(top) / nnz * 100, float(cl.topk(163840).values.sum()) / nnz * 100))
# sorted columns inside rows?
d = ci[1:] >= ci[:-1]
print("adjacent entries ascending: %.3f" % float(d.float().mean()))
H = 2441
nb = (m + H - 1) // H
bins = torch.bincount(torch.repeat_interleave(torch.arange(m, device="cuda") // H, rl), minlength=nb)
print("bins", nb, "max", int(bins.max()), "avg", float(bins.float().mean()), "bins > 4x avg:", int((bins > 4 * bins.float().mean()).sum()))
