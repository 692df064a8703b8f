"""cfg4: how much would pre-summing the products of one row inside one x slice shrink the product stream of the tiles?
(round 4 probe: distinct (row, slice) pairs against entries, before and after the hot columns are taken out)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spblas_reference_amd import generate
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
values, rowptr, colind, shape, nnz = generate.rmat_csr_device(scale, 16, dtype=torch.float64, seed=0)
n = shape[0]
rows = torch.repeat_interleave(torch.arange(n, device="cuda"), (rowptr[1:] - rowptr[:-1]).long())
cnt = torch.bincount(colind.long(), minlength=n)
hot = torch.zeros(n, dtype=torch.bool, device="cuda")
hot[torch.topk(cnt, 16320).indices] = True
for W in (20480,):
    for name, mask in (("all entries", torch.ones(nnz, dtype=torch.bool, device="cuda")), ("without the 16 320 hottest columns", ~hot[colind.long()])):
        r, c = rows[mask], colind[mask].long()
        key = r * ((n + W - 1) // W) + c // W
        uniq = torch.unique(key).numel()
        print(f"W={W} {name}: entries {r.numel()}, distinct (row, slice) pairs {uniq} = {uniq / r.numel():.3f} of the entries")
        # by row length class
        lens = (rowptr[1:] - rowptr[:-1]).long()
        for lo, hi in ((0, 64), (64, 1024), (1024, 16384), (16384, 1 << 30)):
            sel = (lens[r] >= lo) & (lens[r] < hi)
            e = int(sel.sum())
            if e:
                u = torch.unique(key[sel]).numel()
                print(f"     rows of {lo}..{hi} entries: {e} entries ({e / r.numel():.3f}), pairs / entries {u / e:.3f}")
