"""Synthetic inputs for tests and bench (data plumbing: numpy on host, torch on device).

generate_csr imitates the DISTRIBUTION of the reference's test generator
(/root/reference/include/spblas/backend/generate.hpp:49-120: nnz distinct (i,j) drawn
uniformly, values U[0,100), columns left UNSORTED within each row) -- not its exact
mt19937 stream.  The *_device generators build BASELINE.json's large configs directly
in HBM; the reference generator is O(nnz log nnz) with a
std::set node per entry and cannot reach 1e8 nonzeros.
"""
import numpy as np
import torch


def generate_csr(m, n, nnz, seed=0, dtype=np.float32, sorted_cols=False):
    rng = np.random.default_rng(seed)
    nnz = min(nnz, m * n)
    # distinct flat positions
    if m * n <= 4 * nnz:
        flat = rng.permutation(m * n)[:nnz]
    else:
        flat = np.unique(rng.integers(0, m * n, size=int(nnz * 1.2) + 16))
        while flat.shape[0] < nnz:
            flat = np.unique(np.concatenate([flat, rng.integers(0, m * n, size=nnz)]))
        flat = rng.permutation(flat)[:nnz]
    flat = np.sort(flat)
    rows = (flat // n).astype(np.int64)
    cols = (flat % n).astype(np.int32)
    values = (rng.random(nnz) * 100).astype(dtype)
    rowptr = np.zeros(m + 1, dtype=np.int32)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr, dtype=np.int64).astype(np.int32)
    if not sorted_cols:
        for r in range(m):  # shuffle the column order inside each row (generate.hpp:112-117)
            a, b = rowptr[r], rowptr[r + 1]
            if b - a > 1:
                perm = rng.permutation(b - a)
                cols[a:b] = cols[a:b][perm]
    return values, rowptr, cols, (m, n), nnz


def generate_dense(m, n, seed=0, dtype=np.float32):
    rng = np.random.default_rng(seed + 7919)
    return (rng.random((m, n)) * 100).astype(dtype)


def uniform_csr_device(m, n, nnz_per_row, dtype=torch.float32, seed=0, device="cuda", poisson=False,
                       offset_dtype=torch.int32):
    """cfg2/cfg3/cfg5: m x n, columns iid U[0,n) unsorted within a row, values U[0,1).
    Row lengths exactly nnz_per_row, or Poisson(nnz_per_row) when poisson=True."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if poisson:
        lens = torch.poisson(torch.full((m,), float(nnz_per_row), device=device), generator=g).to(torch.int64)
    else:
        lens = torch.full((m,), int(nnz_per_row), dtype=torch.int64, device=device)
    rowptr = torch.zeros(m + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens, 0, out=rowptr[1:])
    nnz = int(rowptr[-1].item())
    colind = torch.randint(0, n, (nnz,), dtype=torch.int32, device=device, generator=g)
    values = torch.rand(nnz, dtype=dtype, device=device, generator=g)
    return values, rowptr.to(offset_dtype), colind, (m, n), nnz


def rmat_csr_device(scale, edge_factor=16, abcd=(0.57, 0.19, 0.19, 0.05), dtype=torch.float64, seed=0,
                    device="cuda", chunk=1 << 26):
    """cfg4: R-MAT graph, 2**scale vertices, edge_factor * 2**scale edges, duplicates KEPT
    (nnz is exactly edge_factor * 2**scale), values U[0,1), columns unsorted within a row."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = 1 << scale
    nedges = edge_factor * n
    a, b, c, _ = abcd
    rows_all, cols_all = [], []
    for start in range(0, nedges, chunk):
        cnt = min(chunk, nedges - start)
        r = torch.zeros(cnt, dtype=torch.int64, device=device)
        cc = torch.zeros(cnt, dtype=torch.int64, device=device)
        for _bit in range(scale):
            u = torch.rand(cnt, device=device, generator=g)
            rbit = (u >= a + b).to(torch.int64)                       # quadrants c, d -> lower half
            cbit = (((u >= a) & (u < a + b)) | (u >= a + b + c)).to(torch.int64)  # quadrants b, d
            r = (r << 1) | rbit
            cc = (cc << 1) | cbit
        rows_all.append(r)
        cols_all.append(cc.to(torch.int32))
    rows = torch.cat(rows_all)
    cols = torch.cat(cols_all)
    del rows_all, cols_all
    order = torch.argsort(rows, stable=True)
    cols = cols[order]
    counts = torch.bincount(rows, minlength=n)
    del rows, order
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    values = torch.rand(nedges, dtype=dtype, device=device, generator=g)
    return values, rowptr.to(torch.int32), cols, (n, n), nedges
