"""Conjugate gradients with the whole iteration in one HIP graph.

The call shape is the reference's inspect / execute split (README.md:36-46 of the reference; examples/device/
matrix_opt_example.cpp): multiply_inspect once, multiply(info, A, p, q) every iteration.  On this backend the execute call
only launches kernels on the current stream, so one CG iteration -- SpMV, two dot products, three vector updates, all scalars
kept on the device -- can be recorded once with torch.cuda.graph and replayed; the host then issues one graph launch per
iteration instead of a dozen kernel launches.

    python examples/cg_graph.py [n] [iterations]
"""
import os
import sys

import numpy as np
import scipy.sparse as sps
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spblas_reference_amd as sp  # noqa: E402


def spd_matrix(n, per_row, seed=0):
    """Symmetric, strictly diagonally dominant: off-diagonal pattern random, diagonal = 1 + sum |row|."""
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n), per_row)
    cols = rng.integers(0, n, n * per_row)
    vals = rng.random(n * per_row) * 0.5
    a = sps.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    a = a + a.T
    a.setdiag(0)
    a.eliminate_zeros()
    d = np.asarray(abs(a).sum(axis=1)).ravel() + 1.0
    return (a + sps.diags(d)).tocsr()


def main(n=200_000, iterations=60):
    dev = torch.device("cuda:0")
    a_h = spd_matrix(n, 6)
    t = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x.astype(dt))).to(dev)
    a = sp.csr_view(t(a_h.data, np.float64), t(a_h.indptr, np.int32), t(a_h.indices, np.int32), a_h.shape, a_h.nnz)
    b_h = np.random.default_rng(1).random(n)
    b = t(b_h, np.float64)
    x = torch.zeros(n, dtype=torch.float64, device=dev)
    r = b.clone()
    p = r.clone()
    q = torch.empty_like(p)
    rs = (r * r).sum()  # device scalars throughout: nothing in the loop needs the host
    info = sp.multiply_inspect(sp.matrix_opt(a), p, q)  # outside the graph: inspect allocates and sizes things

    def iteration():
        nonlocal rs
        sp.multiply(info, a, p, q)          # q = A p
        alpha = rs / (p * q).sum()
        x.add_(alpha * p)
        r.sub_(alpha * q)
        rs_new = (r * r).sum()
        p.mul_(rs_new / rs).add_(r)
        rs.copy_(rs_new)

    side = torch.cuda.Stream()              # warm-up on a side stream (the usual torch recipe), then record
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        iteration()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        iteration()
    for _ in range(iterations):
        graph.replay()
    torch.cuda.synchronize()
    res = float(torch.linalg.norm(b - torch.from_numpy(a_h @ x.cpu().numpy()).to(dev)) / torch.linalg.norm(b))
    print(f"CG on a {n} x {n} SPD matrix ({a_h.nnz} entries): {iterations + 2} iterations, relative residual {res:.3e}")
    return res


if __name__ == "__main__":
    args = [int(v) for v in sys.argv[1:3]]
    sys.exit(0 if main(*args) < 1e-8 else 1)
