"""Round 5: what a PLAIN inspected csr_view gets at several shapes (plan, value-free or not, ms per SpMV, parity on sampled
rows with values rewritten in place after inspect).  Run on the GPU box: python tools/checks/plain_shapes.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import spblas_reference_amd as sp
from spblas_reference_amd import generate
from oracle import oracle
from bench_extra import rows_subproblem, parity_rows

dev = torch.device("cuda:0")
cases = [  # (rows, cols, per_row, dtype, poisson, offset64)
    (1_700_000, 1_700_000, 10, torch.float32, False, False),
    (4_000_000, 4_000_000, 6, torch.float32, True, False),
    (3_000_000, 12_000_000, 10, torch.float64, False, False),
    (6_000_000, 6_000_000, 10, torch.float64, True, True),
    (2_000_000, 2_000_000, 40, torch.float32, True, False),
    (20_000_000, 20_000_000, 3, torch.float32, True, False),
    (10_000_000, 10_000_000, 10, torch.float32, True, False),
]
for (m, n, per, dt, poisson, o64) in cases:
    values, rowptr, colind, shape, nnz = generate.uniform_csr_device(m, n, per, seed=1, dtype=dt, device=dev, poisson=poisson,
                                                                      offset_dtype=torch.int64 if o64 else torch.int32)
    a = sp.csr_view(values, rowptr, colind, shape, nnz)
    x = torch.rand(n, dtype=dt, device=dev)
    y = torch.full((m,), float("nan"), dtype=dt, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    info = sp.multiply_inspect(a, x, y)
    torch.cuda.synchronize(); insp = (time.perf_counter() - t0) * 1e3
    values.mul_(-0.5).add_(0.125)
    for _ in range(3):
        sp.multiply(info, a, x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        sp.multiply(info, a, x, y)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    pi = info.state_.info(); si = info.state_.sliced_info()
    rows = np.unique(np.concatenate([np.arange(0, 1000), np.arange(m - 1000, m), np.random.default_rng(3).integers(0, m, 2000)]))
    sub_rp, sub_ci, sub_v = rows_subproblem(rows, rowptr, colind, values)
    xh = x.cpu().numpy()
    ref = oracle.spmv((len(rows), n), sub_rp, sub_ci, sub_v, xh)
    absrow = oracle.spmv_absrow(sub_rp, sub_ci, sub_v, xh)
    tol, eps = (1e-6, np.finfo(np.float32).eps) if dt == torch.float32 else (1e-12, np.finfo(np.float64).eps)
    nbad, worst = parity_rows(y[torch.from_numpy(rows).to(dev)].cpu().numpy(), ref, absrow.astype(np.float64), tol, float(eps), np.diff(sub_rp))
    alg_bytes = nnz * (values.element_size() + 4) + (m + 1) * rowptr.element_size() + (n + m) * values.element_size()
    print(f"{m:>9} x {n:<9} {per:>2}/row {'poisson' if poisson else 'exact  '} {str(dt)[6:]:8s} o64={int(o64)} nnz={nnz:>10}: alg {pi['alg']} "
          f"value_free {si.get('value_free', 0)} bins {si.get('n_bins', 0)} H {pi.get('rows_per_bin')} u8 {si.get('row_code_u8', 0)} "
          f"inspect {insp:6.1f} ms  {ms:.3f} ms = {alg_bytes / ms / 1e6 / 8000 * 100:.1f} % of 8 TB/s  plan {pi['device_bytes'] / 1e9:.2f} GB  "
          f"parity {'pass' if nbad == 0 else 'FAIL'} ({worst:.1e})", flush=True)
    del info, a, values, rowptr, colind, x, y
    torch.cuda.empty_cache()
