"""-m gpu: plans, states and handles give their device memory back.

Every inspect / compute call allocates backend-owned arrays; a solver that re-inspects whenever its
matrix changes does so thousands of times.  After a warm-up (the stream-ordered pool keeps what it has seen) the free
device memory must stay flat over further create / use / destroy cycles of every plan and state kind.
"""
import gc

import numpy as np
import pytest
import scipy.sparse as sps
import torch

import gpu_util as G
import spblas_reference_amd as sp
from spblas_reference_amd import _capi, generate

pytestmark = pytest.mark.gpu


def _cycle(dev):
    values, rowptr, colind, shape, nnz = dev["a"]
    a = sp.csr_view(values, rowptr, colind, shape, nnz)
    x, y, B, C = dev["x"], dev["y"], dev["B"], dev["C"]
    for alg in (_capi.SPMV_SLICED, _capi.SPMV_ROWBLOCK, _capi.SPMV_AUTO):
        info = sp.multiply_inspect(a, x, y, alg=alg)
        sp.multiply(info, a, x, y)
        del info
    info = sp.multiply_inspect(a, B, C)
    sp.multiply(info, a, B, C)
    del info
    opt = sp.matrix_opt(a)
    sp.multiply(opt, x, y)
    sp.multiply(sp.transposed(a), y, dev["xt"])  # CSC path (materialised transpose or atomics)
    del opt
    bv, br, bc, bsh, bnnz = dev["b"]
    b = sp.csr_view(bv, br, bc, bsh, bnnz)
    c_rp = dev["c_rp"]
    c = sp.csr_view(None, c_rp, None, (shape[0], bsh[1]), 0)
    cinfo = sp.multiply_compute(a, b, c)
    cn = cinfo.result_nnz()
    c.update(dev["c_val"][:cn], c_rp, dev["c_col"][:cn], (shape[0], bsh[1]), cn)
    for _ in range(3):
        sp.multiply_fill(cinfo, a, b, c)
    del cinfo, c
    tv, trp, tci, tsh, tnnz = dev["t"]
    t = sp.csr_view(tv, trp, tci, tsh, tnnz)
    tinfo = sp.triangular_solve_inspect(t, sp.lower_triangle, sp.explicit_diagonal, dev["tb"], dev["tx"])
    sp.triangular_solve(tinfo, t, sp.lower_triangle, sp.explicit_diagonal, dev["tb"], dev["tx"])
    del tinfo
    torch.cuda.synchronize()
    gc.collect()


def test_create_use_destroy_cycles_do_not_leak_device_memory(gpu):
    m, n = 40000, 30000
    values, rowptr, colind, shape, nnz = generate.generate_csr(m, n, 600000, seed=61)
    bv, br, bc, bsh, bnnz = generate.generate_csr(n, 20000, 200000, seed=62)
    rng = np.random.default_rng(1)
    tn = 20000
    S = sps.tril(sps.random(tn, tn, density=0.0005, format="csr", random_state=rng), -1)
    T = (S + sps.diags(np.asarray(abs(S).sum(axis=1)).ravel() + 1.0)).tocsr()
    dev = {
        "a": (G.dev(values), G.dev(rowptr), G.dev(colind), shape, nnz),
        "b": (G.dev(bv), G.dev(br), G.dev(bc), bsh, bnnz),
        "t": (G.dev(T.data.astype(np.float32)), G.dev(T.indptr.astype(np.int32)), G.dev(T.indices.astype(np.int32)),
              T.shape, T.nnz),
        "x": torch.rand(n, device="cuda"), "y": torch.zeros(m, device="cuda"), "xt": torch.zeros(n, device="cuda"),
        "B": torch.rand((n, 16), device="cuda"), "C": torch.zeros((m, 16), device="cuda"),
        "c_rp": torch.zeros(m + 1, dtype=torch.int32, device="cuda"),
        "c_val": torch.zeros(8_000_000, device="cuda"), "c_col": torch.zeros(8_000_000, dtype=torch.int32, device="cuda"),
        "tb": torch.rand(tn, device="cuda"), "tx": torch.zeros(tn, device="cuda"),
    }
    for _ in range(4):  # warm-up: pools and caches reach their size
        _cycle(dev)
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(25):
        _cycle(dev)
    free1, _ = torch.cuda.mem_get_info()
    lost = free0 - free1
    # one cycle allocates ~200 MB of plans and states; a leak of any one of them shows up as hundreds of MB over 25 cycles
    assert lost < 32 * 2 ** 20, f"free device memory fell by {lost / 2**20:.1f} MiB over 25 cycles"
