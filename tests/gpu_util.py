"""Helpers for the -m gpu parity tests: move host CSR data to HBM and wrap it in views."""
import numpy as np
import torch

import spblas_reference_amd as sp


def dev(a, device="cuda"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def csr_on_device(values, rowptr, colind, shape, nnz, offset64=False):
    rp = rowptr.astype(np.int64) if offset64 else rowptr.astype(np.int32)
    return sp.csr_view(dev(values), dev(rp), dev(colind.astype(np.int32)), shape, nnz)


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()
