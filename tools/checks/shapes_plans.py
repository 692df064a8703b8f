"""Which plan a forced / automatic inspect ends up with on the shapes of tests/test_gpu_shapes.py (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np, torch
import gpu_util as G
import spblas_reference_amd as sp
from spblas_reference_amd import _capi
import test_gpu_shapes as T
for case in T.CASES:
    shape, rowptr, colind, values = T.make(case)
    values = values.astype(np.float32)
    a = sp.csr_view(G.dev(values), G.dev(rowptr), G.dev(colind), shape, int(rowptr[-1]))
    x = torch.rand(shape[1], device="cuda"); y = torch.zeros(shape[0], device="cuda")
    out = []
    for name in ("auto", "sliced", "rowblock"):
        info = sp.multiply_inspect(a, x, y, alg=T.ALGS[name])
        i = info.state_.info()
        out.append(f"{name}->alg {i['alg']} slices {i.get('n_slices')}")
    print(f"{case:30s}", " | ".join(out))
