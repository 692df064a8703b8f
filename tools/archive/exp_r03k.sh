#!/bin/bash
# cfg4 inspect with and without the heaviest-first bin order of the scatter
cat > /tmp/insp_rmat.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import spblas_reference_amd as sp
from spblas_reference_amd import generate, _capi
v, rp, ci, shape, nnz = generate.rmat_csr_device(24, 16, dtype=torch.float64, seed=0, device="cuda")
a = sp.csr_view(v, rp, ci, shape, nnz)
x = torch.rand(shape[1], dtype=torch.float64, device="cuda"); y = torch.empty(shape[0], dtype=torch.float64, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    info = sp.multiply_inspect(a, x, y, alg=_capi.SPMV_SLICED)
    torch.cuda.synchronize(); print(f"sliced build #{rep}: {(time.perf_counter()-t0)*1e3:.2f} ms", flush=True)
    del info
PY
python /tmp/insp_rmat.py 2>/dev/null
SPBLAS_GFX950_PB_LPT=0 python /tmp/insp_rmat.py 2>/dev/null
