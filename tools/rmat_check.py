"""R-MAT matrices of several scales / types / offset widths through AUTO (matrix_opt) and the forced plans, every
result compared with the CPU oracle (full matrix).  tools/rmat_check.py [scales...]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import spblas_reference_amd as sp
from spblas_reference_amd import generate, _capi
from oracle import oracle
import util
scales = [int(a) for a in sys.argv[1:]] or [16, 19, 21]
bad = 0
for scale in scales:
    for dtype in (torch.float32, torch.float64):
        for off64 in (False, True):
            v, rp, ci, shape, nnz = generate.rmat_csr_device(scale, 16, dtype=dtype, device="cuda", seed=scale)
            v = v - 0.5
            rp_d = rp.long() if off64 else rp
            m, n = shape
            x = torch.rand(n, dtype=dtype, device="cuda") - 0.5
            a = sp.csr_view(v, rp_d, ci, shape, nnz)
            npd = np.float32 if dtype == torch.float32 else np.float64
            vh, rh, ch, xh = v.cpu().numpy(), rp.cpu().numpy(), ci.cpu().numpy(), x.cpu().numpy()
            ref = oracle.spmv(shape, rh, ch, vh, xh)
            absrow = oracle.spmv_absrow(rh, ch, vh, xh)
            lens = np.diff(rh)
            for name, alg, opt in (("auto", _capi.SPMV_AUTO, True), ("sliced", _capi.SPMV_SLICED, False),
                                   ("rowblock", _capi.SPMV_ROWBLOCK, False)):
                y = torch.full((m,), float("nan"), dtype=dtype, device="cuda")
                try:
                    info = sp.multiply_inspect(sp.matrix_opt(a) if opt else a, x, y, alg=alg)
                except Exception as e:  # noqa: BLE001
                    print("SKIP", scale, dtype, off64, name, str(e)[:60]); continue
                sp.multiply(info, a, x, y); sp.multiply(info, a, x, y)
                torch.cuda.synchronize()
                pi = info.state_.info()
                si = info.state_.sliced_info()
                try:
                    util.assert_parity(y.cpu().numpy(), ref, absrow, npd, row_len=lens, what=f"rmat {scale} {name}")
                    print(f"ok   scale {scale} {str(dtype)[6:]} off64={off64} {name}: alg {pi['alg']} bins {si.get('n_bins')} var {si.get('variable_bins')} hub {si.get('hub_rows')} trial {si.get('auto_trial')} rb {si.get('trial_rowblock_ns')/1e3:.0f} us sl {si.get('trial_sliced_ns')/1e3:.0f} us")
                except AssertionError as e:
                    bad += 1; print("FAIL", scale, dtype, off64, name, str(e)[:200])
                del info
print("failures:", bad)
sys.exit(1 if bad else 0)
