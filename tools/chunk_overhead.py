"""One-GPU cost of the N = 8 dependent step with chunk flags (round 4), on the REAL shard shape: 1.25 M rows x 10 M columns
of cfg2 (508 wave-bins, K-split reduce, one expand item per x slice).  One process, straight through the C ABI: the "peer"
table holds this process's own y (10 M entries, of which the step rewrites the first 1.25 M), x of the chain is that y, the
wait descriptor names one rank -- its own, for which there is nothing to wait -- so what is measured is the kernels, the
launches and the flag traffic of one rank's step, not the links.  Prints: expand + reduce_rows_bcast + step_signal +
step_wait (the barrier step of round 3), the chunked step whose combine kernel publishes the chunks, and the chunked step
that cuts the reduce into stripes."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spblas_reference_amd as sp  # noqa: E402
from spblas_reference_amd import _capi, api, generate  # noqa: E402

dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
n = 10_000_000
v, rp, ci, shape, nnz = generate.uniform_csr_device(rows, n, 10, seed=0, device=dev)
a = sp.csr_view(v, rp, ci, shape, nnz)
y = torch.rand(n, device=dev)
info = sp.multiply_inspect(sp.matrix_opt(a), y, y[:rows], alg=_capi.SPMV_SLICED)
plan = info.state_.plan
si, pi = info.state_.sliced_info(), info.state_.info()
lib = _capi.lib()
hd = api._Handle.current(dev)
ytab = torch.tensor([y.data_ptr()], dtype=torch.int64, device=dev)
flags = torch.zeros(64, dtype=torch.int64, device=dev)
ftab = torch.tensor([flags.data_ptr()], dtype=torch.int64, device=dev)
status = torch.zeros(2, dtype=torch.int32, device=dev)
alpha = ctypes.c_float(0.4)
vp = ctypes.c_void_p


def timed(fn, k=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6


step = [0]


def barrier_step():
    step[0] += 1
    api.check(lib.spblas_gfx950_spmv_step_bcast(hd.h, plan, ctypes.byref(alpha), vp(y.data_ptr()), vp(ytab.data_ptr()), 1, 0, 1), "step")
    api.check(lib.spblas_gfx950_step_signal(hd.h, vp(ftab.data_ptr()), 1, 0, step[0]), "signal")
    api.check(lib.spblas_gfx950_step_wait(hd.h, vp(flags.data_ptr()), 1, step[0], 5000, vp(status.data_ptr())), "wait")


print(f"shard {rows} x {n}: bins {si['n_bins']}, ksplit {si['ksplit']}, slices {pi['n_slices']}, expand items {pi['expand_items']}")
print(f"barrier step (expand + reduce + combine + signal + wait): {timed(barrier_step):7.1f} us")
for mode in ("combine", "stripes"):
    os.environ["SPBLAS_GFX950_CHUNK_STRIPES"] = "1" if mode == "stripes" else "0"
    for chunks in (2, 4, 8):
        crows = (ctypes.c_int64 * (chunks + 1))()
        api.check(lib.spblas_gfx950_spmv_chunk_rows(plan, chunks, crows), "chunk_rows")
        ctab = torch.tensor(list(crows), dtype=torch.int64, device=dev)
        flags.zero_()

        def chunk_step():
            step[0] += 1
            w = _capi.chunk_wait(flags.data_ptr(), ctab.data_ptr(), 1, chunks, step[0] - 1, 5000, status.data_ptr(), 0)
            api.check(lib.spblas_gfx950_spmv_step_bcast_chunked(hd.h, plan, ctypes.byref(alpha), vp(y.data_ptr()),
                                                                vp(ytab.data_ptr()), 1, 0, chunks, vp(ftab.data_ptr()), 0,
                                                                step[0], ctypes.byref(w)), "chunked")

        def chunk_step_nowait():  # (the combine publishes, the expand loads x as usual: prices the two halves apart)
            step[0] += 1
            api.check(lib.spblas_gfx950_spmv_step_bcast_chunked(hd.h, plan, ctypes.byref(alpha), vp(y.data_ptr()),
                                                                vp(ytab.data_ptr()), 1, 0, chunks, vp(ftab.data_ptr()), 0,
                                                                step[0], None), "chunked")

        t = timed(chunk_step)
        t_nw = timed(chunk_step_nowait)
        torch.cuda.synchronize()
        assert int(status[0]) == 0
        print(f"   ... with a plain expand (no waits): {t_nw:7.1f} us")
        print(f"chunked step, {mode:8s} chunks={chunks} (rows {list(crows)[:3]}...): {t:7.1f} us")
