"""-m gpu: several host threads, each on its own stream, calling the backend at once.

The reference's vendor back ends keep their library handle inside each operation state
(/root/reference/include/spblas/vendor/rocsparse/detail/operation_state_t.hpp), so two threads that work on different
operands never share one.  The Python layer keeps one handle per (thread, device) for the same reason: with a shared
handle a launch can land on the other thread's stream and read a right-hand side that is still being written.  The
numerical part of the test is a stress run (the window is narrow: `tools/checks/shared_handle_check.py` re-creates the
shared table and did not hit it in a handful of runs); the handle identity check at the end is the deterministic part.
"""
import threading

import numpy as np
import pytest
import torch

import gpu_util as G
import spblas_reference_amd as sp
import util
from oracle import oracle
from spblas_reference_amd import _capi, generate

pytestmark = pytest.mark.gpu


def test_threads_on_their_own_streams_do_not_share_a_handle(gpu):
    n_threads, rounds = 4, 12
    problems = []
    for t in range(n_threads):
        m, n, nnz = 20000 + 3000 * t, 30000 + 1000 * t, 400000 + 50000 * t
        values, rowptr, colind, shape, _ = generate.generate_csr(m, n, nnz, seed=40 + t)
        xs = [np.random.default_rng(100 * t + r).standard_normal(n).astype(np.float32) for r in range(rounds)]
        refs = [(oracle.spmv(shape, rowptr, colind, values, x), oracle.spmv_absrow(rowptr, colind, values, x)) for x in xs]
        problems.append((values, rowptr, colind, shape, nnz, xs, refs))
    errors, handles = [], {}
    start = threading.Barrier(n_threads)

    def worker(t):
        try:
            values, rowptr, colind, shape, nnz, xs, refs = problems[t]
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                a = sp.csr_view(G.dev(values), G.dev(rowptr), G.dev(colind), shape, nnz)
                x = torch.zeros(shape[1], device="cuda")
                y = torch.full((shape[0],), float("nan"), device="cuda")
                alg = (_capi.SPMV_SLICED, _capi.SPMV_ROWBLOCK, _capi.SPMV_VECTOR, None)[t % 4]
                info = sp.multiply_inspect(a, x, y, alg=alg) if alg is not None else None
                handles[t] = sp.api._Handle.current(x.device).h.value
                staged = [G.dev(v) for v in xs]
                stream.synchronize()
                start.wait()
                outs = []
                for r in range(rounds):
                    # the copy and the multiply are ordered by THIS thread's stream only: a launch that strays onto
                    # another stream reads x while the copy is in flight
                    x.copy_(staged[r], non_blocking=True)
                    if info is not None:
                        sp.multiply(info, a, x, y)
                    else:
                        sp.multiply(a, x, y)
                    outs.append(y.clone())
                stream.synchronize()
            lens = np.diff(rowptr)
            for r in range(rounds):
                util.assert_parity(G.host(outs[r]), refs[r][0], refs[r][1], np.float32, row_len=lens,
                                   what=f"thread {t} round {r}")
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(e)))
            try:
                start.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors
    assert len(set(handles.values())) == n_threads, f"threads shared a handle: {handles}"
