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


def test_plan_made_in_one_thread_refreshes_on_the_calling_threads_stream(gpu):
    """A SLICED plan (matrix_opt: it multiplies with a snapshot of the values) created in thread A and used in thread B on
    B's own stream after the values changed in place: the refresh of the snapshot and the SpGEMM fill of a state created
    in A must be issued through B's handle -- B's stream, in order with B's multiply (round-3 advice: they used the
    creator's handle, i.e. whatever stream A last bound)."""
    m, n, nnz = 30000, 40000, 500000
    values, rowptr, colind, shape, _ = generate.generate_csr(m, n, nnz, seed=77)
    x_h = np.random.default_rng(5).standard_normal(n).astype(np.float32)
    made, errors, used = {}, [], {}

    def maker():
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                a = sp.csr_view(G.dev(values), G.dev(rowptr), G.dev(colind), shape, nnz)
                x = G.dev(x_h)
                y = torch.empty(m, device="cuda")
                info = sp.multiply_inspect(sp.matrix_opt(a), x, y, alg=_capi.SPMV_SLICED)
                sp.multiply(info, a, x, y)
                torch.cuda.current_stream().synchronize()
                made.update(a=a, x=x, info=info, handle=sp.api._Handle.current(x.device).h.value)
        except BaseException as e:  # noqa: BLE001
            errors.append(("maker", repr(e)))

    def user():
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                a, x, info = made["a"], made["x"], made["info"]
                used["handle"] = sp.api._Handle.current(x.device).h.value
                y = torch.full((m,), float("nan"), device="cuda")
                for k in (2.0, -0.5):
                    a.values().mul_(k)  # in place, on THIS thread's stream, right before the multiply
                    sp.multiply(info, a, x, y)
                    used[k] = y.clone()
                stream.synchronize()
        except BaseException as e:  # noqa: BLE001
            errors.append(("user", repr(e)))

    for fn in (maker, user):
        th = threading.Thread(target=fn)
        th.start()
        th.join(timeout=300)
    assert not errors, errors
    assert used["handle"] != made["handle"]
    lens = np.diff(rowptr)
    scale = 1.0
    for k in (2.0, -0.5):
        scale *= k
        v = (values * np.float32(scale)).astype(np.float32)
        util.assert_parity(G.host(used[k]), oracle.spmv(shape, rowptr, colind, v, x_h),
                           oracle.spmv_absrow(rowptr, colind, v, x_h), np.float32, row_len=lens,
                           what=f"cross-thread plan after in-place scaling by {k}")


@pytest.mark.gpu
@pytest.mark.parametrize("preload", ["1", "0"])
def test_first_handle_loads_the_code_objects_or_leaves_it_to_the_first_use(gpu, preload):
    """spblas_gfx950_create: the first handle of a process loads the library's code objects (SPBLAS_GFX950_PRELOAD=0:
    the runtime loads each at the first launch, as before).  Either way a fresh process computes the same product; with
    the preload the first inspect no longer pays for the loading."""
    import os
    import subprocess
    import sys
    code = (
        "import time, numpy as np, torch, spblas_reference_amd as sp\n"
        "from spblas_reference_amd.api import _Handle\n"
        "dev = torch.device('cuda:0')\n"
        "torch.cuda.synchronize(); t0 = time.perf_counter(); _Handle.current(dev); t_handle = time.perf_counter() - t0\n"
        "rng = np.random.default_rng(5); m, n = 3000, 2000\n"
        "lens = rng.integers(0, 12, m); rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32); nnz = int(rp[-1])\n"
        "ci = rng.integers(0, n, nnz).astype(np.int32); v = rng.random(nnz).astype(np.float32); x = rng.random(n).astype(np.float32)\n"
        "a = sp.csr_view(torch.from_numpy(v).to(dev), torch.from_numpy(rp).to(dev), torch.from_numpy(ci).to(dev), (m, n), nnz)\n"
        "xd = torch.from_numpy(x).to(dev); y = torch.empty(m, device=dev)\n"
        "torch.cuda.synchronize(); t0 = time.perf_counter(); info = sp.multiply_inspect(a, xd, y); torch.cuda.synchronize()\n"
        "t_inspect = time.perf_counter() - t0\n"
        "sp.multiply(info, a, xd, y)\n"
        "ref = np.zeros(m, dtype=np.float64); np.add.at(ref, np.repeat(np.arange(m), lens), v.astype(np.float64) * x[ci])\n"
        "assert np.allclose(y.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)\n"
        "print('TIMES', t_handle * 1e3, t_inspect * 1e3)\n")
    env = dict(os.environ, SPBLAS_GFX950_PRELOAD=preload)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    t_handle, t_inspect = (float(t) for t in r.stdout.split("TIMES")[1].split())
    if preload == "0":
        assert t_handle < 5.0, f"no preload asked for, yet the handle took {t_handle:.1f} ms"
