#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: CSR SpMV GFLOP/s + achieved HBM GB/s (% of roofline).

  python bench.py [--gpus N --steps K --warmup W] [--workload spmv|spmm|spgemm|spmv_rmat]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default (N=1): cfg2 = fp32 CSR SpMV, 10M x 10M, exactly 10 nnz/row, columns iid uniform
(unsorted), values/x U[0,1), int32 indices, inputs resident in HBM before the timed region.
A "step" = one multiply(info, A, x, y) (the inspect phase runs once, outside the timed
region, and is reported separately).  N>1: the SAME global matrix row-sharded over N
ranks (strong scaling), step = local SpMV + ONE all-gather of y (in the timed region); the JSON's
"multi_gpu" object says where the step time goes (local kernels / gather / RCCL step / fused step).
--workload spmv_rmat is BASELINE cfg4 on the same harness: fp64 R-MAT scale 24, rows sharded by nnz prefix.
One JSON line on rank 0.  roofline.achieved = algorithmic bytes per launch (SURVEY.md
section 8d: nnz*(sizeof(T)+4) + (m+1)*4 + n*sizeof(T) + m*sizeof(T)) / average kernel
duration from HIP events recorded on the launch stream around the timed steps.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
CHUNK_ROWS = 250_000   # the global matrix is generated per fixed row chunk => identical for every N


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--workload", default="spmv", choices=["spmv", "spmv_poisson", "spmv_poisson1", "spmv_rmat", "spmv_rmat1", "spmv_plain", "spmm", "spmm_banded", "spmm_rmat", "spgemm", "add", "transpose", "sptrsv", "csc_spmv", "spgemm4", "spmv_rmat_shards"])
    p.add_argument("--rows", type=int, default=None, help="override the row count (debug only; reported)")
    p.add_argument("--cols", type=int, default=None, help="override the column count (debug: emulate one row shard)")
    p.add_argument("--alg", default="auto", choices=["auto", "vector", "rowblock", "sliced", "noplan"])
    p.add_argument("--chunks", type=int, default=0,
                   help="N>1: stripes per step whose all-gathers overlap the next stripe's compute (0 = auto)")
    p.add_argument("--flag-chunks", type=int, default=4,
                   help="N>1, fused path on a square matrix: chunks of the dependent chain without a step barrier (diagnostic "
                        "`fused_chunked_step_ms`; 0 = off)")
    p.add_argument("--fused", default="auto", choices=["auto", "off"],
                   help="N>1: auto = peer stores of y from the reduce kernels (hipIpc) when every rank can, "
                        "validated against the RCCL all-gather path; off = RCCL all-gather only")
    p.add_argument("--debug-multi", action="store_true",
                   help="debug: run the N>1 code path (process group, sharded operators, fused all-gather) with "
                        "a single rank; launch through torch.distributed.run --nproc-per-node 1")
    p.add_argument("--overlap", action="store_true", help="debug: use the N>1 overlapped step at N=1 (no collective)")
    p.add_argument("--debug-one-gpu", action="store_true",
                   help="debug/tests: all N ranks share cuda:0 with gloo as the process-group backend (RCCL refuses two "
                        "ranks on one device): exercises the N>1 code path end to end on a one-GPU box")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-secondary", action="store_true",
                   help="default N=1 cfg2 run only: skip the cfg4 / cfg3 / cfg5 / 8(f) records of the `secondary` object")
    p.add_argument("--full-line", action="store_true",
                   help="print the complete result object instead of the compact headline (tools/*.sh that read the plan)")
    p.add_argument("--no-8f", dest="no_8f", action="store_true",
                   help="`secondary` without the SURVEY 8(f) records (add, transpose, triangular solve)")
    return p.parse_args()


def launch_ranks(nproc):
    """One rank per GPU under torch.distributed.run on a free loopback port; stdout/stderr pass straight through."""
    import socket
    import subprocess
    # (device_count() may call hipGetDeviceCount on ROCm builds without amdsmi; that is harmless here because this
    # parent only ever STARTS a child process -- it never execs over itself and never launches GPU work)
    if "--debug-one-gpu" not in sys.argv and torch.cuda.device_count() < nproc:
        print(f"bench.py: --gpus {nproc} but {torch.cuda.device_count()} device(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL and hipIpc across processes need it here
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def gen_rows(row_begin, row_end, n, per_row, poisson, dtype, device):
    """Rows [row_begin,row_end) of the global cfg matrix, generated chunk by chunk with
    seed = chunk index so that every world size sees the same matrix."""
    from spblas_reference_amd import generate
    vals, cols, lens = [], [], []
    c0 = row_begin // CHUNK_ROWS
    c1 = (row_end + CHUNK_ROWS - 1) // CHUNK_ROWS
    for c in range(c0, c1):
        lo, hi = c * CHUNK_ROWS, (c + 1) * CHUNK_ROWS
        v, rp, ci, _, _ = generate.uniform_csr_device(hi - lo, n, per_row, dtype=dtype, seed=1000 + c, device=device,
                                                      poisson=poisson, offset_dtype=torch.int64)
        a, b = max(row_begin, lo) - lo, min(row_end, hi) - lo
        p0, p1 = int(rp[a]), int(rp[b])
        vals.append(v[p0:p1])
        cols.append(ci[p0:p1])
        lens.append(rp[a + 1:b + 1] - rp[a:b])
    lens = torch.cat(lens)
    rowptr = torch.zeros(row_end - row_begin + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens, 0, out=rowptr[1:])
    return torch.cat(vals), rowptr.to(torch.int32), torch.cat(cols), int(rowptr[-1])


def spmv_bytes(m, n, nnz, tsize):
    return nnz * (tsize + 4) + (m + 1) * 4 + n * tsize + m * tsize


from bench_extra import read_pmc_traffic  # noqa: E402  (committed PMC constants + staleness stamp)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_spmv(values, rowptr, colind, shape, x, nnz):
    """Reference CPU path restated (oracle, kind 'port'): 1 core, reference flags
    (-O3 -march=native, built on THIS host), best of 3 on the full workload."""
    from oracle import oracle
    v, rp, ci, xh = values.cpu().numpy(), rowptr.cpu().numpy(), colind.cpu().numpy(), x.cpu().numpy()
    try:
        oracle.load(native=True)
        native = True
    except Exception:
        native = False
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        y_ref = oracle.spmv(shape, rp, ci, v, xh, native=native)
        best = min(best, time.perf_counter() - t0)
    best_omp = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        oracle.spmv_omp(rp, ci, v, xh, native=native)
        best_omp = min(best_omp, time.perf_counter() - t0)
    ncpu = os.cpu_count()
    absrow = oracle.spmv_absrow(rp, ci, v, xh)
    return y_ref, absrow, {"value": 2.0 * nnz / best / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"full workload ({nnz} nnz), best of 3 runs of oracle_spmv (-O3 -march={'native' if native else 'x86-64-v3'})",
            "seconds": best,
            # (scalar fields: a parser that keeps only the flat keys of cpu_baseline keeps the all-core figure too)
            "all_cores_value": 2.0 * nnz / best_omp / 1e9, "all_cores": ncpu, "all_cores_seconds": best_omp,
            "all_cores_note": "OpenMP static row-parallel variant of the same loop, every host core"}


def parity_spmv(y, y_ref, absrow, tol, row_len=None):
    """Norm-wise parity of SURVEY.md section 8c (tests/util.py:assert_parity): |y - y_ref| <= tol * sum_p |a_p x_p| per row;
    like there, never tighter than the rounding the reference's own sequential sum of a k-entry row carries (k/2 * eps:
    only the hub rows of the R-MAT graph are long enough for that to matter).  numpy arrays in, a small report out."""
    err = np.abs(y.astype(np.float64) - y_ref.astype(np.float64))
    tol_row = np.full(err.shape, tol)
    if row_len is not None:
        tol_row = np.maximum(tol_row, 0.5 * np.asarray(row_len, dtype=np.float64) * float(np.finfo(y.dtype).eps))
    bound = tol_row * absrow.astype(np.float64) + float(np.finfo(y.dtype).tiny)
    bad = ~(err <= bound)  # NaN must fail
    ratio = float((err / np.maximum(absrow.astype(np.float64), 1e-300)).max()) if err.size else 0.0
    return {"status": "pass" if not bad.any() else "fail", "rows": int(err.size), "rows_out_of_bound": int(bad.sum()),
            "tol": tol, "worst_err_over_rownorm": ratio}


def measure(step, steps, warmup, multi, device):
    """W untimed steps, then EXACTLY K steps between barrier + synchronize on both sides; ONE pair of HIP events on
    the launch stream brackets the region (per-step event records were measured to cost up to 25 us per step).
    Returns (wall seconds: max over ranks, average ms per step between the events)."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    t0 = time.perf_counter()
    region[0].record()
    for _ in range(steps):
        step()
    region[1].record()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if multi:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return float(el.item()), region[0].elapsed_time(region[1]) / steps


def measure_guarded(op, step, steps, warmup, device):
    """measure() for a path that may fail on SOME ranks (the fused exchange on its first contact with N devices): every
    rank issues the same collectives whatever happens locally -- a short probe with a short time-out and an agreement, then
    the timed loop and a second agreement.  Returns (valid on every rank, wall seconds: max over ranks, ms per step between
    the events, the local exception or None).  Device-side waits are bounded (op._timeout), so a rank whose peers never
    publish comes back with a time-out status instead of hanging."""
    failed = None

    def run(n):
        nonlocal failed
        if failed is None:
            try:
                for _ in range(n):
                    step()
                torch.cuda.synchronize()
                op.check_status()
            except Exception as e:  # noqa: BLE001 - reported, the RCCL number stands
                failed = e

    def agree():
        flag = torch.tensor([0 if failed is not None else 1], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    saved, op._timeout = op._timeout, 2000
    run(2)
    if not agree():
        op._timeout = saved
        return False, None, None, failed
    op._timeout = min(saved, 5000)  # (no step takes seconds: a flag that stops arriving must not cost the run)
    run(warmup)
    dist.barrier()
    torch.cuda.synchronize()
    region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    t0 = time.perf_counter()
    region[0].record()
    run(steps)
    region[1].record()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ok = agree()
    op._timeout = saved
    return ok, float(el.item()), region[0].elapsed_time(region[1]) / steps, failed


def build_cfg2(args, world, rank, device, sharded, sp):
    """cfg2: the SAME global 10M x 10M matrix for every N (generated per fixed row chunk), equal row shards."""
    poisson = args.workload == "spmv_poisson"
    m = n = args.rows or 10_000_000
    if args.cols:
        n = args.cols
    dtype = torch.float32
    chunks = args.chunks if args.chunks > 0 else 1
    ranges = sharded.striped_row_ranges(m, world, chunks)
    if ranges is None or poisson and world > 1:
        chunks = 1
        ranges = sharded.striped_row_ranges(m, world, 1)
    if ranges is None:
        sys.exit(f"rows={m} is not divisible by the number of ranks")
    a_chunks, nnz_local = [], 0
    for c in range(chunks):
        lo, hi = ranges[c][rank]
        v_c, rp_c, ci_c, nnz_c = gen_rows(lo, hi, n, 10, poisson, dtype, device)
        a_chunks.append(sp.csr_view(v_c, rp_c, ci_c, (hi - lo, n), nnz_c))
        nnz_local += nnz_c
    label = (f"cfg2: fp32 CSR SpMV {m}x{n}, {'Poisson(10)' if poisson else 'exactly 10'} nnz/row, "
             f"uniform random unsorted columns, int32 indices")
    return {"m": m, "n": n, "dtype": dtype, "tsize": 4, "chunks": chunks, "ranges": ranges, "a_chunks": a_chunks,
            "nnz_local": nnz_local, "label": label, "dtype_name": "f32", "bounds": [ranges[0][r][0] for r in range(world)] + [m],
            "pmc_key": "spmv_cfg2" if (world == 1 and not poisson and args.rows is None and args.cols is None) else None}


def build_rmat(args, world, rank, device, sharded, sp):
    """cfg4: fp64 R-MAT scale 24 (edge factor 16, duplicates kept, 268 M entries), rows sharded by NNZ PREFIX on rowptr
    (the shards differ several-fold in rows).  Every rank generates the same graph from the same seed on its own
    GPU, keeps its row range and drops the rest; a checksum all-reduce confirms the ranks agree."""
    from spblas_reference_amd import generate
    scale = 24 if args.rows is None else int(np.log2(args.rows))
    values, rowptr, colind, shape, nnz = generate.rmat_csr_device(scale, 16, dtype=torch.float64, seed=0, device=device)
    m = n = shape[0]
    if world > 1:
        chk = torch.stack([rowptr.long().sum(), colind.long().sum()]).double()
        lo_hi = torch.stack([chk, -chk])
        dist.all_reduce(lo_hi, op=dist.ReduceOp.MAX)
        if not bool((lo_hi[0] == -lo_hi[1]).all()):
            sys.exit("ranks generated different R-MAT graphs")
    bounds = sharded.partition_rows_by_nnz(rowptr, world)
    a_local = sharded.shard_csr(values, rowptr, colind, shape, bounds[rank], bounds[rank + 1]) if world > 1 else \
        sp.csr_view(values, rowptr, colind, shape, nnz)
    del values, rowptr, colind
    torch.cuda.empty_cache()
    label = f"cfg4: fp64 CSR SpMV, R-MAT scale {scale}, edge factor 16, duplicates kept, rows sharded by nnz prefix"
    return {"m": m, "n": n, "dtype": torch.float64, "tsize": 8, "chunks": 1, "ranges": None, "a_chunks": [a_local],
            "nnz_local": a_local.size(), "label": label, "dtype_name": "f64", "bounds": bounds,
            "pmc_key": "spmv_rmat" if world == 1 else None}


def main():
    args = parse()
    exit_code = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.debug_multi):
        # Plain `python bench.py --gpus N`: start the N ranks ourselves as a CHILD torch.distributed.run -- this
        # parent has not touched the GPU (no HIP call, no torch.cuda query so far) and never does; it relays the
        # ranks' output (rank 0 prints the ONE JSON line) and exits with the child's code.
        return launch_ranks(max(1, args.gpus))
    if args.gpus != world:
        args.gpus = world
    # stdout carries exactly ONE line (the JSON, rank 0): libraries that print banners to fd 1 (RCCL announces its
    # version there when the first communicator is created) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.debug_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    multi = world > 1 or args.debug_multi
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.debug_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    import spblas_reference_amd as sp
    from spblas_reference_amd import _capi, sharded
    sp._capi.lib()  # fail loudly if the HIP library is missing
    # the first handle of the process loads the library's code objects (csrc/handle.hip): a one-time cost that used to fall
    # on the first inspect / compute call -- reported as config.handle_create_ms next to the first-call figures
    torch.cuda.synchronize()
    t_h = time.perf_counter()
    from spblas_reference_amd.api import _Handle
    _Handle.current(device)
    args.handle_create_ms = (time.perf_counter() - t_h) * 1e3

    if args.workload in ("spmv_rmat1", "spmv_plain", "spmv_poisson1", "spmm", "spmm_banded", "spmm_rmat", "spgemm", "add", "transpose", "sptrsv", "csc_spmv", "spgemm4", "spmv_rmat_shards"):
        from bench_extra import run_extra  # secondary configs (cfg3 / cfg5 / 8f rows), 1 GPU
        sys.stdout.flush()
        os.dup2(json_fd, 1)  # single-GPU secondary workloads print their own line
        return run_extra(args, device)

    rmat = args.workload == "spmv_rmat"
    prob = (build_rmat if rmat else build_cfg2)(args, world, rank, device, sharded, sp)
    m, n, dtype, tsize, chunks = prob["m"], prob["n"], prob["dtype"], prob["tsize"], prob["chunks"]
    a_chunks, ranges, bounds, nnz_local = prob["a_chunks"], prob["ranges"], prob["bounds"], prob["nnz_local"]
    rows_local = sum(a.shape()[0] for a in a_chunks)
    algs = {"auto": _capi.SPMV_AUTO, "vector": _capi.SPMV_VECTOR, "rowblock": _capi.SPMV_ROWBLOCK,
            "sliced": _capi.SPMV_SLICED}
    g = torch.Generator(device=device).manual_seed(7)
    x = torch.rand(n, dtype=dtype, device=device, generator=g)  # same on every rank (replicated)
    nnz_t = torch.tensor([nnz_local], dtype=torch.int64, device=device)
    if multi:
        dist.all_reduce(nnz_t)
    nnz = int(nnz_t.item())

    # N = 1: one plan, no collective.  N > 1: local SpMV of the rank's rows + ONE all-gather of y per step
    # (RCCL; in place for equal shards, direct sends of the exact shard sizes for nnz-balanced ones), or -- when
    # every rank can set it up and it reproduces the RCCL path bit for bit on four different vectors -- the
    # all-gather fused into the reduce kernels (peer stores into hipIpc-mapped copies of y + device-side barrier).
    # --chunks C > 1 enables the striped variants whose all-gathers overlap the next stripe's kernels.
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mode, op, rccl_op, fused_op = "plain", None, None, None
    if not rmat and (multi or args.overlap) and chunks > 1 and args.alg in ("auto", "sliced"):
        try:
            lens = torch.cat([a.rowptr()[1:].long() - a.rowptr()[:-1].long() for a in a_chunks])
            rp = torch.zeros(rows_local + 1, dtype=torch.int64, device=device)
            torch.cumsum(lens, 0, out=rp[1:])
            a_cat = sp.csr_view(torch.cat([a.values() for a in a_chunks]), rp.to(torch.int32),
                                torch.cat([a.colind() for a in a_chunks]), (rows_local, n), nnz_local)
            op = sharded.OverlappedShardedSpMV(a_cat, ranges, alg=algs[args.alg])
            a_chunks = [a_cat]
            mode = "overlapped"
        except RuntimeError:
            op = None
        if multi:  # all ranks or none: a rank on the striped path and a rank on the plain one would not meet in a collective
            agree = torch.tensor([int(op is not None)], dtype=torch.int32, device=device)
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            if not int(agree.item()):
                op, mode, a_chunks = None, "plain", prob["a_chunks"]
    if op is None:
        if rmat:
            rccl_op = sharded.ShardedSpMV(a_chunks[0], bounds, inspect=args.alg != "noplan",
                                          alg=None if args.alg == "noplan" else algs[args.alg])
        else:
            rccl_op = sharded.PipelinedShardedSpMV(a_chunks, ranges, inspect=args.alg != "noplan",
                                                   alg=None if args.alg == "noplan" else algs[args.alg])
        op = rccl_op
        mode = "pipelined" if chunks > 1 else "plain"
    if multi and mode == "plain" and args.fused == "auto" and args.alg in ("auto", "sliced") and \
            len({bounds[r + 1] - bounds[r] for r in range(world)}) == 1:
        fused_op = sharded.try_fused(a_chunks[0], bounds, x, lambda xk: rccl_op.step(xk), alg=algs[args.alg],
                                     info=rccl_op.infos[0], chunks=args.flag_chunks if m == n else 0,
                                     shared_device=args.debug_one_gpu,
                                     log=(lambda msg: print(f"[bench] {msg}; using RCCL all-gather", file=sys.stderr))
                                     if rank == 0 else None)
        if fused_op is not None:
            op, mode = fused_op, "fused"
    torch.cuda.synchronize()
    inspect_ms = (time.perf_counter() - t0) * 1e3
    info0 = op.info if mode in ("overlapped", "fused") else op.infos[0]
    plan_info = info0.state_.info() if info0.state_ is not None else {"alg": "plan-free"}
    if info0.state_ is not None and hasattr(info0.state_, "sliced_info"):
        si = info0.state_.sliced_info()
        if plan_info.get("alg") == 3:
            plan_info["sliced"] = si
        elif si.get("auto_trial"):
            plan_info["auto_trial"] = {k: si[k] for k in ("trial_rowblock_ns", "trial_sliced_ns")}

    # First contact with N devices (nothing of the fused exchange has run across devices before a SCALE run): the RCCL
    # path is ALWAYS timed first and is the number that stands unless the fused path -- validated before (try_fused), timed
    # behind a collective-safe guard, checked again after its timed loop -- is valid AND faster.  A fused failure of any
    # kind (time-out, exception, mismatch) on any rank leaves the RCCL number as ms_per_step and exit code 0.
    fused_post_check, fused_elapsed, fused_kern_ms, fused_fail = None, None, None, None
    if multi and mode == "fused" and rccl_op is not None:
        rccl_elapsed, rccl_kern_ms = measure(lambda: rccl_op.step(x), args.steps, args.warmup, multi, device)
        ok_f, fused_elapsed, fused_kern_ms, fused_fail = measure_guarded(op, lambda: op.step(x), args.steps, args.warmup, device)
        if ok_f:
            # ... and once more AFTER hundreds of steps, with vectors it has not seen, so that a peer store that only goes
            # stale under load cannot publish a number (every rank runs the same collectives whatever happens locally)
            ok, y_rs = 1, [rccl_op.step(x_k).clone() for x_k in (1.25 * x + 0.5, 0.75 * x - 0.25)]
            try:
                for x_k, y_r in zip((1.25 * x + 0.5, 0.75 * x - 0.25), y_rs):
                    y_f = op.step(x_k)
                    torch.cuda.synchronize()
                    op.check_status()
                    ok &= int(torch.equal(y_f, y_r))
            except Exception as e:  # noqa: BLE001
                ok, fused_fail = 0, e
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            fused_post_check = bool(int(flag.item()))
        fused_valid = bool(ok_f and fused_post_check)
        if not fused_valid:
            if rank == 0:
                print(f"[bench] fused all-gather not used ({fused_fail or 'failed its check on some rank'}); the RCCL "
                      "all-gather path is the timed one", file=sys.stderr)
            try:
                fused_op.close()
            except Exception:  # noqa: BLE001
                pass
            fused_op, fused_elapsed, fused_kern_ms = None, None, None
            op, mode = rccl_op, "plain"
            elapsed, kern_avg_ms = rccl_elapsed, rccl_kern_ms
        elif fused_elapsed <= rccl_elapsed:
            elapsed, kern_avg_ms = fused_elapsed, fused_kern_ms
        else:  # valid but slower than the library collective on this machine: the faster valid path is the timed one
            op, mode = rccl_op, "plain"
            elapsed, kern_avg_ms = rccl_elapsed, rccl_kern_ms
    else:
        elapsed, kern_avg_ms = measure(lambda: op.step(x), args.steps, args.warmup, multi, device)
        rccl_elapsed = elapsed if (multi and rccl_op is not None and op is rccl_op) else None

    # Where the time of a multi-GPU step goes (outside the timed region, same K): the local kernels alone, the
    # all-gather alone, the RCCL step and -- when it could be set up -- the fused step.
    diag = None
    if multi and rccl_op is not None:
        k = max(5, min(args.steps, 50))
        local_s, local_ev = measure(lambda: rccl_op.local(x), k, 2, multi, device)
        gather_s, _ = measure(lambda: rccl_op.gather(), k, 2, multi, device)
        rccl_s = (rccl_elapsed / args.steps * k) if rccl_elapsed is not None else measure(lambda: rccl_op.step(x), k, 2, multi, device)[0]
        # Throughput form of the fused step (independent right-hand sides: the wait for step k-1 sits between the
        # expand and the reduce of step k, so link time and expand overlap).  A diagnostic next to the timed, dependent
        # form: never the metric value, and any failure here leaves the line as it is.
        pipe_ms, pipe_ok = None, None
        fop = fused_op  # (valid, whether or not it is the timed path)
        if fop is not None:
            # every collective below is executed by every rank whatever happened locally: a rank whose local part raised
            # (say a barrier time-out reported by check_status) only lowers the flag that is reduced at the end
            failed, y_p, same = None, None, 0

            def local(fn):
                nonlocal failed
                if failed is None:
                    try:
                        return fn()
                    except Exception as e:  # noqa: BLE001 - diagnostics only
                        failed = e
                return None

            def warm():
                for _ in range(2):
                    fop.step_pipelined(x)
                fop.flush()
                torch.cuda.synchronize()

            def timed():
                for _ in range(k):
                    fop.step_pipelined(x)
                y = fop.flush()
                torch.cuda.synchronize()
                return y

            def compare():
                fop.check_status()
                y_keep = y_p.clone()
                y_d = fop.step(x)
                torch.cuda.synchronize()
                return int(torch.equal(y_keep, y_d))

            local(warm)
            dist.barrier()
            t1 = time.perf_counter()
            y_p = local(timed)
            dist.barrier()
            el = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            same = local(compare) or 0
            okf = torch.tensor([0 if failed is not None else same], dtype=torch.int32, device=device)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)
            if failed is not None:
                print(f"[bench] rank {rank}: pipelined fused step not measured: {failed}", file=sys.stderr)
            pipe_ok = bool(int(okf.item()))
            pipe_ms = float(el.item()) / k * 1e3 if pipe_ok else None
        # Dependent chain WITHOUT a step barrier (round 4): y_{j+1} = A y_j, the peers' rows of step j arriving chunk by
        # chunk behind the expand of step j + 1.  A diagnostic like the pipelined form (the timed step multiplies a fixed
        # x and ends in the barrier), checked against the barrier chain's bits; any failure leaves the line as it is.
        chunk_ms, chunk_ok, chunk_wait_us = None, None, None
        if fop is not None and getattr(fop, "chunks", 0):
            failed, same = None, 0

            def local2(fn):
                nonlocal failed
                if failed is None:
                    try:
                        return fn()
                    except Exception as e:  # noqa: BLE001 - diagnostics only
                        failed = e
                return None

            def chain(n_steps, barrier):
                # (x in [0, 1): |y| grows ~2.5x per step -- alpha keeps a long chain in range)
                if barrier:
                    y = fop.step(x)
                    for _ in range(n_steps - 1):
                        y = fop.step(y.clone())
                    return y.clone()
                y = fop.step_dependent(x, alpha=1.0 if n_steps <= 8 else 0.4)
                for _ in range(n_steps - 1):
                    y = fop.step_dependent(alpha=1.0 if n_steps <= 8 else 0.4)
                y = fop.flush_chain()
                torch.cuda.synchronize()
                return y

            def check():
                a, b = chain(4, True), chain(4, False).clone()
                fop.check_status()
                return int(torch.equal(a, b))

            # (short time-outs while the form is being checked: across devices none of this has run before, and a flag that
            # never shows up must cost seconds, not the run; every rank goes through the same collectives whatever happened)
            saved_timeout, fop._timeout = fop._timeout, 1500
            same = local2(check) or 0
            okc = torch.tensor([0 if failed is not None else same], dtype=torch.int32, device=device)
            dist.all_reduce(okc, op=dist.ReduceOp.MIN)
            go = bool(int(okc.item()))
            if go:
                local2(lambda: chain(4, False))  # warm-up
            dist.barrier()
            t1 = time.perf_counter()
            if go:
                local2(lambda: chain(k, False))
            dist.barrier()
            el = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            wait_t = torch.tensor([local2(lambda: fop.chunk_wait_us()) or 0.0], dtype=torch.float64, device=device)
            dist.all_reduce(wait_t, op=dist.ReduceOp.MAX)
            okc = torch.tensor([0 if (failed is not None or not go) else 1], dtype=torch.int32, device=device)
            dist.all_reduce(okc, op=dist.ReduceOp.MIN)
            fop._timeout = saved_timeout
            if failed is not None:
                print(f"[bench] rank {rank}: chunked dependent chain not measured: {failed}", file=sys.stderr)
            chunk_ok = bool(int(okc.item()))
            # whatever happened, every kernel of the attempt has finished on every rank before anything else uses the
            # operator; a failed attempt leaves no armed wait and no time-out mark behind (the timed value, its post-check
            # and the parity check below are about the barrier step)
            torch.cuda.synchronize()
            dist.barrier()
            if not chunk_ok:
                fop._chunk_status.zero_()
                fop._chained = False
                torch.cuda.synchronize()
                dist.barrier()
            chunk_ms = float(el.item()) / k * 1e3 if chunk_ok else None
            chunk_wait_us = float(wait_t.item())
        fused_ms = fused_elapsed / args.steps * 1e3 if fused_elapsed is not None else None
        local_ms = local_s / k * 1e3
        shard_bytes = float(max(bounds[r + 1] - bounds[r] for r in range(world)) * tsize)
        # what one link and direction carried per second if the step time beyond the local kernels was all link time (every
        # rank stores its shard to each of the N - 1 peers over that peer's own link): a floor for the link rate, null when
        # the exchange hid behind the kernels or nothing crossed a link
        link_gbs = (shard_bytes / ((fused_ms - local_ms) * 1e-3) / 1e9
                    if (fused_ms is not None and world > 1 and not args.debug_one_gpu and fused_ms > local_ms) else None)
        diag = {"mode_timed": mode, "path_used": "fused" if mode == "fused" else "rccl",
                "local_spmv_ms": local_ms, "local_spmv_event_ms": local_ev,
                "rccl_nranks": dist.get_world_size(), "backend": dist.get_backend(),
                "chunks": getattr(fop, "chunks", 0) if fop is not None else 0,
                "chunked_step_ms": chunk_ms, "fused_chunked_step_ms": chunk_ms, "fused_chunked_check": chunk_ok,
                "expand_wait_us": chunk_wait_us,
                "fused_pipelined_step_ms": pipe_ms, "fused_pipelined_check": pipe_ok,
                "gather_ms": gather_s / k * 1e3, "rccl_step_ms": rccl_s / k * 1e3,
                "fused_step_ms": fused_ms,
                "fused_check": (None if args.fused != "auto" else bool(fused_op is not None and fused_post_check)),
                "fused_failure": str(fused_fail) if fused_fail is not None else None,
                "link_gbs_estimate": link_gbs,
                "fused_post_check": fused_post_check,
                "gather": getattr(rccl_op, "gather_mode", "inplace"),
                "rows_per_rank": [bounds[r + 1] - bounds[r] for r in range(world)],
                "note": "each figure: K steps between barriers, max over ranks; local = kernels only, gather = "
                        "collective only (y already computed), rccl_step = local + gather"}
    # Diagnostic pass outside the timed region: per-step events give the spread of single steps
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(1 if mode in ("overlapped", "fused") or rmat else chunks)]
          for _ in range(args.steps)]
    for i in range(args.steps):
        op.step(x, events=ev[i])
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    kern_ms = sorted(sum(a.elapsed_time(b) for a, b in step_ev) for step_ev in ev)
    values, rowptr, colind = a_chunks[0].values(), a_chunks[0].rowptr(), a_chunks[0].colind()

    # a second inspect of the same matrix (N = 1 only): the first one of a process also loads the code objects and
    # first-touches the pool, this one is what a caller pays from then on
    # (twice, the smaller figure: the first of the two still grows the memory pool by a second plan next to the live one)
    inspect_warm_ms = None
    if not multi and args.alg != "noplan" and mode == "plain":
        y_tmp = torch.empty(rows_local, dtype=dtype, device=device)
        for _ in range(2):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            info_w = sp.multiply_inspect(sp.matrix_opt(a_chunks[0]), x, y_tmp, alg=algs[args.alg])
            torch.cuda.synchronize()
            ms_w = (time.perf_counter() - t1) * 1e3
            inspect_warm_ms = ms_w if inspect_warm_ms is None else min(inspect_warm_ms, ms_w)
            del info_w
        del y_tmp

    # What was timed is also what is checked (outside the timed region): y of the timed operator for this x.
    y_timed = op.step(x)
    torch.cuda.synchronize()
    if fused_op is not None and mode == "fused":
        fused_op.check_status()
    y_timed = y_timed.clone()
    parity = None
    tol = 1e-6 if tsize == 4 else 1e-12
    if multi:
        # N > 1: the gathered y of the timed path against PLAN-FREE local SpMVs (spmv_vector_kernel on the caller's own
        # arrays: no re-tiled copy, no plan) gathered by the plain RCCL path.  Both sides re-associate the row sums, so
        # the bound is 2 * tol * sum |a x| (the row norms come from the same plan-free kernel on |A|, |x|).
        if rmat:
            mk = lambda chunks_: sharded.ShardedSpMV(chunks_[0], bounds, inspect=False)
        else:
            mk = lambda chunks_: sharded.PipelinedShardedSpMV(chunks_, ranges, inspect=False)
        if mode == "overlapped":
            parity = {"status": "skipped", "why": "striped ownership: covered by tests/test_sharded_cpu.py"}
        else:
            ref_op = mk(a_chunks)
            y_ref = ref_op.step(x).clone()
            abs_chunks = [sp.csr_view(a.values().abs(), a.rowptr(), a.colind(), a.shape(), a.size()) for a in a_chunks]
            y_abs = mk(abs_chunks).step(x.abs()).clone()
            # row lengths, gathered the same way: a k-entry row summed in another order differs by up to ~k * eps * sum|.|
            # (tests/util.py:assert_parity has the same floor; it only matters for the hub rows of the R-MAT graph)
            locals_ = ref_op.y_local if isinstance(ref_op.y_local, list) else [ref_op.y_local]
            for a, yl in zip(a_chunks, locals_):
                yl[:a.shape()[0]].copy_((a.rowptr()[1:].long() - a.rowptr()[:-1].long()).to(dtype))
            row_len = ref_op.gather().double().clone()
            torch.cuda.synchronize()
            eps = float(np.finfo(np.float32 if tsize == 4 else np.float64).eps)
            err = (y_timed.double() - y_ref.double()).abs()
            tol_row = torch.clamp(row_len * eps, min=2.0 * tol)
            bad = ~(err <= tol_row * y_abs.double() + float(np.finfo(np.float32 if tsize == 4 else np.float64).tiny))
            stat = torch.stack([bad.sum().double(), (err / y_abs.double().clamp_min(1e-300)).max()])
            dist.all_reduce(stat, op=dist.ReduceOp.MAX)
            # The reference above comes through the same kind of collective as the timed y: rows that never ARRIVED would be
            # the same stale values on both sides.  Independent of any gather: every rank sums ITS OWN rows of the timed y
            # (they are computed locally) in fp64, the sums are all-reduced, and the total must equal the sum over the whole
            # gathered vector on every rank -- a shard that a peer did not deliver, delivered twice or put at the wrong offset
            # changes that sum.  (Round 6; first exercised across devices by the SCALE run.)
            own = y_timed[bounds[rank]:bounds[rank + 1]].double()
            sums = torch.stack([own.sum(), own.abs().sum()])
            dist.all_reduce(sums)
            whole = torch.stack([y_timed.double().sum(), y_timed.double().abs().sum()])
            gather_bad = torch.tensor([float(((whole - sums).abs() > 1e-9 * sums[1] + 1e-300).any())], dtype=torch.float64, device=device)
            dist.all_reduce(gather_bad, op=dist.ReduceOp.MAX)
            gather_ok = not bool(gather_bad.item())
            parity = {"status": "pass" if (int(stat[0].item()) == 0 and gather_ok) else "fail", "rows": int(y_timed.numel()),
                      "rows_out_of_bound": int(stat[0].item()), "tol": f"max({2.0 * tol:g}, row_length * eps) * sum|a x|",
                      "worst_err_over_rownorm": float(stat[1].item()),
                      "gather_checksum": "pass" if gather_ok else "fail",
                      "against": "plan-free local SpMV (spmv_vector_kernel) + RCCL all-gather, every row, every rank; the gather "
                                 "itself by an all-reduced fp64 checksum of every rank's own rows against the gathered vector"}
            del y_ref, y_abs, abs_chunks, err, bad, row_len, tol_row, ref_op

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        gflops = 2.0 * nnz / (elapsed / args.steps) / 1e9
        local_bytes = spmv_bytes(rows_local, n, nnz_local, tsize)
        achieved = local_bytes / (kern_avg_ms * 1e-3) / 1e9
        alg_id = plan_info.get("alg")
        u8 = plan_info.get("sliced", {}).get("row_code_u8")
        nt = plan_info.get("sliced", {}).get("nt_product_stores")
        # (the counters were taken for both flavours of the expand's product stores; the plan says which one this box picked)
        pmc_key = prob["pmc_key"] + ("_nt" if (nt and prob["pmc_key"] == "spmv_cfg2") else "") if prob["pmc_key"] else None
        traffic, traffic_src = read_pmc_traffic(pmc_key) if (pmc_key and alg_id == 3) else (None, None)
        kernels = {3: f"pb_expand_kernel<{'float' if tsize == 4 else 'double'},{'true' if nt else 'false'}> + pb_reduce_kernel<{'float' if tsize == 4 else 'double'},4,{'2,true' if u8 else '4,false'}> "
                      "(one SpMV = this launch pair)",
                   2: f"spmv_rowblock_kernel<{'float,int,2048' if tsize == 4 else 'double,int,1024'}> (+ spmv_long_fixup_kernel)",
                   1: "spmv_vector_kernel<T,int,LPR>"}
        out = {
            "metric": "csr_spmv_gflops", "value": gflops, "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": prob["dtype_name"], "data": "synthetic",
            "config": {"workload": f"{prob['label']}, nnz={nnz}",
                       "rows": m, "cols": n, "nnz": nnz, "index_type": "int32",
                       "parallelism": ("single GPU" if not multi else
                                       f"row-sharded x{world}, all-gather(y) fused into the reduce kernels (peer stores "
                                       "into hipIpc-mapped copies of y + device-side step barrier)" if mode == "fused" else
                                       f"row-sharded x{world} ({'nnz-prefix' if rmat else 'equal'} shards), {chunks} stripe(s) per step "
                                       f"({mode}), one RCCL all-gather(y) per stripe"),
                       # what the headline is a number FOR: the sliced plan multiplies with a re-tiled snapshot of A, which
                       # is taken once for operands wrapped in matrix_opt; a plain inspected csr_view gets the
                       # tiles WITHOUT a copy of the values (round 5: the reduce reads the caller's array through LDS): 0.37 ms
                       # at cfg2 (secondary.cfg2_plain_csr_view; the row-block kernel on the caller's arrays: 1.71 ms)
                       "operand": "matrix_opt(csr_view) + multiply_inspect" if args.alg != "noplan" else "csr_view, no inspect",
                       # whose values a multiply reads: the reference's CPU path reads the caller's array of that call
                       # (multiply_impl.hpp:48-52); a snapshot plan reads the copy taken at inspect
                       "value_contract": ("snapshot" if (plan_info.get("alg") == 3 and not plan_info.get("sliced", {}).get("value_free")
                                                         and not plan_info.get("sliced", {}).get("refresh_each_call"))
                                          else "reads caller's values per call"),
                       "alg": args.alg, "plan": plan_info,
                       # what the plan holds on the device next to the caller's CSR arrays (which it does not copy or free)
                       "plan_bytes": plan_info.get("device_bytes"),
                       "plan_bytes_over_matrix": (plan_info.get("device_bytes") or 0) / float(nnz_local * (tsize + 4) + (rows_local + 1) * 4),
                       "inspect_ms_untimed": inspect_ms, "inspect_warm_ms_untimed": inspect_warm_ms,
                       "handle_create_ms": getattr(args, "handle_create_ms", None)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # fraction of the rate a streaming copy reaches on this part (6.29 TB/s, MI355X_MICROARCH.md)
                         "frac_of_achievable": achieved / 6290.0,
                         # PMC traffic was measured for the default cfg2 / 1 GPU / sliced plan only
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernels.get(alg_id, kernels[1]),
                         "algorithmic_bytes_per_launch": local_bytes, "kernel_avg_ms": kern_avg_ms,
                         "step_events_pass": {"note": "separate untimed pass with one event pair per step", "min_ms": kern_ms[0],
                                              "median_ms": kern_ms[len(kern_ms) // 2], "avg_ms": sum(kern_ms) / len(kern_ms)},
                         "algorithmic_gbs_whole_step": spmv_bytes(m, n, nnz, tsize) / (elapsed / args.steps) / 1e9},
            "multi_gpu": diag,
        }
        if world == 1 and chunks == 1 and not args.no_cpu_baseline:
            # the cpu_baseline leg computes the oracle's y for the same inputs; the timed plan's y is held to it
            y_ref, absrow, out["cpu_baseline"] = cpu_baseline_spmv(values, rowptr, colind, (m, n), x, nnz)
            if parity is None:
                parity = parity_spmv(y_timed.cpu().numpy(), y_ref, absrow, tol, row_len=np.diff(rowptr.cpu().numpy()))
                parity["against"] = "oracle_spmv (CPU restatement of multiply_impl.hpp:33-53), every row"
        else:
            out["cpu_baseline"] = None
        out["parity_check"] = parity["status"] if parity else "not run (--no-cpu-baseline)"
        out["parity"] = parity
        if world == 1 and not multi and args.workload == "spmv" and args.rows is None and args.cols is None \
                and args.alg == "auto" and not args.no_secondary:
            # The other single-GPU BASELINE configs, after the timed cfg2 loop and its check, in the same process: cfg4
            # (1 GPU), cfg3, cfg5, then the SURVEY 8(f) operations add / transpose / triangular solve -- each inspected,
            # warmed up, timed between one event pair and checked against the oracle.  The headline fields above are untouched; a failing secondary check fails the run (exit code 3).
            from bench_extra import secondary
            del y_timed, values, rowptr, colind, op, rccl_op, a_chunks, prob, info0, x
            torch.cuda.empty_cache()
            out["secondary"] = secondary(args, device, log=lambda msg: print(f"[bench] {msg}", file=sys.stderr))
            if any(v.get("parity_check") == "fail" for v in out["secondary"].values()):
                sys.stderr.write("[bench] SECONDARY PARITY CHECK FAILED\n")
                exit_code = 3
        # stdout carries the HEADLINE object alone (< 4 KB: bench_line.compact, tests/test_bench_line.py); the complete
        # result -- every secondary record with its plan, roofline, cpu_baseline and parity report -- goes to
        # bench_secondary.json (next to this file, and under gpurun_out/ where that exists) and, one short line per
        # record, to stderr
        from bench_line import compact, secondary_stderr_line
        detail = "bench_secondary.json" if world == 1 else f"bench_secondary_n{world}.json"
        for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
            if os.path.isdir(d):
                try:
                    with open(os.path.join(d, detail), "w") as f:
                        json.dump(out, f, indent=1)
                except OSError as e:
                    print(f"[bench] {detail} not written in {d}: {e}", file=sys.stderr)
        for name, rec in (out.get("secondary") or {}).items():
            print("[bench] " + secondary_stderr_line(name, rec), file=sys.stderr)
        sys.stderr.flush()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out if args.full_line else compact(out, detail_file=detail)) + "\n").encode())
        if parity and parity["status"] == "fail":
            sys.stderr.write("[bench] PARITY CHECK FAILED: the timed operator's y is outside the parity bound\n")
            exit_code = 3
    if multi:
        if fused_op is not None:
            try:
                fused_op.check_status()
            except Exception as e:  # noqa: BLE001 - a diagnostic form of the fused step timed out: reported, not fatal
                print(f"[bench] rank {rank}: {e}", file=sys.stderr)
            fused_op.close()
        dist.destroy_process_group()
    return exit_code


if __name__ == "__main__":
    sys.exit(main())
