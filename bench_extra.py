"""Secondary BASELINE configs on one GPU (not the headline bench line):
  --workload spmm      cfg3: fp32 CSR x dense, A 2M x 2M 32 nnz/row, B 2M x 128 row-major
  --workload spgemm    cfg5: fp32 CSR x CSR, 1M x 1M, 16 nnz/row, multiply_compute + multiply_fill
  --workload add | transpose | sptrsv   SURVEY 8f rows: CSR + CSR, CSR transpose, lower-triangular solve
Same JSON contract as bench.py; `value` is GFLOP/s of the timed operation.  The CPU baseline
(oracle, 1 core) is timed on a bounded row sample and scaled by nnz (stated in `sample`)."""
import json
import os
import time

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0


def _time_steps(fn, warmup, steps):
    """K steps between synchronisations, ONE pair of HIP events around the region (an event record per step is
    a release point between consecutive kernels and was measured to cost up to 25 us per step); then a
    diagnostic pass with per-step events for the spread.  Returns (wall seconds, [avg_ms, min_ms of the pass])."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    region = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    t0 = time.perf_counter()
    region[0].record()
    for _ in range(steps):
        fn()
    region[1].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    avg = region[0].elapsed_time(region[1]) / steps
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return elapsed, [avg, min(a.elapsed_time(b) for a, b in ev)]


def _emit(args, metric, flops, alg_bytes, elapsed, ms, workload, extra, cpu):
    avg = ms[0]
    out = {"metric": metric, "value": flops / (elapsed / args.steps) / 1e9, "unit": extra.pop("unit", "GFLOP/s"), "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": extra.pop("dtype"),
           "data": "synthetic", "config": {"workload": workload, **extra},
           "roofline": {"bound": "hbm", "achieved": alg_bytes / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg_bytes / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "algorithmic_bytes_per_launch": alg_bytes, "kernel_avg_ms": avg, "step_events_pass_min_ms": ms[1]},
           "cpu_baseline": cpu}
    print(json.dumps(out))


def run_extra(args, device):
    import spblas_reference_amd as sp
    from oracle import oracle
    from spblas_reference_amd import generate

    if args.workload in ("spmm", "spmm_banded", "spmm_rmat"):
        m = args.rows or 2_000_000
        ncols = 128
        banded = args.workload == "spmm_banded"
        rmat = args.workload == "spmm_rmat"
        if rmat:
            # cfg3's size class with a skewed A (R-MAT scale 21, 32 entries per row on average, duplicates kept): hub rows go
            # to the split long-row kernel, hot columns give the L2s B rows to reuse (tests/test_gpu_configs.py)
            values, rowptr, colind, shape, nnz = generate.rmat_csr_device(21, 32, dtype=torch.float32, seed=1, device=device)
            m = shape[0]
        elif banded:
            # cfg3's shape with 64 entries per row, all within 48 columns of the diagonal: neighbouring rows share B
            # rows, every block of 32 rows is >= 1/5 dense over the 2-3 tiles of 64 columns it touches, and
            # multiply_inspect hands it to the LDS-staged matrix-core kernel
            g0 = torch.Generator(device=device).manual_seed(11)
            per = 64
            off = torch.randint(-48, 49, (m, per), device=device, generator=g0)
            colind = ((torch.arange(m, device=device)[:, None] + off) % m).to(torch.int32).reshape(-1)
            rowptr = (torch.arange(m + 1, device=device, dtype=torch.int64) * per).to(torch.int32)
            values = torch.rand(m * per, device=device, generator=g0)
            shape, nnz = (m, m), m * per
        else:
            values, rowptr, colind, shape, nnz = generate.uniform_csr_device(m, m, 32, seed=0, device=device)
        a = sp.csr_view(values, rowptr, colind, shape, nnz)
        g = torch.Generator(device=device).manual_seed(3)
        B = torch.rand((m, ncols), device=device, generator=g)
        C = torch.empty((m, ncols), device=device)
        info = sp.multiply_inspect(sp.matrix_opt(a) if rmat else a, B, C)
        elapsed, ms = _time_steps(lambda: sp.multiply(info, a, B, C), args.warmup, args.steps)
        alg_bytes = nnz * 8 + (m + 1) * 4 + 2 * m * ncols * 4
        cpu = None
        if not args.no_cpu_baseline:
            rows = 20_000  # bounded sample: first 20k rows (640k nnz x 128 columns)
            rp = rowptr[:rows + 1].cpu().numpy()
            ci, v = colind[:rp[-1]].cpu().numpy(), values[:rp[-1]].cpu().numpy()
            Bh = B.cpu().numpy()
            t0 = time.perf_counter()
            oracle.spmm((rows, m), rp, ci, v, Bh)
            dt = time.perf_counter() - t0
            cpu = {"value": 2.0 * rp[-1] * ncols / dt / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "port",
                   "sample": f"first {rows} rows ({int(rp[-1])} nnz x {ncols} columns) of the same A and B, 1 run of oracle_spmm"}
        mi = info.state_.spmm_info()
        _emit(args, "csr_spmm_gflops", 2.0 * nnz * ncols, alg_bytes, elapsed, ms,
              (f"R-MAT variant of cfg3: fp32 CSR x dense SpMM, A R-MAT scale 21 ({m}x{m}, {nnz} entries, duplicates kept), "
               f"B {m}x{ncols} row-major" if rmat else
               f"banded variant of cfg3: fp32 CSR x dense SpMM, A {m}x{m} 64 nnz/row within 48 columns of the diagonal, "
               f"B {m}x{ncols} row-major" if banded else
               f"cfg3: fp32 CSR x dense SpMM, A {m}x{m} 32 nnz/row uniform random, B {m}x{ncols} row-major"),
              {"dtype": "f32", "rows": m, "nnz": nnz, "ncols": ncols, "spmm_inspect": mi,
               "kernel": "spmm_panel_kernel<int> (v_mfma_f32_32x32x2_f32)" if mi["panel_blocks"] > 0 else
                         "spmm_rowgroup_kernel<float,int,4>"}, cpu)
        return

    if args.workload == "spgemm":
        m = args.rows or 1_000_000
        av, ar, ac, ash, annz = generate.uniform_csr_device(m, m, 16, seed=0, device=device)
        bv, br, bc, bsh, bnnz = generate.uniform_csr_device(m, m, 16, seed=1, device=device)
        a, b = sp.csr_view(av, ar, ac, ash, annz), sp.csr_view(bv, br, bc, bsh, bnnz)
        c_rp = torch.zeros(m + 1, dtype=torch.int32, device=device)
        c = sp.csr_view(None, c_rp, None, (m, m), 0)
        state = sp.spgemm_state_t()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sp.multiply_compute(state, a, b, c)
        torch.cuda.synchronize()
        compute_ms = (time.perf_counter() - t0) * 1e3
        cn = state.result_nnz()
        c.update(torch.empty(cn, device=device), c_rp, torch.empty(cn, dtype=torch.int32, device=device), (m, m), cn)
        products = int((br.long()[ac.long() + 1] - br.long()[ac.long()]).sum().item())
        # The timed step is the ONE-SHOT fill -- what the reference's call shape (examples/simple_spgemm.cpp:52-60: one
        # multiply_compute, one multiply_fill) pays: the hash kernels, columns and values written (recording of ranks
        # for later fills switched off for these steps so that every step is a first fill).
        os.environ["SPBLAS_GFX950_SPGEMM_REUSE"] = "0"
        elapsed, ms = _time_steps(lambda: sp.multiply_fill(state, a, b, c), args.warmup, args.steps)
        del os.environ["SPBLAS_GFX950_SPGEMM_REUSE"]
        alg_bytes = 2 * (annz * 8 + (m + 1) * 4) + cn * 8 + (m + 1) * 4
        # ... and, as a secondary figure, repeated fills of the same structure (multiply_numeric / symbolic-numeric reuse):
        # the second fill records the product ranks once, later ones accumulate by rank and leave the columns alone
        fills = []
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sp.multiply_fill(state, a, b, c)
            torch.cuda.synchronize()
            fills.append((time.perf_counter() - t0) * 1e3)
        reuse_elapsed, reuse_ms = _time_steps(lambda: sp.multiply_fill(state, a, b, c), args.warmup, args.steps)
        reuse_bytes = alg_bytes - cn * 4  # the column indices are neither read nor written by those fills
        # the symbolic phase once more on a fresh state: the first call above also loaded the code object of spgemm.hip
        c2_rp = torch.zeros(m + 1, dtype=torch.int32, device=device)
        c2 = sp.csr_view(None, c2_rp, None, (m, m), 0)
        state2 = sp.spgemm_state_t()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sp.multiply_compute(state2, a, b, c2)
        torch.cuda.synchronize()
        compute_warm_ms = (time.perf_counter() - t0) * 1e3
        assert state2.result_nnz() == cn and torch.equal(c2_rp, c_rp)
        del state2
        cpu = None
        if not args.no_cpu_baseline:
            rows = 20_000
            rp = ar[:rows + 1].cpu().numpy()
            sub = ((rows, m), rp, ac[:rp[-1]].cpu().numpy(), av[:rp[-1]].cpu().numpy())
            bh = ((m, m), br.cpu().numpy(), bc.cpu().numpy(), bv.cpu().numpy())
            t0 = time.perf_counter()
            n_ref, _ = oracle.spgemm_symbolic(sub[0], sub[1], sub[2], bh[0], bh[1], bh[2])
            t1 = time.perf_counter()
            oracle.spgemm_numeric(sub[0], sub[1], sub[2], sub[3], bh[0], bh[1], bh[2], bh[3], capacity=n_ref)
            t2 = time.perf_counter()
            cpu = {"value": 2.0 * (products * rows / m) / (t2 - t1) / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "port",
                   "sample": f"first {rows} rows of A x full B: symbolic {t1 - t0:.3f} s, numeric {t2 - t1:.3f} s (numeric timed)"}
        reuse_step = reuse_elapsed / args.steps
        _emit(args, "csr_spgemm_fill_gflops", 2.0 * products, alg_bytes, elapsed, ms,
              f"cfg5: fp32 CSR x CSR SpGEMM {m}x{m}, 16 nnz/row uniform random; timed step = one-shot multiply_fill "
              "(hash accumulators, sorted columns and values written), after multiply_compute",
              {"dtype": "f32", "rows": m, "products": products, "nnz_c": cn, "multiply_compute_ms_untimed": compute_warm_ms,
               "multiply_compute_first_call_ms": compute_ms,
               "compute_plus_one_shot_fill_ms": compute_warm_ms + elapsed / args.steps * 1e3,
               "repeated_fills": {"note": "numeric reuse on one symbolic result (accumulation by recorded product ranks, "
                                          "columns kept): secondary figure, not the metric value",
                                  "first_fill_ms_untimed": fills[0], "second_fill_ms_untimed_records_ranks": fills[1],
                                  "ms_per_fill": reuse_step * 1e3, "gflops": 2.0 * products / reuse_step / 1e9,
                                  "algorithmic_bytes": reuse_bytes,
                                  "roofline_frac": reuse_bytes / (reuse_ms[0] * 1e-3) / 1e9 / 8000.0,
                                  "kernel": "spg_ranked_fill_kernel<float,16,256,4,false> (+ spg_rank_record_kernel<64,256> once)"},
               "kernel": "spg_hash_kernel<float,9,64,true>"}, cpu)
        return

    if args.workload == "add":  # C = A + B, timed step = add_compute (numeric)
        m = args.rows or 1_000_000
        av, ar, ac, ash, annz = generate.uniform_csr_device(m, m, 16, seed=0, device=device)
        bv, br, bc, bsh, bnnz = generate.uniform_csr_device(m, m, 16, seed=1, device=device)
        a, b = sp.csr_view(av, ar, ac, ash, annz), sp.csr_view(bv, br, bc, bsh, bnnz)
        c_rp = torch.zeros(m + 1, dtype=torch.int32, device=device)
        c = sp.csr_view(None, c_rp, None, (m, m), 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info = sp.add_inspect(a, b, c)
        torch.cuda.synchronize()
        inspect_ms = (time.perf_counter() - t0) * 1e3
        cn = info.result_nnz()
        c.update(torch.empty(cn, device=device), c_rp, torch.empty(cn, dtype=torch.int32, device=device), (m, m), cn)
        elapsed, ms = _time_steps(lambda: sp.add_compute(info, a, b, c), args.warmup, args.steps)
        alg_bytes = (annz + bnnz + cn) * 8 + 3 * (m + 1) * 4
        cpu = None
        if not args.no_cpu_baseline:
            rows = 200_000
            ra, rb = ar[:rows + 1].cpu().numpy(), br[:rows + 1].cpu().numpy()
            t0 = time.perf_counter()
            oracle.add((rows, m), ra, ac[:ra[-1]].cpu().numpy(), av[:ra[-1]].cpu().numpy(), (rows, m), rb,
                       bc[:rb[-1]].cpu().numpy(), bv[:rb[-1]].cpu().numpy())
            dt = time.perf_counter() - t0
            cpu = {"value": (ra[-1] + rb[-1]) / dt / 1e9, "unit": "Gentries/s", "cores": 1, "kind": "port",
                   "sample": f"first {rows} rows, oracle_add_f32 (SPA + sort per row)"}
        _emit(args, "csr_add_gentries", float(annz + bnnz), alg_bytes, elapsed, ms,
              f"8f: fp32 CSR + CSR add {m}x{m}, 16 nnz/row each, uniform random; timed step = add_compute; value = input entries/ns",
              {"dtype": "f32", "unit": "Gentries/s", "rows": m, "nnz_c": cn, "add_inspect_ms_untimed": inspect_ms}, cpu)
        return

    if args.workload == "transpose":  # B = A^T (stable counting sort)
        m = args.rows or 10_000_000
        av, ar, ac, ash, annz = generate.uniform_csr_device(m, m, 10, seed=0, device=device)
        a = sp.csr_view(av, ar, ac, ash, annz)
        t_rp = torch.empty(m + 1, dtype=torch.int32, device=device)
        t_ci = torch.empty(annz, dtype=torch.int32, device=device)
        t_v = torch.empty(annz, device=device)
        bview = sp.csr_view(t_v, t_rp, t_ci, (m, m), annz)
        elapsed, ms = _time_steps(lambda: sp.transpose(a, bview), args.warmup, args.steps)
        alg_bytes = 2 * (annz * 8 + (m + 1) * 4)
        cpu = None
        if not args.no_cpu_baseline:
            rows = 1_000_000
            rp = ar[:rows + 1].cpu().numpy()
            t0 = time.perf_counter()
            oracle.transpose((rows, m), rp, ac[:rp[-1]].cpu().numpy(), av[:rp[-1]].cpu().numpy())
            dt = time.perf_counter() - t0
            cpu = {"value": rp[-1] / dt / 1e9, "unit": "Gentries/s", "cores": 1, "kind": "port",
                   "sample": f"first {rows} rows, oracle_transpose_f32"}
        _emit(args, "csr_transpose_gentries", float(annz), alg_bytes, elapsed, ms,
              f"8f: fp32 CSR transpose {m}x{m}, 10 nnz/row uniform random; value = entries/ns",
              {"dtype": "f32", "unit": "Gentries/s", "rows": m, "nnz": annz}, cpu)
        return

    if args.workload == "sptrsv":  # x = inv(L) b, L random lower triangular + diagonal
        m = args.rows or 4_000_000
        k = 8
        g = torch.Generator(device=device).manual_seed(0)
        rows = torch.arange(m, device=device).repeat_interleave(k)
        cols = (torch.rand(m * k, device=device, generator=g, dtype=torch.float64) * rows.double()).long().clamp_(min=0)
        cols = torch.minimum(cols, rows)                       # col <= row; col == row only in row 0
        vals = (torch.rand(m * k, device=device, generator=g) - 0.5) * (0.5 / k)
        # append the diagonal as the last entry of every row
        rp = torch.arange(m + 1, device=device, dtype=torch.int64) * (k + 1)
        colind = torch.empty(m * (k + 1), dtype=torch.int32, device=device)
        values = torch.empty(m * (k + 1), device=device)
        colind.view(m, k + 1)[:, :k] = cols.view(m, k).int()
        colind.view(m, k + 1)[:, k] = torch.arange(m, device=device, dtype=torch.int32)
        values.view(m, k + 1)[:, :k] = vals.view(m, k)
        values.view(m, k + 1)[:, k] = 1.0 + torch.rand(m, device=device, generator=g)
        nnz = m * (k + 1)
        a = sp.csr_view(values, rp.int(), colind, (m, m), nnz)
        b = torch.rand(m, device=device, generator=g)
        x = torch.empty(m, device=device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info = sp.triangular_solve_inspect(a, sp.lower_triangle, sp.explicit_diagonal, b, x)
        torch.cuda.synchronize()
        inspect_first_ms = (time.perf_counter() - t0) * 1e3   # includes loading the code object of sptrsv.hip
        del info
        t0 = time.perf_counter()
        info = sp.triangular_solve_inspect(a, sp.lower_triangle, sp.explicit_diagonal, b, x)
        torch.cuda.synchronize()
        inspect_ms = (time.perf_counter() - t0) * 1e3
        elapsed, ms = _time_steps(lambda: sp.triangular_solve(info, a, sp.lower_triangle, sp.explicit_diagonal, b, x),
                                  args.warmup, args.steps)
        alg_bytes = nnz * 8 + (m + 1) * 4 + 2 * m * 4
        cpu = None
        if not args.no_cpu_baseline:
            t0 = time.perf_counter()
            oracle.triangular_solve((m, m), rp.int().cpu().numpy(), colind.cpu().numpy(), values.cpu().numpy(),
                                    b.cpu().numpy())
            dt = time.perf_counter() - t0
            cpu = {"value": 2.0 * nnz / dt / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "port",
                   "sample": f"full workload ({nnz} nnz), oracle_trsv_f32 (sequential reference loop)"}
        _emit(args, "csr_sptrsv_gflops", 2.0 * nnz, alg_bytes, elapsed, ms,
              f"8f: fp32 lower-triangular solve {m}x{m}, {k} random sub-diagonal entries per row + diagonal",
              {"dtype": "f32", "rows": m, "nnz": nnz, "plan": info.state_.info(),
               "triangular_solve_inspect_ms_untimed": inspect_ms,
               "triangular_solve_inspect_first_call_ms": inspect_first_ms}, cpu)
        return

    raise SystemExit(f"unknown workload {args.workload}")
