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
ROOT = os.path.dirname(os.path.abspath(__file__))


def read_pmc_traffic(name):
    """(HBM bytes per launch, provenance) from the committed PMC profile (profiles/pmc_traffic.json), or (None, None).
    The provenance names the profile and the hash of the kernel source it was measured on (csrc/spmv_sliced.hip unless
    the stamp names another `source_file`), and says whether that file has changed since: a stale constant must be
    visible in the JSON line."""
    import hashlib
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if name not in d:
            return None, None
        stamp = dict((d.get("_stamp") or {}).get(name) or {})
        src = stamp.get("source_file", "spmv_sliced.hip")
        key = src.replace(".", "_") + "_sha256_16"
        with open(os.path.join(ROOT, "spblas-reference_amd", "csrc", src), "rb") as f:
            now = hashlib.sha256(f.read()).hexdigest()[:16]
        stamp[key + "_now"] = now
        stamp["stale"] = stamp.get(key) != now
        return d.get(name), stamp
    except Exception:
        return None, None


def rows_subproblem(rows, rowptr_d, colind_d, values_d):
    """CSR of the selected rows as host arrays (gathered on the device): what the parity checks hand to the oracle."""
    rows_d = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(rowptr_d.device)
    rp = rowptr_d.long()
    lo, ln = rp[rows_d], rp[rows_d + 1] - rp[rows_d]
    sub_rp = torch.zeros(len(rows) + 1, dtype=torch.int64, device=rowptr_d.device)
    torch.cumsum(ln, 0, out=sub_rp[1:])
    total = int(sub_rp[-1])
    owner = torch.repeat_interleave(torch.arange(len(rows), device=rowptr_d.device), ln)
    idx = lo[owner] + (torch.arange(total, device=rowptr_d.device) - sub_rp[owner])
    return sub_rp.cpu().numpy().astype(np.int32), colind_d[idx].cpu().numpy(), values_d[idx].cpu().numpy()


def parity_rows(got, ref, absref, tol, eps, row_len):
    """Norm-wise bound of SURVEY.md section 8c (tests/util.py:assert_parity): |got - ref| <= max(tol, k/2 * eps) * sum|a b| per row."""
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    tol_row = np.maximum(tol, 0.5 * np.asarray(row_len, dtype=np.float64) * eps)
    if err.ndim == 2:
        tol_row = tol_row[:, None]
    bad = ~(err <= tol_row * absref + 1e-300)
    return int(bad.sum()), float((err / np.maximum(absref, 1e-300)).max()) if err.size else 0.0


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


def _emit(args, metric, flops, alg_bytes, elapsed, ms, workload, extra, cpu, parity=None, pmc_key=None, mfma_key=None):
    avg = ms[0]
    traffic, traffic_src = read_pmc_traffic(pmc_key) if pmc_key else (None, None)
    # matrix-core utilisation of the SpMM panel kernel (north_star: "MFMA utilisation for SpMM"): committed PMC figures
    # (SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES and friends, tools/prof_mfma.sh), stamped like the traffic
    mfma, mfma_src = read_pmc_traffic(mfma_key) if (mfma_key and mfma_key != "none") else (None, None)
    kernel = extra.pop("kernel", None)
    out = {"metric": metric, "value": flops / (elapsed / args.steps) / 1e9, "unit": extra.pop("unit", "GFLOP/s"), "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": extra.pop("dtype"),
           "data": "synthetic", "config": {"workload": workload, **extra},
           "roofline": {"bound": "hbm", "achieved": alg_bytes / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg_bytes / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                        "traffic_source": traffic_src, "kernel": kernel,
                        "algorithmic_bytes_per_launch": alg_bytes, "kernel_avg_ms": avg, "step_events_pass_min_ms": ms[1]},
           "cpu_baseline": cpu}
    if mfma_key == "none":
        # north_star: "MFMA utilisation for SpMM".  The kernel that runs by default since round 6 (spmm_band_kernel) issues no
        # matrix-core instruction -- fp32 MFMA runs at the fp32 vector rate on gfx950 and both matrix-core forms were measured
        # slower on this workload; their utilisation, from the committed counters, stays in the record
        out["roofline"]["mfma_util"] = 0.0
        alt = {}
        for name, key in (("spmm_panel_kernel (SPBLAS_GFX950_SPMM_BAND=0)", "spmm_banded_mfma"),
                          ("spmm_band_mfma_kernel (SPBLAS_GFX950_SPMM_BAND_DENSE=250)", "spmm_banded_mfma_window")):
            v, src = read_pmc_traffic(key)
            if isinstance(v, dict):
                alt[name] = {"mfma_util": v.get("mfma_util"), "avg_us": v.get("avg_us"), "profile": (src or {}).get("profile"),
                             "stale": (src or {}).get("stale")}
        out["roofline"]["mfma_alternatives"] = alt
    elif mfma_key:
        out["roofline"]["mfma_util"] = (mfma or {}).get("mfma_util") if isinstance(mfma, dict) else mfma
        out["roofline"]["mfma_detail"] = mfma if isinstance(mfma, dict) else None
        out["roofline"]["mfma_source"] = mfma_src
    if parity is not None:
        out["parity_check"] = parity["status"]
        out["parity"] = parity
    return out


def run_extra(args, device):
    """One secondary workload as its own bench line (python bench.py --workload spmm|spgemm|...)."""
    out = _run(args, device)
    # like the headline: the compact object on stdout, the complete record next to bench.py (and under gpurun_out/)
    from bench_line import compact
    detail = f"bench_secondary_{args.workload}.json"
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, detail), "w") as f:
                    json.dump(out, f, indent=1)
            except OSError:
                pass
    print(json.dumps(out if getattr(args, "full_line", False) else compact(out, detail_file=detail)))
    return 3 if out.get("parity_check") == "fail" else 0


def secondary(args, device, log=None):
    """BASELINE cfg4 (single-GPU leg), cfg3 and cfg5 after the headline cfg2 loop, in the same process, then the three
    SURVEY 8(f) operations (add, transpose, triangular solve): each one inspected, warmed up, timed between one pair of
    HIP events and checked against the oracle; compact records for the `secondary` object of bench.py's JSON line."""
    import copy
    res = {}
    todo = [("cfg2_plain_csr_view", "spmv_plain"), ("cfg2_poisson", "spmv_poisson1"), ("cfg4", "spmv_rmat1"), ("cfg4_shards_of_8", "spmv_rmat_shards"), ("cfg3", "spmm"), ("spmm_banded", "spmm_banded"),
            ("cfg5", "spgemm")]
    if not getattr(args, "no_8f", False):  # add, transpose, triangular solve at their bench sizes
        todo += [("f_csc_spmv", "csc_spmv"), ("f_add", "add"), ("f_spgemm4", "spgemm4"), ("f_transpose", "transpose"),
                 ("f_sptrsv", "sptrsv")]
    for name, workload in todo:
        a2 = copy.copy(args)
        a2.workload, a2.rows, a2.cols = workload, None, None
        a2.steps, a2.warmup = max(10, min(args.steps, 20)), 5
        t0 = time.perf_counter()
        try:
            r = _run(a2, device)
            cfg = r["config"]
            res[name] = {"workload": cfg.pop("workload"), "metric": r["metric"], "value": r["value"], "unit": r["unit"],
                         "dtype": r["dtype"], "steps": r["steps"], "warmup": r["warmup"], "ms_per_step": r["ms_per_step"],
                         "roofline": r["roofline"], "parity_check": r.get("parity_check", "not run"),
                         "parity": r.get("parity"), "cpu_baseline": r["cpu_baseline"], "detail": cfg}
        except Exception as e:  # noqa: BLE001 - a broken secondary config must show in the line, not hide the headline
            res[name] = {"workload": workload, "parity_check": "fail", "error": f"{type(e).__name__}: {e}"}
        res[name]["wall_s"] = time.perf_counter() - t0
        if log:
            log(f"secondary {name}: {res[name].get('ms_per_step')} ms/step, parity {res[name]['parity_check']}, "
                f"{res[name]['wall_s']:.1f} s")
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return res


def _spgemm_parity(oracle, m, a_d, b_d, c_rp, c, cn, one_shot):
    """cfg5: nnz(C) and EVERY row offset exact against a full symbolic run of the oracle on the host; the columns
    (exact, ascending) and values (1e-6 norm-wise) of sampled rows of the timed one-shot fill against oracle_spgemm."""
    (ar, ac, av), (br, bc, bv) = a_d, b_d
    a_h, b_h = (ar.cpu().numpy(), ac.cpu().numpy()), (br.cpu().numpy(), bc.cpu().numpy())
    n_ref, row_nnz = oracle.spgemm_symbolic((m, m), a_h[0], a_h[1], (m, m), b_h[0], b_h[1])
    rowptr_ok = bool(cn == n_ref and np.array_equal(c_rp.cpu().numpy().astype(np.int64),
                                                    np.concatenate([[0], np.cumsum(row_nnz)])))
    rows = np.unique(np.concatenate([np.arange(0, m, 1009), [m - 1]]))
    sub_rp, sub_c, sub_v = rows_subproblem(rows, ar, ac, av)
    bv_h = bv.cpu().numpy()
    n_sub, _ = oracle.spgemm_symbolic((len(rows), m), sub_rp, sub_c, (m, m), b_h[0], b_h[1])
    cr, cc, cv = oracle.spgemm_numeric((len(rows), m), sub_rp, sub_c, sub_v, (m, m), b_h[0], b_h[1], bv_h, capacity=n_sub)
    _, _, cabs = oracle.spgemm_numeric((len(rows), m), sub_rp, sub_c, np.abs(sub_v), (m, m), b_h[0], b_h[1], np.abs(bv_h),
                                       capacity=n_sub)
    got_rp, got_c, got_v = rows_subproblem(rows, c_rp, one_shot[0], one_shot[1])
    cols_ok = bool(np.array_equal(got_rp, cr) and np.array_equal(got_c, cc))
    nbad, worst = (1, float("inf"))
    if cols_ok:
        # a row of C sums <= 16 products per entry: the k/2*eps floor never applies
        nbad, worst = parity_rows(got_v, cv, cabs.astype(np.float64), 1e-6, float(np.finfo(np.float32).eps), np.zeros(len(cv)))
    finite = bool(torch.isfinite(one_shot[1]).all())
    ok = rowptr_ok and cols_ok and nbad == 0 and finite
    return {"status": "pass" if ok else "fail", "nnz_c": int(cn), "nnz_c_oracle": int(n_ref), "rowptr_exact_all_rows": rowptr_ok,
            "sampled_rows": int(len(rows)), "sampled_entries": int(len(cv)), "sampled_colind_exact": cols_ok,
            "values_out_of_bound": nbad, "tol": 1e-6, "worst_err_over_norm": worst,
            "against": "oracle_spgemm_symbolic over all rows (spgemm_gustavsons.hpp:57-89); oracle_spgemm_numeric on every "
                       "1 009th row (spgemm_gustavsons.hpp:17-52): sorted columns exact, values norm-wise"}


def _run_spmv_rmat1(args, device, sp, oracle, generate):
    """BASELINE cfg4, single-GPU leg: fp64 CSR SpMV on the R-MAT scale-24 graph (edge factor 16, duplicates kept), the
    matrix_opt + multiply_inspect call shape; every row of the timed y against oracle_spmv on the host."""
    scale = 24 if args.rows is None else int(np.log2(args.rows))
    values, rowptr, colind, shape, nnz = generate.rmat_csr_device(scale, 16, dtype=torch.float64, seed=0, device=device)
    m = n = shape[0]
    a = sp.csr_view(values, rowptr, colind, shape, nnz)
    g = torch.Generator(device=device).manual_seed(7)
    x = torch.rand(n, dtype=torch.float64, device=device, generator=g)
    y = torch.empty(m, dtype=torch.float64, device=device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = sp.multiply_inspect(sp.matrix_opt(a), x, y)
    torch.cuda.synchronize()
    inspect_ms = (time.perf_counter() - t0) * 1e3
    elapsed, ms = _time_steps(lambda: sp.multiply(info, a, x, y), args.warmup, args.steps)
    plan = info.state_.info()
    si = info.state_.sliced_info() if hasattr(info.state_, "sliced_info") else {}
    if plan.get("alg") == 3:
        plan["sliced"] = si
    elif si.get("auto_trial"):
        plan["auto_trial"] = {k: si[k] for k in ("trial_rowblock_ns", "trial_sliced_ns")}
    y.fill_(float("nan"))
    sp.multiply(info, a, x, y)
    torch.cuda.synchronize()
    alg_bytes = nnz * 12 + (m + 1) * 4 + (n + m) * 8
    cpu, parity = None, None
    if not args.no_cpu_baseline:
        v, rp, ci, xh = values.cpu().numpy(), rowptr.cpu().numpy(), colind.cpu().numpy(), x.cpu().numpy()
        try:
            oracle.load(native=True)
            native = True
        except Exception:
            native = False
        t0 = time.perf_counter()
        y_ref = oracle.spmv(shape, rp, ci, v, xh, native=native)
        dt = time.perf_counter() - t0
        absrow = oracle.spmv_absrow(rp, ci, v, xh)
        cpu = {"value": 2.0 * nnz / dt / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "port", "seconds": dt,
               "sample": f"full workload ({nnz} nnz), 1 run of oracle_spmv (-O3 -march={'native' if native else 'x86-64-v3'})"}
        lens = np.diff(rp)
        nbad, worst = parity_rows(y.cpu().numpy(), y_ref, absrow.astype(np.float64), 1e-12, float(np.finfo(np.float64).eps), lens)
        parity = {"status": "pass" if nbad == 0 else "fail", "rows": int(m), "rows_out_of_bound": nbad, "tol": 1e-12,
                  "worst_err_over_rownorm": worst,
                  "against": "oracle_spmv (CPU restatement of multiply_impl.hpp:33-53), every row"}
    dbl = "double"
    kern = {3: f"pb_expand_kernel<{dbl},{'true' if si.get('nt_product_stores') else 'false'}> + pb_reduce_kernel<{dbl},...> "
               "(+ pb_split_finish_kernel, pb_empty_rows_kernel; one SpMV = this launch group)",
            2: "spmv_rowblock_kernel<double,int,1024> (+ spmv_long_fixup_kernel)"}.get(plan.get("alg"), "spmv_vector_kernel")
    return _emit(args, "csr_spmv_gflops", 2.0 * nnz, alg_bytes, elapsed, ms,
                 f"cfg4 (single-GPU leg): fp64 CSR SpMV, R-MAT scale {scale}, edge factor 16, duplicates kept, nnz={nnz}",
                 {"dtype": "f64", "rows": m, "nnz": nnz, "operand": "matrix_opt(csr_view) + multiply_inspect", "plan": plan,
                  "plan_bytes": plan.get("device_bytes"),
                  "plan_bytes_over_matrix": (plan.get("device_bytes") or 0) / float(nnz * 12 + (m + 1) * 4),
                  "inspect_ms_untimed": inspect_ms, "kernel": kern}, cpu, parity=parity,
                 pmc_key="spmv_rmat" if (args.rows is None and plan.get("alg") == 3) else None)


def _run_spmv_plain(args, device, sp, oracle, generate):
    """cfg2's matrix as a PLAIN inspected csr_view (no matrix_opt): north_star's call shape taken literally.  Every
    multiply has to read the caller's values of that call (multiply_impl.hpp:48-52), so the values are rewritten IN PLACE
    between inspect and the timed loop and the checked y must follow them.  Since the end of round 4 the re-tiled plan is
    chosen here too when "refresh the plan's copy + tiles" beats the row-block kernel in inspect's timed trial."""
    m = n = args.rows or 10_000_000
    values, rowptr, colind, shape, nnz = generate.uniform_csr_device(m, n, 10, seed=0, device=device)
    a = sp.csr_view(values, rowptr, colind, shape, nnz)
    g = torch.Generator(device=device).manual_seed(11)
    x = torch.rand(n, device=device, generator=g)
    y = torch.empty(m, device=device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = sp.multiply_inspect(a, x, y)
    torch.cuda.synchronize()
    inspect_ms = (time.perf_counter() - t0) * 1e3
    values.mul_(-0.5).add_(0.125)  # in place, after inspect: a plan that kept the old values would be caught below
    elapsed, ms = _time_steps(lambda: sp.multiply(info, a, x, y), args.warmup, args.steps)
    plan = info.state_.info()
    si = info.state_.sliced_info() if hasattr(info.state_, "sliced_info") else {}
    if plan.get("alg") == 3:
        plan["sliced"] = si
    y.fill_(float("nan"))
    sp.multiply(info, a, x, y)
    torch.cuda.synchronize()
    alg_bytes = nnz * 8 + (m + 1) * 4 + (n + m) * 4
    rows = np.unique(np.concatenate([np.arange(0, min(m, 1500)), np.arange(max(0, m - 1500), m),
                                     np.random.default_rng(5).integers(0, m, 3000)]))
    sub_rp, sub_ci, sub_v = rows_subproblem(rows, rowptr, colind, values)
    xh = x.cpu().numpy()
    y_ref = oracle.spmv((len(rows), n), sub_rp, sub_ci, sub_v, xh)
    absrow = oracle.spmv_absrow(sub_rp, sub_ci, sub_v, xh)
    nbad, worst = parity_rows(y[torch.from_numpy(rows).to(device)].cpu().numpy(), y_ref, absrow.astype(np.float64), 1e-6,
                              float(np.finfo(np.float32).eps), np.diff(sub_rp))
    parity = {"status": "pass" if nbad == 0 else "fail", "rows": int(len(rows)), "rows_out_of_bound": nbad, "tol": 1e-6,
              "worst_err_over_rownorm": worst,
              "against": "oracle_spmv on the first / last 1 500 and 3 000 sampled rows, with the values the caller wrote in "
                         "place AFTER multiply_inspect"}
    vf = bool(si.get("value_free", 0))
    kern = {3: ("pb_expand_kernel<float,false,true> + pb_reduce_vf_kernel<float,8,2,true> (one SpMV = this launch pair: the "
                "plan holds no values -- the expand moves x[col], the reduce multiplies by the caller's array through an LDS "
                "window per bin)") if vf else
               ("pb_refresh_bins_kernel<float> + pb_expand_kernel<float,false> + pb_reduce_kernel<float,...> (one SpMV = this "
                "launch group: the plan takes A's values again on every multiply)"),
            2: "spmv_rowblock_kernel<float,int,1024>"}.get(plan.get("alg"), "spmv_vector_kernel")
    return _emit(args, "csr_spmv_gflops", 2.0 * nnz, alg_bytes, elapsed, ms,
                 f"cfg2's matrix as a plain inspected csr_view: fp32 CSR SpMV {m}x{n}, 10 nnz/row uniform random, nnz={nnz}",
                 {"dtype": "f32", "rows": m, "nnz": nnz, "operand": "csr_view + multiply_inspect (no matrix_opt)", "plan": plan,
                  "values_taken_again_every_multiply": bool(si.get("refresh_each_call", 0)),
                  "plan_holds_no_values": vf, "plan_bytes": plan.get("device_bytes"),
                  "plan_bytes_over_matrix": (plan.get("device_bytes") or 0) / float(nnz * 8 + (m + 1) * 4),
                  "inspect_ms_untimed": inspect_ms, "kernel": kern}, None, parity=parity,
                 pmc_key="spmv_plain_cfg2" if (args.rows is None and vf) else None)


def _run_spmv_poisson(args, device, sp, oracle, generate):
    """cfg2 with Poisson(10) row lengths instead of exactly 10 ("row lengths = 10 exactly and a
    Poisson(10) variant (report both)"): the same columns / values distributions, matrix_opt operand like the headline."""
    m = n = args.rows or 10_000_000
    values, rowptr, colind, shape, nnz = generate.uniform_csr_device(m, n, 10, seed=0, device=device, poisson=True)
    a = sp.csr_view(values, rowptr, colind, shape, nnz)
    g = torch.Generator(device=device).manual_seed(11)
    x = torch.rand(n, device=device, generator=g)
    y = torch.empty(m, device=device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a_opt = sp.matrix_opt(a)
    info = sp.multiply_inspect(a_opt, x, y)
    torch.cuda.synchronize()
    inspect_ms = (time.perf_counter() - t0) * 1e3
    elapsed, ms = _time_steps(lambda: sp.multiply(info, a_opt, x, y), args.warmup, args.steps)
    plan = info.state_.info()
    if plan.get("alg") == 3 and hasattr(info.state_, "sliced_info"):
        plan["sliced"] = info.state_.sliced_info()
    y.fill_(float("nan"))
    sp.multiply(info, a_opt, x, y)
    torch.cuda.synchronize()
    alg_bytes = nnz * 8 + (m + 1) * 4 + (n + m) * 4
    rows = np.unique(np.concatenate([np.arange(0, min(m, 1500)), np.arange(max(0, m - 1500), m),
                                     np.random.default_rng(5).integers(0, m, 3000)]))
    sub_rp, sub_ci, sub_v = rows_subproblem(rows, rowptr, colind, values)
    xh = x.cpu().numpy()
    y_ref = oracle.spmv((len(rows), n), sub_rp, sub_ci, sub_v, xh)
    absrow = oracle.spmv_absrow(sub_rp, sub_ci, sub_v, xh)
    nbad, worst = parity_rows(y[torch.from_numpy(rows).to(device)].cpu().numpy(), y_ref, absrow.astype(np.float64), 1e-6,
                              float(np.finfo(np.float32).eps), np.diff(sub_rp))
    # every row: the fp64 checksum of y against the one computed from the entries (a dropped or doubled entry anywhere shows)
    chk = float(y.double().sum().item())
    ref_chk = float((values.double() * x[colind.long()].double()).sum().item())
    chk_ok = abs(chk - ref_chk) <= 1e-6 * abs(ref_chk)
    parity = {"status": "pass" if nbad == 0 and chk_ok else "fail", "rows": int(len(rows)), "rows_out_of_bound": nbad,
              "tol": 1e-6, "worst_err_over_rownorm": worst, "checksum_rel_err": abs(chk - ref_chk) / max(abs(ref_chk), 1e-300),
              "against": "oracle_spmv on the first / last 1 500 and 3 000 sampled rows; fp64 checksum of all of y against "
                         "sum(values * x[colind]) on the device"}
    return _emit(args, "csr_spmv_gflops", 2.0 * nnz, alg_bytes, elapsed, ms,
                 f"cfg2, Poisson variant: fp32 CSR SpMV {m}x{n}, Poisson(10) nnz/row, uniform random unsorted columns, nnz={nnz}",
                 {"dtype": "f32", "rows": m, "nnz": nnz, "operand": "matrix_opt(csr_view) + multiply_inspect", "plan": plan,
                  "plan_bytes": plan.get("device_bytes"), "inspect_ms_untimed": inspect_ms,
                  "kernel": "pb_expand_kernel<float,false,false> + pb_reduce_kernel<float,...>" if plan.get("alg") == 3
                  else "spmv_rowblock_kernel<float,int,1024>"}, None, parity=parity,
                 pmc_key="spmv_poisson_cfg2" if args.rows is None else None)


def _run_csc_spmv(args, device, sp, oracle, generate):
    """y = A^T x on cfg2's matrix handed over as a csc_view (= transposed(csr_view): the stored arrays are
    the CSR of the transpose; backend/algorithms.hpp:21-29, vendor/rocsparse/detail/get_transpose.hpp:19-29).  Two call
    shapes: un-inspected multiply(a_csc, x, y) -- the op = T scatter kernel on the caller's arrays -- and the inspected one
    (multiply_inspect materialises the operand row-major and plans it); the record's value is the inspected multiply, the
    un-inspected time, the inspect time and the bytes the plan holds are in `config`.  Parity: EVERY element of both y
    against oracle_spmv_csc on the host."""
    k = args.rows or 10_000_000                     # stored CSR: k x k with 10 entries per row; the operand is its transpose
    values, rowptr, colind, shape, nnz = generate.uniform_csr_device(k, k, 10, seed=0, device=device)
    a_csc = sp.csc_view(values, rowptr, colind, (k, k), nnz)
    g = torch.Generator(device=device).manual_seed(13)
    x = torch.rand(k, device=device, generator=g)
    y = torch.empty(k, device=device)
    # un-inspected
    el_t, ms_t = _time_steps(lambda: sp.multiply(a_csc, x, y), args.warmup, max(3, args.steps // 2))
    y.fill_(float("nan"))
    sp.multiply(a_csc, x, y)
    torch.cuda.synchronize()
    y_scatter = y.clone()
    # inspected
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = sp.multiply_inspect(a_csc, x, y)
    torch.cuda.synchronize()
    inspect_ms = (time.perf_counter() - t0) * 1e3
    elapsed, ms = _time_steps(lambda: sp.multiply(info, a_csc, x, y), args.warmup, args.steps)
    y.fill_(float("nan"))
    sp.multiply(info, a_csc, x, y)
    torch.cuda.synchronize()
    plan = info.state_.info()
    held = int(plan.get("device_bytes") or 0) + int(getattr(info.state_, "held_bytes", lambda: 0)())
    alg_bytes = nnz * 8 + (k + 1) * 4 + 2 * k * 4
    cpu, parity = None, None
    if not args.no_cpu_baseline:
        v, cp, ri, xh = values.cpu().numpy(), rowptr.cpu().numpy(), colind.cpu().numpy(), x.cpu().numpy()
        t0 = time.perf_counter()
        y_ref = oracle.spmv_csc((k, k), cp, ri, v, xh)
        dt = time.perf_counter() - t0
        cpu = {"value": 2.0 * nnz / dt / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "port", "seconds": dt,
               "sample": f"full workload ({nnz} nnz), 1 run of oracle_spmv_csc (column traversal, scatter into y)"}
        # sum_p |a_p x_p| per output element and the entries per element (for the k/2 * eps floor), on the device in fp64
        src_row = torch.repeat_interleave(torch.arange(k, device=device), (rowptr[1:] - rowptr[:-1]).long())
        absy = torch.zeros(k, dtype=torch.float64, device=device).index_add_(
            0, colind.long(), values.double().abs() * x.double()[src_row].abs())
        del src_row
        cnt = torch.bincount(colind.long(), minlength=k).cpu().numpy()
        absy = absy.cpu().numpy()
        eps = float(np.finfo(np.float32).eps)
        nb1, w1 = parity_rows(y.cpu().numpy(), y_ref, absy, 1e-6, eps, cnt)
        nb2, w2 = parity_rows(y_scatter.cpu().numpy(), y_ref, absy, 1e-6, eps, cnt)
        parity = {"status": "pass" if nb1 == 0 and nb2 == 0 else "fail", "rows": int(k), "rows_out_of_bound": nb1 + nb2,
                  "tol": 1e-6, "worst_err_over_rownorm": max(w1, w2),
                  "against": "oracle_spmv_csc (CPU restatement of backend/algorithms.hpp:21-29 + multiply_impl.hpp:33-53), every "
                             "element, both the inspected and the un-inspected multiply"}
    return _emit(args, "csc_spmv_gflops", 2.0 * nnz, alg_bytes, elapsed, ms,
                 f"8f: fp32 SpMV y = A^T x, csc_view of cfg2's matrix ({k}x{k}, 10 entries per stored row, nnz={nnz}); "
                 "timed step = multiply(info, csc_view, x, y) after multiply_inspect",
                 {"dtype": "f32", "rows": k, "nnz": nnz, "operand": "csc_view + multiply_inspect", "plan": plan,
                  "uninspected_ms_per_step": el_t / max(3, args.steps // 2) * 1e3,
                  "uninspected_roofline_frac": alg_bytes / (ms_t[0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "uninspected_traffic": (lambda tr: None if args.rows or tr[0] is None else
                                          {"bytes": tr[0], "over_algorithmic": tr[0] / alg_bytes, "profile": (tr[1] or {}).get("profile"),
                                           "stale": (tr[1] or {}).get("stale")})(read_pmc_traffic("csc_spmv_uninspected_8f")),
                  "uninspected_kernel": "t2_hist_kernel + scan + t2_plan_kernel + t2_scatter_kernel<float,int> + t2_accumulate_kernel<float> (round 6: products "
                                        "binned by column slice through the handle's workspace; SPBLAS_GFX950_SPMV_T2=0: the float-atomic scatter)",
                  "inspect_ms_untimed": inspect_ms, "plan_bytes": held,
                  "plan_bytes_over_matrix": held / float(nnz * 8 + (k + 1) * 4),
                  "plan_detached_from_materialised_arrays": bool(getattr(info.state_, "detached", False)),
                  "kernel": "device transpose at inspect (spt_* kernels), then the CSR plan's kernels (pb_expand_kernel + pb_reduce_kernel); "
                            "the materialised row-major arrays are released once the plan is built (round 6)"},
                 cpu, parity=parity, pmc_key=None if args.rows else "csc_spmv_8f")


def _run_spgemm4(args, device, sp, oracle, generate):
    """C = alpha * A B + beta * D (vendor/rocsparse/multiply_spgemm.hpp:237-250) at cfg5's size with an
    addend of 16 entries per row; timed step = one multiply_fill with the addend, multiply_compute reported beside it."""
    m = args.rows or 1_000_000
    av, ar, ac, ash, annz = generate.uniform_csr_device(m, m, 16, seed=0, device=device)
    bv, br, bc, bsh, bnnz = generate.uniform_csr_device(m, m, 16, seed=1, device=device)
    dv, dr, dc, dsh, dnnz = generate.uniform_csr_device(m, m, 16, seed=2, device=device)
    # the addend's rows with ascending columns (the reference's operands come from sorted generators, test/gtest/util.hpp)
    key = torch.repeat_interleave(torch.arange(m, device=device), 16) * m + dc.long()
    order = torch.argsort(key)
    dc, dv = dc[order].contiguous(), dv[order].contiguous()
    del key, order
    a, b, d = sp.csr_view(av, ar, ac, ash, annz), sp.csr_view(bv, br, bc, bsh, bnnz), sp.csr_view(dv, dr, dc, dsh, dnnz)
    alpha, beta = 1.5, -0.5
    a_s, d_s = sp.scaled(alpha, a), sp.scaled(beta, d)
    c_rp = torch.zeros(m + 1, dtype=torch.int32, device=device)
    c = sp.csr_view(None, c_rp, None, (m, m), 0)
    state = sp.spgemm_state_t()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sp.multiply_compute(state, a_s, b, c, d_s)
    torch.cuda.synchronize()
    compute_first_ms = (time.perf_counter() - t0) * 1e3
    cn = state.result_nnz()
    c.update(torch.empty(cn, device=device), c_rp, torch.empty(cn, dtype=torch.int32, device=device), (m, m), cn)
    products = int((br.long()[ac.long() + 1] - br.long()[ac.long()]).sum().item())
    os.environ["SPBLAS_GFX950_SPGEMM_REUSE"] = "0"
    try:
        elapsed, ms = _time_steps(lambda: sp.multiply_fill(state, a_s, b, c, d_s), args.warmup, args.steps)
    finally:
        del os.environ["SPBLAS_GFX950_SPGEMM_REUSE"]
    torch.cuda.synchronize()
    got_ci, got_v = c.colind().clone(), c.values().clone()
    compute_ms = float("inf")
    for _ in range(3):
        c2_rp = torch.zeros(m + 1, dtype=torch.int32, device=device)
        c2 = sp.csr_view(None, c2_rp, None, (m, m), 0)
        st2 = sp.spgemm_state_t()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sp.multiply_compute(st2, a_s, b, c2, d_s)
        torch.cuda.synchronize()
        compute_ms = min(compute_ms, (time.perf_counter() - t0) * 1e3)
        assert st2.result_nnz() == cn and torch.equal(c2_rp, c_rp)
        del st2
    alg_bytes = 3 * (annz * 8 + (m + 1) * 4) + cn * 8 + (m + 1) * 4
    cpu, parity = None, None
    if not args.no_cpu_baseline:
        a_h, b_h, d_h = (ar.cpu().numpy(), ac.cpu().numpy()), (br.cpu().numpy(), bc.cpu().numpy()), (dr.cpu().numpy(), dc.cpu().numpy())
        n_ref, row_nnz = oracle.spgemm_symbolic_d((m, m), a_h[0], a_h[1], (m, m), b_h[0], b_h[1], (m, m), d_h[0], d_h[1])
        rowptr_ok = bool(cn == n_ref and np.array_equal(c_rp.cpu().numpy().astype(np.int64),
                                                        np.concatenate([[0], np.cumsum(row_nnz)])))
        rows = np.unique(np.concatenate([np.arange(0, m, 1009), [m - 1]]))
        sa_rp, sa_c, sa_v = rows_subproblem(rows, ar, ac, av)
        sd_rp, sd_c, sd_v = rows_subproblem(rows, dr, dc, dv)
        bv_h = bv.cpu().numpy()
        sub = (len(rows), m)
        n_sub, _ = oracle.spgemm_symbolic_d(sub, sa_rp, sa_c, (m, m), b_h[0], b_h[1], sub, sd_rp, sd_c)
        t0 = time.perf_counter()
        cr, cc, cv = oracle.spgemm_numeric_d(sub, sa_rp, sa_c, sa_v, (m, m), b_h[0], b_h[1], bv_h, sub, sd_rp, sd_c, sd_v, n_sub,
                                             alpha, beta)
        dt = time.perf_counter() - t0
        _, _, cabs = oracle.spgemm_numeric_d(sub, sa_rp, sa_c, np.abs(sa_v), (m, m), b_h[0], b_h[1], np.abs(bv_h), sub, sd_rp, sd_c,
                                             np.abs(sd_v), n_sub, abs(alpha), abs(beta))
        sub_products = int((br.long()[torch.from_numpy(sa_c).to(device).long() + 1]
                            - br.long()[torch.from_numpy(sa_c).to(device).long()]).sum().item())
        cpu = {"value": 2.0 * sub_products / dt / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "port",
               "sample": f"every 1 009th row of A and D ({len(rows)} rows) x full B, oracle_spgemm_numeric_d (numeric timed)"}
        g_rp, g_c, g_v = rows_subproblem(rows, c_rp, got_ci, got_v)
        cols_ok = bool(np.array_equal(g_rp, cr) and np.array_equal(g_c, cc))
        nbad, worst = (1, float("inf"))
        if cols_ok:
            nbad, worst = parity_rows(g_v, cv, cabs.astype(np.float64), 1e-6, float(np.finfo(np.float32).eps), np.zeros(len(cv)))
        ok = rowptr_ok and cols_ok and nbad == 0 and bool(torch.isfinite(got_v).all())
        parity = {"status": "pass" if ok else "fail", "nnz_c": int(cn), "nnz_c_oracle": int(n_ref), "rows": int(m),
                  "rowptr_exact_all_rows": rowptr_ok, "sampled_rows": int(len(rows)), "sampled_colind_exact": cols_ok,
                  "values_out_of_bound": nbad, "tol": 1e-6, "worst_err_over_norm": worst,
                  "against": "oracle_spgemm_symbolic_d over all rows; oracle_spgemm_numeric_d (alpha*A*B + beta*D, "
                             "multiply_spgemm.hpp:118-214) on every 1 009th row: sorted columns exact, values norm-wise"}
    return _emit(args, "csr_spgemm4_fill_gflops", 2.0 * products, alg_bytes, elapsed, ms,
                 f"8f: fp32 C = alpha*A*B + beta*D, {m}x{m}, 16 nnz/row each, uniform random; timed step = one multiply_fill "
                 "with the addend, after multiply_compute",
                 {"dtype": "f32", "rows": m, "products": products, "nnz_c": cn, "alpha": alpha, "beta": beta,
                  "state": state.info(), "multiply_compute_ms_untimed": compute_ms,
                  "multiply_compute_first_call_ms": compute_first_ms,
                  "compute_plus_one_shot_fill_ms": compute_ms + elapsed / args.steps * 1e3,
                  "kernel": "spg_pack_b_kernel + spg_direct_kernel<float,...> with the addend merged (one fill = this launch group)"},
                 cpu, parity=parity, pmc_key=None if args.rows else "spgemm4_8f")


def _run_rmat_shards(args, device, sp, oracle, generate):
    """BASELINE cfg4 as the 8-GPU run will see it, measured on ONE GPU: the eight nnz-prefix row shards of the R-MAT scale-24
    graph (sharded.partition_rows_by_nnz, exactly what bench.py --gpus 8 --workload spmv_rmat gives rank r), each inspected
    and timed one after another.  The slowest shard's local step IS the critical path of the 8-GPU step before the gather;
    max / mean says how well the nnz-prefix split balances TIME (it balances entries by construction).  value = the whole
    matrix's flops over the slowest shard's step; parity: every row of the eight local results against oracle_spmv."""
    from spblas_reference_amd import sharded
    world = 8
    scale = 24 if args.rows is None else int(np.log2(args.rows))
    values, rowptr, colind, shape, nnz = generate.rmat_csr_device(scale, 16, dtype=torch.float64, seed=0, device=device)
    m = n = shape[0]
    bounds = sharded.partition_rows_by_nnz(rowptr, world)
    g = torch.Generator(device=device).manual_seed(7)
    x = torch.rand(n, dtype=torch.float64, device=device, generator=g)
    y_all = torch.full((m,), float("nan"), dtype=torch.float64, device=device)
    per = []
    wall = 0.0
    for r in range(world):
        lo, hi = bounds[r], bounds[r + 1]
        a = sharded.shard_csr(values, rowptr, colind, shape, lo, hi)
        y = y_all[lo:hi]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        algs = {"auto": sp._capi.SPMV_AUTO, "vector": sp._capi.SPMV_VECTOR, "rowblock": sp._capi.SPMV_ROWBLOCK,
                "sliced": sp._capi.SPMV_SLICED}
        info = sp.multiply_inspect(sp.matrix_opt(a), x, y, alg=algs.get(getattr(args, "alg", "auto"), sp._capi.SPMV_AUTO))
        torch.cuda.synchronize()
        inspect_ms = (time.perf_counter() - t0) * 1e3
        elapsed, ms = _time_steps(lambda: sp.multiply(info, a, x, y), args.warmup, args.steps)
        wall += elapsed
        plan = info.state_.info()
        y.fill_(float("nan"))
        sp.multiply(info, a, x, y)
        torch.cuda.synchronize()
        per.append({"rank": r, "rows": int(hi - lo), "nnz": int(a.size()), "ms": elapsed / args.steps * 1e3, "event_ms": ms[0],
                    "plan_alg": plan.get("alg"), "plan_bytes": plan.get("device_bytes"), "inspect_ms": inspect_ms,
                    "alg_bytes": int(a.size()) * 12 + (hi - lo + 1) * 4 + (n + hi - lo) * 8})
        del info, a
        torch.cuda.empty_cache()
    t = [p["ms"] for p in per]
    worst = max(range(world), key=lambda r: t[r])
    cpu, parity = None, None
    if not args.no_cpu_baseline:
        v, rp, ci, xh = values.cpu().numpy(), rowptr.cpu().numpy(), colind.cpu().numpy(), x.cpu().numpy()
        try:
            oracle.load(native=True)
            native = True
        except Exception:
            native = False
        t0 = time.perf_counter()
        y_ref = oracle.spmv(shape, rp, ci, v, xh, native=native)
        dt = time.perf_counter() - t0
        absrow = oracle.spmv_absrow(rp, ci, v, xh)
        cpu = {"value": 2.0 * nnz / dt / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "port", "seconds": dt,
               "sample": f"full workload ({nnz} nnz), 1 run of oracle_spmv over all rows (the eight shards together)"}
        nbad, w = parity_rows(y_all.cpu().numpy(), y_ref, absrow.astype(np.float64), 1e-12, float(np.finfo(np.float64).eps), np.diff(rp))
        parity = {"status": "pass" if nbad == 0 else "fail", "rows": int(m), "rows_out_of_bound": nbad, "tol": 1e-12,
                  "worst_err_over_rownorm": w, "against": "oracle_spmv, every row of the eight shards' local results"}
    # the record's step = the slowest shard (what an 8-GPU step waits for); roofline on that shard's own algorithmic bytes
    out = _emit(args, "csr_spmv_gflops", 2.0 * nnz, per[worst]["alg_bytes"], t[worst] * 1e-3 * args.steps, [per[worst]["event_ms"], per[worst]["event_ms"]],
                f"cfg4 as eight nnz-prefix row shards (the local steps of an 8-GPU run), one after another on one GPU: fp64 "
                f"R-MAT scale {scale}, nnz={nnz}; step = the SLOWEST shard's local SpMV",
                {"dtype": "f64", "rows": m, "nnz": nnz, "operand": "matrix_opt(csr_view of the shard) + multiply_inspect",
                 "shard_ms": [round(v_, 5) for v_ in t], "shard_rows": [p["rows"] for p in per], "shard_nnz": [p["nnz"] for p in per],
                 "shard_plan_alg": [p["plan_alg"] for p in per], "shard_inspect_ms": [round(p["inspect_ms"], 2) for p in per],
                 "max_ms": max(t), "mean_ms": sum(t) / world, "max_over_mean": max(t) / (sum(t) / world),
                 "slowest_rank": worst, "sum_ms": sum(t),
                 "speedup_bound_vs_one_gpu_note": "one-GPU cfg4 step / max_ms bounds the 8-GPU speed-up before the all-gather",
                 "kernel": "the shard's plan (see shard_plan_alg: 3 = SLICED tiles, 2 = row blocks)"}, cpu, parity=parity)
    return out


def _run(args, device):
    import spblas_reference_amd as sp
    from oracle import oracle
    from spblas_reference_amd import generate

    if args.workload == "spmv_plain":
        return _run_spmv_plain(args, device, sp, oracle, generate)
    if args.workload == "spmv_poisson1":
        return _run_spmv_poisson(args, device, sp, oracle, generate)
    if args.workload == "spmv_rmat1":
        return _run_spmv_rmat1(args, device, sp, oracle, generate)
    if args.workload == "spmv_rmat_shards":
        return _run_rmat_shards(args, device, sp, oracle, generate)
    if args.workload == "csc_spmv":
        return _run_csc_spmv(args, device, sp, oracle, generate)
    if args.workload == "spgemm4":
        return _run_spgemm4(args, device, sp, oracle, generate)

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
                   "sample": f"first {rows} rows ({int(rp[-1])} nnz x {ncols} columns) of the same A and B, 1 run of oracle_spmm",
                   # (round-5 review: say it in the record)
                   "note": "NOT the reference as written: the C restatement's inner loop runs over the n columns of one B row and "
                           "the compiler vectorises it; the reference looks every element up through its mdspan accessors "
                           "(multiply_impl.hpp:85-91) and was measured at 0.41 GFLOP/s (BASELINE.md section 2) -- as a timing proxy "
                           "this port flatters the CPU by 10 - 75x; the values are the reference's"}
        mi = info.state_.spmm_info()
        # parity (outside the timed region): the first and last 1 500 rows and 1 500 sampled rows of the timed C against
        # oracle_spmm on the compacted sub-problem (tests/test_gpu_configs.py does the same), plus the fp64 column checksum
        rows = np.unique(np.concatenate([np.arange(1500), np.arange(m - 1500, m),
                                         np.random.default_rng(0).integers(0, m, 1500)]))
        sub_rp, sub_c, sub_v = rows_subproblem(rows, rowptr, colind, values)
        uniq, inv = np.unique(sub_c, return_inverse=True)
        inv = inv.astype(np.int32)
        B_sub = B[torch.from_numpy(uniq.astype(np.int64)).to(device)].cpu().numpy()
        C_ref = oracle.spmm((len(rows), len(uniq)), sub_rp, inv, sub_v, B_sub)
        C_abs = oracle.spmm((len(rows), len(uniq)), sub_rp, inv, np.abs(sub_v), np.abs(B_sub)).astype(np.float64)
        got = C[torch.from_numpy(rows.astype(np.int64)).to(device)].cpu().numpy()
        nbad, worst = parity_rows(got, C_ref, C_abs, 1e-6, float(np.finfo(np.float32).eps), np.diff(sub_rp))
        colsum = C.double().sum(0)
        w = torch.zeros(shape[1], dtype=torch.float64, device=device).index_add_(0, colind.long(), values.double())
        ref = (w[:, None] * B.double()).sum(0)
        chk = bool(((colsum - ref).abs() <= 1e-6 * ref.abs()).all())
        del w, ref, colsum
        parity = {"status": "pass" if (nbad == 0 and chk) else "fail", "rows": int(len(rows)), "elements_out_of_bound": nbad,
                  "tol": 1e-6, "worst_err_over_rownorm": worst, "column_checksum_fp64": "pass" if chk else "fail",
                  "against": "oracle_spmm (CPU restatement of multiply_impl.hpp:66-92) on the first / last 1 500 and 1 500 "
                             "sampled rows; every column's sum against an fp64 evaluation on the device"}
        return _emit(args, "csr_spmm_gflops", 2.0 * nnz * ncols, alg_bytes, elapsed, ms,
              (f"R-MAT variant of cfg3: fp32 CSR x dense SpMM, A R-MAT scale 21 ({m}x{m}, {nnz} entries, duplicates kept), "
               f"B {m}x{ncols} row-major" if rmat else
               f"banded variant of cfg3: fp32 CSR x dense SpMM, A {m}x{m} 64 nnz/row within 48 columns of the diagonal, "
               f"B {m}x{ncols} row-major" if banded else
               f"cfg3: fp32 CSR x dense SpMM, A {m}x{m} 32 nnz/row uniform random, B {m}x{ncols} row-major"),
              {"dtype": "f32", "rows": m, "nnz": nnz, "ncols": ncols, "spmm_inspect": mi,
               # which kernel owns the qualifying row blocks (csrc/spmm.hip): round 6's default is the band kernel -- LDS-staged B
               # window, vector FMAs over the STORED entries; the matrix-core forms (tile kernel of rounds 3 - 5:
               # SPBLAS_GFX950_SPMM_BAND=0; dense-window kernel: SPBLAS_GFX950_SPMM_BAND_DENSE=250) were measured slower and are opt-in
               "kernel": (("spmm_panel_kernel<int> (v_mfma_f32_32x32x2_f32)" if os.environ.get("SPBLAS_GFX950_SPMM_BAND") == "0" else
                           "spmm_band_mfma_kernel<int> (v_mfma_f32_16x16x4_f32)" if int(os.environ.get("SPBLAS_GFX950_SPMM_BAND_DENSE", "1001")) <= 1000 else
                           "spmm_band_kernel<int,8> (LDS-staged B window, v_pk_fma_f32 over the stored entries; no matrix-core work)")
                          if mi["panel_blocks"] > 0 else "spmm_rowgroup_kernel<float,int,4>"),
               "matrix_core_alternatives_ms": ({"note": "same workload, measured in round 6 (profiles/r06_spmm_band.md)",
                                                "spmm_panel_kernel (tiles, 32x32x2)": 2.37, "spmm_band_mfma_kernel (window, 16x16x4)": 2.38}
                                               if banded and not args.rows else None)}, cpu, parity=parity,
                     pmc_key=None if (rmat or args.rows) else ("spmm_banded" if banded else "spmm_cfg3"),
                     mfma_key=(None if not (banded and not args.rows and mi["panel_blocks"] > 0) else
                               "spmm_banded_mfma" if os.environ.get("SPBLAS_GFX950_SPMM_BAND") == "0" else
                               "spmm_banded_mfma_window" if int(os.environ.get("SPBLAS_GFX950_SPMM_BAND_DENSE", "1001")) <= 1000 else
                               "none"))

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
        torch.cuda.synchronize()
        one_shot = (c.colind().clone(), c.values().clone())  # what the timed steps wrote: checked below
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
        # (best of three: the call is host-driven -- a dozen launches and two waits -- and the figure is taken right after other
        # workloads' CPU baselines, whose OpenMP threads may still be spinning on the host's cores)
        compute_warm_ms = float("inf")
        for _ in range(3):
            c2_rp = torch.zeros(m + 1, dtype=torch.int32, device=device)
            c2 = sp.csr_view(None, c2_rp, None, (m, m), 0)
            state2 = sp.spgemm_state_t()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sp.multiply_compute(state2, a, b, c2)
            torch.cuda.synchronize()
            compute_warm_ms = min(compute_warm_ms, (time.perf_counter() - t0) * 1e3)
            assert state2.result_nnz() == cn and torch.equal(c2_rp, c_rp)
            del state2
        cpu = None
        if not args.no_cpu_baseline:
            rows = min(20_000, m)
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
        parity = _spgemm_parity(oracle, m, (ar, ac, av), (br, bc, bv), c_rp, c, cn, one_shot)
        return _emit(args, "csr_spgemm_fill_gflops", 2.0 * products, alg_bytes, elapsed, ms,
              f"cfg5: fp32 CSR x CSR SpGEMM {m}x{m}, 16 nnz/row uniform random; timed step = one-shot multiply_fill "
              "(sorted columns and values written; at this size every row is sorted by the direct kernel, none hashed), "
              "after multiply_compute",
              {"dtype": "f32", "rows": m, "products": products, "nnz_c": cn, "state": state.info(),
               "multiply_compute_ms_untimed": compute_warm_ms,
               "multiply_compute_first_call_ms": compute_ms,
               "compute_plus_one_shot_fill_ms": compute_warm_ms + elapsed / args.steps * 1e3,
               "repeated_fills": {"note": "numeric reuse on one symbolic result (accumulation by recorded product ranks, "
                                          "columns kept): secondary figure, not the metric value",
                                  "first_fill_ms_untimed": fills[0], "second_fill_ms_untimed_records_ranks": fills[1],
                                  "ms_per_fill": reuse_step * 1e3, "gflops": 2.0 * products / reuse_step / 1e9,
                                  "algorithmic_bytes": reuse_bytes,
                                  "roofline_frac": reuse_bytes / (reuse_ms[0] * 1e-3) / 1e9 / 8000.0,
                                  "kernel": "spg_ranked_fill_kernel<float,16,256,4,false> (+ spg_rank_record_kernel<64,256> once)"},
               "kernel": "spg_pack_b_kernel + spg_direct_kernel<float,true,false> + <float,true,true> (rows with shared columns); "
                         "one fill = this launch group"}, cpu, parity=parity,
                     pmc_key=None if args.rows else "spgemm_cfg5")

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
            rows = min(200_000, m)
            ra, rb = ar[:rows + 1].cpu().numpy(), br[:rows + 1].cpu().numpy()
            t0 = time.perf_counter()
            oracle.add((rows, m), ra, ac[:ra[-1]].cpu().numpy(), av[:ra[-1]].cpu().numpy(), (rows, m), rb,
                       bc[:rb[-1]].cpu().numpy(), bv[:rb[-1]].cpu().numpy())
            dt = time.perf_counter() - t0
            cpu = {"value": (ra[-1] + rb[-1]) / dt / 1e9, "unit": "Gentries/s", "cores": 1, "kind": "port",
                   "sample": f"first {rows} rows, oracle_add_f32 (SPA + sort per row)"}
        # parity: every row against the oracle (structure exact, values norm-wise)
        torch.cuda.synchronize()
        r_rp, r_ci, r_v = oracle.add((m, m), ar.cpu().numpy(), ac.cpu().numpy(), av.cpu().numpy(), (m, m), br.cpu().numpy(),
                                     bc.cpu().numpy(), bv.cpu().numpy())
        _, _, r_abs = oracle.add((m, m), ar.cpu().numpy(), ac.cpu().numpy(), np.abs(av.cpu().numpy()), (m, m),
                                 br.cpu().numpy(), bc.cpu().numpy(), np.abs(bv.cpu().numpy()))
        g_rp, g_ci, g_v = c_rp.cpu().numpy(), c.colind().cpu().numpy(), c.values().cpu().numpy()
        struct_ok = bool(cn == int(r_rp[-1]) and np.array_equal(g_rp, r_rp) and np.array_equal(g_ci, r_ci[:cn]))
        n_bad, worst = (parity_rows(g_v, r_v[:cn], r_abs[:cn].astype(np.float64), 1e-6, float(np.finfo(np.float32).eps),
                                    np.full(cn, 2)) if struct_ok else (cn, float("inf")))
        parity = {"status": "pass" if struct_ok and n_bad == 0 else "fail", "rows": m, "rowptr_colind_exact": struct_ok,
                  "values_out_of_bound": int(n_bad), "tol": 1e-6, "worst_err_over_norm": float(worst),
                  "against": "oracle_add (CPU restatement of add_impl.hpp:40-77), every row"}
        return _emit(args, "csr_add_gentries", float(annz + bnnz), alg_bytes, elapsed, ms,
              f"8f: fp32 CSR + CSR add {m}x{m}, 16 nnz/row each, uniform random; timed step = add_compute; value = input entries/ns",
              {"dtype": "f32", "unit": "Gentries/s", "rows": m, "nnz_c": cn, "add_inspect_ms_untimed": inspect_ms,
               "kernel": "spg_ranked_fill_kernel / spg_hash_kernel (identity B + addend)"}, cpu, parity=parity,
                     pmc_key=None if args.rows else "add_8f")

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
            rows = min(1_000_000, m)
            rp = ar[:rows + 1].cpu().numpy()
            t0 = time.perf_counter()
            oracle.transpose((rows, m), rp, ac[:rp[-1]].cpu().numpy(), av[:rp[-1]].cpu().numpy())
            dt = time.perf_counter() - t0
            cpu = {"value": rp[-1] / dt / 1e9, "unit": "Gentries/s", "cores": 1, "kind": "port",
                   "sample": f"first {rows} rows, oracle_transpose_f32"}
        # parity: bit-exact against the oracle's stable counting sort, every entry
        torch.cuda.synchronize()
        r_rp, r_ci, r_v = oracle.transpose((m, m), ar.cpu().numpy(), ac.cpu().numpy(), av.cpu().numpy())
        ok = bool(np.array_equal(t_rp.cpu().numpy(), r_rp) and np.array_equal(t_ci.cpu().numpy(), r_ci) and
                  np.array_equal(t_v.cpu().numpy().view(np.uint32), r_v.view(np.uint32)))
        parity = {"status": "pass" if ok else "fail", "entries": annz, "bit_exact": ok,
                  "against": "oracle_transpose (CPU restatement of transpose_impl.hpp:14-53), every row offset, column and value bit"}
        return _emit(args, "csr_transpose_gentries", float(annz), alg_bytes, elapsed, ms,
              f"8f: fp32 CSR transpose {m}x{m}, 10 nnz/row uniform random; value = entries/ns",
              {"dtype": "f32", "unit": "Gentries/s", "rows": m, "nnz": annz,
               "kernel": "spt_count_kernel + spt_scatter_kernel, three 8-bit passes"}, cpu, parity=parity,
                     pmc_key=None if args.rows else "transpose_8f")

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
        # parity: row-wise backward error in fp64 on the device (every row) and forward error against the oracle's solve
        torch.cuda.synchronize()
        xd, vd, bd = x.double(), values.double(), b.double()
        rows_e = torch.arange(m, device=device).repeat_interleave(k + 1)
        tx = torch.zeros(m, dtype=torch.float64, device=device).index_add_(0, rows_e, vd * xd[colind.long()])
        ab = torch.zeros(m, dtype=torch.float64, device=device).index_add_(0, rows_e, vd.abs() * xd[colind.long()].abs())
        tol_r = max(1e-6, 0.5 * (k + 3) * float(np.finfo(np.float32).eps))
        ok_r = (tx - bd).abs() <= tol_r * (bd.abs() + ab)
        ok_r[0] = True  # row 0: its eight sub-diagonal draws land ON the diagonal (col <= row), which L x above adds up while
        bad_r = int((~ok_r).sum().item())  # the solve takes one diagonal entry; the oracle comparison below covers the row
        x_ref = oracle.triangular_solve((m, m), rp.int().cpu().numpy(), colind.cpu().numpy(), values.cpu().numpy(),
                                        b.cpu().numpy()).astype(np.float64)
        scale = np.maximum(np.abs(x_ref), np.abs(x_ref).max() * 1e-3 + 1e-30)
        ferr = float((np.abs(x.cpu().numpy().astype(np.float64) - x_ref) / scale).max())
        parity = {"status": "pass" if bad_r == 0 and ferr <= 1e-4 else "fail", "rows": m, "rows_out_of_residual_bound": bad_r,
                  "residual_tol": tol_r, "max_forward_error_vs_oracle": ferr, "forward_tol": 1e-4,
                  "against": "row-wise |b - Lx| <= tol (|b| + |L||x|) in fp64 on the device, rows 1..m-1; x (every row) against oracle_trsv "
                             "(CPU restatement of triangular_solve_impl.hpp:41-94) at 100x the tolerance (conditioning), "
                             "as tests/test_gpu_sptrsv.py"}
        del xd, vd, bd, rows_e, tx, ab
        return _emit(args, "csr_sptrsv_gflops", 2.0 * nnz, alg_bytes, elapsed, ms,
              f"8f: fp32 lower-triangular solve {m}x{m}, {k} random sub-diagonal entries per row + diagonal",
              {"dtype": "f32", "rows": m, "nnz": nnz, "plan": info.state_.info(),
               "triangular_solve_inspect_ms_untimed": inspect_ms,
               "triangular_solve_inspect_first_call_ms": inspect_first_ms,
               "kernel": "sptrsv cooperative level kernel"}, cpu, parity=parity,
                     pmc_key=None if args.rows else "sptrsv_8f")

    raise SystemExit(f"unknown workload {args.workload}")
