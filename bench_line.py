"""The ONE stdout line of bench.py, kept small enough for the driver to parse (round 5's line had grown to 21 KB and was
cut off in the driver's record).  compact(full) maps the full result -- headline + every secondary record, which goes to
bench_secondary.json untouched -- onto the headline object: the contract keys, a flat `config`, `roofline`, `cpu_baseline`,
a one-line `parity`, the eight `multi_gpu` keys of DESIGN.md section 3, and a `secondary_summary` of <= 120 bytes per record.
Pure Python (tests/test_bench_line.py runs it on canned values)."""
import json

LIMIT = 4096  # bytes of the line, newline included


def _r(v, sig=6):
    """floats to `sig` significant digits (a 17-digit repr per number is a third of the old line)"""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float(f"{v:.{sig}g}")
    return v


def _pick(d, keys):
    return {k: _r(d.get(k)) for k in keys if d is not None and k in d}


def _parity_line(p):
    if not p:
        return None
    if p.get("status") == "skipped":
        return f"skipped: {p.get('why', '')}"[:160]
    n = p.get("rows", p.get("entries"))
    bad = p.get("rows_out_of_bound", p.get("elements_out_of_bound", p.get("values_out_of_bound")))
    worst = p.get("worst_err_over_rownorm", p.get("worst_err_over_norm"))
    s = f"{p.get('status')}: {n} rows"
    if bad is not None:
        s += f", {bad} out of bound"
    if p.get("tol") is not None:
        s += f", tol {p['tol']}"
    if isinstance(worst, float):
        s += f", worst {worst:.3g}"
    against = str(p.get("against", ""))
    return (s + (f"; vs {against}" if against else ""))[:200]


def summarize_secondary(rec):
    """<= 120 bytes: ms per step, fraction of the HBM roofline, measured traffic over algorithmic bytes, parity."""
    if "error" in rec:
        return {"parity": "fail", "error": str(rec["error"])[:60]}
    roof = rec.get("roofline") or {}
    out = {"ms": _r(rec.get("ms_per_step"), 4), "frac": _r(roof.get("frac"), 3)}
    tr, ab = roof.get("traffic"), roof.get("algorithmic_bytes_per_launch")
    if tr and ab:
        out["traffic_x"] = _r(tr / ab, 3)
    if roof.get("mfma_util") is not None:
        out["mfma"] = _r(roof["mfma_util"], 3)
    if (rec.get("detail") or rec.get("config") or {}).get("max_over_mean") is not None:  # cfg4_shards_of_8: balance of the shards
        out["max_over_mean"] = _r((rec.get("detail") or rec.get("config"))["max_over_mean"], 3)
    out["parity"] = rec.get("parity_check", "not run")
    return out


def secondary_stderr_line(name, rec):
    """`name ms frac traffic_ratio parity` -- one short line per secondary record for the stderr log"""
    s = summarize_secondary(rec)
    return (f"{name} ms={s.get('ms')} frac={s.get('frac')} traffic_x={s.get('traffic_x')} parity={s.get('parity')}"
            + (f" error={s['error']}" if "error" in s else ""))


MULTI_KEYS = ("backend", "rccl_nranks", "path_used", "mode_timed", "rccl_step_ms", "fused_step_ms", "chunked_step_ms",
              "fused_pipelined_step_ms", "fused_check", "fused_post_check", "fused_failure", "link_gbs_estimate",
              "local_spmv_ms", "gather_ms", "gather", "rows_per_rank")


def compact(full, detail_file=None):
    """The headline object for stdout.  `full` is bench.py's complete result (with `secondary` if it ran)."""
    cfg = full.get("config") or {}
    plan = cfg.get("plan") or {}
    sl = plan.get("sliced") or {}
    roof = full.get("roofline") or {}
    src = roof.get("traffic_source") or {}
    cpu = full.get("cpu_baseline")
    out = {k: _r(full.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                        "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {
        **_pick(cfg, ("workload", "rows", "cols", "nnz", "index_type", "parallelism", "operand", "value_contract", "alg")),
        "plan_alg": plan.get("alg"), "plan_value_free": sl.get("value_free"), "plan_hot_split": plan.get("hot_split"),
        "plan_bytes": cfg.get("plan_bytes"), "plan_bytes_over_matrix": _r(cfg.get("plan_bytes_over_matrix"), 4),
        "inspect_ms": _r(cfg.get("inspect_ms_untimed"), 4), "inspect_warm_ms": _r(cfg.get("inspect_warm_ms_untimed"), 4),
        "handle_create_ms": _r(cfg.get("handle_create_ms"), 4)}
    out["roofline"] = {
        **_pick(roof, ("bound", "achieved", "peak", "unit", "frac", "frac_of_achievable", "traffic")),
        "traffic_source": ({"profile": src.get("profile"), "stale": src.get("stale"),
                            "note": "PMC constant from the committed profile, not measured in this run"} if src else None),
        **_pick(roof, ("kernel", "algorithmic_bytes_per_launch", "kernel_avg_ms"))}
    if isinstance(out["roofline"].get("kernel"), str):
        out["roofline"]["kernel"] = out["roofline"]["kernel"][:140]
    if cpu:
        out["cpu_baseline"] = _pick(cpu, ("value", "unit", "cores", "kind", "cpu_model", "sample", "seconds",
                                          "all_cores_value", "all_cores"))
        if isinstance(out["cpu_baseline"].get("sample"), str):
            out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"][:140]
    else:
        out["cpu_baseline"] = None
    out["parity_check"] = full.get("parity_check")
    out["parity"] = _parity_line(full.get("parity"))
    mg = full.get("multi_gpu")
    out["multi_gpu"] = _pick(mg, MULTI_KEYS) if mg else None
    if out["multi_gpu"] and isinstance(out["multi_gpu"].get("fused_failure"), str):
        out["multi_gpu"]["fused_failure"] = out["multi_gpu"]["fused_failure"][:120]
    sec = full.get("secondary")
    if sec:
        out["secondary_summary"] = {name: summarize_secondary(rec) for name, rec in sec.items()}
    if detail_file:
        out["detail_file"] = detail_file
    # whatever a future record adds, the line stays under the limit: drop the optional parts, longest first
    for drop in (None, ("cpu_baseline", "sample"), ("roofline", "kernel"), ("config", "parallelism"), ("parity",),
                 ("secondary_summary",), ("multi_gpu",)):
        if drop:
            tgt = out
            for k in drop[:-1]:
                tgt = tgt.get(k) or {}
            tgt.pop(drop[-1], None)
        if len(json.dumps(out)) + 1 < LIMIT:
            break
    return out
