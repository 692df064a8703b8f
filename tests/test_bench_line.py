"""CPU: the ONE stdout line of bench.py stays small enough for the driver to parse (round 5's 21 KB line was cut off in the
driver's record and the headline went unmeasured).  Canned input: the complete record of round 5's default run
(tests/golden/bench_full_r05.json -- our own bench output, kept as data)."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench_line import LIMIT, compact, secondary_stderr_line, summarize_secondary  # noqa: E402


def _full():
    with open(os.path.join(ROOT, "tests", "golden", "bench_full_r05.json")) as f:
        return json.load(f)


def test_headline_line_is_small_and_round_trips():
    full = _full()
    assert len(json.dumps(full)) > 15000          # the shape that broke the driver's parser
    head = compact(full, detail_file="bench_secondary.json")
    line = json.dumps(head)
    assert len(line) + 1 < LIMIT == 4096
    assert "\n" not in line and json.loads(line) == head
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_check"):
        assert k in head, k
    assert head["config"]["workload"].startswith("cfg2") and head["config"]["nnz"] == 100000000
    r = head["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert r["traffic"] and r["traffic_source"]["stale"] is False and r["kernel_avg_ms"] > 0
    assert abs(head["ms_per_step"] - full["ms_per_step"]) < 1e-5 * full["ms_per_step"]
    c = head["cpu_baseline"]
    assert c["cores"] == 1 and c["kind"] == "port" and c["cpu_model"] and c["all_cores"] == 256
    assert head["parity"].startswith("pass: 10000000 rows")


def test_every_secondary_record_has_a_short_summary():
    full = _full()
    head = compact(full)
    assert set(head["secondary_summary"]) == set(full["secondary"])
    for name, rec in full["secondary"].items():
        s = summarize_secondary(rec)
        assert len(json.dumps(s)) <= 120, (name, s)
        assert s["parity"] == rec["parity_check"] and s["ms"] > 0 and 0 < s["frac"] < 1
        assert name in secondary_stderr_line(name, rec)
    assert head["secondary_summary"]["spmm_banded"]["mfma"] > 0
    broken = {"workload": "x", "parity_check": "fail", "error": "RuntimeError: " + "y" * 500}
    assert len(json.dumps(summarize_secondary(broken))) <= 120


def test_line_stays_under_the_limit_whatever_is_added():
    """twenty more secondary records, a long failure text, an eight-rank multi_gpu object: still one line under 4 KB."""
    full = _full()
    for i in range(20):
        full["secondary"][f"extra_{i}"] = copy.deepcopy(full["secondary"]["cfg4"])
    full["n_gpus"] = 8
    full["multi_gpu"] = {"mode_timed": "plain", "path_used": "rccl", "local_spmv_ms": 0.05, "rccl_nranks": 8, "backend": "nccl",
                         "chunked_step_ms": None, "gather_ms": 0.04, "rccl_step_ms": 0.1, "fused_step_ms": None,
                         "fused_check": False, "fused_failure": "z" * 3000, "link_gbs_estimate": None, "fused_post_check": None,
                         "gather": "inplace", "rows_per_rank": [1250000] * 8, "note": "n" * 400,
                         "fused_pipelined_step_ms": None, "expand_wait_us": 1.0}
    line = json.dumps(compact(full, detail_file="bench_secondary_n8.json"))
    assert len(line) + 1 < LIMIT
    head = json.loads(line)
    assert head["value"] == compact(full)["value"] and head["roofline"]["frac"] > 0 and head["cpu_baseline"]["cores"] == 1
    mg = head.get("multi_gpu")
    if mg is not None:  # (dropped last, only if nothing else made room)
        assert mg["rccl_nranks"] == 8 and mg["path_used"] == "rccl" and len(mg["fused_failure"]) <= 120


def test_multi_gpu_keys_of_the_compact_line():
    full = _full()
    full.pop("secondary")
    full["multi_gpu"] = {"mode_timed": "fused", "path_used": "fused", "local_spmv_ms": 0.05, "rccl_nranks": 8, "backend": "nccl",
                         "chunked_step_ms": 0.07, "gather_ms": 0.04, "rccl_step_ms": 0.1, "fused_step_ms": 0.08,
                         "fused_check": True, "fused_failure": None, "link_gbs_estimate": 120.0, "fused_post_check": True,
                         "gather": "inplace", "rows_per_rank": [1250000] * 8, "note": "n" * 400}
    mg = compact(full)["multi_gpu"]
    for k in ("backend", "rccl_nranks", "path_used", "rccl_step_ms", "fused_step_ms", "chunked_step_ms", "fused_check",
              "fused_failure", "link_gbs_estimate"):
        assert k in mg, k
    assert "note" not in mg and "secondary_summary" not in compact(full)
