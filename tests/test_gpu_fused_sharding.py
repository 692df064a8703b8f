"""-m gpu: the fused all-gather of the row-sharded SpMV (hipIpc-mapped y copies + peer stores from the
reduce kernels + device-side step barrier), exercised with several processes on the one GPU of the test
box.  See tests/mp_fused_worker.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_full(line):
    """bench.py prints a compact headline (< 4 KB, bench_line.compact); the complete record -- plan, parity report, every
    multi_gpu figure -- is in the `detail_file` it names, written next to bench.py.  Returns that record after checking
    that the line is the small one and agrees with it."""
    import json
    head = json.loads(line)
    assert len(line) + 1 < 4096, len(line)
    with open(os.path.join(ROOT, head["detail_file"])) as f:
        full = json.load(f)
    assert abs(full["ms_per_step"] - head["ms_per_step"]) <= 1e-5 * full["ms_per_step"]
    assert full.get("parity_check") == head.get("parity_check")
    return full


@pytest.mark.parametrize("world,dt,stripes", [(2, "f32", 1), (4, "f32", 3), (2, "f64", 4)])
def test_fused_sharded_spmv_multiprocess(gpu, world, dt, stripes):
    """stripes > 1: the reduces of contiguous bin groups alternate between two streams (the groups need not
    divide the bins evenly).  FUSED_VFREE: the same exchange on a value-free local plan (round 6), values rewritten in place
    between steps."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", FUSED_STRIPES=str(stripes), FUSED_VFREE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29650 + world + (7 if dt == "f64" else 0)),
           os.path.join(ROOT, "tests", "mp_fused_worker.py"), dt]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "FUSED_OK" in r.stdout and "FUSED_VFREE_OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


@pytest.mark.parametrize("world,dt,chunks,tri", [(2, "f32", 4, 0), (4, "f32", 3, 0), (3, "f64", 2, 0), (8, "f32", 4, 0),
                                                 (4, "f32", 3, 1), (3, "f64", 2, 1)])
def test_fused_dependent_chain_with_chunk_flags(gpu, world, dt, chunks, tri):
    """The dependent iteration y_{k+1} = A y_k without a step barrier: the reduce publishes a flag per (rank, chunk) after
    the chunk's peer stores, the next expand waits per x slice for exactly the chunks it needs.  Bit-compared with the
    barrier chain on every rank, also with one rank's chunk 1 made 4 ms late in every step (tests/mp_fused_worker.py).
    tri = 1: a lower block-triangular matrix -- ranks that read none of a late peer's rows must still not run ahead of
    it (one expand workgroup waits for every peer's chunks, whatever the sparsity pattern)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", FUSED_STRIPES="1",
               FUSED_CHUNKS=str(chunks), FUSED_TRIANGULAR=str(tri))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29690 + world + (7 if dt == "f64" else 0) + 20 * tri),
           os.path.join(ROOT, "tests", "mp_fused_worker.py"), dt]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0 and "FUSED_OK" in r.stdout and f"chunks {chunks}" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


def test_fused_sharded_spmv_nccl_control_plane(gpu):
    """Same worker with RCCL as the control plane (what bench.py uses on a multi-GPU node); one process,
    because RCCL refuses two ranks on one device: agreement all-reduces on device tensors,
    all_gather_object and barriers under the nccl backend."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", FUSED_BACKEND="nccl",
               FUSED_STRIPES="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
           "127.0.0.1", "--master-port", "29671", os.path.join(ROOT, "tests", "mp_fused_worker.py"), "f32"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "FUSED_OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


@pytest.mark.parametrize("fused", ["auto", "off"])
def test_bench_multi_gpu_code_path_with_one_rank(gpu, fused):
    """bench.py's N > 1 branch (process group, sharded operators, try_fused selection, timed loop, JSON) run
    with a single rank under the nccl backend -- the closest this one-GPU box gets to the driver's
    `torch.distributed.run --nproc-per-node N bench.py --gpus N`."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
           "127.0.0.1", "--master-port", "29677" if fused == "auto" else "29678", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--debug-multi", "--fused", fused, "--rows", "2000000", "--steps", "5", "--warmup", "2",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    out = _bench_full(lines[0])
    assert out["value"] > 0 and out["steps"] == 5
    mg = out["multi_gpu"]  # where the step time goes: one SCALE run must be diagnostic
    # (round 5: the RCCL path is always timed; the fused path is the timed one when it is valid AND faster)
    assert mg["mode_timed"] in (("fused", "plain") if fused == "auto" else ("plain",))
    assert mg["path_used"] == ("fused" if mg["mode_timed"] == "fused" else "rccl")
    assert ("fused into the reduce kernels" in out["config"]["parallelism"]) == (mg["mode_timed"] == "fused")
    assert mg["local_spmv_ms"] > 0 and mg["gather_ms"] >= 0 and mg["rccl_step_ms"] > 0
    assert (mg["fused_step_ms"] is not None) == (fused == "auto")
    assert mg["fused_check"] is (True if fused == "auto" else None)
    assert mg["fused_post_check"] is (True if fused == "auto" else None)   # fused results re-checked after the timed loop
    want = min(mg["rccl_step_ms"], mg["fused_step_ms"]) if fused == "auto" else mg["rccl_step_ms"]
    assert abs(out["ms_per_step"] - want) <= 1e-6 * want + 1e-9  # the faster valid path is the published number


def test_bench_cfg4_rmat_multi_gpu_code_path_with_one_rank(gpu):
    """BASELINE cfg4's multi-GPU leg (`bench.py --gpus N --workload spmv_rmat`: fp64 R-MAT, rows sharded by nnz
    prefix, uneven all-gather) through the N > 1 code path with one rank, at scale 18."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
           "127.0.0.1", "--master-port", "29679", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--debug-multi",
           "--workload", "spmv_rmat", "--rows", str(1 << 18), "--steps", "5", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    out = _bench_full(lines[0])
    assert out["value"] > 0 and out["dtype"] == "f64" and out["config"]["nnz"] == 16 << 18
    assert "R-MAT scale 18" in out["config"]["workload"] and out["multi_gpu"]["rccl_step_ms"] > 0


@pytest.mark.parametrize("workload", ["spmv", "spmv_rmat"])
def test_bench_launches_its_own_ranks_and_checks_what_it_timed(gpu, workload):
    """`python3 bench.py --gpus N` WITHOUT an outer torch.distributed.run (the shape of the driver's N = 1 command):
    the parent starts the ranks as a child process before touching the GPU, relays rank 0's one JSON line and the
    exit code; the line carries `multi_gpu` and a `parity_check` of the timed operator's y (round-2 VERDICT item 1)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--debug-multi", "--steps", "5", "--warmup", "2",
           "--no-cpu-baseline", "--workload", workload, "--rows", "2000000" if workload == "spmv" else str(1 << 18)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    out = _bench_full(lines[0])
    assert out["multi_gpu"]["rccl_step_ms"] > 0
    assert out["parity_check"] == "pass" and out["parity"]["rows_out_of_bound"] == 0, out["parity"]
    assert out["parity"]["rows"] == out["config"]["rows"]


def test_bench_single_gpu_line_is_self_checking(gpu):
    """N = 1 (a reduced cfg2 so the test stays short): the y of the timed plan is compared with the oracle's y that the
    cpu_baseline leg computes anyway -- the published number and the checked number come from one process."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "1000000", "--steps", "5", "--warmup", "2"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    out = _bench_full(lines[0])
    assert out["parity_check"] == "pass" and out["parity"]["rows"] == 1000000
    assert out["cpu_baseline"]["cores"] == 1 and out["cpu_baseline"]["cpu_model"]
    assert out["multi_gpu"] is None


@pytest.mark.parametrize("world,workload,fused", [(2, "spmv", "auto"), (4, "spmv", "auto"), (2, "spmv", "off"), (2, "spmv_rmat", "auto"), (3, "spmv_rmat", "auto")])
def test_bench_with_several_ranks_on_one_gpu(gpu, world, workload, fused):
    """The whole N > 1 path of bench.py -- its own launcher, shard generation, row bounds (equal for cfg2, nnz-prefix
    for the R-MAT graph), try_fused adoption over hipIpc between processes, timed loop, post-check, diagnostics and the
    parity check of the gathered y on every rank -- with N real ranks that share cuda:0 (`--debug-one-gpu`: gloo as the
    process-group backend, since RCCL refuses two ranks on one device).  The closest a one-GPU box gets to the driver's
    SCALE run."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    rows = str(world * 600_000) if workload == "spmv" else str(1 << 18)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--debug-one-gpu", "--steps", "5", "--warmup",
           "2", "--workload", workload, "--rows", rows, "--fused", fused]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1 and r.stdout.strip() == lines[0], (r.stdout[-2000:], r.stderr[-3000:])
    out = _bench_full(lines[0])
    assert out["n_gpus"] == world and out["parity_check"] == "pass", out["parity"]
    mg = out["multi_gpu"]
    assert len(mg["rows_per_rank"]) == world and sum(mg["rows_per_rank"]) == out["config"]["rows"]
    if workload == "spmv":
        assert mg["mode_timed"] in (("fused", "plain") if fused == "auto" else ("plain",))
        assert mg["fused_post_check"] is (True if fused == "auto" else None)
        assert mg["backend"] == "gloo" and mg["rccl_nranks"] == world and mg["rccl_step_ms"] > 0
        if fused == "auto":  # the throughput form is measured next to the timed one and reproduces its bits
            assert mg["fused_check"] is True and mg["fused_step_ms"] > 0 and mg["chunked_step_ms"] is not None
            assert mg["fused_pipelined_step_ms"] > 0 and mg["fused_pipelined_check"] is True
    else:
        assert len(set(mg["rows_per_rank"])) > 1 and mg["gather"] == "p2p"   # nnz-prefix shards differ in rows


@pytest.mark.parametrize("workload,rows", [("add", 60000), ("transpose", 300000), ("sptrsv", 200000), ("spgemm", 50000)])
def test_bench_secondary_workloads_check_themselves(gpu, workload, rows):
    """The SURVEY 8(f) operations (and cfg5) as bench workloads at a reduced size: every line carries a parity object from
    the oracle -- structure exact for add / transpose / SpGEMM, residual + oracle comparison for the triangular solve --
    the same records the default `python bench.py` run puts into `secondary` at full size."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--rows", str(rows),
                        "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    out = _bench_full(lines[0])
    assert out["parity_check"] == "pass", out["parity"]
    assert out["roofline"]["kernel"] and out["value"] > 0


@pytest.mark.parametrize("mute_after", [0, 12])
def test_bench_falls_back_to_rccl_when_a_rank_stops_publishing_its_flags(gpu, mute_after):
    """First-contact safety of the SCALE run: a rank whose step flag never reaches its peers (mute_after = 0: from the
    first fused step, so the validation in try_fused already fails) or stops reaching them in the middle of the timed fused
    loop (mute_after = 12: after validation and probe) must cost bounded waits, not a hang: every rank falls back to the
    RCCL path collectively, the line carries the RCCL number as ms_per_step, fused_check false, parity pass, exit code 0."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SPBLAS_GFX950_TEST_MUTE_RANK="1", SPBLAS_GFX950_TEST_MUTE_AFTER=str(mute_after))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--debug-one-gpu", "--steps", "5", "--warmup", "2",
           "--rows", "1200000", "--fused", "auto", "--flag-chunks", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    out = _bench_full(lines[0])
    mg = out["multi_gpu"]
    assert out["parity_check"] == "pass" and mg["path_used"] == "rccl" and mg["mode_timed"] == "plain"
    assert mg["fused_check"] is False and mg["fused_step_ms"] is None
    assert abs(out["ms_per_step"] - mg["rccl_step_ms"]) <= 1e-6 * mg["rccl_step_ms"] + 1e-9
