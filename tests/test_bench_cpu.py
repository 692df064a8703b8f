"""CPU tests of bench.py's launcher and self-check helpers (no GPU work)."""
import importlib.util
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_plain_gpus_n_starts_a_child_torchrun_before_any_gpu_call(monkeypatch):
    """`python3 bench.py --gpus 4` with no WORLD_SIZE in the environment must become a CHILD
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 ... bench.py --gpus 4` (never an exec of a process
    that touched the GPU) and return the child's exit code."""
    bench = _bench()
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    import torch
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *_: (_ for _ in ()).throw(AssertionError("GPU touched in the parent")))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert bench.main() == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_more_ranks_than_devices_is_refused_at_once(monkeypatch, capsys):
    """--gpus 8 on a box with 2 devices: say so and exit 2 instead of starting ranks that cannot get a device."""
    bench = _bench()
    import torch
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(AssertionError("ranks were started")))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    assert bench.main() == 2
    assert "2 device(s) visible" in capsys.readouterr().err


def test_parity_report_bites():
    bench = _bench()
    rng = np.random.default_rng(0)
    absrow = rng.random(1000) + 1.0
    y_ref = (absrow * 0.5).astype(np.float32)
    ok = bench.parity_spmv(y_ref + np.float32(1e-7) * y_ref, y_ref, absrow, 1e-6)
    assert ok["status"] == "pass" and ok["rows"] == 1000
    y_bad = y_ref.copy()
    y_bad[17] *= np.float32(1.0005)
    bad = bench.parity_spmv(y_bad, y_ref, absrow, 1e-6)
    assert bad["status"] == "fail" and bad["rows_out_of_bound"] == 1
    y_nan = y_ref.copy()
    y_nan[3] = np.nan  # a row the kernel never wrote
    assert bench.parity_spmv(y_nan, y_ref, absrow, 1e-6)["status"] == "fail"
