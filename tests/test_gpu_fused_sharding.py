"""-m gpu: the fused all-gather of the row-sharded SpMV (hipIpc-mapped y copies + peer stores from the
reduce kernels + device-side step barrier), exercised with several processes on the one GPU of the test
box.  See tests/mp_fused_worker.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,dt,stripes", [(2, "f32", 1), (4, "f32", 3), (2, "f64", 4)])
def test_fused_sharded_spmv_multiprocess(gpu, world, dt, stripes):
    """stripes > 1: the reduces of contiguous bin groups alternate between two streams (the groups need not
    divide the bins evenly)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", FUSED_STRIPES=str(stripes))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29650 + world + (7 if dt == "f64" else 0)),
           os.path.join(ROOT, "tests", "mp_fused_worker.py"), dt]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "FUSED_OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
