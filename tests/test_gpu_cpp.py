"""-m gpu: the C++20 host layer end to end.  tests/cpp/device_tests.cpp restates the reference's
device gtests (test/gtest/device/spmv_test.cpp, spgemm_test.cpp, spgemm_reuse_test.cpp and the
SpMM cases of test/gtest/spmm_test.cpp) against include/spblas_gfx950/spblas.hpp, which calls
the same C ABI through the same __gfx950 layer the drop-in backend headers use."""
import os
import subprocess

import pytest

from spblas_reference_amd import _build

pytestmark = pytest.mark.gpu


def test_cpp_device_tests(gpu):
    exe = _build.CPP_TEST_BIN
    if not os.path.exists(exe):
        exe = _build.build_cpp_tests()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "PASSED" in r.stdout


@pytest.mark.parametrize("name", _build.EXAMPLES)
def test_cpp_examples(gpu, name):
    """examples/*.cpp: the programs a user of the reference would write (views over device arrays, inspect,
    multiply / multiply_compute + multiply_fill / add / triangular_solve), each checking itself on the host."""
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", name)
    if not os.path.exists(exe):
        _build.build_examples()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert name in r.stdout


def test_cpp_sharded_spmv_over_rccl_one_rank(gpu):
    """include/spblas/vendor/gfx950/sharded_spmv.hpp: the row-sharded SpMV a C++ caller can use -- local SpMV into its slot of
    the full y + ONE ncclAllGather (equal shards) / grouped ncclBroadcast (nnz-balanced shards) on the same stream -- with
    the communicator of one rank this box allows (tests/cpp/sharded_rccl_test.cpp; librccl is linked by the PROGRAM, the
    backend library stays free of it)."""
    exe = _build.RCCL_TEST_BIN
    if not os.path.exists(exe):
        exe = _build.build_rccl_test()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "SHARDED_RCCL_OK ranks=1" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
