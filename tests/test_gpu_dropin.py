"""-m gpu: the drop-in backend headers RUNNING behind the reference's own API (SURVEY.md section 8 rows a10 / b).

tests/compile_check/_build/dropin_run is tests/compile_check/dropin_run.cpp compiled in the build container against
the reference tree's <spblas/spblas.hpp> with -DSPBLAS_ENABLE_GFX950 (INTEGRATION.md section 2's edits applied to a
scratch copy; tests/compile_check/build_dropin.py, called from __graft_entry__.build()).  It calls
spblas::multiply / multiply_inspect / multiply_compute / multiply_fill / multiply_symbolic_* / multiply_numeric /
add_inspect / add_compute / transpose / scale / triangular_solve[_inspect] on spblas::csr_view, csc_view,
transposed(), scaled(), matrix_opt and row-major mdspans over device pointers, exactly as a user of the reference
would, and checks every result against host loops.  The GPU box has no reference tree: only the binary travels."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "compile_check", "_build", "dropin_run")


def test_dropin_backend_runs_behind_the_reference_api(gpu):
    if not os.path.exists(BIN):
        pytest.skip("tests/compile_check/_build/dropin_run was not built (it needs the reference tree: run "
                    "__graft_entry__.build() in the build container before the snapshot is taken)")
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, f"dropin_run failed:\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    assert "checks, 0 failed" in r.stdout, r.stdout
    n = int(r.stdout.split("dropin_run:")[1].split("checks")[0])
    assert n >= 25, r.stdout


def test_dropin_backend_built_through_cmake_runs(gpu):
    """The same program, built by CMake: cmake/SpblasGfx950.cmake -- the `option(ENABLE_GFX950)` block of INTEGRATION.md
    section 2 as an includable module -- configured with -DENABLE_GFX950=ON by tests/compile_check/cmake_project
    (tests/compile_check/build_dropin.py:build_dropin_run_with_cmake, called from build())."""
    binp = os.path.join(ROOT, "tests", "compile_check", "_build", "dropin_run_cmake")
    if not os.path.exists(binp):
        pytest.skip("dropin_run_cmake was not built (needs the reference tree and cmake: __graft_entry__.build())")
    r = subprocess.run([binp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "checks, 0 failed" in r.stdout, f"{r.stdout[-2000:]}\n{r.stderr[-3000:]}"


REF_BIN = os.path.join(ROOT, "tests", "compile_check", "_build", "reference_device_tests")


def test_reference_device_tests_pass_on_this_backend(gpu):
    """The reference's OWN device tests -- test/gtest/device/{spmv,spgemm,spgemm_reuse}_test.cpp and
    device/rocsparse/spgemm_4args_test.cpp, 14 TESTs -- compiled unmodified from the reference tree (never copied
    here) against <spblas/spblas.hpp> with -DSPBLAS_ENABLE_GFX950, i.e. with this backend behind spblas::multiply,
    multiply_compute / multiply_fill, multiply_symbolic_* / multiply_numeric and the four-argument forms
    (tests/compile_check/build_dropin.py: hipcc, thrust device vectors, a stand-in for the GoogleTest macros).
    Their expected values are the host loops those files contain (spa_accumulator over __backend::rows)."""
    if not os.path.exists(REF_BIN):
        pytest.skip("tests/compile_check/_build/reference_device_tests was not built (needs the reference tree: "
                    "__graft_entry__.build() in the build container)")
    r = subprocess.run([REF_BIN], capture_output=True, text=True, timeout=600)
    tail = r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "14 tests ran, 0 failed" in r.stdout, tail


@pytest.mark.parametrize("name", ["ref_example_device_spmv", "ref_example_rocsparse_simple_spmv"])
def test_reference_device_examples_run_on_this_backend(gpu, name):
    """examples/device/device_spmv.cpp and examples/rocsparse/rocsparse_simple_spmv.cpp of the reference, compiled
    unmodified against this backend (tests/compile_check/build_dropin.py): they run spblas::multiply on device arrays
    and end with "Example is completed!"."""
    binp = os.path.join(ROOT, "tests", "compile_check", "_build", name)
    if not os.path.exists(binp):
        pytest.skip(f"{name} was not built (needs the reference tree and a fmt header: __graft_entry__.build())")
    r = subprocess.run([binp], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "Example is completed!" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]


def test_reference_host_tests_pass_on_this_backend(gpu):
    """The reference's HOST tests -- test/gtest/{spmv,spmm,spgemm,spgemm_csr_csc,add,transpose,triangular_solve}_test.cpp
    and mdspan_overlays.cpp, the list its CMake builds for CPU backends (test/gtest/CMakeLists.txt:7-15), 28 TESTs --
    compiled UNMODIFIED against this device backend.  They keep their operands in std::vector and never synchronise;
    tests/compile_check/gtest_main_pinned.cpp gives the test binary a heap of pinned, device-visible memory
    (operator new -> one hipHostMalloc slab) and the binary runs with AMD_SERIALIZE_KERNEL / AMD_SERIALIZE_COPY = 3 so
    that every launch has finished when the call returns.  Neither the tests nor the backend are changed for it.
    Covers what the device tests do not: SpMM (n = 1 ... 512, csr and csc A), csc / transposed SpMV with scaling, all
    CSR / CSC combinations of SpGEMM, add, transpose, both triangular solves, the mdspan overlays."""
    binp = os.path.join(ROOT, "tests", "compile_check", "_build", "reference_host_tests")
    if not os.path.exists(binp):
        pytest.skip("reference_host_tests was not built (needs the reference tree and a fmt header: __graft_entry__.build())")
    env = dict(os.environ, AMD_SERIALIZE_KERNEL="3", AMD_SERIALIZE_COPY="3")
    r = subprocess.run([binp], capture_output=True, text=True, timeout=900, env=env)
    tail = r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "28 tests ran, 0 failed" in r.stdout, tail


@pytest.mark.parametrize("name", ["simple_spmv", "simple_spmm", "simple_spgemm", "simple_sptrsv", "spmm_csc", "spmm_csr",
                                  "sptrsv_csr", "matrix_opt_example"])
def test_reference_host_examples_run_on_this_backend(gpu, name):
    """examples/*.cpp of the reference (host vectors), compiled unmodified against this device backend and linked with
    tests/compile_check/pinned_heap.cpp (pinned, device-visible heap before main()); serialised launches as above."""
    binp = os.path.join(ROOT, "tests", "compile_check", "_build", "ref_host_example_" + name)
    if not os.path.exists(binp):
        pytest.skip(f"{name} was not built (needs the reference tree and a fmt header: __graft_entry__.build())")
    env = dict(os.environ, AMD_SERIALIZE_KERNEL="3", AMD_SERIALIZE_COPY="3")
    r = subprocess.run([binp], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and "Example is completed!" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
