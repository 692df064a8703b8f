"""The CPU oracle pinned by the reference's OWN tests ("check it against every ... known-answer test
... the reference's own tests hold for this path").

The reference stores no golden vectors: its known answers are the comparator loops inside its test files.  Here the eight
host test files its CMake builds for CPU backends (test/gtest/CMakeLists.txt:7-15: spmv, spmm, spgemm, spgemm_csr_csc,
add, transpose, triangular_solve, mdspan_overlays -- 28 TESTs) are compiled UNMODIFIED, from where they lie in the
reference tree, behind the drop-in header layer, and linked to tests/compile_check/oracle_shim.c, which implements the C
ABI those headers call with oracle/spblas_oracle.c on ordinary host memory (no GPU).  So spblas::multiply,
multiply_compute / multiply_fill, add, transpose and triangular_solve of those tests END IN THE ORACLE, and the
reference's own EXPECT_EQ_ comparators (test/gtest/util.hpp:7-23) judge it.

What this is not: a build of the reference's algorithms (with a vendor backend selected none of them is compiled in; the
stand-in headers under tests/compile_check/stubs/ only let its views, generators and test macros parse).  The binary is
built where the reference tree exists (this container); where it does not (the GPU box) the prebuilt binary still runs.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from compile_check import build_dropin  # noqa: E402


def test_oracle_passes_the_reference_host_tests():
    binp = build_dropin.ORACLE_HOST_TESTS
    if os.path.isdir(build_dropin.REF):
        built = build_dropin.build_reference_host_tests_on_oracle()
        assert built, "could not build the reference's host tests against the oracle (fmt header missing?)"
        binp = built
    elif not os.path.exists(binp):
        pytest.skip("no reference tree here and no prebuilt reference_host_tests_on_oracle")
    r = subprocess.run([binp], capture_output=True, text=True, timeout=900)
    tail = r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "28 tests ran, 0 failed" in r.stdout, tail
    ran = [line.split("]", 1)[1].strip() for line in r.stdout.splitlines() if line.startswith("[       OK ]")]
    for must in ("CsrView.SpMV", "CscView.SpMV", "CsrView.SpMM", "CscView.SpMM", "CsrView.SpGEMM", "CscView.SpGEMM",
                 "CsrView.Add_CSR_CSR_CSR", "CsrView.Transpose", "CsrView.TriangularSolveLowerImplicit"):
        assert must in ran, (must, ran)


def test_oracle_passes_the_reference_device_tests():
    """The four DEVICE test files of the reference (test/gtest/CMakeLists.txt:24-26: spmv, spgemm, spgemm_reuse,
    rocsparse/spgemm_4args -- 14 TESTs) judge the oracle as well: thrust::device_vector is a host vector here
    (tests/compile_check/stubs_host/thrust/device_vector.h), everything else as above.  This is what pins
    oracle_spgemm_symbolic_d / oracle_spgemm_numeric_d_* (C = alpha A B + beta D has no CPU implementation in the
    reference: its known answers are spgemm_4args_test.cpp:78-108) and the symbolic / numeric reuse family."""
    binp = build_dropin.ORACLE_DEVICE_TESTS
    if os.path.isdir(build_dropin.REF):
        built = build_dropin.build_reference_device_tests_on_oracle()
        assert built
        binp = built
    elif not os.path.exists(binp):
        pytest.skip("no reference tree here and no prebuilt reference_device_tests_on_oracle")
    r = subprocess.run([binp], capture_output=True, text=True, timeout=900)
    tail = r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "14 tests ran, 0 failed" in r.stdout, tail


@pytest.mark.parametrize("which", ["host", "device"])
def test_cpu_stack_is_clean_under_asan_and_ubsan(which):
    """The reference's -DENABLE_SANITIZERS configuration (/root/reference/CMakeLists.txt:9,113-117:
    -fsanitize=address,undefined) applied to everything of this build that runs on the CPU: the reference's unmodified
    test files, the drop-in overloads (include/spblas/vendor/gfx950/*.hpp), the state objects and lifetimes of
    detail/backend_calls.hpp, tests/compile_check/oracle_shim.c and oracle/spblas_oracle.c -- 28 host + 14 device tests
    with -fno-sanitize-recover=all, so any heap / stack error, leak or undefined operation fails the run.  (CPU only: GPU
    sanitizers are not available on this pool.)"""
    build = build_dropin.build_reference_host_tests_on_oracle if which == "host" else \
        build_dropin.build_reference_device_tests_on_oracle
    binp = (build_dropin.ORACLE_HOST_TESTS if which == "host" else build_dropin.ORACLE_DEVICE_TESTS) + "_asan"
    if os.path.isdir(build_dropin.REF):
        binp = build(sanitize=True)
        assert binp
    elif not os.path.exists(binp):
        pytest.skip("no reference tree here and no prebuilt sanitizer binary")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([binp], capture_output=True, text=True, timeout=900, env=env)
    tail = r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    assert r.returncode == 0, tail
    assert ("28 tests ran, 0 failed" if which == "host" else "14 tests ran, 0 failed") in r.stdout, tail
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, tail


def test_reference_tests_notice_an_oracle_that_stops_sorting(tmp_path):
    """Mutation check of the pin itself.  The reference's SpGEMM / add comparators re-accumulate C's rows through a sparse
    accumulator (test/gtest/spgemm_test.cpp:56-65), so they accept any column order -- while the result must come back
    sorted (spgemm_gustavsons.hpp:42).  oracle_shim.c therefore checks the order of every SpGEMM / add
    result it hands back; an oracle built WITHOUT its column sort (-DORACLE_MUTATION_NO_SORT), preloaded over the shim of
    the very same test binary, must make the reference's own tests fail."""
    import shutil
    binp = build_dropin.ORACLE_HOST_TESTS
    if os.path.isdir(build_dropin.REF):
        binp = build_dropin.build_reference_host_tests_on_oracle() or binp
    if not os.path.exists(binp):
        pytest.skip("no reference tree here and no prebuilt reference_host_tests_on_oracle")
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    mutant = str(tmp_path / "liboracle_shim_nosort.so")
    here = os.path.join(ROOT, "tests", "compile_check")
    r = subprocess.run([gcc, "-O2", "-fPIC", "-shared", "-DORACLE_MUTATION_NO_SORT", "-I", os.path.join(ROOT, "include"),
                        os.path.join(here, "oracle_shim.c"), os.path.join(ROOT, "oracle", "spblas_oracle.c"), "-o", mutant],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    good = subprocess.run([binp], capture_output=True, text=True, timeout=900)
    assert good.returncode == 0 and "0 failed" in good.stdout
    bad = subprocess.run([binp], capture_output=True, text=True, timeout=900, env=dict(os.environ, LD_PRELOAD=mutant))
    out = bad.stdout + bad.stderr
    assert bad.returncode != 0, "the reference's tests passed on an oracle that does not sort its rows"
    assert "not in ascending column order" in out, out[-3000:]
    failed = [line for line in bad.stdout.splitlines() if line.startswith("[  FAILED  ]")]
    assert any("SpGEMM" in line for line in failed), failed
