"""Builds libspblas_gfx950.so (the C-ABI library, include/spblas_gfx950.h) with hipcc
for gfx950.  hipcc cross-compiles without a GPU; the .so is built IN-TREE
(spblas-reference_amd/lib/) so it travels with the repo snapshot to the GPU box."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
LIBDIR = os.path.join(_PKG, "lib")
LIBPATH = os.path.join(LIBDIR, "libspblas_gfx950.so")
SOURCES = ["handle.hip", "spmv.hip", "spmv_sliced.hip", "spmv_hot.hip", "spmm.hip", "spgemm.hip", "transpose.hip", "sptrsv.hip", "multigpu.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function", "-I", os.path.join(_ROOT, "include"), "-I", CSRC]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the gfx950 backend cannot be built")
    return exe


def _newest_input():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    files.append(os.path.join(_ROOT, "include", "spblas_gfx950.h"))
    return max(os.path.getmtime(f) for f in files)


def needs_build():
    return not os.path.exists(LIBPATH) or os.path.getmtime(LIBPATH) < _newest_input()


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIBPATH
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []

    def compile_one(src):
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBPATH] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIBPATH


CPP_TEST_SRC = os.path.join(_ROOT, "tests", "cpp", "device_tests.cpp")
CPP_TEST_BIN = os.path.join(_ROOT, "tests", "cpp", "device_tests")


def build_cpp_tests(force=False):
    """C++20 host program over the standalone API mirror (include/spblas_gfx950/spblas.hpp),
    compiled by g++ (the reference's host compiler class) and linked to the C-ABI library."""
    deps = [CPP_TEST_SRC, os.path.join(_ROOT, "include", "spblas_gfx950", "spblas.hpp"),
            os.path.join(_ROOT, "include", "spblas", "vendor", "gfx950", "detail", "backend_calls.hpp"), LIBPATH]
    if not force and os.path.exists(CPP_TEST_BIN) and os.path.getmtime(CPP_TEST_BIN) >= max(map(os.path.getmtime, deps)):
        return CPP_TEST_BIN
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-std=c++20", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(_ROOT, "include"),
           "-I", os.path.join(rocm, "include"), CPP_TEST_SRC, "-L", LIBDIR, "-lspblas_gfx950",
           "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath,$ORIGIN/../../spblas-reference_amd/lib",
           "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", CPP_TEST_BIN]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"g++ failed on device_tests.cpp:\n{r.stderr}")
    return CPP_TEST_BIN


RCCL_TEST_SRC = os.path.join(_ROOT, "tests", "cpp", "sharded_rccl_test.cpp")
RCCL_TEST_BIN = os.path.join(_ROOT, "tests", "cpp", "sharded_rccl_test")


def build_rccl_test(force=False):
    """The C++ row-sharded SpMV over RCCL (include/spblas/vendor/gfx950/sharded_spmv.hpp): a caller-side program that links
    librccl itself (the backend library does not)."""
    deps = [RCCL_TEST_SRC, os.path.join(_ROOT, "include", "spblas", "vendor", "gfx950", "sharded_spmv.hpp"),
            os.path.join(_ROOT, "include", "spblas", "vendor", "gfx950", "fused_sharded_spmv.hpp"),
            os.path.join(_ROOT, "include", "spblas", "vendor", "gfx950", "detail", "backend_calls.hpp"), LIBPATH]
    if not force and os.path.exists(RCCL_TEST_BIN) and os.path.getmtime(RCCL_TEST_BIN) >= max(map(os.path.getmtime, deps)):
        return RCCL_TEST_BIN
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-std=c++20", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(_ROOT, "include"),
           "-I", os.path.join(rocm, "include"), RCCL_TEST_SRC, "-L", LIBDIR, "-lspblas_gfx950",
           "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-lrccl", "-lpthread",
           "-Wl,-rpath,$ORIGIN/../../spblas-reference_amd/lib", "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", RCCL_TEST_BIN]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"g++ failed on sharded_rccl_test.cpp:\n{r.stderr}")
    return RCCL_TEST_BIN


EXAMPLES = ["device_spmv", "device_spgemm", "device_sptrsv"]


def build_examples(force=False):
    """The example programs under examples/ (C++20 over the standalone API mirror), compiled by g++."""
    exdir = os.path.join(_ROOT, "examples")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    hdrs = [os.path.join(exdir, "common.hpp"), os.path.join(_ROOT, "include", "spblas_gfx950", "spblas.hpp"),
            os.path.join(_ROOT, "include", "spblas", "vendor", "gfx950", "detail", "backend_calls.hpp"), LIBPATH]
    out = []
    for name in EXAMPLES:
        src, binp = os.path.join(exdir, name + ".cpp"), os.path.join(exdir, name)
        out.append(binp)
        if not force and os.path.exists(binp) and os.path.getmtime(binp) >= max(map(os.path.getmtime, hdrs + [src])):
            continue
        cmd = ["g++", "-std=c++20", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(_ROOT, "include"),
               "-I", os.path.join(rocm, "include"), src, "-L", LIBDIR, "-lspblas_gfx950", "-L",
               os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath,$ORIGIN/../spblas-reference_amd/lib",
               "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", binp]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"g++ failed on {name}.cpp:\n{r.stderr}")
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_cpp_tests(force=True))
    print(build_rccl_test(force=True))
    print(build_examples(force=True))
