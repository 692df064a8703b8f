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
SOURCES = ["handle.hip", "spmv.hip", "spmv_sliced.hip", "spmm.hip", "spgemm.hip"]
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

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBPATH] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIBPATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
