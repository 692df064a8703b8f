"""Compiler evidence for the drop-in backend headers.

include/spblas/vendor/gfx950/*.hpp are the overloads a reference maintainer would compile in with
-DSPBLAS_ENABLE_GFX950.  The reference tree is only present in the build container, and its own headers need
range-v3 and kokkos-mdspan, which the image lacks.  This test
  1. copies the six reference headers INTEGRATION.md section 2 names into a scratch directory and applies exactly
     the edits that section lists (the scratch directory shadows the originals on the include path; nothing
     under /root/reference is written),
  2. puts tests/compile_check/stubs/ (a declared-only views::zip and a minimal rank-2 mdspan: COMPILE-CHECK
     infrastructure, they pin nothing and are not part of the product) where <range/v3/all.hpp> and
     <experimental/mdspan> are looked up,
  3. runs `g++ -std=c++20 -fsyntax-only -DSPBLAS_ENABLE_GFX950` on tests/compile_check/dropin_check.cpp, which
     instantiates every overload: multiply / multiply_inspect (SpMV on span, SpMM on row-major mdspan),
     multiply_compute / multiply_fill / multiply_symbolic_* / multiply_numeric (3- and 4-argument), add*,
     transpose*, scale, triangular_solve*.
Skipped where /root/reference does not exist (the GPU box).
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/include"
CHECK = os.path.join(ROOT, "tests", "compile_check")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


def _edit(text, anchor, addition, after=True, count=1):
    assert anchor in text, f"anchor not found: {anchor!r}"
    return text.replace(anchor, anchor + addition if after else addition + anchor, count)


def patched_reference_headers(dst):
    """INTEGRATION.md section 2, applied to a scratch copy."""
    def load(rel):
        with open(os.path.join(REF, rel)) as f:
            return f.read()

    def store(rel, text):
        path = os.path.join(dst, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)

    guard_old = "defined(SPBLAS_ENABLE_CUSPARSE)\n#define SPBLAS_VENDOR_BACKEND"
    guard_new = "defined(SPBLAS_ENABLE_CUSPARSE) || defined(SPBLAS_ENABLE_GFX950)\n#define SPBLAS_VENDOR_BACKEND"
    # spblas.hpp:3-7 -- the SPBLAS_VENDOR_BACKEND guard
    t = load("spblas/spblas.hpp")
    assert guard_old in t
    store("spblas/spblas.hpp", t.replace(guard_old, guard_new))
    # algorithms/algorithms.hpp: the CPU multiply / triangular_solve are already excluded by SPBLAS_VENDOR_BACKEND
    # (:8-11); the CPU scale / add / transpose loops cannot dereference device memory and their signatures are the
    # ones this backend provides for device operands, so they are excluded for this backend as well
    t = load("spblas/algorithms/algorithms.hpp")
    for impl in ("scale_impl", "add_impl", "transpose_impl"):
        t = t.replace(f"#include <spblas/algorithms/{impl}.hpp>\n",
                      f"#ifndef SPBLAS_ENABLE_GFX950\n#include <spblas/algorithms/{impl}.hpp>\n#endif\n")
    assert t.count("#ifndef SPBLAS_ENABLE_GFX950") == 3
    store("spblas/algorithms/algorithms.hpp", t)
    # backend/backend.hpp:9-27
    t = load("spblas/backend/backend.hpp")
    t = _edit(t, "#ifdef SPBLAS_ENABLE_CUSPARSE\n#include <spblas/vendor/cusparse/cusparse.hpp>\n#endif\n",
              "\n#ifdef SPBLAS_ENABLE_GFX950\n#include <spblas/vendor/gfx950/gfx950.hpp>\n#endif\n")
    store("spblas/backend/backend.hpp", t)
    # detail/types.hpp:6-24
    t = load("spblas/detail/types.hpp")
    t = _edit(t, "#ifdef SPBLAS_ENABLE_CUSPARSE\n#include <spblas/vendor/cusparse/types.hpp>\n#endif\n",
              "\n#ifdef SPBLAS_ENABLE_GFX950\n#include <spblas/vendor/gfx950/index_types.hpp>\n#endif\n")
    store("spblas/detail/types.hpp", t)
    # detail/operation_info_t.hpp:22-24 and :100-103
    t = load("spblas/detail/operation_info_t.hpp")
    t = _edit(t, "#ifdef SPBLAS_ENABLE_ROCSPARSE\n#include <spblas/vendor/rocsparse/operation_state_t.hpp>\n#endif\n",
              "\n#ifdef SPBLAS_ENABLE_GFX950\n#include <spblas/vendor/gfx950/detail/backend_calls.hpp>\n#endif\n")
    t = _edit(t, "#ifdef SPBLAS_ENABLE_ROCSPARSE\npublic:\n  __rocsparse::operation_state_t state_;\n#endif\n",
              "\n#ifdef SPBLAS_ENABLE_GFX950\npublic:\n  __gfx950::operation_state_t state_;\n#endif\n")
    store("spblas/detail/operation_info_t.hpp", t)
    # views/matrix_opt_impl.hpp (optional edit of INTEGRATION.md section 2: plan cache in the matrix_opt)
    return dst


def test_dropin_headers_compile_inside_the_reference_tree(tmp_path):
    gxx = shutil.which("g++")
    assert gxx, "g++ not found"
    scratch = patched_reference_headers(str(tmp_path / "patched"))
    cmd = [gxx, "-std=c++20", "-fsyntax-only", "-Wall", "-Wno-unused-variable", "-DSPBLAS_ENABLE_GFX950",
           "-D__HIP_PLATFORM_AMD__",
           "-I", scratch,                                  # the edited copies shadow the originals
           "-I", os.path.join(ROOT, "include"),             # spblas/vendor/gfx950/*.hpp, spblas_gfx950.h
           "-I", REF,                                      # the rest of the reference tree, untouched
           "-I", os.path.join(CHECK, "stubs"),              # <range/v3/all.hpp>, <experimental/mdspan> stand-ins
           "-I", "/opt/rocm/include",                       # hip_runtime_api.h for stream_memory.hpp
           os.path.join(CHECK, "dropin_check.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, "drop-in headers do not compile inside the reference tree:\n" + r.stderr[-6000:]
