"""Compile-check / run-check infrastructure for the drop-in backend headers (include/spblas/vendor/gfx950/*.hpp).

patched_reference_headers(dst)  applies INTEGRATION.md section 2's edits to a scratch copy of the six reference
                                headers it names (nothing under /root/reference is written).
build_dropin_run()              compiles tests/compile_check/dropin_run.cpp -- the reference's own <spblas/spblas.hpp>
                                with -DSPBLAS_ENABLE_GFX950 -- and links it to libspblas_gfx950.so.  Only where the
                                reference tree exists (the build container); the binary
                                (tests/compile_check/_build/dropin_run, git-ignored) travels to the GPU box with the
                                snapshot and tests/test_gpu_dropin.py runs it there.
The stubs under tests/compile_check/stubs/ stand in for range-v3's views::zip and for mdspan ONLY so that the
reference's views and concepts parse; with a vendor backend selected none of the reference's algorithms is compiled in,
and every expected value in dropin_run.cpp comes from host loops written in that file.  This is not a build of the
reference, and it pins nothing for the oracle.
"""
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/include"
OUT_DIR = os.path.join(HERE, "_build")
DROPIN_RUN = os.path.join(OUT_DIR, "dropin_run")
REF_DEVICE_TESTS = os.path.join(OUT_DIR, "reference_device_tests")
# the reference's own device tests, compiled UNMODIFIED from where they lie (never copied into this repository)
REF_TEST_DIR = "/root/reference/test/gtest"
# ... and its two device examples (they print a banner and "Example is completed!"; fmt comes header-only from the
# copy that ships inside the image's PyTorch)
REF_EXAMPLES = {"ref_example_device_spmv": "/root/reference/examples/device/device_spmv.cpp",
                "ref_example_rocsparse_simple_spmv": "/root/reference/examples/rocsparse/rocsparse_simple_spmv.cpp"}
# its HOST tests (the ones the reference builds for CPU backends, test/gtest/CMakeLists.txt:7-15; conjugate_test.cpp is
# complex-only and built for two backends only, :17-19), run against this DEVICE backend through gtest_main_pinned.cpp
REF_HOST_TESTS = os.path.join(OUT_DIR, "reference_host_tests")
REF_HOST_TEST_SOURCES = ["spmv_test.cpp", "spmm_test.cpp", "spgemm_test.cpp", "spgemm_csr_csc.cpp", "add_test.cpp",
                         "transpose_test.cpp", "triangular_solve_test.cpp", "mdspan_overlays.cpp"]
# ... and its eight host examples (their own main(): linked with pinned_heap.cpp)
REF_HOST_EXAMPLES = ["simple_spmv", "simple_spmm", "simple_spgemm", "simple_sptrsv", "spmm_csc", "spmm_csr", "sptrsv_csr",
                     "matrix_opt_example"]
REF_TEST_SOURCES = ["device/spmv_test.cpp", "device/spgemm_test.cpp", "device/spgemm_reuse_test.cpp",
                    "device/rocsparse/spgemm_4args_test.cpp"]


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
    # views/matrix_opt_impl.hpp:25-28,90-92 -- the plan cache of a matrix_opt (oneMKL keeps its optimised handle the same way)
    t = load("spblas/views/matrix_opt_impl.hpp")
    t = _edit(t, "#include <spblas/concepts.hpp>\n",
              "\n#ifdef SPBLAS_ENABLE_GFX950\n#include <memory>\n#include <spblas/vendor/gfx950/detail/backend_calls.hpp>\n#endif\n")
    t = _edit(t, "  matrix_opt(M matrix) : matrix_(matrix) {\n",
              "#ifdef SPBLAS_ENABLE_GFX950\n    gfx950_state_ = std::make_shared<__gfx950::opt_cache_t>();\n#endif\n")
    t = _edit(t, "public:\n  M matrix_;\n",
              "\n#ifdef SPBLAS_ENABLE_GFX950\n  std::shared_ptr<__gfx950::opt_cache_t> gfx950_state_;  // shared by the copies of the view\n#endif\n")
    store("spblas/views/matrix_opt_impl.hpp", t)
    return dst


def compile_flags(scratch):
    return ["-std=c++20", "-Wall", "-Wno-unused-variable", "-DSPBLAS_ENABLE_GFX950", "-D__HIP_PLATFORM_AMD__",
            "-I", scratch,                                  # the edited copies shadow the originals
            "-I", os.path.join(ROOT, "include"),             # spblas/vendor/gfx950/*.hpp, spblas_gfx950.h
            "-I", REF,                                      # the rest of the reference tree, untouched
            "-I", os.path.join(HERE, "stubs"),               # <range/v3/all.hpp>, <experimental/mdspan> stand-ins
            "-I", os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "include")]


def build_dropin_run(libdir):
    """Returns the binary's path, or None where the reference tree does not exist."""
    if not os.path.isdir(REF):
        return None
    gxx = shutil.which("g++")
    if not gxx:
        raise RuntimeError("g++ not found")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    os.makedirs(OUT_DIR, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        scratch = patched_reference_headers(os.path.join(tmp, "patched"))
        cmd = [gxx, "-O1"] + compile_flags(scratch) + [
            os.path.join(HERE, "dropin_run.cpp"), "-L", libdir, "-lspblas_gfx950", "-L", os.path.join(rocm, "lib"),
            "-lamdhip64", "-Wl,-rpath,$ORIGIN/../../../spblas-reference_amd/lib", "-Wl,-rpath," + os.path.join(rocm, "lib"),
            "-o", DROPIN_RUN]
        r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("g++ failed on dropin_run.cpp:\n" + r.stderr[-8000:])
    return DROPIN_RUN


DROPIN_RUN_CMAKE = os.path.join(OUT_DIR, "dropin_run_cmake")


def build_dropin_run_with_cmake(libdir):
    """The same program through CMake: cmake/SpblasGfx950.cmake (the `option(ENABLE_GFX950)` block INTEGRATION.md section 2
    gives the reference's CMakeLists.txt, as an includable module) is configured with -DENABLE_GFX950=ON by
    tests/compile_check/cmake_project/CMakeLists.txt and builds dropin_run.cpp against the patched header tree.  Returns
    the binary's path, or None without the reference tree / cmake."""
    cmake = shutil.which("cmake")
    if not os.path.isdir(REF) or not cmake:
        return None
    os.makedirs(OUT_DIR, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        scratch = patched_reference_headers(os.path.join(tmp, "patched"))
        headers = ";".join([scratch, REF, os.path.join(HERE, "stubs")])
        gen = ["-G", "Ninja"] if shutil.which("ninja") else []
        cfg = [cmake, "-S", os.path.join(HERE, "cmake_project"), "-B", os.path.join(tmp, "build")] + gen + [
            "-DENABLE_GFX950=ON", f"-DSPBLAS_HEADERS={headers}", f"-DSPBLAS_GFX950_LIBDIR={libdir}",
            "-DCMAKE_BUILD_TYPE=Release", "-DCMAKE_CXX_COMPILER=" + shutil.which("g++")]
        r = subprocess.run(cfg, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("cmake configure failed:\n" + r.stdout[-3000:] + r.stderr[-3000:])
        r = subprocess.run([cmake, "--build", os.path.join(tmp, "build"), "-j", "4"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("cmake build failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
        shutil.copy2(os.path.join(tmp, "build", "dropin_run_cmake"), DROPIN_RUN_CMAKE)
    return DROPIN_RUN_CMAKE


def build_reference_device_tests(libdir, jobs=4):
    """The reference's device test files (test/gtest/device/*.cpp: thrust device vectors + GoogleTest macros)
    compiled as they are against the patched tree with -DSPBLAS_ENABLE_GFX950 -- i.e. with THIS backend behind
    spblas::multiply & co. -- by hipcc (thrust needs a device compiler), with stubs/gtest/gtest.h standing in for
    GoogleTest (TEST / EXPECT_EQ / EXPECT_NEAR + a main that runs the registry).  Their expected values are the
    loops those test files contain.  Returns the binary's path, or None where the reference tree does not exist."""
    if not os.path.isdir(REF) or not os.path.isdir(REF_TEST_DIR):
        return None
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    hipcc = os.path.join(rocm, "bin", "hipcc")
    os.makedirs(OUT_DIR, exist_ok=True)
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory() as tmp:
        scratch = patched_reference_headers(os.path.join(tmp, "patched"))
        flags = [f for f in compile_flags(scratch) if f != "-D__HIP_PLATFORM_AMD__"]  # hipcc defines it
        common = [hipcc, "--offload-arch=gfx950", "-O1", "-Wno-reorder-ctor", "-Wno-unused-but-set-variable"] + flags
        sources = [os.path.join(REF_TEST_DIR, s) for s in REF_TEST_SOURCES] + [os.path.join(HERE, "gtest_main.cpp")]
        objs = [os.path.join(tmp, f"t{i}.o") for i in range(len(sources))]

        def compile_one(i):
            return subprocess.run(common + ["-c", sources[i], "-o", objs[i]], capture_output=True, text=True)

        with ThreadPoolExecutor(max_workers=jobs) as pool:
            for i, r in enumerate(pool.map(compile_one, range(len(sources)))):
                if r.returncode != 0:
                    raise RuntimeError(f"hipcc failed on {sources[i]}:\n" + r.stderr[-8000:])
        r = subprocess.run([hipcc, "--offload-arch=gfx950"] + objs + [
            "-L", libdir, "-lspblas_gfx950", "-Wl,-rpath,$ORIGIN/../../../spblas-reference_amd/lib",
            "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", REF_DEVICE_TESTS], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link of reference_device_tests failed:\n" + r.stderr[-8000:])
    return REF_DEVICE_TESTS


def _fmt_include():
    try:
        import torch
        inc = os.path.join(os.path.dirname(torch.__file__), "include")
        return inc if os.path.exists(os.path.join(inc, "fmt", "core.h")) else None
    except Exception:  # noqa: BLE001
        return None


def build_reference_examples(libdir):
    """The reference's device examples (examples/device/device_spmv.cpp, examples/rocsparse/rocsparse_simple_spmv.cpp),
    compiled unmodified against this backend.  Returns the binaries' paths ([] where the reference tree or a fmt header
    is missing)."""
    fmt = _fmt_include()
    if not os.path.isdir(REF) or fmt is None or not all(os.path.exists(p) for p in REF_EXAMPLES.values()):
        return []
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    hipcc = os.path.join(rocm, "bin", "hipcc")
    os.makedirs(OUT_DIR, exist_ok=True)
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        scratch = patched_reference_headers(os.path.join(tmp, "patched"))
        flags = [f for f in compile_flags(scratch) if f != "-D__HIP_PLATFORM_AMD__"]
        for name, src in REF_EXAMPLES.items():
            binp = os.path.join(OUT_DIR, name)
            r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-Wno-reorder-ctor", "-DFMT_HEADER_ONLY", "-I", fmt] +
                               flags + [src, "-L", libdir, "-lspblas_gfx950",
                                        "-Wl,-rpath,$ORIGIN/../../../spblas-reference_amd/lib",
                                        "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", binp],
                               capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n" + r.stderr[-6000:])
            out.append(binp)
    return out


def build_reference_host_tests(libdir, jobs=4):
    """The reference's HOST test files (std::vector operands, no synchronisation) compiled unmodified against this
    device backend and linked with gtest_main_pinned.cpp, which makes every heap allocation of the test binary
    pinned, device-visible memory.  Returns the binary's path, or None where the reference tree / a fmt header is
    missing."""
    fmt = _fmt_include()
    if not os.path.isdir(REF) or not os.path.isdir(REF_TEST_DIR) or fmt is None:
        return None
    gxx = shutil.which("g++")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    os.makedirs(OUT_DIR, exist_ok=True)
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory() as tmp:
        scratch = patched_reference_headers(os.path.join(tmp, "patched"))
        common = [gxx, "-O1", "-DFMT_HEADER_ONLY", "-I", fmt] + compile_flags(scratch)
        sources = [os.path.join(REF_TEST_DIR, s) for s in REF_HOST_TEST_SOURCES] + [os.path.join(HERE, "gtest_main_pinned.cpp")]
        objs = [os.path.join(tmp, f"h{i}.o") for i in range(len(sources))]

        def compile_one(i):
            return subprocess.run(common + ["-c", sources[i], "-o", objs[i]], capture_output=True, text=True)

        with ThreadPoolExecutor(max_workers=jobs) as pool:
            for i, r in enumerate(pool.map(compile_one, range(len(sources)))):
                if r.returncode != 0:
                    raise RuntimeError(f"g++ failed on {sources[i]}:\n" + r.stderr[-8000:])
        r = subprocess.run([gxx] + objs + ["-L", libdir, "-lspblas_gfx950", "-L", os.path.join(rocm, "lib"), "-lamdhip64",
                                           "-Wl,-rpath,$ORIGIN/../../../spblas-reference_amd/lib",
                                           "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", REF_HOST_TESTS],
                           capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link of reference_host_tests failed:\n" + r.stderr[-8000:])
    return REF_HOST_TESTS


def build_reference_host_examples(libdir, jobs=4):
    """examples/*.cpp of the reference (host vectors, fmt output) against this device backend, each linked with
    pinned_heap.cpp (the pinned, device-visible heap set up before main()).  Returns the binaries' paths."""
    fmt = _fmt_include()
    if not os.path.isdir(REF) or fmt is None:
        return []
    gxx = shutil.which("g++")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    os.makedirs(OUT_DIR, exist_ok=True)
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory() as tmp:
        scratch = patched_reference_headers(os.path.join(tmp, "patched"))
        common = [gxx, "-O1", "-DFMT_HEADER_ONLY", "-I", fmt] + compile_flags(scratch)
        tail = [os.path.join(HERE, "pinned_heap.cpp"), "-L", libdir, "-lspblas_gfx950", "-L", os.path.join(rocm, "lib"),
                "-lamdhip64", "-Wl,-rpath,$ORIGIN/../../../spblas-reference_amd/lib", "-Wl,-rpath," + os.path.join(rocm, "lib")]

        def build_one(name):
            binp = os.path.join(OUT_DIR, "ref_host_example_" + name)
            return binp, subprocess.run(common + [os.path.join("/root/reference/examples", name + ".cpp")] + tail + ["-o", binp],
                                        capture_output=True, text=True)

        out = []
        with ThreadPoolExecutor(max_workers=jobs) as pool:
            for binp, r in pool.map(build_one, REF_HOST_EXAMPLES):
                if r.returncode != 0:
                    raise RuntimeError(f"g++ failed on {binp}:\n" + r.stderr[-6000:])
                out.append(binp)
    return out


ORACLE_HOST_TESTS = os.path.join(OUT_DIR, "reference_host_tests_on_oracle")


SANITIZE_FLAGS = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g"]


def build_reference_host_tests_on_oracle(jobs=4, sanitize=False):
    """The same eight host test files of the reference, behind the same drop-in header layer, but linked to
    oracle_shim.c + oracle/spblas_oracle.c instead of the GPU library: spblas::multiply & co. end in the CPU oracle, on
    ordinary host memory, no GPU involved.  Passing = the oracle reproduces every known answer the reference's tests
    hold for this path (tests/test_oracle_reference_tests.py).  Returns the binary's path or None."""
    fmt = _fmt_include()
    if not os.path.isdir(REF) or not os.path.isdir(REF_TEST_DIR) or fmt is None:
        return None
    gxx, gcc = shutil.which("g++"), shutil.which("gcc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    os.makedirs(OUT_DIR, exist_ok=True)
    # sanitize=True: the reference's -DENABLE_SANITIZERS build (/root/reference/CMakeLists.txt:9,113-117:
    # -fsanitize=address,undefined) of the whole CPU-only stack -- test files, drop-in headers, backend_calls.hpp state
    # lifetimes, the shim and the oracle -- as a second binary next to the plain one
    san = SANITIZE_FLAGS if sanitize else []
    sfx = "_asan" if sanitize else ""
    shim = os.path.join(OUT_DIR, f"liboracle_shim{sfx}.so")
    r = subprocess.run([gcc, "-O1" if sanitize else "-O2", "-fPIC", "-shared"] + san +
                       ["-I", os.path.join(ROOT, "include"), os.path.join(HERE, "oracle_shim.c"),
                        os.path.join(ROOT, "oracle", "spblas_oracle.c"), "-o", shim], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("gcc failed on oracle_shim.c:\n" + r.stderr[-4000:])
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory() as tmp:
        scratch = patched_reference_headers(os.path.join(tmp, "patched"))
        common = [gxx, "-O1", "-DFMT_HEADER_ONLY", "-I", fmt] + san + compile_flags(scratch)
        sources = [os.path.join(REF_TEST_DIR, s) for s in REF_HOST_TEST_SOURCES] + [os.path.join(HERE, "gtest_main.cpp")]
        objs = [os.path.join(tmp, f"o{i}.o") for i in range(len(sources))]

        def compile_one(i):
            return subprocess.run(common + ["-c", sources[i], "-o", objs[i]], capture_output=True, text=True)

        with ThreadPoolExecutor(max_workers=jobs) as pool:
            for i, r in enumerate(pool.map(compile_one, range(len(sources)))):
                if r.returncode != 0:
                    raise RuntimeError(f"g++ failed on {sources[i]}:\n" + r.stderr[-8000:])
        # (libamdhip64 only satisfies the inline stream helpers of stream_memory.hpp; nothing calls into it)
        r = subprocess.run([gxx] + san + objs + [shim, "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath,$ORIGIN",
                                                 "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", ORACLE_HOST_TESTS + sfx],
                           capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link of reference_host_tests_on_oracle failed:\n" + r.stderr[-8000:])
    return ORACLE_HOST_TESTS + sfx


ORACLE_DEVICE_TESTS = os.path.join(OUT_DIR, "reference_device_tests_on_oracle")


def build_reference_device_tests_on_oracle(jobs=4, sanitize=False):
    """The reference's four DEVICE test files linked to the oracle shim as well: stubs_host/thrust/device_vector.h makes
    thrust::device_vector a host vector, g++ compiles them, and the symbolic / numeric reuse family and the four-argument
    SpGEMM of those tests end in oracle_spgemm_* (tests/test_oracle_reference_tests.py).  Returns the path or None."""
    if not os.path.isdir(REF) or not os.path.isdir(REF_TEST_DIR):
        return None
    gxx, gcc = shutil.which("g++"), shutil.which("gcc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    os.makedirs(OUT_DIR, exist_ok=True)
    san = SANITIZE_FLAGS if sanitize else []
    sfx = "_asan" if sanitize else ""
    shim = os.path.join(OUT_DIR, f"liboracle_shim{sfx}.so")
    r = subprocess.run([gcc, "-O1" if sanitize else "-O2", "-fPIC", "-shared"] + san +
                       ["-I", os.path.join(ROOT, "include"), os.path.join(HERE, "oracle_shim.c"),
                        os.path.join(ROOT, "oracle", "spblas_oracle.c"), "-o", shim], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("gcc failed on oracle_shim.c:\n" + r.stderr[-4000:])
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory() as tmp:
        scratch = patched_reference_headers(os.path.join(tmp, "patched"))
        common = [gxx, "-O1", "-I", os.path.join(HERE, "stubs_host")] + san + compile_flags(scratch)
        sources = [os.path.join(REF_TEST_DIR, s) for s in REF_TEST_SOURCES] + [os.path.join(HERE, "gtest_main.cpp")]
        objs = [os.path.join(tmp, f"d{i}.o") for i in range(len(sources))]

        def compile_one(i):
            return subprocess.run(common + ["-c", sources[i], "-o", objs[i]], capture_output=True, text=True)

        with ThreadPoolExecutor(max_workers=jobs) as pool:
            for i, r in enumerate(pool.map(compile_one, range(len(sources)))):
                if r.returncode != 0:
                    raise RuntimeError(f"g++ failed on {sources[i]}:\n" + r.stderr[-8000:])
        r = subprocess.run([gxx] + san + objs + [shim, "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath,$ORIGIN",
                                                 "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", ORACLE_DEVICE_TESTS + sfx],
                           capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link of reference_device_tests_on_oracle failed:\n" + r.stderr[-8000:])
    return ORACLE_DEVICE_TESTS + sfx
