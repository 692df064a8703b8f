"""CPU tests: the C-ABI library loads and exports every symbol include/spblas_gfx950.h
declares, and the ctypes prototypes cover exactly that set (no compute calls here)."""
import ctypes
import os
import re

import spblas_reference_amd as sp
from spblas_reference_amd import _build, _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "spblas_gfx950.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spblas_gfx950_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_in_tree():
    _build.build()  # no-op when up to date; hipcc cross-compiles gfx950 without a GPU
    assert os.path.exists(_capi.library_path())
    assert os.path.commonpath([ROOT, _capi.library_path()]) == ROOT


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 17
    dll = ctypes.CDLL(_capi.library_path())
    for n in names:
        assert hasattr(dll, n), f"{n} declared in the header but not exported"
    bound = sorted(n for n, _, _ in _capi.PROTOTYPES)
    assert bound == names, "ctypes prototypes and header declarations differ"


def test_status_strings_and_version():
    lib = _capi.lib()
    assert lib.spblas_gfx950_version() >= 100
    assert lib.spblas_gfx950_status_string(0) == b"success"
    assert b"out of memory" in lib.spblas_gfx950_status_string(_capi.INSUFFICIENT_SPACE)
    assert b"incompatible" in lib.spblas_gfx950_status_string(_capi.INVALID_SIZE)


def test_null_handle_is_rejected_without_touching_a_gpu():
    lib = _capi.lib()
    one = ctypes.c_float(1)
    assert lib.spblas_gfx950_spmv(None, None, 0, 1, 1, 0, ctypes.byref(one), None, None, None, None,
                                  ctypes.byref(one), None, 0, 0) == _capi.INVALID_HANDLE
    assert lib.spblas_gfx950_destroy(None) == _capi.INVALID_HANDLE
    assert lib.spblas_gfx950_plan_info(None, None) == _capi.INVALID_POINTER


def test_package_surface_mirrors_reference_names():
    for name in ["csr_view", "csc_view", "scaled", "conjugated", "transposed", "matrix_opt", "operation_info_t",
                 "spgemm_state_t", "multiply", "multiply_inspect", "multiply_compute", "multiply_fill",
                 "multiply_symbolic_compute", "multiply_symbolic_fill", "multiply_numeric"]:
        assert hasattr(sp, name)


def test_cpp_host_layer_compiles_with_gxx():
    """The C++20 host layer (standalone API mirror + shared __gfx950 call layer) builds with g++
    and links against the C-ABI library -- no GPU needed to compile."""
    exe = _build.build_cpp_tests()
    assert os.path.exists(exe)
