"""ctypes binding of the C ABI in include/spblas_gfx950.h.

This is the ONLY route from the Python host layer to compute: there is no CPU or
torch fallback.  If the HIP library is missing or fails to load, every operation
raises (the product path must fail loudly, never degrade silently).
"""
import ctypes
import os

from . import _build

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_i64 = ctypes.c_int64

SUCCESS = 0
INVALID_HANDLE = 1
INVALID_POINTER = 2
INVALID_SIZE = 3
INVALID_VALUE = 4
NOT_SUPPORTED = 5
ALLOC_FAILED = 6
HIP_ERROR = 7
INSUFFICIENT_SPACE = 8
PLAN_MISMATCH = 9

F32, F64 = 0, 1
I32, I64 = 0, 1
OP_N, OP_T = 0, 1
LOWER, UPPER = 0, 1
DIAG_EXPLICIT, DIAG_UNIT = 0, 1
SPMV_AUTO, SPMV_VECTOR, SPMV_ROWBLOCK, SPMV_SLICED = 0, 1, 2, 3
OPT_BIN_ROW_ALIGN = 1
OPT_MAX_KSPLIT = 2
OPT_VALUE_SNAPSHOT = 3
OPT_SPGEMM_KEEP_COLIND = 4
OPT_STORE_TRIAL = 5

# every symbol include/spblas_gfx950.h declares: (name, restype, argtypes)
PROTOTYPES = [
    ("spblas_gfx950_version", c_int, []),
    ("spblas_gfx950_status_string", ctypes.c_char_p, [c_int]),
    ("spblas_gfx950_last_hip_error", c_int, []),
    ("spblas_gfx950_create", c_int, [ctypes.POINTER(c_void_p), c_void_p]),
    ("spblas_gfx950_destroy", c_int, [c_void_p]),
    ("spblas_gfx950_set_stream", c_int, [c_void_p, c_void_p]),
    ("spblas_gfx950_get_stream", c_int, [c_void_p, ctypes.POINTER(c_void_p)]),
    ("spblas_gfx950_set_option", c_int, [c_void_p, c_int, c_i64]),
    ("spblas_gfx950_spmv_expand", c_int, [c_void_p, c_void_p, c_void_p]),
    ("spblas_gfx950_spmv_reduce_rows", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64]),
    ("spblas_gfx950_spmv_plan_create", c_int,
     [c_void_p, ctypes.POINTER(c_void_p), c_i64, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    ("spblas_gfx950_spmv_plan_update_values", c_int, [c_void_p, c_void_p, c_void_p]),
    ("spblas_gfx950_spmv_plan_detach", c_int, [c_void_p, c_void_p]),
    ("spblas_gfx950_plan_destroy", c_int, [c_void_p, c_void_p]),
    ("spblas_gfx950_plan_info", c_int, [c_void_p, ctypes.POINTER(c_i64)]),
    ("spblas_gfx950_plan_info_sliced", c_int, [c_void_p, ctypes.POINTER(c_i64)]),
    ("spblas_gfx950_plan_info_hot", c_int, [c_void_p, ctypes.POINTER(c_i64)]),
    ("spblas_gfx950_spmv", c_int,
     [c_void_p, c_void_p, c_int, c_i64, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
      c_void_p, c_int, c_int]),
    ("spblas_gfx950_spmm", c_int,
     [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64,
      c_void_p, c_void_p, c_i64, c_int, c_int]),
    ("spblas_gfx950_spmm_strided", c_int,
     [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64,
      c_void_p, c_void_p, c_i64, c_i64, c_int, c_int]),
    ("spblas_gfx950_spmm_inspect", c_int, [c_void_p, c_void_p]),
    ("spblas_gfx950_spmm_plan_info", c_int, [c_void_p, ctypes.POINTER(c_i64)]),
    ("spblas_gfx950_csr_transpose", c_int,
     [c_void_p, c_i64, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    ("spblas_gfx950_scale", c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_int]),
    ("spblas_gfx950_narrow_indices", c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_i64]),
    ("spblas_gfx950_spgemm_create", c_int, [c_void_p, ctypes.POINTER(c_void_p)]),
    ("spblas_gfx950_spgemm_destroy", c_int, [c_void_p, c_void_p]),
    ("spblas_gfx950_spgemm_info", c_int, [c_void_p, ctypes.POINTER(ctypes.c_int64)]),
    ("spblas_gfx950_spgemm_symbolic", c_int,
     [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_i64, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p,
      ctypes.POINTER(c_i64)]),
    ("spblas_gfx950_spgemm_numeric", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
      c_void_p, c_i64, c_int]),
    ("spblas_gfx950_csr_add_symbolic", c_int,
     [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p,
      ctypes.POINTER(c_i64)]),
    ("spblas_gfx950_csr_add_numeric", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
      c_void_p, c_void_p, c_i64, c_int]),
    ("spblas_gfx950_ipc_alloc", c_int, [ctypes.c_size_t, c_int, ctypes.POINTER(c_void_p)]),
    ("spblas_gfx950_ipc_free", c_int, [c_void_p]),
    ("spblas_gfx950_ipc_export", c_int, [c_void_p, c_void_p]),
    ("spblas_gfx950_ipc_open", c_int, [c_void_p, ctypes.POINTER(c_void_p)]),
    ("spblas_gfx950_ipc_close", c_int, [c_void_p]),
    ("spblas_gfx950_spmv_reduce_rows_bcast", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_i64, c_i64]),
    ("spblas_gfx950_spmv_step_bcast", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int]),
    ("spblas_gfx950_spmv_chunk_rows", c_int, [c_void_p, c_int, ctypes.POINTER(c_i64)]),
    ("spblas_gfx950_spmv_step_bcast_chunked", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_void_p, c_int, c_i64, c_void_p]),
    ("spblas_gfx950_step_signal", c_int, [c_void_p, c_void_p, c_int, c_int, c_i64]),
    ("spblas_gfx950_step_wait", c_int, [c_void_p, c_void_p, c_int, c_i64, c_i64, c_void_p]),
    ("spblas_gfx950_wall_clock_khz", c_int, [c_void_p, ctypes.POINTER(c_int)]),
    ("spblas_gfx950_bcast_wait_before", c_int, [c_void_p, c_void_p, c_int, c_i64, c_i64, c_void_p]),
    ("spblas_gfx950_sptrsv_create", c_int,
     [c_void_p, ctypes.POINTER(c_void_p), c_i64, c_i64, c_void_p, c_void_p, c_int, c_int]),
    ("spblas_gfx950_sptrsv_destroy", c_int, [c_void_p, c_void_p]),
    ("spblas_gfx950_sptrsv_info", c_int, [c_void_p, ctypes.POINTER(c_i64)]),
    ("spblas_gfx950_sptrsv_status", c_int, [c_void_p, c_void_p, ctypes.POINTER(c_int)]),
    ("spblas_gfx950_sptrsv_solve", c_int,
     [c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    ("spblas_gfx950_spgemm_set_addend", c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
    ("spblas_gfx950_spgemm_numeric_addend", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int]),
]

_LIB = None


class BackendError(RuntimeError):
    """Non-success status from the gfx950 library (std::runtime_error in the C++ layer)."""

    def __init__(self, status, where):
        self.status = status
        msg = lib().spblas_gfx950_status_string(status).decode()
        if status == HIP_ERROR:
            msg += f" (hipError_t {lib().spblas_gfx950_last_hip_error()})"
        super().__init__(f"{where}: {msg}")


def library_path():
    # SPBLAS_GFX950_LIB: another build of the same library (tools/build_variant.sh: same-box A/B measurements only)
    return os.environ.get("SPBLAS_GFX950_LIB") or _build.LIBPATH


class chunk_wait(ctypes.Structure):
    """spblas_gfx950_chunk_wait (include/spblas_gfx950.h)."""
    _fields_ = [("flags", c_void_p), ("chunk_rows", c_void_p), ("n_ranks", c_int), ("chunks", c_int), ("step", c_i64),
                ("timeout_ms", c_i64), ("status_dev", c_void_p), ("max_expand_workgroups", c_int)]


def lib():
    """Load (never build silently on a GPU box: build() is explicit) the C-ABI library."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: the gfx950 HIP backend is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                "There is no CPU fallback.")
        dll = ctypes.CDLL(path)
        for name, restype, argtypes in PROTOTYPES:
            fn = getattr(dll, name)  # AttributeError = ABI drift, surface it
            fn.restype = restype
            fn.argtypes = argtypes
        _LIB = dll
    return _LIB


def check(status, where):
    """Map a status to the exception type the reference throws at this boundary
    (SURVEY.md section 8b 'Error convention')."""
    if status == SUCCESS:
        return
    if status == INVALID_SIZE:
        raise ValueError(f"{where}: matrix dimensions are incompatible.")  # std::invalid_argument
    if status == ALLOC_FAILED:
        raise MemoryError(f"{where}: device allocation failed")  # std::bad_alloc
    raise BackendError(status, where)
