"""spblas-reference_amd: MI355X (gfx950) backend for the spblas-reference multiply() path.

Layout (only what the hot path needs):
  csrc/      hand-written HIP kernels + the C ABI (include/spblas_gfx950.h)
  _capi.py   ctypes binding of that ABI (no fallback path)
  api.py     host-side mirror of the reference's view/operator interface
  sharded.py row-sharded multi-GPU SpMV (one process per GPU, RCCL all-gather of y)
  generate.py synthetic inputs
Import name: `spblas_reference_amd` (see the shim module at the repo root).
"""
from . import _build, _capi, generate  # noqa: F401
from ._capi import BackendError  # noqa: F401
from .api import (  # noqa: F401
    add, add_compute, add_inspect, conjugated, csc_view, csr_view, get_scaling_factor, get_ultimate_base, has_matrix_opt, index, is_conjugated,
    matrix_opt, multiply, multiply_compute, multiply_fill, multiply_inspect, multiply_numeric,
    multiply_symbolic_compute, multiply_symbolic_fill, operation_info_t, prepared_multiply, scale, scaled, scaled_view,
    spgemm_state_t, transpose, transpose_inspect,
    transposed, triangular_solve, triangular_solve_inspect, upper_triangle_t, lower_triangle_t,
    implicit_unit_diagonal_t, explicit_diagonal_t, upper_triangle, lower_triangle, implicit_unit_diagonal,
    explicit_diagonal)
