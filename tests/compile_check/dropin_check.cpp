// Compile check of the drop-in backend headers (include/spblas/vendor/gfx950/*.hpp) INSIDE the reference tree:
// <spblas/spblas.hpp> with -DSPBLAS_ENABLE_GFX950 after INTEGRATION.md section 2's edits (applied to a scratch
// copy of the six reference headers by tests/test_dropin_headers.py), then every overload the reference's callers
// would bind is instantiated on csr_view<float, int32_t, int32_t> + std::span / row-major mdspan.
// g++ -fsyntax-only: nothing is linked or run.
#include <cstdint>
#include <span>

#include <spblas/spblas.hpp>

#include <spblas/algorithms/transposed.hpp>  // not pulled in by spblas.hpp when a vendor backend is selected

using T = float;
using I = spblas::index_t;
using O = spblas::offset_t;
static_assert(std::is_same_v<I, std::int32_t> && std::is_same_v<O, std::int32_t>,
              "index_types.hpp must make index_t = offset_t = int32_t");

void dropin_instantiations(spblas::csr_view<T, I, O> a, spblas::csr_view<T, I, O> b, spblas::csr_view<T, I, O> c,
                           spblas::csr_view<T, I, O> d, spblas::csc_view<T, I, O> a_csc, std::span<T> x, std::span<T> y,
                           spblas::mdspan_row_major<T, I> B, spblas::mdspan_row_major<T, I> C) {
  using namespace spblas;
  // SpMV: vendor/rocsparse/detail/spmv_impl.hpp:18-90 call shapes + multiply_inspect (onemkl_sycl/spmv_impl.hpp:100-108)
  multiply(a, x, y);
  multiply(scaled(2.0f, a), x, y);
  multiply(a, scaled(2.0f, x), y);
  multiply(a_csc, x, y);
  multiply(transposed(a), x, y);
  operation_info_t info = multiply_inspect(a, x, y);
  multiply_inspect(info, a, x, y);
  multiply(info, a, x, y);
  matrix_opt a_opt(a);
  operation_info_t info_opt = multiply_inspect(a_opt, x, y);
  multiply(info_opt, a_opt, x, y);
  // SpMM: vendor/onemkl_sycl/spmm_impl.hpp:133-198 call shapes (examples/spmm_csr.cpp:45-46)
  multiply(a, B, C);
  multiply(scaled(3.0f, a), B, C);
  operation_info_t info_mm = multiply_inspect(a, B, C);
  multiply(info_mm, a, B, C);
  multiply(a_csc, B, C);  // csc_view A (test/gtest/spmm_test.cpp:181)
  // SpGEMM: vendor/rocsparse/multiply_spgemm.hpp:232-317
  operation_info_t info_g = multiply_compute(a, b, c);
  multiply_compute(info_g, a, b, c);
  multiply_fill(info_g, a, b, c);
  spgemm_state_t state;
  multiply_compute(state, a, b, c);
  multiply_fill(state, a, b, c);
  multiply_symbolic_compute(state, a, b, c);
  multiply_symbolic_fill(state, a, b, c);
  multiply_numeric(state, scaled(2.0f, a), b, c);
  // CSR / CSC operand combinations (test/gtest/spgemm_test.cpp:205, spgemm_csr_csc.cpp)
  spblas::csc_view<T, I, O> c_csc(static_cast<T*>(nullptr), static_cast<O*>(nullptr), static_cast<I*>(nullptr), {0, 0}, 0);
  operation_info_t info_gc = multiply_compute(a_csc, a_csc, c_csc);
  multiply_fill(info_gc, a_csc, a_csc, c_csc);
  operation_info_t info_gm = multiply_compute(a, a_csc, c);
  multiply_fill(info_gm, a, a_csc, c);
  // four-argument forms  C = alpha*A*B + beta*D  (multiply_spgemm.hpp:237-274)
  multiply_compute(state, a, b, c, d);
  multiply_fill(state, a, b, c, d);
  multiply_symbolic_compute(state, a, b, c, d);
  multiply_symbolic_fill(state, a, b, c, d);
  multiply_numeric(state, a, b, c, scaled(0.5f, d));
  // add / transpose / scale / triangular_solve
  operation_info_t info_a = add_inspect(a, b, c);
  add_compute(info_a, a, b, c);
  add(a, b, c);
  transpose(a, b);
  operation_info_t info_t = transpose_inspect(a, b);
  transpose(info_t, a, b);
  scale(2.0f, a);
  scale(2.0f, x);
  operation_info_t info_s = triangular_solve_inspect(a, lower_triangle_t{}, explicit_diagonal_t{}, x, y);
  triangular_solve(info_s, a, lower_triangle_t{}, explicit_diagonal_t{}, x, y);
  triangular_solve(a, upper_triangle_t{}, implicit_unit_diagonal_t{}, x, y);
}
