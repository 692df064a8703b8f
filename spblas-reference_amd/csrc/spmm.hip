// CSR x dense SpMM for gfx950:  C = alpha * A * B + beta * C, B (k x n) and C (m x n)
// row-major.
//
// The rocSPARSE slot of the reference has no SpMM; the device call site this
// replaces is oneapi::mkl::sparse::gemm at
// /root/reference/include/spblas/vendor/onemkl_sycl/spmm_impl.hpp:116-120 and the
// maths is the CPU path include/spblas/algorithms/multiply_impl.hpp:66-92.
//
// spmm_rowgroup_kernel: a group of G lanes (power of two) owns one row of A and a
// panel of G*V columns of B/C, V = elements per 16-byte (or narrower) lane access.
// The group loads G (colind, value) pairs with one coalesced streaming access,
// broadcasts them lane by lane and gathers whole B rows: every gather is one
// contiguous G*V*sizeof(T)-byte segment (512 B for n = 128, fp32, V = 4, G = 32).
// HBM/L2-bound on the B gathers; algorithmic bytes = nnz*(sizeof(T)+4) +
// (m+1)*sizeof(O) + (k*n + m*n)*sizeof(T).
#include "common.hpp"
#include "plan.hpp"

namespace spb {

template <typename T, int V>
struct vec_of;
template <>
struct vec_of<float, 1> { typedef float type; };
template <>
struct vec_of<float, 2> { typedef f32x2 type; };
template <>
struct vec_of<float, 4> { typedef f32x4 type; };
template <>
struct vec_of<double, 1> { typedef double type; };
template <>
struct vec_of<double, 2> { typedef f64x2 type; };

template <typename T, int V>
__device__ __forceinline__ void fma_vec(T (&acc)[V], T a, const typename vec_of<T, V>::type& b) {
  if constexpr (V == 1) {
    acc[0] += a * b;
  } else {
#pragma unroll
    for (int i = 0; i < V; ++i)
      acc[i] += a * b[i];
  }
}

template <typename T, typename O, int V>
__global__ __launch_bounds__(256) void spmm_rowgroup_kernel(int64_t m, int64_t n, const O* __restrict__ rowptr,
                                                            const int32_t* __restrict__ colind,
                                                            const T* __restrict__ values,
                                                            const T* __restrict__ B, int64_t ldb,
                                                            T* __restrict__ C, int64_t ldc, T alpha, T beta,
                                                            int G) {
  typedef typename vec_of<T, V>::type vec_t;
  const int rows_per_block = 256 / G;
  const int64_t row = (int64_t) blockIdx.x * rows_per_block + threadIdx.x / G;
  const int lig = threadIdx.x % G;
  const int64_t panel_cols = (int64_t) G * V;
  O p0 = 0, p1 = 0;
  if (row < m) {
    p0 = rowptr[row];
    p1 = rowptr[row + 1];
  }
  for (int64_t col0 = (int64_t) lig * V; col0 - (int64_t) lig * V < n; col0 += panel_cols) {
    const bool active = row < m && col0 < n;
    T acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i)
      acc[i] = T(0);
    const T* __restrict__ Bc = B + col0;
    for (O base = p0; base < p1; base += G) {
      int32_t c = 0;
      T v = T(0);
      if (base + lig < p1) {
        c = stream_load(colind + base + lig);
        v = stream_load(values + base + lig);
      }
      const int cnt = (int) ((p1 - base) < (O) G ? (p1 - base) : (O) G);
      int j = 0;
      for (; j + 4 <= cnt; j += 4) {
        const int64_t k0 = __shfl(c, j, G), k1 = __shfl(c, j + 1, G), k2 = __shfl(c, j + 2, G),
                      k3 = __shfl(c, j + 3, G);
        const T a0 = __shfl(v, j, G), a1 = __shfl(v, j + 1, G), a2 = __shfl(v, j + 2, G),
                a3 = __shfl(v, j + 3, G);
        if (active) {
          const vec_t b0 = *reinterpret_cast<const vec_t*>(Bc + k0 * ldb);
          const vec_t b1 = *reinterpret_cast<const vec_t*>(Bc + k1 * ldb);
          const vec_t b2 = *reinterpret_cast<const vec_t*>(Bc + k2 * ldb);
          const vec_t b3 = *reinterpret_cast<const vec_t*>(Bc + k3 * ldb);
          fma_vec<T, V>(acc, a0, b0);
          fma_vec<T, V>(acc, a1, b1);
          fma_vec<T, V>(acc, a2, b2);
          fma_vec<T, V>(acc, a3, b3);
        }
      }
      for (; j < cnt; ++j) {
        const int64_t k0 = __shfl(c, j, G);
        const T a0 = __shfl(v, j, G);
        if (active) {
          const vec_t b0 = *reinterpret_cast<const vec_t*>(Bc + k0 * ldb);
          fma_vec<T, V>(acc, a0, b0);
        }
      }
    }
    if (active) {
      T* cp = C + row * ldc + col0;
      if constexpr (V == 1) {
        cp[0] = beta == T(0) ? alpha * acc[0] : alpha * acc[0] + beta * cp[0];
      } else {
        vec_t out;
        if (beta == T(0)) {
#pragma unroll
          for (int i = 0; i < V; ++i)
            out[i] = alpha * acc[i];
        } else {
          const vec_t old = *reinterpret_cast<const vec_t*>(cp);
#pragma unroll
          for (int i = 0; i < V; ++i)
            out[i] = alpha * acc[i] + beta * old[i];
        }
        *reinterpret_cast<vec_t*>(cp) = out;
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void scale_matrix_kernel(int64_t m, int64_t n, T* __restrict__ C, int64_t ldc,
                                                           T beta) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i < m * n) {
    const int64_t r = i / n, c = i % n;
    T* p = C + r * ldc + c;
    *p = beta == T(0) ? T(0) : beta * *p;
  }
}

template <typename T, typename O, int V>
static void launch_spmm(hipStream_t s, int64_t m, int64_t n, const O* rowptr, const int32_t* colind,
                        const T* values, const T* B, int64_t ldb, T* C, int64_t ldc, T alpha, T beta) {
  int G = 1;
  while (G < 64 && (int64_t) G * V < n)
    G <<= 1;
  const int rows_per_block = 256 / G;
  hipLaunchKernelGGL((spmm_rowgroup_kernel<T, O, V>), dim3((unsigned) cdiv(m, rows_per_block)), dim3(256), 0, s,
                     m, n, rowptr, colind, values, B, ldb, C, ldc, alpha, beta, G);
}

template <typename T, typename O>
static int spmm_typed(spblas_gfx950_handle_t h, int64_t m, int64_t k, int64_t n, int64_t nnz,
                      const void* alpha_p, const void* rowptr_p, const int32_t* colind, const void* values_p,
                      const void* B_p, int64_t ldb, const void* beta_p, void* C_p, int64_t ldc) {
  const T alpha = *static_cast<const T*>(alpha_p);
  const T beta = *static_cast<const T*>(beta_p);
  const O* rowptr = static_cast<const O*>(rowptr_p);
  const T* values = static_cast<const T*>(values_p);
  const T* B = static_cast<const T*>(B_p);
  T* C = static_cast<T*>(C_p);
  hipStream_t s = h->stream;
  if (m == 0 || n == 0)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  if (nnz == 0) {
    hipLaunchKernelGGL((scale_matrix_kernel<T>), dim3((unsigned) cdiv(m * n, 256)), dim3(256), 0, s, m, n, C,
                       ldc, beta);
    SPB_HIP(hipGetLastError());
    return SPBLAS_GFX950_STATUS_SUCCESS;
  }
  // widest lane access (<= 16 B) that divides n and keeps every row segment aligned
  constexpr int VMAX = 16 / sizeof(T);
  const uintptr_t bits = (uintptr_t) B | (uintptr_t) C;
  int V = 1;
  for (int cand = VMAX; cand > 1; cand >>= 1) {
    const size_t bytes = cand * sizeof(T);
    if (n % cand == 0 && ldb % cand == 0 && ldc % cand == 0 && (bits % bytes) == 0) {
      V = cand;
      break;
    }
  }
  if constexpr (sizeof(T) == 4) {
    if (V == 4)
      launch_spmm<T, O, 4>(s, m, n, rowptr, colind, values, B, ldb, C, ldc, alpha, beta);
    else if (V == 2)
      launch_spmm<T, O, 2>(s, m, n, rowptr, colind, values, B, ldb, C, ldc, alpha, beta);
    else
      launch_spmm<T, O, 1>(s, m, n, rowptr, colind, values, B, ldb, C, ldc, alpha, beta);
  } else {
    if (V == 2)
      launch_spmm<T, O, 2>(s, m, n, rowptr, colind, values, B, ldb, C, ldc, alpha, beta);
    else
      launch_spmm<T, O, 1>(s, m, n, rowptr, colind, values, B, ldb, C, ldc, alpha, beta);
  }
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

} // namespace spb

using namespace spb;

extern "C" int spblas_gfx950_spmm(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, int64_t m, int64_t k,
                                  int64_t n, int64_t nnz, const void* alpha, const void* rowptr,
                                  const int32_t* colind, const void* values, const void* B, int64_t ldb,
                                  const void* beta, void* C, int64_t ldc, int offset_type, int value_type) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (m < 0 || k < 0 || n < 0 || nnz < 0 || m > INT32_MAX || k > INT32_MAX || ldb < n || ldc < n)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  if (offset_type == SPBLAS_GFX950_I32 && nnz > INT32_MAX)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  if ((offset_type != SPBLAS_GFX950_I32 && offset_type != SPBLAS_GFX950_I64) ||
      (value_type != SPBLAS_GFX950_F32 && value_type != SPBLAS_GFX950_F64))
    return SPBLAS_GFX950_STATUS_INVALID_VALUE;
  if (!alpha || !beta || !rowptr || (nnz > 0 && (!colind || !values || !B)) || (m > 0 && n > 0 && !C))
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (plan && (plan->m != m || plan->n != k || plan->nnz != nnz || plan->rowptr != rowptr ||
               plan->colind != colind || plan->offset_type != offset_type))
    return SPBLAS_GFX950_STATUS_PLAN_MISMATCH;
  if (value_type == SPBLAS_GFX950_F32) {
    return offset_type == SPBLAS_GFX950_I32
               ? spmm_typed<float, int32_t>(handle, m, k, n, nnz, alpha, rowptr, colind, values, B, ldb, beta, C, ldc)
               : spmm_typed<float, int64_t>(handle, m, k, n, nnz, alpha, rowptr, colind, values, B, ldb, beta, C, ldc);
  }
  return offset_type == SPBLAS_GFX950_I32
             ? spmm_typed<double, int32_t>(handle, m, k, n, nnz, alpha, rowptr, colind, values, B, ldb, beta, C, ldc)
             : spmm_typed<double, int64_t>(handle, m, k, n, nnz, alpha, rowptr, colind, values, B, ldb, beta, C, ldc);
}
