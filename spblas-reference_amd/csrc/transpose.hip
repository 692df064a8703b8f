// Device CSR -> CSR transpose:  B = A^T  (SURVEY.md section 8f ranks 1-2).
//
// Result identical to the reference's counting sort
// (/root/reference/include/spblas/algorithms/transpose_impl.hpp:14-53): entries of every output row
// are in source order (ascending original row, ties in storage order), i.e. a STABLE sort of the
// entries by column.  Steps, all on the handle's stream:
//   1. payload of every entry = (row id, value)                    -- spt_payload_kernel (expand rowptr)
//   2. stable LSD radix sort of (column, payload) pairs            -- rocprim::radix_sort_pairs
//   3. t_rowptr[j] = first sorted position with column >= j        -- spt_rowptr_kernel
//   4. t_colind[k], t_values[k] = sorted payload k                 -- spt_split_kernel (streaming)
// The payload travels with the key, so no pass gathers from a permutation (the first version sorted
// (column, position) pairs and gathered rows / values afterwards: 3.8 of 6.6 ms at 1e8 entries were that
// random gather).  The device-wide radix sort is the one generic primitive taken from rocPRIM
// (header-only, part of ROCm); it runs at inspect time only.  Everything on the multiply() hot path is
// hand-written.  Used by multiply_inspect on csc_view / transposed(csr) operands, which then run the
// regular (row-block or LDS-sliced) SpMV kernels on the materialised transpose instead of the atomic
// scatter kernel.
#include <cstring>

#include "common.hpp"

#include <rocprim/rocprim.hpp>

namespace spb {

template <typename T>
struct spt_payload {
  int32_t row;
  T val;
};

template <typename T>
__global__ __launch_bounds__(256) void spt_payload_kernel(int64_t m, const int32_t* __restrict__ rowptr,
                                                          const T* __restrict__ values,
                                                          spt_payload<T>* __restrict__ out) {
  const int64_t row = (int64_t) blockIdx.x * 32 + threadIdx.x / 8;
  const int lane = threadIdx.x % 8;
  if (row >= m)
    return;
  for (int p = rowptr[row] + lane; p < rowptr[row + 1]; p += 8) {
    spt_payload<T> e;
    e.row = (int32_t) row;
    e.val = values[p];
    out[p] = e;
  }
}

__global__ __launch_bounds__(256) void spt_rowptr_kernel(int64_t n, int64_t nnz, const int32_t* __restrict__ sorted_cols,
                                                         int32_t* __restrict__ t_rowptr) {
  const int64_t j = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (j > n)
    return;
  int64_t lo = 0, hi = nnz;  // first k with sorted_cols[k] >= j
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted_cols[mid] < j)
      lo = mid + 1;
    else
      hi = mid;
  }
  t_rowptr[j] = (int32_t) lo;
}

template <typename T>
__global__ __launch_bounds__(256) void spt_split_kernel(int64_t nnz, const spt_payload<T>* __restrict__ sorted,
                                                        int32_t* __restrict__ t_colind, T* __restrict__ t_values) {
  const int64_t k = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (k >= nnz)
    return;
  const spt_payload<T> e = sorted[k];
  t_colind[k] = e.row;
  t_values[k] = e.val;
}

template <typename T>
static int transpose_typed(spblas_gfx950_handle_t handle, int64_t m, int64_t n, int64_t nnz, const int32_t* rowptr,
                           const int32_t* colind, const T* values, int32_t* t_rowptr, int32_t* t_colind,
                           T* t_values) {
  hipStream_t s = handle->stream;
  using P = spt_payload<T>;
  int bits = 1;
  while (bits < 32 && ((int64_t) 1 << bits) < n)
    ++bits;
  // scratch: payload in / out, sorted keys and the sort's temporary storage, carved out of the handle's
  // grow-only buffer (no allocation on repeated calls, nothing handed back to an allocator while
  // kernels may still be using it)
  size_t tmp_bytes = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, colind, (int32_t*) nullptr, (P*) nullptr, (P*) nullptr,
                                           (size_t) nnz, 0, bits, s);
  if (e != hipSuccess)
    return hip_fail(e);
  const size_t pay = (((size_t) nnz * sizeof(P)) + 255) & ~(size_t) 255;
  const size_t keys = (((size_t) nnz * 4) + 255) & ~(size_t) 255;
  void* base = nullptr;
  int rc = handle_scratch(handle, 2 * pay + keys + tmp_bytes + 256, &base);
  if (rc)
    return rc;
  char* bp = static_cast<char*>(base);
  P* pay_in = reinterpret_cast<P*>(bp);
  P* pay_out = reinterpret_cast<P*>(bp + pay);
  int32_t* sorted_cols = reinterpret_cast<int32_t*>(bp + 2 * pay);
  void* tmp = bp + 2 * pay + keys;
  hipLaunchKernelGGL((spt_payload_kernel<T>), dim3((unsigned) cdiv(m, 32)), dim3(256), 0, s, m, rowptr, values, pay_in);
  e = rocprim::radix_sort_pairs(tmp, tmp_bytes, colind, sorted_cols, pay_in, pay_out, (size_t) nnz, 0, bits, s);
  if (e != hipSuccess)
    return hip_fail(e);
  hipLaunchKernelGGL(spt_rowptr_kernel, dim3((unsigned) cdiv(n + 1, 256)), dim3(256), 0, s, n, nnz, sorted_cols, t_rowptr);
  hipLaunchKernelGGL((spt_split_kernel<T>), dim3((unsigned) cdiv(nnz, 256)), dim3(256), 0, s, nnz, pay_out, t_colind,
                     t_values);
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

// scale(alpha, t): 16-byte accesses over the aligned body, scalar head and tail
template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(int64_t n, T alpha, T* __restrict__ v) {
  constexpr int V = 16 / (int) sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(V)));
  const int64_t head = ((16 - (reinterpret_cast<uintptr_t>(v) & 15)) & 15) / sizeof(T);
  const int64_t h = head < n ? head : n;
  const int64_t nv = (n - h) / V;
  const int64_t tid = (int64_t) blockIdx.x * 256 + threadIdx.x, stride = (int64_t) gridDim.x * 256;
  vec_t* body = reinterpret_cast<vec_t*>(v + h);
  for (int64_t i = tid; i < nv; i += stride)
    body[i] = body[i] * alpha;
  if (tid < h)
    v[tid] *= alpha;
  const int64_t t0 = h + nv * V;
  if (t0 + tid < n && tid < V)
    v[t0 + tid] *= alpha;
}

} // namespace spb

using namespace spb;

extern "C" int spblas_gfx950_csr_transpose(spblas_gfx950_handle_t handle, int64_t m, int64_t n, int64_t nnz,
                                           const int32_t* rowptr, const int32_t* colind, const void* values,
                                           int32_t* t_rowptr, int32_t* t_colind, void* t_values, int value_type) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (m < 0 || n < 0 || nnz < 0 || m > INT32_MAX || n >= INT32_MAX || nnz > INT32_MAX)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  if (value_type != SPBLAS_GFX950_F32 && value_type != SPBLAS_GFX950_F64)
    return SPBLAS_GFX950_STATUS_INVALID_VALUE;
  if (!rowptr || !t_rowptr || (nnz > 0 && (!colind || !values || !t_colind || !t_values)))
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (nnz == 0) {
    SPB_HIP(hipMemsetAsync(t_rowptr, 0, (size_t) (n + 1) * 4, handle->stream));
    return SPBLAS_GFX950_STATUS_SUCCESS;
  }
  if (value_type == SPBLAS_GFX950_F32)
    return transpose_typed<float>(handle, m, n, nnz, rowptr, colind, static_cast<const float*>(values), t_rowptr,
                                  t_colind, static_cast<float*>(t_values));
  return transpose_typed<double>(handle, m, n, nnz, rowptr, colind, static_cast<const double*>(values), t_rowptr,
                                 t_colind, static_cast<double*>(t_values));
}

extern "C" int spblas_gfx950_scale(spblas_gfx950_handle_t handle, int64_t n, const void* alpha, void* values,
                                   int value_type) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (n < 0)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  if (value_type != SPBLAS_GFX950_F32 && value_type != SPBLAS_GFX950_F64)
    return SPBLAS_GFX950_STATUS_INVALID_VALUE;
  if (!alpha || (n > 0 && !values))
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (n == 0)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  const int64_t per = value_type == SPBLAS_GFX950_F32 ? 4 : 2;
  int64_t blocks = cdiv(cdiv(n, per), 256);
  const int64_t cap = (int64_t) (handle->num_cus > 0 ? handle->num_cus : 256) * 16;
  if (blocks > cap)
    blocks = cap;
  if (blocks < 1)
    blocks = 1;
  if (value_type == SPBLAS_GFX950_F32)
    hipLaunchKernelGGL(scale_kernel<float>, dim3((unsigned) blocks), dim3(256), 0, handle->stream, n,
                       *static_cast<const float*>(alpha), static_cast<float*>(values));
  else
    hipLaunchKernelGGL(scale_kernel<double>, dim3((unsigned) blocks), dim3(256), 0, handle->stream, n,
                       *static_cast<const double*>(alpha), static_cast<double*>(values));
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}
