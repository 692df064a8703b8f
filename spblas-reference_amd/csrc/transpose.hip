// Device CSR -> CSR transpose:  B = A^T  (SURVEY.md section 8f ranks 1-2).
//
// Result identical to the reference's counting sort
// (/root/reference/include/spblas/algorithms/transpose_impl.hpp:14-53): entries of every output row
// are in source order (ascending original row, ties in storage order), i.e. a STABLE sort of the
// entries by column.  Steps, all on the handle's stream:
//   1. row id of every entry (expand rowptr)                       -- spt_rowid_kernel
//   2. stable LSD radix sort of (column, source position) pairs    -- rocprim::radix_sort_pairs
//   3. t_rowptr[j] = first sorted position with column >= j        -- spt_rowptr_kernel
//   4. t_colind[k] = rowid[perm[k]],  t_values[k] = values[perm[k]] -- spt_gather_kernel
// The device-wide radix sort is the one generic primitive taken from rocPRIM (header-only, part of
// ROCm); it runs at inspect time only.  Everything on the multiply() hot path is hand-written.
// Used by multiply_inspect on csc_view / transposed(csr) operands, which then run the regular
// (row-block or LDS-sliced) SpMV kernels on the materialised transpose instead of the atomic
// scatter kernel.
#include <cstring>

#include "common.hpp"

#include <rocprim/rocprim.hpp>

namespace spb {

__global__ __launch_bounds__(256) void spt_rowid_kernel(int64_t m, const int32_t* __restrict__ rowptr,
                                                        int32_t* __restrict__ rowid, int32_t* __restrict__ pos) {
  const int64_t row = (int64_t) blockIdx.x * 32 + threadIdx.x / 8;
  const int lane = threadIdx.x % 8;
  if (row >= m)
    return;
  for (int p = rowptr[row] + lane; p < rowptr[row + 1]; p += 8) {
    rowid[p] = (int32_t) row;
    pos[p] = p;
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
__global__ __launch_bounds__(256) void spt_gather_kernel(int64_t nnz, const int32_t* __restrict__ perm,
                                                         const int32_t* __restrict__ rowid,
                                                         const T* __restrict__ values, int32_t* __restrict__ t_colind,
                                                         T* __restrict__ t_values) {
  const int64_t k = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (k >= nnz)
    return;
  const int p = perm[k];
  t_colind[k] = rowid[p];
  t_values[k] = values[p];
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
  hipStream_t s = handle->stream;
  if (nnz == 0) {
    SPB_HIP(hipMemsetAsync(t_rowptr, 0, (size_t) (n + 1) * 4, s));
    return SPBLAS_GFX950_STATUS_SUCCESS;
  }
  int bits = 1;
  while (bits < 32 && ((int64_t) 1 << bits) < n)
    ++bits;
  // scratch: four int32 arrays of nnz entries + the sort's temporary storage, carved out of the
  // handle's grow-only buffer (no allocation on repeated calls, nothing handed back to an allocator
  // while kernels may still be using it)
  size_t tmp_bytes = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, colind, (int32_t*) nullptr, (int32_t*) nullptr,
                                           (int32_t*) nullptr, (size_t) nnz, 0, bits, s);
  if (e != hipSuccess)
    return hip_fail(e);
  const size_t arr = (((size_t) nnz * 4) + 255) & ~(size_t) 255;
  void* base = nullptr;
  int rc = handle_scratch(handle, 4 * arr + tmp_bytes + 256, &base);
  if (rc)
    return rc;
  char* bp = static_cast<char*>(base);
  int32_t* rowid = reinterpret_cast<int32_t*>(bp);
  int32_t* pos = reinterpret_cast<int32_t*>(bp + arr);
  int32_t* sorted_cols = reinterpret_cast<int32_t*>(bp + 2 * arr);
  int32_t* perm = reinterpret_cast<int32_t*>(bp + 3 * arr);
  void* tmp = bp + 4 * arr;
  hipLaunchKernelGGL(spt_rowid_kernel, dim3((unsigned) cdiv(m, 32)), dim3(256), 0, s, m, rowptr, rowid, pos);
  e = rocprim::radix_sort_pairs(tmp, tmp_bytes, colind, sorted_cols, pos, perm, (size_t) nnz, 0, bits, s);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(spt_rowptr_kernel, dim3((unsigned) cdiv(n + 1, 256)), dim3(256), 0, s, n, nnz, sorted_cols,
                       t_rowptr);
    if (value_type == SPBLAS_GFX950_F32)
      hipLaunchKernelGGL((spt_gather_kernel<float>), dim3((unsigned) cdiv(nnz, 256)), dim3(256), 0, s, nnz, perm, rowid,
                         static_cast<const float*>(values), t_colind, static_cast<float*>(t_values));
    else
      hipLaunchKernelGGL((spt_gather_kernel<double>), dim3((unsigned) cdiv(nnz, 256)), dim3(256), 0, s, nnz, perm, rowid,
                         static_cast<const double*>(values), t_colind, static_cast<double*>(t_values));
    e = hipGetLastError();
  }
  if (e != hipSuccess)
    return hip_fail(e);
  return SPBLAS_GFX950_STATUS_SUCCESS;
}
