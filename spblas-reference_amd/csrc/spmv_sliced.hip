// Column-sliced ("propagation blocking") SpMV for matrices whose x gathers miss every
// cache: SPBLAS_GFX950_SPMV_SLICED.
//
// Why: on BASELINE cfg2 (10M x 10M, uniform random columns) the CSR kernels gather each
// x[col] as a separate 128-byte L2 miss (rocprofv3: TCC_EA0_RDREQ_128B = 0.97 per nonzero,
// 12.4 GB of fabric reads for 0.92 GB of algorithmic bytes -- profiles/r01a_summary.md).
// The only memory on the CU that sustains random 4-byte accesses at the needed rate is LDS,
// so multiply_inspect re-tiles A once on the device:
//
//   A' order   entries grouped by (column slice s, row bin b); slice = W consecutive columns
//              (W*sizeof(T) <= 80 KiB of LDS), bin = H consecutive rows (H*sizeof(T) <= 80 KiB)
//   s_val[i]   value,   s_col[i] 16-bit column inside the slice,  s_row[i] 16-bit row inside the bin
//
// and multiply() runs two streaming kernels (no global gathers, no global atomics):
//   expand  one workgroup per slice: x slice -> LDS, then P[i] = s_val[i] * xs[s_col[i]] over the
//           slice's contiguous range of A' (coalesced 4+2 byte reads, 4 byte writes)
//   reduce  one workgroup per bin: zero H accumulators in LDS, walk the bin's S segments of P
//           (one wavefront per segment), ds_add_f32 into LDS, write y = alpha*acc + beta*y
// HBM traffic per nonzero: 6 B + 4 B (expand) + 6 B (reduce) = 16 B vs 8 B algorithmic, all of it
// coalesced streams.  LDS float atomics make the summation order (not the set of addends)
// vary between runs: results are reproducible to rounding, not bitwise (documented in DESIGN.md).
#include "common.hpp"
#include "plan.hpp"
#include "scan.hpp"

#include <cstdlib>

namespace spb {

static constexpr int PB_THREADS = 1024;
static constexpr int PB_LDS_BYTES = 80 * 1024;  // two workgroups per CU (160 KiB LDS)

// ---- inspect --------------------------------------------------------------------------
// one 8-lane group per CSR row: count entries per (slice, bin)
template <typename O>
__global__ __launch_bounds__(256) void pb_count_kernel(int64_t m, const O* __restrict__ rowptr,
                                                       const int32_t* __restrict__ colind, int W, int H, int NB,
                                                       int32_t* __restrict__ cnt) {
  const int64_t row = (int64_t) blockIdx.x * 32 + threadIdx.x / 8;
  const int lane = threadIdx.x % 8;
  if (row >= m)
    return;
  const int b = (int) (row / H);
  for (O p = rowptr[row] + lane; p < rowptr[row + 1]; p += 8) {
    const int s = colind[p] / W;
    atomicAdd(&cnt[(int64_t) s * NB + b], 1);
  }
}

template <typename T, typename O>
__global__ __launch_bounds__(256) void pb_scatter_kernel(int64_t m, const O* __restrict__ rowptr,
                                                         const int32_t* __restrict__ colind,
                                                         const T* __restrict__ values, int W, int H, int NB,
                                                         const int32_t* __restrict__ seg, int32_t* __restrict__ cursor,
                                                         T* __restrict__ s_val, uint16_t* __restrict__ s_col,
                                                         uint16_t* __restrict__ s_row, int32_t* __restrict__ perm) {
  const int64_t row = (int64_t) blockIdx.x * 32 + threadIdx.x / 8;
  const int lane = threadIdx.x % 8;
  if (row >= m)
    return;
  const int b = (int) (row / H);
  const uint16_t lr = (uint16_t) (row - (int64_t) b * H);
  for (O p = rowptr[row] + lane; p < rowptr[row + 1]; p += 8) {
    const int c = colind[p];
    const int s = c / W;
    const int64_t key = (int64_t) s * NB + b;
    const int i = seg[key] + atomicAdd(&cursor[key], 1);
    s_val[i] = values[p];
    s_col[i] = (uint16_t) (c - s * W);
    s_row[i] = lr;
    perm[i] = (int32_t) p;
  }
}

// locality probe: number of non-empty (slice, bin) segments
__global__ __launch_bounds__(256) void pb_nonempty_kernel(int64_t nseg, const int32_t* __restrict__ cnt,
                                                          unsigned long long* __restrict__ out) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  const bool ne = i < nseg && cnt[i] > 0;
  const unsigned long long c = __popcll(__ballot(ne));
  if ((threadIdx.x & 63) == 0 && c)
    atomicAdd(out, c);
}

// segT[b*S + s] = (start, length) of segment (s, b) in A' order
__global__ __launch_bounds__(256) void pb_transpose_seg_kernel(int S, int NB, const int32_t* __restrict__ seg,
                                                               int2* __restrict__ segT) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t) S * NB)
    return;
  const int b = (int) (i / S), s = (int) (i % S);
  const int64_t key = (int64_t) s * NB + b;
  segT[i] = make_int2(seg[key], seg[key + 1] - seg[key]);
}

template <typename T>
__global__ __launch_bounds__(256) void pb_update_values_kernel(int64_t nnz, const int32_t* __restrict__ perm,
                                                               const T* __restrict__ values, T* __restrict__ s_val) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i < nnz)
    s_val[i] = values[perm[i]];
}

// ---- execute ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(PB_THREADS) void pb_expand_kernel(int64_t n, int W, int NB, const int32_t* __restrict__ seg,
                                                               const T* __restrict__ s_val,
                                                               const uint16_t* __restrict__ s_col,
                                                               const T* __restrict__ x, T* __restrict__ P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* xs = reinterpret_cast<T*>(smem);
  const int s = blockIdx.x;
  const int64_t c0 = (int64_t) s * W;
  const int cw = (int) ((n - c0) < W ? (n - c0) : W);
  for (int i = threadIdx.x; i < cw; i += PB_THREADS)
    xs[i] = x[c0 + i];
  __syncthreads();
  const int a0 = seg[(int64_t) s * NB], a1 = seg[(int64_t) (s + 1) * NB];
  // flat streaming pass over the slice's contiguous range of A'
  int i = a0 + threadIdx.x;
  for (; i + 3 * PB_THREADS < a1; i += 4 * PB_THREADS) {
    const T v0 = stream_load(s_val + i), v1 = stream_load(s_val + i + PB_THREADS),
            v2 = stream_load(s_val + i + 2 * PB_THREADS), v3 = stream_load(s_val + i + 3 * PB_THREADS);
    const int k0 = stream_load(s_col + i), k1 = stream_load(s_col + i + PB_THREADS),
              k2 = stream_load(s_col + i + 2 * PB_THREADS), k3 = stream_load(s_col + i + 3 * PB_THREADS);
    __builtin_nontemporal_store(v0 * xs[k0], P + i);
    __builtin_nontemporal_store(v1 * xs[k1], P + i + PB_THREADS);
    __builtin_nontemporal_store(v2 * xs[k2], P + i + 2 * PB_THREADS);
    __builtin_nontemporal_store(v3 * xs[k3], P + i + 3 * PB_THREADS);
  }
  for (; i < a1; i += PB_THREADS)
    __builtin_nontemporal_store(stream_load(s_val + i) * xs[stream_load(s_col + i)], P + i);
}

template <typename T>
__global__ __launch_bounds__(PB_THREADS) void pb_reduce_kernel(int64_t m, int H, int S, const int2* __restrict__ segT,
                                                               const T* __restrict__ P,
                                                               const uint16_t* __restrict__ s_row,
                                                               T* __restrict__ y, T alpha, T beta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* acc = reinterpret_cast<T*>(smem);
  const int b = blockIdx.x;
  const int64_t r0 = (int64_t) b * H;
  const int rh = (int) ((m - r0) < H ? (m - r0) : H);
  for (int i = threadIdx.x; i < rh; i += PB_THREADS)
    acc[i] = T(0);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int2* mine = segT + (int64_t) b * S;
  for (int s = wave; s < S; s += PB_THREADS / 64) {
    const int2 sg = mine[s];
    const int a1 = sg.x + sg.y;
    int i = sg.x + lane;
    for (; i + 192 < a1; i += 256) {
      const T p0 = stream_load(P + i), p1 = stream_load(P + i + 64), p2 = stream_load(P + i + 128),
              p3 = stream_load(P + i + 192);
      const int q0 = stream_load(s_row + i), q1 = stream_load(s_row + i + 64), q2 = stream_load(s_row + i + 128),
                q3 = stream_load(s_row + i + 192);
      unsafeAtomicAdd(acc + q0, p0);
      unsafeAtomicAdd(acc + q1, p1);
      unsafeAtomicAdd(acc + q2, p2);
      unsafeAtomicAdd(acc + q3, p3);
    }
    for (; i < a1; i += 64)
      unsafeAtomicAdd(acc + stream_load(s_row + i), stream_load(P + i));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < rh; i += PB_THREADS) {
    const T v = alpha * acc[i];
    y[r0 + i] = beta == T(0) ? v : v + beta * y[r0 + i];
  }
}

// ---- host -------------------------------------------------------------------------------
static int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return v && *v ? std::atoi(v) : dflt;
}

// number of pieces: enough that one piece fits the LDS budget; for big problems a multiple
// of 512 (2 workgroups x 256 CUs) so the single wave of workgroups fills the chip evenly.
static void pick_tiling(int64_t extent, int max_elems, int* pieces, int* width) {
  int64_t p = cdiv(extent, max_elems);
  if (p < 1)
    p = 1;
  if (p > 256)
    p = cdiv(p, 512) * 512;
  int64_t w = cdiv(extent, p);
  if (w < 1)
    w = 1;
  p = cdiv(extent, w);
  if (p < 1)
    p = 1;
  *pieces = (int) p;
  *width = (int) w;
}

template <typename T, typename O>
static int sliced_build_typed(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values_p,
                              bool auto_mode) {
  hipStream_t s = h->stream;
  const int64_t m = pl->m, n = pl->n, nnz = pl->nnz;
  if (nnz > INT32_MAX - 8)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  int max_elems = PB_LDS_BYTES / (int) sizeof(T);
  if (max_elems > 65536)
    max_elems = 65536;  // 16-bit local indices
  int S, W, NB, H;
  const int w_env = env_int("SPBLAS_GFX950_SLICE_COLS", 0);  // test hooks: force small tiles
  const int h_env = env_int("SPBLAS_GFX950_SLICE_ROWS", 0);
  pick_tiling(n, w_env > 0 && w_env < max_elems ? w_env : max_elems, &S, &W);
  pick_tiling(m, h_env > 0 && h_env < max_elems ? h_env : max_elems, &NB, &H);
  const int64_t nseg = (int64_t) S * NB;
  if (nseg > (int64_t) 64 << 20)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  pl->n_slices = S;
  pl->slice_cols = W;
  pl->n_rblk = NB;
  pl->rows_per_blk = H;

  int rc;
  int32_t* seg = nullptr;
  int32_t* cursor = nullptr;
  long long* partials = nullptr;
  if ((rc = dev_alloc((void**) &seg, (size_t) (nseg + 1) * 4, s)))
    return rc;
  pl->seg_ptr = seg;
  SPB_HIP(hipMemsetAsync(seg, 0, (size_t) (nseg + 1) * 4, s));
  const O* rowptr = static_cast<const O*>(pl->rowptr);
  const unsigned grid = (unsigned) cdiv(m, 32);
  hipLaunchKernelGGL((pb_count_kernel<O>), dim3(grid), dim3(256), 0, s, m, rowptr, pl->colind, W, H, NB, seg);
  if (auto_mode) {
    // AUTO only: a matrix whose entries cluster in few (slice, bin) tiles (banded, block
    // structured) already gets its x reuse from L2 with the CSR kernels -- decline.
    unsigned long long* d_ne = nullptr;
    unsigned long long ne = 0;
    if ((rc = dev_alloc((void**) &d_ne, sizeof(unsigned long long), s)))
      return rc;
    SPB_HIP(hipMemsetAsync(d_ne, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(pb_nonempty_kernel, dim3((unsigned) cdiv(nseg, 256)), dim3(256), 0, s, nseg, seg, d_ne);
    SPB_HIP(hipMemcpyAsync(&ne, d_ne, sizeof(ne), hipMemcpyDeviceToHost, s));
    SPB_HIP(hipStreamSynchronize(s));
    dev_free(d_ne, s);
    if ((double) ne < 0.25 * (double) nseg)
      return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  }
  if ((rc = dev_alloc((void**) &cursor, (size_t) nseg * 4, s)))
    return rc;
  if ((rc = dev_alloc((void**) &partials, (size_t) (cdiv(nseg, 2048) + 2) * sizeof(long long), s)))
    return rc;
  if ((rc = dev_alloc((void**) &pl->s_values, (size_t) nnz * sizeof(T), s)))
    return rc;
  if ((rc = dev_alloc((void**) &pl->s_colind, (size_t) nnz * 2, s)))
    return rc;
  if ((rc = dev_alloc((void**) &pl->s_lrow, (size_t) nnz * 2, s)))
    return rc;
  if ((rc = dev_alloc((void**) &pl->s_perm, (size_t) nnz * 4, s)))
    return rc;
  if ((rc = dev_alloc((void**) &pl->s_products, (size_t) nnz * sizeof(T), s)))
    return rc;
  if ((rc = dev_alloc((void**) &pl->s_segT, (size_t) nseg * sizeof(int2), s)))
    return rc;
  pl->device_bytes += (size_t) nnz * (2 * sizeof(T) + 8) + (size_t) nseg * 16;
  SPB_HIP(hipMemsetAsync(cursor, 0, (size_t) nseg * 4, s));
  scan_counts_i32(s, nseg, seg, partials);
  hipLaunchKernelGGL((pb_scatter_kernel<T, O>), dim3(grid), dim3(256), 0, s, m, rowptr, pl->colind,
                     static_cast<const T*>(values_p), W, H, NB, seg, cursor, static_cast<T*>(pl->s_values),
                     reinterpret_cast<uint16_t*>(pl->s_colind), pl->s_lrow, reinterpret_cast<int32_t*>(pl->s_perm));
  hipLaunchKernelGGL(pb_transpose_seg_kernel, dim3((unsigned) cdiv(nseg, 256)), dim3(256), 0, s, S, NB, seg,
                     static_cast<int2*>(pl->s_segT));
  SPB_HIP(hipGetLastError());
  SPB_HIP(hipStreamSynchronize(s));
  dev_free(cursor, s);
  dev_free(partials, s);
  // both kernels may use up to 80 KiB of dynamic LDS
  SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pb_expand_kernel<T>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_BYTES));
  SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pb_reduce_kernel<T>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_BYTES));
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spmv_sliced_build(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values, bool auto_mode) {
  const bool f32 = pl->value_type == SPBLAS_GFX950_F32, o32 = pl->offset_type == SPBLAS_GFX950_I32;
  if (f32)
    return o32 ? sliced_build_typed<float, int32_t>(h, pl, values, auto_mode)
               : sliced_build_typed<float, int64_t>(h, pl, values, auto_mode);
  return o32 ? sliced_build_typed<double, int32_t>(h, pl, values, auto_mode)
             : sliced_build_typed<double, int64_t>(h, pl, values, auto_mode);
}

template <typename T>
static int sliced_update_typed(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values) {
  if (pl->nnz == 0)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  hipLaunchKernelGGL((pb_update_values_kernel<T>), dim3((unsigned) cdiv(pl->nnz, 256)), dim3(256), 0, h->stream,
                     pl->nnz, reinterpret_cast<const int32_t*>(pl->s_perm), static_cast<const T*>(values),
                     static_cast<T*>(pl->s_values));
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spmv_sliced_update(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values) {
  return pl->value_type == SPBLAS_GFX950_F32 ? sliced_update_typed<float>(h, pl, values)
                                             : sliced_update_typed<double>(h, pl, values);
}

template <typename T>
static int sliced_exec_typed(spblas_gfx950_handle_t h, const spblas_gfx950_plan_s* pl, const void* alpha_p,
                             const void* x, const void* beta_p, void* y) {
  hipStream_t s = h->stream;
  const T alpha = *static_cast<const T*>(alpha_p), beta = *static_cast<const T*>(beta_p);
  const int32_t* seg = reinterpret_cast<const int32_t*>(pl->seg_ptr);
  hipLaunchKernelGGL((pb_expand_kernel<T>), dim3((unsigned) pl->n_slices), dim3(PB_THREADS),
                     (size_t) pl->slice_cols * sizeof(T), s, pl->n, pl->slice_cols, (int) pl->n_rblk, seg,
                     static_cast<const T*>(pl->s_values), reinterpret_cast<const uint16_t*>(pl->s_colind),
                     static_cast<const T*>(x), static_cast<T*>(pl->s_products));
  hipLaunchKernelGGL((pb_reduce_kernel<T>), dim3((unsigned) pl->n_rblk), dim3(PB_THREADS),
                     (size_t) pl->rows_per_blk * sizeof(T), s, pl->m, pl->rows_per_blk, pl->n_slices,
                     static_cast<const int2*>(pl->s_segT), static_cast<const T*>(pl->s_products), pl->s_lrow,
                     static_cast<T*>(y), alpha, beta);
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spmv_sliced_exec(spblas_gfx950_handle_t h, const spblas_gfx950_plan_s* pl, const void* alpha, const void* x,
                     const void* beta, void* y) {
  return pl->value_type == SPBLAS_GFX950_F32 ? sliced_exec_typed<float>(h, pl, alpha, x, beta, y)
                                             : sliced_exec_typed<double>(h, pl, alpha, x, beta, y);
}

void spmv_sliced_free(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl) {
  hipStream_t s = h->stream;
  dev_free(pl->seg_ptr, s);
  dev_free(pl->s_colind, s);
  dev_free(pl->s_values, s);
  dev_free(pl->s_lrow, s);
  dev_free(pl->s_perm, s);
  dev_free(pl->s_products, s);
  dev_free(pl->s_segT, s);
  pl->seg_ptr = pl->s_colind = pl->s_values = pl->s_products = pl->s_segT = pl->s_perm = nullptr;
  pl->s_lrow = nullptr;
  pl->n_slices = 0;
}

} // namespace spb
