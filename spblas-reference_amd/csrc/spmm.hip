// CSR x dense SpMM for gfx950:  C = alpha * A * B + beta * C, B (k x n) and C (m x n)
// row-major.
//
// The rocSPARSE slot of the reference has no SpMM; the device call site this
// replaces is oneapi::mkl::sparse::gemm at
// /root/reference/include/spblas/vendor/onemkl_sycl/spmm_impl.hpp:116-120 and the
// maths is the CPU path include/spblas/algorithms/multiply_impl.hpp:66-92.
//
// Three kernels; which rows go where is decided by multiply_inspect (spblas_gfx950_spmm_inspect, the counterpart of
// oneMKL's optimize_gemm at vendor/onemkl_sycl/spmm_impl.hpp:40-67) -- without a plan only the first one runs:
//   spmm_rowgroup_kernel   a group of G lanes (power of two) owns one row of A and a panel of G*V columns of
//                          B/C, V = elements per 16-byte (or narrower) lane access.  The group loads G
//                          (colind, value) pairs with one coalesced streaming access, broadcasts them lane by
//                          lane and gathers whole B rows: every gather is one contiguous G*V*sizeof(T)-byte
//                          segment (512 B for n = 128, fp32, V = 4, G = 32).  Bound by the B-row gathers.
//   spmm_long_rows_kernel  rows longer than the plan's nnz window (hub rows of a power-law matrix: one lane group
//                          would walk 1e5 entries) are cut into parts of ~4 K entries, one workgroup each; a
//                          finish kernel adds the parts in order.
//   spmm_panel_kernel      fp32: row blocks (32 rows) whose entries fall into <= 16 aligned tiles of 64 columns,
//                          at >= 1/5 density: the B tile (128 x n) is staged in LDS ONCE for the block, the block
//                          of A is scattered into a dense 32 x 128 LDS tile, and the contraction runs on the matrix
//                          cores (v_mfma_f32_32x32x2_f32: exact f32, a k-ordered fma chain).  Banded / block
//                          structured matrices; for uniform random columns no block qualifies and MFMA use is 0
//                          by construction.
// Algorithmic bytes = nnz*(sizeof(T)+4) + (m+1)*sizeof(O) + (k*n + m*n)*sizeof(T).
#include "common.hpp"
#include "plan.hpp"
#include "scan.hpp"

#include <cstdlib>

namespace spb {

static bool env_flag(const char* name) {
  const char* v = std::getenv(name);
  return v && *v && *v != '0';
}

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
                                                            int G, const unsigned char* __restrict__ is_panel,
                                                            int long_len) {
  // is_panel[row / 32] != 0: the panel kernel owns the row block; long_len > 0: rows longer than that belong to
  // the long-row kernel
  typedef typename vec_of<T, V>::type vec_t;
  const int rows_per_block = 256 / G;
  const int64_t row = (int64_t) blockIdx.x * rows_per_block + threadIdx.x / G;
  const int lig = threadIdx.x % G;
  const int64_t panel_cols = (int64_t) G * V;
  O p0 = 0, p1 = 0;
  bool mine = row < m;
  if (mine && is_panel && is_panel[row >> 5])
    mine = false;
  if (mine) {
    p0 = rowptr[row];
    p1 = rowptr[row + 1];
    if (long_len > 0 && p1 - p0 > (O) long_len)
      mine = false;
  }
  if (!mine)
    p0 = p1 = 0;
  for (int64_t col0 = (int64_t) lig * V; col0 - (int64_t) lig * V < n; col0 += panel_cols) {
    const bool active = mine && col0 < n;
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
                        const T* values, const T* B, int64_t ldb, T* C, int64_t ldc, T alpha, T beta,
                        const unsigned char* is_panel, int long_len) {
  int G = 1;
  while (G < 64 && (int64_t) G * V < n)
    G <<= 1;
  const int rows_per_block = 256 / G;
  hipLaunchKernelGGL((spmm_rowgroup_kernel<T, O, V>), dim3((unsigned) cdiv(m, rows_per_block)), dim3(256), 0, s,
                     m, n, rowptr, colind, values, B, ldb, C, ldc, alpha, beta, G, is_panel, long_len);
}

// ---- long rows ------------------------------------------------------------------------------------------
// Workgroup (i, part) sums entries [lo, hi) of long row i for all n columns into part_buf[(i*parts + part)*n ..]:
// thread t owns column t % cpp of the current pass of cpp = min(n, 256) columns and walks every eg-th entry
// (eg = 256 / cpp entry groups); the groups' sums meet in LDS.
template <typename T, typename O>
__global__ __launch_bounds__(256) void spmm_long_rows_kernel(const int32_t* __restrict__ long_rows, int parts,
                                                             int64_t n, const O* __restrict__ rowptr,
                                                             const int32_t* __restrict__ colind,
                                                             const T* __restrict__ values, const T* __restrict__ B,
                                                             int64_t ldb, T* __restrict__ part_buf) {
  __shared__ T red[256];
  const int64_t i = blockIdx.x;
  const int part = blockIdx.y;
  const int64_t r = long_rows[i];
  const O p0 = rowptr[r], p1 = rowptr[r + 1];
  const O per = ((p1 - p0) + (O) parts - 1) / (O) parts;
  const O lo = p0 + (O) part * per, hi = (lo + per) < p1 ? (lo + per) : p1;
  const int cpp = n < 256 ? (int) n : 256;   // columns per pass
  const int eg = 256 / cpp;                   // entry groups
  const int j = threadIdx.x % cpp, e = threadIdx.x / cpp;
  T* out = part_buf + ((int64_t) i * parts + part) * n;
  for (int64_t c0 = 0; c0 < n; c0 += cpp) {
    const bool col_ok = e < eg && c0 + j < n;
    T acc = T(0);
    if (col_ok) {
      const T* Bc = B + c0 + j;
      O p = lo + (O) e;
      for (; p + (O) (3 * eg) < hi; p += (O) (4 * eg)) {  // four gathers in flight
        const int64_t k0 = colind[p], k1 = colind[p + eg], k2 = colind[p + 2 * eg], k3 = colind[p + 3 * eg];
        const T b0 = Bc[k0 * ldb], b1 = Bc[k1 * ldb], b2 = Bc[k2 * ldb], b3 = Bc[k3 * ldb];
        acc += values[p] * b0;
        acc += values[p + eg] * b1;
        acc += values[p + 2 * eg] * b2;
        acc += values[p + 3 * eg] * b3;
      }
      for (; p < hi; p += (O) eg)
        acc += values[p] * Bc[(int64_t) colind[p] * ldb];
    }
    __syncthreads();
    red[threadIdx.x] = acc;
    __syncthreads();
    if (e == 0 && c0 + j < n) {
      T sum = red[j];
      for (int g = 1; g < eg; ++g)
        sum += red[g * cpp + j];
      out[c0 + j] = sum;
    }
  }
}

// C[row] = alpha * (parts in order) + beta * C[row] for every long row
template <typename T>
__global__ __launch_bounds__(256) void spmm_long_finish_kernel(const int32_t* __restrict__ long_rows, int parts,
                                                               int64_t n, const T* __restrict__ part_buf,
                                                               T* __restrict__ C, int64_t ldc, T alpha, T beta) {
  const int64_t i = blockIdx.x;
  const int64_t r = long_rows[i];
  for (int64_t j = threadIdx.x; j < n; j += 256) {
    T sum = T(0);
    for (int q = 0; q < parts; ++q)
      sum += part_buf[((int64_t) i * parts + q) * n + j];
    T* cp = C + r * ldc + j;
    *cp = beta == T(0) ? alpha * sum : alpha * sum + beta * *cp;
  }
}

// ---- panel path (fp32, matrix cores) ---------------------------------------------------------------------
static constexpr int MM_RB = 32;      // rows per block (the M of v_mfma_f32_32x32x2_f32)
static constexpr int MM_KT = 64;      // columns of A / rows of B per tile (32 KiB of B + 8 KiB of A in LDS: 4 workgroups per CU)
static constexpr int MM_MAXT = 16;    // tiles per block at most
static constexpr int MM_NP = 128;     // columns of B / C per pass (4 wavefronts x 32)
static constexpr int MM_TS = MM_MAXT + 1;  // ints per block in the tile table: count + ids

// inspect: one wavefront per row block.  tiles[b*9] = number of distinct aligned column tiles of the block (9 = more
// than MM_MAXT), tiles[b*9 + 1 ..] = their ids ascending; is_panel[b] = 1 when the block qualifies: <= MM_MAXT
// tiles, >= min_per_tile entries per tile on average, no row longer than long_len.  flag32[b] repeats is_panel[b]
// as an int for the scan that turns the flags into the ASCENDING list of qualifying blocks (neighbouring list
// entries share B tiles, which the panel kernel's XCD mapping relies on).
template <typename O>
__global__ __launch_bounds__(64) void spmm_panel_probe_kernel(int64_t m, int64_t nblk, const O* __restrict__ rowptr,
                                                              const int32_t* __restrict__ colind, int long_len,
                                                              int min_per_tile, int32_t* __restrict__ tiles,
                                                              unsigned char* __restrict__ is_panel,
                                                              int32_t* __restrict__ flag32,
                                                              unsigned long long* __restrict__ counters,
                                                              int2* __restrict__ win, int32_t* __restrict__ flag_dense,
                                                              int band_ch, int dense_pm) {
  __shared__ int slot[MM_MAXT];
  __shared__ int overflow;
  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x;
  const int64_t r0 = b * MM_RB, r1 = (r0 + MM_RB) < m ? (r0 + MM_RB) : m;
  if (lane < MM_MAXT)
    slot[lane] = -1;
  if (lane == 0)
    overflow = 0;
  __syncthreads();
  const O p0 = rowptr[r0], p1 = rowptr[r1];
  bool has_long = false;
  if (long_len > 0 && r0 + lane < r1 && lane < MM_RB)
    has_long = rowptr[r0 + lane + 1] - rowptr[r0 + lane] > (O) long_len;
  has_long = __any(has_long);
  int cmin = INT32_MAX, cmax = -1;  // the block's column window (exact when the scan below ran to its end)
  // cheap reject before the scan: too few entries for even one tile
  if (!has_long && p1 - p0 >= (O) min_per_tile) {
    for (O p = p0 + lane; p < p1 && !overflow; p += 64) {
      const int cc = colind[p];
      cmin = cc < cmin ? cc : cmin;
      cmax = cc > cmax ? cc : cmax;
      const int t = cc / MM_KT;
      bool placed = false;
      for (int q = 0; q < MM_MAXT && !placed; ++q) {
        const int old = atomicCAS(&slot[q], -1, t);
        placed = old == -1 || old == t;
      }
      if (!placed)
        overflow = 1;
    }
  }
  __syncthreads();
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const int a = __shfl_xor(cmin, o, 64), c = __shfl_xor(cmax, o, 64);
    cmin = a < cmin ? a : cmin;
    cmax = c > cmax ? c : cmax;
  }
  if (lane == 0) {
    win[b] = make_int2(cmin, cmax);
    int ids[MM_MAXT], nt = 0;
    for (int q = 0; q < MM_MAXT; ++q)
      if (slot[q] >= 0)
        ids[nt++] = slot[q];
    for (int a = 1; a < nt; ++a) {  // insertion sort: ascending tiles = ascending k order of the contraction
      const int v = ids[a];
      int c = a - 1;
      while (c >= 0 && ids[c] > v) {
        ids[c + 1] = ids[c];
        --c;
      }
      ids[c + 1] = v;
    }
    const bool ok = !has_long && !overflow && nt > 0 && (int64_t) (p1 - p0) >= (int64_t) min_per_tile * nt;
    tiles[b * MM_TS] = overflow ? MM_MAXT + 1 : nt;
    for (int q = 0; q < nt; ++q)
      tiles[b * MM_TS + 1 + q] = ids[q];
    // 2 = dense in its window: the matrix-core kernel of the band path (spmm_band_mfma_kernel); 1 = the entry-loop kernel
    const int64_t W = (int64_t) cmax - cmin + 1;
    const bool dense = ok && ((W + 3) & ~(int64_t) 3) <= band_ch && (int64_t) (p1 - p0) * 1000 >= (int64_t) dense_pm * MM_RB * W;
    is_panel[b] = ok ? (dense ? 2 : 1) : 0;
    flag32[b] = ok ? 1 : 0;
    flag_dense[b] = dense ? 1 : 0;
    if (ok)
      atomicAdd(&counters[1], (unsigned long long) (p1 - p0));
  }
}

// panel_blocks[offset[b]] = b for every qualifying block (offset = exclusive scan of the flags)
__global__ __launch_bounds__(256) void spmm_panel_list_kernel(int64_t nblk, const unsigned char* __restrict__ is_panel,
                                                              const int32_t* __restrict__ offset,
                                                              int32_t* __restrict__ panel_blocks) {
  const int64_t b = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (b < nblk && is_panel[b])
    panel_blocks[offset[b]] = (int32_t) b;
}

// the band path's two lists: dense blocks (is_panel == 2) and the others (1), each ascending
__global__ __launch_bounds__(256) void spmm_band_lists_kernel(int64_t nblk, const unsigned char* __restrict__ is_panel,
                                                              const int32_t* __restrict__ off_all,
                                                              const int32_t* __restrict__ off_dense,
                                                              int32_t* __restrict__ vec_blocks,
                                                              int32_t* __restrict__ dense_blocks) {
  const int64_t b = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (b >= nblk)
    return;
  if (is_panel[b] == 2)
    dense_blocks[off_dense[b]] = (int32_t) b;
  else if (is_panel[b] == 1)
    vec_blocks[off_all[b] - off_dense[b]] = (int32_t) b;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// One workgroup (4 wavefronts) per qualifying row block.  Per tile: A entries -> dense At[k][row] (k-major: the
// MFMA A operand of lane l is At[2s + (l >> 5)][l & 31], consecutive lanes consecutive addresses), the B tile ->
// Bt[k][col]; 64 MFMA steps of k = 2 contract it; wavefront w owns output columns [32w, 32w + 32) of the pass.
// A B tile holding a non-finite value is contracted entry by entry instead (0 * inf would poison rows that do not
// reference it; the reference only multiplies stored entries, multiply_impl.hpp:85-91).
template <typename O>
__global__ __launch_bounds__(256) void spmm_panel_kernel(int64_t m, int64_t k, int64_t n,
                                                         const O* __restrict__ rowptr,
                                                         const int32_t* __restrict__ colind,
                                                         const float* __restrict__ values,
                                                         const float* __restrict__ B, int64_t ldb,
                                                         float* __restrict__ C, int64_t ldc, float alpha, float beta,
                                                         const int32_t* __restrict__ panel_blocks,
                                                         const int32_t* __restrict__ tiles, int64_t nlist_arg, int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Bt = reinterpret_cast<float*>(smem);       // [MM_KT][MM_NP]
  float* At = Bt + MM_KT * MM_NP;                   // [MM_KT][MM_RB]
  int& nonfinite = *reinterpret_cast<int*>(At + MM_KT * MM_RB);  // (kept in the dynamic region: 16-byte aligned base)
  int* rp_s = reinterpret_cast<int*>(At + MM_KT * MM_RB) + 4;      // [MM_RB + 1] row offsets relative to the block
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // consecutive list entries share B tiles: keep them on ONE XCD (workgroup i runs on XCD i % 8; each XCD has its own
  // L2) by giving XCD x the x-th contiguous eighth of the list -- 8.5 GB of fabric reads for 1 GB of B otherwise
  const int64_t nlist = nlist_arg, per_xcd = (nlist + 7) / 8;
  const int64_t li = (int64_t) (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (li >= nlist)
    return;
  const int64_t b = panel_blocks[li];
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(B) | (uintptr_t) (ldb * 4)) & 15) == 0;
  const int64_t r0 = b * MM_RB, r1 = (r0 + MM_RB) < m ? (r0 + MM_RB) : m;
  const O p0 = rowptr[r0], p1 = rowptr[r1];
  const int nt = tiles[b * MM_TS];
  if (tid <= (int) (r1 - r0))
    rp_s[tid] = (int) (rowptr[r0 + tid] - p0);
  __syncthreads();
  auto row_of = [&](int q) {  // last row of the block starting at or before entry q (offsets relative to the block)
    int lo = 0, hi = (int) (r1 - r0);
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (rp_s[mid] <= q)
        lo = mid;
      else
        hi = mid;
    }
    return lo;
  };
  // this thread's four of the block's first 1 024 entries: column (-1 = none), value, row inside the block
  int64_t ec[4];
  float ev[4];
  int er[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const O p = p0 + (O) (u * 256 + tid);
    const bool ok = p < p1;
    const O pc = ok ? p : (p1 > p0 ? p1 - 1 : p0);
    ec[u] = ok ? (int64_t) colind[pc] : -1;
    ev[u] = ok ? values[pc] : 0.f;
    er[u] = ok ? row_of((int) (p - p0)) : 0;
  }
  for (int64_t c0 = 0; c0 < n; c0 += MM_NP) {
    const int ncol = (int) ((n - c0) < MM_NP ? (n - c0) : MM_NP);
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q)
      acc[q] = 0.f;
    for (int ti = 0; ti < nt; ++ti) {
      const int64_t kbase = (int64_t) tiles[b * MM_TS + 1 + ti] * MM_KT;
      __syncthreads();  // the previous tile's operands are no longer read
      for (int q = tid; q < MM_KT * MM_RB; q += 256)
        At[q] = 0.f;
      if (tid == 0)
        nonfinite = 0;
      // B tile: rows [kbase, kbase + 128) x columns [c0, c0 + ncol), zero elsewhere.  16 x 16-byte loads per thread,
      // all issued before the first LDS write (one load at a time left the workgroup waiting 64 memory round trips)
      bool bad = false;
      if (dbg & 4) {
      } else if (vec_ok && ncol == MM_NP) {
        constexpr int PER = MM_KT * MM_NP / 4 / 256;  // 8 float4 per thread
        f32x4 buf[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
          const int q4 = tid + u * 256, kk = q4 / (MM_NP / 4), j4 = q4 % (MM_NP / 4);
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
          buf[u] = kbase + kk < k ? *reinterpret_cast<const f32x4*>(B + (kbase + kk) * ldb + c0 + 4 * j4) : zero;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
          const int q4 = tid + u * 256;
          bad |= !(__builtin_fabsf(buf[u].x) <= 3.402823466e38f) | !(__builtin_fabsf(buf[u].y) <= 3.402823466e38f) |
                 !(__builtin_fabsf(buf[u].z) <= 3.402823466e38f) | !(__builtin_fabsf(buf[u].w) <= 3.402823466e38f);
          reinterpret_cast<f32x4*>(Bt)[q4] = buf[u];
        }
      } else {
        for (int q = tid; q < MM_KT * MM_NP; q += 256) {
          const int kk = q / MM_NP, jj = q % MM_NP;
          float v = 0.f;
          if (kbase + kk < k && jj < ncol)
            v = B[(kbase + kk) * ldb + c0 + jj];
          bad |= !(__builtin_fabsf(v) <= 3.402823466e38f);  // NaN or infinity
          Bt[q] = v;
        }
      }
      __syncthreads();
      if (bad)
        nonfinite = 1;
      // A block restricted to this tile (duplicates of one (row, column) accumulate: LDS float atomics, few entries).
      // The first 1 024 entries of the block sit in registers (column, value, row) since before the tile loop; only
      // larger blocks go back to memory for the rest.
      if (!(dbg & 2)) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (ec[u] >= kbase && ec[u] < kbase + MM_KT)
            unsafeAtomicAdd(&At[(int) (ec[u] - kbase) * MM_RB + er[u]], ev[u]);
        for (O base = p0 + (O) 1024; base < p1; base += (O) (4 * 256)) {
          int32_t cc[4];
          float vv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const O p = base + (O) (u * 256 + tid);
            const O pc = p < p1 ? p : p1 - 1;
            cc[u] = colind[pc];
            vv[u] = values[pc];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const O p = base + (O) (u * 256 + tid);
            const int64_t c = cc[u];
            if (p < p1 && c >= kbase && c < kbase + MM_KT)
              unsafeAtomicAdd(&At[(int) (c - kbase) * MM_RB + row_of((int) (p - p0))], vv[u]);
          }
        }
      }
      __syncthreads();
      if (!nonfinite) {
        if (wave * 32 < ncol && !(dbg & 1)) {
          const float* ap = At + (lane >> 5) * MM_RB + (lane & 31);
          const float* bp = Bt + (lane >> 5) * MM_NP + wave * 32 + (lane & 31);
#pragma unroll 8
          for (int sidx = 0; sidx < MM_KT / 2; ++sidx)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[sidx * 2 * MM_RB], bp[sidx * 2 * MM_NP], acc, 0, 0, 0);
        }
      } else if (wave * 32 < ncol) {
        // entry-by-entry contraction in the accumulator layout: D row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
        for (int kk = 0; kk < MM_KT; ++kk)
          for (int rr = 0; rr < MM_RB; ++rr) {
            const float a = At[kk * MM_RB + rr];
            if (a != 0.f) {  // only stored entries contribute (an explicit zero entry times inf is lost here: see DESIGN)
              const float prod = a * Bt[kk * MM_NP + wave * 32 + (lane & 31)];
              const bool owner = ((rr >> 2) & 1) == (lane >> 5);
              const int reg = (rr & 3) + 4 * (rr >> 3);
#pragma unroll
              for (int q = 0; q < 16; ++q)
                if (owner && q == reg)
                  acc[q] += prod;
            }
          }
      }
    }
    // C block: lane holds column c0 + 32*wave + (lane & 31), rows (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int jj = wave * 32 + (lane & 31);
    if (jj < ncol) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t row = r0 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (row < r1) {
          float* cp = C + row * ldc + c0 + jj;
          *cp = beta == 0.f ? alpha * acc[q] : alpha * acc[q] + beta * *cp;
        }
      }
    }
  }
}

// ---- band path (fp32): LDS-staged B window, vector FMAs over the STORED entries only ----------------------------------
// Round 6.  The matrix-core kernel above contracts DENSE 32 x 64 tiles: at the 1/5 density that admits a block it issues
// 3 - 5x the useful flops, fp32 MFMA runs at the fp32 vector rate on gfx950, and building the dense A tile costs more than
// the contraction (measured on the banded bench, 2.36 ms: A tile by LDS float atomics 0.77, skeleton -- five dependent
// global round trips per block at three workgroups per CU -- 0.72, MFMA 0.57, B staging 0.30; the phases add up, nothing
// overlaps).  This kernel keeps what pays -- B rows staged ONCE per block in LDS, shared by the 32 rows -- and drops the
// dense A tile: a wavefront owns 32 / NW rows, its lanes two output columns each; an entry (k, a) of a row is
// wave-uniform (v_readlane from the lanes that loaded the row's entries with one coalesced load), one ds_read_b64 fetches the
// lane's two elements of B[k] from the window and two FMAs add them -- 512 B of LDS traffic per entry, the bound (a CU's LDS
// delivers 128 B per clock: 4 clocks per entry).  The window is the block's exact column range [kmin, kmax] (inspect), staged
// CH rows at a time by LDS-DMA (global_load_lds_dwordx4: no registers, one wave-instruction per two B rows); a band of
// +-48 columns is ONE chunk.  Only stored entries are multiplied, so a non-finite element of B reaches exactly the rows
// the reference's loop gives it to (multiply_impl.hpp:85-91) and no dense fallback is needed.  Per row the entries are
// added in storage order within a chunk, chunks in ascending column order.
typedef float mb_f2 __attribute__((ext_vector_type(2)));
template <typename O, int NW>
__global__ __launch_bounds__(NW * 64) void spmm_band_kernel(int64_t m, int64_t n, int64_t k, const O* __restrict__ rowptr,
                                                            const int32_t* __restrict__ colind,
                                                            const float* __restrict__ values,
                                                            const float* __restrict__ B, int64_t ldb,
                                                            float* __restrict__ C, int64_t ldc, float alpha, float beta,
                                                            const int32_t* __restrict__ panel_blocks,
                                                            const int2* __restrict__ win,
                                                            const int32_t* __restrict__ tiles, int64_t nlist_arg, int CH,
                                                            int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Bt = reinterpret_cast<float*>(smem);  // [CH][MM_NP]
  constexpr int RPW = MM_RB / NW;              // rows per wavefront
  constexpr int EB = 8;                        // entries per batch of the single-chunk path
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // neighbouring list entries share B rows: XCD x takes the x-th contiguous eighth of the list (as the panel kernel)
  const int64_t nlist = nlist_arg, per_xcd = (nlist + 7) / 8;
  const int64_t li = (int64_t) (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (li >= nlist)
    return;
  const int64_t b = panel_blocks[li];
  const int64_t r0 = b * MM_RB, r1 = (r0 + MM_RB) < m ? (r0 + MM_RB) : m;
  const int2 w = win[b];
  const bool single = w.y - w.x < CH;  // the whole window in one chunk: every entry of every row lies inside it
  const int64_t rw0 = r0 + (int64_t) wave * RPW;
  // row offsets of the wave's rows (rows past the end of the matrix are empty): wave-uniform, scalar loads
  O p[RPW + 1];
#pragma unroll
  for (int i = 0; i <= RPW; ++i)
    p[i] = rowptr[(rw0 + i) < r1 ? (rw0 + i) : r1];
  // ... and the first 64 entries of every row: loaded once per block (in flight while the window is staged), kept in
  // registers across column passes
  int ecol[RPW], eval_[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const O q = p[i] + (O) lane;
    const bool ok = q < p[i + 1];
    ecol[i] = ok ? colind[q] : w.x;  // (lanes without an entry: a valid window row, never read back)
    eval_[i] = ok ? __float_as_int(values[q]) : 0;
  }
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(B) | (uintptr_t) (ldb * 4)) & 15) == 0;
  const bool c2_ok = ((reinterpret_cast<uintptr_t>(C) | (uintptr_t) (ldc * 4)) & 7) == 0;
  for (int64_t c0 = 0; c0 < n; c0 += MM_NP) {
    const int ncol = (int) ((n - c0) < MM_NP ? (n - c0) : MM_NP);
    mb_f2 acc[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i)
      acc[i] = mb_f2{0.f, 0.f};
    if (single) {
      const int kb = w.x, rows_here = w.y - w.x + 1;
      __syncthreads();  // the previous pass is no longer read
      if (dbg & 4) {  // timing experiment (results wrong): no staging
      } else if (vec_ok && ncol == MM_NP) {
        // two B rows (2 x 512 B) per wave-instruction: lanes 0..31 row 2q, lanes 32..63 row 2q + 1
        const int npair = (rows_here + 1) >> 1;
        for (int q = wave; q < npair; q += NW) {
          const int kk = 2 * q + (lane >> 5);
          if (kk < rows_here)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*) (B + (int64_t) (kb + kk) * ldb + c0 + 4 * (lane & 31)),
                (__attribute__((address_space(3))) void*) (Bt + (size_t) q * 2 * MM_NP), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0);  // (the form the compiler's wait-count pass sees: DESIGN section 5)
      } else {
        for (int q = tid; q < rows_here * MM_NP; q += NW * 64) {
          const int kk = q / MM_NP, jj = q % MM_NP;
          Bt[q] = jj < ncol ? B[(int64_t) (kb + kk) * ldb + c0 + jj] : 0.f;
        }
      }
      __syncthreads();
      // The first 64 entries of each row sit in the wave's lanes (one coalesced load per row, issued before the staging above
      // was waited for).  An entry is handed to all lanes through a 512-byte piece of LDS that belongs to the wave: the lanes
      // store (byte offset of B row k inside the window, value) once per row, every lane then reads entry j back with ONE
      // broadcast ds_read_b64 at an immediate offset -- no vector-ALU work to distribute an entry (two v_readlane per entry
      // were measured at 0.57 of 1.82 ms here: the vector ALU, not the LDS, bounds this loop).  Per entry: 1 address add,
      // 1 packed FMA, 2 LDS reads.  LDS operations of one wave execute in order: the store of the next row cannot pass the
      // reads of this one.
      int2* const ebuf = reinterpret_cast<int2*>(Bt + (size_t) CH * MM_NP) + wave * 64;
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int len = (dbg & 1) ? 0 : (int) (p[i + 1] - p[i]);  // (dbg bit 0: timing experiment, no contraction)
        for (int base = 0; base < len; base += 64) {
          int cv = ecol[i], vi = eval_[i];
          if (base != 0) {  // rows longer than 64 entries: the rest, 64 at a time
            const O q = p[i] + (O) (base + lane);
            const bool ok = q < p[i + 1];
            cv = ok ? colind[q] : kb;
            vi = ok ? __float_as_int(values[q]) : 0;
          }
          ebuf[lane] = make_int2((cv - kb) * (MM_NP * 4), vi);
          const int cnt = (len - base) < 64 ? (len - base) : 64;
          const char* bl8 = reinterpret_cast<const char*>(Bt) + 8 * lane;
          int j = 0;
          for (; j + EB <= cnt; j += EB) {
            int2 ent[EB];
            mb_f2 bv[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u)
              ent[u] = ebuf[j + u];
#pragma unroll
            for (int u = 0; u < EB; ++u)
              bv[u] = *reinterpret_cast<const mb_f2*>(bl8 + ent[u].x);
#pragma unroll
            for (int u = 0; u < EB; ++u) {
              const float av = __int_as_float(ent[u].y);
              acc[i] = __builtin_elementwise_fma(mb_f2{av, av}, bv[u], acc[i]);
            }
          }
          for (; j < cnt; ++j) {
            const int2 ent = ebuf[j];
            const float av = __int_as_float(ent.y);
            const mb_f2 bv = *reinterpret_cast<const mb_f2*>(bl8 + ent.x);
            acc[i] = __builtin_elementwise_fma(mb_f2{av, av}, bv, acc[i]);
          }
        }
      }
    } else {
      // A window wider than the staging area (a block that straddles a wrap-around, scattered dense tiles): the block's
      // aligned 64-column tiles from inspect (ascending, at most MM_MAXT), G = CH / 64 of them staged at a time; the entries
      // of a row that lie in the staged tiles are found by a ballot over 64 entries at a time and handed round by v_readlane
      const int nt = tiles[b * MM_TS];
      const int G = CH / MM_KT < 4 ? CH / MM_KT : 4;
      for (int g0 = 0; g0 < nt; g0 += G) {
        int tid_s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          tid_s[u] = (u < G && g0 + u < nt) ? tiles[b * MM_TS + 1 + g0 + u] : -2;
        __syncthreads();  // the previous group is no longer read
        for (int q = tid; q < G * MM_KT * (MM_NP / 4); q += NW * 64) {
          const int slot = q / (MM_KT * (MM_NP / 4)), rem = q % (MM_KT * (MM_NP / 4));
          const int kk = rem / (MM_NP / 4), j4 = (rem % (MM_NP / 4)) * 4;
          const int t = slot == 0 ? tid_s[0] : slot == 1 ? tid_s[1] : slot == 2 ? tid_s[2] : tid_s[3];
          const int64_t krow = (int64_t) t * MM_KT + kk;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (t >= 0 && krow < k) {
            const float* src = B + krow * ldb + c0 + j4;
            if (vec_ok && ncol == MM_NP) {
              v = *reinterpret_cast<const f32x4*>(src);
            } else {
              v.x = j4 + 0 < ncol ? src[0] : 0.f;
              v.y = j4 + 1 < ncol ? src[1] : 0.f;
              v.z = j4 + 2 < ncol ? src[2] : 0.f;
              v.w = j4 + 3 < ncol ? src[3] : 0.f;
            }
          }
          *reinterpret_cast<f32x4*>(Bt + (size_t) (slot * MM_KT + kk) * MM_NP + j4) = v;
        }
        __syncthreads();
        const float* bl = Bt + 2 * lane;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
          for (O base = p[i]; base < p[i + 1]; base += 64) {
            const O q = base + (O) lane;
            const bool ok = q < p[i + 1];
            const int cv = ok ? colind[q] : -1;
            const int vi = ok ? __float_as_int(values[q]) : 0;
            const int ct = cv >> 6;  // (-1 for lanes without an entry: matches no tile id)
            int slot = -1;
#pragma unroll
            for (int u = 0; u < 4; ++u)
              slot = ct == tid_s[u] ? u : slot;
            const int lrow = slot * MM_KT + (cv & (MM_KT - 1));
            unsigned long long mask = __ballot(slot >= 0);
            while (mask) {
              const int j = __builtin_ctzll(mask);
              mask &= mask - 1;
              const int kk = __builtin_amdgcn_readlane(lrow, j);
              const float av = __int_as_float(__builtin_amdgcn_readlane(vi, j));
              const mb_f2 bv = *reinterpret_cast<const mb_f2*>(bl + kk * MM_NP);
              acc[i] = __builtin_elementwise_fma(mb_f2{av, av}, bv, acc[i]);
            }
          }
        }
      }
    }
    // C: lane holds columns c0 + 2 lane, + 1 of the wave's rows
    const int jj = 2 * lane;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int64_t row = rw0 + i;
      if (row < r1 && jj < ncol) {
        float* cp = C + row * ldc + c0 + jj;
        if (c2_ok && jj + 1 < ncol) {
          mb_f2 o = alpha * acc[i];
          if (beta != 0.f)
            o += beta * *reinterpret_cast<const mb_f2*>(cp);
          *reinterpret_cast<mb_f2*>(cp) = o;
        } else {
          cp[0] = beta == 0.f ? alpha * acc[i].x : alpha * acc[i].x + beta * cp[0];
          if (jj + 1 < ncol)
            cp[1] = beta == 0.f ? alpha * acc[i].y : alpha * acc[i].y + beta * cp[1];
        }
      }
    }
  }
}

// ---- dense blocks of the band path: the contraction on the matrix cores ------------------------------------------------
// spmm_band_kernel's entry loop is bound by the LDS (two reads per entry: 4 clocks of a CU's LDS, 8 192 per block of the
// banded bench).  When a block fills its window densely enough (inspect: >= mm_band_dense_pm / 1000 of the 32 x W slots, W <=
// the staged rows), a DENSE 32 x W tile of A in LDS (At[k][row], 16 KiB beside the 64 KiB B window: two workgroups per CU)
// turns the same contraction into W / 4 steps of v_mfma_f32_16x16x4_f32 per wave and row half (wave w: output columns
// 16 w .. 16 w + 15), whose operands are two 4-byte LDS reads per step.  Building At costs three LDS accesses per entry-lane
// and no atomics: the rows of a wave are its own, every lane stores a TAG at its entry's slot, reads it back, and a slot
// that holds another lane's tag is a duplicate (row, column) pair -- only then the wave falls back to LDS float adds.  A
// zero of At times a non-finite element of B would put a NaN into rows that do not reference that row of B (the reference
// multiplies stored entries only, multiply_impl.hpp:85-91): a block whose result holds a non-finite value is computed
// again entry by entry (plain loop, one row at a time), which has the reference's semantics.
template <typename O>
__global__ __launch_bounds__(512) void spmm_band_mfma_kernel(int64_t m, int64_t n, const O* __restrict__ rowptr,
                                                             const int32_t* __restrict__ colind,
                                                             const float* __restrict__ values,
                                                             const float* __restrict__ B, int64_t ldb,
                                                             float* __restrict__ C, int64_t ldc, float alpha, float beta,
                                                             const int32_t* __restrict__ dense_blocks,
                                                             const int2* __restrict__ win, int64_t nlist_arg, int CH, int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NW = 8, RPW = MM_RB / NW;
  float* Bt = reinterpret_cast<float*>(smem);  // [CH][MM_NP]
  // [MM_RB][CHs], element (row, k) at row * CHs + (k ^ 2 (row & 15)), CHs = CH rounded up to 32: the lanes of a wave store
  // the entries of ONE row (distinct k: distinct banks; [k][row] put all 64 on one bank), and the matrix-core operand read
  // -- lane l: row l & 15, k = 4 s + (l >> 4) -- meets 32 different banks per half wave through the XOR
  float* At = Bt + (size_t) CH * MM_NP;
  const int CHs = (CH + 31) & ~31;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t nlist = nlist_arg, per_xcd = (nlist + 7) / 8;
  const int64_t li = (int64_t) (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (li >= nlist)
    return;
  const int64_t b = dense_blocks[li];
  const int64_t r0 = b * MM_RB, r1 = (r0 + MM_RB) < m ? (r0 + MM_RB) : m;
  const int2 w = win[b];
  const int kb = w.x, W = w.y - w.x + 1, Wp = (W + 3) & ~3;  // (inspect: Wp <= CH, n a multiple of 4 handled below)
  const int64_t rw0 = r0 + (int64_t) wave * RPW;
  O p[RPW + 1];
#pragma unroll
  for (int i = 0; i <= RPW; ++i)
    p[i] = rowptr[(rw0 + i) < r1 ? (rw0 + i) : r1];
  int ecol[RPW], eval_[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const O q = p[i] + (O) lane;
    const bool ok = q < p[i + 1];
    ecol[i] = ok ? colind[q] : kb;
    eval_[i] = ok ? __float_as_int(values[q]) : 0;
  }
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(B) | (uintptr_t) (ldb * 4)) & 15) == 0;
  const bool c2_ok = ((reinterpret_cast<uintptr_t>(C) | (uintptr_t) (ldc * 4)) & 7) == 0;
  // the dense A tile does not depend on the column pass: built once per block
  for (int q = tid * 4; q < CHs * MM_RB; q += NW * 64 * 4)
    *reinterpret_cast<f32x4*>(At + q) = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  if (!(dbg & 2)) {  // (dbg: timing experiments, results wrong)
    typedef __attribute__((address_space(3))) volatile int lds_vint;  // (explicit LDS pointers: DESIGN section 5)
    bool dup = false;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int len = (int) (p[i + 1] - p[i]);
      for (int base = 0; base < len; base += 64) {
        int cv = ecol[i];
        if (base != 0) {
          const O q = p[i] + (O) (base + lane);
          cv = q < p[i + 1] ? colind[q] : kb;
        }
        const bool ok = base + lane < len;
        const int row = wave * RPW + i;
        lds_vint* slot = (lds_vint*) (reinterpret_cast<int*>(At) + row * CHs + ((cv - kb) ^ (2 * (row & 15))));
        const int tag = base + lane + 1;
        if (ok)
          *slot = tag;
        if (ok)
          dup |= *slot != tag;
      }
    }
    dup = __ballot(dup) != 0ull;  // (rows are wave-private: a duplicate concerns this wave only)
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int len = (int) (p[i + 1] - p[i]);
      for (int base = 0; base < len; base += 64) {
        int cv = ecol[i], vi = eval_[i];
        if (base != 0) {
          const O q = p[i] + (O) (base + lane);
          const bool okq = q < p[i + 1];
          cv = okq ? colind[q] : kb;
          vi = okq ? __float_as_int(values[q]) : 0;
        }
        const int row = wave * RPW + i;
        if (base + lane < len)
          At[row * CHs + ((cv - kb) ^ (2 * (row & 15)))] = dup ? 0.f : __int_as_float(vi);
      }
      if (dup)
        for (int base = 0; base < len; base += 64) {
          const O q = p[i] + (O) (base + lane);
          const int row = wave * RPW + i;
          if (q < p[i + 1])
            unsafeAtomicAdd(At + row * CHs + ((colind[q] - kb) ^ (2 * (row & 15))), values[q]);
        }
    }
  }
  for (int64_t c0 = 0; c0 < n; c0 += MM_NP) {
    const int ncol = (int) ((n - c0) < MM_NP ? (n - c0) : MM_NP);
    __syncthreads();  // At complete / the previous pass's window no longer read
    if (vec_ok && ncol == MM_NP) {
      const int npair = (W + 1) >> 1;
      for (int q = wave; q < npair; q += NW) {
        const int kk = 2 * q + (lane >> 5);
        if (kk < W)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*) (B + (int64_t) (kb + kk) * ldb + c0 + 4 * (lane & 31)),
              (__attribute__((address_space(3))) void*) (Bt + (size_t) q * 2 * MM_NP), 16, 0, 0);
      }
      for (int q = tid; q < (Wp - W) * MM_NP; q += NW * 64)
        Bt[(size_t) W * MM_NP + q] = 0.f;
      __builtin_amdgcn_s_waitcnt(0);
    } else {
      for (int q = tid; q < Wp * MM_NP; q += NW * 64) {
        const int kk = q / MM_NP, jj = q % MM_NP;
        Bt[q] = (kk < W && jj < ncol) ? B[(int64_t) (kb + kk) * ldb + c0 + jj] : 0.f;
      }
    }
    __syncthreads();
    typedef float mb_f4 __attribute__((ext_vector_type(4)));
    mb_f4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    // A operand of lane l: At[4 s + (l >> 4)][l & 15] (rows 0..15) / [.. + 16] (rows 16..31); B: Bt[4 s + (l >> 4)][16 w + (l & 15)]
    const float* ap0 = At + (lane & 15) * CHs, *ap1 = ap0 + 16 * CHs;
    const int ksw = 2 * (lane & 15), kl = lane >> 4;
    const float* bp = Bt + (lane >> 4) * MM_NP + 16 * wave + (lane & 15);
    const int nsteps = (dbg & 1) ? 0 : Wp / 4;
    int sidx = 0;
    for (; sidx + 8 <= nsteps; sidx += 8) {  // (unrolled by hand: twenty-four LDS reads in flight, then sixteen MFMAs)
      float a0[8], a1[8], bb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        bb[u] = bp[(sidx + u) * 4 * MM_NP];
        a0[u] = ap0[((sidx + u) * 4 + kl) ^ ksw];
        a1[u] = ap1[((sidx + u) * 4 + kl) ^ ksw];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], bb[u], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], bb[u], d1, 0, 0, 0);
      }
    }
    for (; sidx < nsteps; ++sidx) {
      const float bb = bp[sidx * 4 * MM_NP];
      d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap0[(sidx * 4 + kl) ^ ksw], bb, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap1[(sidx * 4 + kl) ^ ksw], bb, d1, 0, 0, 0);
    }
    bool bad = false;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bad |= !(__builtin_fabsf(d0[q]) <= 3.402823466e38f) | !(__builtin_fabsf(d1[q]) <= 3.402823466e38f);
    if (dbg & 8)
      bad = false;
    if (!__syncthreads_or(bad)) {
      // D of lane l: rows 4 (l >> 4) + q (+ 16), column 16 w + (l & 15) -- 64-byte pieces of eight rows per wave.  The block
      // goes through LDS (the window is no longer read: the barrier above) and leaves as whole 512-byte rows.
      constexpr int CS = MM_NP + 4;  // row stride of the staged block (two-way bank conflicts at most)
      float* Ct = Bt;
      const int jc = 16 * wave + (lane & 15);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        Ct[(4 * (lane >> 4) + q) * CS + jc] = d0[q];
        Ct[(4 * (lane >> 4) + q + 16) * CS + jc] = d1[q];
      }
      __syncthreads();
      const int jj = 2 * lane;
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int64_t row = rw0 + i;
        if (row < r1 && jj < ncol) {
          const float sx = Ct[(wave * RPW + i) * CS + jj], sy = Ct[(wave * RPW + i) * CS + jj + 1];
          float* cp = C + row * ldc + c0 + jj;
          if (c2_ok && jj + 1 < ncol) {
            mb_f2 o = mb_f2{alpha * sx, alpha * sy};
            if (beta != 0.f)
              o += beta * *reinterpret_cast<const mb_f2*>(cp);
            *reinterpret_cast<mb_f2*>(cp) = o;
          } else {
            cp[0] = beta == 0.f ? alpha * sx : alpha * sx + beta * cp[0];
            if (jj + 1 < ncol)
              cp[1] = beta == 0.f ? alpha * sy : alpha * sy + beta * cp[1];
          }
        }
      }
    } else {
      // a non-finite value in the window (or an overflow): entry by entry, one row at a time, lane = two output columns
      for (int i = 0; i < RPW; ++i) {
        const int64_t row = rw0 + i;
        if (row >= r1)
          break;
        float sx = 0.f, sy = 0.f;
        const O q0 = rowptr[row], q1 = rowptr[row + 1];
        for (O q = q0; q < q1; ++q) {
          const float a = values[q];
          const float* br = Bt + (size_t) (colind[q] - kb) * MM_NP + 2 * lane;
          sx = __builtin_fmaf(a, br[0], sx);
          sy = __builtin_fmaf(a, br[1], sy);
        }
        const int jj = 2 * lane;
        float* cp = C + row * ldc + c0 + jj;
        if (jj < ncol)
          cp[0] = beta == 0.f ? alpha * sx : alpha * sx + beta * cp[0];
        if (jj + 1 < ncol)
          cp[1] = beta == 0.f ? alpha * sy : alpha * sy + beta * cp[1];
      }
    }
  }
}


template <typename T, typename O>
static int spmm_typed(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, int64_t m, int64_t k, int64_t n, int64_t nnz,
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
  // what multiply_inspect decided: rows longer than the nnz window go to the long-row kernels, qualifying row
  // blocks (fp32) to the panel kernel; everything else (and every row without a plan) to the row-group kernel
  const bool planned = pl && pl->mm_ready;
  const int long_len = planned && pl->n_long > 0 ? pl->win : 0;
  const unsigned char* is_panel = planned && pl->mm_npanel > 0 && sizeof(T) == 4 ? pl->mm_is_panel : nullptr;
  if (long_len > 0) {
    // ~4 K entries per workgroup, at most 64 parts per row; the partial rows live in the plan (grown on demand)
    int64_t parts = cdiv(pl->max_row_len, 4096);
    parts = parts < 1 ? 1 : (parts > 64 ? 64 : parts);
    const int64_t need = pl->n_long * parts * n;
    if (pl->mm_long_cap < need || pl->mm_long_parts != (int) parts) {
      if (stream_capturing(s))  // the first call with this many columns has to run outside the capture
        return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
      dev_free(pl->mm_long_part, s);
      pl->mm_long_part = nullptr;
      pl->mm_long_cap = 0;
      int rc = dev_alloc(&pl->mm_long_part, (size_t) need * sizeof(T), s);
      if (rc)
        return rc;
      pl->mm_long_cap = need;
      pl->mm_long_parts = (int) parts;
    }
  }
  const bool all_panel = is_panel && pl->mm_npanel == pl->mm_nblk;  // nothing left for the row-group kernel
  if (all_panel) {
  } else if constexpr (sizeof(T) == 4) {
    if (V == 4)
      launch_spmm<T, O, 4>(s, m, n, rowptr, colind, values, B, ldb, C, ldc, alpha, beta, is_panel, long_len);
    else if (V == 2)
      launch_spmm<T, O, 2>(s, m, n, rowptr, colind, values, B, ldb, C, ldc, alpha, beta, is_panel, long_len);
    else
      launch_spmm<T, O, 1>(s, m, n, rowptr, colind, values, B, ldb, C, ldc, alpha, beta, is_panel, long_len);
  } else {
    if (V == 2)
      launch_spmm<T, O, 2>(s, m, n, rowptr, colind, values, B, ldb, C, ldc, alpha, beta, is_panel, long_len);
    else
      launch_spmm<T, O, 1>(s, m, n, rowptr, colind, values, B, ldb, C, ldc, alpha, beta, is_panel, long_len);
  }
  if constexpr (sizeof(T) == 4) {
    if (is_panel && pl->mm_band) {
      int ch = pl->mm_band_ch;
      int64_t mm = m, nn = n, kk_ = k, ldb_ = ldb, ldc_ = ldc;
      const O* rp = rowptr;
      const int32_t* ci = colind;
      const float* va = reinterpret_cast<const float*>(values);
      const float* Bp = reinterpret_cast<const float*>(B);
      float* Cp = reinterpret_cast<float*>(C);
      float al = (float) alpha, be = (float) beta;
      const int2* wn = static_cast<const int2*>(pl->mm_win);
      const int64_t n_vec = pl->mm_npanel - pl->mm_ndense;
      if (n_vec > 0) {
        const void* fn = pl->mm_band_waves == 4    ? reinterpret_cast<const void*>(spmm_band_kernel<O, 4>)
                         : pl->mm_band_waves == 16 ? reinterpret_cast<const void*>(spmm_band_kernel<O, 16>)
                                                   : reinterpret_cast<const void*>(spmm_band_kernel<O, 8>);
        int64_t nl = n_vec;
        const int32_t* pb = pl->mm_vec_blocks;
        const int32_t* tl = pl->mm_tiles;
        int dbg = env_flag("SPBLAS_GFX950_SPMM_DBG") ? std::atoi(std::getenv("SPBLAS_GFX950_SPMM_DBG")) : 0;
        void* args[] = {&mm, &nn, &kk_, &rp, &ci, &va, &Bp, &ldb_, &Cp, &ldc_, &al, &be, &pb, &wn, &tl, &nl, &ch, &dbg};
        SPB_HIP(hipLaunchKernel(fn, dim3((unsigned) (8 * cdiv(n_vec, 8))), dim3(pl->mm_band_waves * 64), args,
                                (size_t) ch * MM_NP * sizeof(float) + (size_t) pl->mm_band_waves * 512, s));
      }
      if (pl->mm_ndense > 0) {
        int64_t nl = pl->mm_ndense;
        const int32_t* db = pl->mm_dense_blocks;
        int dbg = env_flag("SPBLAS_GFX950_SPMM_DBG") ? std::atoi(std::getenv("SPBLAS_GFX950_SPMM_DBG")) : 0;
        void* args[] = {&mm, &nn, &rp, &ci, &va, &Bp, &ldb_, &Cp, &ldc_, &al, &be, &db, &wn, &nl, &ch, &dbg};
        SPB_HIP(hipLaunchKernel(reinterpret_cast<const void*>(spmm_band_mfma_kernel<O>), dim3((unsigned) (8 * cdiv(nl, 8))),
                                dim3(512), args, ((size_t) ch * MM_NP + (size_t) ((ch + 31) & ~31) * MM_RB) * sizeof(float), s));
      }
    } else if (is_panel) {
      const size_t lds = (size_t) (MM_KT * MM_NP + MM_KT * MM_RB + 4 + MM_RB + 4) * sizeof(float);
      hipLaunchKernelGGL((spmm_panel_kernel<O>), dim3((unsigned) (8 * cdiv(pl->mm_npanel, 8))), dim3(256), lds, s, m, k, n,
                         rowptr, colind, values, B, ldb, C, ldc, alpha, beta, pl->mm_panel_blocks, pl->mm_tiles,
                         pl->mm_npanel, env_flag("SPBLAS_GFX950_SPMM_DBG") ? std::atoi(std::getenv("SPBLAS_GFX950_SPMM_DBG")) : 0);
    }
  }
  if (long_len > 0) {
    T* part = static_cast<T*>(pl->mm_long_part);
    hipLaunchKernelGGL((spmm_long_rows_kernel<T, O>), dim3((unsigned) pl->n_long, (unsigned) pl->mm_long_parts), dim3(256), 0,
                       s, pl->long_rows, pl->mm_long_parts, n, rowptr, colind, values, B, ldb, part);
    hipLaunchKernelGGL((spmm_long_finish_kernel<T>), dim3((unsigned) pl->n_long), dim3(256), 0, s, pl->long_rows,
                       pl->mm_long_parts, n, part, C, ldc, alpha, beta);
  }
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}


// Dense operands of any mdspan layout (layout_left B and / or C): element (i, j) at i*rs + j*cs.  A group of G lanes owns
// one row of A; the lanes walk the row's entries G at a time and keep JT output columns in registers, so every (column,
// value) pair is loaded once per tile of JT columns and every gather is one element -- for a layout_left B the JT gathers
// of an entry are JT different lines, which is what that layout costs.  Group reduction by shuffles, lane j of the group
// writes column j of the tile.  The maths (per row, entries in storage order within a lane, lanes combined in a fixed
// tree) is multiply_impl.hpp:66-92 up to the association of the row sum.
template <typename T, typename O, int JT>
__global__ __launch_bounds__(256) void spmm_strided_kernel(int64_t m, int64_t n, const O* __restrict__ rowptr,
                                                           const int32_t* __restrict__ colind,
                                                           const T* __restrict__ values, const T* __restrict__ B,
                                                           int64_t brs, int64_t bcs, T* __restrict__ C, int64_t crs,
                                                           int64_t ccs, T alpha, T beta, int G) {
  const int64_t row = (int64_t) blockIdx.x * (256 / G) + threadIdx.x / G;
  const int lig = threadIdx.x % G;
  O p0 = 0, p1 = 0;
  if (row < m) {
    p0 = rowptr[row];
    p1 = rowptr[row + 1];
  }
  for (int64_t j0 = (int64_t) blockIdx.y * JT; j0 < n; j0 += (int64_t) gridDim.y * JT) {
    T acc[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j)
      acc[j] = T(0);
    for (O p = p0 + lig; p < p1; p += G) {
      const int64_t c = colind[p];
      const T v = values[p];
      const T* __restrict__ bp = B + c * brs + j0 * bcs;
#pragma unroll
      for (int j = 0; j < JT; ++j)
        if (j0 + j < n)
          acc[j] += v * bp[j * bcs];
    }
#pragma unroll
    for (int j = 0; j < JT; ++j)
      for (int o = G >> 1; o > 0; o >>= 1)
        acc[j] += __shfl_xor(acc[j], o, G);
    if (row < m) {
#pragma unroll
      for (int j = 0; j < JT; ++j)
        if (lig == (j % G) && j0 + j < n) {
          T* cp = C + row * crs + (j0 + j) * ccs;
          *cp = beta == T(0) ? alpha * acc[j] : alpha * acc[j] + beta * *cp;
        }
    }
  }
}

template <typename T, typename O>
static int spmm_strided_typed(spblas_gfx950_handle_t h, int64_t m, int64_t n, int64_t nnz, const void* alpha_p,
                              const void* rowptr_v, const int32_t* colind, const void* values_v, const void* B_v,
                              int64_t brs, int64_t bcs, const void* beta_p, void* C_v, int64_t crs, int64_t ccs) {
  if (m == 0 || n == 0)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  const T alpha = *static_cast<const T*>(alpha_p), beta = *static_cast<const T*>(beta_p);
  int G = 2;
  const int64_t avg = m > 0 ? nnz / m : 0;
  while (G < 64 && G < avg)
    G <<= 1;
  constexpr int JT = 8;
  const int64_t tiles = cdiv(n, JT);
  hipLaunchKernelGGL((spmm_strided_kernel<T, O, JT>), dim3((unsigned) cdiv(m, 256 / G), (unsigned) (tiles < 64 ? tiles : 64)),
                     dim3(256), 0, h->stream, m, n, static_cast<const O*>(rowptr_v), colind, static_cast<const T*>(values_v),
                     static_cast<const T*>(B_v), brs, bcs, static_cast<T*>(C_v), crs, ccs, alpha, beta, G);
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

template <typename O>
static int spmm_inspect_typed(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl) {
  hipStream_t s = h->stream;
  pl->mm_ready = 1;
  pl->mm_nblk = cdiv(pl->m, MM_RB);
  pl->mm_npanel = 0;
  pl->mm_panel_nnz = 0;
  // the panel kernel is fp32 only (v_mfma_f32_32x32x2_f32); a block needs >= 256 entries to qualify at all
  if (pl->value_type != SPBLAS_GFX950_F32 || pl->mm_nblk == 0 || pl->nnz < 256 || env_flag("SPBLAS_GFX950_SPMM_NO_PANEL"))
    return SPBLAS_GFX950_STATUS_SUCCESS;
  int rc;
  // entries a block must hold per 32 x 64 tile it touches.  Rounds 3 - 5 (tile kernel on the matrix cores): 1/5 dense, the measured crossover with the row-group
  // kernel on banded matrices (tools/spmm_density.py: 1 M rows, n = 128 -- 0.94 vs 0.98 ms at 0.2, 1.49 vs 2.50 ms
  // at 0.6, 0.78 vs 0.41 ms at 0.05); SPBLAS_GFX950_SPMM_PANEL_MIN overrides it (tests)
  // round 6, band kernel (same tool, profiles/r06_spmm_band.md): 0.38 vs 0.41 ms at 0.05, 0.43 vs 0.61 at 0.1, 0.54 vs 0.99 at
  // 0.2, 1.10 vs 2.51 at 0.6 (0.38 vs 0.31 at 0.025): admission from 1/12
  int min_per_tile = MM_RB * MM_KT / 12;
  if (const char* e = std::getenv("SPBLAS_GFX950_SPMM_PANEL_MIN"))
    min_per_tile = std::atoi(e) > 0 ? std::atoi(e) : min_per_tile;
  // which kernel takes the qualifying blocks: the band kernels (round 6: entry loop over the stored entries, matrix cores for
  // blocks dense in their window) unless SPBLAS_GFX950_SPMM_BAND=0 asks for the tile kernel of rounds 3 - 5.
  // SPBLAS_GFX950_SPMM_BAND_CH: B rows staged per chunk (128 rows: 64 KiB of B + 16 KiB for the dense A tile or the hand-over
  // buffers = two workgroups per CU; a +-48 band fits); _WAVES: wavefronts per workgroup of the entry-loop kernel (4, 8, 16);
  // _DENSE: window density, per mille, from which a block goes to the matrix cores (0: all that fit, 1001: none)
  {
    const char* e = std::getenv("SPBLAS_GFX950_SPMM_BAND");
    pl->mm_band = !(e && std::atoi(e) == 0);
    const char* c = std::getenv("SPBLAS_GFX950_SPMM_BAND_CH");
    int ch = c ? std::atoi(c) : 128;
    pl->mm_band_ch = ch < 64 ? 64 : (ch > 312 ? 312 : ch);  // (>= one aligned tile of 64 rows: the wide-window path)
    const char* wv = std::getenv("SPBLAS_GFX950_SPMM_BAND_WAVES");
    pl->mm_band_waves = (wv && std::atoi(wv) == 4) ? 4 : (wv && std::atoi(wv) == 16) ? 16 : 8;
    const char* dn = std::getenv("SPBLAS_GFX950_SPMM_BAND_DENSE");
    // (measured on the banded bench, 2 M rows, 64 entries per row in a window of 128: entry loop 1.41 ms, matrix-core kernel
    // 2.38 ms -- tile build 0.70, contraction 0.59, the rest a skeleton of five barriers and three dependent round trips per
    // block at two workgroups per CU -- so the default sends no block there; SPBLAS_GFX950_SPMM_BAND_DENSE=250 opts in)
    pl->mm_band_dense_pm = !pl->mm_band ? 1001 : dn ? std::atoi(dn) : 1001;
  }
  unsigned long long* d_cnt = nullptr;
  int32_t* flag32 = nullptr;
  int32_t* flag_dense = nullptr;
  long long* partials = nullptr;
  if ((rc = dev_alloc((void**) &pl->mm_is_panel, (size_t) pl->mm_nblk, s)) ||
      (rc = dev_alloc((void**) &pl->mm_tiles, (size_t) pl->mm_nblk * MM_TS * 4, s)) ||
      (rc = dev_alloc((void**) &pl->mm_panel_blocks, (size_t) pl->mm_nblk * 4, s)) ||
      (rc = dev_alloc(&pl->mm_win, (size_t) pl->mm_nblk * sizeof(int2), s)) ||
      (rc = dev_alloc((void**) &pl->mm_vec_blocks, (size_t) pl->mm_nblk * 4, s)) ||
      (rc = dev_alloc((void**) &pl->mm_dense_blocks, (size_t) pl->mm_nblk * 4, s)) ||
      (rc = dev_alloc((void**) &d_cnt, 16, s)))
    return rc;
  if ((rc = dev_alloc((void**) &flag32, (size_t) (pl->mm_nblk + 1) * 4, s)) ||
      (rc = dev_alloc((void**) &flag_dense, (size_t) (pl->mm_nblk + 1) * 4, s)) ||
      (rc = dev_alloc((void**) &partials, (size_t) (cdiv(pl->mm_nblk, 2048) + 2) * sizeof(long long), s))) {
    dev_free(d_cnt, s);
    dev_free(flag32, s);
    dev_free(flag_dense, s);
    return rc;
  }
  SPB_HIP(hipMemsetAsync(d_cnt, 0, 16, s));
  hipLaunchKernelGGL((spmm_panel_probe_kernel<O>), dim3((unsigned) pl->mm_nblk), dim3(64), 0, s, pl->m, pl->mm_nblk,
                     static_cast<const O*>(pl->rowptr), pl->colind, pl->n_long > 0 ? pl->win : 0, min_per_tile,
                     pl->mm_tiles, pl->mm_is_panel, flag32, d_cnt, static_cast<int2*>(pl->mm_win), flag_dense, pl->mm_band_ch,
                     pl->mm_band_dense_pm);
  (void) scan_counts_i32(s, pl->mm_nblk, flag32, partials);  // flag32[nblk] = number of qualifying blocks
  hipLaunchKernelGGL(spmm_panel_list_kernel, dim3((unsigned) cdiv(pl->mm_nblk, 256)), dim3(256), 0, s, pl->mm_nblk,
                     pl->mm_is_panel, flag32, pl->mm_panel_blocks);
  unsigned long long h_cnt[2] = {0, 0};
  int32_t h_np = 0, h_nd = 0;
  hipError_t e = hipMemcpyAsync(h_cnt, d_cnt, 16, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess)
    e = hipMemcpyAsync(&h_np, flag32 + pl->mm_nblk, 4, hipMemcpyDeviceToHost, s);
  // (the second scan reuses `partials`: the copy of the first total above is ordered before these kernels)
  (void) scan_counts_i32(s, pl->mm_nblk, flag_dense, partials);
  hipLaunchKernelGGL(spmm_band_lists_kernel, dim3((unsigned) cdiv(pl->mm_nblk, 256)), dim3(256), 0, s, pl->mm_nblk,
                     pl->mm_is_panel, flag32, flag_dense, pl->mm_vec_blocks, pl->mm_dense_blocks);
  if (e == hipSuccess)
    e = hipMemcpyAsync(&h_nd, flag_dense + pl->mm_nblk, 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess)
    e = hipStreamSynchronize(s);
  dev_free(d_cnt, s);
  dev_free(flag32, s);
  dev_free(flag_dense, s);
  dev_free(partials, s);
  if (e != hipSuccess)
    return hip_fail(e);
  pl->mm_npanel = (int64_t) h_np;
  pl->mm_ndense = (int64_t) h_nd;
  pl->mm_panel_nnz = (int64_t) h_cnt[1];
  pl->device_bytes += (size_t) pl->mm_nblk * (1 + 4 * MM_TS + 4 + sizeof(int2) + 8);
  {
    const int lds = pl->mm_band_ch * MM_NP * (int) sizeof(float) +
                    std::max(((pl->mm_band_ch + 31) & ~31) * MM_RB * (int) sizeof(float), 16 * 512);
    SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(spmm_band_kernel<O, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(spmm_band_kernel<O, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(spmm_band_kernel<O, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(spmm_band_mfma_kernel<O>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(spmm_panel_kernel<O>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (MM_KT * MM_NP + MM_KT * MM_RB + 4 + MM_RB + 4) * (int) sizeof(float)));
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

void spmm_plan_free(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl) {
  hipStream_t s = h->stream;
  dev_free(pl->mm_is_panel, s);
  dev_free(pl->mm_tiles, s);
  dev_free(pl->mm_panel_blocks, s);
  dev_free(pl->mm_win, s);
  dev_free(pl->mm_vec_blocks, s);
  dev_free(pl->mm_dense_blocks, s);
  pl->mm_win = nullptr;
  pl->mm_vec_blocks = pl->mm_dense_blocks = nullptr;
  dev_free(pl->mm_long_part, s);
  pl->mm_is_panel = nullptr;
  pl->mm_tiles = pl->mm_panel_blocks = nullptr;
  pl->mm_long_part = nullptr;
  pl->mm_long_cap = 0;
  pl->mm_ready = 0;
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
               plan->colind != colind || plan->offset_type != offset_type || plan->value_type != value_type))
    return SPBLAS_GFX950_STATUS_PLAN_MISMATCH;
  if (plan) {  // the long-row partials are the plan's
    plan->last_stream = handle->stream;
    plan->used = true;
  }
  if (value_type == SPBLAS_GFX950_F32) {
    return offset_type == SPBLAS_GFX950_I32
               ? spmm_typed<float, int32_t>(handle, plan, m, k, n, nnz, alpha, rowptr, colind, values, B, ldb, beta, C, ldc)
               : spmm_typed<float, int64_t>(handle, plan, m, k, n, nnz, alpha, rowptr, colind, values, B, ldb, beta, C, ldc);
  }
  return offset_type == SPBLAS_GFX950_I32
             ? spmm_typed<double, int32_t>(handle, plan, m, k, n, nnz, alpha, rowptr, colind, values, B, ldb, beta, C, ldc)
             : spmm_typed<double, int64_t>(handle, plan, m, k, n, nnz, alpha, rowptr, colind, values, B, ldb, beta, C, ldc);
}

extern "C" int spblas_gfx950_spmm_strided(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, int64_t m, int64_t k,
                                          int64_t n, int64_t nnz, const void* alpha, const void* rowptr,
                                          const int32_t* colind, const void* values, const void* B, int64_t brs, int64_t bcs,
                                          const void* beta, void* C, int64_t crs, int64_t ccs, int offset_type,
                                          int value_type) {
  if (bcs == 1 && ccs == 1) {  // both layout_right: the regular kernels, plan included
    // (a layout_left mdspan with ONE row also has strides (1, 1): its row stride says nothing -- an operand of at most one
    // row gets the leading dimension the regular entry point asks for; round-4 advisor finding)
    const int64_t ldb = k <= 1 ? (n > 1 ? n : 1) : brs, ldc = m <= 1 ? (n > 1 ? n : 1) : crs;
    return spblas_gfx950_spmm(handle, plan, m, k, n, nnz, alpha, rowptr, colind, values, B, ldb, beta, C, ldc, offset_type,
                              value_type);
  }
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (m < 0 || k < 0 || n < 0 || nnz < 0 || m > INT32_MAX || k > INT32_MAX)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  // each operand is layout_right (column stride 1, rows >= n apart) or layout_left (row stride 1, columns >= rows apart)
  const auto layout_ok = [n](int64_t rows, int64_t rs, int64_t cs) {
    return (cs == 1 && rs >= n) || (rs == 1 && cs >= rows) || rows <= 1 || n <= 1;
  };
  if (brs < 0 || bcs < 0 || crs < 0 || ccs < 0 || !layout_ok(k, brs, bcs) || !layout_ok(m, crs, ccs))
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  if (offset_type == SPBLAS_GFX950_I32 && nnz > INT32_MAX)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  if ((offset_type != SPBLAS_GFX950_I32 && offset_type != SPBLAS_GFX950_I64) ||
      (value_type != SPBLAS_GFX950_F32 && value_type != SPBLAS_GFX950_F64))
    return SPBLAS_GFX950_STATUS_INVALID_VALUE;
  if (!alpha || !beta || !rowptr || (nnz > 0 && (!colind || !values || !B)) || (m > 0 && n > 0 && !C))
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (plan && (plan->m != m || plan->n != k || plan->nnz != nnz || plan->rowptr != rowptr ||
               plan->colind != colind || plan->offset_type != offset_type || plan->value_type != value_type))
    return SPBLAS_GFX950_STATUS_PLAN_MISMATCH;
  if (value_type == SPBLAS_GFX950_F32)
    return offset_type == SPBLAS_GFX950_I32
               ? spmm_strided_typed<float, int32_t>(handle, m, n, nnz, alpha, rowptr, colind, values, B, brs, bcs, beta, C, crs, ccs)
               : spmm_strided_typed<float, int64_t>(handle, m, n, nnz, alpha, rowptr, colind, values, B, brs, bcs, beta, C, crs, ccs);
  return offset_type == SPBLAS_GFX950_I32
             ? spmm_strided_typed<double, int32_t>(handle, m, n, nnz, alpha, rowptr, colind, values, B, brs, bcs, beta, C, crs, ccs)
             : spmm_strided_typed<double, int64_t>(handle, m, n, nnz, alpha, rowptr, colind, values, B, brs, bcs, beta, C, crs, ccs);
}

extern "C" int spblas_gfx950_spmm_inspect(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!plan)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (plan->mm_ready)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  if (stream_capturing(handle->stream))  // inspect-class call: sizes its output on the host, never part of a graph
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  const int rc = plan->offset_type == SPBLAS_GFX950_I32 ? spmm_inspect_typed<int32_t>(handle, plan)
                                                         : spmm_inspect_typed<int64_t>(handle, plan);
  if (rc != SPBLAS_GFX950_STATUS_SUCCESS)
    spmm_plan_free(handle, plan);  // a failed inspect leaves no half-built panel tables behind and is retried (mm_ready = 0)
  return rc;
}

extern "C" int spblas_gfx950_spmm_plan_info(spblas_gfx950_plan_t plan, int64_t info[4]) {
  if (!plan || !info)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  info[0] = plan->mm_ready;
  info[1] = plan->mm_npanel;      // row blocks (32 rows) multiplied on the matrix cores
  info[2] = plan->mm_panel_nnz;   // entries inside them
  info[3] = plan->mm_ready ? plan->n_long : 0;  // rows cut into parts by the long-row kernel
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

// Loads this file's code object (the runtime loads a code object at the first use of one of its kernels: milliseconds
// that would otherwise fall on the caller's first inspect / compute call -- handle.hip: spblas_gfx950_create).
namespace spb {
void preload_spmm() {
  hipFuncAttributes attr;
  (void) hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&scale_matrix_kernel<float>));
  (void) hipGetLastError();
}
} // namespace spb
