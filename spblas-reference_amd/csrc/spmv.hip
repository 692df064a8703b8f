// CSR SpMV for gfx950:  y = alpha * op(A) * x + beta * y.
//
// Replaces rocsparse_spmv(..., rocsparse_spmv_alg_csr_stream, ...) at
// /root/reference/include/spblas/vendor/rocsparse/detail/spmv_impl.hpp:60-77; the
// maths is the reference CPU path include/spblas/algorithms/multiply_impl.hpp:33-53
// with alpha folded to one scalar as the rocSPARSE slot does (spmv_impl.hpp:35-37).
//
// Kernels (all HBM-bound; algorithmic bytes per nonzero = sizeof(T)+4, per row =
// sizeof(O)+sizeof(T), per column = sizeof(T)):
//   spmv_vector_kernel    plan-free; a power-of-two group of lanes per row.
//   spmv_rowblock_kernel  plan (multiply_inspect); one 256-thread workgroup per
//                         nnz window: coalesced 16-byte streaming loads of
//                         colind/values, x gathers, products staged in LDS, then a
//                         sub-wavefront group per row reduces out of LDS.
//   spmv_long_fixup_kernel sums the per-window partials of rows longer than a window.
//   spmv_transpose_kernel  op = T (CSC / transposed(csr)): scatter with HW float atomics.
#include "common.hpp"
#include "plan.hpp"
#include "scan.hpp"

#include <atomic>
#include <cstdlib>

namespace spb {

// ---------------------------------------------------------------------------
// plan-free kernel: LPR lanes per row
// ---------------------------------------------------------------------------
template <typename T, typename O, int LPR>
__global__ __launch_bounds__(256) void spmv_vector_kernel(int64_t m, const O* __restrict__ rowptr,
                                                          const int32_t* __restrict__ colind,
                                                          const T* __restrict__ values,
                                                          const T* __restrict__ x, T* __restrict__ y,
                                                          T alpha, T beta) {
  constexpr int ROWS = 256 / LPR;
  const int64_t row = (int64_t) blockIdx.x * ROWS + threadIdx.x / LPR;
  const int lane = threadIdx.x % LPR;
  T s = 0;
  if (row < m) {
    const O p0 = rowptr[row], p1 = rowptr[row + 1];
    for (O p = p0 + lane; p < p1; p += LPR)
      s += stream_load(values + p) * x[stream_load(colind + p)];
  }
  s = group_sum_c<LPR>(s);
  if (row < m && lane == 0)
    y[row] = beta == T(0) ? alpha * s : alpha * s + beta * y[row];
}

// ---------------------------------------------------------------------------
// row-block kernel
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void load4_stream(const T* p, T (&out)[4]);
template <>
__device__ __forceinline__ void load4_stream<float>(const float* p, float (&out)[4]) {
  f32x4 v = stream_load(reinterpret_cast<const f32x4*>(p));
  out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
template <>
__device__ __forceinline__ void load4_stream<double>(const double* p, double (&out)[4]) {
  f64x2 a = stream_load(reinterpret_cast<const f64x2*>(p));
  f64x2 b = stream_load(reinterpret_cast<const f64x2*>(p) + 1);
  out[0] = a.x; out[1] = a.y; out[2] = b.x; out[3] = b.y;
}

// Sum of values[p]*x[colind[p]] for p in [lo, hi) over the whole workgroup.
// Result valid in thread 0.
template <typename T, typename O>
__device__ T block_segment_dot(O lo, O hi, const int32_t* __restrict__ colind,
                               const T* __restrict__ values, const T* __restrict__ x, T* red) {
  T s = 0;
  for (O p = lo + (O) threadIdx.x; p < hi; p += 256)
    s += stream_load(values + p) * x[stream_load(colind + p)];
  s = group_sum_c<64>(s);
  __syncthreads();
  if ((threadIdx.x & 63) == 0)
    red[threadIdx.x >> 6] = s;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// Window w owns the rows whose first entry lies in [w*WIN, (w+1)*WIN).  A row no
// longer than WIN is computed entirely by its owner (its entries end before
// (w+2)*WIN, hence the 2*WIN LDS slots).  A row longer than WIN ("long") is
// split: every window it covers reduces its own slice into part_tail[w] (the
// window where the row starts) or part_head[w] (later windows).
template <typename T, typename O, int WIN, bool HAS_LONG, bool VEC>
__global__ __launch_bounds__(256) void spmv_rowblock_kernel(
    int64_t nnz, const O* __restrict__ rowptr, const int32_t* __restrict__ colind,
    const T* __restrict__ values, const T* __restrict__ x, T* __restrict__ y, T alpha, T beta,
    const int32_t* __restrict__ win_row, T* __restrict__ part_head, T* __restrict__ part_tail) {
  constexpr int CAP = 2 * WIN;
  constexpr int ITERS = CAP / 4 / 256;
  static_assert(CAP % 1024 == 0, "window must be a multiple of 512");
  __shared__ T prod[CAP];
  __shared__ T red[4];

  const int tid = threadIdx.x;
  const int64_t w = blockIdx.x;
  const int r_begin = win_row[w];
  int r_end = win_row[w + 1];
  const O wlo = (O) (w * WIN);
  const O whi = (O) ((w + 1) * WIN < nnz ? (w + 1) * WIN : nnz);

  const O a = rowptr[r_begin];  // first entry of the first owned row (>= wlo)
  O e = rowptr[r_end];          // one past the last entry of the last owned row

  if (HAS_LONG) {
    // long row entering this window from an earlier one
    if (r_begin > 0 && a > wlo) {
      const O hs = rowptr[r_begin - 1];
      if (a - hs > (O) WIN) {
        T s = block_segment_dot<T, O>(wlo, a < whi ? a : whi, colind, values, x, red);
        if (tid == 0)
          part_head[w] = s;
        __syncthreads();
      }
    }
    // long row starting in this window (necessarily the last owned row)
    if (r_end > r_begin) {
      const O ls = rowptr[r_end - 1];
      if (e - ls > (O) WIN) {
        T s = block_segment_dot<T, O>(ls, whi, colind, values, x, red);
        if (tid == 0)
          part_tail[w] = s;
        __syncthreads();
        r_end -= 1;
        e = ls;
      }
    }
  }

  const O a_al = a & ~(O) 3;  // 16-byte aligned start; a_al >= wlo because WIN % 4 == 0
  const int total = (int) (e - a_al);

  // ---- phase 1: stream colind/values (16 B per lane), gather x, stage products
  int32_t c[ITERS][4];
  T v[ITERS][4];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int q = (it * 256 + tid) * 4;
    const O p = a_al + (O) q;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      c[it][j] = 0;
      v[it][j] = T(0);
    }
    if (q < total) {
      if (VEC && (int64_t) p + 4 <= nnz) {
        i32x4 cc = stream_load(reinterpret_cast<const i32x4*>(colind + p));
        c[it][0] = cc.x; c[it][1] = cc.y; c[it][2] = cc.z; c[it][3] = cc.w;
        load4_stream<T>(values + p, v[it]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if ((int64_t) p + j < nnz) {
            c[it][j] = stream_load(colind + p + j);
            v[it][j] = stream_load(values + p + j);
          }
      }
    }
  }
  T xv[ITERS][4];
#pragma unroll
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      xv[it][j] = x[c[it][j]];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int q = (it * 256 + tid) * 4;
    if (q < total) {
      const O p = a_al + (O) q;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool in = (p + j >= a) && (p + j < e);
        prod[q + j] = in ? v[it][j] * xv[it][j] : T(0);
      }
    }
  }
  __syncthreads();

  // ---- phase 2: a group of `lpr` lanes reduces each owned row out of LDS
  const int nrows = r_end - r_begin;
  int lpr = 1;
  while (lpr < 64 && nrows * lpr * 2 <= 256)
    lpr <<= 1;
  const int grp = tid / lpr, lig = tid % lpr, ngrp = 256 / lpr;
  for (int r = r_begin + grp; r < r_end; r += ngrp) {
    const int s0 = (int) (rowptr[r] - a_al), s1 = (int) (rowptr[r + 1] - a_al);
    T s = 0;
    for (int q = s0 + lig; q < s1; q += lpr)
      s += prod[q];
    s = group_sum(s, lpr);
    if (lig == 0)
      y[r] = beta == T(0) ? alpha * s : alpha * s + beta * y[r];
  }
}

// One wavefront per long row: y[r] = alpha * (tail + heads) + beta * y[r].
template <typename T, typename O>
__global__ __launch_bounds__(64) void spmv_long_fixup_kernel(int64_t n_long, int win,
                                                             const int32_t* __restrict__ long_rows,
                                                             const O* __restrict__ rowptr,
                                                             const T* __restrict__ part_head,
                                                             const T* __restrict__ part_tail,
                                                             T* __restrict__ y, T alpha, T beta) {
  const int64_t i = blockIdx.x;
  if (i >= n_long)
    return;
  const int r = long_rows[i];
  const int64_t p0 = (int64_t) rowptr[r], p1 = (int64_t) rowptr[r + 1];
  const int64_t w0 = p0 / win, w1 = (p1 - 1) / win;
  T s = 0;
  for (int64_t w = w0 + 1 + threadIdx.x; w <= w1; w += 64)
    s += part_head[w];
  s = group_sum_c<64>(s);
  if (threadIdx.x == 0) {
    s += part_tail[w0];
    y[r] = beta == T(0) ? alpha * s : alpha * s + beta * y[r];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void scale_vector_kernel(int64_t n, T* __restrict__ y, T beta) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i < n)
    y[i] = beta == T(0) ? T(0) : beta * y[i];
}

// op = T: y (n entries) += alpha * A^T x.  One 8-lane group per row of A.
template <typename T, typename O>
__global__ __launch_bounds__(256) void spmv_transpose_kernel(int64_t m, const O* __restrict__ rowptr,
                                                             const int32_t* __restrict__ colind,
                                                             const T* __restrict__ values,
                                                             const T* __restrict__ x, T* __restrict__ y,
                                                             T alpha) {
  const int64_t row = (int64_t) blockIdx.x * 32 + threadIdx.x / 8;
  const int lane = threadIdx.x % 8;
  if (row >= m)
    return;
  const T xi = alpha * x[row];
  const O p0 = rowptr[row], p1 = rowptr[row + 1];
  for (O p = p0 + lane; p < p1; p += 8)
    unsafeAtomicAdd(y + stream_load(colind + p), stream_load(values + p) * xi);
}

// ---- op = T without a plan, large matrices: two passes through a workspace instead of 1e8 float atomics --------------------
// Round 6.  The scatter kernel above issues one global float atomic per entry: 21 G / s chip-wide whatever the kernel does
// (profiles/r03_*: L2 atomics), 4.75 ms at cfg2's size = 2.4 % of the HBM roofline (bench.py --workload csc_spmv).  y = A^T x
// needs no gather -- x[i] is one scalar per ROW of A, streamed -- only its accumulation is scattered.  So: (1) t2_hist: the
// entries of every tile of 8 192 consecutive entries per column slice (S slices of W columns, W sized for an LDS-resident slice
// of y), scanned into the place of every (slice, tile) piece; (2) t2_plan: a list of work items -- a slice whose share exceeds
// SEG entries is cut into segments (hot columns); (3) t2_scatter: a tile per workgroup, the row of an entry from marks + a
// running maximum over the tile, the product alpha * a * x[i] and its 16-bit local column staged in LDS by slice and written
// to the slices' runs (the pieces of a tile leave as runs, those of neighbouring tiles as whole lines); (4) t2_accumulate: a
// work item sums its segment into an LDS copy of the slice (LDS float add by compare-and-swap: integer LDS atomics run 12 x
// faster than ds_add_f32) and writes y = beta y + sum -- or, for the segments of a cut slice, adds its partial sums to a y
// that (2) has scaled already.  The products of a slice lie in row order, but 1 024 lanes add them to the LDS copy as they
// come: like the scatter kernel's, the bits of y can differ from run to run (reproducibility).
static constexpr int T2_TILE = 8192, T2_THREADS = 1024, T2_SMAX = 1024;
// Where a tile's piece of a slice's run goes is COUNTED, not reserved: t2_hist writes the tile's entries per slice to
// counts[slice][tile], one exclusive scan over that matrix (scan.hpp, the transpose's) gives every piece its place -- the
// pieces of a slice in tile order, those of neighbouring tiles adjacent in memory.  (Reserved by atomics -- 511 per tile, 6.2 M
// at cfg2's size, on cursors padded to one per 128-byte line and split 16 ways per slice because atomics on one line, let alone
// one address, are served one after another -- the scatter kernel spent a third of its 0.39 ms on them: the chip takes about
// 21 G device-scope atomics per second whatever else a kernel does.)
// Workgroup -> tile: every XCD takes a contiguous range of tiles (workgroups are dealt round-robin to the 8 XCDs), so that the
// adjacent pieces of neighbouring tiles meet in ONE L2 and leave it as whole lines, and the counters of neighbouring tiles --
// neighbours in counts[slice][tile] -- are written through one L2 as well (transpose.hip: 4.0 -> 3.3 ms from this mapping).
__device__ __forceinline__ int64_t t2_tile_of(int64_t b, int64_t ntile) {
  const int64_t per = ntile / 8, body = per * 8;
  return b < body ? (b & 7) * per + (b >> 3) : b;
}

// slice = col / W without a division: rec = floor(2^32 / W) under-estimates by at most one
__device__ __forceinline__ int t2_slice(int col, int W, unsigned rec) {
  const unsigned q = __umulhi((unsigned) col, rec);
  return (int) (q + (unsigned) ((unsigned) col - q * (unsigned) W >= (unsigned) W));
}

// four consecutive elements in one access at the element's own alignment (global memory takes unaligned vector accesses; the
// address unit spends its cycles per INSTRUCTION, however wide)
typedef int t2_i4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float t2_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef double t2_d4u __attribute__((ext_vector_type(4), aligned(8)));
typedef unsigned short t2_h4u __attribute__((ext_vector_type(4), aligned(2)));
template <typename T> struct t2_vec4;
template <> struct t2_vec4<float> { typedef t2_f4u type; };
template <> struct t2_vec4<double> { typedef t2_d4u type; };

// wavefront scans by DPP moves (row_shr 1 / 2 / 4 / 8, row_bcast:15 / :31; a lane without a source reads 0): six vector
// instructions, no LDS round trip (the shuffle form is six dependent ds_bpermute)
template <int CTRL, int ROWS>
__device__ __forceinline__ int t2_dpp(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWS, 0xf, true);
}
__device__ __forceinline__ int t2_wave_incl_max(int v) {  // v >= 0
  int t;
  t = t2_dpp<0x111, 0xf>(v), v = v > t ? v : t;
  t = t2_dpp<0x112, 0xf>(v), v = v > t ? v : t;
  t = t2_dpp<0x114, 0xf>(v), v = v > t ? v : t;
  t = t2_dpp<0x118, 0xf>(v), v = v > t ? v : t;
  t = t2_dpp<0x142, 0xa>(v), v = v > t ? v : t;
  t = t2_dpp<0x143, 0xc>(v), v = v > t ? v : t;
  return v;
}
__device__ __forceinline__ unsigned t2_wave_incl_sum(unsigned v) {
  v += (unsigned) t2_dpp<0x111, 0xf>((int) v);
  v += (unsigned) t2_dpp<0x112, 0xf>((int) v);
  v += (unsigned) t2_dpp<0x114, 0xf>((int) v);
  v += (unsigned) t2_dpp<0x118, 0xf>((int) v);
  v += (unsigned) t2_dpp<0x142, 0xa>((int) v);
  v += (unsigned) t2_dpp<0x143, 0xc>((int) v);
  return v;
}

__global__ __launch_bounds__(256) void t2_hist_kernel(int64_t nnz, const int32_t* __restrict__ colind, int W, unsigned rec, int S,
                                                      int64_t ntile, int32_t* __restrict__ counts) {
  // one tile per workgroup: eight 16-byte loads per lane, all in flight together (one 4-byte load per lane and round: 139 us
  // at cfg2's size)
  __shared__ unsigned hist[T2_SMAX];
  for (int i = threadIdx.x; i < S; i += 256)
    hist[i] = 0;
  __syncthreads();
  const int64_t t = t2_tile_of(blockIdx.x, ntile);
  const int64_t e0 = t * T2_TILE, e1 = (e0 + T2_TILE) < nnz ? (e0 + T2_TILE) : nnz;
  if (e1 - e0 == T2_TILE) {
    constexpr int ROUNDS = T2_TILE / (256 * 4);
    t2_i4u c[ROUNDS];
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k)
      c[k] = __builtin_nontemporal_load(reinterpret_cast<const t2_i4u*>(colind + e0 + (int64_t) (k * 256 + threadIdx.x) * 4));
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
      atomicAdd(&hist[t2_slice(c[k].x, W, rec)], 1u);
      atomicAdd(&hist[t2_slice(c[k].y, W, rec)], 1u);
      atomicAdd(&hist[t2_slice(c[k].z, W, rec)], 1u);
      atomicAdd(&hist[t2_slice(c[k].w, W, rec)], 1u);
    }
  } else {
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 256)
      atomicAdd(&hist[t2_slice(stream_load(colind + e), W, rec)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < S; i += 256)
    counts[(int64_t) i * ntile + t] = (int32_t) hist[i];
}

// one workgroup: work items (slice, first, last, cut) with segments of at most seg entries -- a slice's run is
// offsets[slice][0] .. offsets[slice + 1][0] of the scanned counts --; the slices that are cut get their part of y scaled by beta
// here (their segments only ADD)
template <typename T>
__global__ __launch_bounds__(1024) void t2_plan_kernel(int S, int W, int64_t n, int64_t ntile, unsigned seg,
                                                       const int32_t* __restrict__ offsets, int4* __restrict__ items,
                                                       int* __restrict__ n_items, T* __restrict__ y, T beta) {
  // one slice per thread (S <= 1024): exclusive scan of the item counts by DPP scans inside the wavefronts + the 16 wave totals
  // through LDS (a Hillis-Steele scan over the workgroup -- twenty barriers -- and a serial walk over the slices for the cut
  // ones were 35 of this kernel's first 49 us)
  __shared__ unsigned s_wb[16];
  __shared__ int s_ncut, s_cut[T2_SMAX];
  const int i = threadIdx.x, wv = i >> 6, lane = i & 63;
  const unsigned off = i < S ? (unsigned) offsets[(int64_t) i * ntile] : 0u;
  const unsigned end = i < S ? (unsigned) offsets[(int64_t) (i + 1) * ntile] : 0u;  // (offsets[S * ntile] = the total)
  const unsigned c = end - off;
  const unsigned t = i < S ? (c == 0 ? 1u : (c + seg - 1) / seg) : 0u;  // (an empty slice still owns its part of y)
  const unsigned b = t2_wave_incl_sum(t);
  if (lane == 63)
    s_wb[wv] = b;
  if (i == 0)
    s_ncut = 0;
  __syncthreads();
  unsigned bb = 0;
  for (int w = 0; w < wv; ++w)
    bb += s_wb[w];
  const unsigned it0 = bb + b - t;
  if (i < S) {
    if (t > 1)
      s_cut[atomicAdd(&s_ncut, 1)] = i;
    for (unsigned q = 0; q < t; ++q) {
      const unsigned lo = off + q * seg, hi = (lo + seg) < end ? (lo + seg) : end;
      items[it0 + q] = make_int4(i, (int) lo, (int) hi, t > 1 ? 1 : 0);
    }
  }
  if (i == 1023)  // (the threads behind the last slice add nothing: the last thread's inclusive sum is the total)
    *n_items = (int) (bb + b);
  __syncthreads();
  // the cut slices' part of y: scaled now, added to by every segment later
  const int ncut = s_ncut;
  for (int q = 0; q < ncut; ++q) {
    const int sc = s_cut[q];
    const int64_t c0 = (int64_t) sc * W, c1 = (c0 + W) < n ? (c0 + W) : n;
    for (int64_t cc = c0 + threadIdx.x; cc < c1; cc += 1024)
      y[cc] = beta == T(0) ? T(0) : beta * y[cc];
  }
}

// LDS of the scatter kernel: products + packed (local column, slice) words of a tile, three tables of Sa words
static inline size_t t2_scatter_lds(size_t value_size, int S) {
  const size_t Sa = ((size_t) S + 4) & ~(size_t) 3;
  return (size_t) T2_TILE * (value_size + 4) + 3 * Sa * 4;
}

template <typename T, typename O>
__global__ __launch_bounds__(T2_THREADS, sizeof(T) == 4 ? 8 : 4) void t2_scatter_kernel(int64_t m, int64_t nnz, const O* __restrict__ rowptr,
                                                                const int32_t* __restrict__ colind,
                                                                const T* __restrict__ values, const T* __restrict__ x,
                                                                T alpha, int W, unsigned rec, int S, const int32_t* __restrict__ tile_row,
                                                                int64_t ntile, const int32_t* __restrict__ offsets,
                                                                T* __restrict__ prod, uint16_t* __restrict__ lcol) {
  // The tile's products leave SORTED BY SLICE through LDS: 64 lanes storing 4 + 2 bytes each at 64 unrelated addresses were
  // two address-unit passes of 64 separate accesses per wave-instruction (first version: 1.7 ms at cfg2's size, the whole
  // kernel); staged, consecutive lanes write consecutive elements of a run.
  // Second version (0.72 ms): the kernel was bound by its LDS instructions -- a binary search per entry for its row (11 reads)
  // and one per staged position for its slice (10 reads) next to the two atomics and three stores an entry needs.  Now the row
  // of an entry comes from marks + a running maximum (the rows that start inside the tile leave their number at their first
  // entry; DPP scans inside the wavefronts, one carry per 64 positions -- the scheme of transpose.hip's first pass), and a
  // staged entry carries its slice next to its local column (one 32-bit word), so the write-out reads three words per entry.
  extern __shared__ __attribute__((aligned(16))) unsigned char t2s_smem[];
  typedef __attribute__((address_space(3))) int lds_int;
  const int Sa = (S + 4) & ~3;
  T* s_val = reinterpret_cast<T*>(t2s_smem);                         // [TILE] products; before that: the marks of the row search
  unsigned* s_pack = reinterpret_cast<unsigned*>(s_val + T2_TILE);   // [TILE] local column | slice << 16
  unsigned* s_cnt = s_pack + T2_TILE;                                // [Sa] entries of the tile per slice, then the fill counters
  unsigned* s_loff = s_cnt + Sa;                                     // [Sa] exclusive scan of the counts (positions in the stage)
  unsigned* s_delta = s_loff + Sa;                                   // [Sa] global position - staged position of a slice's piece
  __shared__ unsigned s_scan[T2_THREADS / 64];
  constexpr int PER = T2_TILE / T2_THREADS, NSEG = T2_TILE / 64;
  __shared__ int s_seg[NSEG];                                        // running maximum at the end of every 64 positions
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int64_t tile = t2_tile_of(blockIdx.x, ntile);
  const int64_t e0 = tile * T2_TILE, e1 = (e0 + T2_TILE) < nnz ? (e0 + T2_TILE) : nnz;
  const int count = (int) (e1 - e0);
  // where this tile's piece of slice `tid` goes (needed behind the third barrier)
  const unsigned gb = tid < S ? (unsigned) offsets[(int64_t) tid * ntile + tile] : 0u;
  // rows that have entries in [e0, e1): tile_row[w] = first row r with rowptr[r] >= w * TILE; the row before it may reach in
  int64_t r_lo = tile_row[tile], r_hi = tile_row[tile + 1];
  r_lo = r_lo > 0 ? r_lo - 1 : 0;
  r_hi = r_hi < m ? r_hi : m;  // (r_hi itself starts at or beyond e1)
  const int64_t nrows = r_hi - r_lo;
  int col[PER];
  T val[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int64_t e = e0 + tid + (int64_t) k * T2_THREADS;
    col[k] = e < e1 ? stream_load(colind + e) : -1;
    val[k] = e < e1 ? stream_load(values + e) : T(0);
  }
  lds_int* mark = (lds_int*) (int*) t2s_smem;
  for (int i = tid; i < S; i += T2_THREADS)
    s_cnt[i] = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k)
    mark[k * T2_THREADS + tid] = 0;
  __syncthreads();
  // row r_lo + j (j >= 1) that starts inside the tile: j at its first entry; of several rows starting at one position -- empty
  // ones -- the last owns the entry (LDS max)
  for (int64_t j = 1 + tid; j <= nrows; j += T2_THREADS) {
    const int64_t pos = (int64_t) rowptr[r_lo + j] - e0;
    if (pos >= 0 && pos < count)
      __hip_atomic_fetch_max(&mark[pos], (int) j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  int sl[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    sl[k] = col[k] >= 0 ? t2_slice(col[k], W, rec) : 0;
    if (col[k] >= 0)
      atomicAdd(&s_cnt[sl[k]], 1u);
  }
  __syncthreads();
  int rrow[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k)
    rrow[k] = mark[k * T2_THREADS + tid];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    rrow[k] = t2_wave_incl_max(rrow[k]);
    if (lane == 63)
      s_seg[k * (T2_THREADS / 64) + wv] = rrow[k];  // positions k * THREADS + wv * 64 .. + 63: segment k * 16 + wv
  }
  // exclusive scan of s_cnt[0 .. S) -> s_loff: one counter per thread (S <= THREADS), scans inside the wavefronts, the 16
  // wave totals through LDS
  const unsigned c_mine = tid < S ? s_cnt[tid] : 0u;
  const unsigned a_mine = t2_wave_incl_sum(c_mine);
  if (lane == 63)
    s_scan[wv] = a_mine;
  __syncthreads();
  if (tid < S) {
    unsigned before = 0;
    for (int w = 0; w < wv; ++w)
      before += s_scan[w];
    const unsigned loff = before + a_mine - c_mine;
    s_loff[tid] = loff;
    s_delta[tid] = gb - loff;
    s_cnt[tid] = 0;  // from here on: the next free place inside the slice's piece of the stage
  }
  if (wv == T2_THREADS / 64 - 1) {
    // (the last wavefront has no counters for S <= 960) exclusive running maximum over the NSEG segment ends, two per lane
    static_assert(NSEG == 128, "two segment ends per lane");
    const int a = s_seg[2 * lane], b = s_seg[2 * lane + 1];
    const int ab = a > b ? a : b;
    const int inc = t2_wave_incl_max(ab);
    int exc = __shfl_up(inc, 1, 64);  // the inclusive maximum of the lanes before this one
    exc = lane == 0 ? 0 : exc;
    __builtin_amdgcn_wave_barrier();
    s_seg[2 * lane] = exc;
    s_seg[2 * lane + 1] = exc > a ? exc : a;
  }
  __syncthreads();
  T xv[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int pre = s_seg[k * (T2_THREADS / 64) + wv];
    rrow[k] = rrow[k] > pre ? rrow[k] : pre;
#ifdef T2_EXP_NOSEARCH
    rrow[k] = 0;
#endif
    xv[k] = x[r_lo + rrow[k]];
  }
  // (places first, products second: the gathers of x stay in flight behind the LDS atomics)
  unsigned at[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k)
    at[k] = col[k] >= 0 ? s_loff[sl[k]] + atomicAdd(&s_cnt[sl[k]], 1u) : 0u;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    if (col[k] < 0)
      continue;
    s_pack[at[k]] = (unsigned) (col[k] - sl[k] * W) | ((unsigned) sl[k] << 16);
    s_val[at[k]] = alpha * val[k] * xv[k];
  }
  __syncthreads();
  for (int q = tid; q < count; q += T2_THREADS) {
    const unsigned pk = s_pack[q];
    const unsigned g = (unsigned) q + s_delta[pk >> 16];
#ifndef T2_EXP_NOWRITE
    prod[g] = s_val[q];
    lcol[g] = (uint16_t) pk;
#else
    if (g == 0xffffffffu)
      prod[0] = s_val[q];
#endif
  }
}

template <typename T>
__device__ __forceinline__ void t2_lds_add(T* addr, T v);
template <>
__device__ __forceinline__ void t2_lds_add<float>(float* addr, float v) {
  int* ai = reinterpret_cast<int*>(addr);
  int old = *(volatile __attribute__((address_space(3))) int*) ai;
  while (true) {
    const int assumed = old;
    old = atomicCAS(ai, assumed, __float_as_int(__int_as_float(assumed) + v));
    if (old == assumed)
      break;
  }
}
template <>
__device__ __forceinline__ void t2_lds_add<double>(double* addr, double v) {
  unsigned long long* ai = reinterpret_cast<unsigned long long*>(addr);
  unsigned long long old = *(volatile __attribute__((address_space(3))) unsigned long long*) ai;
  while (true) {
    const unsigned long long assumed = old;
    old = atomicCAS(ai, assumed, (unsigned long long) __double_as_longlong(__longlong_as_double((long long) assumed) + v));
    if (old == assumed)
      break;
  }
}

template <typename T>
__global__ __launch_bounds__(1024) void t2_accumulate_kernel(int W, int64_t n, const int4* __restrict__ items,
                                                             const int* __restrict__ n_items, const T* __restrict__ prod,
                                                             const uint16_t* __restrict__ lcol, T* __restrict__ y, T beta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char t2_smem[];
  T* acc = reinterpret_cast<T*>(t2_smem);  // [W]
  typedef typename t2_vec4<T>::type vec4;
  const int count = *n_items;
  for (int it = blockIdx.x; it < count; it += gridDim.x) {
    const int4 w = items[it];
    for (int i = threadIdx.x; i < W; i += 1024)
      acc[i] = T(0);
    __syncthreads();
    // four consecutive entries per lane and access, two accesses in flight (one 4 + 2 byte pair per lane and round: 267 us at
    // cfg2's size, 2.2 TB/s)
    const int64_t lo = (int64_t) (unsigned) w.y, hi = (int64_t) (unsigned) w.z;
    int64_t e = lo + (int64_t) threadIdx.x * 4;
    for (; e + 4096 + 4 <= hi; e += 8192) {
      const vec4 p0 = __builtin_nontemporal_load(reinterpret_cast<const vec4*>(prod + e));
      const t2_h4u c0 = __builtin_nontemporal_load(reinterpret_cast<const t2_h4u*>(lcol + e));
      const vec4 p1 = __builtin_nontemporal_load(reinterpret_cast<const vec4*>(prod + e + 4096));
      const t2_h4u c1 = __builtin_nontemporal_load(reinterpret_cast<const t2_h4u*>(lcol + e + 4096));
      t2_lds_add<T>(acc + c0.x, p0.x);
      t2_lds_add<T>(acc + c0.y, p0.y);
      t2_lds_add<T>(acc + c0.z, p0.z);
      t2_lds_add<T>(acc + c0.w, p0.w);
      t2_lds_add<T>(acc + c1.x, p1.x);
      t2_lds_add<T>(acc + c1.y, p1.y);
      t2_lds_add<T>(acc + c1.z, p1.z);
      t2_lds_add<T>(acc + c1.w, p1.w);
    }
    for (; e < hi; e += 4096)
      for (int j = 0; j < 4; ++j)
        if (e + j < hi)
          t2_lds_add<T>(acc + stream_load(lcol + e + j), stream_load(prod + e + j));
    __syncthreads();
    const int64_t c0 = (int64_t) w.x * W, c1 = (c0 + W) < n ? (c0 + W) : n;
    if (w.w) {  // a segment of a cut slice: y was scaled by t2_plan_kernel, the segments add
      for (int64_t c = c0 + threadIdx.x; c < c1; c += 1024)
        if (acc[c - c0] != T(0))
          unsafeAtomicAdd(y + c, acc[c - c0]);
    } else {
      for (int64_t c = c0 + threadIdx.x; c < c1; c += 1024)
        y[c] = beta == T(0) ? acc[c - c0] : acc[c - c0] + beta * y[c];
    }
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------
// inspect kernels
// ---------------------------------------------------------------------------
// win_row[w] = first row r with rowptr[r] >= w*win  (w = 0..nwin-1); win_row[nwin] = m.
template <typename O>
__global__ __launch_bounds__(256) void plan_window_rows_kernel(int64_t m, int64_t nwin, int win,
                                                               const O* __restrict__ rowptr,
                                                               int32_t* __restrict__ win_row) {
  const int64_t w = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (w > nwin)
    return;
  if (w == nwin) {
    win_row[w] = (int32_t) m;
    return;
  }
  const int64_t target = w * win;
  int64_t lo = 0, hi = m;  // answer in [0, m]
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t) rowptr[mid] < target)
      lo = mid + 1;
    else
      hi = mid;
  }
  win_row[w] = (int32_t) lo;
}

// stats[0] = max row length, stats[1] = #long rows, stats[2] = #empty rows, stats[3] = entries in long rows.
// Long rows are appended to long_rows (capacity nnz/win + 1 always suffices).
template <typename O>
__global__ __launch_bounds__(256) void plan_row_stats_kernel(int64_t m, int win,
                                                             const O* __restrict__ rowptr,
                                                             unsigned long long* __restrict__ stats,
                                                             int32_t* __restrict__ long_rows) {
  // grid-stride: a few thousand workgroups, one set of global atomics per workgroup (one per wave
  // cost 1.8 ms of same-address contention at 10M rows)
  __shared__ unsigned long long s_max[4], s_empty[4];
  unsigned long long mx = 0, nempty = 0;
  for (int64_t r = (int64_t) blockIdx.x * 256 + threadIdx.x; r < m; r += (int64_t) gridDim.x * 256) {
    const unsigned long long len = (unsigned long long) (rowptr[r + 1] - rowptr[r]);
    nempty += len == 0;
    mx = len > mx ? len : mx;
    if (len > (unsigned long long) win) {
      unsigned long long slot = atomicAdd(&stats[1], 1ull);
      long_rows[slot] = (int32_t) r;
      atomicAdd(&stats[3], len);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long other = __shfl_xor(mx, o, SPB_WAVE);
    mx = other > mx ? other : mx;
    nempty += __shfl_xor(nempty, o, SPB_WAVE);
  }
  if ((threadIdx.x & 63) == 0) {
    s_max[threadIdx.x >> 6] = mx;
    s_empty[threadIdx.x >> 6] = nempty;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      mx = s_max[w] > mx ? s_max[w] : mx;
      nempty += s_empty[w];
    }
    if (mx > 0)
      atomicMax(&stats[0], mx);
    if (nempty)
      atomicAdd(&stats[2], nempty);
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename T, typename O, int LPR>
static void launch_vector(hipStream_t s, int64_t m, const O* rowptr, const int32_t* colind,
                          const T* values, const T* x, T* y, T alpha, T beta) {
  constexpr int ROWS = 256 / LPR;
  const int64_t grid = cdiv(m, ROWS);
  hipLaunchKernelGGL((spmv_vector_kernel<T, O, LPR>), dim3((unsigned) grid), dim3(256), 0, s, m, rowptr,
                     colind, values, x, y, alpha, beta);
}

static int pick_lpr(int64_t m, int64_t nnz) {
  const double avg = m > 0 ? (double) nnz / (double) m : 0.0;
  int lpr = 2;
  while (lpr < 64 && (double) lpr * 1.5 < avg)
    lpr <<= 1;
  return lpr;
}

template <typename T, typename O>
static int run_vector(hipStream_t s, int lpr, int64_t m, const O* rowptr, const int32_t* colind,
                      const T* values, const T* x, T* y, T alpha, T beta) {
  switch (lpr) {
  case 2: launch_vector<T, O, 2>(s, m, rowptr, colind, values, x, y, alpha, beta); break;
  case 4: launch_vector<T, O, 4>(s, m, rowptr, colind, values, x, y, alpha, beta); break;
  case 8: launch_vector<T, O, 8>(s, m, rowptr, colind, values, x, y, alpha, beta); break;
  case 16: launch_vector<T, O, 16>(s, m, rowptr, colind, values, x, y, alpha, beta); break;
  case 32: launch_vector<T, O, 32>(s, m, rowptr, colind, values, x, y, alpha, beta); break;
  default: launch_vector<T, O, 64>(s, m, rowptr, colind, values, x, y, alpha, beta); break;
  }
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

template <typename T>
struct window_of {
  // f32: 2*2048*4 B = 16 KiB LDS; f64: 2*1024*8 B = 16 KiB LDS -> 8+ workgroups/CU.
  static constexpr int value = sizeof(T) == 4 ? 2048 : 1024;
};

template <typename T, typename O, bool HAS_LONG>
static void launch_rowblock(hipStream_t s, const spblas_gfx950_plan_s* pl, const O* rowptr,
                            const int32_t* colind, const T* values, const T* x, T* y, T alpha, T beta) {
  constexpr int WIN = window_of<T>::value;
  const bool vec = (((uintptr_t) colind | (uintptr_t) values) & 15) == 0;
  T* ph = static_cast<T*>(pl->part_head);
  T* pt = static_cast<T*>(pl->part_tail);
  if (vec)
    hipLaunchKernelGGL((spmv_rowblock_kernel<T, O, WIN, HAS_LONG, true>), dim3((unsigned) pl->nwin),
                       dim3(256), 0, s, pl->nnz, rowptr, colind, values, x, y, alpha, beta, pl->win_row,
                       ph, pt);
  else
    hipLaunchKernelGGL((spmv_rowblock_kernel<T, O, WIN, HAS_LONG, false>), dim3((unsigned) pl->nwin),
                       dim3(256), 0, s, pl->nnz, rowptr, colind, values, x, y, alpha, beta, pl->win_row,
                       ph, pt);
}

int spmv_sliced_exec(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* alpha, const void* x,
                     const void* beta, void* y);
int spmv_sliced_expand(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* x);
int spmv_sliced_reduce_rows(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* alpha, const void* beta,
                            void* y, int64_t row_begin, int64_t row_end, void* const* peers, int n_peers,
                            int64_t peer_off);
int spmv_sliced_full_ksplit(spblas_gfx950_plan_s* pl);
int spmv_sliced_reserve_partial(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, int K);
int spmv_sliced_build(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values, bool auto_mode);
int spmv_sliced_update(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values);
void spmv_sliced_free(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl);
void spmm_plan_free(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl);
int spmv_hot_rows(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* alpha, const void* x, void* y);

static int env_int_spmv(const char* name, int def);

template <typename T, typename O>
static int spmv_typed(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, int op, int64_t m, int64_t n,
                      int64_t nnz, const void* alpha_p, const void* rowptr_p, const int32_t* colind,
                      const void* values_p, const void* x_p, const void* beta_p, void* y_p) {
  const T alpha = *static_cast<const T*>(alpha_p);
  const T beta = *static_cast<const T*>(beta_p);
  const O* rowptr = static_cast<const O*>(rowptr_p);
  const T* values = static_cast<const T*>(values_p);
  const T* x = static_cast<const T*>(x_p);
  T* y = static_cast<T*>(y_p);
  hipStream_t s = h->stream;

  if (op == SPBLAS_GFX950_OP_T) {
    // Large matrices: two passes through a workspace (t2_* kernels above) instead of one float atomic per entry.  Needs the
    // handle's scratch (6 or 10 B per entry + tables), hence not inside stream captures; SPBLAS_GFX950_SPMV_T2=0 keeps the
    // scatter kernel, =1 forces the two-pass form for any size (tests).
    const int t2_env = env_int_spmv("SPBLAS_GFX950_SPMV_T2", -1);
    const int cus = h->num_cus > 0 ? h->num_cus : 256;
    // slice width: as many slices as the accumulate pass has workgroup slots (two per CU: ONE round of work items -- 611 slices
    // of 16 384 columns on 512 slots were two rounds, 0.236 against 0.170 ms), a multiple of 64, an LDS-resident slice of y of
    // at most 76 KiB (two workgroups per CU), 16-bit local columns, at most T2_SMAX slices
    int64_t Wsl = (cdiv(n, (int64_t) 2 * cus) + 63) & ~(int64_t) 63;
    const int64_t Wcap = (76 * 1024) / (int64_t) sizeof(T);
    Wsl = Wsl < 4096 ? 4096 : Wsl > Wcap ? Wcap : Wsl;
    if (cdiv(n, Wsl) > T2_SMAX)  // (a matrix with more columns than 1 024 such slices: wider slices, one workgroup per CU)
      Wsl = std::min<int64_t>((cdiv(n, (int64_t) T2_SMAX) + 63) & ~(int64_t) 63, std::min<int64_t>(65536, (152 * 1024) / (int64_t) sizeof(T)));
    if (t2_env == 1) {  // tests / tools/fuzz_spmv_t.py: any slice width (a multiple of 64 that the LDS holds), any segment length
      const int w_env = env_int_spmv("SPBLAS_GFX950_SPMV_T2_W", 0);
      if (w_env >= 64 && w_env <= Wcap)
        Wsl = w_env & ~63;
    }
    const unsigned wrec = (unsigned) (((uint64_t) 1 << 32) / (uint64_t) Wsl);
    const int64_t Ssl = cdiv(n, Wsl), ntile = cdiv(nnz, T2_TILE);
    const bool t2_fits = m > 0 && nnz > 0 && nnz < INT32_MAX - T2_TILE && Ssl <= T2_SMAX && ntile < INT32_MAX &&
                         Ssl * ntile < INT32_MAX;
    const bool t2_want = t2_env == 1 || (t2_env != 0 && nnz >= ((int64_t) 4 << 20) && n >= 65536);
    if (t2_fits && t2_want) {
      const int64_t nscan = Ssl * ntile, nblk = cdiv(nscan, 2048);  // counts[slice][tile] + the total; the scan's block sums
      const size_t off_cnt = 0, off_part = (off_cnt + (size_t) (nscan + 1) * 4 + 255) & ~(size_t) 255,
                   off_nit = (off_part + (size_t) (nblk + 2) * sizeof(long long) + 15) & ~(size_t) 15,
                   off_tile = (off_nit + 4 + 15) & ~(size_t) 15;
      unsigned seg = (unsigned) std::max<int64_t>(65536, 2 * cdiv(nnz, Ssl));
      if (t2_env == 1 && env_int_spmv("SPBLAS_GFX950_SPMV_T2_SEG", 0) >= 64)
        seg = (unsigned) env_int_spmv("SPBLAS_GFX950_SPMV_T2_SEG", 0);
      const size_t max_items = (size_t) Ssl + (size_t) (nnz / seg) + 1;
      const size_t off_items = (off_tile + (size_t) (ntile + 1) * 4 + 15) & ~(size_t) 15;
      const size_t off_prod = (off_items + max_items * sizeof(int4) + 255) & ~(size_t) 255;
      const size_t off_lcol = (off_prod + (size_t) nnz * sizeof(T) + 255) & ~(size_t) 255;
      const size_t bytes = off_lcol + (size_t) nnz * 2 + 256;
      void* ws = nullptr;
      // (never inside a stream capture: the recorded launches would keep pointers into the handle's scratch, which a later call
      // may grow, i.e. free and allocate again -- a captured un-inspected transposed multiply keeps the scatter kernel)
      if (!stream_capturing(s) && handle_scratch(h, bytes, &ws) == SPBLAS_GFX950_STATUS_SUCCESS) {
        char* w8 = static_cast<char*>(ws);
        int32_t* counts = reinterpret_cast<int32_t*>(w8 + off_cnt);
        long long* partials = reinterpret_cast<long long*>(w8 + off_part);
        int* n_items = reinterpret_cast<int*>(w8 + off_nit);
        int32_t* tile_row = reinterpret_cast<int32_t*>(w8 + off_tile);
        int4* items = reinterpret_cast<int4*>(w8 + off_items);
        T* prod = reinterpret_cast<T*>(w8 + off_prod);
        uint16_t* lcol = reinterpret_cast<uint16_t*>(w8 + off_lcol);
        hipLaunchKernelGGL((plan_window_rows_kernel<O>), dim3((unsigned) cdiv(ntile + 1, 256)), dim3(256), 0, s, m, ntile, T2_TILE,
                           rowptr, tile_row);
        hipLaunchKernelGGL(t2_hist_kernel, dim3((unsigned) ntile), dim3(256), 0, s, nnz, colind, (int) Wsl, wrec, (int) Ssl, ntile,
                           counts);
        scan_counts_i32(s, nscan, counts, partials);
        hipLaunchKernelGGL((t2_plan_kernel<T>), dim3(1), dim3(1024), 0, s, (int) Ssl, (int) Wsl, n, ntile, seg, counts, items,
                           n_items, y, beta);
        static std::atomic<bool> t2_attr[64][2][2] = {};
        const int dv = h->device >= 0 && h->device < 64 ? h->device : 0;
        constexpr int ti = sizeof(T) == 4 ? 0 : 1, oi = sizeof(O) == 4 ? 0 : 1;
        const size_t lds_sc = t2_scatter_lds(sizeof(T), (int) Ssl);
        if (!t2_attr[dv][ti][oi].load(std::memory_order_acquire) || h->device >= 64) {
          SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(t2_accumulate_kernel<T>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
          SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(t2_scatter_kernel<T, O>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int) t2_scatter_lds(sizeof(T), T2_SMAX)));
          t2_attr[dv][ti][oi].store(true, std::memory_order_release);
        }
        hipLaunchKernelGGL((t2_scatter_kernel<T, O>), dim3((unsigned) ntile), dim3(T2_THREADS), lds_sc, s, m, nnz, rowptr, colind,
                           values, x, alpha, (int) Wsl, wrec, (int) Ssl, tile_row, ntile, counts, prod, lcol);
        hipLaunchKernelGGL((t2_accumulate_kernel<T>), dim3((unsigned) std::min<size_t>(max_items, (size_t) cus * 2)), dim3(1024),
                           (size_t) Wsl * sizeof(T), s, (int) Wsl, n, items, n_items, prod, lcol, y, beta);
        SPB_HIP(hipGetLastError());
        return SPBLAS_GFX950_STATUS_SUCCESS;
      }
      (void) hipGetLastError();
    }
    // y has n entries: scale, then scatter-add.
    if (n > 0)
      hipLaunchKernelGGL((scale_vector_kernel<T>), dim3((unsigned) cdiv(n, 256)), dim3(256), 0, s, n, y,
                         beta);
    if (m > 0 && nnz > 0)
      hipLaunchKernelGGL((spmv_transpose_kernel<T, O>), dim3((unsigned) cdiv(m, 32)), dim3(256), 0, s, m,
                         rowptr, colind, values, x, y, alpha);
    SPB_HIP(hipGetLastError());
    return SPBLAS_GFX950_STATUS_SUCCESS;
  }

  if (m == 0)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  if (nnz == 0) {
    hipLaunchKernelGGL((scale_vector_kernel<T>), dim3((unsigned) cdiv(m, 256)), dim3(256), 0, s, m, y, beta);
    SPB_HIP(hipGetLastError());
    return SPBLAS_GFX950_STATUS_SUCCESS;
  }

  if (pl && pl->alg == SPBLAS_GFX950_SPMV_SLICED)
    return spmv_sliced_exec(h, pl, alpha_p, x_p, beta_p, y_p);

  if (pl && pl->alg == SPBLAS_GFX950_SPMV_ROWBLOCK) {
    pl->last_stream = s;  // part_head / part_tail are the plan's
    pl->used = true;
    if (pl->n_long > 0) {
      launch_rowblock<T, O, true>(s, pl, rowptr, colind, values, x, y, alpha, beta);
      hipLaunchKernelGGL((spmv_long_fixup_kernel<T, O>), dim3((unsigned) pl->n_long), dim3(64), 0, s,
                         pl->n_long, pl->win, pl->long_rows, rowptr, static_cast<const T*>(pl->part_head),
                         static_cast<const T*>(pl->part_tail), y, alpha, beta);
    } else {
      launch_rowblock<T, O, false>(s, pl, rowptr, colind, values, x, y, alpha, beta);
    }
    SPB_HIP(hipGetLastError());
    return SPBLAS_GFX950_STATUS_SUCCESS;
  }

  const int lpr = pl ? pl->vector_lpr : pick_lpr(m, nnz);
  return run_vector<T, O>(s, lpr, m, rowptr, colind, values, x, y, alpha, beta);
}

template <typename O>
static int plan_build(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, int alg_req) {
  hipStream_t s = h->stream;
  const O* rowptr = static_cast<const O*>(pl->rowptr);
  const int64_t m = pl->m, nnz = pl->nnz;
  pl->win = pl->win_req > 0 ? pl->win_req
                            : pl->value_type == SPBLAS_GFX950_F32 ? window_of<float>::value : window_of<double>::value;
  pl->nwin = nnz / pl->win + 1;
  pl->vector_lpr = pick_lpr(m, nnz);
  const size_t tsz = pl->value_type == SPBLAS_GFX950_F32 ? 4 : 8;

  int rc;
  unsigned long long* d_stats = nullptr;
  const int64_t long_cap = nnz / pl->win + 1;
  if ((rc = dev_alloc((void**) &d_stats, 4 * sizeof(unsigned long long), s)))
    return rc;
  struct stats_guard {  // released on every exit path (plan-owned arrays are freed by plan_destroy)
    void* p;
    hipStream_t s;
    ~stats_guard() { dev_free(p, s); }
  } guard{d_stats, s};
  if ((rc = dev_alloc((void**) &pl->long_rows, (size_t) long_cap * 4, s)))
    return rc;
  if ((rc = dev_alloc((void**) &pl->win_row, (size_t) (pl->nwin + 1) * 4, s)))
    return rc;
  SPB_HIP(hipMemsetAsync(d_stats, 0, 4 * sizeof(unsigned long long), s));
  if (m > 0) {
    hipLaunchKernelGGL((plan_row_stats_kernel<O>), dim3((unsigned) (cdiv(m, 256) < 2048 ? cdiv(m, 256) : 2048)), dim3(256), 0, s, m,
                       pl->win, rowptr, d_stats, pl->long_rows);
  }
  hipLaunchKernelGGL((plan_window_rows_kernel<O>), dim3((unsigned) cdiv(pl->nwin + 1, 256)), dim3(256), 0,
                     s, m, pl->nwin, pl->win, rowptr, pl->win_row);
  SPB_HIP(hipGetLastError());
  unsigned long long stats[4];
  SPB_HIP(hipMemcpyAsync(stats, d_stats, sizeof(stats), hipMemcpyDeviceToHost, s));
  SPB_HIP(hipStreamSynchronize(s));
  pl->max_row_len = (int64_t) stats[0];
  pl->n_long = (int64_t) stats[1];
  pl->empty_rows = (int64_t) stats[2];
  pl->long_nnz = (int64_t) stats[3];
  pl->device_bytes = (size_t) long_cap * 4 + (size_t) (pl->nwin + 1) * 4;
  if (pl->n_long > 0) {
    if ((rc = dev_alloc(&pl->part_head, (size_t) pl->nwin * tsz, s)))
      return rc;
    if ((rc = dev_alloc(&pl->part_tail, (size_t) pl->nwin * tsz, s)))
      return rc;
    pl->device_bytes += 2 * (size_t) pl->nwin * tsz;
  }

  int alg = alg_req;
  if (alg == SPBLAS_GFX950_SPMV_AUTO) {
    // Row blocks need enough entries per window to amortise the block; matrices
    // that are mostly empty rows (nnz << m) are served by the plan-free kernel.
    alg = (nnz >= m / 2) ? SPBLAS_GFX950_SPMV_ROWBLOCK : SPBLAS_GFX950_SPMV_VECTOR;
  }
  pl->alg = alg;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

// for the builders of composite plans (spmv_hot.hip): the row statistics / window partition of one CSR matrix ...
int spmv_plan_structures(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, int alg_req) {
  return pl->offset_type == SPBLAS_GFX950_I32 ? plan_build<int32_t>(h, pl, alg_req) : plan_build<int64_t>(h, pl, alg_req);
}
// ... and the end of a plan they own: everything it holds, then the object
void spmv_plan_release(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl) {
  if (!pl)
    return;
  hipStream_t s = h->stream;
  dev_free(pl->win_row, s);
  dev_free(pl->long_rows, s);
  dev_free(pl->part_head, s);
  dev_free(pl->part_tail, s);
  spmv_sliced_free(h, pl);
  spmm_plan_free(h, pl);
  delete pl;
}

// AUTO considers the sliced re-tiling only when x approaches an XCD's 4 MiB L2 (measured
// crossover on square 10-per-row matrices: n between 0.5M and 1M columns, tools/auto_sweep.sh), the
// matrix is big enough to amortise two launches, rows are short (LDS atomics serialise on
// hub rows) and the average (slice, bin) segment keeps a wavefront busy.
static int env_int_spmv(const char* name, int def) {
  const char* v = std::getenv(name);
  return v && *v ? std::atoi(v) : def;
}

static bool sliced_candidate(const spblas_gfx950_plan_s* pl) {
  const size_t tsz = pl->value_type == SPBLAS_GFX950_F32 ? 4 : 8;
  const double tile = 80.0 * 1024 / tsz;
  const double nseg = (pl->n / tile + 1) * (pl->m / tile + 1);
  return (size_t) pl->n * tsz >= ((size_t) 3 << 20) && pl->nnz >= ((int64_t) 2 << 20) &&
         // (entries per 80 KiB x 80 KiB tile.  48 until round 5; the last of the eight nnz-prefix row shards of R-MAT scale 24 --
         // 7.3 M rows of 4.6 entries, 28.7 per tile -- was kept off the tiles by it and ran the row-block kernel at 0.463 ms
         // against 0.374 tiled: the build's own padding guard (padded stream <= 2 nnz) is the better judge)
         pl->nnz < INT32_MAX - 8 && (double) pl->nnz / nseg >= 24.0 &&
         // rows longer than the window: a few dense rows are fine; a matrix living in its long rows (power law) gets
         // variable-height bins and is decided by the timed trial (below) -- unless trials are switched off
         (pl->max_row_len <= 4096 || (pl->n_long <= 65536 && pl->long_nnz * 4 <= pl->nnz) ||
          env_int_spmv("SPBLAS_GFX950_AUTO_TRIAL", 1) != 0);
}

// AUTO trial: both plans exist; multiply a zero vector with each (a warm-up and one timed run) (the traffic does not depend on the values of x)
// and keep the faster.  *sliced_wins is left true when anything about the trial itself fails.
static int auto_trial(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values, bool* sliced_wins) {
  *sliced_wins = true;
  hipStream_t s = h->stream;
  const size_t tsz = pl->value_type == SPBLAS_GFX950_F32 ? 4 : 8;
  void *x = nullptr, *y = nullptr;
  if (dev_alloc(&x, (size_t) pl->n * tsz, s) != SPBLAS_GFX950_STATUS_SUCCESS)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  if (dev_alloc(&y, (size_t) pl->m * tsz, s) != SPBLAS_GFX950_STATUS_SUCCESS) {
    dev_free(x, s);
    return SPBLAS_GFX950_STATUS_SUCCESS;
  }
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  bool ok = hipMemsetAsync(x, 0, (size_t) pl->n * tsz, s) == hipSuccess;
  for (int i = 0; i < 3 && ok; ++i)
    ok = hipEventCreate(&ev[i]) == hipSuccess;
  const double one = 1.0, zero = 0.0;
  const float onef = 1.f, zerof = 0.f;
  const void* alpha = tsz == 4 ? (const void*) &onef : (const void*) &one;
  const void* beta = tsz == 4 ? (const void*) &zerof : (const void*) &zero;
  auto run = [&](int alg) {
    const int saved = pl->alg;
    pl->alg = alg;
    const int rc = spblas_gfx950_spmv(h, pl, SPBLAS_GFX950_OP_N, pl->m, pl->n, pl->nnz, alpha, pl->rowptr, pl->colind, values, x,
                                      beta, y, pl->offset_type, pl->value_type);
    pl->alg = saved;
    return rc;
  };
  float best[2] = {1e30f, 1e30f};  // {row-block, sliced}
  auto timed = [&](int alg, float* ms) {
    *ms = 1e30f;
    return hipEventRecord(ev[0], s) == hipSuccess && run(alg) == SPBLAS_GFX950_STATUS_SUCCESS &&
           hipEventRecord(ev[1], s) == hipSuccess && hipEventSynchronize(ev[1]) == hipSuccess &&
           hipEventElapsedTime(ms, ev[0], ev[1]) == hipSuccess;
  };
  // the sliced plan: a warm-up (first launch of the kernels, workspace growth), then one timed run -- the plans differ by
  // far more than the run-to-run noise.  The row-block kernel: its FIRST run is timed too, and when that is already 1.5x
  // the sliced plan's time the second run is not spent (cfg4: 3.4 of the 38 ms of inspect; a first run is slower than a
  // later one by far less than that)
  ok = ok && run(SPBLAS_GFX950_SPMV_SLICED) == SPBLAS_GFX950_STATUS_SUCCESS && timed(SPBLAS_GFX950_SPMV_SLICED, &best[1]);
  if (ok) {
    float first = 1e30f;
    ok = timed(SPBLAS_GFX950_SPMV_ROWBLOCK, &first);
    best[0] = first;
    if (ok && first <= 1.5f * best[1]) {
      float second = 1e30f;
      ok = timed(SPBLAS_GFX950_SPMV_ROWBLOCK, &second);
      if (ok && second < best[0])
        best[0] = second;
    }
  }
  (void) hipStreamSynchronize(s);
  for (int i = 0; i < 3; ++i)
    if (ev[i])
      (void) hipEventDestroy(ev[i]);
  dev_free(x, s);
  dev_free(y, s);
  if (ok) {
    pl->trial_ms[0] = best[0];
    pl->trial_ms[1] = best[1];
    *sliced_wins = best[1] < 0.95f * best[0];  // the re-tiled copy costs memory: it has to win clearly
  }
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

// Product-store flavour of a SLICED plan.  Which one is faster is a property of the box and the moment, not of the matrix:
// where the reduce pays for the expand's write-backs the non-temporal hint wins 1-4 %, elsewhere it loses 3 %
// (tools/exp_r03o.sh, profiles/r03_store_trial.md; on the round-3 driver box the two differed by 0.7 %).  Default: plain
// stores, no trial.  Opt-in (SPBLAS_GFX950_OPT_STORE_TRIAL = 2 on the handle, or SPBLAS_GFX950_PB_NT=-2): the first plan of
// the HANDLE with >= 32 M placed entries per value size times the plan both ways (a warm-up and three samples of two SpMVs
// each, interleaved, on a zero vector) and the handle keeps the decision for its later plans -- no process-wide state.
// SPBLAS_GFX950_PB_NT = 0 / 1 forces a flavour (reproducible runs, profiles); option value 0 / 1 does the same per handle.
// SPBLAS_GFX950_PB_TUNE_MIN: test hook, the number of placed entries from which a plan counts as large.
static void store_trial(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values) {
  const int env = env_int_spmv("SPBLAS_GFX950_PB_NT", -1);
  const int mode = env == 0 || env == 1 ? env : env == -2 ? 2 : (int) h->store_flavour;
  pl->nt_products = mode == 1;
  if (mode != 2 || pl->s_placed < (int64_t) env_int_spmv("SPBLAS_GFX950_PB_TUNE_MIN", 32 << 20))
    return;
  const size_t tsz = pl->value_type == SPBLAS_GFX950_F32 ? 4 : 8;
  int& slot = h->nt_choice[tsz == 4 ? 0 : 1];
  if (slot != 0) {
    pl->nt_products = slot == 2;
    return;
  }
  hipStream_t s = h->stream;
  void *x = nullptr, *y = nullptr;
  if (dev_alloc(&x, (size_t) pl->n * tsz, s) != SPBLAS_GFX950_STATUS_SUCCESS)
    return;
  if (dev_alloc(&y, (size_t) pl->m * tsz, s) != SPBLAS_GFX950_STATUS_SUCCESS) {
    dev_free(x, s);
    return;
  }
  hipEvent_t ev[2] = {nullptr, nullptr};
  bool ok = hipMemsetAsync(x, 0, (size_t) pl->n * tsz, s) == hipSuccess;
  for (int i = 0; i < 2 && ok; ++i)
    ok = hipEventCreate(&ev[i]) == hipSuccess;
  const double one = 1.0, zero = 0.0;
  const float onef = 1.f, zerof = 0.f;
  const void* alpha = tsz == 4 ? (const void*) &onef : (const void*) &one;
  const void* beta = tsz == 4 ? (const void*) &zerof : (const void*) &zero;
  const int saved_alg = pl->alg;
  pl->alg = SPBLAS_GFX950_SPMV_SLICED;
  auto run = [&](int nt) {
    pl->nt_products = nt;
    return spblas_gfx950_spmv(h, pl, SPBLAS_GFX950_OP_N, pl->m, pl->n, pl->nnz, alpha, pl->rowptr, pl->colind, values, x, beta, y,
                              pl->offset_type, pl->value_type) == SPBLAS_GFX950_STATUS_SUCCESS;
  };
  // one sample = two SpMVs (the second one's expand runs behind a reduce, as in a solver loop), in ms per SpMV
  auto sample = [&](int nt, float* out) {
    float ms = 0.f;
    const bool good = hipEventRecord(ev[0], s) == hipSuccess && run(nt) && run(nt) && hipEventRecord(ev[1], s) == hipSuccess &&
                      hipEventSynchronize(ev[1]) == hipSuccess && hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess;
    *out = 0.5f * ms;
    return good;
  };
  if (ok) {
    float best[2] = {1e30f, 1e30f};
    ok = run(0) && run(1);  // warm-up of both kernels
    for (int rep = 0; rep < 3 && ok; ++rep)
      for (int nt = 0; nt < 2 && ok; ++nt) {
        float ms = 0.f;
        ok = sample(nt, &ms);
        if (ok && ms < best[nt])
          best[nt] = ms;
      }
    pl->nt_products = 0;
    if (ok) {
      pl->store_trial_ms[0] = best[0];
      pl->store_trial_ms[1] = best[1];
      // the hint has to win by more than the noise of three samples (where it matters it wins by 2-4 %; where it does
      // not, it loses by as much)
      pl->nt_products = best[1] < 0.995f * best[0];
      slot = pl->nt_products ? 2 : 1;
    }
  }
  (void) hipStreamSynchronize(s);
  for (int i = 0; i < 2; ++i)
    if (ev[i])
      (void) hipEventDestroy(ev[i]);
  dev_free(x, s);
  dev_free(y, s);
  pl->alg = saved_alg;
}

} // namespace spb

using namespace spb;

extern "C" {

int spblas_gfx950_spmv_plan_create(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t* plan, int64_t m,
                                   int64_t n, int64_t nnz, const void* rowptr, const int32_t* colind,
                                   const void* values, int offset_type, int value_type, int alg) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (stream_capturing(handle->stream))  // inspect-class call: sizes its output on the host, never part of a graph
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  if (!plan)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  *plan = nullptr;
  if (m < 0 || n < 0 || nnz < 0 || m > INT32_MAX || n > INT32_MAX)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  if (offset_type == SPBLAS_GFX950_I32 && nnz > INT32_MAX)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  if ((offset_type != SPBLAS_GFX950_I32 && offset_type != SPBLAS_GFX950_I64) ||
      (value_type != SPBLAS_GFX950_F32 && value_type != SPBLAS_GFX950_F64) || alg < 0 ||
      alg > SPBLAS_GFX950_SPMV_SLICED)
    return SPBLAS_GFX950_STATUS_INVALID_VALUE;
  if (!rowptr || (nnz > 0 && !colind))
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (alg == SPBLAS_GFX950_SPMV_SLICED && nnz > 0 && !values)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;

  auto* pl = new (std::nothrow) spblas_gfx950_plan_s();
  if (!pl)
    return SPBLAS_GFX950_STATUS_ALLOC_FAILED;
  pl->m = m;
  pl->n = n;
  pl->nnz = nnz;
  pl->rowptr = rowptr;
  pl->colind = colind;
  pl->offset_type = offset_type;
  pl->value_type = value_type;
  // (SPBLAS_GFX950_OPT_VALUE_SNAPSHOT = 2: the caller announces that the values WILL change -- a snapshot plan then keeps
  // its source positions from the start and the first update_values is a gather, not a second inspect)
  pl->keep_src = handle->value_snapshot == 2 ? 1 : 0;
  int rc = offset_type == SPBLAS_GFX950_I32 ? plan_build<int32_t>(handle, pl, alg)
                                            : plan_build<int64_t>(handle, pl, alg);
  pl->base_device_bytes = pl->device_bytes;
  if (rc == SPBLAS_GFX950_STATUS_SUCCESS && pl->alg == SPBLAS_GFX950_SPMV_SLICED) {
    // (an explicitly requested SLICED plan keeps its copy of the values -- the documented snapshot -- unless the test hook
    // SPBLAS_GFX950_PB_VFREE=2 asks for the value-free form, which then reads the caller's array on every multiply)
    if (env_int_spmv("SPBLAS_GFX950_PB_VFREE", 1) == 2) {
      pl->vfree = 1;
      pl->refresh_each_call = 1;
    }
    rc = spmv_sliced_build(handle, pl, values, false);
    if (rc == SPBLAS_GFX950_STATUS_SUCCESS && !pl->vfree && env_int_spmv("SPBLAS_GFX950_PB_VFREE", 1) == 2)
      pl->refresh_each_call = 0;
    if (rc == SPBLAS_GFX950_STATUS_SUCCESS)
      store_trial(handle, pl, values);
  } else if (rc == SPBLAS_GFX950_STATUS_SUCCESS && alg == SPBLAS_GFX950_SPMV_AUTO && values &&
             (handle->value_snapshot != 0 ||
              // (plain operands: large and NOT skewed -- a power-law matrix would get the hot-column split, whose refresh
              // gathers through two source maps: 6.9 against 3.4 ms for the row-block kernel at cfg4, after 50 ms of inspect)
              (pl->nnz >= ((int64_t) 16 << 20) && env_int_spmv("SPBLAS_GFX950_PLAIN_SLICED", 1) &&
               (double) pl->max_row_len <= 16.0 * ((double) pl->nnz / (double) (pl->m > 0 ? pl->m : 1)) + 64.0 &&
               pl->empty_rows * 4 <= pl->m)) &&
             pl->alg == SPBLAS_GFX950_SPMV_ROWBLOCK && sliced_candidate(pl)) {
    // x far larger than an XCD's L2 and no long rows: try the LDS-sliced re-tiling; it
    // declines (NOT_SUPPORTED) when the entries cluster in few tiles.
    // The sliced plan multiplies with a re-tiled COPY of the values.  With SPBLAS_GFX950_OPT_VALUE_SNAPSHOT (matrix_opt)
    // the copy is taken at inspect and when the caller passes another array.  WITHOUT the opt-in (a plain inspected
    // csr_view, whose multiply must read the caller's values of that call: multiply_impl.hpp:48-52) the plan takes the
    // values again on EVERY multiply -- pb_refresh_bins_kernel, 0.38 ms at cfg2 -- and is kept only if refresh + tiles
    // beat the row-block kernel in a timed trial (cfg2: 0.76 against 1.71 ms).  Large matrices only: the plan is a
    // second copy of A (SPBLAS_GFX950_PLAIN_SLICED=0: row-block plan as before round 4's end).
    // Round 5: such a plan is built VALUE-FREE when the matrix allows it (spmv_sliced.hip, pb_reduce_vf_kernel): no copy
    // of the values at all -- the expand moves x[col], the reduce multiplies by the caller's array through an LDS window
    // per bin -- and falls back to the copying form otherwise.
    pl->refresh_each_call = handle->value_snapshot != 0 ? 0 : 1;
    pl->vfree = pl->refresh_each_call;
    const int rc2 = spmv_sliced_build(handle, pl, values, true);
    if (rc2 == SPBLAS_GFX950_STATUS_SUCCESS) {
      bool keep = true;
      // (the copying form of a refreshing plan pays a value refresh per multiply and is only kept when it beats the
      // row-block kernel in a timed trial; the value-free form costs what a snapshot plan costs plus the window reads --
      // 0.37 against 1.71 ms at cfg2 -- and is decided by the same static rules as a snapshot plan: no trial, whose two
      // row-block multiplies alone were 3.4 of the 10 ms of this inspect)
      // Round 6: by default a RULE, not a stopwatch, so that the same matrix gets the same plan -- and the same bits -- on
      // every box: skewed matrices (s_uncertain) keep the tiles when x is far larger than what the caches hold for the
      // row-block kernel's gathers (measured pairs, tiled vs row-block: R-MAT scale 20, x of 4 - 8 MB: 0.14 / 0.18 vs 0.13 /
      // 0.14 ms; scale 22, x of 16 / 32 MB: 0.40 vs 0.47 and 0.59 vs 0.52; scale 24, x of 64 / 128 MB: 1.47 vs 2.33 and 1.7
      // vs 3.4; its eight row shards 0.31 - 0.37 vs 0.44 - 0.58), copying plans that refresh their values on every multiply
      // from 32 MB (cfg2, 40 MB: 0.76 vs 1.71).  SPBLAS_GFX950_AUTO_TRIAL=1 brings the timed trial back, =0 the static
      // decline of rounds 2 - 5.
      if (pl->s_uncertain || (pl->refresh_each_call && !pl->vfree)) {
        if (env_int_spmv("SPBLAS_GFX950_AUTO_TRIAL", -1) == 1) {
          (void) auto_trial(handle, pl, values, &keep);
        } else {
          const double x_bytes = (double) pl->n * (pl->value_type == SPBLAS_GFX950_F32 ? 4.0 : 8.0);
          keep = x_bytes >= (pl->s_uncertain ? 40.0 : 32.0) * 1024.0 * 1024.0;
        }
      }
      if (keep) {
        pl->alg = SPBLAS_GFX950_SPMV_SLICED;
        store_trial(handle, pl, values);
      } else {
        spmv_sliced_free(handle, pl);
        pl->refresh_each_call = 0;
      }
    } else {
      spmv_sliced_free(handle, pl);
      pl->refresh_each_call = 0;
      pl->vfree = 0;
      if (rc2 != SPBLAS_GFX950_STATUS_NOT_SUPPORTED)
        rc = rc2;
    }
  }
  if (rc != SPBLAS_GFX950_STATUS_SUCCESS) {
    spblas_gfx950_plan_destroy(handle, pl);
    return rc;
  }
  *plan = pl;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_spmv_plan_detach(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!plan)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (plan->detached)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  // self-contained: the tiles hold their own copy of the values and every row (no hub rows multiplied from the caller's
  // arrays, no hot-column split with its own CSR parts, no per-call value refresh, not the value-free form)
  if (plan->alg != SPBLAS_GFX950_SPMV_SLICED || plan->vfree || plan->refresh_each_call || plan->n_hub > 0 || plan->rest_plan ||
      plan->hot_plan || plan->nnz == 0)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  plan->detached = 1;
  plan->rowptr = nullptr;
  plan->colind = nullptr;
  plan->values_ptr = nullptr;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_spmv_plan_update_values(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan,
                                          const void* values) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!plan || !values)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (plan->detached)  // (the source positions index arrays that are gone: inspect again)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  if (plan->alg != SPBLAS_GFX950_SPMV_SLICED)
    return SPBLAS_GFX950_STATUS_SUCCESS;  // other algorithms read the caller's values directly
  return spmv_sliced_update(handle, plan, values);
}

int spblas_gfx950_spmv_expand(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, const void* x) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!plan || !x)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (plan->alg != SPBLAS_GFX950_SPMV_SLICED || plan->refresh_each_call)  // (the two-stage form is not given A's values)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  if (plan->nnz == 0)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  return spmv_sliced_expand(handle, plan, x);
}

int spblas_gfx950_spmv_reduce_rows(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, const void* alpha,
                                   const void* beta, void* y, int64_t row_begin, int64_t row_end) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!plan || !alpha || !beta || !y)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (plan->alg != SPBLAS_GFX950_SPMV_SLICED)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  if (row_begin < 0 || row_end > plan->m || row_begin > row_end)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  return spmv_sliced_reduce_rows(handle, plan, alpha, beta, y, row_begin, row_end, nullptr, 0, 0);
}

// The step wait armed by spblas_gfx950_bcast_wait_before belongs to exactly ONE bcast call: whatever path that call
// leaves by (argument checks, NOT_SUPPORTED for hub rows, a refused workspace, an empty range, a failed launch), the wait
// must not stay armed on the handle and fire inside a later, unrelated reduce with a stale step and status pointer.
struct bcast_wait_disarm {
  spblas_gfx950_handle_t h;
  ~bcast_wait_disarm() {
    h->bcast_wait.flags = nullptr;
  }
};

int spblas_gfx950_spmv_reduce_rows_bcast(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, const void* alpha,
                                         void* const* y_peers, int n_peers, int64_t y_row_offset,
                                         int64_t row_begin, int64_t row_end) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  bcast_wait_disarm disarm{handle};
  if (!plan || !alpha || !y_peers)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  // (a plan made without the snapshot opt-in must take A's values of THIS call: these entry points are not given them.  A
  // value-free plan holds no copy that could go stale: it reads the array registered with the plan -- plan_create, the last
  // spblas_gfx950_spmv, plan_update_values -- as it is when the step runs, which is the same promise.  Round 6.)
  if (plan->alg != SPBLAS_GFX950_SPMV_SLICED || (plan->refresh_each_call && !plan->vfree))
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  if (n_peers < 1 || y_row_offset < 0 || row_begin < 0 || row_end > plan->m || row_begin > row_end)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  const double zero = 0.0;  // all-zero bit pattern: beta = 0 for both value types
  return spmv_sliced_reduce_rows(handle, plan, alpha, &zero, nullptr, row_begin, row_end, y_peers, n_peers,
                                 y_row_offset);
}

// One fused multi-GPU step in a single host call: expand, then the reduce of `stripes` contiguous groups
// of row bins, alternating between the handle's stream and an auxiliary stream of the handle.  With the
// slice split active each stripe ends in a combine kernel whose peer stores are bound by the xGMI
// links, not by the CUs; on the other stream the next stripe's reduce kernel runs meanwhile, so the
// link time hides behind the remaining reduces instead of following them.  The caller's stream is
// joined with the auxiliary one before returning (stream order, no host wait).
int spblas_gfx950_spmv_step_bcast(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, const void* alpha,
                                  const void* x, void* const* y_peers, int n_peers, int64_t y_row_offset,
                                  int stripes) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  bcast_wait_disarm disarm{handle};
  if (!plan || !alpha || !x || !y_peers)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (plan->alg != SPBLAS_GFX950_SPMV_SLICED || (plan->refresh_each_call && !plan->vfree))  // (as above)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  if (n_peers < 1 || y_row_offset < 0 || stripes < 1)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  if (plan->nnz == 0 || plan->m == 0)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  const double zero = 0.0;
  const int64_t NB = plan->n_rblk, H = plan->rows_per_blk, RW = plan->rwaves;
  int64_t per = cdiv(cdiv(NB, RW), stripes) * RW;  // bins per stripe: whole workgroups
  if (per < RW)
    per = RW;
  const int n_str = plan->s_binrow ? 1 : (int) cdiv(NB, per);  // stripes are cut on the arithmetic bin grid only
  hipStream_t main_s = handle->stream;
  int rc;
  if (n_str <= 1) {
    if ((rc = spmv_sliced_expand(handle, plan, x)))
      return rc;
    return spmv_sliced_reduce_rows(handle, plan, alpha, &zero, nullptr, 0, plan->m, y_peers, n_peers, y_row_offset);
  }
  if (!handle->aux_stream) {
    SPB_HIP(hipStreamCreateWithFlags(&handle->aux_stream, hipStreamNonBlocking));
    SPB_HIP(hipEventCreateWithFlags(&handle->ev_fork, hipEventDisableTiming));
    SPB_HIP(hipEventCreateWithFlags(&handle->ev_join, hipEventDisableTiming));
  }
  const int K = spmv_sliced_full_ksplit(plan);
  if ((rc = spmv_sliced_reserve_partial(handle, plan, K)))
    return rc;
  if ((rc = spmv_sliced_expand(handle, plan, x)))
    return rc;
  SPB_HIP(hipEventRecord(handle->ev_fork, main_s));
  SPB_HIP(hipStreamWaitEvent(handle->aux_stream, handle->ev_fork, 0));
  const int64_t saved_cap = handle->max_ksplit;
  handle->max_ksplit = K;
  rc = SPBLAS_GFX950_STATUS_SUCCESS;
  for (int c = 0; c < n_str && rc == SPBLAS_GFX950_STATUS_SUCCESS; ++c) {
    const int64_t b0 = c * per, b1 = (c + 1) * per < NB ? (c + 1) * per : NB;
    const int64_t r_lo = b0 * H, r_hi = b1 * H < plan->m ? b1 * H : plan->m;
    handle->stream = (c & 1) ? handle->aux_stream : main_s;
    rc = spmv_sliced_reduce_rows(handle, plan, alpha, &zero, nullptr, r_lo, r_hi, y_peers, n_peers, y_row_offset);
  }
  handle->stream = main_s;
  handle->max_ksplit = saved_cap;
  SPB_HIP(hipEventRecord(handle->ev_join, handle->aux_stream));
  SPB_HIP(hipStreamWaitEvent(main_s, handle->ev_join, 0));
  return rc;
}

// Row boundaries of the chunks a chunked step cuts this plan's rows into (local rows, chunks + 1 entries, host array):
// the stripes of spmv_step_bcast -- contiguous groups of whole reduce workgroups on the arithmetic bin grid.  A plan with
// fewer stripes than asked for leaves the last chunks empty.
static int chunk_layout(const spblas_gfx950_plan_s* plan, int chunks, int64_t* per_bins, int* n_str) {
  const int64_t NB = plan->n_rblk, RW = plan->rwaves;
  int64_t per = cdiv(cdiv(NB, RW), chunks) * RW;
  if (per < RW)
    per = RW;
  *per_bins = per;
  *n_str = plan->s_binrow ? 1 : (int) cdiv(NB, per);
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

// K-split plans (the shards of N >= 8 ranks): the combine kernel publishes the chunks itself, the reduce stays ONE launch
static bool chunk_in_combine(spblas_gfx950_plan_s* plan) {
  return !plan->s_nzrow && plan->n_split == 0 && spmv_sliced_full_ksplit(plan) > 1 &&
         env_int_spmv("SPBLAS_GFX950_CHUNK_STRIPES", 0) == 0;
}

int spblas_gfx950_spmv_chunk_rows(spblas_gfx950_plan_t plan, int chunks, int64_t* rows) {
  if (!plan || !rows)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (chunks < 1 || chunks > 64)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  if (plan->alg != SPBLAS_GFX950_SPMV_SLICED || plan->rest_plan || plan->m == 0 || plan->nnz == 0)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  if (chunk_in_combine(plan)) {  // chunks of whole combine workgroups (2 048 rows: spmv_sliced.hip PB_PUB_ROWS)
    const int64_t rpc = cdiv(cdiv(plan->m, chunks), 2048) * 2048;
    for (int c = 0; c <= chunks; ++c)
      rows[c] = c * rpc < plan->m ? c * rpc : plan->m;
    rows[chunks] = plan->m;
    return SPBLAS_GFX950_STATUS_SUCCESS;
  }
  int64_t per = 0;
  int n_str = 0;
  chunk_layout(plan, chunks, &per, &n_str);
  const int64_t H = plan->rows_per_blk;
  for (int c = 0; c <= chunks; ++c) {
    const int64_t b = c < n_str ? (int64_t) c * per : plan->n_rblk;
    const int64_t r = n_str <= 1 ? (c == 0 ? 0 : plan->m) : b * H;
    rows[c] = r < plan->m ? r : plan->m;
  }
  rows[chunks] = plan->m;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

// Dependent iteration over row shards with the all-gather fused in AND overlapped with the next expand (round 4).  x must
// be this rank's copy of the previous step's y (all ranks' rows, the buffer the peers stored into).  The expand waits per x
// slice for the chunks of the peers' rows that slice is made of (wait->flags: this rank's flag array, slot q * chunks + c);
// the reduce runs in `chunks` stripes as in spmv_step_bcast, each stripe's peer stores followed -- on the stripe's stream --
// by a kernel that publishes slot rank * chunks + c = step in every rank's flag array.  wait == NULL: the first step (x is
// complete everywhere: a plain expand).  No step barrier: buffer safety comes from the data dependence itself (a rank
// overwrites copy k & 1 in step k + 2, whose expand has waited for every rank's chunks of step k + 1, which they publish
// after their expand of step k + 1 has read copy k & 1).
int spblas_gfx950_spmv_step_bcast_chunked(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, const void* alpha,
                                          const void* x, void* const* y_peers, int n_peers, int64_t y_row_offset,
                                          int chunks, void* const* flag_peers, int rank, int64_t step,
                                          const spblas_gfx950_chunk_wait* wait) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  struct disarm_t {
    spblas_gfx950_handle_t h;
    ~disarm_t() {
      h->chunk_wait.flags = nullptr;
    }
  } disarm{handle};
  if (!plan || !alpha || !x || !y_peers || !flag_peers)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (plan->alg != SPBLAS_GFX950_SPMV_SLICED || plan->rest_plan || plan->refresh_each_call)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  if (n_peers < 1 || n_peers > 64 || rank < 0 || rank >= n_peers || y_row_offset < 0 || chunks < 1 || chunks > 64)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  if (plan->nnz == 0 || plan->m == 0 || plan->n_split > 0 || plan->s_binrow)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  if (wait) {
    if (!wait->flags || !wait->chunk_rows || !wait->status_dev || wait->n_ranks != n_peers || wait->chunks != chunks)
      return SPBLAS_GFX950_STATUS_INVALID_VALUE;
    auto& cw = handle->chunk_wait;
    cw.flags = static_cast<const long long*>(wait->flags);
    cw.chunk_rows = reinterpret_cast<const long long*>(wait->chunk_rows);
    cw.n_ranks = wait->n_ranks;
    cw.chunks = wait->chunks;
    cw.rank = rank;
    cw.step = (long long) wait->step;
    cw.timeout_ticks = (long long) wait->timeout_ms * wall_clock_khz(handle);
    cw.status_dev = wait->status_dev;
    cw.max_wgs = wait->max_expand_workgroups;
  }
  const double zero = 0.0;
  const int64_t delay_us = env_int_spmv("SPBLAS_GFX950_CHUNK_DELAY_US", 0);  // test hook: chunk 1 of every step is late
  hipStream_t main_s = handle->stream;
  int rc;
  if (chunk_in_combine(plan)) {
    // one expand, one reduce, one combine that publishes its chunks as their last workgroup finishes
    if (!handle->chunk_done) {
      if (stream_capturing(main_s))
        return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
      SPB_HIP(hipMalloc((void**) &handle->chunk_done, 64 * sizeof(int)));
      SPB_HIP(hipMemset(handle->chunk_done, 0, 64 * sizeof(int)));
    }
    if ((rc = spmv_sliced_expand(handle, plan, x)))
      return rc;
    auto& cp = handle->chunk_pub;
    cp.flag_peers = flag_peers;
    cp.n_peers = n_peers;
    cp.slot0 = rank * chunks;
    cp.chunks = chunks;
    cp.step = (long long) step;
    cp.rows_per_chunk = cdiv(cdiv(plan->m, chunks), 2048) * 2048;
    cp.delay_ticks = (long long) (delay_us * wall_clock_khz(handle) / 1000);
    rc = spmv_sliced_reduce_rows(handle, plan, alpha, &zero, nullptr, 0, plan->m, y_peers, n_peers, y_row_offset);
    const bool consumed = cp.flag_peers == nullptr;
    cp.flag_peers = nullptr;
    if (rc)
      return rc;
    // (a reduce that did not end in the K-split combine after all: publish at the kernel boundary)
    return consumed ? SPBLAS_GFX950_STATUS_SUCCESS
                    : launch_chunk_signal(handle, main_s, flag_peers, n_peers, rank * chunks, chunks, step, 0);
  }
  int64_t per = 0;
  int n_str = 0;
  chunk_layout(plan, chunks, &per, &n_str);
  const int64_t NB = plan->n_rblk, H = plan->rows_per_blk;
  if (n_str <= 1) {
    if ((rc = spmv_sliced_expand(handle, plan, x)))
      return rc;
    if ((rc = spmv_sliced_reduce_rows(handle, plan, alpha, &zero, nullptr, 0, plan->m, y_peers, n_peers, y_row_offset)))
      return rc;
    return launch_chunk_signal(handle, main_s, flag_peers, n_peers, rank * chunks, chunks, step, delay_us);
  }
  if (!handle->aux_stream) {
    SPB_HIP(hipStreamCreateWithFlags(&handle->aux_stream, hipStreamNonBlocking));
    SPB_HIP(hipEventCreateWithFlags(&handle->ev_fork, hipEventDisableTiming));
    SPB_HIP(hipEventCreateWithFlags(&handle->ev_join, hipEventDisableTiming));
  }
  const int K = spmv_sliced_full_ksplit(plan);
  if ((rc = spmv_sliced_reserve_partial(handle, plan, K)))
    return rc;
  if ((rc = spmv_sliced_expand(handle, plan, x)))
    return rc;
  SPB_HIP(hipEventRecord(handle->ev_fork, main_s));
  SPB_HIP(hipStreamWaitEvent(handle->aux_stream, handle->ev_fork, 0));
  const int64_t saved_cap = handle->max_ksplit;
  handle->max_ksplit = K;
  rc = SPBLAS_GFX950_STATUS_SUCCESS;
  for (int c = 0; c < n_str && rc == SPBLAS_GFX950_STATUS_SUCCESS; ++c) {
    const int64_t b0 = c * per, b1 = (c + 1) * per < NB ? (c + 1) * per : NB;
    const int64_t r_lo = b0 * H, r_hi = b1 * H < plan->m ? b1 * H : plan->m;
    handle->stream = (c & 1) ? handle->aux_stream : main_s;
    rc = spmv_sliced_reduce_rows(handle, plan, alpha, &zero, nullptr, r_lo, r_hi, y_peers, n_peers, y_row_offset);
    if (rc == SPBLAS_GFX950_STATUS_SUCCESS)  // (the chunks past the last stripe are empty: published with it)
      rc = launch_chunk_signal(handle, handle->stream, flag_peers, n_peers, rank * chunks + c, c == n_str - 1 ? chunks - c : 1, step,
                               c == 1 ? delay_us : 0);
  }
  handle->stream = main_s;
  handle->max_ksplit = saved_cap;
  SPB_HIP(hipEventRecord(handle->ev_join, handle->aux_stream));
  SPB_HIP(hipStreamWaitEvent(main_s, handle->ev_join, 0));
  return rc;
}

int spblas_gfx950_plan_destroy(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!plan)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  hipStream_t s = handle->stream;
  if (plan->used && plan->last_stream != s) {
    // the frees below are ordered on the handle's stream: put them behind the last launch that used the plan
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
      if (hipEventRecord(ev, plan->last_stream) != hipSuccess || hipStreamWaitEvent(s, ev, 0) != hipSuccess)
        (void) hipStreamSynchronize(plan->last_stream);
      (void) hipEventDestroy(ev);
    } else {
      (void) hipStreamSynchronize(plan->last_stream);
    }
    (void) hipGetLastError();
  }
  spmv_plan_release(handle, plan);
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_plan_info(spblas_gfx950_plan_t plan, int64_t info[12]) {
  if (!plan || !info)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  info[0] = plan->alg;
  info[1] = plan->win;
  info[2] = plan->nwin;
  info[3] = plan->n_long;
  info[4] = plan->max_row_len;
  info[5] = (int64_t) plan->device_bytes;
  const spblas_gfx950_plan_s* tp = plan->rest_plan ? plan->rest_plan : plan;  // the tiles of a split plan are A_rest's
  info[6] = tp->n_slices;
  info[7] = plan->empty_rows;
  info[8] = tp->rows_per_blk;
  info[9] = tp->bin_aligned;
  info[10] = tp->n_xitems;
  info[11] = tp->n_ritems;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_plan_info_sliced(spblas_gfx950_plan_t plan, int64_t info[12]) {
  if (!plan || !info)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  const bool sl = plan->alg == SPBLAS_GFX950_SPMV_SLICED;
  const spblas_gfx950_plan_s* tp = plan->rest_plan ? plan->rest_plan : plan;  // the tiles of a split plan are A_rest's
  info[0] = sl ? tp->n_rblk : 0;
  info[1] = sl && tp->s_binrow ? 1 : 0;
  info[2] = sl ? tp->a_blocks : 0;
  info[3] = sl ? tp->p_blocks : 0;
  info[4] = sl ? tp->s_placed : 0;
  info[5] = sl && tp->hub_len > 0 ? tp->n_hub : 0;
  info[6] = sl ? tp->hub_len : 0;
  info[7] = sl ? tp->n_ksplit : 0;
  info[8] = sl ? tp->s_m : 0;
  // bit 0: AUTO ran its trial, bit 1: one-byte row codes, bit 2: non-temporal product stores, bit 3: this plan ran the store
  // trial, bit 4: hot-column split (spblas_gfx950_plan_info_hot has the numbers), bit 6: the plan takes A's values again on
  // every multiply (made without the snapshot opt-in)
  info[9] = (plan->trial_ms[0] > 0.f ? 1 : 0) | (sl && tp->enc8 ? 2 : 0) | (sl && plan->nt_products ? 4 : 0) |
            (plan->store_trial_ms[0] > 0.f ? 8 : 0) | (sl && plan->rest_plan ? 16 : 0) | (sl && plan->refresh_each_call ? 64 : 0) |
            (sl && tp->vfree ? 128 : 0);  // bit 7: value-free tiles (no copy of A's values in the plan)
  const float* tms = plan->trial_ms[0] > 0.f ? plan->trial_ms : plan->store_trial_ms;  // AUTO's times, else the store trial's
  info[10] = (int64_t) (tms[0] * 1e6f);
  info[11] = (int64_t) (tms[1] * 1e6f);
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_plan_info_hot(spblas_gfx950_plan_t plan, int64_t info[8]) {
  if (!plan || !info)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  const bool on = plan->alg == SPBLAS_GFX950_SPMV_SLICED && plan->hot_plan && plan->rest_plan;
  info[0] = on ? plan->hot_k : 0;                  // hot columns (their x values live in LDS)
  info[1] = on ? plan->hot_nnz : 0;                // entries multiplied in row order out of LDS
  info[2] = on ? plan->hot_m : 0;                  // rows that have such entries
  info[3] = on ? plan->hot_ncross : 0;             // ... whose entries lie in more than one window of 256
  info[4] = on ? plan->rest_plan->nnz : 0;         // entries left to the tiled plan
  info[5] = on ? (int64_t) plan->rest_plan->device_bytes : 0;
  info[6] = on ? plan->hot_plan->nwin : 0;         // windows of 256 entries of A_hot
  info[7] = 0;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_spmv(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, int op, int64_t m, int64_t n,
                       int64_t nnz, const void* alpha, const void* rowptr, const int32_t* colind,
                       const void* values, const void* x, const void* beta, void* y, int offset_type,
                       int value_type) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (m < 0 || n < 0 || nnz < 0 || m > INT32_MAX || n > INT32_MAX)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  if (offset_type == SPBLAS_GFX950_I32 && nnz > INT32_MAX)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  if ((op != SPBLAS_GFX950_OP_N && op != SPBLAS_GFX950_OP_T) ||
      (offset_type != SPBLAS_GFX950_I32 && offset_type != SPBLAS_GFX950_I64) ||
      (value_type != SPBLAS_GFX950_F32 && value_type != SPBLAS_GFX950_F64))
    return SPBLAS_GFX950_STATUS_INVALID_VALUE;
  const int64_t ylen = op == SPBLAS_GFX950_OP_N ? m : n, xlen = op == SPBLAS_GFX950_OP_N ? n : m;
  const bool detached = plan && plan->detached;  // (spblas_gfx950_spmv_plan_detach: the multiply takes no matrix arrays)
  if (!alpha || !beta || (!detached && (!rowptr || (nnz > 0 && (!colind || !values)))) || (ylen > 0 && !y) ||
      (xlen > 0 && nnz > 0 && !x))
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (detached) {
    if (plan->m != m || plan->n != n || plan->nnz != nnz || plan->offset_type != offset_type || plan->value_type != value_type)
      return SPBLAS_GFX950_STATUS_PLAN_MISMATCH;
    if (op != SPBLAS_GFX950_OP_N)
      return SPBLAS_GFX950_STATUS_INVALID_VALUE;
  } else if (plan) {
    if (plan->m != m || plan->n != n || plan->nnz != nnz || plan->rowptr != rowptr ||
        plan->colind != colind || plan->offset_type != offset_type || plan->value_type != value_type)
      return SPBLAS_GFX950_STATUS_PLAN_MISMATCH;
    if (op != SPBLAS_GFX950_OP_N)
      plan = nullptr;  // plans describe op = N only
    // a SLICED plan multiplies with its own re-tiled copy of the values: when the caller hands over another
    // array than the one the copy was taken from, take the copy again first
    // (a plan made without the snapshot opt-in does so on EVERY multiply: the caller may have rewritten the array in place,
    // and a plain inspected csr_view promises what multiply_impl.hpp:48-52 does -- the values of this call)
    if (plan && plan->alg == SPBLAS_GFX950_SPMV_SLICED && nnz > 0 && (values != plan->values_ptr || plan->refresh_each_call)) {
      const int rc = spmv_sliced_update(handle, plan, values);
      if (rc != SPBLAS_GFX950_STATUS_SUCCESS)
        return rc;
    }
  }
  if (value_type == SPBLAS_GFX950_F32) {
    return offset_type == SPBLAS_GFX950_I32
               ? spmv_typed<float, int32_t>(handle, plan, op, m, n, nnz, alpha, rowptr, colind, values, x, beta, y)
               : spmv_typed<float, int64_t>(handle, plan, op, m, n, nnz, alpha, rowptr, colind, values, x, beta, y);
  }
  return offset_type == SPBLAS_GFX950_I32
             ? spmv_typed<double, int32_t>(handle, plan, op, m, n, nnz, alpha, rowptr, colind, values, x, beta, y)
             : spmv_typed<double, int64_t>(handle, plan, op, m, n, nnz, alpha, rowptr, colind, values, x, beta, y);
}

} // extern "C"

// Loads this file's code object (the runtime loads a code object at the first use of one of its kernels: milliseconds
// that would otherwise fall on the caller's first inspect / compute call -- handle.hip: spblas_gfx950_create).
namespace spb {
void preload_spmv() {
  hipFuncAttributes attr;
  (void) hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&scale_vector_kernel<float>));
  (void) hipGetLastError();
}
} // namespace spb
