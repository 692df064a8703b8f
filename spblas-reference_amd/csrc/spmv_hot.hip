// Hot-column split of the SLICED SpMV plan (round 4) -- the same maths as spmv_sliced.hip,
//   y <- alpha * A x + beta * y     (/root/reference/include/spblas/algorithms/multiply_impl.hpp:33-53),
// for matrices whose entries concentrate in few COLUMNS (power-law graphs: on the R-MAT scale-24 matrix of BASELINE cfg4
// the 12 288 most referenced of 16.8 M columns hold 27 % of the entries, tools/rmat_stats.py).
//
// The tiled plan moves 28 B per stored fp64 entry by design (value + 16-bit column in, product out; product + row word
// in): a gather from x is only cheap out of LDS and an entry cannot be ordered by column slice and by row at once.  For
// the hot columns it can: their x values -- ONE slice, whatever their numbers are -- fit the LDS of every CU at the same
// time, so the entries that reference them are multiplied in ROW order, straight into y, at 10 B per entry:
//
//   inspect   sample the column indices (1 in 16) into a histogram, take the K most referenced columns (K = what LDS
//             holds next to the scan strips: 16 320 fp64 / 36 800 fp32) if they cover >= 15 % of the sample, and
//             split A once, stably, into A_hot (values, 16-bit index into the hot list, row offsets) and A_rest (an
//             ordinary CSR matrix with the remaining entries), each with the source position of every entry (for
//             update_values).  A_rest gets the regular tiled plan; A_hot the nnz-window row partition of the row-block
//             plan (spmv.hip: plan_build) with windows of 256 entries.
//   multiply  the tiled plan of A_rest with the caller's alpha and beta (it writes every row of y), then
//             pb_hot_rows_kernel: y += alpha * A_hot x over the rows that HAVE hot entries (A_hot is kept over those rows
//             only, hot_rows[] names them: a window never owns more rows than it has entries, however many rows of the
//             matrix are empty).  One workgroup of 16 wavefronts per CU, the hot x values gathered into LDS once; every
//             wavefront walks its own windows: products staged in a wave-private LDS strip, a lane group per row sums
//             them -- the row-block kernel's scheme at wavefront scope, no workgroup barrier after the fill; rows longer
//             than a window go through per-window partials and pb_hot_fixup_kernel.
//
// Results: a row's sum is associated differently (tiled part, then the hot part added to it) -- inside the parity bound
// like every other plan; bit-reproducible from run to run (no atomics on this path).
// Chosen by spmv_sliced_build for row-skewed matrices (the ones that get variable-height bins) unless the handle asks for
// row-range reduces (SPBLAS_GFX950_OPT_BIN_ROW_ALIGN); SPBLAS_GFX950_PB_HOT = 0 / 1 switches it off / forces the attempt,
// SPBLAS_GFX950_PB_HOT_MIN_PCT the coverage from which the split is taken (default 15).
#include "common.hpp"
#include "plan.hpp"
#include "scan.hpp"

#include <cstdlib>
#include <new>
#include <type_traits>
#include <vector>

namespace spb {

int spmv_plan_structures(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, int alg_req);  // spmv.hip
int spmv_sliced_build(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values, bool auto_mode);
int spmv_sliced_update(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values);
int spmv_sliced_exec(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* alpha, const void* x,
                     const void* beta, void* y);
int spmv_sliced_expand(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* x);
void spmv_sliced_free(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl);
void spmv_plan_release(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl);  // spmv.hip: everything a plan owns, and the plan
int spmv_sliced_reflag_dups(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl);

static int hot_env(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return v && *v ? std::atoi(v) : dflt;
}

#ifndef HOT_EPL
#define HOT_EPL 4                         // consecutive entries per lane: 4 or 8
#endif
static constexpr int HOT_WIN = 64 * HOT_EPL;  // entries per window (one wavefront)
static constexpr int HOT_THREADS = 1024;  // 16 wavefronts share the hot x values
static constexpr int HOT_WAVES = HOT_THREADS / 64;
static constexpr int HOT_LDS = 160 * 1024;
static constexpr int HOT_HIST = 1024;     // sampled reference counts 0 .. 1022, last bucket = more

template <typename T>
static constexpr int hot_max_cols() {  // 16 320 fp64, 36 800 fp32 (a multiple of 64 entries; 512 B of row-start bitmaps)
  return (HOT_LDS - HOT_WAVES * HOT_WIN * (int) sizeof(T) - HOT_WAVES * HOT_WIN / 8) / (int) sizeof(T) / 64 * 64;
}

// ---------------------------------------------------------------------------------------------------------- inspect
// one entry in 16, at a position inside its group of 16 that changes from group to group (matrices with rows of 16
// sorted columns would otherwise show the sampler one column range only)
__global__ __launch_bounds__(256) void hot_sample_kernel(int64_t nnz, const int32_t* __restrict__ colind,
                                                         int32_t* __restrict__ cnt) {
  const int64_t g = (int64_t) blockIdx.x * 256 + threadIdx.x;
  const uint32_t hsh = (uint32_t) g * 2654435761u;
  const int64_t p = g * 16 + (hsh >> 28);
  if (p < nnz)
    atomicAdd(cnt + colind[p], 1);
}

// hist[c] = columns referenced c times by the sample (c clipped to HOT_HIST - 1), mass[c] = sum of their counts
__global__ __launch_bounds__(256) void hot_hist_kernel(int64_t n, const int32_t* __restrict__ cnt,
                                                       unsigned long long* __restrict__ hist,
                                                       unsigned long long* __restrict__ mass) {
  __shared__ unsigned int sh[HOT_HIST];
  __shared__ unsigned long long sm[HOT_HIST];
  for (int i = threadIdx.x; i < HOT_HIST; i += 256) {
    sh[i] = 0;
    sm[i] = 0;
  }
  __syncthreads();
  for (int64_t c = (int64_t) blockIdx.x * 256 + threadIdx.x; c < n; c += (int64_t) gridDim.x * 256) {
    const int v = cnt[c];
    if (v > 0) {
      const int b = v < HOT_HIST - 1 ? v : HOT_HIST - 1;
      atomicAdd(&sh[b], 1u);
      atomicAdd(&sm[b], (unsigned long long) v);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < HOT_HIST; i += 256)
    if (sh[i]) {
      atomicAdd(&hist[i], (unsigned long long) sh[i]);
      atomicAdd(&mass[i], sm[i]);
    }
}

__global__ __launch_bounds__(256) void hot_flag_cols_kernel(int64_t n, const int32_t* __restrict__ cnt, int thr,
                                                            int32_t* __restrict__ flag) {
  const int64_t c = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (c < n)
    flag[c] = cnt[c] >= thr;
}

// pos = exclusive scan of the column flags: colmap[c] = index in the hot list or -1, hot_cols[index] = c (ascending)
__global__ __launch_bounds__(256) void hot_colmap_kernel(int64_t n, const int32_t* __restrict__ pos,
                                                         int32_t* __restrict__ colmap, int32_t* __restrict__ hot_cols) {
  const int64_t c = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (c >= n)
    return;
  const int32_t i = pos[c];
  const bool hot = pos[c + 1] != i;
  colmap[c] = hot ? i : -1;
  if (hot)
    hot_cols[i] = (int32_t) c;
}

__global__ __launch_bounds__(256) void hot_flag_entries_kernel(int64_t nnz, const int32_t* __restrict__ colind,
                                                               const int32_t* __restrict__ colmap,
                                                               int32_t* __restrict__ flag) {
  const int64_t p = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (p < nnz)
    flag[p] = colmap[stream_load(colind + p)] >= 0;
}

// hotpos = exclusive scan of the entry flags (hotpos[nnz] = hot entries): a stable two-way split in one pass
template <typename T>
__global__ __launch_bounds__(256) void hot_split_kernel(int64_t nnz, const int32_t* __restrict__ colind,
                                                        const T* __restrict__ values, const int32_t* __restrict__ colmap,
                                                        const int32_t* __restrict__ hotpos, T* __restrict__ hot_val,
                                                        uint16_t* __restrict__ hot_col, int32_t* __restrict__ hot_src,
                                                        T* __restrict__ rest_val, int32_t* __restrict__ rest_col,
                                                        int32_t* __restrict__ rest_src) {
  const int64_t p = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (p >= nnz)
    return;
  const int32_t c = stream_load(colind + p);
  const T v = stream_load(values + p);
  const int32_t hp = hotpos[p];
  const int32_t k = colmap[c];
  if (k >= 0) {
    hot_val[hp] = v;
    hot_col[hp] = (uint16_t) k;
    hot_src[hp] = (int32_t) p;
  } else {
    const int64_t rp = p - hp;
    rest_val[rp] = v;
    rest_col[rp] = c;
    rest_src[rp] = (int32_t) p;
  }
}

// rest_rowptr[r] = entries of A_rest before row r; rowflag[r] = row r has hot entries
template <typename O>
__global__ __launch_bounds__(256) void hot_rowptr_kernel(int64_t m, const O* __restrict__ rowptr,
                                                         const int32_t* __restrict__ hotpos, O* __restrict__ rest_rowptr,
                                                         int32_t* __restrict__ rowflag) {
  const int64_t r = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (r > m)
    return;
  const O p = rowptr[r];
  const O hp = (O) hotpos[p];
  rest_rowptr[r] = p - hp;
  if (r < m)
    rowflag[r] = hotpos[rowptr[r + 1]] != (int32_t) hp;
}

// rowpos = exclusive scan of rowflag: A_hot over the rows that have hot entries only
template <typename O>
__global__ __launch_bounds__(256) void hot_compact_rows_kernel(int64_t m, const O* __restrict__ rowptr,
                                                               const int32_t* __restrict__ hotpos,
                                                               const int32_t* __restrict__ rowpos, int64_t n_hot,
                                                               O* __restrict__ hot_rowptr, int32_t* __restrict__ hot_rows) {
  const int64_t r = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (r > m)
    return;
  const int32_t i = rowpos[r];
  if (r == m) {
    hot_rowptr[i] = (O) n_hot;
    return;
  }
  if (rowpos[r + 1] != i) {
    hot_rowptr[i] = (O) hotpos[rowptr[r]];
    hot_rows[i] = (int32_t) r;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void hot_gather_values_kernel(int64_t cnt, const int32_t* __restrict__ src,
                                                                const T* __restrict__ values, T* __restrict__ out) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i < cnt) {
    const int32_t p = src[i];
    out[i] = p >= 0 ? values[p] : T(0);  // (pads of the slice-ordered stream have no source)
  }
}

// ---------------------------------------------------------------------------------------------------------- multiply
template <typename T>
__device__ __forceinline__ void hot_load4(const T* p, T (&out)[4]);
template <>
__device__ __forceinline__ void hot_load4<float>(const float* p, float (&out)[4]) {
  const f32x4 v = stream_load(reinterpret_cast<const f32x4*>(p));
  out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
template <>
__device__ __forceinline__ void hot_load4<double>(const double* p, double (&out)[4]) {
  const f64x2 a = stream_load(reinterpret_cast<const f64x2*>(p));
  const f64x2 b = stream_load(reinterpret_cast<const f64x2*>(p) + 1);
  out[0] = a.x; out[1] = a.y; out[2] = b.x; out[3] = b.y;
}
typedef unsigned short hot_u16x4 __attribute__((ext_vector_type(4)));

// Window w (HOT_WIN entries of A_hot) owns the rows of A_hot whose first entry lies in it (every row of A_hot has entries,
// so at most HOT_WIN of them); y[hot_rows[r]] += alpha * sum.  A row no longer than a window is summed entirely by its
// owner: it ends before the end of the NEXT window, so a wavefront stages the products of entries [w, w + 2) * HOT_WIN in
// its LDS strip (the second half is read again by the neighbouring wavefront, out of L2).  A longer row leaves a partial
// per window it covers (part_tail where it starts, part_head in the later ones), summed out of the same strip, for
// pb_hot_fixup_kernel.  spmv.hip: spmv_rowblock_kernel is this scheme at workgroup scope with x in global memory.
//
// A wavefront's windows are latency chains (window -> rows -> rows of y -> y), and a CU holds only the 16 wavefronts its
// LDS-resident x slice allows, so the chain is software-pipelined over four windows: the row numbers of window w + 3
// strides, the row offsets / row bounds / entries of window w + 2, the y values of window w + 1 and the arithmetic of window
// w (its entries arrive with its y values) are in flight together -- nothing an iteration loads is used before the next one; every load is unconditional (clamped
// indices, padded arrays) so that the waits count loads, not branches.
// DPP moves for the wave-wide segmented scan (no LDS round trips): CTRL 0x110 + n = row_shr:n inside the 16-lane rows,
// 0x142 / 0x143 = row_bcast:15 / row_bcast:31 (lane 15 of a row to the next row / lane 31 to the upper half; gfx9 family),
// 0x138 = wave_shr:1.  Lanes without a source (and rows outside ROWS) read 0.
template <int CTRL, int ROWS>
__device__ __forceinline__ unsigned hot_dpp(unsigned v) {
  return (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, CTRL, ROWS, 0xf, true);
}
template <int CTRL, int ROWS>
__device__ __forceinline__ float hot_dpp(float v) {
  return __uint_as_float(hot_dpp<CTRL, ROWS>(__float_as_uint(v)));
}
template <int CTRL, int ROWS>
__device__ __forceinline__ double hot_dpp(double v) {
  const unsigned long long b = (unsigned long long) __double_as_longlong(v);
  const unsigned lo = hot_dpp<CTRL, ROWS>((unsigned) b), hi = hot_dpp<CTRL, ROWS>((unsigned) (b >> 32));
  return __longlong_as_double((long long) (((unsigned long long) hi << 32) | lo));
}
// one step of the inclusive segmented scan over lanes: (v, f) of this lane absorbs (vp, fp) of the lanes before it
template <int CTRL, int ROWS, typename T>
__device__ __forceinline__ void hot_seg_step(T& v, unsigned& f, bool take) {
  const T vp = hot_dpp<CTRL, ROWS>(v);
  const unsigned fp = hot_dpp<CTRL, ROWS>(f);
  if (take) {
    if (!f)
      v += vp;
    f |= fp;
  }
}

// Window w = entries [w, w + 1) * HOT_WIN of A_hot, one wavefront, four consecutive entries per lane.  Every row of A_hot
// has entries, so a window sees at most HOT_WIN row starts.  The products stay in registers; the row starts of the window
// go into a bitmap (LDS, 256 bits); an exact segmented inclusive scan -- sums restart at every row start: no differences
// of prefix sums, a row's rounding depends on its own terms only -- runs over lanes with DPP moves; the scan values go to
// the wavefront's LDS strip, where the sum of a row is the value at its last entry:
//   rows that start and end inside the window      y[hot_rows[r]] += alpha * sum         (the lane that loaded the row's bounds)
//   the piece of a row that began in an earlier window   head[w] = value before the window's first row start
//   the piece of the last row if it runs on               tail[w] = value at the window's last entry
// and pb_hot_fixup_kernel adds tail + heads for the rows that cross a window boundary (listed at inspect).  A window
// without any row start (inside a long row: about half of the windows of a power-law graph) is a plain wave reduction.
//
// A wavefront's windows are latency chains (window -> rows -> rows of y -> y) and a CU holds only the 16 wavefronts its
// LDS-resident x slice allows, so the chain is software-pipelined over four windows: the row numbers of window w + 3
// strides, the row bounds of window w + 2, the y values and entries of window w + 1 and the arithmetic of window w are in
// flight together -- nothing a step loads is used before the next one; every load is unconditional (clamped indices,
// padded arrays) so that the waits count loads, not branches; the four stages rotate through four register sets by
// unrolling (copying a stage that was loaded in the same step would wait for it).
template <typename T, typename O>
struct hot_stage {
  int rb, re;      // rows of A_hot that start in the window: [rb, re)
  O a, e;          // first entry of row rb, one past the last entry of row re - 1
  O s0[2], s1[2];  // bounds of the rows rb + lane (+ 64)
  int hr[2];       // their rows of y
  T yv[2];         // ... and what y holds there (loaded one window ahead: nobody else writes these rows in this launch)
  T v[HOT_EPL];    // entries w * HOT_WIN + HOT_EPL * lane ...
  hot_u16x4 c[HOT_EPL / 4];
};

// The walk of one wavefront over its windows w, w + stride, ... <= last: y[hot_rows[r]] += alpha * sum (read-modify-write, y
// loaded one window ahead).
template <typename T, typename O>
__device__ __forceinline__ void hot_walk(int64_t w, const int64_t last, const int64_t stride, const int64_t nnz,
                                         const O* __restrict__ rowptr, const uint16_t* __restrict__ col,
                                         const T* __restrict__ val, const T* xs, T* strip, unsigned* bits, T* __restrict__ y,
                                         const T alpha, const int32_t* __restrict__ hot_rows, const int m_hot,
                                         const int32_t* __restrict__ win_row, T* __restrict__ part_head,
                                         T* __restrict__ part_tail, const int lane) {
  constexpr int BW = HOT_WIN / 32;
  if (w > last)
    return;
  auto clampw = [&](int64_t ww) { return ww < last ? ww : last; };
  auto load_rows = [&](int64_t ww, hot_stage<T, O>& st) {
    st.rb = win_row[ww];
    st.re = win_row[ww + 1];
  };
  auto load_bounds = [&](hot_stage<T, O>& st) {
    const int rb = st.rb, re = st.re;
    st.a = rowptr[rb];
    st.e = rowptr[re];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int r = rb + lane + 64 * k < re ? rb + lane + 64 * k : re;
      st.s0[k] = rowptr[r];
      st.s1[k] = rowptr[r < re ? r + 1 : re];
      st.hr[k] = hot_rows[r < m_hot ? r : m_hot - 1];
    }
  };
  auto load_entries = [&](int64_t ww, hot_stage<T, O>& st) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
      st.yv[k] = y[st.hr[k]];
    const int64_t base = ww * HOT_WIN + HOT_EPL * lane;  // (the arrays end in HOT_WIN zero entries)
#pragma unroll
    for (int q = 0; q < HOT_EPL / 4; ++q) {
      hot_load4<T>(val + base + 4 * q, reinterpret_cast<T (&)[4]>(st.v[4 * q]));
      st.c[q] = stream_load(reinterpret_cast<const hot_u16x4*>(col + base + 4 * q));
    }
  };
  const int lir = lane & 15;  // lane in its DPP row
  auto step = [&](hot_stage<T, O>& cur, hot_stage<T, O>& mid, hot_stage<T, O>& far, hot_stage<T, O>& next) {
    load_rows(clampw(w + 3 * stride), next);
    load_bounds(far);
    load_entries(clampw(w + stride), mid);
    const int rb = cur.rb, re = cur.re;
    const O wlo = (O) (w * HOT_WIN);
    const O whi = (O) ((w + 1) * HOT_WIN < nnz ? (w + 1) * HOT_WIN : nnz);
    T t[HOT_EPL];
#pragma unroll
    for (int j = 0; j < HOT_EPL; ++j) {
      t[j] = cur.v[j] * xs[cur.c[j / 4][j % 4]];  // (pads: 0 * xs[0], past the last row: never part of a sum)
    }
    if (rb == re) {  // no row starts here: one wave reduction, the piece of the row that covers the window
      T v = t[0];
#pragma unroll
      for (int j = 1; j < HOT_EPL; ++j)
        v += t[j];
      v += hot_dpp<0x111, 0xf>(v);
      v += hot_dpp<0x112, 0xf>(v);
      v += hot_dpp<0x114, 0xf>(v);
      v += hot_dpp<0x118, 0xf>(v);
      v += hot_dpp<0x142, 0xa>(v);
      v += hot_dpp<0x143, 0xc>(v);
      if (lane == 63)
        part_head[w] = v;
      return;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (rb + lane + 64 * k < re)
        atomicOr(&bits[(int) (cur.s0[k] - wlo) >> 5], 1u << ((int) (cur.s0[k] - wlo) & 31));
    for (int r = rb + 128 + lane; r < re; r += 64) {  // (more than 128 rows in one window: rare)
      const int q = (int) (rowptr[r] - wlo);
      atomicOr(&bits[q >> 5], 1u << (q & 31));
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the row starts are in the bitmap
    constexpr unsigned EMASK = (1u << HOT_EPL) - 1u;
    const unsigned nib = (bits[(lane * HOT_EPL) >> 5] >> ((lane * HOT_EPL) & 31)) & EMASK;
#pragma unroll
    for (int j = 1; j < HOT_EPL; ++j)
      t[j] = (nib & (1u << j)) ? t[j] : t[j - 1] + t[j];
    T v = t[HOT_EPL - 1];
    unsigned f = nib != 0u;
    hot_seg_step<0x111, 0xf>(v, f, lir >= 1);
    hot_seg_step<0x112, 0xf>(v, f, lir >= 2);
    hot_seg_step<0x114, 0xf>(v, f, lir >= 4);
    hot_seg_step<0x118, 0xf>(v, f, lir >= 8);
    hot_seg_step<0x142, 0xa>(v, f, (lane & 16) != 0);
    hot_seg_step<0x143, 0xc>(v, f, lane >= 32);
    const T c = hot_dpp<0x138, 0xf>(v);  // what the lanes before me carry into my first segment (lane 0: nothing)
    T* dst = strip + HOT_EPL * lane;
#pragma unroll
    for (int j = 0; j < HOT_EPL; ++j)
      dst[j] = (nib & ((2u << j) - 1u)) ? t[j] : t[j] + c;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // the scan values have landed in the strip
    if (lane < BW)
      bits[lane] = 0;  // (read by nobody before the next window's barrier)
    const bool runs_on = cur.e > whi;  // the last row that starts here ends in a later window
    if (lane == 0) {
      if (cur.a > wlo)
        part_head[w] = strip[(int) (cur.a - wlo) - 1];
      if (runs_on)
        part_tail[w] = strip[(int) (whi - wlo) - 1];
    }
    const int r_end = re - (runs_on ? 1 : 0);
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (rb + lane + 64 * k < r_end)
        y[cur.hr[k]] = cur.yv[k] + alpha * strip[(int) (cur.s1[k] - wlo) - 1];
    for (int r = rb + 128 + lane; r < r_end; r += 64)
      y[hot_rows[r]] += alpha * strip[(int) (rowptr[r + 1] - wlo) - 1];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // the strip is read out before the next window overwrites it
  };
  hot_stage<T, O> sa, sb, sc, sd;
  load_rows(w, sa);
  load_rows(clampw(w + stride), sb);
  load_rows(clampw(w + 2 * stride), sc);
  load_bounds(sa);
  load_bounds(sb);
  load_entries(w, sa);
  for (;;) {
    step(sa, sb, sc, sd);
    if ((w += stride) > last)
      break;
    step(sb, sc, sd, sa);
    if ((w += stride) > last)
      break;
    step(sc, sd, sa, sb);
    if ((w += stride) > last)
      break;
    step(sd, sa, sb, sc);
    if ((w += stride) > last)
      break;
  }
}

template <typename T, typename O>
__global__ __launch_bounds__(HOT_THREADS) void pb_hot_rows_kernel(int64_t nnz, int64_t nwin, const O* __restrict__ rowptr,
                                                                  const uint16_t* __restrict__ col,
                                                                  const T* __restrict__ val,
                                                                  const int32_t* __restrict__ hot_cols, int K,
                                                                  const T* __restrict__ x, T* __restrict__ y, T alpha,
                                                                  const int32_t* __restrict__ hot_rows, int m_hot,
                                                                  const int32_t* __restrict__ win_row,
                                                                  T* __restrict__ part_head, T* __restrict__ part_tail) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* xs = reinterpret_cast<T*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  T* strip = xs + hot_max_cols<T>() + wave * HOT_WIN;
  // row starts of the current window: one bit per entry (HOT_WIN / 32 words per wavefront, behind the strips)
  constexpr int BW = HOT_WIN / 32;
  unsigned* bits = reinterpret_cast<unsigned*>(xs + hot_max_cols<T>() + HOT_WAVES * HOT_WIN) + wave * BW;
  for (int i0 = 0; i0 < K; i0 += 4 * HOT_THREADS) {  // (four gathers in flight per thread)
    int idx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      idx[u] = i0 + u * HOT_THREADS + tid < K ? hot_cols[i0 + u * HOT_THREADS + tid] : 0;
    T xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      xv[u] = x[idx[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * HOT_THREADS + tid < K)
        xs[i0 + u * HOT_THREADS + tid] = xv[u];
  }
  if (lane < BW)
    bits[lane] = 0;
  __syncthreads();
  hot_walk<T, O>((int64_t) blockIdx.x * HOT_WAVES + wave, nwin - 1, (int64_t) gridDim.x * HOT_WAVES, nnz, rowptr, col, val,
                        xs, strip, bits, y, alpha, hot_rows, m_hot, win_row, part_head, part_tail, lane);
}

// Pre-summing expand of the tiles (round 4, second half): A re-ordered by x slice (natural column ranges of W columns, each
// slice's entries in row order, padded to whole windows), and the sum of every (row, slice) PAIR -- not every product -- is
// stored into the product stream of a tiled plan built over the pairs; that plan's reduce finishes y.  A power-law graph
// has far fewer pairs than entries (R-MAT scale 24, W = 20 480: 0.45), and that many fewer values make the round trip through
// HBM.  Every workgroup takes an equal range of windows and loads the x slice of every slice its range touches
// (slice_win[s] = first window of slice s).  The pair structure travels IN the entry stream and per window, not per pair
// (a first form with row offsets and a slot per pair in tables read 12 B per pair and ran 2.2 ms; this one 0.8):
//   col16 bit 15 = "a (row, slice) pair starts at this entry"; column 0x7fff = pad (exact zero, never a pair start);
//   per window 32 bytes: the product-stream slot of the k-th pair that starts in the window is sbase0 + k, with up to three
//   break points (kb, sbase) where the slots jump (the pairs of a window are consecutive in the stream until the row bin
//   changes); bit 0 of `info`: the window's last pair runs on into the next window; bit 1: more than three breaks -- the
//   slots then come from the per-pair table.
// A wavefront needs no row offsets, no bitmap, no LDS strip: products -> segmented scan over lanes (the sums restart at pair
// starts) -> every entry that ENDS a pair stores its value: into the slot of that pair, or -- when the pair started in an
// earlier window / runs on into the next -- into head[w] / tail[w] for pb_hot_fixup_kernel.  LDS holds only the x slice.
struct ps_windesc {
  int32_t sbase0, info, kb1, sb1, kb2, sb2, kb3, sb3;
};
static constexpr unsigned PS_PAD = 0x7fffu, PS_HEAD = 0x8000u;

template <typename T>
__global__ __launch_bounds__(HOT_THREADS) void pb_presum2_kernel(int64_t n, int W, int S, int64_t nwin,
                                                                 const int32_t* __restrict__ slice_win,
                                                                 const uint16_t* __restrict__ col, const T* __restrict__ val,
                                                                 const T* __restrict__ x, T* __restrict__ P,
                                                                 const ps_windesc* __restrict__ desc,
                                                                 const int32_t* __restrict__ ppos,
                                                                 const int32_t* __restrict__ win_pair0,
                                                                 T* __restrict__ part_head, T* __restrict__ part_tail) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* xs = reinterpret_cast<T*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lir = lane & 15;
  const int64_t per = (nwin + gridDim.x - 1) / gridDim.x;
  int64_t w0 = (int64_t) blockIdx.x * per;
  const int64_t w1 = w0 + per < nwin ? w0 + per : nwin;
  if (w0 >= w1)
    return;
  int lo = 0, hi = S;  // last slice starting at or before w0
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t) slice_win[mid] <= w0)
      lo = mid;
    else
      hi = mid;
  }
  struct stage_t {
    T v[4];
    hot_u16x4 c;
    ps_windesc d;
  };
  for (int s = lo; w0 < w1 && s < S; ++s) {
    const int64_t send = slice_win[s + 1];
    if (send <= w0)
      continue;  // an empty slice
    const int64_t seg_hi = w1 < send ? w1 : send;
    __syncthreads();  // everyone is done with the previous x slice
    const int64_t c0 = (int64_t) s * W;
    const int cw = (int) ((n - c0) < W ? (n - c0) : W);
    for (int i = tid; i < cw; i += HOT_THREADS)
      xs[i] = x[c0 + i];
    __syncthreads();
    const int64_t last = seg_hi - 1;
    int64_t w = w0 + wave;
    if (w <= last) {
      auto load = [&](int64_t ww, stage_t& st) {
        const int64_t wc = ww < last ? ww : last;
        const int64_t base = wc * HOT_WIN + 4 * lane;
        hot_load4<T>(val + base, st.v);
        st.c = stream_load(reinterpret_cast<const hot_u16x4*>(col + base));
        st.d = desc[wc];  // (wave-uniform address: scalar loads)
      };
      // (OVER: the window's slots come from the per-pair table -- a load whose result is used at once, i.e. a wait for
      // everything in flight, the next window's prefetch included; kept out of the common path as a whole: a wait on ONE
      // arm of an inner branch is hoisted by the compiler in front of every store)
      auto process = [&](const stage_t& cur, auto over_tag) {
        constexpr bool OVER = decltype(over_tag)::value;
        unsigned nib = 0;
        T t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned cj = cur.c[j];
          nib |= (cj >> 15) << j;
          const unsigned cc = cj & PS_PAD;
          t[j] = cc == PS_PAD ? T(0) : cur.v[j] * xs[cc];
        }
        if (__ballot(nib != 0u) == 0ull) {  // no pair starts here: the piece of the pair that covers the window
          T v = (t[0] + t[1]) + (t[2] + t[3]);
          v += hot_dpp<0x111, 0xf>(v);
          v += hot_dpp<0x112, 0xf>(v);
          v += hot_dpp<0x114, 0xf>(v);
          v += hot_dpp<0x118, 0xf>(v);
          v += hot_dpp<0x142, 0xa>(v);
          v += hot_dpp<0x143, 0xc>(v);
          if (lane == 63)
            part_head[w] = v;
          return;
        }
        // pairs started before each of my entries (exclusive over lanes, then inside the lane)
        int pc = __popc(nib);
        int incl = pc;
        {
          int u;
          u = (int) hot_dpp<0x111, 0xf>((unsigned) incl); incl += lir >= 1 ? u : 0;
          u = (int) hot_dpp<0x112, 0xf>((unsigned) incl); incl += lir >= 2 ? u : 0;
          u = (int) hot_dpp<0x114, 0xf>((unsigned) incl); incl += lir >= 4 ? u : 0;
          u = (int) hot_dpp<0x118, 0xf>((unsigned) incl); incl += lir >= 8 ? u : 0;
          u = (int) hot_dpp<0x142, 0xa>((unsigned) incl); incl += (lane & 16) ? u : 0;
          u = (int) hot_dpp<0x143, 0xc>((unsigned) incl); incl += lane >= 32 ? u : 0;
        }
        const int before = incl - pc;  // pair starts in the lanes before me
        // inclusive segmented sums
        t[1] = (nib & 2u) ? t[1] : t[0] + t[1];
        t[2] = (nib & 4u) ? t[2] : t[1] + t[2];
        t[3] = (nib & 8u) ? t[3] : t[2] + t[3];
        T v = t[3];
        unsigned f = nib != 0u;
        hot_seg_step<0x111, 0xf>(v, f, lir >= 1);
        hot_seg_step<0x112, 0xf>(v, f, lir >= 2);
        hot_seg_step<0x114, 0xf>(v, f, lir >= 4);
        hot_seg_step<0x118, 0xf>(v, f, lir >= 8);
        hot_seg_step<0x142, 0xa>(v, f, (lane & 16) != 0);
        hot_seg_step<0x143, 0xc>(v, f, lane >= 32);
        const T c = hot_dpp<0x138, 0xf>(v);  // what the lanes before me carry into my first segment (lane 0: nothing)
        // does the entry after my last one start a pair?  (the lane after me knows; after lane 63: the window descriptor)
        const unsigned nxt = __shfl_down(nib, 1, 64);
        const bool runs_on = (cur.d.info & 1) != 0;
        const unsigned heads_after = (nib >> 1) | ((lane == 63 ? (runs_on ? 0u : 1u) : (nxt & 1u)) << 3);  // bit j: entry j + 1 starts a pair
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (!((heads_after >> j) & 1u))
            continue;  // entry j does not end a pair
          const T sum = (nib & ((2u << j) - 1u)) ? t[j] : t[j] + c;
          const int k = before + __popc(nib & ((2u << j) - 1u)) - 1;  // the pair's number among those that start in this window
          if (k < 0) {
            part_head[w] = sum;  // started in an earlier window
          } else {
            int slot;
            if constexpr (OVER)
              slot = ppos[win_pair0[w] + k];
            else
              slot = k >= cur.d.kb3 ? cur.d.sb3 + (k - cur.d.kb3)
                                    : k >= cur.d.kb2 ? cur.d.sb2 + (k - cur.d.kb2)
                                                     : k >= cur.d.kb1 ? cur.d.sb1 + (k - cur.d.kb1) : cur.d.sbase0 + k;
            P[slot] = sum;
          }
        }
        if (runs_on && lane == 63)  // the open pair at the end of the window
          part_tail[w] = (nib & 15u) ? t[3] : t[3] + c;
      };
      auto run = [&](const stage_t& cur) {
        if (cur.d.info & 2)
          process(cur, std::true_type{});
        else
          process(cur, std::false_type{});
      };
      // four windows in flight per wavefront (the loads of window w + 3 strides are issued before window w is summed): a
      // CU holds only 16 wavefronts next to its x slice, and one window each does not cover the memory latency
      stage_t sa, sb, sc, sd;
      load(w, sa);
      load(w + HOT_WAVES, sb);
      load(w + 2 * HOT_WAVES, sc);
      for (;;) {
        load(w + 3 * HOT_WAVES, sd);
        run(sa);
        if ((w += HOT_WAVES) > last)
          break;
        load(w + 3 * HOT_WAVES, sa);
        run(sb);
        if ((w += HOT_WAVES) > last)
          break;
        load(w + 3 * HOT_WAVES, sb);
        run(sc);
        if ((w += HOT_WAVES) > last)
          break;
        load(w + 3 * HOT_WAVES, sc);
        run(sd);
        if ((w += HOT_WAVES) > last)
          break;
      }
    }
    w0 = seg_hi;
  }
}

// the rows of A_hot that cross a window boundary: y[hot_rows[r]] += alpha * (tail of the window the row starts in + heads of
// the windows it runs through), 16 lanes per row
template <typename T, typename O, bool PS>
__global__ __launch_bounds__(256) void pb_hot_fixup_kernel(int64_t n_cross, const int32_t* __restrict__ cross_rows,
                                                           const O* __restrict__ rowptr, const T* __restrict__ part_head,
                                                           const T* __restrict__ part_tail, T* __restrict__ y, T alpha,
                                                           const int32_t* __restrict__ hot_rows) {
  const int64_t i = ((int64_t) blockIdx.x * 256 + threadIdx.x) >> 4;
  const int lig = threadIdx.x & 15;
  T s = 0;
  int r = 0;
  int64_t w0 = 0;
  if (i < n_cross) {
    r = cross_rows[i];
    const int64_t p0 = (int64_t) rowptr[r], p1 = (int64_t) rowptr[r + 1];
    w0 = p0 / HOT_WIN;
    const int64_t w1 = (p1 - 1) / HOT_WIN;
    for (int64_t w = w0 + 1 + lig; w <= w1; w += 16)
      s += part_head[w];
  }
  s = group_sum_c<16>(s);
  if (i < n_cross && lig == 0) {
    if (PS)
      y[hot_rows[r]] = s + part_tail[w0];  // the pair's slot of the product stream
    else
      y[hot_rows[r]] += alpha * (s + part_tail[w0]);
  }
}

// inspect: the rows of A_hot whose entries lie in more than one window
template <typename O>
__global__ __launch_bounds__(256) void hot_cross_rows_kernel(int64_t m_hot, const O* __restrict__ rowptr,
                                                             int32_t* __restrict__ cross_rows,
                                                             unsigned long long* __restrict__ n_cross) {
  const int64_t r = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (r >= m_hot)
    return;
  const int64_t p0 = (int64_t) rowptr[r], p1 = (int64_t) rowptr[r + 1];
  if (p0 / HOT_WIN != (p1 - 1) / HOT_WIN)
    cross_rows[atomicAdd(n_cross, 1ull)] = (int32_t) r;
}

template <typename T, typename O>
static int hot_launch(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* alpha_p, const void* x, void* y) {
  spblas_gfx950_plan_s* hp = pl->hot_plan;
  hipStream_t s = h->stream;
  const T alpha = *static_cast<const T*>(alpha_p);
  const int cus = h->num_cus > 0 ? h->num_cus : 256;
  const int64_t grid = cdiv(hp->nwin, HOT_WAVES) < cus ? cdiv(hp->nwin, HOT_WAVES) : cus;
  const O* rowptr = static_cast<const O*>(hp->rowptr);
  T* head = static_cast<T*>(pl->hot_part);
  T* tail = head + hp->nwin;
  hp->last_stream = s;
  hp->used = true;
  hipLaunchKernelGGL((pb_hot_rows_kernel<T, O>), dim3((unsigned) grid), dim3(HOT_THREADS), HOT_LDS, s, hp->nnz, hp->nwin, rowptr,
                     pl->hot_col, static_cast<const T*>(pl->hot_val), pl->hot_cols, pl->hot_k, static_cast<const T*>(x),
                     static_cast<T*>(y), alpha, pl->hot_rows, (int) pl->hot_m, hp->win_row, head, tail);
  if (pl->hot_ncross > 0)
    hipLaunchKernelGGL((pb_hot_fixup_kernel<T, O, false>), dim3((unsigned) cdiv(pl->hot_ncross * 16, 256)), dim3(256), 0, s,
                       pl->hot_ncross, pl->hot_cross, rowptr, head, tail, static_cast<T*>(y), alpha, pl->hot_rows);
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

// pre-summing plan: products of every (row, slice) pair into the pair plan's product stream
template <typename T>
static int presum_launch(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* x) {
  spblas_gfx950_plan_s *hp = pl->hot_plan, *rp = pl->rest_plan;
  hipStream_t s = h->stream;
  const int cus = h->num_cus > 0 ? h->num_cus : 256;
  const int64_t grid = hp->nwin < cus ? hp->nwin : cus;
  T* head = static_cast<T*>(pl->hot_part);
  T* tail = head + hp->nwin;
  const int32_t* rowptr = static_cast<const int32_t*>(pl->hot_rowptr);
  hp->last_stream = rp->last_stream = pl->last_stream = s;
  hp->used = rp->used = pl->used = true;
  hipLaunchKernelGGL((pb_presum2_kernel<T>), dim3((unsigned) grid), dim3(HOT_THREADS), (size_t) pl->ps_W * sizeof(T), s, pl->n, pl->ps_W,
                     pl->ps_S, hp->nwin, pl->ps_slice_win, pl->hot_col, static_cast<const T*>(pl->hot_val), static_cast<const T*>(x),
                     static_cast<T*>(rp->s_products), static_cast<const ps_windesc*>(pl->ps_desc), pl->hot_rows, pl->ps_win_pair0,
                     head, tail);
  if (pl->hot_ncross > 0)
    hipLaunchKernelGGL((pb_hot_fixup_kernel<T, int32_t, true>), dim3((unsigned) cdiv(pl->hot_ncross * 16, 256)), dim3(256), 0, s,
                       pl->hot_ncross, pl->hot_cross, rowptr, head, tail, static_cast<T*>(rp->s_products), T(1), pl->hot_rows);
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spmv_presum_expand(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* x) {
  return pl->value_type == SPBLAS_GFX950_F32 ? presum_launch<float>(h, pl, x) : presum_launch<double>(h, pl, x);
}

// y += alpha * A_hot x (the second half of a multiply with a split plan: the tiled plan of A_rest has written every row)
int spmv_hot_rows(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* alpha, const void* x, void* y) {
  const bool o32 = pl->offset_type == SPBLAS_GFX950_I32;
  if (pl->value_type == SPBLAS_GFX950_F32)
    return o32 ? hot_launch<float, int32_t>(h, pl, alpha, x, y) : hot_launch<float, int64_t>(h, pl, alpha, x, y);
  return o32 ? hot_launch<double, int32_t>(h, pl, alpha, x, y) : hot_launch<double, int64_t>(h, pl, alpha, x, y);
}

int spmv_sliced_reduce_rows(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* alpha, const void* beta,
                            void* y, int64_t row_begin, int64_t row_end, void* const* peers, int n_peers,
                            int64_t peer_off);

int spmv_hot_exec(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* alpha, const void* x, const void* beta,
                  void* y) {
  pl->last_stream = h->stream;
  pl->used = true;
  if (pl->ps_mode) {  // pre-summed products into the pair plan's stream, then that plan's reduce finishes y
    const int rc_p = spmv_presum_expand(h, pl, x);
    return rc_p ? rc_p : spmv_sliced_reduce_rows(h, pl->rest_plan, alpha, beta, y, 0, pl->m, nullptr, 0, 0);
  }
  pl->rest_plan->nt_products = pl->nt_products;
  const int rc = spmv_sliced_exec(h, pl->rest_plan, alpha, x, beta, y);
  if (rc)
    return rc;
  return spmv_hot_rows(h, pl, alpha, x, y);
}

// ---------------------------------------------------------------------------------------------------------- build
void spmv_hot_free(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl) {
  hipStream_t s = h->stream;
  if (pl->hot_plan) {
    spmv_plan_release(h, pl->hot_plan);
    pl->hot_plan = nullptr;
  }
  if (pl->rest_plan) {
    spmv_plan_release(h, pl->rest_plan);
    pl->rest_plan = nullptr;
  }
  dev_free(pl->hot_cols, s);
  dev_free(pl->hot_col, s);
  dev_free(pl->hot_val, s);
  dev_free(pl->hot_src, s);
  dev_free(pl->hot_rowptr, s);
  dev_free(pl->hot_rows, s);
  dev_free(pl->hot_part, s);
  dev_free(pl->hot_cross, s);
  dev_free(pl->ps_slice_win, s);
  dev_free(pl->ps_desc, s);
  dev_free(pl->ps_win_pair0, s);
  pl->ps_desc = nullptr;
  pl->ps_win_pair0 = nullptr;
  dev_free(pl->ps_ap_rowptr, s);
  dev_free(pl->ps_ap_col, s);
  pl->ps_slice_win = pl->ps_ap_rowptr = pl->ps_ap_col = nullptr;
  pl->ps_mode = pl->ps_W = pl->ps_S = 0;
  pl->hot_part = nullptr;
  pl->hot_cross = nullptr;
  pl->hot_ncross = 0;
  dev_free(pl->rest_rowptr, s);
  dev_free(pl->rest_col, s);
  dev_free(pl->rest_val, s);
  dev_free(pl->rest_src, s);
  pl->hot_cols = nullptr;
  pl->hot_col = nullptr;
  pl->hot_val = pl->rest_val = nullptr;
  pl->hot_src = pl->rest_src = pl->rest_col = pl->hot_rows = nullptr;
  pl->hot_rowptr = pl->rest_rowptr = nullptr;
  pl->hot_k = 0;
  pl->hot_nnz = 0;
}

template <typename T, typename O>
static int hot_build_typed(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values_p, bool auto_mode) {
  hipStream_t s = h->stream;
  const int64_t m = pl->m, n = pl->n, nnz = pl->nnz;
  const int32_t* colind = pl->colind;
  const T* values = static_cast<const T*>(values_p);
  const O* rowptr = static_cast<const O*>(pl->rowptr);
  int rc;
  readback_scope rb_scope(h);
  // temporaries of the inspect, released on every exit path
  struct guard_t {
    hipStream_t s;
    std::vector<void*> p;
    ~guard_t() {
      for (void* q : p)
        dev_free(q, s);
    }
    int alloc(void** out, size_t bytes) {
      const int rc_a = dev_alloc(out, bytes, s);
      if (rc_a == SPBLAS_GFX950_STATUS_SUCCESS && *out)
        p.push_back(*out);
      return rc_a;
    }
  } g{s, {}};
  int32_t *cnt = nullptr, *pos = nullptr;
  unsigned long long* hist = nullptr;
  long long* partials = nullptr;
  const int64_t scan_len = nnz > n ? nnz : n;
  if ((rc = g.alloc((void**) &cnt, (size_t) n * 4)) || (rc = g.alloc((void**) &pos, (size_t) (scan_len + 1) * 4)) ||
      (rc = g.alloc((void**) &hist, (size_t) 2 * HOT_HIST * sizeof(unsigned long long))) ||
      (rc = g.alloc((void**) &partials, (size_t) (cdiv(scan_len, 2048) + 2) * sizeof(long long))))
    return rc;
  SPB_HIP(hipMemsetAsync(cnt, 0, (size_t) n * 4, s));
  SPB_HIP(hipMemsetAsync(hist, 0, (size_t) 2 * HOT_HIST * sizeof(unsigned long long), s));
  const int64_t nsamp = cdiv(nnz, 16);
  hipLaunchKernelGGL(hot_sample_kernel, dim3((unsigned) cdiv(nsamp, 256)), dim3(256), 0, s, nnz, colind, cnt);
  hipLaunchKernelGGL(hot_hist_kernel, dim3((unsigned) (cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048)), dim3(256), 0, s, n, cnt, hist,
                     hist + HOT_HIST);
  std::vector<unsigned long long> h_hist((size_t) 2 * HOT_HIST);
  if ((rc = readback_add(h, h_hist.data(), hist, h_hist.size() * sizeof(unsigned long long))) || (rc = readback_flush(h)))
    return rc;
  SPB_HIP(hipGetLastError());
  // the smallest count threshold (>= 2 sampled references) whose columns still fit the LDS
  const int kmax = hot_max_cols<T>();
  unsigned long long cols = 0, mass = 0, sampled = 0;
  for (int c = 1; c < HOT_HIST; ++c)
    sampled += h_hist[(size_t) HOT_HIST + c];
  int thr = HOT_HIST;  // nothing
  for (int c = HOT_HIST - 1; c >= 2; --c) {
    if (cols + h_hist[(size_t) c] > (unsigned long long) kmax)
      break;
    cols += h_hist[(size_t) c];
    mass += h_hist[(size_t) HOT_HIST + c];
    thr = c;
  }
  // (a second or third split takes less: the columns get colder; SPBLAS_GFX950_PB_HOT_MIN_PCT2 is its threshold)
  const int min_pct = pl->is_child > 0 ? hot_env("SPBLAS_GFX950_PB_HOT_MIN_PCT2", 8) : hot_env("SPBLAS_GFX950_PB_HOT_MIN_PCT", 15);
  if (cols == 0 || sampled == 0 || mass * 100 < sampled * (unsigned long long) min_pct)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;  // no set of columns that small carries enough of the matrix
  // hot list (ascending columns) and the column -> hot index map
  int32_t* colmap = nullptr;
  if ((rc = g.alloc((void**) &colmap, (size_t) n * 4)) || (rc = dev_alloc((void**) &pl->hot_cols, (size_t) cols * 4, s)))
    return rc;
  hipLaunchKernelGGL(hot_flag_cols_kernel, dim3((unsigned) cdiv(n, 256)), dim3(256), 0, s, n, cnt, thr, pos);
  (void) scan_counts_i32(s, n, pos, partials);
  hipLaunchKernelGGL(hot_colmap_kernel, dim3((unsigned) cdiv(n, 256)), dim3(256), 0, s, n, pos, colmap, pl->hot_cols);
  pl->hot_k = (int) cols;
  // stable split of the entries
  hipLaunchKernelGGL(hot_flag_entries_kernel, dim3((unsigned) cdiv(nnz, 256)), dim3(256), 0, s, nnz, colind, colmap, pos);
  long long* total_dev = scan_counts_i32(s, nnz, pos, partials);
  long long n_hot = 0;
  if ((rc = readback_add(h, &n_hot, total_dev, sizeof(n_hot))) || (rc = readback_flush(h)))
    return rc;
  SPB_HIP(hipGetLastError());
  const int64_t n_rest = nnz - (int64_t) n_hot;
  if (n_hot * 100 < nnz * (long long) min_pct || n_rest < 1)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;  // (the sample promised more than the matrix holds)
  pl->hot_nnz = (int64_t) n_hot;
  // (the multiply loads whole windows: the arrays end in zero entries up to the end of the last window)
  const size_t hot_pad = (size_t) (n_hot / HOT_WIN + 1) * HOT_WIN + HOT_WIN;
  if ((rc = dev_alloc(&pl->hot_val, hot_pad * sizeof(T), s)) ||
      (rc = dev_alloc((void**) &pl->hot_col, hot_pad * 2, s)) ||
      (rc = dev_alloc((void**) &pl->hot_src, (size_t) n_hot * 4, s)) ||
      (rc = dev_alloc(&pl->rest_rowptr, (size_t) (m + 1) * sizeof(O), s)) ||
      (rc = dev_alloc(&pl->rest_val, (size_t) n_rest * sizeof(T), s)) ||
      (rc = dev_alloc((void**) &pl->rest_col, (size_t) n_rest * 4, s)) ||
      (rc = dev_alloc((void**) &pl->rest_src, (size_t) n_rest * 4, s)))
    return rc;
  SPB_HIP(hipMemsetAsync(static_cast<T*>(pl->hot_val) + n_hot, 0, (hot_pad - (size_t) n_hot) * sizeof(T), s));
  SPB_HIP(hipMemsetAsync(pl->hot_col + n_hot, 0, (hot_pad - (size_t) n_hot) * 2, s));
  hipLaunchKernelGGL((hot_split_kernel<T>), dim3((unsigned) cdiv(nnz, 256)), dim3(256), 0, s, nnz, colind, values, colmap, pos,
                     static_cast<T*>(pl->hot_val), pl->hot_col, pl->hot_src, static_cast<T*>(pl->rest_val), pl->rest_col,
                     pl->rest_src);
  int32_t *rowflag = nullptr;
  long long* rpartials = nullptr;
  if ((rc = g.alloc((void**) &rowflag, (size_t) (m + 1) * 4)) ||
      (rc = g.alloc((void**) &rpartials, (size_t) (cdiv(m, 2048) + 2) * sizeof(long long))))
    return rc;
  hipLaunchKernelGGL((hot_rowptr_kernel<O>), dim3((unsigned) cdiv(m + 1, 256)), dim3(256), 0, s, m, rowptr, pos,
                     static_cast<O*>(pl->rest_rowptr), rowflag);
  long long* mc_dev = scan_counts_i32(s, m, rowflag, rpartials);
  long long m_hot = 0;
  if ((rc = readback_add(h, &m_hot, mc_dev, sizeof(m_hot))) || (rc = readback_flush(h)))
    return rc;
  SPB_HIP(hipGetLastError());
  if (m_hot < 1)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  if ((rc = dev_alloc(&pl->hot_rowptr, (size_t) (m_hot + 1) * sizeof(O), s)) ||
      (rc = dev_alloc((void**) &pl->hot_rows, (size_t) m_hot * 4, s)))
    return rc;
  hipLaunchKernelGGL((hot_compact_rows_kernel<O>), dim3((unsigned) cdiv(m + 1, 256)), dim3(256), 0, s, m, rowptr, pos, rowflag,
                     (int64_t) n_hot, static_cast<O*>(pl->hot_rowptr), pl->hot_rows);
  SPB_HIP(hipGetLastError());
  // A_hot: the nnz-window row partition (windows of HOT_WIN entries), long-row list and partials of the row-block plan
  auto* hp = new (std::nothrow) spblas_gfx950_plan_s();
  if (!hp)
    return SPBLAS_GFX950_STATUS_ALLOC_FAILED;
  pl->hot_plan = hp;
  hp->m = (int64_t) m_hot;
  hp->n = pl->hot_k;
  hp->nnz = (int64_t) n_hot;
  hp->rowptr = pl->hot_rowptr;
  hp->colind = nullptr;
  hp->offset_type = pl->offset_type;
  hp->value_type = pl->value_type;
  hp->win_req = HOT_WIN;
  hp->is_child = -1;
  if ((rc = spmv_plan_structures(h, hp, SPBLAS_GFX950_SPMV_ROWBLOCK)))
    return rc;
  {
    // the rows that cross a window boundary (a row of n entries crosses with probability ~ n / HOT_WIN) and the two
    // partial sums per window they are put together from
    unsigned long long* n_cross_dev = nullptr;
    if ((rc = g.alloc((void**) &n_cross_dev, sizeof(unsigned long long))) ||
        (rc = dev_alloc((void**) &pl->hot_cross, (size_t) (hp->nwin + 1) * 4, s)) ||
        (rc = dev_alloc(&pl->hot_part, (size_t) 2 * hp->nwin * sizeof(T), s)))
      return rc;
    SPB_HIP(hipMemsetAsync(n_cross_dev, 0, sizeof(unsigned long long), s));
    SPB_HIP(hipMemsetAsync(pl->hot_part, 0, (size_t) 2 * hp->nwin * sizeof(T), s));
    hipLaunchKernelGGL((hot_cross_rows_kernel<O>), dim3((unsigned) cdiv(m_hot, 256)), dim3(256), 0, s, (int64_t) m_hot,
                       static_cast<const O*>(pl->hot_rowptr), pl->hot_cross, n_cross_dev);
    unsigned long long n_cross = 0;
    if ((rc = readback_add(h, &n_cross, n_cross_dev, sizeof(n_cross))) || (rc = readback_flush(h)))
      return rc;
    SPB_HIP(hipGetLastError());
    pl->hot_ncross = (int64_t) n_cross;  // (at most one row crosses each boundary: <= nwin - 1)
  }
  // A_rest: an ordinary CSR matrix for the tiled plan
  auto* rp = new (std::nothrow) spblas_gfx950_plan_s();
  if (!rp)
    return SPBLAS_GFX950_STATUS_ALLOC_FAILED;
  pl->rest_plan = rp;
  rp->m = m;
  rp->n = n;
  rp->nnz = n_rest;
  rp->rowptr = pl->rest_rowptr;
  rp->colind = pl->rest_col;
  rp->offset_type = pl->offset_type;
  rp->value_type = pl->value_type;
  rp->is_child = pl->is_child + 1;
  if ((rc = spmv_plan_structures(h, rp, SPBLAS_GFX950_SPMV_ROWBLOCK)))
    return rc;
  if ((rc = spmv_sliced_build(h, rp, pl->rest_val, auto_mode)))
    return rc;
  rp->alg = SPBLAS_GFX950_SPMV_SLICED;
  rp->nt_products = pl->nt_products;
  pl->s_uncertain = rp->s_uncertain;
  pl->s_placed = rp->s_placed;
  pl->values_ptr = values_p;
  pl->device_bytes += hp->device_bytes + rp->device_bytes + (size_t) cols * 4 + (size_t) n_hot * (sizeof(T) + 6) +
                      (size_t) n_rest * (sizeof(T) + 8) + (size_t) (m + 1) * sizeof(O) + (size_t) (m_hot + 1) * (sizeof(O) + 4);
  pl->hot_m = (int64_t) m_hot;
  pl->device_bytes += (size_t) (hp->nwin + 1) * 4 + (size_t) 2 * hp->nwin * sizeof(T);
  SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pb_hot_rows_kernel<T, O>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, HOT_LDS));
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

// ---------------------------------------------------------------------------------------------------------- pre-summing plan
int spblas_gfx950_csr_transpose_internal(spblas_gfx950_handle_t handle, int64_t m, int64_t n, int64_t nnz, const int32_t* rowptr,
                                         const int32_t* colind, const void* values, int32_t* t_rowptr, int32_t* t_colind,
                                         void* t_values, int value_type);  // = spblas_gfx950_csr_transpose (transpose.hip)

template <typename O>
__global__ __launch_bounds__(256) void ps_rowptr32_kernel(int64_t m, const O* __restrict__ rowptr, int32_t* __restrict__ out) {
  const int64_t r = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (r <= m)
    out[r] = (int32_t) rowptr[r];
}
// slice of every entry and its position as a payload the transpose carries along (32 bits, moved as a float)
__global__ __launch_bounds__(256) void ps_sid_kernel(int64_t nnz, const int32_t* __restrict__ colind, int W,
                                                     int32_t* __restrict__ sid, int32_t* __restrict__ pos) {
  const int64_t p = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (p < nnz) {
    sid[p] = colind[p] / W;
    pos[p] = (int32_t) p;
  }
}
__global__ __launch_bounds__(256) void ps_iota_kernel(int64_t cnt, int32_t* __restrict__ out) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i < cnt)
    out[i] = (int32_t) i;
}
// entries in (slice, row) order -> the padded stream: entry i of slice s goes to i + shift[s]
template <typename T>
__global__ __launch_bounds__(256) void ps_place_kernel(int64_t nnz, int S, const int32_t* __restrict__ t_rowptr,
                                                       const int32_t* __restrict__ shift, const int32_t* __restrict__ t_rows,
                                                       const int32_t* __restrict__ t_pos, int W,
                                                       const int32_t* __restrict__ colind, const T* __restrict__ values,
                                                       int32_t* __restrict__ prow, int32_t* __restrict__ psrc,
                                                       T* __restrict__ val, uint16_t* __restrict__ col) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= nnz)
    return;
  int lo = 0, hi = S;  // last slice starting at or before i
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t) t_rowptr[mid] <= i)
      lo = mid;
    else
      hi = mid;
  }
  const int64_t j = i + shift[lo];
  const int32_t p = t_pos[i];
  prow[j] = t_rows[i];
  psrc[j] = p;
  val[j] = values[p];
  col[j] = (uint16_t) (colind[p] - lo * W);
}
// a pair starts where the row changes (pads, row -1, stay with the pair before them) ...
__global__ __launch_bounds__(256) void ps_flags_kernel(int64_t cnt, const int32_t* __restrict__ prow, int32_t* __restrict__ flag) {
  const int64_t j = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (j < cnt) {
    const int32_t r = prow[j];
    flag[j] = r >= 0 && (j == 0 || prow[j - 1] != r);
  }
}
// ... and at the first entry of every non-empty slice
__global__ __launch_bounds__(256) void ps_slice_flags_kernel(int S, const int32_t* __restrict__ slice_win,
                                                             int32_t* __restrict__ flag) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s < S && slice_win[s + 1] > slice_win[s])
    flag[(int64_t) slice_win[s] * HOT_WIN] = 1;
}
// pidx = exclusive scan of the flags: first entry and row of every pair
__global__ __launch_bounds__(256) void ps_pairs_kernel(int64_t cnt, const int32_t* __restrict__ pidx,
                                                       const int32_t* __restrict__ prow, int32_t* __restrict__ pair_rowptr,
                                                       int32_t* __restrict__ pair_row) {
  const int64_t j = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (j > cnt)
    return;
  const int32_t i = pidx[j];
  if (j == cnt) {
    pair_rowptr[i] = (int32_t) cnt;
    return;
  }
  if (pidx[j + 1] != i) {
    pair_rowptr[i] = (int32_t) j;
    pair_row[i] = prow[j];
  }
}
__global__ __launch_bounds__(256) void ps_slice_pairs_kernel(int S, const int32_t* __restrict__ slice_win,
                                                             const int32_t* __restrict__ pidx, int32_t* __restrict__ rowptr2) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s <= S)
    rowptr2[s] = pidx[(int64_t) slice_win[s] * HOT_WIN];
}
// compact row (the pair plan's row numbering when it took the empty rows out) of every original row
__global__ __launch_bounds__(256) void ps_rowinv_kernel(int64_t s_m, const int32_t* __restrict__ nzrow, int32_t* __restrict__ inv) {
  const int64_t c = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (c < s_m)
    inv[nzrow[c]] = (int32_t) c;
}
// bin of every pair (slice-major numbering); cnt[bin * S + slice] = pairs of the tile, first[bin * S + slice] = its first pair
__global__ __launch_bounds__(256) void ps_tiles_kernel(int64_t m_pairs, int S, int H, int64_t NB, const int32_t* __restrict__ pair_row,
                                                       const int32_t* __restrict__ rowptr2, const int32_t* __restrict__ inv,
                                                       const int32_t* __restrict__ binrow, int32_t* __restrict__ pbin,
                                                       int32_t* __restrict__ cnt, int32_t* __restrict__ first) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= m_pairs)
    return;
  auto bin_of = [&](int64_t k) {
    const int32_t r = pair_row[k];
    const int32_t c = inv ? inv[r] : r;
    if (!binrow)
      return (int32_t) (c / H);
    int64_t lo = 0, hi = NB;  // last bin whose first row is <= c
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (binrow[mid] <= c)
        lo = mid;
      else
        hi = mid;
    }
    return (int32_t) lo;
  };
  int lo = 0, hi = S;  // slice of pair i: last slice whose first pair is <= i
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t) rowptr2[mid] <= i)
      lo = mid;
    else
      hi = mid;
  }
  const int32_t b = bin_of(i);
  pbin[i] = b;
  const int64_t t = (int64_t) b * S + lo;
  atomicAdd(cnt + t, 1);
  if (i == (int64_t) rowptr2[lo] || bin_of(i - 1) != b)
    first[t] = (int32_t) i;
}
// base = exclusive scan of cnt over (bin, slice): slot of pair i in the product stream, and its row word
__global__ __launch_bounds__(256) void ps_slots_kernel(int64_t m_pairs, int S, int H, int BLK, const int32_t* __restrict__ pair_row,
                                                       const int32_t* __restrict__ rowptr2, const int32_t* __restrict__ inv,
                                                       const int32_t* __restrict__ binrow, const int32_t* __restrict__ pbin,
                                                       const int32_t* __restrict__ base, const int32_t* __restrict__ first,
                                                       const int32_t* __restrict__ binblk, int32_t* __restrict__ ppos,
                                                       uint16_t* __restrict__ s_lrow) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= m_pairs)
    return;
  int lo = 0, hi = S;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t) rowptr2[mid] <= i)
      lo = mid;
    else
      hi = mid;
  }
  const int32_t b = pbin[i];
  const int64_t t = (int64_t) b * S + lo;
  const int64_t pp = (int64_t) binblk[b] * BLK + (base[t] - base[(int64_t) b * S]) + (i - first[t]);
  const int32_t r = pair_row[i];
  const int32_t c = inv ? inv[r] : r;
  ppos[i] = (int32_t) pp;
  s_lrow[pp] = (uint16_t) (c - (binrow ? binrow[b] : b * H));
}

// pair starts into bit 15 of the 16-bit columns (pads: 0x7fff, no start)
__global__ __launch_bounds__(256) void ps_headbits_kernel(int64_t cnt, const int32_t* __restrict__ pidx, uint16_t* __restrict__ col) {
  const int64_t j = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (j >= cnt)
    return;
  const unsigned c = col[j];
  col[j] = (uint16_t) ((c == 0xffffu ? PS_PAD : c) | (pidx[j + 1] != pidx[j] ? PS_HEAD : 0u));
}
// per window: first pair that starts in it, the slots of its pairs as a base + up to three break points, "runs on"
__global__ __launch_bounds__(256) void ps_windesc_kernel(int64_t nwin, int64_t np, const int32_t* __restrict__ pidx,
                                                         const int32_t* __restrict__ ppos, ps_windesc* __restrict__ desc,
                                                         int32_t* __restrict__ win_pair0, unsigned long long* __restrict__ n_over) {
  const int64_t w = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (w >= nwin)
    return;
  const int64_t e0 = w * HOT_WIN, e1 = e0 + HOT_WIN;
  const int p0 = pidx[e0], p1 = pidx[e1];
  win_pair0[w] = p0;
  ps_windesc d = {0, 0, 0x7fffffff, 0, 0x7fffffff, 0, 0x7fffffff, 0};
  // the last entry's pair runs on iff the next window starts without a pair start (slices start with one; pads never do,
  // but pads only fill a slice's LAST window, and the window after it starts a slice or does not exist)
  if (e1 < np && pidx[e1 + 1] == pidx[e1])
    d.info |= 1;
  if (p1 > p0) {
    d.sbase0 = ppos[p0];
    int nb = 0;
    for (int k = 1; k < p1 - p0; ++k)
      if (ppos[p0 + k] != ppos[p0 + k - 1] + 1) {
        ++nb;
        if (nb == 1) {
          d.kb1 = k;
          d.sb1 = ppos[p0 + k];
        } else if (nb == 2) {
          d.kb2 = k;
          d.sb2 = ppos[p0 + k];
        } else if (nb == 3) {
          d.kb3 = k;
          d.sb3 = ppos[p0 + k];
        }
      }
    if (nb > 3) {
      d.info |= 2;
      atomicAdd(n_over, 1ull);
    }
  }
  desc[w] = d;
}

template <typename T, typename O>
static int presum_build_typed(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values_p) {
  hipStream_t s = h->stream;
  const int64_t m = pl->m, n = pl->n, nnz = pl->nnz;
  // slice width: the whole LDS of a CU holds x (the kernel needs nothing else there); bit 15 of a column is the pair start
  const int W = (int) std::min<int64_t>(HOT_LDS / (int64_t) sizeof(T), 32704) / 64 * 64;  // 20 480 fp64, 32 704 fp32
  const int64_t S64 = cdiv(n, W);
  if (S64 > 8192 || nnz > INT32_MAX - 8 - 8192 * (int64_t) HOT_WIN || m >= INT32_MAX)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  const int S = (int) S64;
  int rc;
  readback_scope rb_scope(h);
  struct guard_t {
    hipStream_t s;
    std::vector<void*> p;
    ~guard_t() {
      for (void* q : p)
        dev_free(q, s);
    }
    int alloc(void** out, size_t bytes) {
      const int rc_a = dev_alloc(out, bytes, s);
      if (rc_a == SPBLAS_GFX950_STATUS_SUCCESS && *out)
        p.push_back(*out);
      return rc_a;
    }
  } g{s, {}};
  // 1. entries by slice, rows in order inside a slice: a stable counting sort = the device transpose of (row, slice)
  int32_t *rowptr32 = nullptr, *sid = nullptr, *pos = nullptr, *t_rowptr = nullptr, *t_rows = nullptr, *t_pos = nullptr;
  if ((rc = g.alloc((void**) &sid, (size_t) nnz * 4)) || (rc = g.alloc((void**) &pos, (size_t) nnz * 4)) ||
      (rc = g.alloc((void**) &t_rowptr, (size_t) (S + 1) * 4)) || (rc = g.alloc((void**) &t_rows, (size_t) nnz * 4)) ||
      (rc = g.alloc((void**) &t_pos, (size_t) nnz * 4)))
    return rc;
  if (sizeof(O) == 4) {
    rowptr32 = const_cast<int32_t*>(static_cast<const int32_t*>(pl->rowptr));
  } else {
    if ((rc = g.alloc((void**) &rowptr32, (size_t) (m + 1) * 4)))
      return rc;
    hipLaunchKernelGGL((ps_rowptr32_kernel<O>), dim3((unsigned) cdiv(m + 1, 256)), dim3(256), 0, s, m,
                       static_cast<const O*>(pl->rowptr), rowptr32);
  }
  hipLaunchKernelGGL(ps_sid_kernel, dim3((unsigned) cdiv(nnz, 256)), dim3(256), 0, s, nnz, pl->colind, W, sid, pos);
  SPB_HIP(hipGetLastError());
  if ((rc = spblas_gfx950_csr_transpose_internal(h, m, S, nnz, rowptr32, sid, pos, t_rowptr, t_rows, t_pos, SPBLAS_GFX950_F32)))
    return rc;
  std::vector<int32_t> h_trp((size_t) S + 1);
  SPB_HIP(hipMemcpyAsync(h_trp.data(), t_rowptr, (size_t) (S + 1) * 4, hipMemcpyDeviceToHost, s));
  SPB_HIP(hipStreamSynchronize(s));
  // 2. the padded stream: every slice starts on a window boundary
  std::vector<int32_t> h_win((size_t) S + 1), h_shift((size_t) S);
  int64_t nwin = 0;
  for (int t = 0; t < S; ++t) {
    h_win[(size_t) t] = (int32_t) nwin;
    h_shift[(size_t) t] = (int32_t) (nwin * HOT_WIN - h_trp[(size_t) t]);
    nwin += cdiv((int64_t) h_trp[(size_t) t + 1] - h_trp[(size_t) t], HOT_WIN);
  }
  h_win[(size_t) S] = (int32_t) nwin;
  const int64_t np = nwin * HOT_WIN;  // padded entries
  if (np > INT32_MAX - 8 || nwin < 1)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  int32_t *shift = nullptr, *prow = nullptr, *flag = nullptr;
  long long* partials = nullptr;
  if ((rc = dev_alloc((void**) &pl->ps_slice_win, (size_t) (S + 1) * 4, s)) || (rc = g.alloc((void**) &shift, (size_t) S * 4)) ||
      (rc = g.alloc((void**) &prow, (size_t) np * 4)) || (rc = g.alloc((void**) &flag, (size_t) (np + 1) * 4)) ||
      (rc = g.alloc((void**) &partials, (size_t) (cdiv(np > m ? np : m, 2048) + 2) * sizeof(long long))) ||
      (rc = dev_alloc(&pl->hot_val, (size_t) (np + HOT_WIN) * sizeof(T), s)) ||
      (rc = dev_alloc((void**) &pl->hot_col, (size_t) (np + HOT_WIN) * 2, s)) ||
      (rc = dev_alloc((void**) &pl->hot_src, (size_t) np * 4, s)))
    return rc;
  SPB_HIP(hipMemcpyAsync(pl->ps_slice_win, h_win.data(), (size_t) (S + 1) * 4, hipMemcpyHostToDevice, s));
  SPB_HIP(hipMemcpyAsync(shift, h_shift.data(), (size_t) S * 4, hipMemcpyHostToDevice, s));
  SPB_HIP(hipMemsetAsync(prow, 0xFF, (size_t) np * 4, s));                            // pads: row -1,
  SPB_HIP(hipMemsetAsync(pl->hot_src, 0xFF, (size_t) np * 4, s));                     // no source,
  SPB_HIP(hipMemsetAsync(pl->hot_val, 0, (size_t) (np + HOT_WIN) * sizeof(T), s));    // value 0,
  SPB_HIP(hipMemsetAsync(pl->hot_col, 0xFF, (size_t) (np + HOT_WIN) * 2, s));         // column 0xffff
  hipLaunchKernelGGL((ps_place_kernel<T>), dim3((unsigned) cdiv(nnz, 256)), dim3(256), 0, s, nnz, S, t_rowptr, shift, t_rows, t_pos, W,
                     pl->colind, static_cast<const T*>(values_p), prow, pl->hot_src, static_cast<T*>(pl->hot_val), pl->hot_col);
  SPB_HIP(hipStreamSynchronize(s));  // (h_win / h_shift leave scope with this frame only, but keep the uploads simple)
  // 3. the (row, slice) pairs
  hipLaunchKernelGGL(ps_flags_kernel, dim3((unsigned) cdiv(np, 256)), dim3(256), 0, s, np, prow, flag);
  hipLaunchKernelGGL(ps_slice_flags_kernel, dim3((unsigned) cdiv(S, 256)), dim3(256), 0, s, S, pl->ps_slice_win, flag);
  long long* total_dev = scan_counts_i32(s, np, flag, partials);
  long long m_pairs = 0;
  if ((rc = readback_add(h, &m_pairs, total_dev, sizeof(m_pairs))) || (rc = readback_flush(h)))
    return rc;
  SPB_HIP(hipGetLastError());
  if (m_pairs < 2 || m_pairs >= INT32_MAX - 8)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  // (worth it only where rows meet slices more than once: SPBLAS_GFX950_PB_PS_MAX_PCT, pairs per 100 entries)
  if (m_pairs * 100 > nnz * (long long) hot_env("SPBLAS_GFX950_PB_PS_MAX_PCT", 75))
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  int32_t *pair_row = nullptr, *rowptr2 = nullptr, *iota = nullptr, *ap_slice = nullptr, *ap_pay = nullptr;
  if ((rc = dev_alloc(&pl->hot_rowptr, (size_t) (m_pairs + 1) * 4, s)) ||
      (rc = dev_alloc((void**) &pl->hot_rows, (size_t) m_pairs * 4, s)) || (rc = g.alloc((void**) &pair_row, (size_t) m_pairs * 4)) ||
      (rc = g.alloc((void**) &rowptr2, (size_t) (S + 1) * 4)) || (rc = g.alloc((void**) &iota, (size_t) m_pairs * 4)) ||
      (rc = dev_alloc((void**) &pl->ps_ap_rowptr, (size_t) (m + 1) * 4, s)) ||
      (rc = g.alloc((void**) &ap_slice, (size_t) m_pairs * 4)) || (rc = g.alloc((void**) &ap_pay, (size_t) m_pairs * 4)) ||
      (rc = dev_alloc((void**) &pl->ps_ap_col, (size_t) m_pairs * 4, s)))
    return rc;
  hipLaunchKernelGGL(ps_pairs_kernel, dim3((unsigned) cdiv(np + 1, 256)), dim3(256), 0, s, np, flag, prow,
                     static_cast<int32_t*>(pl->hot_rowptr), pair_row);
  hipLaunchKernelGGL(ps_slice_pairs_kernel, dim3((unsigned) cdiv(S + 1, 256)), dim3(256), 0, s, S, pl->ps_slice_win, flag, rowptr2);
  hipLaunchKernelGGL(ps_iota_kernel, dim3((unsigned) cdiv(m_pairs, 256)), dim3(256), 0, s, (int64_t) m_pairs, iota);
  SPB_HIP(hipGetLastError());
  // 4. the pair matrix by row (S x m by slice -> its transpose), each pair carrying its slice-major number
  if ((rc = spblas_gfx950_csr_transpose_internal(h, S, m, (int64_t) m_pairs, rowptr2, pair_row, iota, pl->ps_ap_rowptr, ap_slice, ap_pay,
                                                 SPBLAS_GFX950_F32)))
    return rc;
  // (the pair matrix is m x S: its "column" is the slice number, so the tiled plan over it has ONE x slice and its tiles
  // are the row bins; inside a bin this builder orders the pairs itself, by slice then row)
  SPB_HIP(hipMemcpyAsync(pl->ps_ap_col, ap_slice, (size_t) m_pairs * 4, hipMemcpyDeviceToDevice, s));
  // 5. the tiled plan over the pairs: its bins, row map, work lists, row words and duplicate flags are those of the product
  //    stream the pre-summing expand writes; its own A' stream is never multiplied
  void* zeros = nullptr;
  if ((rc = g.alloc(&zeros, (size_t) m_pairs * sizeof(T))))
    return rc;
  SPB_HIP(hipMemsetAsync(zeros, 0, (size_t) m_pairs * sizeof(T), s));
  auto* rp = new (std::nothrow) spblas_gfx950_plan_s();
  if (!rp)
    return SPBLAS_GFX950_STATUS_ALLOC_FAILED;
  pl->rest_plan = rp;
  rp->m = m;
  rp->n = S;
  rp->nnz = (int64_t) m_pairs;
  rp->rowptr = pl->ps_ap_rowptr;
  rp->colind = pl->ps_ap_col;
  rp->offset_type = SPBLAS_GFX950_I32;
  rp->value_type = pl->value_type;
  rp->is_child = -1;
  if ((rc = spmv_plan_structures(h, rp, SPBLAS_GFX950_SPMV_ROWBLOCK)))
    return rc;
  if ((rc = spmv_sliced_build(h, rp, zeros, false)))
    return rc;
  if (rp->enc8 || !rp->s_lrow || rp->hub_len > 0 || rp->n_split > 0 || rp->n_slices != 1)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;  // (the pair plan: 16-bit row words, whole rows, one x slice)
  rp->alg = SPBLAS_GFX950_SPMV_SLICED;
  {
    // the slots of the pairs inside every bin, by slice then row: consecutive pairs of the slice-major stream then lie
    // next to each other in the product stream until the bin changes
    constexpr int BLK = sizeof(T) == 4 ? 32 : 16;
    const int64_t NB = rp->n_rblk, tiles = NB * S;
    if (tiles > INT32_MAX - 8)
      return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
    int32_t *inv = nullptr, *pbin = nullptr, *tcnt = nullptr, *tfirst = nullptr;
    long long* tpart = nullptr;
    if ((rc = g.alloc((void**) &pbin, (size_t) m_pairs * 4)) || (rc = g.alloc((void**) &tcnt, (size_t) (tiles + 1) * 4)) ||
        (rc = g.alloc((void**) &tfirst, (size_t) tiles * 4)) ||
        (rc = g.alloc((void**) &tpart, (size_t) (cdiv(tiles, 2048) + 2) * sizeof(long long))))
      return rc;
    if (rp->s_nzrow) {
      if ((rc = g.alloc((void**) &inv, (size_t) m * 4)))
        return rc;
      SPB_HIP(hipMemsetAsync(inv, 0, (size_t) m * 4, s));
      hipLaunchKernelGGL(ps_rowinv_kernel, dim3((unsigned) cdiv(rp->s_m, 256)), dim3(256), 0, s, rp->s_m,
                         static_cast<const int32_t*>(rp->s_nzrow), inv);
    }
    SPB_HIP(hipMemsetAsync(tcnt, 0, (size_t) (tiles + 1) * 4, s));
    const int32_t* binrow = static_cast<const int32_t*>(rp->s_binrow);
    hipLaunchKernelGGL(ps_tiles_kernel, dim3((unsigned) cdiv(m_pairs, 256)), dim3(256), 0, s, (int64_t) m_pairs, S, rp->rows_per_blk, NB,
                       pair_row, rowptr2, inv, binrow, pbin, tcnt, tfirst);
    (void) scan_counts_i32(s, tiles, tcnt, tpart);
    hipLaunchKernelGGL(ps_slots_kernel, dim3((unsigned) cdiv(m_pairs, 256)), dim3(256), 0, s, (int64_t) m_pairs, S, rp->rows_per_blk, BLK,
                       pair_row, rowptr2, inv, binrow, pbin, tcnt, tfirst, static_cast<const int32_t*>(rp->s_binblk), pl->hot_rows,
                       rp->s_lrow);
    SPB_HIP(hipGetLastError());
    if ((rc = spmv_sliced_reflag_dups(h, rp)))
      return rc;
  }
  // 6. pair starts into the entry stream, a descriptor per window, the pairs that cross a window boundary, partials
  auto* hp = new (std::nothrow) spblas_gfx950_plan_s();
  if (!hp)
    return SPBLAS_GFX950_STATUS_ALLOC_FAILED;
  pl->hot_plan = hp;  // (carries the window count only: the second form needs no per-window row tables)
  hp->m = (int64_t) m_pairs;
  hp->n = W;
  hp->nnz = np;
  hp->offset_type = SPBLAS_GFX950_I32;
  hp->value_type = pl->value_type;
  hp->is_child = -1;
  hp->nwin = nwin;
  {
    unsigned long long* n_over_dev = nullptr;
    if ((rc = g.alloc((void**) &n_over_dev, sizeof(unsigned long long))) ||
        (rc = dev_alloc(&pl->ps_desc, (size_t) nwin * sizeof(ps_windesc), s)) ||
        (rc = dev_alloc((void**) &pl->ps_win_pair0, (size_t) nwin * 4, s)))
      return rc;
    SPB_HIP(hipMemsetAsync(n_over_dev, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(ps_headbits_kernel, dim3((unsigned) cdiv(np, 256)), dim3(256), 0, s, np, flag, pl->hot_col);
    hipLaunchKernelGGL(ps_windesc_kernel, dim3((unsigned) cdiv(nwin, 256)), dim3(256), 0, s, nwin, np, flag, pl->hot_rows,
                       static_cast<ps_windesc*>(pl->ps_desc), pl->ps_win_pair0, n_over_dev);
    unsigned long long n_over = 0;
    if ((rc = readback_add(h, &n_over, n_over_dev, sizeof(n_over))) || (rc = readback_flush(h)))
      return rc;
    SPB_HIP(hipGetLastError());
    pl->ps_over = (int64_t) n_over;
  }
  {
    unsigned long long* n_cross_dev = nullptr;
    if ((rc = g.alloc((void**) &n_cross_dev, sizeof(unsigned long long))) ||
        (rc = dev_alloc((void**) &pl->hot_cross, (size_t) (nwin + 1) * 4, s)) ||
        (rc = dev_alloc(&pl->hot_part, (size_t) 2 * nwin * sizeof(T), s)))
      return rc;
    SPB_HIP(hipMemsetAsync(n_cross_dev, 0, sizeof(unsigned long long), s));
    SPB_HIP(hipMemsetAsync(pl->hot_part, 0, (size_t) 2 * nwin * sizeof(T), s));
    hipLaunchKernelGGL((hot_cross_rows_kernel<int32_t>), dim3((unsigned) cdiv(m_pairs, 256)), dim3(256), 0, s, (int64_t) m_pairs,
                       static_cast<const int32_t*>(pl->hot_rowptr), pl->hot_cross, n_cross_dev);
    unsigned long long n_cross = 0;
    if ((rc = readback_add(h, &n_cross, n_cross_dev, sizeof(n_cross))) || (rc = readback_flush(h)))
      return rc;
    SPB_HIP(hipGetLastError());
    pl->hot_ncross = (int64_t) n_cross;
  }
  pl->ps_mode = 1;
  pl->ps_W = W;
  pl->ps_S = S;
  pl->hot_k = W;
  pl->hot_nnz = nnz;
  pl->hot_m = (int64_t) m_pairs;
  pl->s_uncertain = 1;  // AUTO: let the timed trial decide against the row-block plan
  pl->s_placed = nnz;
  pl->values_ptr = values_p;
  pl->device_bytes += rp->device_bytes + (size_t) np * (sizeof(T) + 6) + (size_t) m_pairs * 12 + (size_t) (m + 1) * 4 +
                      (size_t) 2 * nwin * sizeof(T) + (size_t) nwin * (sizeof(ps_windesc) + 8) + (size_t) (S + 2) * 4;
  SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pb_presum2_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              HOT_LDS));
  SPB_HIP(hipStreamSynchronize(s));
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

// Tries the pre-summing plan.  Same contract as spmv_hot_build.
int spmv_presum_build(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values) {
  const bool o32 = pl->offset_type == SPBLAS_GFX950_I32;
  int rc;
  if (pl->value_type == SPBLAS_GFX950_F32)
    rc = o32 ? presum_build_typed<float, int32_t>(h, pl, values) : presum_build_typed<float, int64_t>(h, pl, values);
  else
    rc = o32 ? presum_build_typed<double, int32_t>(h, pl, values) : presum_build_typed<double, int64_t>(h, pl, values);
  if (rc != SPBLAS_GFX950_STATUS_SUCCESS)
    spmv_hot_free(h, pl);
  return rc;
}

// Tries the split.  SUCCESS: pl->hot_plan / pl->rest_plan are set and the plan multiplies through spmv_hot_exec.
// NOT_SUPPORTED: no small set of columns carries enough of the matrix (or the tiled plan declined A_rest): nothing is left
// behind and the caller builds the ordinary tiled plan.
int spmv_hot_build(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values, bool auto_mode) {
  const bool o32 = pl->offset_type == SPBLAS_GFX950_I32;
  int rc;
  if (pl->value_type == SPBLAS_GFX950_F32)
    rc = o32 ? hot_build_typed<float, int32_t>(h, pl, values, auto_mode) : hot_build_typed<float, int64_t>(h, pl, values, auto_mode);
  else
    rc = o32 ? hot_build_typed<double, int32_t>(h, pl, values, auto_mode) : hot_build_typed<double, int64_t>(h, pl, values, auto_mode);
  if (rc != SPBLAS_GFX950_STATUS_SUCCESS)
    spmv_hot_free(h, pl);
  return rc;
}

// the caller's value array changed: both halves take their values again through the source positions of the split
int spmv_hot_update(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values) {
  hipStream_t s = h->stream;
  spblas_gfx950_plan_s* rp = pl->rest_plan;
  if (pl->ps_mode) {  // the entry stream by x slice takes its values again; the pair plan holds structure only
    const int64_t np = pl->hot_plan->nnz;
    if (pl->value_type == SPBLAS_GFX950_F32)
      hipLaunchKernelGGL((hot_gather_values_kernel<float>), dim3((unsigned) cdiv(np, 256)), dim3(256), 0, s, np, pl->hot_src,
                         static_cast<const float*>(values), static_cast<float*>(pl->hot_val));
    else
      hipLaunchKernelGGL((hot_gather_values_kernel<double>), dim3((unsigned) cdiv(np, 256)), dim3(256), 0, s, np, pl->hot_src,
                         static_cast<const double*>(values), static_cast<double*>(pl->hot_val));
    SPB_HIP(hipGetLastError());
    pl->values_ptr = values;
    return SPBLAS_GFX950_STATUS_SUCCESS;
  }
  if (pl->value_type == SPBLAS_GFX950_F32) {
    hipLaunchKernelGGL((hot_gather_values_kernel<float>), dim3((unsigned) cdiv(pl->hot_nnz, 256)), dim3(256), 0, s, pl->hot_nnz,
                       pl->hot_src, static_cast<const float*>(values), static_cast<float*>(pl->hot_val));
    hipLaunchKernelGGL((hot_gather_values_kernel<float>), dim3((unsigned) cdiv(rp->nnz, 256)), dim3(256), 0, s, rp->nnz,
                       pl->rest_src, static_cast<const float*>(values), static_cast<float*>(pl->rest_val));
  } else {
    hipLaunchKernelGGL((hot_gather_values_kernel<double>), dim3((unsigned) cdiv(pl->hot_nnz, 256)), dim3(256), 0, s, pl->hot_nnz,
                       pl->hot_src, static_cast<const double*>(values), static_cast<double*>(pl->hot_val));
    hipLaunchKernelGGL((hot_gather_values_kernel<double>), dim3((unsigned) cdiv(rp->nnz, 256)), dim3(256), 0, s, rp->nnz,
                       pl->rest_src, static_cast<const double*>(values), static_cast<double*>(pl->rest_val));
  }
  SPB_HIP(hipGetLastError());
  const int rc = spmv_sliced_update(h, rp, pl->rest_val);
  if (rc)
    return rc;
  pl->values_ptr = values;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

} // namespace spb
