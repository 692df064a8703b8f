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
// one entry in 16: WHOLE 128-byte lines of the column indices (32 consecutive entries -- two or three rows of a graph), one
// line out of every 16, its place inside the group of 16 changing from group to group.  (Round 4, end: one entry out of
// every 16 consecutive ones touched every line of the array -- 1.07 GB for a 67 MB sample.)
__global__ __launch_bounds__(256) void hot_sample_kernel(int64_t nnz, const int32_t* __restrict__ colind,
                                                         int32_t* __restrict__ cnt) {
  const int64_t t = (int64_t) blockIdx.x * 256 + threadIdx.x;
  const int64_t g = t >> 5;  // group of 16 lines
  const uint32_t hsh = (uint32_t) g * 2654435761u;
  const int64_t p = (g * 16 + (hsh >> 28)) * 32 + (t & 31);
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

// pos = exclusive scan of the column flags: colmap[c] = index in the hot list or -1, hot_cols[index] = c (ascending),
// bitmap = one bit per column (2 MB for 16.8 M columns: what the per-entry test below reads -- it stays in the L2s, where
// the 64 MB column map made every entry's lookup a line from memory: 12.5 GB for 268 M entries)
__global__ __launch_bounds__(256) void hot_colmap_kernel(int64_t n, const int32_t* __restrict__ pos,
                                                         int32_t* __restrict__ colmap, int32_t* __restrict__ hot_cols,
                                                         unsigned long long* __restrict__ bitmap) {
  const int64_t c = (int64_t) blockIdx.x * 256 + threadIdx.x;
  bool hot = false;
  if (c < n) {
    const int32_t i = pos[c];
    hot = pos[c + 1] != i;
    colmap[c] = hot ? i : -1;
    if (hot)
      hot_cols[i] = (int32_t) c;
  }
  const unsigned long long word = __ballot(hot);  // columns 64 w .. 64 w + 63 (a workgroup starts on a multiple of 256)
  if ((threadIdx.x & 63) == 0 && (c & ~(int64_t) 63) < n)
    bitmap[c >> 6] = word;
}

__global__ __launch_bounds__(256) void hot_flag_entries_kernel(int64_t nnz, const int32_t* __restrict__ colind,
                                                               const unsigned long long* __restrict__ bitmap,
                                                               int32_t* __restrict__ flag) {
  const int64_t p = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (p < nnz) {
    const int32_t c = stream_load(colind + p);
    flag[p] = (int32_t) ((bitmap[c >> 6] >> (c & 63)) & 1ull);
  }
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
  if (hotpos[p + 1] != hp) {  // a hot entry: only those look their column up (the few thousand map lines of the hot columns)
    hot_val[hp] = v;
    hot_col[hp] = (uint16_t) colmap[c];
    if (hot_src)  // (source positions only for plans that refresh their values: plan.hpp keep_src)
      hot_src[hp] = (int32_t) p;
  } else {
    const int64_t rp = p - hp;
    rest_val[rp] = v;
    rest_col[rp] = c;
    if (rest_src)
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

// the rows of A_hot that cross a window boundary: y[hot_rows[r]] += alpha * (tail of the window the row starts in + heads of
// the windows it runs through), 16 lanes per row
template <typename T, typename O>
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
  if (i < n_cross && lig == 0)
    y[hot_rows[r]] += alpha * (s + part_tail[w0]);
}

// inspect: the rows of A_hot whose entries lie in more than one window
template <typename O>
__global__ __launch_bounds__(256) void hot_cross_rows_kernel(int64_t m_hot, const O* __restrict__ rowptr,
                                                             int32_t* __restrict__ cross_rows,
                                                             unsigned long long* __restrict__ n_cross) {
  const int64_t r = (int64_t) blockIdx.x * 256 + threadIdx.x;
  bool cross = false;
  if (r < m_hot) {
    const int64_t p0 = (int64_t) rowptr[r], p1 = (int64_t) rowptr[r + 1];
    cross = p0 / HOT_WIN != (p1 - 1) / HOT_WIN;
  }
  const unsigned long long mask = __ballot(cross);  // one atomic per wavefront on the single counter
  if (mask) {
    const int lane = threadIdx.x & 63, leader = __builtin_ctzll(mask);
    unsigned long long base = 0;
    if (lane == leader)
      base = atomicAdd(n_cross, (unsigned long long) __popcll(mask));
    base = __shfl(base, leader);
    if (cross)
      cross_rows[base + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t) r;
  }
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
    hipLaunchKernelGGL((pb_hot_fixup_kernel<T, O>), dim3((unsigned) cdiv(pl->hot_ncross * 16, 256)), dim3(256), 0, s,
                       pl->hot_ncross, pl->hot_cross, rowptr, head, tail, static_cast<T*>(y), alpha, pl->hot_rows);
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
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
  unsigned long long* colbits = nullptr;
  if ((rc = g.alloc((void**) &colmap, (size_t) n * 4)) || (rc = g.alloc((void**) &colbits, (size_t) (n / 64 + 1) * 8)) ||
      (rc = dev_alloc((void**) &pl->hot_cols, (size_t) cols * 4, s)))
    return rc;
  hipLaunchKernelGGL(hot_flag_cols_kernel, dim3((unsigned) cdiv(n, 256)), dim3(256), 0, s, n, cnt, thr, pos);
  (void) scan_counts_i32(s, n, pos, partials);
  hipLaunchKernelGGL(hot_colmap_kernel, dim3((unsigned) cdiv(n, 256)), dim3(256), 0, s, n, pos, colmap, pl->hot_cols, colbits);
  pl->hot_k = (int) cols;
  // stable split of the entries
  hipLaunchKernelGGL(hot_flag_entries_kernel, dim3((unsigned) cdiv(nnz, 256)), dim3(256), 0, s, nnz, colind, colbits, pos);
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
  // the source position of every entry (what a value refresh gathers through) only when the plan is known to refresh:
  // spmv_sliced.hip, keep_src (the first change of values builds a plan without them again, with them)
  const bool keep_src = pl->keep_src || pl->refresh_each_call || hot_env("SPBLAS_GFX950_PB_KEEP_SRC", 0);
  if ((rc = dev_alloc(&pl->hot_val, hot_pad * sizeof(T), s)) ||
      (rc = dev_alloc((void**) &pl->hot_col, hot_pad * 2, s)) ||
      (keep_src && (rc = dev_alloc((void**) &pl->hot_src, (size_t) n_hot * 4, s))) ||
      (rc = dev_alloc(&pl->rest_rowptr, (size_t) (m + 1) * sizeof(O), s)) ||
      (rc = dev_alloc(&pl->rest_val, (size_t) n_rest * sizeof(T), s)) ||
      (rc = dev_alloc((void**) &pl->rest_col, (size_t) n_rest * 4, s)) ||
      (keep_src && (rc = dev_alloc((void**) &pl->rest_src, (size_t) n_rest * 4, s))))
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
  rp->keep_src = keep_src ? 1 : 0;
  if ((rc = spmv_plan_structures(h, rp, SPBLAS_GFX950_SPMV_ROWBLOCK)))
    return rc;
  if ((rc = spmv_sliced_build(h, rp, pl->rest_val, auto_mode)))
    return rc;
  rp->alg = SPBLAS_GFX950_SPMV_SLICED;
  rp->nt_products = pl->nt_products;
  pl->s_uncertain = rp->s_uncertain;
  pl->s_placed = rp->s_placed;
  pl->values_ptr = values_p;
  // A_rest as a CSR matrix was the INPUT of its tiles; only a plan that refreshes values (the gather goes caller -> rest_val
  // -> tiles) or whose tiles left rows out (hub rows are multiplied from the CSR arrays) needs it afterwards: 12 B per
  // entry of A_rest otherwise given back (cfg4: 2.27 of 9.5 GB)
  size_t rest_csr_bytes = (size_t) n_rest * (sizeof(T) + 4);
  {
    spblas_gfx950_plan_s* tp = rp->rest_plan ? nullptr : rp;  // (a remainder that was split again keeps its input)
    if (!keep_src && tp && !(tp->hub_len > 0 && tp->n_hub > 0) && !hot_env("SPBLAS_GFX950_PB_KEEP_REST", 0)) {
      dev_free(pl->rest_val, s);
      dev_free(pl->rest_col, s);
      pl->rest_val = nullptr;
      pl->rest_col = nullptr;
      rp->colind = nullptr;
      rp->values_ptr = nullptr;
      rest_csr_bytes = 0;
    }
  }
  pl->device_bytes += hp->device_bytes + rp->device_bytes + (size_t) cols * 4 + (size_t) n_hot * (sizeof(T) + 2 + (keep_src ? 4 : 0)) +
                      rest_csr_bytes + (keep_src ? (size_t) n_rest * 4 : 0) + (size_t) (m + 1) * sizeof(O) +
                      (size_t) (m_hot + 1) * (sizeof(O) + 4);
  pl->hot_m = (int64_t) m_hot;
  pl->device_bytes += (size_t) (hp->nwin + 1) * 4 + (size_t) 2 * hp->nwin * sizeof(T);
  SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pb_hot_rows_kernel<T, O>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, HOT_LDS));
  return SPBLAS_GFX950_STATUS_SUCCESS;
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

// Loads this file's code object (the runtime loads a code object at the first use of one of its kernels: milliseconds
// that would otherwise fall on the caller's first inspect / compute call -- handle.hip: spblas_gfx950_create).
namespace spb {
void preload_hot() {
  hipFuncAttributes attr;
  (void) hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&hot_sample_kernel));
  (void) hipGetLastError();
}
} // namespace spb
