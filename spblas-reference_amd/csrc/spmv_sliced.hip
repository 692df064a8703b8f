// Column-sliced ("propagation blocking") SpMV for matrices whose x gathers miss every
// cache: SPBLAS_GFX950_SPMV_SLICED.
//
// Why: on BASELINE cfg2 (10M x 10M, uniform random columns) the CSR kernels gather each
// x[col] as a separate 128-byte L2 miss (rocprofv3: TCC_EA0_RDREQ_128B = 0.97 per nonzero,
// 12.4 GB of fabric reads for 0.92 GB of algorithmic bytes -- profiles/r01a_summary.md).
// The only memory on the CU that sustains random 4-byte accesses at the needed rate is LDS,
// so multiply_inspect re-tiles A once on the device:
//
//   tile       (column slice s, wave-bin b): slice = W consecutive columns (W*sizeof(T) <= 160 KiB of LDS),
//              wave-bin = H consecutive rows owned by ONE wavefront of the reduce kernel
//   run        the entries of one tile, padded to whole BLOCKS of 32 entries (pads: value 0, row = H)
//   A' order   blocks sorted by (slice, bin): s_val[] value, s_col[] 16-bit column inside the slice
//   P order    the same blocks sorted by (bin, slice), every bin padded to whole GROUPS of 8 blocks:
//              s_row[] 15-bit row inside the bin | bit 15 = "duplicate" flag, P[] the products
//   blkdst[]   P-order block of every A'-order block
//
// and multiply() runs two streaming kernels (no global gathers, no global atomics):
//   expand  x slice -> LDS; flat pass over the slice's contiguous blocks of A':
//           P[blkdst[blk]] = s_val[blk] * xs[s_col[blk]]   -- linear 16-byte reads, and writes that are
//           linear inside every 128-byte block (a block-permuted stream: tools/ubench/hbm_pieces.hip shows
//           it runs within 3-5 % of the fully linear copy as long as whole lines are written)
//   reduce  one wavefront per wave-bin: H accumulators in LDS; the bin's part of P and s_row is ONE
//           contiguous stream, read with 16-byte / 8-byte lane accesses (4 entries per lane, no run
//           descriptors, no line over-fetch);  y = alpha*acc + beta*y
// Round 1 stored P in A' order, so that the reduce read it in ~100-entry runs: 142 us for 0.77 GB, issue- and
// over-fetch-bound (34 instructions per 64 entries).  Moving the small pieces to the WRITE side (whole 128-byte
// lines at permuted places) makes the reduce a pure stream: 4x fewer instructions per entry.
// HBM traffic per stored entry: 6 B + 4 B (expand) + 6 B (reduce) = 16 B vs 8 B algorithmic, plus the block
// padding (half a block per run; tall bins keep runs ~200 entries long at cfg2).
//
// Why wave-owned bins: ds_add_f32 retires ~0.33 lanes/clk/CU on gfx950 (tools/ubench/lds_atomic:
// 1e8 LDS float atomics = 509 us, 12x slower than integer atomics or a plain read-add-write),
// so accumulation must be a plain LDS read-modify-write.  That is race free iff (a) no other
// wavefront touches the rows -- each wave owns its bin -- and (b) the 256 entries one reduce step
// handles (64 lanes x 4) hit distinct rows.  (b) is arranged at inspect time: within each
// 256-entry group all but one entry of a repeated row carry the duplicate flag and are
// applied afterwards with the (slow, rare: ~3 % of entries) LDS atomic.
// The order of additions into a row is fixed by the plan, so results are run-to-run
// reproducible for a given plan (plans built twice may order entries differently).
#include <algorithm>
#include <atomic>
#include <type_traits>
#include <vector>

#include "common.hpp"
#include "plan.hpp"
#include "scan.hpp"

#include <cstdlib>
#include <ctime>

namespace spb {

// hot-column split of a column-skewed matrix (spmv_hot.hip)
int spmv_hot_build(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values, bool auto_mode);
int spmv_hot_exec(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* alpha, const void* x, const void* beta,
                  void* y);
int spmv_hot_rows(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* alpha, const void* x, void* y);
int spmv_hot_update(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values);
void spmv_hot_free(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl);
void spmv_sliced_free(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl);

static int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return v && *v ? std::atoi(v) : dflt;
}

static constexpr int PB_THREADS = 1024;         // expand: 16 waves share one x slice
static constexpr int PB_FTHREADS = 512;         // inspect (flag kernel): 8 waves, one wave-bin each
static constexpr int PB_FWAVES = PB_FTHREADS / 64;
static constexpr int PB_RWAVES_DEFAULT = 4;     // reduce: wave-bins per workgroup (plan->rwaves)
static constexpr int PB_LDS_BYTES = 80 * 1024;  // two workgroups per CU (160 KiB LDS)
// entries per block -- the padding unit of a run: 128 B of products (32 fp32, 16 fp64), so the expand's scattered
// product stores are whole aligned 128-byte pieces for either type (smaller or unaligned pieces write 28-30 % slower,
// tools/ubench/hbm_pieces.hip) and an fp64 run pads to 16 entries, not 32 (R-MAT scale 24: 47 % -> 25 % padding)
template <typename T>
struct pb_geom {
  static constexpr int BLK = 128 / (int) sizeof(T);
  static constexpr int GBLK = 256 / BLK;  // blocks per reduce group (PB_GRP entries): the padding unit of a bin
};
static inline int pb_blk_of(int value_type) { return value_type == SPBLAS_GFX950_F32 ? 32 : 16; }
static constexpr int PB_GRP = 256;              // entries per reduce step: 64 lanes x 4

// ---- one-byte row codes (round 3) -----------------------------------------------------------------------------
// The reduce streams 4 B of product + 2 B of row word per padded entry; with the runs sorted by row the row word
// shrinks to ONE byte (tools/build_variant.sh u8rows, cfg2: reduce 140.6 -> 118.1 us).  Encoding ("enc8"), P order:
//   s_code[e]   u8 per entry: the row advance since the previous entry of the same BLOCK (0..254), or 255 =
//               "advance 255 rows and leave this entry out of the main pass" (a pad, or an EXCEPTION: an entry whose
//               row lies >= 255 rows beyond the decoder's position -- 3e-5 of the entries at cfg2's tile density)
//   s_hdr[blk]  per block: row of the block's first entry (u16) + one duplicate flag per entry (the bit the u16
//               encoding keeps in bit 15 of the row word); 8 B per 32 fp32 entries, 4 B per 16 fp64 entries
//   exceptions  per wave-bin, at most exc_cap of them: (P index, row); the reduce adds them with the LDS float atomic
//               after its main pass.  A bin with more exceptions, or a run larger than the inspect's staging area, sets
//               enc_fail and the plan is rebuilt with the 16-bit rows.
// The decoder's row D of entry j of a block is base + code_1 + ... + code_j: D_j = min(r_j, D_{j-1} + 255), i.e.
// D_j = 255 j + min_{k <= j} (r_k - 255 k) -- a prefix minimum at inspect, a prefix sum in the reduce.
// Bytes per padded entry in the reduce: 4 + 1 + 0.25 = 5.25 (was 6); nothing changes for the expand.
template <typename T>
struct pb_hdr;
template <>
struct pb_hdr<float> {
  typedef unsigned long long type;  // bits 0..31 duplicate flags, 32..47 base row
  static __host__ __device__ __forceinline__ type make(unsigned base, unsigned flags) {
    return ((type) base << 32) | flags;
  }
  static __device__ __forceinline__ unsigned base(type h) { return (unsigned) (h >> 32) & 0xffffu; }
  static __device__ __forceinline__ unsigned flags(type h) { return (unsigned) h; }
};
template <>
struct pb_hdr<double> {
  typedef unsigned type;  // bits 0..15 duplicate flags, 16..31 base row
  static __host__ __device__ __forceinline__ type make(unsigned base, unsigned flags) { return (base << 16) | flags; }
  static __device__ __forceinline__ unsigned base(type h) { return h >> 16; }
  static __device__ __forceinline__ unsigned flags(type h) { return h & 0xffffu; }
};
static constexpr int PB_EXC_CAP = 128;   // exceptions a wave-bin may hold

// value of lane - o (o = 1, 2, 4) inside the 16-lane DPP row; lanes without a source get 0
template <int O>
__device__ __forceinline__ unsigned pb_row_shr(unsigned v) {
  return (unsigned) __builtin_amdgcn_update_dpp(0, (int) v, 0x110 + O, 0xf, 0xf, true);
}
// rows of a lane's 4 consecutive entries from their packed codes: row[j] = base + (codes of the block's earlier
// lanes) + c0 + ... + cj, clamped to Hw (tail pads may run past the bin); LPB = lanes per block (8 fp32 / 4 fp64)
template <int LPB>
__device__ __forceinline__ void pb_decode_rows(unsigned cw, unsigned base, int lane, unsigned Hw, unsigned (&row)[4],
                                               bool (&skip)[4]) {
  const unsigned c0 = cw & 0xffu, c1 = (cw >> 8) & 0xffu, c2 = (cw >> 16) & 0xffu, c3 = cw >> 24;
  const unsigned s1 = c0 + c1, s2 = s1 + c2, s3 = s2 + c3;
  const unsigned li = (unsigned) lane & (LPB - 1);
  unsigned t = s3, u;
  u = pb_row_shr<1>(t);
  t += li >= 1 ? u : 0u;
  u = pb_row_shr<2>(t);
  t += li >= 2 ? u : 0u;
  if (LPB > 4) {
    u = pb_row_shr<4>(t);
    t += li >= 4 ? u : 0u;
  }
  const unsigned b = base + t - s3;
  row[0] = min(b + c0, Hw);
  row[1] = min(b + s1, Hw);
  row[2] = min(b + s2, Hw);
  row[3] = min(b + s3, Hw);
  skip[0] = c0 == 255u;
  skip[1] = c1 == 255u;
  skip[2] = c2 == 255u;
  skip[3] = c3 == 255u;
}


// ---- inspect --------------------------------------------------------------------------
// One workgroup per wave-bin: all entries of the bin's rows share wb, so the per-slice counts
// live in an LDS histogram and are written out without any global atomic (the first version
// issued one global atomic per entry: 3.7 ms + 6.7 ms at cfg2).
// last row r in [lo, hi) with rowptr[r] <= p
template <typename O>
__device__ __forceinline__ int64_t pb_row_of(const O* __restrict__ rowptr, int64_t lo, int64_t hi, O p) {
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (rowptr[mid] <= p)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// rows [*r0, *r1) of wave-bin wb: equal heights, or -- row-skewed matrices -- the variable heights of binrow[]
__device__ __forceinline__ void pb_bin_rows(const int32_t* __restrict__ binrow, int64_t wb, int H, int64_t m,
                                            int64_t* r0, int64_t* r1) {
  if (binrow) {
    *r0 = binrow[wb];
    *r1 = binrow[wb + 1];
  } else {
    *r0 = wb * H;
    *r1 = (*r0 + H) < m ? (*r0 + H) : m;
  }
}

// Variable-height bins: row r starts a bin when it lies on the H-row grid (no bin is taller than the LDS
// accumulators allow) or when the entry count before it crosses a multiple of E (no bin holds much more than E
// entries plus one row).  flag[] is scanned into bin numbers; pb_bin_rows_kernel writes the boundaries.
template <typename O>
__global__ __launch_bounds__(256) void pb_bin_flags_kernel(int64_t m, int H, int64_t E, const O* __restrict__ rowptr,
                                                           int32_t* __restrict__ flag) {
  const int64_t r = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (r >= m)
    return;
  flag[r] = (r % H == 0) || ((int64_t) rowptr[r] / E != (int64_t) rowptr[r - 1] / E);
}
// The row map of a plan whose tiles are NOT built over the rows of y one to one.  pieces[r] = compact rows that row r
// of y becomes: 0 for an empty row that is taken out, 1 for an ordinary row, cdiv(len, L) for a row longer than L
// entries (L = 0: no splitting).  After the scan pos[r] = compact rows before row r.
static constexpr int32_t PB_PIECE_BIT = (int32_t) 0x80000000;
template <typename O>
__global__ __launch_bounds__(256) void pb_row_pieces_kernel(int64_t m, const O* __restrict__ rowptr, int drop_empty, int L,
                                                            int32_t* __restrict__ pieces) {
  const int64_t r = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (r >= m)
    return;
  const int64_t len = (int64_t) (rowptr[r + 1] - rowptr[r]);
  pieces[r] = len == 0 ? (drop_empty ? 0 : 1) : (L > 0 && len > L ? (int32_t) ((len + L - 1) / L) : 1);
}
// rowptr_c / nzrow of the compact rows (nzrow = row of y, PB_PIECE_BIT set on the pieces of a split row), the list of
// empty rows that were taken out and the list of split rows (both appended in arbitrary order)
template <typename O>
__global__ __launch_bounds__(1024) void pb_row_map_kernel(int64_t m, const O* __restrict__ rowptr,
                                                         const int32_t* __restrict__ pos, int L, O* __restrict__ rowptr_c,
                                                         int32_t* __restrict__ nzrow, int32_t* __restrict__ zrow,
                                                         int4* __restrict__ split_rows,
                                                         unsigned long long* __restrict__ counters) {
  const int64_t r = (int64_t) blockIdx.x * 1024 + threadIdx.x;
  const int32_t i = r < m ? pos[r] : 0, k = r < m ? pos[r + 1] - i : -1;
  {
    // workgroup-aggregated append of the empty rows: ONE atomic per 1 024 rows on the single counter (R-MAT scale 24 has
    // 9.4 M empty rows: an atomic each took 3 ms, one per wavefront -- 262 k same-address atomics at ~11 ns -- still 2.9 ms)
    __shared__ unsigned s_cnt[16];
    __shared__ unsigned long long s_base;
    const unsigned long long mask = __ballot(k == 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
      s_cnt[wave] = (unsigned) __popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned tot = 0;
      for (int w = 0; w < 16; ++w) {
        const unsigned c = s_cnt[w];
        s_cnt[w] = tot;
        tot += c;
      }
      s_base = tot ? atomicAdd(&counters[0], (unsigned long long) tot) : 0ull;
    }
    __syncthreads();
    if (k == 0)
      zrow[s_base + s_cnt[wave] + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t) r;
  }
  if (r > m)
    return;
  if (r == m) {
    rowptr_c[pos[m]] = rowptr[m];
    return;
  }
  if (k == 0) {
  } else if (k == 1) {
    nzrow[i] = (int32_t) r;
    rowptr_c[i] = rowptr[r];
  } else {
    split_rows[atomicAdd(&counters[1], 1ull)] = make_int4((int) r, i, k, 0);
    for (int j = 0; j < k; ++j) {
      nzrow[i + j] = (int32_t) r | PB_PIECE_BIT;
      rowptr_c[i + j] = rowptr[r] + (O) j * (O) L;
    }
  }
}
// y[row] = (sum of the row's pieces, in order) + beta * y[row]: one wavefront per split row
template <typename T>
__global__ __launch_bounds__(256) void pb_split_finish_kernel(int64_t n_split, const int4* __restrict__ split_rows,
                                                              const T* __restrict__ piece_out, T* __restrict__ y, T beta) {
  const int64_t i = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= n_split)
    return;
  const int4 e = split_rows[i];
  T s = T(0);
  for (int j = lane; j < e.z; j += 64)
    s += piece_out[(int64_t) e.y + j];
  s = group_sum_c<64>(s);
  if (lane == 0)
    y[e.x] = beta == T(0) ? s : s + beta * y[e.x];
}
// y = beta * y on the empty rows inside [row_begin, row_end)
template <typename T>
__global__ __launch_bounds__(256) void pb_empty_rows_kernel(int64_t n_zero, const int32_t* __restrict__ zrow,
                                                            T* __restrict__ y, T beta, int64_t row_begin, int64_t row_end) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= n_zero)
    return;
  const int64_t r = zrow[i];
  if (r >= row_begin && r < row_end)
    y[r] = beta == T(0) ? T(0) : beta * y[r];
}
// out[b] = original row of the first (compact) row of wave-bin b; out[NB] = m
__global__ __launch_bounds__(256) void pb_bin_orig_rows_kernel(int64_t NB, int64_t s_m, int64_t m,
                                                               const int32_t* __restrict__ binrow,
                                                               const int32_t* __restrict__ nzrow, int32_t* __restrict__ out) {
  const int64_t b = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (b > NB)
    return;
  const int64_t c = binrow[b];
  out[b] = c < s_m ? (nzrow[c] & ~PB_PIECE_BIT) : (int32_t) m;
}

// rows longer than `thr` entries, appended to rows[] in arbitrary order (wave-aggregated append)
template <typename O>
__global__ __launch_bounds__(256) void pb_hub_list_kernel(int64_t m, int thr, const O* __restrict__ rowptr,
                                                          unsigned long long* __restrict__ count,
                                                          int32_t* __restrict__ rows) {
  const int lane = threadIdx.x & 63;
  for (int64_t r0 = (int64_t) blockIdx.x * 256; r0 < m; r0 += (int64_t) gridDim.x * 256) {
    const int64_t r = r0 + threadIdx.x;
    const bool hub = r < m && (int64_t) (rowptr[r + 1] - rowptr[r]) > (int64_t) thr;
    const unsigned long long mask = __ballot(hub);
    if (mask == 0)
      continue;
    unsigned long long base = 0;
    const int leader = __builtin_ctzll(mask);
    if (lane == leader)
      base = atomicAdd(count, (unsigned long long) __popcll(mask));
    base = __shfl(base, leader);
    if (hub)
      rows[base + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t) r;
  }
}
__global__ __launch_bounds__(256) void pb_bin_rows_kernel(int64_t m, const int32_t* __restrict__ pos,
                                                          int32_t* __restrict__ binrow) {
  const int64_t r = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (r > m)
    return;
  if (r == m)
    binrow[pos[m]] = (int32_t) m;
  else if (pos[r + 1] != pos[r])
    binrow[pos[r]] = (int32_t) r;
}

// hub_len > 0: entries of rows longer than hub_len are left out of the tiles (a run that repeats one
// row hundreds of times would serialise on the LDS atomic); pb_hub_rows_kernel adds those rows.
template <typename O>
__global__ __launch_bounds__(256) void pb_count_kernel(int64_t m, const O* __restrict__ rowptr,
                                                       const int32_t* __restrict__ colind, int W, int H, int S, int NB,
                                                       int32_t* __restrict__ cnt, int hub_len,
                                                       const int32_t* __restrict__ binrow,
                                                       uint16_t* __restrict__ wpart) {
  // wpart != nullptr (round 5, plans whose scatter stages in order: one-byte row codes): the counts are taken per PART of
  // the bin as well -- part w = the positions [w L, (w + 1) L) the w-th wavefront of the staged scatter owns -- so that the
  // scatter need not walk the bin's columns once more just to count them: wpart[(bin * 16 + w) * S + slice]
  extern __shared__ int hist[];  // [S] (+ [16][S] per part)
  const int wb = blockIdx.x;
  const int nh = wpart ? 17 * S : S;
  for (int i = threadIdx.x; i < nh; i += 256)
    hist[i] = 0;
  __syncthreads();
  int64_t r0, r1;
  pb_bin_rows(binrow, wb, H, m, &r0, &r1);
  if (r0 < m) {
    const O p0 = rowptr[r0], p1 = rowptr[r1];
    if (hub_len <= 0) {
      // four columns per lane and load (round 5: one 4-byte load and one integer division per entry made this pass 150 us at
      // cfg2 -- 2.7 TB/s); the slice comes from a float reciprocal with one correction step, exact for S <= 16 384 slices
      const float inv_w = 1.0f / (float) W;
      auto slice_of = [&](int c) {
        int sl = (int) ((float) c * inv_w);
        if (sl * W > c)
          --sl;
        else if ((sl + 1) * W <= c)
          ++sl;
        return sl;
      };
      O pa = (p0 + 3) & ~(O) 3;  // first 16-byte aligned position (colind itself is at least 16-byte aligned: hipMalloc / torch)
      if ((reinterpret_cast<uintptr_t>(colind) & 15) != 0 || pa > p1)
        pa = p1;
      const O pb = pa + ((p1 - pa) & ~(O) 3);
      // (per-part histograms: the total per slice is summed from them below)
      const int ne = (int) (p1 - p0);
      const int L = (((ne + 15) / 16) + 63) & ~63;  // = the staged scatter's share of one wavefront
      const float inv_l = 1.0f / (float) (L > 0 ? L : 1);
      auto add = [&](O p, int c) {
        if (wpart) {
          const int q = (int) (p - p0);
          int w = (int) ((float) q * inv_l);
          if (w * L > q)
            --w;
          else if ((w + 1) * L <= q)
            ++w;
          atomicAdd(&hist[S + w * S + slice_of(c)], 1);
        } else {
          atomicAdd(&hist[slice_of(c)], 1);
        }
      };
      for (O p = p0 + threadIdx.x; p < pa; p += 256)
        add(p, colind[p]);
      for (O p = pa + 4 * (O) threadIdx.x; p < pb; p += 4 * 256) {
        const int4 c4 = *reinterpret_cast<const int4*>(colind + p);
        add(p, c4.x);
        add(p + 1, c4.y);
        add(p + 2, c4.z);
        add(p + 3, c4.w);
      }
      for (O p = pb + threadIdx.x; p < p1; p += 256)
        add(p, colind[p]);
      if (wpart) {
        __syncthreads();
        for (int i = threadIdx.x; i < S; i += 256) {
          int t = 0;
          for (int w = 0; w < 16; ++w) {
            const int c = hist[S + w * S + i];
            wpart[((size_t) wb * 16 + w) * S + i] = (uint16_t) (c < 65535 ? c : 65535);
            t += c;
          }
          hist[i] = t;
        }
      }
    } else {
      for (O p = p0 + threadIdx.x; p < p1; p += 256) {
        const int64_t r = pb_row_of(rowptr, r0, r1, p);
        if (rowptr[r + 1] - rowptr[r] > (O) hub_len)
          continue;
        atomicAdd(&hist[colind[p] / W], 1);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < S; i += 256)
    cnt[(int64_t) i * NB + wb] = hist[i];
}

// blocks per run, in A' order (key = s*NB + b); scanned in place into the block offsets aoff[]
__global__ __launch_bounds__(256) void pb_nblk_kernel(int64_t nseg, const int32_t* __restrict__ cnt,
                                                      int32_t* __restrict__ aoff, int blk) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i < nseg)
    aoff[i] = (cnt[i] + blk - 1) / blk;
}

// entries per run in the COMPACT A' stream (round 3): a run occupies its count rounded up to 4 entries -- the lanes of
// the expand own 4 entries each, 16 / 32-byte aligned -- not whole blocks; scanned in place into eoff[]
__global__ __launch_bounds__(256) void pb_ecnt_kernel(int64_t nseg, const int32_t* __restrict__ cnt,
                                                      int32_t* __restrict__ eoff) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i < nseg)
    eoff[i] = (cnt[i] + 3) & ~3;
}

// P order: one workgroup per wave-bin b scans the block counts of its S runs: prel[b*S + s] = blocks of the bin
// before slice s; bintot[b] = the bin's blocks rounded up to whole groups (scanned afterwards into binblk[]).
__global__ __launch_bounds__(256) void pb_bin_prefix_kernel(int S, int NB, const int32_t* __restrict__ cnt,
                                                            int32_t* __restrict__ prel, int32_t* __restrict__ bintot,
                                                            int blk) {
  const int gblk = PB_GRP / blk;
  __shared__ int sm[256];
  __shared__ int carry;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0)
    carry = 0;
  __syncthreads();
  for (int s0 = 0; s0 < S; s0 += 256) {
    const int s = s0 + tid;
    const int v = s < S ? (cnt[(int64_t) s * NB + b] + blk - 1) / blk : 0;
    sm[tid] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const int t = tid >= o ? sm[tid - o] : 0;
      __syncthreads();
      sm[tid] += t;
      __syncthreads();
    }
    const int c = carry;
    if (s < S)
      prel[(int64_t) b * S + s] = c + sm[tid] - v;
    __syncthreads();
    if (tid == 255)
      carry = c + sm[255];
    __syncthreads();
  }
  if (tid == 0)
    bintot[b] = (carry + gblk - 1) / gblk * gblk;
}

// sliceblk[s] = aoff[s*NB] (first A'-order block of slice s), sliceblk[S] = all blocks
__global__ __launch_bounds__(256) void pb_slice_blocks_kernel(int S, int NB, const int32_t* __restrict__ aoff,
                                                              int32_t* __restrict__ sliceblk) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i <= S)
    sliceblk[i] = aoff[(int64_t) i * NB];
}

// Direct scatter (S > PB_STAGE_MAX_S): LDS cursors start at the runs' A' positions; the row of entry p is
// found by a binary search in the bin's slice of rowptr (<= 12 probes, L1/L2 resident).  Pads were set by
// the memsets of the build; this kernel writes the real entries and the block map of its runs.
template <typename T, typename O>
__global__ __launch_bounds__(256) void pb_scatter_kernel(int64_t m, const O* __restrict__ rowptr,
                                                         const int32_t* __restrict__ colind,
                                                         const T* __restrict__ values, int W, int H, int S, int NB,
                                                         const int32_t* __restrict__ aoff,
                                                         const int32_t* __restrict__ prel,
                                                         const int32_t* __restrict__ binblk, T* __restrict__ s_val,
                                                         uint16_t* __restrict__ s_col, uint16_t* __restrict__ s_row,
                                                         int32_t* __restrict__ perm, int32_t* __restrict__ blkdst,
                                                         int hub_len, const int32_t* __restrict__ binrow,
                                                         const int32_t* __restrict__ eoff, int32_t* __restrict__ blksrc) {
  constexpr int PB_BLK = pb_geom<T>::BLK;
  extern __shared__ int smem_i[];
  int* cursor = smem_i;    // [S] next (compact) A' position of the run
  int* pdelta = smem_i + S;  // [S] P position minus A' position of the run's entries
  const int wb = blockIdx.x;
  const int pb0 = binblk[wb];
  for (int i = threadIdx.x; i < S; i += 256) {
    const int64_t key = (int64_t) i * NB + wb;
    const int p0b = pb0 + prel[(int64_t) wb * S + i];
    cursor[i] = eoff[key];
    pdelta[i] = p0b * PB_BLK - eoff[key];
  }
  __syncthreads();
  // the block map of the bin's runs: one wavefront per run, a lane per block
  for (int sl = threadIdx.x >> 6; sl < S; sl += 4) {
    const int64_t key = (int64_t) sl * NB + wb;
    const int a0 = aoff[key], nb = aoff[key + 1] - a0, e0 = eoff[key], p0b = (e0 + pdelta[sl]) / PB_BLK;
    for (int k = threadIdx.x & 63; k < nb; k += 64) {
      blkdst[a0 + k] = p0b + k;
      blksrc[a0 + k] = e0 + k * PB_BLK;
    }
  }
  __syncthreads();
  int64_t r0, r1;
  pb_bin_rows(binrow, wb, H, m, &r0, &r1);
  if (r0 >= m)
    return;
  const O p0 = rowptr[r0], p1 = rowptr[r1];
  for (O p = p0 + threadIdx.x; p < p1; p += 256) {
    const int64_t lo = pb_row_of(rowptr, r0, r1, p);
    if (hub_len > 0 && rowptr[lo + 1] - rowptr[lo] > (O) hub_len)
      continue;
    const int c = colind[p];
    const int sl = c / W;
    const int i = atomicAdd(&cursor[sl], 1);
    s_val[i] = values[p];
    s_col[i] = (uint16_t) (c - sl * W);
    if (perm)
      perm[i] = (int32_t) p;
    s_row[i + pdelta[sl]] = (uint16_t) (lo - r0);
  }
}

// Staged variant of the scatter (S <= PB_STAGE_MAX_S): the direct kernel above issues four 2..4-byte
// stores per entry to ~S different runs, which the memory side turns into one 32-byte write each
// (rocprofv3: 361 M write requests, 11.5 GB for 1.2 GB of payload at cfg2, 5.7 ms).  Here a workgroup of
// 1 024 threads owns the whole LDS of a CU and makes a few passes over its bin: pass k stages the entries
// (position, value, column inside the slice) of the slices [s0, s1) -- as many as fit -- grouped by run in
// LDS, then every wave writes whole runs with contiguous stores; the row comes from a binary search in an
// LDS copy of the bin's row offsets.  colind is re-read once per pass (coalesced).
static constexpr int PB_STAGE_THREADS = 1024;
static constexpr int PB_STAGE_LDS = 160 * 1024;
static constexpr int PB_STAGE_MAX_S = 2048;
static constexpr int PB_STAGE_SP = 256;  // enc8: at most this many slices (per-wave, per-slice cursors in LDS)

// A staged run of the ordered staging is sorted except inside ROUNDS: the entries one wavefront staged from one round of
// 64 consecutive source positions (equal q >> 6, adjacent in the run, a handful at most) took their places in the
// order the LDS atomic served the lanes.  Sort every such group in place by position: each lane ranks its entry inside
// its group (a walk to both ends of the group) and the chunk is rewritten one chunk behind the reads, so that a group
// straddling two chunks of 64 is read whole before any of it is overwritten (a group has at most 64 entries).
template <typename T, typename Q>
__device__ __forceinline__ void pb_sort_rounds(int n, int lo, int lane, Q* __restrict__ st, T* __restrict__ stv,
                                               uint16_t* __restrict__ stc) {
  struct item_t {
    int q, np;
    T v;
    uint16_t c;
  };
  auto load = [&](int ch) {
    item_t it;
    const int j = ch * 64 + lane;
    it.q = -1;
    it.np = -1;
    it.v = T(0);
    it.c = 0;
    if (j < n) {
      it.q = (int) st[lo + j];
      it.v = stv[lo + j];
      it.c = stc[lo + j];
      const int gid = it.q >> 6;
      int start = j, smaller = 0;
      for (int k = j - 1; k >= 0; --k) {
        const int qk = (int) st[lo + k];
        if ((qk >> 6) != gid)
          break;
        start = k;
        smaller += qk < it.q ? 1 : 0;
      }
      for (int k = j + 1; k < n; ++k) {
        const int qk = (int) st[lo + k];
        if ((qk >> 6) != gid)
          break;
        smaller += qk < it.q ? 1 : 0;
      }
      it.np = start + smaller;
    }
    return it;
  };
  const int nch = (n + 63) >> 6;
  item_t cur = load(0);
  for (int ch = 0; ch < nch; ++ch) {
    item_t nxt = cur;
    if (ch + 1 < nch)
      nxt = load(ch + 1);
    if (cur.np >= 0 && cur.np != ch * 64 + lane) {
      st[lo + cur.np] = (Q) cur.q;
      stv[lo + cur.np] = cur.v;
      stc[lo + cur.np] = cur.c;
    }
    cur = nxt;
  }
}

// enc8 write-out of one staged run by one wavefront.  The run lies in the staging area sorted by source position (= by
// row: positions grow with the row; the ordered staging of pb_scatter_staged_kernel sees to that): emit values /
// columns / source positions in that order and the one-byte row codes, block bases and exceptions.
template <typename T, typename RowOf, typename Q>
__device__ __forceinline__ void pb_emit_sorted_run(int n, int lo, int g, int gp, int p0, int lane, const Q* __restrict__ st,
                                                   const T* __restrict__ stv, const uint16_t* __restrict__ stc, RowOf row_of,
                                                   T* __restrict__ s_val, uint16_t* __restrict__ s_col,
                                                   int32_t* __restrict__ perm, unsigned char* __restrict__ s_code,
                                                   typename pb_hdr<T>::type* __restrict__ s_hdr, int* exc_n, int exc_cap,
                                                   unsigned* __restrict__ exc_idx, uint16_t* __restrict__ exc_row,
                                                   uint16_t* __restrict__ s_src) {
  constexpr int PB_BLK = pb_geom<T>::BLK;
  const int nb = (n + PB_BLK - 1) / PB_BLK;
  for (int j0 = 0; j0 < nb * PB_BLK; j0 += 64) {
    const int j = j0 + lane;
    const bool valid = j < n;
    const int qq = valid ? (int) st[lo + j] : 0;
    const int r = valid ? row_of(qq) : 0;
    const int bi = lane & (PB_BLK - 1);  // j0 is a multiple of 64, 64 a multiple of the block
    // D_j = 255 bi + min over the block's entries k <= j of (r_k - 255 k): the decoder's position after entry j
    int M = valid ? r - 255 * bi : 0x3fffffff;
#pragma unroll
    for (int o = 1; o < PB_BLK; o <<= 1) {
      const int u = __shfl_up(M, o, 64);
      if (bi >= o)
        M = u < M ? u : M;
    }
    const int D = M + 255 * bi;
    const int Dp = __shfl_up(D, 1, 64);
    const bool exc = valid && bi > 0 && r - Dp >= 255;
    if (valid) {
      s_col[g + j] = stc[lo + j];
      if (s_src) {  // value-free tiles: no copy of the value, the source position goes to the REDUCE's stream (16 bits: the
        s_src[gp + j] = (uint16_t) qq;  // bin's window of the caller's array holds < 65 536 entries)
      } else {
        s_val[g + j] = stv[lo + j];
        if (perm)
          perm[g + j] = (int32_t) (p0 + qq);
      }
      s_code[gp + j] = (unsigned char) (bi == 0 ? 0 : (exc ? 255 : r - Dp));
      if (bi == 0)
        s_hdr[(gp + j) / PB_BLK] = pb_hdr<T>::make((unsigned) r, 0u);
    } else if (j < ((n + 3) & ~3)) {  // the run's share of A' ends on a multiple of 4: value 0, column 0, no source position
      s_col[g + j] = 0;               // (written here, so that the plan arrays need no clearing pass: pb_clear_tail)
      if (!s_src) {
        s_val[g + j] = T(0);
        if (perm)
          perm[g + j] = -1;
      }
    }
    if (exc) {
      const int k = atomicAdd(exc_n, 1);
      if (k < exc_cap) {
        exc_idx[k] = (unsigned) (gp + j);
        exc_row[k] = (uint16_t) r;
      }
    }
  }
}

// Q16 (round 5): positions inside the bin, row offsets and the row table are staged as 16-bit words (the build checked that
// no bin's window holds 65 536 entries or more): 8 instead of 10 bytes per staged fp32 entry and 12 KB of tables less --
// cfg2's bins of 48 830 entries go through LDS in three passes instead of five (each pass walks all of the bin's columns)
template <typename T, typename O, bool ENC8, bool Q16>
__global__ __launch_bounds__(PB_STAGE_THREADS) void pb_scatter_staged_kernel(
    int64_t m, const O* __restrict__ rowptr, const int32_t* __restrict__ colind, const T* __restrict__ values, int W,
    int H, int S, int NB, const int32_t* __restrict__ cnt, const int32_t* __restrict__ aoff,
    const int32_t* __restrict__ prel, const int32_t* __restrict__ binblk, T* __restrict__ s_val,
    uint16_t* __restrict__ s_col, uint16_t* __restrict__ s_row, int32_t* __restrict__ perm,
    int32_t* __restrict__ blkdst, int hub_len, int cap, int rt_len, const int32_t* __restrict__ binrow,
    unsigned char* __restrict__ s_code, typename pb_hdr<T>::type* __restrict__ s_hdr, unsigned* __restrict__ exc_idx,
    uint16_t* __restrict__ exc_row, int32_t* __restrict__ exc_cnt, int exc_cap, int32_t* __restrict__ enc_fail,
    const int32_t* __restrict__ eoff, int32_t* __restrict__ blksrc, const int32_t* __restrict__ bin_order,
    uint16_t* __restrict__ s_src, const uint16_t* __restrict__ wpart) {
  constexpr int PB_BLK = pb_geom<T>::BLK;
  __shared__ int exc_n;
  if (ENC8 && threadIdx.x == 0)
    exc_n = 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* lcnt = reinterpret_cast<int*>(smem);  // [S] entries of this bin per slice
  int* gdst = lcnt + S;                      // [S] first A' position of the run
  int* pdst = gdst + S;                      // [S] first P position of the run
  int* lcur = pdst + S;                      // [S] staging cursor (local offset, advanced by the atomics)
  typedef typename std::conditional<Q16, uint16_t, int>::type q_t;
  // enc8: [16 waves][PB_STAGE_SP] entries of the pass per (wave, slice) -> first staging position of that wave's part
  int* wcnt = lcur + S;
  T* stv = reinterpret_cast<T*>(wcnt + (ENC8 ? (PB_STAGE_THREADS / 64) * PB_STAGE_SP : 0));  // [cap] staged entries: value,
  q_t* st = reinterpret_cast<q_t*>(stv + cap);  // [cap] position relative to the bin's first entry,
  uint16_t* stc = reinterpret_cast<uint16_t*>(st + cap);  // [cap] column inside the slice
  q_t* rp = reinterpret_cast<q_t*>(stc + cap);  // [H + 1] the bin's row offsets relative to its first entry
  q_t* rt = rp + H + 1;                         // [rt_len] row of every 64th entry (narrows the row search)
  __shared__ int pass_end, pass_direct;
  // bin_order (row-skewed matrices): the bins heaviest first -- a bin of 131 k entries makes 13 staging passes over all
  // of them, a light one a single pass over 10 k, and a heavy bin that starts last finishes alone
  const int wb = bin_order ? bin_order[blockIdx.x] : (int) blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t r0, r1;
  pb_bin_rows(binrow, wb, H, m, &r0, &r1);
  if (r0 >= m)
    return;
  const int nr = (int) (r1 - r0);
  const O p0 = rowptr[r0], p1 = rowptr[r1];
  const int ne = (int) (p1 - p0);
  if (ne == 0)
    return;  // nothing to place (and p0 may be the end of the arrays: the clamped gathers below need ne > 0)
  const int pb0 = binblk[wb];
  for (int i = tid; i < S; i += PB_STAGE_THREADS) {
    const int64_t key = (int64_t) i * NB + wb;
    const int p0b = pb0 + prel[(int64_t) wb * S + i];
    lcnt[i] = cnt[key];
    gdst[i] = eoff[key];  // first entry of the run in the compact A' stream
    pdst[i] = p0b * PB_BLK;
  }
  for (int i = tid; i <= nr; i += PB_STAGE_THREADS)
    rp[i] = (q_t) (rowptr[r0 + i] - p0);
  __syncthreads();
  // the block map of the bin's runs: one wavefront per run, a lane per block
  for (int sl = wave; sl < S; sl += PB_STAGE_THREADS / 64) {
    const int nb = (lcnt[sl] + PB_BLK - 1) / PB_BLK, a0 = aoff[(int64_t) sl * NB + wb], p0b = pdst[sl] / PB_BLK;
    for (int k = lane; k < nb; k += 64) {
      blkdst[a0 + k] = p0b + k;
      blksrc[a0 + k] = gdst[sl] + k * PB_BLK;
    }
  }
  // row (inside the bin) of the entry at relative position q: last i in [lo, hi) with rp[i] <= q
  auto row_between = [&](int q, int lo, int hi) {
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int) rp[mid] <= q)
        lo = mid;
      else
        hi = mid;
    }
    return lo;
  };
  // rt[k] = row of entry 64k (k beyond the table or the bin: the last row), so that the search for an entry
  // only covers the rows its block of 64 entries spans
  const int nblk = (ne + 63) >> 6;
  const bool use_rt = nblk + 1 <= rt_len;
  if (use_rt) {
    for (int k = tid; k <= nblk; k += PB_STAGE_THREADS)
      rt[k] = (q_t) (k < nblk ? row_between(k << 6, 0, nr) : (nr > 0 ? nr - 1 : 0));
    __syncthreads();
  }
  auto row_of = [&](int q) {
    if (!use_rt)
      return row_between(q, 0, nr);
    const int k = q >> 6;
    return row_between(q, (int) rt[k], (int) rt[k + 1] + 1);
  };
  // slice of column c (exact: float reciprocal + one correction step; S <= 256 slices keep the quotient far inside the
  // float's 24 bits), or -1 for an entry of a hub row that stays out of the tiles
  const float inv_w = 1.0f / (float) W;
  auto slice_of = [&](int q, int c) {
    int sl = (int) ((float) c * inv_w);
    if (sl * W > c)
      --sl;
    else if ((sl + 1) * W <= c)
      ++sl;
    if (hub_len > 0) {
      const int r = row_of(q);
      if ((int) rp[r + 1] - (int) rp[r] > hub_len)
        return -1;
    }
    return sl;
  };
  if (ENC8) {
    // entries per (wave, slice) of the whole bin, then -- per slice -- the exclusive scan over the waves
    constexpr int NW = PB_STAGE_THREADS / 64;
    const int L = (((ne + NW - 1) / NW) + 63) & ~63;
    const int qlo = wave * L, qhi = (qlo + L) < ne ? (qlo + L) : ne;
    for (int i = tid; i < NW * PB_STAGE_SP; i += PB_STAGE_THREADS)
      wcnt[i] = 0;
    __syncthreads();
    constexpr int CU4 = 8;
    if (wpart) {  // counted by pb_count_kernel already (same partition of the bin's positions): 8 KB instead of a walk
      for (int i = tid; i < NW * S; i += PB_STAGE_THREADS)
        wcnt[(i / S) * PB_STAGE_SP + i % S] = (int) wpart[((size_t) wb * NW) * S + i];
    } else
    for (int qb = qlo; qb < qhi; qb += 64 * CU4) {
      int cbuf[CU4];
#pragma unroll
      for (int u = 0; u < CU4; ++u) {
        const int q = qb + 64 * u + lane;
        cbuf[u] = colind[p0 + (q < qhi ? q : qhi - 1)];
      }
#pragma unroll
      for (int u = 0; u < CU4; ++u) {
        const int q = qb + 64 * u + lane;
        const int sl = q < qhi ? slice_of(q, cbuf[u]) : -1;
        if (sl >= 0)
          atomicAdd(&wcnt[wave * PB_STAGE_SP + sl], 1);
      }
    }
    __syncthreads();
    for (int sl = tid; sl < S; sl += PB_STAGE_THREADS) {
      int run = 0;
      for (int w = 0; w < NW; ++w) {
        const int t = wcnt[w * PB_STAGE_SP + sl];
        wcnt[w * PB_STAGE_SP + sl] = run;
        run += t;
      }
    }
    __syncthreads();
  }
  int s0 = 0;
  while (s0 < S) {
    // wave 0: the longest slice range [s0, s1) whose entries fit the staging area, and their local offsets
    if (wave == 0) {
      int total = 0, s1 = s0;
      bool open = true;
      while (open && s1 < S) {
        const int i = s1 + lane;
        const int c = i < S ? lcnt[i] : 0;
        int incl = c;
        for (int o = 1; o < 64; o <<= 1) {
          const int t = __shfl_up(incl, o, 64);
          if (lane >= o)
            incl += t;
        }
        const bool fits = i < S && total + incl <= cap;
        const unsigned long long fm = __ballot(fits);
        // lanes are monotone: the first lane that does not fit ends the range
        const int nfit = fm == ~0ull ? 64 : __builtin_ctzll(~fm);
        if (lane < nfit)
          lcur[i] = total + incl - c;
        total += __shfl(incl, nfit > 0 ? nfit - 1 : 0, 64) * (nfit > 0);
        s1 += nfit;
        open = nfit == 64;
      }
      if (lane == 0) {
        // a single run larger than the staging area goes straight to memory
        pass_direct = s1 == s0;
        if (s1 == s0) {
          lcur[s0] = 0;
          s1 = s0 + 1;
        }
        pass_end = s1;
      }
    }
    __syncthreads();
    const int s1 = pass_end;
    const bool direct = pass_direct != 0;
    if (ENC8 && !direct) {
      // Ordered staging (the one-byte row codes want every run sorted by row = by source position).  Wave w owns the
      // contiguous positions [w L, (w + 1) L) of the bin (L a multiple of 64) and walks them in order, 64 at a time;
      // the per-(wave, slice) counts of the whole bin were taken once, before the passes (wcnt: the offset of the wave's
      // part inside every run), so a pass only places: position = run start + the wave's offset + a wave-private LDS
      // cursor.  Entries of one wave and one slice are staged round by round in position order; only the lanes of ONE
      // round that share a slice may come out permuted -- pb_sort_rounds puts those few back before the run is emitted.
      // (Sorting whole runs by rank cost 2.1 ms of a 3.5 ms scatter at cfg2; a ballot match per round and pass with a
      // second counting sweep per pass 1.6 ms and 6.0 instead of 3.3 GB of reads.)
      constexpr int NW = PB_STAGE_THREADS / 64;
      const int L = (((ne + NW - 1) / NW) + 63) & ~63;
      const int qlo = wave * L, qhi = (qlo + L) < ne ? (qlo + L) : ne;
      int* mycur = wcnt + wave * PB_STAGE_SP;
      for (int i = tid; i < (s1 - s0) * NW; i += PB_STAGE_THREADS) {  // cursors of this pass: run start + wave offset
        const int w = i / (s1 - s0), sl = s0 + i % (s1 - s0);
        wcnt[w * PB_STAGE_SP + sl] += lcur[sl];
      }
      __syncthreads();
      constexpr int CU4 = 8;  // rounds whose loads are issued together
      // Columns AND values of a batch of CU4 rounds are loaded together, unconditionally, one batch ahead of the batch being
      // placed (two register sets): nothing the placing loop waits for is a round trip to memory.  (Round 5: the value load
      // sat inside the placing branch and was waited for round by round -- 48 dependent round trips per wavefront and pass,
      // most of the 2.2 ms of this kernel at cfg2; with predicated loads batched behind the columns 1.42 ms; the price of
      // loading every value in every pass is 0.8 GB of coalesced reads.)
      auto load_batch = [&](int qb, int (&cb)[CU4], T (&vb)[CU4]) {
#pragma unroll
        for (int u = 0; u < CU4; ++u) {
          const int q = qb + 64 * u + lane;
          const int qq = q < qhi ? q : (qhi > qlo ? qhi - 1 : qlo < ne ? qlo : ne - 1);
          cb[u] = colind[p0 + qq];
          if (!s_src)
            vb[u] = values[p0 + qq];
        }
      };
      auto place_batch = [&](int qb, const int (&cb)[CU4], const T (&vb)[CU4]) {
#pragma unroll
        for (int u = 0; u < CU4; ++u) {
          const int q = qb + 64 * u + lane;
          const int sl = q < qhi ? slice_of(q, cb[u]) : -1;
          if (sl >= s0 && sl < s1) {
            const int pos = atomicAdd(&mycur[sl], 1);
            st[pos] = (q_t) q;
            if (!s_src)
              stv[pos] = vb[u];
            stc[pos] = (uint16_t) (cb[u] - sl * W);
          }
        }
      };
      int c_a[CU4], c_b[CU4];
      T v_a[CU4], v_b[CU4];
      load_batch(qlo, c_a, v_a);
      for (int qb = qlo; qb < qhi; qb += 2 * 64 * CU4) {
        load_batch(qb + 64 * CU4, c_b, v_b);
        place_batch(qb, c_a, v_a);
        load_batch(qb + 2 * 64 * CU4, c_a, v_a);
        place_batch(qb + 64 * CU4, c_b, v_b);
      }
      __syncthreads();
    } else {
    // eight column loads per thread are issued before the first is used: with one workgroup per CU the
    // loop is bound by load latency, not bandwidth
    constexpr int LU = 8;
    for (int qb = tid; qb < ne; qb += LU * PB_STAGE_THREADS) {
      int cbuf[LU];
#pragma unroll
      for (int u = 0; u < LU; ++u) {
        const int qq = qb + u * PB_STAGE_THREADS;
        cbuf[u] = colind[p0 + (qq < ne ? qq : ne - 1)];
      }
#pragma unroll
      for (int u = 0; u < LU; ++u) {
      const int q = qb + u * PB_STAGE_THREADS;
      if (q >= ne)
        break;
      const int c = cbuf[u];
      const int sl = c / W;
      if (sl < s0 || sl >= s1)
        continue;
      int r = -1;
      if (hub_len > 0 || direct) {
        r = row_of(q);
        if (hub_len > 0 && (int) rp[r + 1] - (int) rp[r] > hub_len)
          continue;
      }
      const int pos = atomicAdd(&lcur[sl], 1);
      if (direct) {
        if (ENC8)
          continue;  // a run larger than the staging area cannot be sorted here: the plan falls back (flag below)
        const int i = gdst[sl] + pos;
        s_col[i] = (uint16_t) (c - sl * W);
        if (s_src) {
          s_src[pdst[sl] + pos] = (uint16_t) q;
        } else {
          s_val[i] = values[p0 + q];
          if (perm)
            perm[i] = (int32_t) (p0 + q);
        }
        s_row[pdst[sl] + pos] = (uint16_t) r;
      } else {
        st[pos] = (q_t) q;
        if (!s_src)
          stv[pos] = values[p0 + q];  // neighbouring threads: neighbouring addresses
        stc[pos] = (uint16_t) (c - sl * W);
      }
      }
    }
    __syncthreads();
    }
    if (ENC8 && direct && tid == 0)
      atomicExch(enc_fail, 1);
    if (!ENC8 && direct && tid < 3) {  // the run that went straight to memory: pads of its last quad
      const int n = lcnt[s0];
      if (tid < ((n + 3) & ~3) - n) {
        s_col[gdst[s0] + n + tid] = 0;
        if (!s_src) {
          s_val[gdst[s0] + n + tid] = T(0);
          if (perm)
            perm[gdst[s0] + n + tid] = -1;
        }
      }
    }
    if (!direct) {
      // every wave writes whole runs from the staging area (contiguous stores, no global gathers: fetching
      // value and column again by position cost 2.4 of the kernel's 2.75 ms -- random 4-byte reads, even
      // L2 hits, run at ~10 cycles per request and CU)
      constexpr int NW = PB_STAGE_THREADS / 64;
      for (int sl = s0 + wave; sl < s1; sl += NW) {
        const int n = lcnt[sl], lo = ENC8 ? lcur[sl] : lcur[sl] - n, g = gdst[sl], gp = pdst[sl];
        if (ENC8) {
          unsigned* ei = exc_idx + (size_t) wb * exc_cap;
          uint16_t* er = exc_row + (size_t) wb * exc_cap;
#ifndef PB_EXP_NOSORT  // A/B only: what does the in-place sort of the rounds cost? (results wrong without it)
          if (n > 1)
            pb_sort_rounds<T>(n, lo, lane, st, stv, stc);
#endif
          if (n > 0)
            pb_emit_sorted_run<T>(n, lo, g, gp, (int) p0, lane, st, stv, stc, row_of, s_val, s_col, perm, s_code, s_hdr,
                                  &exc_n, exc_cap, ei, er, s_src);
          continue;
        }
        for (int j = lane; j < n; j += 64) {
          const int q = (int) st[lo + j];
          s_col[g + j] = stc[lo + j];
          if (s_src) {
            s_src[gp + j] = (uint16_t) q;
          } else {
            s_val[g + j] = stv[lo + j];
            if (perm)
              perm[g + j] = (int32_t) (p0 + q);
          }
          s_row[gp + j] = (uint16_t) row_of(q);
        }
        if (lane < ((n + 3) & ~3) - n) {  // pads of the run's last quad (see pb_emit_sorted_run)
          s_col[g + n + lane] = 0;
          if (!s_src) {
            s_val[g + n + lane] = T(0);
            if (perm)
              perm[g + n + lane] = -1;
          }
        }
      }
    }
    // end of the pass: the next one reuses the LDS staging area, so LDS traffic must be complete -- but not
    // the global stores above; __syncthreads() would also wait for those (vmcnt(0)) in every wave
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    s0 = s1;
  }
  if (ENC8 && tid == 0) {  // (every wave's exception appends are LDS operations completed before the last barrier)
    const int k = exc_n;
    exc_cnt[wb] = k < exc_cap ? k : exc_cap;
    if (k > exc_cap)
      atomicExch(enc_fail, 1);
  }
}

// balance probe: entries per slice and per bin group (RW bins = one reduce workgroup).  Workgroups
// [0, S) sum one slice each (a contiguous row of NB counters), workgroups [S, S + ngroups) one bin group
// each (RW counters out of every slice's row); no atomics (the first version added every counter to its
// group with a global atomic: 0.59 ms at cfg2).
__global__ __launch_bounds__(256) void pb_balance_kernel(int S, int NB, int RW, const int32_t* __restrict__ cnt,
                                                         unsigned long long* __restrict__ slice_sum,
                                                         unsigned long long* __restrict__ slice_ne,
                                                         unsigned long long* __restrict__ group_sum,
                                                         unsigned long long* __restrict__ slice_max) {
  __shared__ unsigned long long red[4];
  __shared__ unsigned long long red_ne[4];
  __shared__ unsigned long long red_mx[4];
  unsigned long long tot = 0, ne = 0, mx = 0;  // ne: non-empty (slice, bin) tiles -- the locality probe; mx: longest run
  if ((int) blockIdx.x < S) {
    const int sl = blockIdx.x;
    for (int b = threadIdx.x; b < NB; b += 256) {
      const unsigned long long c = (unsigned long long) cnt[(int64_t) sl * NB + b];
      tot += c;
      ne += c != 0;
      mx = c > mx ? c : mx;
    }
  } else {
    const int64_t g = (int64_t) blockIdx.x - S;
    const int64_t b0 = g * RW;
    const int nb = (int) ((b0 + RW) <= NB ? RW : (NB - b0));
    for (int i = threadIdx.x; i < S * nb; i += 256)
      tot += (unsigned long long) cnt[(int64_t) (i / nb) * NB + b0 + (i % nb)];
  }
  tot = group_sum_c<64>(tot);
  ne = group_sum_c<64>(ne);
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long u = __shfl_xor(mx, o, 64);
    mx = u > mx ? u : mx;
  }
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = tot;
    red_ne[threadIdx.x >> 6] = ne;
    red_mx[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = red[0] + red[1] + red[2] + red[3];
    if ((int) blockIdx.x < S) {
      slice_sum[blockIdx.x] = t;
      slice_ne[blockIdx.x] = red_ne[0] + red_ne[1] + red_ne[2] + red_ne[3];
      slice_max[blockIdx.x] = std::max(std::max(red_mx[0], red_mx[1]), std::max(red_mx[2], red_mx[3]));
    } else {
      group_sum[blockIdx.x - S] = t;
    }
  }
}


// s_val[i] = values[perm[i]] over the A'-order array; pads (perm < 0) stay 0
template <typename T>
__global__ __launch_bounds__(256) void pb_update_values_kernel(int64_t n_pad, const int32_t* __restrict__ perm,
                                                               const T* __restrict__ values, T* __restrict__ s_val) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i < n_pad) {
    const int32_t p = stream_load(perm + i);
    s_val[i] = p >= 0 ? values[p] : T(0);
  }
}

// runs[b * S + s] = (first entry, padded length) of run (slice s, bin b): the slice-major offsets turned bin-major
__global__ __launch_bounds__(256) void pb_run_table_kernel(int S, int64_t NB, const int32_t* __restrict__ eoff,
                                                           int2* __restrict__ runs) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;  // = b * S + s
  if (i >= (int64_t) S * NB)
    return;
  const int64_t b = i / S, sl = i % S, key = sl * NB + b;
  const int g = eoff[key];
  runs[i] = make_int2(g, eoff[key + 1] - g);
}

// The same refresh, bin by bin (end of round 4).  The kernel above walks the A' arrays in their own order -- slice-major --
// so the 200 entries of a run gather from a window of the caller's array that every one of the S slices fetches again:
// 12.8 GB of lines for 0.4 GB of values, 1.72 ms at cfg2.  All runs of ONE wave-bin draw from one window (the entries of
// the bin's rows: ~195 KB at cfg2), so a workgroup per bin loads that window into LDS (in passes of what LDS holds) and
// walks the bin's S runs: perm in (coalesced), value out of LDS, s_val out (coalesced).  The pads (perm = -1) stay 0.
// row0[b] / rp[]: first row of every bin and the row offsets IN THE CALLER'S ARRAY (the compacted row map of a plan that has
// one: its offsets are source positions too).
template <typename T, typename O>
__global__ __launch_bounds__(1024) void pb_refresh_bins_kernel(int S, int64_t NB, int H, int64_t rows,
                                                               const int32_t* __restrict__ binrow, const O* __restrict__ rp,
                                                               const int2* __restrict__ runs,
                                                               const int32_t* __restrict__ perm, const T* __restrict__ values,
                                                               T* __restrict__ s_val, int cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int2* lrun = reinterpret_cast<int2*>(smem);                    // [S] the bin's runs (one coalesced read)
  T* win = reinterpret_cast<T*>(smem + (((size_t) S * 8 + 15) & ~(size_t) 15));
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = 16, U = 4;
  const int64_t r0 = binrow ? (int64_t) binrow[b] : b * H;
  int64_t r1 = binrow ? (int64_t) binrow[b + 1] : (b + 1) * H;
  r1 = r1 < rows ? r1 : rows;
  if (r0 >= r1)
    return;
  constexpr int VE = 16 / (int) sizeof(T);
  const bool vec_ok = (reinterpret_cast<uintptr_t>(values) & 15) == 0;  // (cap is a multiple of VE)
  const int64_t p_lo = vec_ok ? (int64_t) rp[r0] & ~(int64_t) (VE - 1) : (int64_t) rp[r0], p_hi = (int64_t) rp[r1];
  for (int i = tid; i < S; i += 1024)
    lrun[i] = runs[b * S + i];
  for (int64_t base = p_lo; base < p_hi; base += cap) {
    const int wn = (int) ((p_hi - base) < cap ? (p_hi - base) : cap);
    __syncthreads();  // everyone is done with the previous window
    if (vec_ok) {  // 16-byte loads: the window starts on a multiple of VE entries of a 16-byte aligned array
      typedef T vec_t __attribute__((ext_vector_type(VE)));
      const int nv = wn / VE;
      for (int i = tid; i < nv; i += 1024)
        reinterpret_cast<vec_t*>(win)[i] = stream_load(reinterpret_cast<const vec_t*>(values + base) + i);
      for (int i = nv * VE + tid; i < wn; i += 1024)
        win[i] = stream_load(values + base + i);
    } else {
      for (int i = tid; i < wn; i += 1024)
        win[i] = stream_load(values + base + i);
    }
    __syncthreads();
    // wave w takes the runs of slices 4 w .. 4 w + 3, then 64 further on: the source positions of FOUR runs are loaded
    // together (a run is ~200 entries at cfg2: one round of a wavefront; one run at a time is a chain of run offsets ->
    // positions -> store per run: 0.64 ms at cfg2 against 0.50 with four in flight, 0.45 with 16-byte window loads, 0.38
    // with the bin's run table read once into LDS instead of two strided 4-byte loads per run)
    constexpr int R = 4;
    for (int s0 = wave * R; s0 < S; s0 += NW * R) {
      int g[R], n[R];
      int32_t pp[R][U];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        g[r] = 0;
        n[r] = 0;
        if (s0 + r < S) {
          const int2 rr = lrun[s0 + r];
          g[r] = rr.x;
          n[r] = rr.y;
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int u = 0; u < U; ++u)
          pp[r][u] = lane + 64 * u < n[r] ? stream_load(perm + g[r] + lane + 64 * u) : -1;
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t d = (int64_t) pp[r][u] - base;
          if (pp[r][u] >= 0 && d >= 0 && d < wn)
            s_val[g[r] + lane + 64 * u] = win[d];
        }
        for (int j = 64 * U + lane; j < n[r]; j += 64) {  // (a run longer than 256 entries: the rest one round at a time)
          const int32_t pq = stream_load(perm + g[r] + j);
          const int64_t d = (int64_t) pq - base;
          if (pq >= 0 && d >= 0 && d < wn)
            s_val[g[r] + j] = win[d];
        }
      }
    }
  }
}

// ---- execute ---------------------------------------------------------------------------
template <typename T>
struct pack4;
template <>
struct pack4<float> {
  static __device__ __forceinline__ void load_cached(const float* p, float (&o)[4]) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  static __device__ __forceinline__ void load(const float* p, float (&o)[4]) {
    const f32x4 v = stream_load(reinterpret_cast<const f32x4*>(p));
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&o)[4]) {
    f32x4 v;
    v.x = o[0]; v.y = o[1]; v.z = o[2]; v.w = o[3];
#ifdef PB_EXP_NT_STORE  // A/B only (tools/build_variant.sh)
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
    *reinterpret_cast<f32x4*>(p) = v;  // plain store: 12 % faster than nt here (measured)
#endif
  }
  static __device__ __forceinline__ void store_nt(float* p, const float (&o)[4]) {
    f32x4 v;
    v.x = o[0]; v.y = o[1]; v.z = o[2]; v.w = o[3];
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  }
};
template <>
struct pack4<double> {
  static __device__ __forceinline__ void load_cached(const double* p, double (&o)[4]) {
    const f64x2 a = reinterpret_cast<const f64x2*>(p)[0];
    const f64x2 b = reinterpret_cast<const f64x2*>(p)[1];
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
  }
  static __device__ __forceinline__ void load(const double* p, double (&o)[4]) {
    const f64x2 a = stream_load(reinterpret_cast<const f64x2*>(p));
    const f64x2 b = stream_load(reinterpret_cast<const f64x2*>(p) + 1);
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
  }
  static __device__ __forceinline__ void store(double* p, const double (&o)[4]) {
    f64x2 a, b;
    a.x = o[0]; a.y = o[1]; b.x = o[2]; b.y = o[3];
    reinterpret_cast<f64x2*>(p)[0] = a;
    reinterpret_cast<f64x2*>(p)[1] = b;
  }
  static __device__ __forceinline__ void store_nt(double* p, const double (&o)[4]) {
    f64x2 a, b;
    a.x = o[0]; a.y = o[1]; b.x = o[2]; b.y = o[3];
    __builtin_nontemporal_store(a, reinterpret_cast<f64x2*>(p));
    __builtin_nontemporal_store(b, reinterpret_cast<f64x2*>(p) + 1);
  }
};

typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

// expand: P[blkdst[blk]] = s_val[blk] * x[slice_base + s_col[blk]] over the slice's contiguous blocks of A'.
// The x slice lives in LDS.  Eight consecutive lanes own one block of 32 entries (4 entries = 16 B of values,
// 8 B of columns each): the wavefront reads 8 consecutive blocks -- 1 KiB of values -- and stores 8 whole
// 128-byte lines of products, each at the place the reduce kernel's stream wants it.
// NT: the products are stored with the non-temporal hint.  Which flavour is faster depends on the BOX (round 3,
// tools/exp_r03o.sh, same binaries): where the reduce pays for the expand's write-backs (reduce 118-122 us after plain
// stores) the hint moves that cost into the expand and the pair gains 1-3 %; where it does not (reduce 105-107 us) the hint
// costs 3 %.  Default plain; a handle can ask for a timed trial at inspect (SPBLAS_GFX950_OPT_STORE_TRIAL, spmv.hip: store_trial).
// what the expand of a chunked multi-GPU step waits for (flags == nullptr: an ordinary expand)
struct pb_chunk_wait {
  const long long* flags;
  const long long* chunk_rows;
  int n_ranks, chunks, rank;
  long long step, timeout_ticks;
  int* status_dev;
};

// VF (value-free tiles, round 5): the plan holds no values -- the "product" is the gathered x[col] alone (2 B read, 4 B
// written per entry); the reduce multiplies by the caller's values (pb_reduce_vf_kernel)
template <typename T, bool NT, bool VF = false>
__global__ __launch_bounds__(PB_THREADS) void pb_expand_kernel(int64_t n, int W, const int32_t* __restrict__ sliceblk,
                                                               const T* __restrict__ s_val,
                                                               const uint16_t* __restrict__ s_col,
                                                               const int32_t* __restrict__ blkdst,
                                                               const T* __restrict__ x, T* __restrict__ P,
                                                               const int4* __restrict__ items, int S, int share,
                                                               const int32_t* __restrict__ blksrc, int n_items, int rot,
                                                               pb_chunk_wait cw) {
  constexpr int PB_BLK = pb_geom<T>::BLK, LPB = PB_BLK / 4;  // lanes per block: 8 (fp32) / 4 (fp64)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* xs = reinterpret_cast<T*>(smem);
  const int tid = threadIdx.x;
  const int sub = (tid & (LPB - 1)) * 4;  // first of this lane's 4 entries inside its block
  const int bsel = tid / LPB;             // block inside the workgroup's pass of PB_THREADS / LPB blocks
  constexpr int PASS = PB_THREADS / LPB;
  // Chunked multi-GPU step (cw.flags != nullptr): x IS the previous step's y, row-sharded over cw.n_ranks ranks, and the
  // peers deliver their rows chunk by chunk (peer stores + one flag per (rank, chunk) after the chunk's stores).  Before a
  // workgroup loads an x slice it waits for exactly the chunks that slice is made of -- bounded, with s_sleep -- so the
  // links drain behind this kernel instead of in front of it; an acquire fence then drops stale copies of the slice's lines.
  auto wait_slice = [&](int s) {
    if (!cw.flags)
      return;
    if (tid == 0) {
      const long long c0 = (long long) s * W, c1 = (c0 + W) < (long long) n ? (c0 + W) : (long long) n;
      const long long t0 = wall_clock64();
      bool late = false;
      for (int q = 0; q < cw.n_ranks && !late; ++q) {
        if (q == cw.rank)
          continue;  // my own rows: stream order
        const long long* cr = cw.chunk_rows + (long long) q * (cw.chunks + 1);
        if (cr[cw.chunks] <= c0 || cr[0] >= c1)
          continue;
        for (int c = 0; c < cw.chunks && !late; ++c) {
          if (cr[c + 1] <= c0 || cr[c] >= c1 || cr[c + 1] <= cr[c])
            continue;
          const long long* f = cw.flags + (long long) q * cw.chunks + c;
          while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < cw.step) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > cw.timeout_ticks) {
              cw.status_dev[0] = 1;
              late = true;
              break;
            }
          }
        }
      }
      const long long waited = wall_clock64() - t0;
      if (waited > 200)  // (2 us at 100 MHz: not every workgroup needs to touch the counter)
        atomicMax(cw.status_dev + 1, (int) (waited < 0x7fffffffLL ? waited : 0x7fffffffLL));
    }
    __syncthreads();
#ifdef PB_CHUNK_ACQ_FENCE
    // the slice's lines were written by other devices after this kernel started: drop what this XCD's caches hold of
    // them (an invalidate, not a write-back), then load the slice with the ordinary 16-byte loads
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
#endif
  };
  // ... and ONE workgroup, at its end, waits for EVERY chunk of every peer, whatever the sparsity pattern.  The reduce /
  // combine that follows this kernel on the stream peer-stores the rows of step k + 1 into the buffer the peers' expand of
  // step k is reading; that is only safe once every peer has published step k (its publish follows its expand on its
  // stream).  The per-slice waits above cover exactly the peers in whose column ranges this shard has entries -- a block
  // triangular matrix, an upwind stencil or a directed graph leaves peers out, and a rank that never waits for a peer can
  // run steps ahead of it (round-4 advisor finding).  Whole-step order per peer, bounded like the slice waits.
  auto wait_all_peers = [&]() {
    if (!cw.flags || blockIdx.x != 0 || tid != 0)
      return;
    const long long t0 = wall_clock64();
    for (int q = 0; q < cw.n_ranks; ++q) {
      if (q == cw.rank)
        continue;
      const long long* cr = cw.chunk_rows + (long long) q * (cw.chunks + 1);
      for (int c = 0; c < cw.chunks; ++c) {
        if (cr[c + 1] <= cr[c])
          continue;  // an empty chunk is never published on its own
        const long long* f = cw.flags + (long long) q * cw.chunks + c;
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < cw.step) {
          __builtin_amdgcn_s_sleep(8);
          if (wall_clock64() - t0 > cw.timeout_ticks) {
            cw.status_dev[0] = 1;
            return;
          }
        }
      }
    }
    const long long waited = wall_clock64() - t0;
    if (waited > 200)
      atomicMax(cw.status_dev + 1, (int) (waited < 0x7fffffffLL ? waited : 0x7fffffffLL));
  };
  auto load_x = [&](int s) {
    const int64_t c0 = (int64_t) s * W;
    const int cw_ = (int) ((n - c0) < W ? (n - c0) : W);
#ifndef PB_CHUNK_ACQ_FENCE  // (A/B: an acquire fence + ordinary loads instead: 141 against 117 us for the shard step)
    if (cw.flags) {
      for (int i = tid; i < cw_; i += PB_THREADS)
        xs[i] = __hip_atomic_load(x + c0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
#endif
    // 16-byte lane loads when the slice starts on a 16-byte boundary (W is a multiple of 4; x itself usually is aligned):
    // a quarter of the load and LDS-store instructions of the 160 KiB fill
    constexpr int VE = 16 / (int) sizeof(T);
    if ((reinterpret_cast<uintptr_t>(x + c0) & 15) == 0) {
      typedef T vec_t __attribute__((ext_vector_type(VE)));
      const int nv = cw_ / VE;
      for (int i = tid; i < nv; i += PB_THREADS)
        reinterpret_cast<vec_t*>(xs)[i] = reinterpret_cast<const vec_t*>(x + c0)[i];
      for (int i = nv * VE + tid; i < cw_; i += PB_THREADS)
        xs[i] = x[c0 + i];
      return;
    }
    for (int i = tid; i < cw_; i += PB_THREADS)
      xs[i] = x[c0 + i];
  };
  // blocks [b0, b1) of the slice whose x values are in LDS; two passes in flight.  nt_tag (A/B builds only): products stored
  // non-temporally
  auto process_impl = [&](auto nt_tag, int b0, int b1) {
    constexpr bool nt_store = decltype(nt_tag)::value;
    auto put = [&](T* ptr, const T (&val)[4]) {
      if constexpr (nt_store)
        pack4<T>::store_nt(ptr, val);
      else
        pack4<T>::store(ptr, val);
    };
    int blk = b0 + bsel;
    // A' is compact (round 3): the entries of block k start at blksrc[k], a multiple of 4; the lanes past the end of a
    // run's last block read the entries that follow it -- their products land on pad positions of P, which the reduce
    // leaves out (row = H / code 255)
    for (; blk + PASS < b1; blk += 2 * PASS) {
      const int ea = stream_load(blksrc + blk) + sub, eb = stream_load(blksrc + blk + PASS) + sub;
      T va[4], vb[4], pa[4], pb[4];
#ifdef PB_EXP_PLAIN_A  // A/B only: cached loads of the (now unaligned) A' blocks
      pack4<T>::load_cached(s_val + ea, va);
      pack4<T>::load_cached(s_val + eb, vb);
      const u16x4 ca = *reinterpret_cast<const u16x4*>(s_col + ea);
      const u16x4 cb = *reinterpret_cast<const u16x4*>(s_col + eb);
#else
      if constexpr (!VF) {
        pack4<T>::load(s_val + ea, va);
        pack4<T>::load(s_val + eb, vb);
      }
      const u16x4 ca = stream_load(reinterpret_cast<const u16x4*>(s_col + ea));
      const u16x4 cb = stream_load(reinterpret_cast<const u16x4*>(s_col + eb));
#endif
      const int da = stream_load(blkdst + blk), db = stream_load(blkdst + blk + PASS);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (VF) {
          pa[j] = xs[ca[j]];
          pb[j] = xs[cb[j]];
        } else {
          pa[j] = va[j] * xs[ca[j]];
          pb[j] = vb[j] * xs[cb[j]];
        }
      }
      put(P + (int64_t) da * PB_BLK + sub, pa);
      put(P + (int64_t) db * PB_BLK + sub, pb);
    }
    for (; blk < b1; blk += PASS) {
      const int ea = stream_load(blksrc + blk) + sub;
      T va[4], pa[4];
      if constexpr (!VF)
        pack4<T>::load(s_val + ea, va);
      const u16x4 ca = stream_load(reinterpret_cast<const u16x4*>(s_col + ea));
      const int da = stream_load(blkdst + blk);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (VF)
          pa[j] = xs[ca[j]];
        else
          pa[j] = va[j] * xs[ca[j]];
      }
      put(P + (int64_t) da * PB_BLK + sub, pa);
    }
  };
  auto process = [&](int b0, int b1) {
#ifdef PB_EXP_NT_TAIL_PCT  // A/B only: the last N % of a workgroup's blocks are stored non-temporally (do the write-backs that
    // the reduce runs into come from the END of the expand?  tools/exp_r03n.sh: no -- 10 / 25 / 50 % buy 2-3 us, all of them 5)
    const int nt_from = b1 - (int) ((long long) (b1 - b0) * PB_EXP_NT_TAIL_PCT / 100);
    process_impl(std::false_type{}, b0, nt_from);
    process_impl(std::true_type{}, nt_from, b1);
#else
    process_impl(std::integral_constant<bool, NT>{}, b0, b1);
#endif
  };
  if (items) {
    // items (column-skewed matrices): workgroup i takes the blocks [items[i].y, items[i].z) of slice items[i].x,
    // so that a slice holding a large share of the matrix is spread over proportionally many workgroups
    // (rot: the chunked multi-GPU step starts every rank on its OWN slices -- nothing to wait for -- and goes round the
    // ranks from there; a grid smaller than the list walks it in strides)
    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
      int idx = it + rot;
      idx = idx >= n_items ? idx - n_items : idx;
      const int4 item = items[idx];
      if (it != (int) blockIdx.x)
        __syncthreads();  // everyone is done with the previous x slice
      wait_slice(item.x);
      load_x(item.x);
      __syncthreads();
      process(item.y, item.z);
    }
    wait_all_peers();
    return;
  }
  // Equal shares: workgroup b takes the blocks [b*share, (b+1)*share) of A' and loads the x slice of every
  // slice its range touches -- one or two for the usual case of about one slice per workgroup.  The grid is
  // exactly one wave of workgroups whatever the slice count is (slices cut for LDS capacity rarely come in
  // multiples of 512; a second, nearly empty round of whole-slice workgroups cost up to 2x).
  const int total = sliceblk[S];
  int bid = (int) blockIdx.x + rot;
  bid = bid >= (int) gridDim.x ? bid - (int) gridDim.x : bid;
  const long long g0l = (long long) bid * share;
  int g0 = g0l < total ? (int) g0l : total;
  const int g1 = (total - g0) < share ? total : g0 + share;
  if (g0 >= g1) {
    wait_all_peers();
    return;
  }
  int lo = 0, hi = S;  // last slice starting at or before g0
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (sliceblk[mid] <= g0)
      lo = mid;
    else
      hi = mid;
  }
  for (int s = lo; g0 < g1 && s < S; ++s) {
    const int slice_end = sliceblk[s + 1];
    if (slice_end <= g0)
      continue;  // empty slice
    const int a1 = g1 < slice_end ? g1 : slice_end;
    __syncthreads();  // everyone is done with the previous x slice
    wait_slice(s);
    load_x(s);
    __syncthreads();
    process(g0, a1);
    g0 = a1;
  }
  wait_all_peers();
}

// inspect: mark duplicates.  Same walk as the reduce kernel: wave-bin -> groups of 256 entries; lane l holds the
// entries 4l .. 4l+3 of the group.  The reduce kernel issues the LDS reads of a whole group before the first
// write, so of all entries of a group that share a row exactly one stays plain; the others get the duplicate
// flag (bit 15) and take the atomic path there.  tag[row] = id (0..255) of the last entry that claimed the row.
__global__ __launch_bounds__(PB_FTHREADS) void pb_flag_dups_kernel(int Hw, int64_t NBw,
                                                                   const int32_t* __restrict__ binblk,
                                                                   uint16_t* __restrict__ s_row, int PB_GBLK) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned char* tag = smem + (size_t) wave * (size_t) ((Hw + 63) & ~63);
  const int64_t wb = (int64_t) blockIdx.x * PB_FWAVES + wave;
  if (wb >= NBw)
    return;
  const int g0 = binblk[wb] / PB_GBLK, g1 = binblk[wb + 1] / PB_GBLK;
  if (g0 >= g1)
    return;
  u16x4 nxt = *reinterpret_cast<const u16x4*>(s_row + (int64_t) g0 * PB_GRP + 4 * lane);
  for (int g = g0; g < g1; ++g) {
    u16x4* ptr = reinterpret_cast<u16x4*>(s_row + (int64_t) g * PB_GRP + 4 * lane);
    u16x4 r = nxt;
    if (g + 1 < g1)
      nxt = *reinterpret_cast<const u16x4*>(s_row + (int64_t) (g + 1) * PB_GRP + 4 * lane);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((int) r[j] < Hw)  // pads carry row = Hw and never collide
        tag[r[j]] = (unsigned char) (4 * lane + j);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the tag stores have landed
    bool changed = false;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((int) r[j] < Hw && tag[r[j]] != (unsigned char) (4 * lane + j)) {
        r[j] = (unsigned short) (r[j] | 0x8000);
        changed = true;
      }
    if (changed)
      *ptr = r;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // the tag reads are done before the next group's stores
  }
}

// The same for the one-byte row codes: rows come from pb_decode_rows, entries with code 255 (pads, exceptions) take no
// part, and the four flag bits of a lane are gathered into the block's header word.
template <typename T>
__global__ __launch_bounds__(PB_FTHREADS) void pb_flag_dups8_kernel(int Hw, int64_t NBw,
                                                                    const int32_t* __restrict__ binblk,
                                                                    const unsigned char* __restrict__ s_code,
                                                                    typename pb_hdr<T>::type* __restrict__ s_hdr) {
  constexpr int PB_BLK = pb_geom<T>::BLK, PB_GBLK = pb_geom<T>::GBLK, LPB = PB_BLK / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned char* tag = smem + (size_t) wave * (size_t) ((Hw + 64 + 63) & ~63);
  const int64_t wb = (int64_t) blockIdx.x * PB_FWAVES + wave;
  if (wb >= NBw)
    return;
  const int g0 = binblk[wb] / PB_GBLK, g1 = binblk[wb + 1] / PB_GBLK;
  if (g0 >= g1)
    return;
  unsigned cw_n = *reinterpret_cast<const unsigned*>(s_code + (int64_t) g0 * PB_GRP + 4 * lane);
  typename pb_hdr<T>::type hd_n = s_hdr[(int64_t) g0 * PB_GBLK + lane / LPB];
  for (int g = g0; g < g1; ++g) {
    // (the next group's words are on their way while this one goes through the tag array: one wave per bin and a load ->
    // LDS -> store chain per group made this kernel 190 us at cfg2)
    const unsigned cw = cw_n;
    const int64_t blk = (int64_t) g * PB_GBLK + lane / LPB;
    const typename pb_hdr<T>::type hd = hd_n;
    const int gn = g + 1 < g1 ? g + 1 : g;
    cw_n = *reinterpret_cast<const unsigned*>(s_code + (int64_t) gn * PB_GRP + 4 * lane);
    hd_n = s_hdr[(int64_t) gn * PB_GBLK + lane / LPB];
    unsigned row[4];
    bool skip[4];
    pb_decode_rows<LPB>(cw, pb_hdr<T>::base(hd), lane, (unsigned) Hw, row, skip);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (!skip[j])
        tag[row[j]] = (unsigned char) (4 * lane + j);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the tag stores have landed
    unsigned fl = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (!skip[j] && tag[row[j]] != (unsigned char) (4 * lane + j))
        fl |= 1u << j;
    fl <<= 4 * (lane & (LPB - 1));
#pragma unroll
    for (int o = 1; o < LPB; o <<= 1)
      fl |= (unsigned) __shfl_xor((int) fl, o, 64);
    if ((lane & (LPB - 1)) == 0 && fl != 0)
      s_hdr[blk] = pb_hdr<T>::make(pb_hdr<T>::base(hd), fl);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // the tag reads are done before the next group's stores
  }
}

// (Duplicates are added with the native LDS float atomic.  A compare-and-swap loop -- 12x the rate per operation,
// tools/ubench/lds_atomic.hip -- was tried in its place: 10-20 % faster where two or three lanes share a row, 4x slower
// where dozens do (R-MAT with whole rows: 3.5 vs 0.8 ms), 35 % slower for fp64, and the larger kernel cost cfg2 6 %.)
// reduce: RW wavefronts per workgroup, one wave-bin each.  The bin's entries are ONE contiguous stream of
// groups (256 entries: lane l holds entries 4l..4l+3 -- a 16-byte product load and an 8-byte row load).
// The loop is software-pipelined by hand: the loads of batch k+1 (UB groups) are issued before batch k is
// applied; loads past the end of the stream are clamped to its last group (same lines, no extra traffic)
// because a load inside a divergent branch makes the compiler drain vmcnt(0) at the join.
//   item (ritems != nullptr: row-skewed matrices; else blockIdx): .x = bin group, .y / .z = this workgroup reduces
//   part y of z equal parts of every wave-bin's stream, .w >= 0 = offset of its partial sums (RW*Hw values,
//   pb_combine_items_kernel adds them), .w < 0 = the group is not split and y is written directly.
//   ENC8: the row stream is s_code (one byte per entry) + s_hdr (per block); `s_row` then points at the codes
template <typename T, int RW, int UB, bool ENC8>
__global__ __launch_bounds__(RW * 64) void pb_reduce_kernel(int64_t m, int Hw, int64_t wb_begin, int64_t NBw,
                                                            const int32_t* __restrict__ binblk,
                                                            const T* __restrict__ P,
                                                            const uint16_t* __restrict__ s_row, T* __restrict__ y,
                                                            T alpha, T beta, int K, T* __restrict__ partial,
                                                            int64_t pstride, T* const* __restrict__ peers,
                                                            int n_peers, int64_t peer_off,
                                                            const int4* __restrict__ ritems, int dbg,
                                                            const int32_t* __restrict__ binrow,
                                                            const int32_t* __restrict__ rowmap,
                                                            T* __restrict__ piece_out,
                                                            const typename pb_hdr<T>::type* __restrict__ s_hdr,
                                                            const unsigned* __restrict__ exc_idx,
                                                            const uint16_t* __restrict__ exc_row,
                                                            const int32_t* __restrict__ exc_cnt, int exc_cap) {
  // dbg (SPBLAS_GFX950_PB_DBG, timing experiments only -- results are wrong): 1 = skip the atomic path of flagged
  // entries, 2 = no LDS traffic at all (the stream alone)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  T* acc = reinterpret_cast<T*>(smem) + (size_t) wave * (Hw + 64);  // Hw accumulators + 64 dummy slots
  const int4 item = ritems ? ritems[blockIdx.x] : make_int4((int) blockIdx.x, (int) blockIdx.y, K, 0);
  const int64_t wb = wb_begin + (int64_t) item.x * RW + wave;  // NBw = end of the bin range
  if (ritems)
    partial = item.w >= 0 ? partial + item.w + (int64_t) wave * Hw : nullptr;
  if (wb >= NBw)
    return;
  int64_t r0, r1;
  pb_bin_rows(binrow, wb, Hw, m, &r0, &r1);
  const int rh = (int) (r1 - r0);
  for (int i = lane; i < rh; i += 64)
    acc[i] = T(0);
  constexpr int PB_GBLK = pb_geom<T>::GBLK;
  const int gb0 = binblk[wb] / PB_GBLK, ng = binblk[wb + 1] / PB_GBLK - gb0;
  const int g_lo = gb0 + (int) ((int64_t) ng * item.y / item.z);
  const int g_hi = gb0 + (int) ((int64_t) ng * (item.y + 1) / item.z);
  constexpr int LPB = pb_geom<T>::BLK / 4;  // lanes per block
  typedef typename pb_hdr<T>::type hdr_t;
  struct batch_t {
    T p[UB][4];
    u16x4 r[UB];   // 16-bit rows
    unsigned cw[UB];  // enc8: four row codes
    hdr_t hd[UB];     // enc8: the header of the lane's block
  };
  // flagged entries store to a per-lane dummy slot behind the accumulators instead of sitting in an
  // exec-masked block (the instruction count per group is what the kernel time follows once the stream is linear)
  T* const dummy = acc + Hw + lane;
  T dbg_sum = T(0);
  auto issue = [&](int g, batch_t& q) {
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int gg = (g + u) < g_hi ? (g + u) : (g_hi - 1);
      // byte offsets fit 32 bits (checked at build): global_load with a scalar base
      const unsigned offR = ((unsigned) gg * PB_GRP + 4u * (unsigned) lane) * 2u;
      const unsigned offP = offR * (unsigned) (sizeof(T) / 2);
      pack4<T>::load(reinterpret_cast<const T*>(reinterpret_cast<const char*>(P) + offP), q.p[u]);
      if (ENC8) {
        q.cw[u] = stream_load(reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(s_row) + (offR >> 1)));
#ifdef PB_EXP_NOHDR  // A/B only: what do the header loads cost? (results wrong)
        q.hd[u] = (hdr_t) 0;
#else
        q.hd[u] = stream_load(s_hdr + ((unsigned) gg * (unsigned) PB_GBLK + (unsigned) lane / LPB));
#endif
        continue;
      }
#ifdef PB_EXP_U8ROWS  // timing experiment (tools/build_variant.sh): one byte of row stream per entry; results are wrong
      const unsigned w8 = stream_load(reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(s_row) + (offR >> 1)));
      q.r[u][0] = (unsigned short) ((w8 & 0xffu) << 3);
      q.r[u][1] = (unsigned short) (((w8 >> 8) & 0xffu) << 3);
      q.r[u][2] = (unsigned short) (((w8 >> 16) & 0xffu) << 3);
      q.r[u][3] = (unsigned short) ((w8 >> 24) << 3);
#else
      q.r[u] = stream_load(reinterpret_cast<const u16x4*>(reinterpret_cast<const char*>(s_row) + offR));
#endif
    }
  };
  auto consume = [&](int g, const batch_t& q) {
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (g + u >= g_hi)
        break;
      if (dbg & 2) {
        dbg_sum += q.p[u][0] + q.p[u][1] + q.p[u][2] + q.p[u][3] +
                   (ENC8 ? T(q.cw[u] ^ (unsigned) q.hd[u]) : T(q.r[u][0] ^ q.r[u][1] ^ q.r[u][2] ^ q.r[u][3]));
        continue;
      }
      // one group: all LDS reads, then all writes.  Rows of unflagged entries are distinct inside a
      // group (pb_flag_dups_kernel), flagged ones are added atomically afterwards.
      T v[4];
      T* slot[4];
      bool flagged[4], atomic[4];
      if (ENC8) {
        unsigned row[4];
        bool skip[4];
        pb_decode_rows<LPB>(q.cw[u], pb_hdr<T>::base(q.hd[u]), lane, (unsigned) Hw, row, skip);
        const unsigned fl = pb_hdr<T>::flags(q.hd[u]) >> (4 * (lane & (LPB - 1)));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          atomic[j] = ((fl >> j) & 1u) != 0 && !skip[j];  // duplicates; pads and exceptions take no part here
          flagged[j] = atomic[j] || skip[j];
          slot[j] = acc + row[j];
          v[j] = *slot[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned r = q.r[u][j];
          asm("" : "+v"(r));  // keep the row word a 32-bit value (the 16-bit forms cost extra masking)
          flagged[j] = atomic[j] = r > 0x7FFFu;
          slot[j] = acc + (r & 0x7FFFu);
          v[j] = *slot[j];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        T* w = flagged[j] ? dummy : slot[j];
        *w = v[j] + q.p[u][j];
      }
      if (!(dbg & 1) && __builtin_amdgcn_ballot_w64(atomic[0] | atomic[1] | atomic[2] | atomic[3]) != 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (atomic[j])
            unsafeAtomicAdd(slot[j], q.p[u][j]);
      }
    }
  };
  if (g_hi > g_lo) {
    batch_t qa, qb;
    issue(g_lo, qa);
    for (int g = g_lo; g < g_hi; g += 2 * UB) {
      issue(g + UB, qb);
      consume(g, qa);
      issue(g + 2 * UB, qa);
      consume(g + UB, qb);
    }
  }
  if (dbg & 2)
    *dummy = dbg_sum;
  if (ENC8 && item.y == 0 && !(dbg & 3)) {
    // exceptions of this wave-bin (entries the one-byte codes could not reach): added once, by the first part
    const int ne = exc_cnt[wb];
    for (int i = lane; i < ne; i += 64) {
      const unsigned e = exc_idx[(size_t) wb * exc_cap + i];
      unsafeAtomicAdd(acc + exc_row[(size_t) wb * exc_cap + i], P[e]);
    }
  }
  if (partial) {
    // uniform split: slot k of the [K][m] array; work items: this wave's part of the item's own RW*Hw block
    T* dst = ritems ? partial : partial + (int64_t) blockIdx.y * pstride + r0;
    for (int i0 = lane; i0 < rh; i0 += 64 * 8) {  // (eight LDS reads in flight, as in the final loop below)
      T v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = acc[i0 + 64 * u < rh ? i0 + 64 * u : 0];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + 64 * u < rh)
          dst[i0 + 64 * u] = v[u];
    }
    return;
  }
  if (peers) {
    // fused all-gather (multi-GPU row shards): local row r is row peer_off + r of the full y, and is
    // stored straight into every rank's copy (peers[] holds the local buffer and the IPC-mapped
    // buffers of the other ranks; stores to those travel over xGMI).  beta = 0 by contract.
    // the accumulators are read once, eight per lane at a time, and every value goes to all peers from registers
    for (int i0 = lane; i0 < rh; i0 += 64 * 8) {
      T v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = alpha * acc[i0 + 64 * u < rh ? i0 + 64 * u : 0];
      for (int p = 0; p < n_peers; ++p) {
        T* dst = peers[p] + peer_off + r0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (i0 + 64 * u < rh)
            dst[i0 + 64 * u] = v[u];
      }
    }
    return;
  }
  if (rowmap) {  // compact row -> row of y, or (a piece of a split row) -> its slot of piece_out
    for (int i = lane; i < rh; i += 64) {
      const T v = alpha * acc[i];
      const int32_t r = rowmap[r0 + i];
      if (r < 0)
        piece_out[r0 + i] = v;
      else
        y[r] = beta == T(0) ? v : v + beta * y[r];
    }
    return;
  }
  // eight accumulator reads in flight per lane before the first store: the epilogue runs while nothing else streams
  // (every wavefront reaches it at the same time), one LDS round trip per row group was 4 % of the kernel
  for (int i0 = lane; i0 < rh; i0 += 64 * 8) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 64 * u;
      v[u] = alpha * acc[i < rh ? i : 0];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 64 * u;
      if (i < rh)
        y[r0 + i] = beta == T(0) ? v[u] : v[u] + beta * y[r0 + i];
    }
  }
}


// ---- value-free reduce (round 5) ---------------------------------------------------------------------------------------
// A plain inspected csr_view must multiply with the caller's values OF THAT CALL (multiply_impl.hpp:48-52), and until round
// 5 the tiled plan did so by copying them into its A' stream before every multiply (pb_refresh_bins_kernel: 12.3 B per
// entry on top of the pair's 16).  Here the values never pass through the expand: the product stream carries the gathered
// x[col] (pb_expand_kernel<.., VF>), and the reduce of a bin multiplies by the caller's values itself.  All entries of a
// bin's rows lie in ONE window of the caller's array (rowptr[r0] .. rowptr[r1]), which the workgroup stages in LDS once;
// every entry of the bin's stream carries its 16-bit position inside that window next to its row code.  Bytes per padded
// entry: expand 2 + 4, reduce 4 + 2 + 1.25, plus 4 per entry for the window: ~17-19 B against 28.
// The window is what sizes the bin (fp32: H rows * (avg entries per row + NW) * 4 B <= 160 KiB), so a bin belongs to a
// WORKGROUP, not to a wavefront: wave w of NW reduces the w-th part of the bin's stream into accumulators of its own (plain
// LDS read-add-write stays race free: nobody else touches them; duplicates inside a group are flagged as before), and the
// NW partial rows are added in wave order at the end -- the order of additions is fixed by the plan.
//   rowptr: the caller's row offsets (o64: 64-bit); values: the caller's array of THIS call; win_cap: entries the window area
//   holds = the widest span the build accepted + 16 for the alignment shift of the staged copy
#ifndef PB_VF_GLDS_AUX
#define PB_VF_GLDS_AUX 2  // non-temporal: every window is read by one CU, once per multiply
#endif
// PERSISTENT: the workgroup walks the bins wb_begin + blockIdx.x, + gridDim.x, ... (one workgroup per CU: a bin's window +
// accumulators fill LDS).  A wavefront's part of one bin's stream is short (~13 groups of 256 entries at cfg2), so the loads
// run across the bin boundary: the wave's stream is a flat sequence of BATCHES (UB groups, never straddling a bin), two in
// flight in two register sets, and the batch issued while the last one of bin b is applied already belongs to bin b + 1.
// Between two bins:  barrier A (every wave is done with bin b: window free, accumulators complete) -> y of bin b out of the
// accumulators, which are zeroed by the threads that read them -> window of bin b + 1 by LDS-DMA -> barrier B.  A
// non-persistent form (one bin per workgroup, loads starting behind the staging) took 259 us at cfg2, this one ... see
// DESIGN I.3.
template <typename T, int NW, int UB, bool ENC8>
__global__ __launch_bounds__(NW * 64) void pb_reduce_vf_kernel(int64_t m, int Hw, int64_t wb_begin, int64_t wb_end,
                                                               const int32_t* __restrict__ binblk, const T* __restrict__ P,
                                                               const uint16_t* __restrict__ s_row,
                                                               const uint16_t* __restrict__ s_src,
                                                               const T* __restrict__ values, const void* __restrict__ rowptr,
                                                               int o64, T* __restrict__ y, T alpha, T beta, int win_cap,
                                                               const typename pb_hdr<T>::type* __restrict__ s_hdr,
                                                               const unsigned* __restrict__ exc_idx,
                                                               const uint16_t* __restrict__ exc_row,
                                                               const int32_t* __restrict__ exc_cnt, int exc_cap,
                                                               T* const* __restrict__ peers, int n_peers, int64_t peer_off) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  T* win = reinterpret_cast<T*>(smem);                      // [win_cap] the bin's window of the caller's values
  T* const acc0 = win + win_cap;                            // NW x (Hw accumulators + 64 dummy slots)
  T* const acc = acc0 + (size_t) wave * (Hw + 64);          // this wave's
  const int64_t stride = gridDim.x;
  int64_t wb = wb_begin + blockIdx.x;                       // the bin being reduced
  if (wb >= wb_end)
    return;
  constexpr int PB_GBLK = pb_geom<T>::GBLK;
  constexpr int LPB = pb_geom<T>::BLK / 4;  // lanes per block
  constexpr int VE = 16 / (int) sizeof(T);
  typedef typename pb_hdr<T>::type hdr_t;
  struct batch_t {
    T p[UB][4];       // gathered x values
    u16x4 s[UB];      // positions inside the window
    u16x4 r[UB];      // 16-bit rows
    unsigned cw[UB];  // enc8: four row codes
    hdr_t hd[UB];     // enc8: the header of the lane's block
  };
  T* const dummy = acc + Hw + lane;
  auto row_off = [&](int64_t r) -> int64_t {
    return o64 ? (int64_t) static_cast<const int64_t*>(rowptr)[r] : (int64_t) static_cast<const int32_t*>(rowptr)[r];
  };
  // this wave's part [lo, hi) of bin b's groups
  auto wave_range = [&](int64_t b, int& lo, int& hi) {
    const int gb0 = binblk[b] / PB_GBLK, ng = binblk[b + 1] / PB_GBLK - gb0;
    lo = gb0 + (int) ((int64_t) ng * wave / NW);
    hi = gb0 + (int) ((int64_t) ng * (wave + 1) / NW);
  };
  // loads of the groups g .. g + UB - 1, clamped to `last` (loads past the end of a part re-read its last group -- same lines,
  // no extra traffic -- because a load inside a branch makes the compiler's wait counts pessimistic; a part without groups
  // reads the group at its start: inside the arrays, which carry one group of slack -- and is never applied)
  auto issue = [&](int g, int last, batch_t& q) {
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int gg = (g + u) < last ? (g + u) : last;
      const unsigned offR = ((unsigned) gg * PB_GRP + 4u * (unsigned) lane) * 2u;
      const unsigned offP = offR * (unsigned) (sizeof(T) / 2);
      pack4<T>::load(reinterpret_cast<const T*>(reinterpret_cast<const char*>(P) + offP), q.p[u]);
      q.s[u] = stream_load(reinterpret_cast<const u16x4*>(reinterpret_cast<const char*>(s_src) + offR));
      if (ENC8) {
        q.cw[u] = stream_load(reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(s_row) + (offR >> 1)));
        q.hd[u] = stream_load(s_hdr + ((unsigned) gg * (unsigned) PB_GBLK + (unsigned) lane / LPB));
      } else {
        q.r[u] = stream_load(reinterpret_cast<const u16x4*>(reinterpret_cast<const char*>(s_row) + offR));
      }
    }
  };
  // ---- the window of bin b: 16-byte pieces straight into LDS (global_load_lds_dwordx4: one wave-instruction moves 1 KiB to a
  // wave-uniform LDS base + 16 B per lane, no registers, no ds_write pass), the whole window in flight at once
  int shift = 0;  // position of the bin's first entry inside the staged window
  auto stage_window = [&](int64_t b) {
#ifdef PB_EXP_VF_NOWIN  // timing experiment only (results wrong): no window staging
    return;
#endif
    const int64_t r0 = b * Hw, r1 = (r0 + Hw) < m ? (r0 + Hw) : m;
    const int64_t p0 = row_off(r0), p1 = row_off(r1);
    const bool vec_ok = (reinterpret_cast<uintptr_t>(values) & 15) == 0;
    const int64_t p_lo = vec_ok ? (p0 & ~(int64_t) (VE - 1)) : p0;
    shift = (int) (p0 - p_lo);
    const int wn = (int) (p1 - p_lo);
    if (vec_ok) {
      typedef T vec_t __attribute__((ext_vector_type(VE)));
      const int nv = wn / VE;
      const vec_t* src = reinterpret_cast<const vec_t*>(values + p_lo);
      const int nchunk = (nv + 63) >> 6;
      for (int c = wave; c < nchunk; c += NW) {
        const int i = c * 64 + lane;
        if (i < nv)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) (src + i),
                                           (__attribute__((address_space(3))) void*) (reinterpret_cast<vec_t*>(win) + c * 64),
                                           16, 0, PB_VF_GLDS_AUX);
      }
      for (int i = nv * VE + tid; i < wn; i += NW * 64)
        win[i] = stream_load(values + p_lo + i);
    } else {
      for (int i = tid; i < wn; i += NW * 64)
        win[i] = stream_load(values + p_lo + i);
    }
  };
  // ---- one batch into the wave's accumulators
  auto consume = [&](int g, int g_hi, const batch_t& q) {
    const T* wv = win + shift;
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (g + u >= g_hi)
        break;
#ifdef PB_EXP_VF_NOACC  // timing experiment only (results wrong): the stream and the window without the LDS accumulation
      {
        T t = q.p[u][0] + q.p[u][1] + q.p[u][2] + q.p[u][3] + T(q.s[u][0] ^ q.s[u][1] ^ q.s[u][2] ^ q.s[u][3]);
        if (ENC8)
          t += T(q.cw[u] ^ (unsigned) q.hd[u]);
        else
          t += T(q.r[u][0]);
        *dummy += t;
        continue;
      }
#endif
      T v[4], pr[4];
      T* slot[4];
      bool flagged[4], atomic[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        pr[j] = wv[q.s[u][j]];
      if (ENC8) {
        unsigned row[4];
        bool skip[4];
        pb_decode_rows<LPB>(q.cw[u], pb_hdr<T>::base(q.hd[u]), lane, (unsigned) Hw, row, skip);
        const unsigned fl = pb_hdr<T>::flags(q.hd[u]) >> (4 * (lane & (LPB - 1)));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          atomic[j] = ((fl >> j) & 1u) != 0 && !skip[j];
          flagged[j] = atomic[j] || skip[j];
          slot[j] = acc + row[j];
          v[j] = *slot[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned r = q.r[u][j];
          asm("" : "+v"(r));
          flagged[j] = atomic[j] = r > 0x7FFFu;
          slot[j] = acc + (r & 0x7FFFu);
          v[j] = *slot[j];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pr[j] *= q.p[u][j];
        T* w = flagged[j] ? dummy : slot[j];
        *w = v[j] + pr[j];
      }
      if (__builtin_amdgcn_ballot_w64(atomic[0] | atomic[1] | atomic[2] | atomic[3]) != 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (atomic[j])
            unsafeAtomicAdd(slot[j], pr[j]);
      }
    }
  };
  // ---- between two bins
  auto end_of_bin = [&]() {
    if (ENC8 && wave == 0) {
      // exceptions of this bin (entries the one-byte codes could not reach): added once, by the first wave
      const T* wv = win + shift;
      const int ne = exc_cnt[wb];
      for (int i = lane; i < ne; i += 64) {
        const unsigned e = exc_idx[(size_t) wb * exc_cap + i];
        unsafeAtomicAdd(acc + exc_row[(size_t) wb * exc_cap + i], P[e] * wv[s_src[e]]);
      }
    }
    // A: every wave is done with the bin (its LDS traffic complete; the loads already issued for the next bin stay in flight)
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    // y = alpha * (the waves' partial rows, in wave order) + beta * y; the accumulators are left zeroed
    const int64_t r0 = wb * Hw;
    const int rh = (int) (((r0 + Hw) < m ? (r0 + Hw) : m) - r0);
    for (int i0 = tid; i0 < rh; i0 += NW * 64 * 4) {
      T sum[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = (i0 + u * NW * 64) < rh ? (i0 + u * NW * 64) : 0;
        T t = acc0[i];
#pragma unroll
        for (int w = 1; w < NW; ++w)
          t += acc0[(size_t) w * (Hw + 64) + i];
        sum[u] = alpha * t;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * NW * 64;
        if (i < rh) {
          if (peers) {
            // fused all-gather (round 6: value-free plans too): local row r is row peer_off + r of the full y, stored into
            // every rank's copy -- the local buffer and the IPC-mapped ones, those over xGMI.  beta = 0 by contract.
            for (int p = 0; p < n_peers; ++p)
              peers[p][peer_off + r0 + i] = sum[u];
          } else {
            y[r0 + i] = beta == T(0) ? sum[u] : sum[u] + beta * y[r0 + i];
          }
#pragma unroll
          for (int w = 0; w < NW; ++w)
            acc0[(size_t) w * (Hw + 64) + i] = T(0);
        }
      }
    }
    wb += stride;
    if (wb < wb_end) {
      stage_window(wb);
      // B: the window has landed, the accumulators are zeroed.  (The builtin, not inline asm: the compiler's wait-count pass
      // has to SEE that the LDS-DMA loads are done -- otherwise it drains vmcnt(0) before every LDS read of the main loop,
      // i.e. right after the next batch's loads were issued.)
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_s_barrier();
    }
  };
  batch_t qa, qb;
  int g_lo, g_hi;
  wave_range(wb, g_lo, g_hi);
  issue(g_lo, g_hi > g_lo ? g_hi - 1 : g_lo, qa);
  for (int i = tid; i < NW * (Hw + 64); i += NW * 64)
    acc0[i] = T(0);
  stage_window(wb);
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_s_barrier();
  for (;;) {
    // the wave's part of the NEXT bin: its first batch is loaded while the last batches of this bin are applied
    const int64_t nb = wb + stride;
    int n_lo = g_hi > g_lo ? g_hi - 1 : g_lo, n_hi = n_lo;  // (no next bin: a part without groups on the last group read)
    if (nb < wb_end)
      wave_range(nb, n_lo, n_hi);
    const int last = g_hi > g_lo ? g_hi - 1 : g_lo, n_last = n_hi > n_lo ? n_hi - 1 : n_lo;
    int g = g_lo;
    do {  // (at least once: a wave without groups in this bin still has the next bin's first batch to load)
      issue(g + UB, last, qb);
      consume(g, g_hi, qa);
      const bool more = g + 2 * UB < g_hi;
      issue(more ? g + 2 * UB : n_lo, more ? last : n_last, qa);
      consume(g + UB, g_hi, qb);
      g += 2 * UB;
    } while (g < g_hi);
    end_of_bin();
    if (wb >= wb_end)
      break;
    g_lo = n_lo;
    g_hi = n_hi;
  }
}

template <typename T>
static const void* pb_reduce_vf_fn(int nw, int ub, bool enc8) {
#define SPB_VF(NW_, UB_, E_) reinterpret_cast<const void*>(pb_reduce_vf_kernel<T, NW_, UB_, E_>)
  if (enc8) {
    if (nw == 8)
      return ub == 4 ? SPB_VF(8, 4, true) : SPB_VF(8, 2, true);
    return ub == 4 ? SPB_VF(4, 4, true) : SPB_VF(4, 2, true);
  }
  if (nw == 8)
    return ub == 2 ? SPB_VF(8, 2, false) : SPB_VF(8, 4, false);
  return ub == 2 ? SPB_VF(4, 2, false) : SPB_VF(4, 4, false);
#undef SPB_VF
}
// widest bin window of the arithmetic bin grid: max over b of rowptr[min((b + 1) H, m)] - rowptr[b H]
template <typename O>
__global__ __launch_bounds__(256) void pb_bin_span_kernel(int64_t m, int H, int64_t NB, const O* __restrict__ rowptr,
                                                          unsigned long long* __restrict__ out,
                                                          const int32_t* __restrict__ binrow = nullptr) {
  const int64_t b = (int64_t) blockIdx.x * 256 + threadIdx.x;
  unsigned long long span = 0;
  if (b < NB) {
    int64_t r0, r1;
    pb_bin_rows(binrow, b, H, m, &r0, &r1);
    if (r0 < m)
      span = (unsigned long long) (rowptr[r1] - rowptr[r0]);
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long u = __shfl_xor(span, o, 64);
    span = u > span ? u : span;
  }
  if ((threadIdx.x & 63) == 0 && span > 0)
    atomicMax(out, span);
}

// y = alpha * (partial[0] + partial[1] + ... in this fixed order) + beta * y
template <typename T>
__global__ __launch_bounds__(256) void pb_combine_kernel(int64_t r_lo, int64_t r_hi, int K,
                                                         const T* __restrict__ partial, int64_t pstride,
                                                         T* __restrict__ y, T alpha, T beta,
                                                         T* const* __restrict__ peers, int n_peers,
                                                         int64_t peer_off, const int32_t* __restrict__ rowmap,
                                                         T* __restrict__ piece_out) {
  int64_t i = r_lo + (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= r_hi)
    return;
  T s = partial[i];
  for (int k = 1; k < K; ++k)
    s += partial[(int64_t) k * pstride + i];
  if (rowmap) {
    const int32_t r = rowmap[i];
    if (r < 0) {
      piece_out[i] = alpha * s;
      return;
    }
    i = r;
  }
  if (peers) {
    for (int p = 0; p < n_peers; ++p)
      peers[p][peer_off + i] = alpha * s;
    return;
  }
  y[i] = beta == T(0) ? alpha * s : alpha * s + beta * y[i];
}

// The same for the chunked multi-GPU step (peer stores only, no row map): the rows are published to the peers chunk by
// chunk WITHOUT a kernel boundary per chunk -- splitting the reduce into stripes costs 28 / 55 us for 2 / 4 stripes on a
// 1.25 M-row shard (profiles/r03_shards_and_fused_floor.md), extra launches cost as much.  Workgroups are dispatched in row
// order; every workgroup ends with a system-scope release fence and counts itself into its chunk's arrival counter, and the
// workgroup that completes a chunk stores `step` into slot slot0 + chunk of every rank's flag array.  (Ordering by hand:
// rows and flags are both system-scope write-through stores; a workgroup counts itself in only after its stores were
// acknowledged; the agent-scope counter is an atomic in memory, not in an L2.  Exercised with ranks that share ONE device --
// across devices nothing of this has run yet.)
static constexpr int PB_PUB_ROWS = 2048;  // rows per workgroup of the publishing combine (chunks are multiples of it)
template <typename T>
__global__ __launch_bounds__(256) void pb_combine_publish_kernel(int64_t r_lo, int64_t r_hi, int K,
                                                                 const T* __restrict__ partial, int64_t pstride, T alpha,
                                                                 T* const* __restrict__ peers, int n_peers, int64_t peer_off,
                                                                 long long* const* __restrict__ flag_peers, int slot0,
                                                                 int chunks, long long rows_per_chunk, long long step,
                                                                 int* __restrict__ done, long long delay_ticks) {
  // (2 048 rows per workgroup, not 256: the arrival counters are same-address atomics, ~11 ns each one after the other --
  // 4 883 of them made the combine of a 1.25 M-row shard 55 us instead of 5)
#pragma unroll
  for (int u = 0; u < PB_PUB_ROWS / 256; ++u) {
    const int64_t i = r_lo + (int64_t) blockIdx.x * PB_PUB_ROWS + u * 256 + threadIdx.x;
    if (i < r_hi) {
      T s = partial[i];
      for (int k = 1; k < K; ++k)
        s += partial[(int64_t) k * pstride + i];
      // write-through, system-scope stores: nothing of y stays dirty in this XCD's L2, so "visible everywhere" is "all my
      // stores have been acknowledged" -- a system-scope release FENCE instead writes the whole L2 back, per workgroup:
      // 241 instead of 62 us for the step of a 1.25 M-row shard (tools/chunk_overhead.py)
      for (int p = 0; p < n_peers; ++p)
        __hip_atomic_store(peers[p] + peer_off + i, alpha * s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wavefront's stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    const long long first = (long long) blockIdx.x * PB_PUB_ROWS;
    long long c = first / rows_per_chunk;
    c = c < chunks ? c : chunks - 1;
    const long long c_lo = c * rows_per_chunk;
    const long long c_hi = (c == chunks - 1 || (c + 1) * rows_per_chunk > (long long) (r_hi - r_lo)) ? (long long) (r_hi - r_lo)
                                                                                                    : (c + 1) * rows_per_chunk;
    const int wgs = (int) ((c_hi - c_lo + PB_PUB_ROWS - 1) / PB_PUB_ROWS);
    // (acquire-release on the arrival counter, release on the flags: only the thread that completes a chunk pays for the
    // fences, and nothing in the memory model lets a flag overtake the rows it stands for -- the hand-made order above
    // [write-through stores, vmcnt(0), barrier] was exercised with ranks sharing ONE device only; round-4 advisor finding)
    if (__hip_atomic_fetch_add(done + c, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == wgs - 1) {
      __hip_atomic_store(done + c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (the next step counts from zero)
      if (delay_ticks > 0 && c == 1) {  // test hook: a deliberately late chunk
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < delay_ticks)
          __builtin_amdgcn_s_sleep(32);
      }
      for (int p = 0; p < n_peers; ++p)
        __hip_atomic_store(flag_peers[p] + slot0 + c, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Rows that were kept out of the tiles (longer than plan->hub_len).  gridDim.y workgroups share a row
// (a 1M-entry row on one workgroup would take milliseconds): each writes the sum of its part to
// part[i * gridDim.y + k]; pb_hub_finish_kernel adds the parts in order (deterministic) and does
// y[row] += alpha * sum -- the reduce kernel has already written beta*y (+ nothing) there.  Only rows
// inside [row_begin, row_end) are touched (two-stage callers reduce row ranges).
template <typename T, typename O>
__global__ __launch_bounds__(256) void pb_hub_rows_kernel(int64_t n_hub, const int32_t* __restrict__ hub_rows,
                                                          const O* __restrict__ rowptr,
                                                          const int32_t* __restrict__ colind,
                                                          const T* __restrict__ values, const T* __restrict__ x,
                                                          T* __restrict__ part, int64_t row_begin, int64_t row_end) {
  __shared__ T red[4];
  const int64_t i = blockIdx.x;
  const int64_t r = hub_rows[i];
  if (r < row_begin || r >= row_end)
    return;
  const O p0 = rowptr[r], p1 = rowptr[r + 1];
  const O per = ((p1 - p0) + (O) gridDim.y - 1) / (O) gridDim.y;
  const O lo = p0 + (O) blockIdx.y * per, hi = (lo + per) < p1 ? (lo + per) : p1;
  T s = T(0);
  for (O p = lo + threadIdx.x; p < hi; p += 256)
    s += stream_load(values + p) * x[stream_load(colind + p)];
  s = group_sum_c<64>(s);
  if ((threadIdx.x & 63) == 0)
    red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0)
    part[i * gridDim.y + blockIdx.y] = red[0] + red[1] + red[2] + red[3];
}

template <typename T>
__global__ __launch_bounds__(256) void pb_hub_finish_kernel(int64_t n_hub, int parts,
                                                            const int32_t* __restrict__ hub_rows,
                                                            const T* __restrict__ part, T* __restrict__ y, T alpha,
                                                            int64_t row_begin, int64_t row_end,
                                                            const int32_t* __restrict__ rowmap) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= n_hub)
    return;
  const int64_t r = hub_rows[i];
  if (r < row_begin || r >= row_end)
    return;
  T s = T(0);
  for (int k = 0; k < parts; ++k)
    s += part[i * parts + k];
  y[rowmap ? (rowmap[r] & ~PB_PIECE_BIT) : r] += alpha * s;
}

// Work-item variant of the combine: one entry of `cg` per SPLIT bin group = (group, K_g, offset of its
// first partial block, rows in the group); blockIdx.y walks the group's rows in chunks of 256.  The K_g
// partial blocks of RW*Hw values lie back to back and are added in that order.
template <typename T>
__global__ __launch_bounds__(256) void pb_combine_items_kernel(const int4* __restrict__ cg, int64_t group_rows,
                                                               int64_t m, const T* __restrict__ partial,
                                                               T* __restrict__ y, T alpha, T beta, int Hw,
                                                               const int32_t* __restrict__ binrow, int64_t NB,
                                                               const int32_t* __restrict__ rowmap,
                                                               T* __restrict__ piece_out) {
  const int4 g = cg[blockIdx.x];
  const int64_t i = (int64_t) blockIdx.y * 256 + threadIdx.x;
  if (i >= group_rows)
    return;
  int64_t row = (int64_t) g.x * group_rows + i;
  if (binrow) {  // slot i of the group's partial block = row (i % Hw) of its wave-bin (i / Hw)
    const int64_t wb = (int64_t) g.x * (group_rows / Hw) + i / Hw;
    if (wb >= NB)
      return;
    const int64_t r0 = binrow[wb], off = i % Hw;
    if (off >= binrow[wb + 1] - r0)
      return;
    row = r0 + off;
  }
  if (row >= m)
    return;
  const T* src = partial + (int64_t) g.z + i;
  T s = src[0];
  for (int k = 1; k < g.y; ++k)
    s += src[(int64_t) k * group_rows];
  if (rowmap) {
    const int32_t r = rowmap[row];
    if (r < 0) {
      piece_out[row] = alpha * s;
      return;
    }
    row = r;
  }
  y[row] = beta == T(0) ? alpha * s : alpha * s + beta * y[row];
}

template <typename T>
static const void* pb_reduce_fn(int rw, int ub, bool enc8 = false) {
  if (enc8) {
    if (rw == 8)
      return ub == 2 ? (const void*) pb_reduce_kernel<T, 8, 2, true>
                     : ub == 8 ? (const void*) pb_reduce_kernel<T, 8, 8, true> : (const void*) pb_reduce_kernel<T, 8, 4, true>;
    return ub == 1 ? (const void*) pb_reduce_kernel<T, 4, 1, true>
           : ub == 2 ? (const void*) pb_reduce_kernel<T, 4, 2, true>
                     : ub == 8 ? (const void*) pb_reduce_kernel<T, 4, 8, true> : (const void*) pb_reduce_kernel<T, 4, 4, true>;
  }
  if (ub == 1 && rw == 4)
    return (const void*) pb_reduce_kernel<T, 4, 1, false>;
#define SPB_RK(RW_, UB_) reinterpret_cast<const void*>(pb_reduce_kernel<T, RW_, UB_, false>)
  if (rw == 8)
    return ub == 2 ? SPB_RK(8, 2) : (ub == 8 ? SPB_RK(8, 8) : SPB_RK(8, 4));
  return ub == 2 ? SPB_RK(4, 2) : (ub == 8 ? SPB_RK(4, 8) : SPB_RK(4, 4));
#undef SPB_RK
}

// ---- host -------------------------------------------------------------------------------
static int pick_ksplit(int64_t waves, int64_t steps_per_bin);

// number of pieces: enough that one piece fits the LDS budget; for big problems a multiple
// of 512 (2 workgroups x 256 CUs) so the single wave of workgroups fills the chip evenly.
// Split `extent` into pieces of at most max_elems.  Big problems get a piece count that is a
// multiple of round_to (whole waves of workgroups over the 256 CUs); trailing pieces may then
// be empty, which every kernel tolerates.
// pieces = as few as fit max_elems, rounded up to a multiple of round_to once there are more than
// round_from of them (whole waves of workgroups), width = the matching piece size (multiple of align)
static void pick_tiling(int64_t extent, int max_elems, int round_to, int round_from, int align, int* pieces,
                        int* width) {
  int64_t p = cdiv(extent, max_elems);
  if (p < 1)
    p = 1;
  const bool rounded = p > round_from;
  if (rounded)
    p = cdiv(p, round_to) * round_to;
  int64_t w = cdiv(extent, p);
  w = cdiv(w, align) * align;
  if (w > max_elems)
    w = (max_elems / align) * align;
  if (w < align)
    w = align;
  if (!rounded || w * p < extent)
    p = cdiv(extent, w);
  *pieces = (int) p;
  *width = (int) w;
}


// SPBLAS_GFX950_TRACE_INSPECT=1: host-side time stamps of the inspect phases on stderr (drains the stream)
struct pb_tracer {
  bool on;
  hipStream_t s;
  double t0;
  static double now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
  }
  pb_tracer(hipStream_t st) : on(env_int("SPBLAS_GFX950_TRACE_INSPECT", 0) != 0), s(st), t0(now()) {}
  void mark(const char* what) {
    if (!on)
      return;
    (void) hipStreamSynchronize(s);
    std::fprintf(stderr, "[inspect] %8.3f ms  %s\n", now() - t0, what);
  }
};


template <typename T, typename O>
static int sliced_build_typed(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values_p,
                              bool auto_mode) {
  hipStream_t s = h->stream;
  const int64_t n = pl->n, nnz = pl->nnz;
  int64_t m = pl->m;
  if (nnz > INT32_MAX - 8)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  int rc;
  readback_scope rb_scope(h);  // several read-backs below land in locals of this frame
  pb_tracer tr(s);
  const O* rowptr = static_cast<const O*>(pl->rowptr);
  // Matrices with many empty rows (graphs: 56 % of the rows of R-MAT scale 24) are tiled over their NON-EMPTY rows
  // only: same colind / values, a compacted row pointer array, and the reduce writes row nzrow[i] of y for compact
  // row i.  Empty rows cost accumulator slots and whole wave-bins otherwise (cfg4: 9 445 -> 5 680 bins, padding
  // 21 % -> 15 %, 2.53 -> 2.32 ms).  In the same row map, rows LONGER than split_len entries become several compact rows
  // ("pieces"): a row of 16 k entries otherwise leaves ~80 entries in each of its runs, which the reduce has to add
  // with LDS float atomics -- 0.47 of 0.80 ms on R-MAT scale 22 fp32 (SPBLAS_GFX950_PB_DBG=1).  Pieces are ordinary rows
  // to every kernel; their sums meet in pb_split_finish_kernel.  Needs variable-height bins (the boundaries live in
  // binrow[], in compact rows).
  int compact = env_int("SPBLAS_GFX950_PB_COMPACT", -1);
  if (compact < 0)
    compact = pl->empty_rows * 4 > m;
  if (pl->empty_rows == 0 || m - pl->empty_rows < 2)
    compact = 0;
  // piece length: the fp32 LDS add is the slow one (0.33 lanes per clock and CU), so fp32 wants pieces short enough that
  // two entries of a piece rarely meet in one group of the reduce (measured on R-MAT scale 22 / 24: 32 beats 16, 64, 128 and
  // 512); fp64 is nearly insensitive and prefers fewer compact rows (2048)
  int split_len = env_int("SPBLAS_GFX950_PB_SPLIT_LEN", sizeof(T) == 4 ? 32 : 2048);
  {
    const double avg0 = m > 0 ? (double) nnz / (double) m : 0.0;
    if (split_len < 8 || pl->max_row_len <= 2 * (int64_t) split_len || (double) pl->max_row_len <= 16.0 * avg0 + 64.0)
      split_len = 0;  // no row is long enough to matter
  }
  if (h->bin_row_align > 1 || env_int("SPBLAS_GFX950_PB_VARBINS", -1) == 0 || m < 2)
    compact = split_len = 0;
  pl->s_m = m;
  pl->split_len = 0;
  if (compact || split_len > 0) {
    int32_t* pieces = nullptr;
    long long* fpart = nullptr;
    unsigned long long* counters = nullptr;
    struct guard_t {
      hipStream_t s;
      void* p[3] = {nullptr, nullptr, nullptr};
      ~guard_t() {
        for (void* q : p)
          dev_free(q, s);
      }
    } g{s};
    if ((rc = dev_alloc((void**) &pieces, (size_t) (m + 1) * 4, s)))
      return rc;
    g.p[0] = pieces;
    if ((rc = dev_alloc((void**) &fpart, (size_t) (cdiv(m, 2048) + 2) * sizeof(long long), s)))
      return rc;
    g.p[1] = fpart;
    if ((rc = dev_alloc((void**) &counters, 2 * sizeof(unsigned long long), s)))
      return rc;
    g.p[2] = counters;
    SPB_HIP(hipMemsetAsync(counters, 0, 2 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL((pb_row_pieces_kernel<O>), dim3((unsigned) cdiv(m, 256)), dim3(256), 0, s, m, rowptr, compact,
                       split_len, pieces);
    long long* total_dev = scan_counts_i32(s, m, pieces, fpart);
    long long m_c = 0;
    if ((rc = readback_add(h, &m_c, total_dev, sizeof(m_c))) || (rc = readback_flush(h)))
      return rc;
    if (m_c < 2 || m_c >= INT32_MAX)
      return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
    const int64_t split_cap = split_len > 0 ? nnz / split_len + 1 : 1;
    if ((rc = dev_alloc(&pl->s_rowptr_c, (size_t) (m_c + 1) * sizeof(O), s)) ||
        (rc = dev_alloc(&pl->s_nzrow, (size_t) m_c * 4, s)) ||
        (rc = dev_alloc(&pl->s_zrow, (size_t) (compact ? pl->empty_rows : 1) * 4, s)) ||
        (rc = dev_alloc(&pl->s_split_rows, (size_t) split_cap * sizeof(int4), s)))
      return rc;
    hipLaunchKernelGGL((pb_row_map_kernel<O>), dim3((unsigned) cdiv(m + 1, 1024)), dim3(1024), 0, s, m, rowptr, pieces, split_len,
                       static_cast<O*>(pl->s_rowptr_c), static_cast<int32_t*>(pl->s_nzrow),
                       static_cast<int32_t*>(pl->s_zrow), static_cast<int4*>(pl->s_split_rows), counters);
    unsigned long long h_cnt[2] = {0, 0};
    if ((rc = readback_add(h, h_cnt, counters, sizeof(h_cnt))) || (rc = readback_flush(h)))
      return rc;
    SPB_HIP(hipGetLastError());
    pl->n_zero = (int64_t) h_cnt[0];
    pl->n_split = (int64_t) h_cnt[1];
    if (pl->n_split > 0) {
      if ((rc = dev_alloc(&pl->s_piece_out, (size_t) m_c * sizeof(T), s)))
        return rc;
      pl->split_len = split_len;
    }
    pl->s_m = m_c;
    pl->device_bytes += (size_t) (m_c + 1) * sizeof(O) + (size_t) m_c * 4 + (size_t) pl->n_zero * 4 +
                        (size_t) pl->n_split * 16 + (pl->n_split > 0 ? (size_t) m_c * sizeof(T) : 0);
    m = m_c;
    rowptr = static_cast<const O*>(pl->s_rowptr_c);
    compact = 1;  // from here on: "the tiles are built over a row map"
    tr.mark("row map (empty rows out, long rows split)");
  }
  // x slice of the expand: 80 KiB for fp32 (two workgroups per CU); for fp64 the whole 160 KiB of a CU
  // (one workgroup), which halves the number of slices and doubles the run length
  // fp32 switches to 160 KiB slices as well from n = 2 M on: half the slices, twice the run length (square
  // matrices with 10 entries per row, tools/xlds_mid.sh: n = 3 M 109 -> 101 us, 6 M 203 -> 193, 7.9 M 285 -> 248;
  // even at 1-2 M, 8 us slower at 0.5 M where 13 slices cannot feed the chip).  Row shards of a multi-GPU run -- the
  // columns of the whole matrix but a fraction of its entries -- were kept on 80 KiB slices until the slice count
  // was aligned to the CU count (one expand workgroup per slice, x read once); with the alignment the wide slices
  // win at every shard size: 10 M columns x 2.5 M / 1.25 M / 0.625 M rows 111 -> 100, 77 -> 62, 60 -> 45 us
  // (tools/shard_sweep.sh).
  const int64_t s80 = cdiv(n, PB_LDS_BYTES / 4);
  const bool wide32 = sizeof(T) == 4 && s80 >= 100;
  const int xlds = env_int("SPBLAS_GFX950_PB_XLDS_KB", (sizeof(T) == 8 || wide32) ? 160 : PB_LDS_BYTES / 1024) * 1024;
  int max_cols = xlds / (int) sizeof(T);
  if (max_cols > 65536)
    max_cols = 65536;  // 16-bit local column
  // reduce shape: RW wave-bins per workgroup share `rlds` bytes of LDS.  Fewer, taller bins make longer runs
  // (less block padding) but leave fewer wavefronts to hide latency.
  int RW = env_int("SPBLAS_GFX950_PB_RWAVES", PB_RWAVES_DEFAULT);
  if (RW != 4 && RW != 8)
    RW = PB_RWAVES_DEFAULT;
  int rlds = env_int("SPBLAS_GFX950_PB_RLDS_KB", PB_LDS_BYTES / 1024) * 1024;
  if (rlds < 16 * 1024 || rlds > 160 * 1024)
    rlds = PB_LDS_BYTES;
  // per wave-bin; < 32768 - 64 (15-bit row + flag, pads carry row = H); 64 dummy slots per wave follow the accumulators
  int max_rows = rlds / RW / (int) sizeof(T) - 64;
  if (max_rows > 32000)
    max_rows = 32000;
  // Value-free tiles (pb_reduce_vf_kernel): asked for by plan_create for a plan that has to read the caller's values on
  // every multiply.  A bin is then a WORKGROUP's (NWv wavefronts with accumulators of their own) and its height follows
  // from the window of the caller's values that has to fit LDS beside them: H * (entries per row + NWv) elements.  Arithmetic
  // bins over the rows of y only (no row map, no variable heights), the widest window checked on the device; anything else
  // falls back to the copying plan (vfree = 0) before a byte is allocated.
  const bool vf_forced = env_int("SPBLAS_GFX950_PB_VFREE", 1) == 2;  // test hook: also for small / skewed matrices
  int vfree = pl->vfree && env_int("SPBLAS_GFX950_PB_VFREE", 1) && !compact && split_len == 0 && h->bin_row_align <= 1 &&
              env_int("SPBLAS_GFX950_SLICE_ROWS", 0) <= 0;
  // (eight wavefronts per bin against four: 243 against 267 us for the reduce at cfg2 -- the shorter runs of the lower bins
  // cost less than the latency four wavefronts per CU cannot hide)
  int NWv = env_int("SPBLAS_GFX950_PB_VF_WAVES", 8);
  if (NWv != 4 && NWv != 8)
    NWv = 8;
  constexpr int VF_LDS = 160 * 1024 - 64;
  // (SPBLAS_GFX950_PB_VF_LDS_KB = 80: two workgroups per CU with bins half as tall -- measured)
  const int vf_lds_req = env_int("SPBLAS_GFX950_PB_VF_LDS_KB", 160) == 80 ? 80 * 1024 - 64 : VF_LDS;
  const int vf_elems = vf_lds_req / (int) sizeof(T);
  int vf_rows = 0, vf_cap = 0;
  if (vfree) {
    const double avg = m > 0 ? (double) nnz / (double) m : 0.0;
    const double skew = (double) pl->max_row_len > 16.0 * avg + 64.0 || pl->empty_rows * 4 > pl->m;
    int64_t hh = (int64_t) ((double) (vf_elems - NWv * 64 - 16) / (avg * 1.02 + (double) NWv));
    hh = std::min<int64_t>(hh, vf_elems / (2 * NWv) - 64);  // never more than half of LDS in accumulators
    hh = std::min<int64_t>(hh, 32000);
    const int vf_rows_env = env_int("SPBLAS_GFX950_PB_VF_ROWS", 0);  // test hook: small bins
    if (vf_rows_env > 0)
      hh = std::min<int64_t>(hh, vf_rows_env);
    if ((skew && !vf_forced) || hh < (vf_forced ? 1 : 64) || m < 2)
      vfree = 0;
    for (int attempt = 0; vfree && attempt < 4; ++attempt) {
      const int64_t nb = cdiv(m, hh);
      const int64_t cap = (int64_t) vf_elems - (int64_t) NWv * (hh + 64) - 16;  // 16: the alignment shift of the window
      unsigned long long* d_span = nullptr;
      unsigned long long h_span = 0;
      if ((rc = dev_alloc((void**) &d_span, sizeof(unsigned long long), s)))
        return rc;
      hipError_t e = hipMemsetAsync(d_span, 0, sizeof(unsigned long long), s);
      if (e == hipSuccess) {
        hipLaunchKernelGGL((pb_bin_span_kernel<O>), dim3((unsigned) cdiv(nb, 256)), dim3(256), 0, s, m, (int) hh, nb, rowptr, d_span);
        e = hipMemcpyAsync(&h_span, d_span, sizeof(h_span), hipMemcpyDeviceToHost, s);
      }
      if (e == hipSuccess)
        e = hipStreamSynchronize(s);
      dev_free(d_span, s);
      if (e != hipSuccess)
        return hip_fail(e);
      if ((int64_t) h_span <= cap && h_span < 65536ull) {
        vf_rows = (int) hh;
        // the window AREA is the accepted span plus the 16 elements set aside above: stage_window aligns the window's start
        // down to 16 bytes and writes span + shift elements (shift <= 16 / sizeof(T) - 1), and the accumulators begin right
        // behind the area -- a bin whose span reaches `cap` with a misaligned rowptr[r0] used to put raw values of A into
        // the first accumulators of wave 0 (round-5 advisor; tests: test_spmv_value_free_tiles_widest_window_misaligned)
        vf_cap = (int) cap + 16;
        break;
      }
      // a denser stretch of rows than the average: shrink the bins in proportion (with a margin) and look again
      hh = (int64_t) ((double) hh * (double) std::min<int64_t>(cap, 65535) / (double) h_span * 0.97);
      if (hh < (vf_forced ? 1 : 64) || attempt == 3)
        vfree = 0;
    }
    if (vf_rows == 0)
      vfree = 0;
    // one workgroup per bin and CU: enough bins for a few rounds over the chip
    if (vfree && !vf_forced && cdiv(m, vf_rows) < 2 * (int64_t) (h->num_cus > 0 ? h->num_cus : 256))
      vfree = 0;
    tr.mark("value-free tiles: bin height from the widest window");
  }
  pl->vfree = vfree;
  if (vfree) {
    RW = 1;  // inspect-side bookkeeping is per bin
    max_rows = vf_rows;
    pl->vf_waves = NWv;
    pl->vf_win_cap = vf_cap;
  }
  pl->rwaves = RW;
  int S, W, NB, H;
  const int w_env = env_int("SPBLAS_GFX950_SLICE_COLS", 0);  // test hooks: force small tiles
  const int h_env = env_int("SPBLAS_GFX950_SLICE_ROWS", 0);
  // Slices: as few (as wide) as LDS allows -- every extra slice shortens all runs.  The expand gives every
  // workgroup an equal share of A' whatever the slice count is, so the count needs no rounding to waves of
  // workgroups (SPBLAS_GFX950_PB_XROUND is a test hook).
  const int xround = env_int("SPBLAS_GFX950_PB_XROUND", 1);
  pick_tiling(n, w_env > 0 && w_env < max_cols ? w_env : max_cols, xround, xround, 4, &S, &W);
  // One slice per expand workgroup when that costs little: with 160 KiB slices the expand runs one workgroup per CU, and
  // a slice count just below the CU count (cfg2: 245 of 256) makes every equal share straddle two slices -- two x-slice
  // loads and two barriers per workgroup, 80 MB of x reads instead of 40.  Rounding the count up to the CU count
  // narrows the slices by < 10 % and lets workgroup i take exactly slice i (SPBLAS_GFX950_PB_SLICE_ALIGN=0: off).
  bool slice_aligned = false;
  {
    const int cus = h->num_cus > 0 ? h->num_cus : 256;
    if (env_int("SPBLAS_GFX950_PB_SLICE_ALIGN", 1) && w_env <= 0 && xlds > PB_LDS_BYTES && S <= cus && S * 10 >= cus * 9) {
      const int64_t w2 = cdiv(cdiv(n, cus), 4) * 4;
      if (w2 <= max_cols && cdiv(n, w2) <= cus) {
        W = (int) w2;
        S = (int) cdiv(n, w2);
        slice_aligned = true;
      }
    }
  }
  // Bins: enough wavefronts to fill the chip (8 per CU = 2 048), but never so many that the average run drops
  // below ~3 blocks (half a block of padding per run), and never taller than the LDS budget allows.  Too few
  // bins for the chip are made up for by the slice-free K split of the reduce (every bin's stream cut in K parts).
  {
    const int64_t nb_min = cdiv(m, max_rows);
    const int64_t nb_fill = env_int("SPBLAS_GFX950_PB_BINS", 2048);
    // (3 blocks per run: row shards of cfg2 2.5 M / 1.25 M rows 99.6 -> 93.2 / 61.3 -> 58.4 us against 4 blocks, 2 blocks
    // no better, square matrices of 1-6 M rows unchanged; tools/shard_run_sweep.sh, tools/run_min_mid.sh)
    const int64_t nb_run = nnz / ((int64_t) S * std::max(1, env_int("SPBLAS_GFX950_PB_RUN_MIN", 96)));
    int64_t nb = std::max<int64_t>(nb_min, std::min<int64_t>(nb_fill, nb_run));
    if (nb < 1)
      nb = 1;
    int64_t hh = cdiv(m > 0 ? m : 1, nb);
    if (h_env > 0 && h_env < max_rows)
      hh = h_env;
    if (hh > max_rows)
      hh = max_rows;
    if (hh < 1)
      hh = 1;
    H = (int) hh;
    NB = (int) cdiv(m > 0 ? m : 1, H);
  }
  if (vfree) {  // exactly the grid whose windows were measured
    H = vf_rows;
    NB = (int) cdiv(m, H);
  }
  if (h->bin_row_align > 1) {
    // caller wants bin boundaries on multiples of bin_row_align (stripe boundaries of the overlapped
    // multi-GPU step): use the largest divisor of it that fits the LDS budget, if a decent one exists
    int best = 0;
    for (int d = max_rows; d >= max_rows / 4 && d >= 1; --d)
      if (h->bin_row_align % d == 0) {
        best = d;
        break;
      }
    if (best > 0) {
      H = best;
      NB = (int) cdiv(m, H);
      pl->bin_aligned = 1;
    }
  }
  // Row-skewed matrices (power-law graphs: most entries in a few thousand rows, half the rows empty) get wave-bins
  // of VARIABLE height: a new bin every H rows and wherever the entry count crosses a multiple of E, so no bin is
  // taller than the LDS accumulators allow and none carries much more than E entries.  With equal heights the
  // R-MAT scale-24 matrix puts 8.9 M entries into one bin (average 39 k): one inspect workgroup and one reduce
  // wavefront per such bin (inspect 310 ms).  Uniform matrices keep the arithmetic bins (binrow = nullptr).
  const double avg_len = m > 0 ? (double) nnz / (double) m : 0.0;
  int varbins = env_int("SPBLAS_GFX950_PB_VARBINS", -1);
  if (varbins < 0)
    varbins = (double) pl->max_row_len > 16.0 * avg_len + 64.0 || pl->empty_rows * 4 > pl->m;
  if (h->bin_row_align > 1 || m < 2 || (NB < 2 && !compact))
    varbins = 0;
  if (compact)
    varbins = 1;  // the reduce maps compact rows through binrow[] + nzrow[]
  if (vfree)
    varbins = 0;
  const int32_t* binrow = nullptr;
  if (varbins) {
    int64_t E = nnz / std::max(1, env_int("SPBLAS_GFX950_PB_BINS", 2048));
    E = std::max<int64_t>(8192, std::min<int64_t>(E, 98304));
    int32_t* flag = nullptr;
    long long* fpart = nullptr;
    if ((rc = dev_alloc((void**) &flag, (size_t) (m + 1) * 4, s)))
      return rc;
    if ((rc = dev_alloc((void**) &fpart, (size_t) (cdiv(m, 2048) + 2) * sizeof(long long), s))) {
      dev_free(flag, s);
      return rc;
    }
    tr.mark("variable bins: temporaries allocated");
    hipLaunchKernelGGL((pb_bin_flags_kernel<O>), dim3((unsigned) cdiv(m, 256)), dim3(256), 0, s, m, H, E, rowptr, flag);
    long long* total_dev = scan_counts_i32(s, m, flag, fpart);
    tr.mark("variable bins: flags + scan");
    long long nb_var = 0;
    hipError_t e = hipMemcpyAsync(&nb_var, total_dev, sizeof(nb_var), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess)
      e = hipStreamSynchronize(s);
    tr.mark("variable bins: total read back");
    if (e == hipSuccess && nb_var > 0 && nb_var < ((int64_t) 1 << 24)) {
      NB = (int) nb_var;
      rc = dev_alloc(&pl->s_binrow, (size_t) (NB + 1) * 4, s);
      if (!rc) {
        hipLaunchKernelGGL(pb_bin_rows_kernel, dim3((unsigned) cdiv(m + 1, 256)), dim3(256), 0, s, m, flag,
                           static_cast<int32_t*>(pl->s_binrow));
      }
    }
    dev_free(flag, s);
    dev_free(fpart, s);
    if (e != hipSuccess)
      return hip_fail(e);
    if (rc)
      return rc;
    binrow = static_cast<const int32_t*>(pl->s_binrow);
    if (!binrow && compact)
      return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;  // the compacted rows are addressed through binrow[]
    if (!binrow)
      varbins = 0;
    pl->device_bytes += (size_t) (NB + 1) * 4;
    tr.mark("variable bins");
  }
  const int64_t nseg = (int64_t) S * NB;
  if (nseg > (int64_t) 64 << 20 || S > 16384)  // 2 * S ints of LDS per inspect workgroup
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  pl->n_slices = S;
  pl->slice_cols = W;
  pl->n_rblk = NB;
  pl->rows_per_blk = H;

  int32_t *cnt = nullptr, *aoff = nullptr, *prel = nullptr;
  long long* partials = nullptr;
  if ((rc = dev_alloc((void**) &cnt, (size_t) (nseg + 1) * 4, s)))
    return rc;
  pl->seg_ptr = cnt;
  tr.mark("cnt allocated");
  SPB_HIP(hipMemsetAsync(cnt, 0, (size_t) (nseg + 1) * 4, s));
  // rows longer than the nnz window (the plan's long_rows list) stay out of the tiles
  pl->hub_len = pl->n_long > 0 ? pl->win : 0;
  pl->s_hub_rows = pl->long_rows;
  pl->n_hub = pl->n_long;
  pl->hub_rows_owned = false;
  if (pl->split_len > 0) {  // every row of the tiles is at most split_len entries long
    pl->hub_len = 0;
    pl->s_hub_rows = nullptr;
    pl->n_hub = 0;
  } else if (varbins && pl->n_long > 0) {
    // variable bins keep all but the very longest rows in the tiles (a row of 16 k entries leaves ~20 entries in
    // each of its runs: duplicates the reduce adds atomically, not a serial chain): own list, higher threshold
    const int hub2 = std::max<int>(pl->win, env_int("SPBLAS_GFX950_PB_HUB_LEN", 16384));
    if (pl->max_row_len <= hub2) {
      pl->hub_len = 0;
      pl->s_hub_rows = nullptr;
      pl->n_hub = 0;
    } else if (hub2 > pl->win || compact) {  // (compaction: the list holds COMPACT row numbers)
      unsigned long long* st2 = nullptr;
      int32_t* rows2 = nullptr;
      if ((rc = dev_alloc((void**) &st2, 4 * sizeof(unsigned long long), s)))
        return rc;
      if ((rc = dev_alloc((void**) &rows2, (size_t) (nnz / hub2 + 1) * 4, s))) {
        dev_free(st2, s);
        return rc;
      }
      unsigned long long hst[4] = {0, 0, 0, 0};
      hipError_t e = hipMemsetAsync(st2, 0, 4 * sizeof(unsigned long long), s);
      if (e == hipSuccess) {
        hipLaunchKernelGGL((pb_hub_list_kernel<O>), dim3((unsigned) (cdiv(m, 256) < 2048 ? cdiv(m, 256) : 2048)), dim3(256),
                           0, s, m, hub2, rowptr, st2 + 1, rows2);
        e = hipMemcpyAsync(hst, st2, sizeof(hst), hipMemcpyDeviceToHost, s);
      }
      if (e == hipSuccess)
        e = hipStreamSynchronize(s);
      dev_free(st2, s);
      if (e != hipSuccess) {
        dev_free(rows2, s);
        return hip_fail(e);
      }
      pl->hub_len = hub2;
      pl->s_hub_rows = rows2;
      pl->n_hub = (int64_t) hst[1];
      pl->hub_rows_owned = true;
    }
  }
  pl->values_ptr = values_p;
  if (pl->hub_len > 0) {
    // workgroups per hub row: ~16K entries each, at most 64
    int64_t parts = cdiv(pl->max_row_len, 16384);
    pl->hub_parts = (int) (parts < 1 ? 1 : (parts > 64 ? 64 : parts));
    int rc_h = dev_alloc(&pl->s_hub_part, (size_t) pl->n_hub * pl->hub_parts * sizeof(T), s);
    if (rc_h)
      return rc_h;
  }
  // plans that are going to stage their scatter in order (one-byte row codes: large, at most PB_STAGE_SP slices, no rows
  // left out of the tiles) have the count pass count per part of the bin as well: the scatter then skips its own counting walk
  uint16_t* wpart = nullptr;
  if (S <= PB_STAGE_SP && pl->hub_len == 0 && NB >= 1536 && nnz >= ((int64_t) 32 << 20) && env_int("SPBLAS_GFX950_PB_ENC8", 1) &&
      env_int("SPBLAS_GFX950_PB_STAGED_SCATTER", 1) && env_int("SPBLAS_GFX950_PB_WPART", 1) &&
      dev_alloc((void**) &wpart, (size_t) NB * 16 * S * 2, s) != SPBLAS_GFX950_STATUS_SUCCESS)
    wpart = nullptr;
  hipLaunchKernelGGL((pb_count_kernel<O>), dim3((unsigned) NB), dim3(256), (size_t) S * 4 * (wpart ? 17 : 1), s, m, rowptr,
                     pl->colind, W, H, S, NB, cnt, pl->hub_len, binrow, wpart);
  tr.mark("count kernel");
  // One probe pass over the counters, read back once: entries per slice, non-empty tiles per slice, entries
  // per bin group.  AUTO uses them to decline matrices the plan does not suit; the work lists below use them
  // to spot column / row skew, and their total is the number of entries placed in tiles.
  const int64_t ngroups = cdiv(NB, RW);
  std::vector<unsigned long long> h_sum((size_t) (2 * S + ngroups + S));  // ..., then the longest run per slice
  unsigned long long* d_sum = nullptr;
  if ((rc = dev_alloc((void**) &d_sum, h_sum.size() * sizeof(unsigned long long), s))) {
    dev_free(wpart, s);
    return rc;
  }
  struct temp_guard {  // inspect temporaries are released on every exit path
    hipStream_t s;
    void* p[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    ~temp_guard() {
      for (void* q : p)
        dev_free(q, s);
    }
  } temps{s};
  temps.p[0] = d_sum;
  temps.p[7] = wpart;
  hipLaunchKernelGGL(pb_balance_kernel, dim3((unsigned) (S + ngroups)), dim3(256), 0, s, S, NB, RW, cnt, d_sum,
                     d_sum + S, d_sum + 2 * S, d_sum + 2 * S + ngroups);
  if ((rc = readback_add(h, h_sum.data(), d_sum, h_sum.size() * sizeof(unsigned long long))))
    return rc;
  // widest window of the caller's arrays a bin covers (entries of its rows, hub rows included): decides whether the staged
  // scatter may keep positions as 16-bit words
  unsigned long long* d_span = nullptr;
  unsigned long long h_span = ~0ull;
  if ((rc = dev_alloc((void**) &d_span, sizeof(unsigned long long), s)))
    return rc;
  temps.p[6] = d_span;
  SPB_HIP(hipMemsetAsync(d_span, 0, sizeof(unsigned long long), s));
  hipLaunchKernelGGL((pb_bin_span_kernel<O>), dim3((unsigned) cdiv(NB, 256)), dim3(256), 0, s, m, H, (int64_t) NB, rowptr, d_span,
                     binrow);
  if ((rc = readback_add(h, &h_span, d_span, sizeof(h_span))))
    return rc;
  // the block offsets of both orders are computed meanwhile (the probe read-back below is the only wait)
  if ((rc = dev_alloc((void**) &aoff, (size_t) (nseg + 1) * 4, s)))
    return rc;
  temps.p[1] = aoff;
  if ((rc = dev_alloc((void**) &prel, (size_t) (nseg + 1) * 4, s)))
    return rc;
  temps.p[2] = prel;
  if ((rc = dev_alloc((void**) &partials, (size_t) (cdiv(nseg, 2048) + 2) * sizeof(long long), s)))
    return rc;
  temps.p[3] = partials;
  if ((rc = dev_alloc(&pl->s_binblk, (size_t) (NB + 1) * 4, s)))
    return rc;
  if ((rc = dev_alloc(&pl->s_sliceblk, (size_t) (S + 1) * 4, s)))
    return rc;
  int32_t* binblk = static_cast<int32_t*>(pl->s_binblk);
  int32_t* sliceblk = static_cast<int32_t*>(pl->s_sliceblk);
  hipLaunchKernelGGL(pb_nblk_kernel, dim3((unsigned) cdiv(nseg, 256)), dim3(256), 0, s, nseg, cnt, aoff, pb_geom<T>::BLK);
  (void) scan_counts_i32(s, nseg, aoff, partials);  // aoff[nseg] = blocks in A' order
  int32_t* eoff = nullptr;  // first entry of every run in the compact A' stream (run lengths rounded up to 4)
  if ((rc = dev_alloc((void**) &eoff, (size_t) (nseg + 1) * 4, s)))
    return rc;
  temps.p[4] = eoff;
  hipLaunchKernelGGL(pb_ecnt_kernel, dim3((unsigned) cdiv(nseg, 256)), dim3(256), 0, s, nseg, cnt, eoff);
  (void) scan_counts_i32(s, nseg, eoff, partials);  // eoff[nseg] = entries of the compact stream
  // the same offsets bin-major, with the padded run lengths: what the value refresh reads (sliced_update_typed)
  if (!vfree && (pl->keep_src || pl->refresh_each_call || env_int("SPBLAS_GFX950_PB_KEEP_SRC", 0))) {
    if ((rc = dev_alloc(&pl->s_eoff, (size_t) nseg * sizeof(int2), s)))
      return rc;
    pl->device_bytes += (size_t) nseg * sizeof(int2);
    hipLaunchKernelGGL(pb_run_table_kernel, dim3((unsigned) cdiv(nseg, 256)), dim3(256), 0, s, S, (int64_t) NB, eoff,
                       static_cast<int2*>(pl->s_eoff));
  }
  hipLaunchKernelGGL(pb_bin_prefix_kernel, dim3((unsigned) NB), dim3(256), 0, s, S, NB, cnt, prel, binblk, pb_geom<T>::BLK);
  (void) scan_counts_i32(s, NB, binblk, partials);  // binblk[NB] = blocks in P order (bins padded to groups)
  hipLaunchKernelGGL(pb_slice_blocks_kernel, dim3((unsigned) cdiv(S + 1, 256)), dim3(256), 0, s, S, NB, aoff, sliceblk);
  std::vector<int32_t> h_sliceblk((size_t) S + 1);
  int32_t h_pblocks = 0, h_epad = 0;
  if ((rc = readback_add(h, h_sliceblk.data(), sliceblk, (size_t) (S + 1) * 4)) ||
      (rc = readback_add(h, &h_pblocks, binblk + NB, 4)) || (rc = readback_add(h, &h_epad, eoff + nseg, 4)) ||
      (rc = readback_flush(h)))
    return rc;
  unsigned long long placed_total = 0, max_slice = 0, max_group = 0, ne = 0, max_run = 0;
  for (int i = 0; i < S; ++i) {
    placed_total += h_sum[(size_t) i];
    max_slice = std::max(max_slice, h_sum[(size_t) i]);
    ne += h_sum[(size_t) (S + i)];
    max_run = std::max(max_run, h_sum[(size_t) (2 * S + ngroups + i)]);
  }
  for (int64_t g = 0; g < ngroups; ++g)
    max_group = std::max(max_group, h_sum[(size_t) (2 * S + g)]);
  if (placed_total > (unsigned long long) INT32_MAX)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  pl->s_placed = (int64_t) placed_total;
  if (auto_mode) {
    // a matrix whose entries cluster in few (slice, bin) tiles (banded, block structured) already gets
    // its x reuse from L2 with the CSR kernels -- decline
    if ((double) ne < 0.25 * (double) nseg)
      return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
    // ... and a matrix whose entries pile up in a few slices (hot columns) or a few bin groups (heavy
    // rows below the hub threshold) would leave most of the chip waiting for one expand / reduce
    // workgroup: decline when the heaviest slice or group carries more than 6x the average.
    const double mean_slice = (double) placed_total / (double) S, mean_group = (double) placed_total / (double) ngroups;
    if (placed_total > 0 &&
        ((double) max_slice > 6.0 * mean_slice + 65536.0 || (double) max_group > 6.0 * mean_group + 65536.0)) {
      // Round 2: the expand spreads a heavy slice over as many workgroups as its share asks for and variable-height
      // bins flatten heavy rows, so such a matrix CAN run well here (R-MAT scale 24 fp64: 2.4 vs 3.4 ms) -- but hot
      // columns also serve the row-block kernel from L2.  No static rule separates the two: the plan is finished
      // and plan_create times it against the row-block kernel.
      if (!env_int("SPBLAS_GFX950_AUTO_TRIAL", -1))
        return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
      pl->s_uncertain = 1;  // (plan_create decides: by rule since round 6, by the stopwatch with SPBLAS_GFX950_AUTO_TRIAL=1)
    }
  }
  tr.mark("probe + block offsets read back");
  const int64_t a_blocks = h_sliceblk[(size_t) S], p_blocks = h_pblocks;
  pl->a_blocks = a_blocks;
  pl->p_blocks = p_blocks;
  constexpr int PB_BLK = pb_geom<T>::BLK, PB_GBLK = pb_geom<T>::GBLK;
  // a_pad: entries of the compact A' stream + one block of slack (the lanes past the end of the last run's last block)
  const int64_t a_pad = (int64_t) h_epad + PB_BLK, p_pad = p_blocks * PB_BLK;
  pl->a_entries = a_pad;
  // 32-bit entry indices in the expand, 32-bit byte offsets into the product stream in the reduce; and the
  // padded copy must stay a small multiple of the matrix (runs of a few entries pad to a whole block: a matrix
  // that sparse per tile should not be tiled)
  if (p_pad > INT32_MAX - 1024 || (uint64_t) (p_pad + 1024) * sizeof(T) >= ((uint64_t) 1 << 32) ||
      (auto_mode && p_pad > 2 * nnz + 65536))
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  if (!vfree && (rc = dev_alloc((void**) &pl->s_values, (size_t) (a_pad + 8) * sizeof(T), s)))
    return rc;
  if ((rc = dev_alloc((void**) &pl->s_colind, (size_t) (a_pad + 8) * 2, s)))
    return rc;
  // Source positions (4 B per entry: what a value refresh gathers through) only for plans that are known to need them: a plan
  // that takes the values again on every multiply, or one whose caller HAS changed the values once (keep_src: the first
  // update of a plan built without them builds it again, spmv_sliced_update).  A plan that is inspected once and multiplied
  // many times -- the common case -- is a quarter smaller without them (cfg2: 1.61 -> 1.18 GB).
  const bool keep_src = !vfree && (pl->keep_src || pl->refresh_each_call || env_int("SPBLAS_GFX950_PB_KEEP_SRC", 0));
  if (keep_src && (rc = dev_alloc((void**) &pl->s_perm, (size_t) (a_pad + 8) * 4, s)))
    return rc;
  if (vfree && (rc = dev_alloc((void**) &pl->s_src, (size_t) (p_pad + PB_GRP) * 2, s)))
    return rc;
  if ((rc = dev_alloc((void**) &pl->s_blkdst, (size_t) (a_blocks + 8) * 4, s)) ||
      (rc = dev_alloc((void**) &pl->s_blksrc, (size_t) (a_blocks + 8) * 4, s)))
    return rc;
  // Row stream of the reduce: one-byte codes when the tiles are dense enough for them (the average row advance inside
  // a run is H * (non-empty tiles) / entries: beyond ~32 rows a bin's exception list (128) gets tight: e^-8 of ~50 k entries) and the staged scatter
  // (whose ordered staging leaves every run sorted by row) is in use; otherwise 16-bit rows.
  typedef typename pb_hdr<T>::type hdr_t;
  const bool staged = S <= PB_STAGE_MAX_S && env_int("SPBLAS_GFX950_PB_STAGED_SCATTER", 1);
  int enc8 = env_int("SPBLAS_GFX950_PB_ENC8", 1);
  // ... and only when the wave-bins alone fill the chip: with fewer wavefronts in flight (row shards of a multi-GPU run,
  // K-split plans) the longer decode chain per group is no longer hidden -- cfg2 shards on one box, one-byte codes vs
  // 16-bit rows: 5 M rows (2 034 bins) 156.9 vs 162.8 us, 2.5 M (1 017 bins) 103.3 vs 93.1, 1.25 M 63.2 vs 58.3.
  (void) max_run;
  if (vfree && !staged)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;  // (more than 2 048 slices: n > 80 M columns)
  if (!staged || placed_total == 0 || S > PB_STAGE_SP ||
      (enc8 == 1 && ((double) H * (double) ne > 32.0 * (double) placed_total || NB < 1536 ||
                     placed_total < (unsigned long long) 32 << 20)))  // below ~32 M entries the SpMV gains nothing (n = 1-3 M
    enc8 = 0;                                                        // at 10 per row: 55.6 / 78.9 / 102.0 vs 56.0 / 78.2 / 102.1 us) and inspect pays 25-50 % more
  pl->enc8 = enc8 ? 1 : 0;
  pl->exc_cap = PB_EXC_CAP;
  const size_t hdr_bytes = (size_t) (p_blocks + PB_GBLK) * sizeof(hdr_t);
  if (enc8) {
    if ((rc = dev_alloc((void**) &pl->s_code, (size_t) (p_pad + PB_GRP), s)) || (rc = dev_alloc(&pl->s_hdr, hdr_bytes, s)) ||
        (rc = dev_alloc((void**) &pl->s_exc_idx, (size_t) NB * PB_EXC_CAP * 4, s)) ||
        (rc = dev_alloc((void**) &pl->s_exc_row, (size_t) NB * PB_EXC_CAP * 2, s)) ||
        (rc = dev_alloc((void**) &pl->s_exc_cnt, (size_t) (NB + 1) * 4, s)))  // [NB] = enc_fail
      return rc;
  } else if ((rc = dev_alloc((void**) &pl->s_lrow, (size_t) (p_pad + PB_GRP) * 2, s))) {
    return rc;
  }
  if ((rc = dev_alloc((void**) &pl->s_products, (size_t) (p_pad + PB_GRP) * sizeof(T), s)))
    return rc;
  pl->s_products_bytes = (size_t) (p_pad + PB_GRP) * sizeof(T);
  pl->device_bytes += (size_t) a_pad * (vfree ? 2 : sizeof(T) + 2 + (keep_src ? 4 : 0)) + (vfree ? (size_t) p_pad * 2 : 0) +
                      (size_t) a_blocks * 8 +
                      (size_t) p_pad * sizeof(T) +
                      (enc8 ? (size_t) p_pad + hdr_bytes + (size_t) NB * PB_EXC_CAP * 6 : (size_t) p_pad * 2) +
                      (size_t) (NB + S + 2) * 4 + (size_t) nseg * 4;
  // pads: value 0, column 0, no source position, row = H (a dummy accumulator); products start finite
  // The staged scatter writes every entry of the compact stream, the pads of the runs' last quads included: only the slack
  // behind the stream is cleared (1 GB of memsets at cfg2: 0.15 of the 3.3 ms of a warm inspect).  The direct scatter
  // (more slices than the staged one takes) and SPBLAS_GFX950_PB_CLEAR=1 clear everything.
  const int64_t a_keep = (staged && !env_int("SPBLAS_GFX950_PB_CLEAR", 0)) ? (int64_t) h_epad & ~(int64_t) 63 : 0;
  if (!vfree)
    SPB_HIP(hipMemsetAsync(static_cast<T*>(pl->s_values) + a_keep, 0, (size_t) (a_pad + 8 - a_keep) * sizeof(T), s));
  SPB_HIP(hipMemsetAsync(static_cast<uint16_t*>(pl->s_colind) + a_keep, 0, (size_t) (a_pad + 8 - a_keep) * 2, s));
  if (keep_src)
    SPB_HIP(hipMemsetAsync(static_cast<int32_t*>(pl->s_perm) + a_keep, 0xFF, (size_t) (a_pad + 8 - a_keep) * 4, s));
  if (vfree)  // pads point at the first entry of the window (their row code keeps them out of the sums)
    SPB_HIP(hipMemsetAsync(pl->s_src, 0, (size_t) (p_pad + PB_GRP) * 2, s));
  SPB_HIP(hipMemsetAsync(pl->s_blkdst, 0, (size_t) (a_blocks + 8) * 4, s));
  SPB_HIP(hipMemsetAsync(pl->s_blksrc, 0, (size_t) (a_blocks + 8) * 4, s));
  if (enc8) {  // pads: code 255 (left out of the main pass); headers and exception counts start at 0
    SPB_HIP(hipMemsetAsync(pl->s_code, 0xFF, (size_t) (p_pad + PB_GRP), s));
    SPB_HIP(hipMemsetAsync(pl->s_hdr, 0, hdr_bytes, s));
    SPB_HIP(hipMemsetAsync(pl->s_exc_cnt, 0, (size_t) (NB + 1) * 4, s));
  } else {
    SPB_HIP(hipMemsetD16Async(reinterpret_cast<hipDeviceptr_t>(pl->s_lrow), (unsigned short) H, (size_t) (p_pad + PB_GRP), s));
  }
  // (the product stream needs no clearing: the expand writes every block of every run, and what lies between -- the pads that
  // round a bin up to whole groups -- is only ever added to a dummy accumulator; 0.43 GB of memset at cfg2)
  if (env_int("SPBLAS_GFX950_PB_CLEAR", 0))
    SPB_HIP(hipMemsetAsync(pl->s_products, 0, (size_t) (p_pad + PB_GRP) * sizeof(T), s));
  tr.mark("plan arrays allocated + cleared");
  pl->n_ksplit = pick_ksplit(NB, NB > 0 ? p_blocks / PB_GBLK / NB : 0);
  {
    // column skew: when one slice is far above the average the expand gets an explicit work list with
    // workgroups in proportion to the slice sizes (in blocks of A')
    const int64_t total = (int64_t) placed_total;
    const bool one_item_per_slice = slice_aligned && total > 0 && (double) max_slice * S <= 1.05 * (double) total;
    if (one_item_per_slice) {
      std::vector<int4> items;
      for (int i = 0; i < S; ++i)
        if (h_sliceblk[(size_t) i + 1] > h_sliceblk[(size_t) i])
          items.push_back(make_int4(i, h_sliceblk[(size_t) i], h_sliceblk[(size_t) i + 1], 0));
      if (!items.empty()) {
        if ((rc = dev_alloc(&pl->s_xitems, items.size() * sizeof(int4), s)))
          return rc;
        if ((rc = upload_add(h, pl->s_xitems, items.data(), items.size() * sizeof(int4))) || (rc = readback_flush(h)))
          return rc;
        pl->n_xitems = (int64_t) items.size();
      }
    } else if (total > 0 && (int64_t) max_slice * S > 3 * total) {
      const int cus = h->num_cus > 0 ? h->num_cus : 256;
      const int64_t target = std::max<int64_t>(cdiv(a_blocks, (int64_t) env_int("SPBLAS_GFX950_PB_XITEM_DIV", 2) * cus),
                                               4 * (int64_t) W / PB_BLK);
      std::vector<int4> items;
      for (int i = 0; i < S; ++i) {
        const int64_t lo = h_sliceblk[(size_t) i], hi = h_sliceblk[(size_t) i + 1];
        if (hi <= lo)
          continue;
        const int64_t np = std::max<int64_t>(1, cdiv(hi - lo, target));
        const int64_t per = cdiv(hi - lo, np);
        for (int64_t k = 0; k < np; ++k) {
          const int64_t a = std::min(lo + k * per, hi), b = std::min(lo + (k + 1) * per, hi);
          if (b > a)
            items.push_back(make_int4(i, (int) a, (int) b, 0));
        }
      }
      if (!items.empty()) {
        // workgroups are dispatched in list order: with more items than CUs the heaviest go first (longest processing
        // time first), so that no large item starts late and finishes alone (SPBLAS_GFX950_PB_LPT=0: list order = slice order)
        if (env_int("SPBLAS_GFX950_PB_LPT", 1))
          std::stable_sort(items.begin(), items.end(), [](const int4& a, const int4& b) { return a.z - a.y > b.z - b.y; });
        if ((rc = dev_alloc(&pl->s_xitems, items.size() * sizeof(int4), s)))
          return rc;
        if ((rc = upload_add(h, pl->s_xitems, items.data(), items.size() * sizeof(int4))) || (rc = readback_flush(h)))
          return rc;
        pl->n_xitems = (int64_t) items.size();
        if (tr.on) {  // list-scheduling makespan of the work list on `cus` workgroup slots, in blocks (+ one x slice per item)
          std::vector<int64_t> slot((size_t) cus, 0);
          const int64_t xcost = (int64_t) W / PB_BLK * (int64_t) sizeof(T) / (int64_t) (2 * sizeof(T) + 2);
          for (const int4& it : items) {
            auto mn = std::min_element(slot.begin(), slot.end());
            *mn += (it.z - it.y) + xcost;
          }
          const int64_t mk = *std::max_element(slot.begin(), slot.end());
          std::fprintf(stderr, "[inspect] expand work list: %zu items, %lld blocks, largest %d, makespan %lld = %.2f x the even share "
                       "(x slice ~ %lld blocks)\n", items.size(), (long long) a_blocks, items.empty() ? 0 : items[0].z - items[0].y,
                       (long long) mk, (double) mk * cus / (double) std::max<int64_t>(1, a_blocks), (long long) xcost);
        }
      }
    }
  }
  {
    // row skew: when one bin group is far above the average the reduce gets a work list: every group's wave-bin
    // streams are cut into as many equal parts as its share of the entries asks for; split groups write partial
    // rows into a compact buffer and pb_combine_items_kernel adds them in part order (bit reproducible).
    const int64_t total = (int64_t) placed_total, max_tot = (int64_t) max_group;
    const int force_items = env_int("SPBLAS_GFX950_PB_RITEMS", 0);  // experiment: work list whatever the skew; value = parts per 1/768 of the entries
    // variable-height bins with more workgroups than the chip holds at once (cfg4: 1 425 groups of 4 bins, 512 slots, the
    // heaviest group 0.87 of a slot's share): a work list ordered heaviest first, so that a heavy group never starts late
    const int cus_r = h->num_cus > 0 ? h->num_cus : 256;
    const bool lpt = env_int("SPBLAS_GFX950_PB_LPT", 1) && varbins && ngroups > 2 * (int64_t) cus_r && max_tot * ngroups > (3 * total) / 2;
    if (!vfree && total > 0 && (max_tot * ngroups > 3 * total || force_items > 0 || lpt) && ngroups > 1) {
      const int64_t target = std::max<int64_t>(total / (768 * (force_items > 0 ? force_items : 1)), 16384);
      const int64_t block = (int64_t) RW * H;  // values per partial block
      std::vector<int4> items, split;
      std::vector<int64_t> weight;  // entries per item (for the heaviest-first order)
      int64_t poff = 0;
      for (int64_t g = 0; g < ngroups; ++g) {
        const int64_t tot_g = (int64_t) h_sum[(size_t) (2 * S + g)];
        int64_t Kg = (tot_g + target / 2) / target;
        if (lpt && force_items <= 0 && max_tot * ngroups <= 3 * total)
          Kg = 1;  // ordering only: no group is heavy enough to be cut
        // a part should still hold a few groups of every wave-bin's stream
        Kg = std::max<int64_t>(1, std::min<int64_t>(Kg, std::max<int64_t>(1, tot_g / ((int64_t) RW * 4 * PB_GRP))));
        Kg = std::min<int64_t>(Kg, 256);
        if (Kg == 1 || poff + Kg * block > (int64_t) INT32_MAX) {  // offsets are 32-bit: stop splitting
          items.push_back(make_int4((int) g, 0, 1, -1));
          weight.push_back(tot_g);
          continue;
        }
        for (int64_t k = 0; k < Kg; ++k) {
          items.push_back(make_int4((int) g, (int) k, (int) Kg, (int) (poff + k * block)));
          weight.push_back(tot_g / Kg);
        }
        split.push_back(make_int4((int) g, (int) Kg, (int) poff, 0));
        poff += Kg * block;
      }
      if (env_int("SPBLAS_GFX950_PB_LPT", 1) && (int64_t) items.size() > 2 * (int64_t) cus_r) {
        std::vector<size_t> order(items.size());
        for (size_t i = 0; i < order.size(); ++i)
          order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return weight[a] > weight[b]; });
        std::vector<int4> sorted(items.size());
        for (size_t i = 0; i < order.size(); ++i)
          sorted[i] = items[order[i]];
        items.swap(sorted);
      }
      if (!split.empty() || lpt) {
        if ((rc = dev_alloc(&pl->s_ritems, items.size() * sizeof(int4), s)) ||
            (rc = dev_alloc(&pl->s_rsplit, split.size() * sizeof(int4), s)) ||
            (rc = dev_alloc(&pl->s_rpartial, (size_t) poff * sizeof(T), s)))
          return rc;
        if ((rc = upload_add(h, pl->s_ritems, items.data(), items.size() * sizeof(int4))) ||
            (rc = upload_add(h, pl->s_rsplit, split.data(), split.size() * sizeof(int4))) || (rc = readback_flush(h)))
          return rc;
        pl->n_ritems = (int64_t) items.size();
        pl->n_rsplit = (int64_t) split.size();
        pl->device_bytes += (size_t) poff * sizeof(T);
      }
    }
  }
  tr.mark("work lists");
  // the row table: one entry per 64 matrix entries of a bin (a bin too long for it searches all of its rows), at most 2 048
  const bool q16 = h_span < 65536ull && H < 65536 && env_int("SPBLAS_GFX950_PB_STAGE_Q16", 1);
  const int rt_len = (int) std::min<unsigned long long>(2048, ((h_span >> 6) + 2 + 63) & ~63ull);
  // bins in the order the scatter should take them: groups of RW bins by weight, heaviest first (variable-height bins with
  // more bins than CUs only; the weights are the group sums of the probe)
  int32_t* bin_order_dev = nullptr;
  if (varbins && NB > 2 * (h->num_cus > 0 ? h->num_cus : 256) && env_int("SPBLAS_GFX950_PB_LPT", 1)) {
    std::vector<int64_t> gorder((size_t) ngroups);
    for (int64_t g = 0; g < ngroups; ++g)
      gorder[(size_t) g] = g;
    std::stable_sort(gorder.begin(), gorder.end(), [&](int64_t a, int64_t b) {
      return h_sum[(size_t) (2 * S + a)] > h_sum[(size_t) (2 * S + b)];
    });
    std::vector<int32_t> order;
    order.reserve((size_t) NB);
    for (int64_t g : gorder)
      for (int64_t b = g * RW; b < std::min<int64_t>((g + 1) * RW, NB); ++b)
        order.push_back((int32_t) b);
    if ((rc = dev_alloc((void**) &bin_order_dev, (size_t) NB * 4, s)))
      return rc;
    temps.p[5] = bin_order_dev;
    if ((rc = upload_add(h, bin_order_dev, order.data(), (size_t) NB * 4)) || (rc = readback_flush(h)))
      return rc;
  }
  auto stage_cap = [&](bool e8) {
    const size_t qb = q16 ? 2 : 4;  // bytes per staged position / row offset / row table entry
    return (int) (((size_t) PB_STAGE_LDS - (size_t) 16 * S - qb * (size_t) (H + 1) - qb * (size_t) rt_len - 256 -
                   (e8 ? (size_t) (PB_STAGE_THREADS / 64) * PB_STAGE_SP * 4 : 0)) /
                  (qb + 2 + sizeof(T))) & ~7;
  };
  auto launch_staged = [&](bool e8) {
    const int cap = stage_cap(e8);
    const void* fn = e8 ? (q16 ? reinterpret_cast<const void*>(pb_scatter_staged_kernel<T, O, true, true>)
                               : reinterpret_cast<const void*>(pb_scatter_staged_kernel<T, O, true, false>))
                        : (q16 ? reinterpret_cast<const void*>(pb_scatter_staged_kernel<T, O, false, true>)
                               : reinterpret_cast<const void*>(pb_scatter_staged_kernel<T, O, false, false>));
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PB_STAGE_LDS - 64);
    if (e != hipSuccess)
      return e;
    int64_t mm = m;
    const int32_t* ci = pl->colind;
    const T* vp = static_cast<const T*>(values_p);
    T* sv = static_cast<T*>(pl->s_values);
    uint16_t *sc = reinterpret_cast<uint16_t*>(pl->s_colind), *sr = pl->s_lrow;
    int32_t *pm = reinterpret_cast<int32_t*>(pl->s_perm), *bd = static_cast<int32_t*>(pl->s_blkdst);
    int hub = pl->hub_len, cap_ = cap, rt_ = rt_len, W_ = W, H_ = H, S_ = S, NB_ = NB, ecap = PB_EXC_CAP;
    unsigned char* code = pl->s_code;
    hdr_t* hdr = static_cast<hdr_t*>(pl->s_hdr);
    unsigned* ei = pl->s_exc_idx;
    uint16_t* er = pl->s_exc_row;
    int32_t *ec = pl->s_exc_cnt, *fail = pl->s_exc_cnt ? pl->s_exc_cnt + NB : nullptr;
    const int32_t *cnt_ = cnt, *aoff_ = aoff, *prel_ = prel, *binblk_ = binblk, *eoff_ = eoff;
    int32_t* bs = static_cast<int32_t*>(pl->s_blksrc);
    const int32_t* bo = bin_order_dev;
    uint16_t* ssrc = pl->s_src;
    // (the per-part counts are 16-bit: only with 16-bit staging is no part of a bin long enough to overflow them)
    const uint16_t* wp = (e8 && q16) ? wpart : nullptr;
    void* args[] = {&mm, &rowptr, &ci, &vp, &W_, &H_, &S_, &NB_, &cnt_, &aoff_, &prel_, &binblk_, &sv, &sc, &sr, &pm, &bd,
                    &hub, &cap_, &rt_, &binrow, &code, &hdr, &ei, &er, &ec, &ecap, &fail, &eoff_, &bs, &bo, &ssrc, &wp};
    return hipLaunchKernel(fn, dim3((unsigned) NB), dim3(PB_STAGE_THREADS), args, (size_t) PB_STAGE_LDS - 64, s);
  };
  if (staged) {
    SPB_HIP(launch_staged(enc8 != 0));
  } else {
    SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pb_scatter_kernel<T, O>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8));
    hipLaunchKernelGGL((pb_scatter_kernel<T, O>), dim3((unsigned) NB), dim3(256), (size_t) S * 8, s, m, rowptr,
                       pl->colind, static_cast<const T*>(values_p), W, H, S, NB, aoff, prel, binblk,
                       static_cast<T*>(pl->s_values), reinterpret_cast<uint16_t*>(pl->s_colind), pl->s_lrow,
                       reinterpret_cast<int32_t*>(pl->s_perm), static_cast<int32_t*>(pl->s_blkdst), pl->hub_len, binrow,
                       static_cast<const int32_t*>(eoff), static_cast<int32_t*>(pl->s_blksrc));
  }
  tr.mark("scatter");
  SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pb_flag_dups_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_BYTES + 16 * 1024));
  SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pb_flag_dups8_kernel<T>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_BYTES + 16 * 1024));
  auto launch_flags = [&]() {
    if (pl->enc8)
      hipLaunchKernelGGL((pb_flag_dups8_kernel<T>), dim3((unsigned) cdiv(NB, PB_FWAVES)), dim3(PB_FTHREADS),
                         (size_t) PB_FWAVES * (size_t) ((H + 64 + 63) & ~63), s, H, (int64_t) NB, binblk, pl->s_code,
                         static_cast<hdr_t*>(pl->s_hdr));
    else
      hipLaunchKernelGGL(pb_flag_dups_kernel, dim3((unsigned) cdiv(NB, PB_FWAVES)), dim3(PB_FTHREADS),
                         (size_t) PB_FWAVES * (size_t) ((H + 63) & ~63), s, H, (int64_t) NB, binblk, pl->s_lrow, PB_GBLK);
  };
  launch_flags();
  SPB_HIP(hipGetLastError());
  int32_t enc_fail = 0;
  if (pl->enc8) {
    if ((rc = readback_add(h, &enc_fail, pl->s_exc_cnt + NB, 4)) || (rc = readback_flush(h)))
      return rc;
  } else {
    SPB_HIP(hipStreamSynchronize(s));
  }
  if (pl->enc8 && (enc_fail || env_int("SPBLAS_GFX950_PB_ENC8_FAIL", 0))) {
    // a wave-bin with more exceptions than its list holds, or a run that could not be sorted: the same plan with
    // 16-bit rows (the tiles, offsets and work lists stay; the scatter and the duplicate flags run again).  Only a plan
    // that WAS built with one-byte codes takes this branch (the test hook alone must not re-allocate live 16-bit rows).
    pl->device_bytes -= (size_t) p_pad + hdr_bytes + (size_t) NB * PB_EXC_CAP * 6;
    pl->device_bytes += (size_t) p_pad * 2;
    dev_free(pl->s_code, s);
    dev_free(pl->s_hdr, s);
    dev_free(pl->s_exc_idx, s);
    dev_free(pl->s_exc_row, s);
    dev_free(pl->s_exc_cnt, s);
    pl->s_code = nullptr;
    pl->s_hdr = nullptr;
    pl->s_exc_idx = nullptr;
    pl->s_exc_row = nullptr;
    pl->s_exc_cnt = nullptr;
    pl->enc8 = 0;
    if ((rc = dev_alloc((void**) &pl->s_lrow, (size_t) (p_pad + PB_GRP) * 2, s)))
      return rc;
    SPB_HIP(hipMemsetD16Async(reinterpret_cast<hipDeviceptr_t>(pl->s_lrow), (unsigned short) H, (size_t) (p_pad + PB_GRP), s));
    SPB_HIP(launch_staged(false));
    launch_flags();
    SPB_HIP(hipGetLastError());
    SPB_HIP(hipStreamSynchronize(s));
    tr.mark("fallback to 16-bit rows");
  }
  tr.mark("flags");
  if (tr.on)  // where the streams of the launch pair live (round 3: the same plan runs 292-315 us depending on where they land)
    std::fprintf(stderr, "[inspect] arrays: s_val %p s_col %p blkdst %p blksrc %p P %p rows %p hdr %p  (a_pad %lld, p_pad %lld)\n",
                 pl->s_values, pl->s_colind, pl->s_blkdst, pl->s_blksrc, pl->s_products,
                 pl->enc8 ? (void*) pl->s_code : (void*) pl->s_lrow, pl->s_hdr, (long long) a_pad, (long long) p_pad);
  SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pb_expand_kernel<T, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, xlds > PB_LDS_BYTES ? xlds : PB_LDS_BYTES));
  SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pb_expand_kernel<T, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, xlds > PB_LDS_BYTES ? xlds : PB_LDS_BYTES));
  if (vfree) {
    SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pb_expand_kernel<T, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, xlds > PB_LDS_BYTES ? xlds : PB_LDS_BYTES));
    SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pb_expand_kernel<T, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, xlds > PB_LDS_BYTES ? xlds : PB_LDS_BYTES));
    for (int ub : {2, 4})
      SPB_HIP(hipFuncSetAttribute(pb_reduce_vf_fn<T>(pl->vf_waves, ub, pl->enc8 != 0), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  VF_LDS));
    return SPBLAS_GFX950_STATUS_SUCCESS;
  }
  for (int ub : {1, 2, 4, 8})
    SPB_HIP(hipFuncSetAttribute(pb_reduce_fn<T>(pl->rwaves, ub, pl->enc8 != 0), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spmv_sliced_build(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values, bool auto_mode) {
  // Row-skewed matrices (the ones that get variable-height bins below) are usually column-skewed too (graphs): try to take
  // the entries of the most referenced columns out of the tiles first (spmv_hot.hip).  Not for the halves of a split plan
  // themselves, not when the caller wants row-range reduces (the split plan multiplies all rows in one call).
  {
    const int hot = env_int("SPBLAS_GFX950_PB_HOT", -1);
    const double avg = pl->m > 0 ? (double) pl->nnz / (double) pl->m : 0.0;
    const bool skewed = (double) pl->max_row_len > 16.0 * avg + 64.0 || pl->empty_rows * 4 > pl->m;
    // (is_child counts the splits above this plan: the remainder of a split may be split again -- the next most referenced
    // columns -- up to SPBLAS_GFX950_PB_HOT_DEPTH levels; the window structures of a hot part are never split, is_child < 0)
    const int depth = env_int("SPBLAS_GFX950_PB_HOT_DEPTH", 1);
    if (pl->is_child >= 0 && pl->is_child < depth && hot != 0 && h->bin_row_align <= 1 && pl->nnz > 0 &&
        pl->nnz <= INT32_MAX - 8 && pl->m >= 2 &&
        (hot == 1 || (skewed && pl->nnz >= (4 << 20) && env_int("SPBLAS_GFX950_PB_VARBINS", -1) != 0))) {
      // (a pre-summing plan -- one product per (row, x slice) pair through the product stream -- was built and measured in
      // round 4: correct, 2.39 ms against 1.57 - 1.75 ms at cfg4; removed again, profiles/r04_presum.md has the numbers
      // and the commit that holds the code)
      const int rc_h = spmv_hot_build(h, pl, values, auto_mode);
      // (the split needs roughly two more copies of A: when THAT does not fit, the ordinary tiled plan -- or, under AUTO, the
      // row-block plan -- still may; round-4 advisor finding)
      if (rc_h == SPBLAS_GFX950_STATUS_ALLOC_FAILED) {
        spmv_hot_free(h, pl);
        (void) hipGetLastError();
      } else if (rc_h != SPBLAS_GFX950_STATUS_NOT_SUPPORTED) {
        return rc_h;
      }
    }
  }
  const bool f32 = pl->value_type == SPBLAS_GFX950_F32, o32 = pl->offset_type == SPBLAS_GFX950_I32;
  if (f32)
    return o32 ? sliced_build_typed<float, int32_t>(h, pl, values, auto_mode)
               : sliced_build_typed<float, int64_t>(h, pl, values, auto_mode);
  return o32 ? sliced_build_typed<double, int32_t>(h, pl, values, auto_mode)
             : sliced_build_typed<double, int64_t>(h, pl, values, auto_mode);
}

template <typename T>
static int sliced_update_typed(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values) {
  pl->values_ptr = values;  // the hub rows read the caller's array directly
  const int64_t a_pad = pl->a_entries;
  if (pl->s_placed == 0 || a_pad == 0 || pl->vfree)  // (value-free tiles: the reduce reads the caller's array itself)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  // bin by bin through LDS when the plan kept its run offsets (SPBLAS_GFX950_PB_UPDATE_BINS=0: the gather in A' order)
  if (pl->s_eoff && pl->n_rblk > 0 && pl->n_slices > 0 && env_int("SPBLAS_GFX950_PB_UPDATE_BINS", 1)) {
    const int cap = (int) ((PB_STAGE_LDS - 1024 - (((size_t) pl->n_slices * 8 + 15) & ~(size_t) 15)) / sizeof(T)) & ~3;
    const bool o32 = pl->offset_type == SPBLAS_GFX950_I32;
    const void* rp = pl->s_nzrow ? pl->s_rowptr_c : pl->rowptr;  // (both hold positions in the caller's arrays)
    const int64_t rows = pl->s_nzrow ? pl->s_m : pl->m;
    const int32_t* binrow = static_cast<const int32_t*>(pl->s_binrow);
    const void* fn = o32 ? (const void*) pb_refresh_bins_kernel<T, int32_t> : (const void*) pb_refresh_bins_kernel<T, int64_t>;
    // (once per device, offset type and value type -- this function is a template --: a plan without the snapshot opt-in
    // comes here on every multiply)
    static std::atomic<bool> attr_set[64][2] = {};  // (written on every multiply of a refreshing plan, from any thread)
    const int dev = h->device >= 0 && h->device < 64 ? h->device : 0;
    if (!attr_set[dev][o32 ? 0 : 1].load(std::memory_order_acquire) || h->device >= 64) {
      SPB_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PB_STAGE_LDS - 1024));
      attr_set[dev][o32 ? 0 : 1].store(true, std::memory_order_release);
    }
    if (o32)
      hipLaunchKernelGGL((pb_refresh_bins_kernel<T, int32_t>), dim3((unsigned) pl->n_rblk), dim3(1024), (size_t) PB_STAGE_LDS - 1024,
                         h->stream, pl->n_slices, pl->n_rblk, pl->rows_per_blk, rows, binrow, static_cast<const int32_t*>(rp),
                         static_cast<const int2*>(pl->s_eoff), reinterpret_cast<const int32_t*>(pl->s_perm),
                         static_cast<const T*>(values), static_cast<T*>(pl->s_values), cap);
    else
      hipLaunchKernelGGL((pb_refresh_bins_kernel<T, int64_t>), dim3((unsigned) pl->n_rblk), dim3(1024), (size_t) PB_STAGE_LDS - 1024,
                         h->stream, pl->n_slices, pl->n_rblk, pl->rows_per_blk, rows, binrow, static_cast<const int64_t*>(rp),
                         static_cast<const int2*>(pl->s_eoff), reinterpret_cast<const int32_t*>(pl->s_perm),
                         static_cast<const T*>(values), static_cast<T*>(pl->s_values), cap);
    SPB_HIP(hipGetLastError());
    return SPBLAS_GFX950_STATUS_SUCCESS;
  }
  hipLaunchKernelGGL((pb_update_values_kernel<T>), dim3((unsigned) cdiv(a_pad, 256)), dim3(256), 0, h->stream, a_pad,
                     reinterpret_cast<const int32_t*>(pl->s_perm), static_cast<const T*>(values),
                     static_cast<T*>(pl->s_values));
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spmv_sliced_update(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values) {
  // A plan built without source positions (the default: see keep_src in sliced_build_typed) meets its first change of
  // values: it is built again from the caller's arrays, this time WITH them, so that every later change is the cheap
  // gather.  Inspect-class work: not inside a stream capture.
  const bool have_src = pl->rest_plan ? pl->hot_src != nullptr : (pl->vfree || pl->s_perm != nullptr || pl->s_placed == 0);
  if (!have_src && !pl->is_child) {
    if (stream_capturing(h->stream))
      return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
    if (pl->used && pl->last_stream != h->stream)
      SPB_HIP(hipStreamSynchronize(pl->last_stream));
    const int nt = pl->nt_products, refresh = pl->refresh_each_call;
    spmv_sliced_free(h, pl);
    pl->device_bytes = pl->base_device_bytes;
    pl->keep_src = 1;
    pl->refresh_each_call = refresh;
    int rc = spmv_sliced_build(h, pl, values, false);
    pl->nt_products = nt;
    if (rc == SPBLAS_GFX950_STATUS_SUCCESS && env_int("SPBLAS_GFX950_TEST_FAIL_REBUILD", 0))
      rc = SPBLAS_GFX950_STATUS_ALLOC_FAILED;  // test hook: the failure path below with a fully built plan to tear down
    if (rc != SPBLAS_GFX950_STATUS_SUCCESS) {
      // The second build needs 4 B per entry more than the first and may not fit (or decline).  A half-built tiled plan
      // must not stay behind: the plan goes back to the structures plan_build always makes (row-block windows / the
      // plan-free kernel, both of which read the caller's arrays of each call -- the values just handed over included), as
      // plan_create does when the tiles cannot be built at inspect.  Out-of-memory and "declined" are absorbed that way; any
      // other error is reported, with the plan equally consistent.
      spmv_sliced_free(h, pl);
      pl->device_bytes = pl->base_device_bytes;
      pl->values_ptr = nullptr;
      pl->refresh_each_call = 0;
      pl->vfree = 0;
      pl->keep_src = 0;
      pl->alg = pl->nnz >= pl->m / 2 ? SPBLAS_GFX950_SPMV_ROWBLOCK : SPBLAS_GFX950_SPMV_VECTOR;
      if (rc == SPBLAS_GFX950_STATUS_ALLOC_FAILED || rc == SPBLAS_GFX950_STATUS_NOT_SUPPORTED) {
        (void) hipGetLastError();
        return SPBLAS_GFX950_STATUS_SUCCESS;
      }
    }
    return rc;
  }
  if (pl->rest_plan)
    return spmv_hot_update(h, pl, values);
  return pl->value_type == SPBLAS_GFX950_F32 ? sliced_update_typed<float>(h, pl, values)
                                             : sliced_update_typed<double>(h, pl, values);
}

// parts every wave-bin's stream is cut into when there are too few wave-bins to fill the chip (a row shard of a
// multi-GPU run): at least 4 wavefronts per CU in flight, but a part keeps >= 8 steps of 256 entries
static int pick_ksplit(int64_t waves, int64_t steps_per_bin) {
  int K = env_int("SPBLAS_GFX950_PB_KSPLIT", 0);
  if (K <= 0) {
    K = 1;
    // ~700 wavefronts fill the chip for this kernel; more parts only add partial sums to combine (row shards of
    // cfg2, tools/shard_sweep.sh: 190 bins K = 4 / 8: 40.9 / 44.5 us, 248 bins: 69.0 / 77.0 us)
    while (waves * K < 700 && K < 32 && steps_per_bin / (2 * K) >= 8)
      K *= 2;
  }
  return K < 1 ? 1 : K;
}

template <typename T>
static int sliced_expand_typed(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* x) {
  pl->last_x = x;  // the hub rows are computed in the reduce stage and gather x themselves
  pl->last_stream = h->stream;
  pl->used = true;
  // one wave of workgroups (2 per CU with 80 KiB slices, 1 with 160 KiB), but never shares so small that
  // re-loading the x slice dominates
  const int4* items = static_cast<const int4*>(pl->s_xitems);
  const int cus = h->num_cus > 0 ? h->num_cus : 256;
  const size_t xbytes = (size_t) pl->slice_cols * sizeof(T);
  int64_t nwg = (int64_t) cus * (xbytes > (size_t) PB_LDS_BYTES ? 1 : 2);
  const int64_t total = pl->a_blocks;
  if (total == 0)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  const int64_t min_share = std::max<int64_t>(1, 2 * (int64_t) pl->slice_cols / pb_geom<T>::BLK);
  if (nwg * min_share > total) {
    // ... but not fewer workgroups than slices (row shards: many slices with few entries each)
    const int64_t floor_wg = pl->n_slices < nwg ? pl->n_slices : nwg;
    nwg = total / min_share;
    if (nwg < floor_wg)
      nwg = floor_wg;
  }
  if (nwg < 1)
    nwg = 1;
  if (env_int("SPBLAS_GFX950_PB_DBG", 0) & 4)
    return SPBLAS_GFX950_STATUS_SUCCESS;  // timing experiment: the reduce alone
  // one-shot from spblas_gfx950_spmv_step_bcast_chunked: this expand waits, slice by slice, for the peers' chunks of x
  pb_chunk_wait cw = {nullptr, nullptr, 0, 0, 0, 0, 0, nullptr};
  int rot_ranks = 0, rot_of = 1, cap = 0;
  {
    auto& hw = h->chunk_wait;
    if (hw.flags) {
      cw = {hw.flags, hw.chunk_rows, hw.n_ranks, hw.chunks, hw.rank, hw.step, hw.timeout_ticks, hw.status_dev};
      rot_ranks = hw.rank;
      rot_of = hw.n_ranks > 0 ? hw.n_ranks : 1;
      cap = hw.max_wgs;
      hw.flags = nullptr;
    }
  }
  if (cap > 0 && nwg > cap)
    nwg = cap;
  const int share = (int) cdiv(total, nwg);
  dim3 grid = items ? dim3((unsigned) pl->n_xitems) : dim3((unsigned) cdiv(total, share));
  if (items && cap > 0 && (int64_t) cap < pl->n_xitems)
    grid = dim3((unsigned) cap);
  // (a waiting expand starts every rank on the work that reads its OWN rows of x -- nothing to wait for -- and goes round
  // the ranks from there: the slices follow the rows of x, and so do the items and the shares of A')
  const int rot = (int) ((int64_t) rot_ranks * (items ? pl->n_xitems : (int64_t) grid.x) / rot_of);
  if (pl->vfree) {
    if (pl->nt_products)
      hipLaunchKernelGGL((pb_expand_kernel<T, true, true>), grid, dim3(PB_THREADS), xbytes, h->stream, pl->n, pl->slice_cols,
                         static_cast<const int32_t*>(pl->s_sliceblk), static_cast<const T*>(nullptr),
                         reinterpret_cast<const uint16_t*>(pl->s_colind), static_cast<const int32_t*>(pl->s_blkdst),
                         static_cast<const T*>(x), static_cast<T*>(pl->s_products), items, (int) pl->n_slices, share,
                         static_cast<const int32_t*>(pl->s_blksrc), (int) pl->n_xitems, rot, cw);
    else
      hipLaunchKernelGGL((pb_expand_kernel<T, false, true>), grid, dim3(PB_THREADS), xbytes, h->stream, pl->n, pl->slice_cols,
                         static_cast<const int32_t*>(pl->s_sliceblk), static_cast<const T*>(nullptr),
                         reinterpret_cast<const uint16_t*>(pl->s_colind), static_cast<const int32_t*>(pl->s_blkdst),
                         static_cast<const T*>(x), static_cast<T*>(pl->s_products), items, (int) pl->n_slices, share,
                         static_cast<const int32_t*>(pl->s_blksrc), (int) pl->n_xitems, rot, cw);
  } else if (pl->nt_products)
    hipLaunchKernelGGL((pb_expand_kernel<T, true>), grid, dim3(PB_THREADS), xbytes, h->stream, pl->n, pl->slice_cols,
                       static_cast<const int32_t*>(pl->s_sliceblk), static_cast<const T*>(pl->s_values),
                       reinterpret_cast<const uint16_t*>(pl->s_colind), static_cast<const int32_t*>(pl->s_blkdst),
                       static_cast<const T*>(x), static_cast<T*>(pl->s_products), items, (int) pl->n_slices, share,
                       static_cast<const int32_t*>(pl->s_blksrc), (int) pl->n_xitems, rot, cw);
  else
    hipLaunchKernelGGL((pb_expand_kernel<T, false>), grid, dim3(PB_THREADS), xbytes, h->stream, pl->n, pl->slice_cols,
                       static_cast<const int32_t*>(pl->s_sliceblk), static_cast<const T*>(pl->s_values),
                       reinterpret_cast<const uint16_t*>(pl->s_colind), static_cast<const int32_t*>(pl->s_blkdst),
                       static_cast<const T*>(x), static_cast<T*>(pl->s_products), items, (int) pl->n_slices, share,
                       static_cast<const int32_t*>(pl->s_blksrc), (int) pl->n_xitems, rot, cw);
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

// Host copy of binrow[] for callers that reduce a PART of the rows (the whole-range path never needs it).  Fetched on
// first use: a 38 KB device-to-host copy is the first SDMA transfer of many processes and cost 15 ms inside inspect.
static int ensure_host_binrow(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl) {
  if (!pl->s_binrow || pl->h_binrow)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  const size_t bytes = (size_t) (pl->n_rblk + 1) * 4;
  readback_scope rb_scope(h);  // hb / ho are freed on the error paths below
  int32_t* hb = static_cast<int32_t*>(std::malloc(bytes));
  int32_t* ho = pl->s_nzrow ? static_cast<int32_t*>(std::malloc(bytes)) : nullptr;
  int32_t* d_orig = nullptr;
  if (!hb || (pl->s_nzrow && !ho)) {
    std::free(hb);
    std::free(ho);
    return SPBLAS_GFX950_STATUS_ALLOC_FAILED;
  }
  int rc = readback_add(h, hb, pl->s_binrow, bytes);
  if (!rc && pl->s_nzrow) {  // compaction: the original row every bin starts at
    rc = dev_alloc((void**) &d_orig, bytes, h->stream);
    if (!rc) {
      hipLaunchKernelGGL(pb_bin_orig_rows_kernel, dim3((unsigned) cdiv(pl->n_rblk + 1, 256)), dim3(256), 0, h->stream,
                         pl->n_rblk, pl->s_m, pl->m, static_cast<const int32_t*>(pl->s_binrow),
                         static_cast<const int32_t*>(pl->s_nzrow), d_orig);
      rc = readback_add(h, ho, d_orig, bytes);
    }
  }
  if (!rc)
    rc = readback_flush(h);
  dev_free(d_orig, h->stream);
  if (rc) {
    std::free(hb);
    std::free(ho);
    return rc;
  }
  pl->h_binrow = hb;
  pl->h_binrow_orig = ho;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

// rows of the wave-bins [wb_begin, wb_end):  y = alpha * (products of the last expand) + beta * y
template <typename T>
static int sliced_reduce_typed(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* alpha_p,
                               const void* beta_p, void* y, int64_t wb_begin, int64_t wb_end,
                               void* const* peers_p = nullptr, int n_peers = 0, int64_t peer_off = 0) {
  hipStream_t s = h->stream;
  if (wb_end <= wb_begin)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  pl->last_stream = s;
  pl->used = true;
  if (peers_p && pl->hub_len > 0 && pl->n_hub > 0)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;  // the fused all-gather epilogue does not cover hub rows
  const T alpha = *static_cast<const T*>(alpha_p), beta = *static_cast<const T*>(beta_p);
  const int RW = pl->rwaves;
  // groups per batch (two batches in flight): 4 for the 16-bit rows; the one-byte codes run best with 2 (cfg2, same box:
  // reduce 118.4 / 124.6 / 129.5 us for 2 / 4 / 8)
  int UB = env_int("SPBLAS_GFX950_PB_RBATCH", pl->enc8 ? 2 : 4);
  if (UB != 1 && UB != 2 && UB != 8)
    UB = 4;
  const int64_t groups = cdiv(wb_end - wb_begin, RW);
  int K = pick_ksplit(wb_end - wb_begin, pl->n_rblk > 0 ? pl->p_blocks / pb_geom<T>::GBLK / pl->n_rblk : 0);
  if (h->max_ksplit > 0 && K > h->max_ksplit)
    K = (int) h->max_ksplit;  // striped callers run several reduces side by side
  if (pl->vfree) {
    // (the multi-GPU steps are not handed A's values: a value-free plan multiplies with the array registered at plan_create,
    // the last spblas_gfx950_spmv or plan_update_values -- its contents at the time of the step)
    if (!pl->values_ptr)
      return SPBLAS_GFX950_STATUS_INVALID_VALUE;
    K = 1;
  }
  // [r_lo, r_hi): the rows of these bins in the index space the tiles were built over (compact rows when the empty
  // rows were taken out); [o_lo, o_hi): the same range in rows of y (hub rows and empty rows are listed by those)
  int64_t r_lo = wb_begin * pl->rows_per_blk;
  int64_t r_hi = wb_end * pl->rows_per_blk < pl->s_m ? wb_end * pl->rows_per_blk : pl->s_m;
  int64_t o_lo = r_lo, o_hi = r_hi;
  if (pl->s_binrow) {  // variable bins
    if (wb_begin == 0 && wb_end == pl->n_rblk) {
      r_lo = o_lo = 0;
      r_hi = pl->s_m;
      o_hi = pl->m;
    } else {
      const int rc_b = ensure_host_binrow(h, pl);
      if (rc_b)
        return rc_b;
      r_lo = o_lo = pl->h_binrow[wb_begin];
      r_hi = o_hi = pl->h_binrow[wb_end];
      if (pl->h_binrow_orig) {  // (empty rows ahead of the first non-empty row belong to the first bin)
        o_lo = wb_begin == 0 ? 0 : pl->h_binrow_orig[wb_begin];
        o_hi = pl->h_binrow_orig[wb_end];
      }
    }
  }
  const int32_t* rowmap = static_cast<const int32_t*>(pl->s_nzrow);
  if (peers_p && rowmap)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;  // the fused all-gather epilogue writes contiguous rows
  // the pieces of a split row may lie in different bins: their sums only meet when every bin has been reduced
  if (pl->n_split > 0 && !(wb_begin == 0 && wb_end == pl->n_rblk))
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  T* piece_out = static_cast<T*>(pl->s_piece_out);
  const bool use_items = pl->s_ritems && !peers_p && wb_begin == 0 && wb_end == pl->n_rblk;
  if (!use_items && K > 1 && pl->s_partial_k < K) {  // grow the partial-sum workspace (stream ordered)
    if (stream_capturing(s))
      return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
    dev_free(pl->s_partial, s);
    pl->s_partial = nullptr;
    int rc = dev_alloc(&pl->s_partial, (size_t) K * pl->s_m * sizeof(T), s);
    if (rc)
      return rc;
    pl->s_partial_k = K;
  }
  if (pl->vfree) {
    // one workgroup per bin: the bin's window of the caller's values in LDS, vf_waves wavefronts on the bin's stream
    const int32_t* binblk = static_cast<const int32_t*>(pl->s_binblk);
    const T* Pp = static_cast<const T*>(pl->s_products);
    const uint16_t* rowp = pl->enc8 ? reinterpret_cast<const uint16_t*>(pl->s_code) : pl->s_lrow;
    const uint16_t* srcp = pl->s_src;
    const T* vals = static_cast<const T*>(pl->values_ptr);
    const void* rp = pl->rowptr;
    int o64 = pl->offset_type == SPBLAS_GFX950_I32 ? 0 : 1;
    typedef typename pb_hdr<T>::type hdr_t;
    const hdr_t* hdr = static_cast<const hdr_t*>(pl->s_hdr);
    const unsigned* exc_idx = pl->s_exc_idx;
    const uint16_t* exc_row = pl->s_exc_row;
    const int32_t* exc_cnt = pl->s_exc_cnt;
    int exc_cap = pl->exc_cap, Hw = pl->rows_per_blk, cap = pl->vf_win_cap;
    int64_t mm = pl->m;
    T* yp = static_cast<T*>(y);
    T a = alpha, b = beta;
    int UBv = env_int("SPBLAS_GFX950_PB_RBATCH", pl->enc8 ? 2 : 4);
    if (UBv != 2)
      UBv = 4;
    T* const* peers = reinterpret_cast<T* const*>(peers_p);
    void* args[] = {&mm, &Hw, &wb_begin, &wb_end, &binblk, &Pp, &rowp, &srcp, &vals, &rp, &o64, &yp, &a, &b, &cap, &hdr,
                    &exc_idx, &exc_row, &exc_cnt, &exc_cap, &peers, &n_peers, &peer_off};
    const size_t lds = ((size_t) cap + (size_t) pl->vf_waves * (Hw + 64)) * sizeof(T);
    // persistent: one workgroup per CU walks the bins (SPBLAS_GFX950_PB_VF_GRID: test / experiment hook)
    int64_t grid = env_int("SPBLAS_GFX950_PB_VF_GRID", h->num_cus > 0 ? h->num_cus : 256);
    if (grid < 1 || grid > wb_end - wb_begin)
      grid = wb_end - wb_begin;
    SPB_HIP(hipLaunchKernel(pb_reduce_vf_fn<T>(pl->vf_waves, UBv, pl->enc8 != 0), dim3((unsigned) grid),
                            dim3(pl->vf_waves * 64), args, lds, s));
  } else {
    const int32_t* binblk = static_cast<const int32_t*>(pl->s_binblk);
    const T* Pp = static_cast<const T*>(pl->s_products);
    const uint16_t* rowp = pl->enc8 ? reinterpret_cast<const uint16_t*>(pl->s_code) : pl->s_lrow;
    typedef typename pb_hdr<T>::type hdr_t;
    const hdr_t* hdr = static_cast<const hdr_t*>(pl->s_hdr);
    const unsigned* exc_idx = pl->s_exc_idx;
    const uint16_t* exc_row = pl->s_exc_row;
    const int32_t* exc_cnt = pl->s_exc_cnt;
    int exc_cap = pl->exc_cap;
    const bool e8 = pl->enc8 != 0;
    T* yp = static_cast<T*>(y);
    T* part = K > 1 ? static_cast<T*>(pl->s_partial) : nullptr;
    int64_t mm = pl->s_m, pstride = pl->s_m;
    int Hw = pl->rows_per_blk, Kk = K;
    T a = alpha, b = beta;
    T* const* peers = reinterpret_cast<T* const*>(peers_p);
    const int4* ritems = nullptr;
    int dbg = env_int("SPBLAS_GFX950_PB_DBG", 0);
    const int32_t* binrow = static_cast<const int32_t*>(pl->s_binrow);
    void* args[] = {&mm, &Hw, &wb_begin, &wb_end, &binblk, &Pp, &rowp, &yp, &a, &b, &Kk, &part, &pstride,
                    &peers, &n_peers, &peer_off, &ritems, &dbg, &binrow, &rowmap, &piece_out, &hdr, &exc_idx,
                    &exc_row, &exc_cnt, &exc_cap};
    const size_t lds = (size_t) RW * (pl->rows_per_blk + 64) * sizeof(T);
    if (use_items) {
      // row-skewed matrix, whole range: explicit work list (built at inspect), compact partial sums
      ritems = static_cast<const int4*>(pl->s_ritems);
      part = static_cast<T*>(pl->s_rpartial);
      pstride = 0;
      SPB_HIP(hipLaunchKernel(pb_reduce_fn<T>(RW, UB, e8), dim3((unsigned) pl->n_ritems), dim3(RW * 64), args, lds, s));
      if (pl->n_rsplit > 0)
        hipLaunchKernelGGL((pb_combine_items_kernel<T>),
                           dim3((unsigned) pl->n_rsplit, (unsigned) cdiv((int64_t) RW * pl->rows_per_blk, 256)), dim3(256),
                           0, s, static_cast<const int4*>(pl->s_rsplit), (int64_t) RW * pl->rows_per_blk, pl->s_m,
                           static_cast<const T*>(pl->s_rpartial), static_cast<T*>(y), alpha, beta, pl->rows_per_blk,
                           binrow, pl->n_rblk, rowmap, piece_out);
    } else {
      // fused multi-GPU step, throughput form: the wait for the PREVIOUS step's barrier goes right before the kernel that
      // stores into the peers' copies of y -- everything before it overlaps the peers' stores still crossing the links
      auto wait_hook = [&]() -> int {
        auto& bw = h->bcast_wait;
        if (!peers_p || !bw.flags)
          return SPBLAS_GFX950_STATUS_SUCCESS;
        const int rc_w = launch_step_wait(h, bw.flags, bw.n_peers, bw.step, bw.timeout_ms, bw.status_dev);
        bw.flags = nullptr;
        return rc_w;
      };
      if (!(K > 1 && r_hi > r_lo)) {
        const int rc_w = wait_hook();
        if (rc_w)
          return rc_w;
      }
      SPB_HIP(hipLaunchKernel(pb_reduce_fn<T>(RW, UB, e8), dim3((unsigned) groups, (unsigned) K), dim3(RW * 64), args, lds, s));
      if (K > 1 && r_hi > r_lo) {
        const int rc_w = wait_hook();
        if (rc_w)
          return rc_w;
      }
      if (K > 1 && r_hi > r_lo && h->chunk_pub.flag_peers && peers_p && !rowmap) {
        // chunked multi-GPU step: the combine publishes its rows chunk by chunk itself (one-shot)
        const auto cp = h->chunk_pub;
        h->chunk_pub.flag_peers = nullptr;
        hipLaunchKernelGGL((pb_combine_publish_kernel<T>), dim3((unsigned) cdiv(r_hi - r_lo, PB_PUB_ROWS)), dim3(256), 0, s, r_lo, r_hi,
                           K, static_cast<const T*>(pl->s_partial), pl->s_m, alpha, reinterpret_cast<T* const*>(peers_p),
                           n_peers, peer_off, reinterpret_cast<long long* const*>(cp.flag_peers), cp.slot0, cp.chunks,
                           cp.rows_per_chunk, cp.step, h->chunk_done, cp.delay_ticks);
      } else if (K > 1 && r_hi > r_lo)
        hipLaunchKernelGGL((pb_combine_kernel<T>), dim3((unsigned) cdiv(r_hi - r_lo, 256)), dim3(256), 0, s, r_lo, r_hi, K,
                           static_cast<const T*>(pl->s_partial), pl->s_m, static_cast<T*>(y), alpha, beta,
                           reinterpret_cast<T* const*>(peers_p), n_peers, peer_off, rowmap, piece_out);
    }
  }
  if (pl->n_split > 0)  // rows that were cut into pieces: y = (sum of the pieces) + beta * y
    hipLaunchKernelGGL((pb_split_finish_kernel<T>), dim3((unsigned) cdiv(pl->n_split, 4)), dim3(256), 0, s, pl->n_split,
                       static_cast<const int4*>(pl->s_split_rows), static_cast<const T*>(piece_out), static_cast<T*>(y), beta);
  if (rowmap && pl->n_zero > 0 && o_hi > o_lo)  // the empty rows inside the range: y = beta * y
    hipLaunchKernelGGL((pb_empty_rows_kernel<T>), dim3((unsigned) cdiv(pl->n_zero, 256)), dim3(256), 0, s, pl->n_zero,
                       static_cast<const int32_t*>(pl->s_zrow), static_cast<T*>(y), beta, o_lo, o_hi);
  if (pl->hub_len > 0 && pl->n_hub > 0) {
    // rows kept out of the tiles: y[row] += alpha * (row . x), for the rows of this bin range
    if (!pl->values_ptr || !pl->last_x)
      return SPBLAS_GFX950_STATUS_INVALID_VALUE;
    if (!pl->s_hub_part)
      return SPBLAS_GFX950_STATUS_INVALID_VALUE;
    const dim3 grid((unsigned) pl->n_hub, (unsigned) pl->hub_parts);
    T* part = static_cast<T*>(pl->s_hub_part);
    const int32_t* hub_rows = static_cast<const int32_t*>(pl->s_hub_rows);
    // (hub rows are numbered like the rows of the tiles: compact numbers and the compacted row pointers when the
    // empty rows were taken out)
    const void* hub_rowptr = rowmap ? pl->s_rowptr_c : pl->rowptr;
    if (pl->offset_type == SPBLAS_GFX950_I32)
      hipLaunchKernelGGL((pb_hub_rows_kernel<T, int32_t>), grid, dim3(256), 0, s, pl->n_hub, hub_rows,
                         static_cast<const int32_t*>(hub_rowptr), pl->colind, static_cast<const T*>(pl->values_ptr),
                         static_cast<const T*>(pl->last_x), part, r_lo, r_hi);
    else
      hipLaunchKernelGGL((pb_hub_rows_kernel<T, int64_t>), grid, dim3(256), 0, s, pl->n_hub, hub_rows,
                         static_cast<const int64_t*>(hub_rowptr), pl->colind, static_cast<const T*>(pl->values_ptr),
                         static_cast<const T*>(pl->last_x), part, r_lo, r_hi);
    hipLaunchKernelGGL((pb_hub_finish_kernel<T>), dim3((unsigned) cdiv(pl->n_hub, 256)), dim3(256), 0, s, pl->n_hub,
                       pl->hub_parts, hub_rows, part, static_cast<T*>(y), alpha, r_lo, r_hi, rowmap);
  }
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

// split the reduce of the WHOLE row range would use, and the partial-sum workspace for it:
// callers that reduce stripe by stripe on several streams cap K with it (handle->max_ksplit) and
// reserve the workspace before they fork, so no stripe allocates.
int spmv_sliced_full_ksplit(spblas_gfx950_plan_s* pl) {
  return pick_ksplit(pl->n_rblk, pl->n_rblk > 0 ? pl->p_blocks / (PB_GRP / pb_blk_of(pl->value_type)) / pl->n_rblk : 0);
}

int spmv_sliced_reserve_partial(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, int K) {
  if (K <= 1 || pl->s_partial_k >= K)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  hipStream_t s = h->stream;
  if (stream_capturing(s))
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  dev_free(pl->s_partial, s);
  pl->s_partial = nullptr;
  pl->s_partial_k = 0;
  const size_t tsz = pl->value_type == SPBLAS_GFX950_F32 ? 4 : 8;
  int rc = dev_alloc(&pl->s_partial, (size_t) K * pl->s_m * tsz, s);
  if (rc)
    return rc;
  pl->s_partial_k = K;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spmv_sliced_expand(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* x) {
  if (pl->rest_plan) {  // split plan: the tiles are A_rest's; the hot part needs x again at reduce time
    pl->last_x = x;
    pl->rest_plan->nt_products = pl->nt_products;
    return spmv_sliced_expand(h, pl->rest_plan, x);
  }
  return pl->value_type == SPBLAS_GFX950_F32 ? sliced_expand_typed<float>(h, pl, x)
                                             : sliced_expand_typed<double>(h, pl, x);
}

// bins whose first row lies in [row_begin, row_end)
int spmv_sliced_reduce_rows(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* alpha, const void* beta,
                            void* y, int64_t row_begin, int64_t row_end, void* const* peers, int n_peers,
                            int64_t peer_off) {
  if (pl->rest_plan) {
    // split plan: all rows in one call, no peer stores (the hot part adds into y after the tiles have written it)
    if (peers || row_begin > 0 || row_end < pl->m || !pl->last_x)
      return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
    const int rc_r = spmv_sliced_reduce_rows(h, pl->rest_plan, alpha, beta, y, 0, pl->m, nullptr, 0, 0);
    return rc_r ? rc_r : spmv_hot_rows(h, pl, alpha, pl->last_x, y);
  }
  const int64_t H = pl->rows_per_blk;
  int64_t wb0 = cdiv(row_begin, H);
  int64_t wb1 = cdiv(row_end, H);
  if (pl->s_binrow && row_begin <= 0 && row_end >= pl->m) {
    wb0 = 0;
    wb1 = pl->n_rblk;
  } else if (pl->s_binrow) {  // variable bins: first bin whose first row is >= the bound
    const int rc_b = ensure_host_binrow(h, pl);
    if (rc_b)
      return rc_b;
    const int32_t* br = pl->h_binrow_orig ? pl->h_binrow_orig : pl->h_binrow;  // first row of y of every bin
    wb0 = std::lower_bound(br, br + pl->n_rblk, (int32_t) std::min<int64_t>(row_begin, pl->m)) - br;
    wb1 = std::lower_bound(br, br + pl->n_rblk, (int32_t) std::min<int64_t>(row_end, pl->m)) - br;
  }
  if (wb1 > pl->n_rblk)
    wb1 = pl->n_rblk;
  return pl->value_type == SPBLAS_GFX950_F32
             ? sliced_reduce_typed<float>(h, pl, alpha, beta, y, wb0, wb1, peers, n_peers, peer_off)
             : sliced_reduce_typed<double>(h, pl, alpha, beta, y, wb0, wb1, peers, n_peers, peer_off);
}

int spmv_sliced_exec(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* alpha, const void* x,
                     const void* beta, void* y) {
  if (pl->rest_plan)
    return spmv_hot_exec(h, pl, alpha, x, beta, y);
  int rc = spmv_sliced_expand(h, pl, x);
  if (rc)
    return rc;
  return spmv_sliced_reduce_rows(h, pl, alpha, beta, y, 0, pl->m, nullptr, 0, 0);
}

void spmv_sliced_free(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl) {
  hipStream_t s = h->stream;
  spmv_hot_free(h, pl);
  dev_free(pl->seg_ptr, s);
  dev_free(pl->s_sliceblk, s);
  dev_free(pl->s_binblk, s);
  dev_free(pl->s_blkdst, s);
  dev_free(pl->s_blksrc, s);
  dev_free(pl->s_eoff, s);
  pl->s_blksrc = pl->s_eoff = nullptr;
  pl->a_entries = 0;
  dev_free(pl->s_colind, s);
  dev_free(pl->s_values, s);
  dev_free(pl->s_lrow, s);
  dev_free(pl->s_code, s);
  dev_free(pl->s_hdr, s);
  dev_free(pl->s_exc_idx, s);
  dev_free(pl->s_exc_row, s);
  dev_free(pl->s_exc_cnt, s);
  pl->s_code = nullptr;
  pl->s_hdr = nullptr;
  pl->s_exc_idx = nullptr;
  pl->s_exc_row = nullptr;
  pl->s_exc_cnt = nullptr;
  pl->enc8 = 0;
  dev_free(pl->s_perm, s);
  dev_free(pl->s_src, s);
  pl->s_src = nullptr;
  pl->vfree = 0;
  dev_free(pl->s_products, s);
  dev_free(pl->s_partial, s);
  dev_free(pl->s_hub_part, s);
  dev_free(pl->s_binrow, s);
  pl->s_binrow = nullptr;
  std::free(pl->h_binrow);
  pl->h_binrow = nullptr;
  std::free(pl->h_binrow_orig);
  pl->h_binrow_orig = nullptr;
  dev_free(pl->s_rowptr_c, s);
  dev_free(pl->s_nzrow, s);
  dev_free(pl->s_zrow, s);
  dev_free(pl->s_piece_out, s);
  dev_free(pl->s_split_rows, s);
  pl->s_piece_out = pl->s_split_rows = nullptr;
  pl->n_split = 0;
  pl->split_len = 0;
  pl->s_rowptr_c = pl->s_nzrow = pl->s_zrow = nullptr;
  pl->n_zero = 0;
  pl->s_m = 0;
  if (pl->hub_rows_owned)
    dev_free(pl->s_hub_rows, s);
  pl->s_hub_rows = nullptr;
  pl->n_hub = 0;
  pl->hub_rows_owned = false;
  dev_free(pl->s_xitems, s);
  dev_free(pl->s_ritems, s);
  dev_free(pl->s_rsplit, s);
  dev_free(pl->s_rpartial, s);
  pl->s_ritems = pl->s_rsplit = pl->s_rpartial = nullptr;
  pl->n_ritems = pl->n_rsplit = 0;
  pl->s_partial = nullptr;
  pl->s_hub_part = nullptr;
  pl->s_xitems = nullptr;
  pl->n_xitems = 0;
  pl->s_partial_k = 0;
  pl->seg_ptr = pl->s_colind = pl->s_values = pl->s_products = pl->s_perm = nullptr;
  pl->s_sliceblk = pl->s_binblk = pl->s_blkdst = nullptr;
  pl->s_lrow = nullptr;
  pl->a_blocks = pl->p_blocks = 0;
  pl->n_slices = 0;
}

} // namespace spb

// Loads this file's code object (the runtime loads a code object at the first use of one of its kernels: milliseconds
// that would otherwise fall on the caller's first inspect / compute call -- handle.hip: spblas_gfx950_create).
namespace spb {
void preload_sliced() {
  hipFuncAttributes attr;
  (void) hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&pb_bin_orig_rows_kernel));
  (void) hipGetLastError();
}
} // namespace spb
