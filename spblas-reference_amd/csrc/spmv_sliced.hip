// Column-sliced ("propagation blocking") SpMV for matrices whose x gathers miss every
// cache: SPBLAS_GFX950_SPMV_SLICED.
//
// Why: on BASELINE cfg2 (10M x 10M, uniform random columns) the CSR kernels gather each
// x[col] as a separate 128-byte L2 miss (rocprofv3: TCC_EA0_RDREQ_128B = 0.97 per nonzero,
// 12.4 GB of fabric reads for 0.92 GB of algorithmic bytes -- profiles/r01a_summary.md).
// The only memory on the CU that sustains random 4-byte accesses at the needed rate is LDS,
// so multiply_inspect re-tiles A once on the device:
//
//   A' order   entries grouped by (column slice s, wave-bin wb); slice = W consecutive columns
//              (W*sizeof(T) <= 80 KiB of LDS), wave-bin = Hw consecutive rows owned by ONE
//              wavefront of the reduce kernel (8 wave-bins * Hw * sizeof(T) <= 80 KiB)
//   s_val[i]   value,  s_col[i] 16-bit column inside the slice,
//   s_row[i]   15-bit row inside the wave-bin | bit 15 = "duplicate" flag (see below)
//
// and multiply() runs two streaming kernels (no global gathers, no global atomics):
//   expand  one workgroup per slice: x slice -> LDS, then P[i] = s_val[i] * xs[s_col[i]] over the
//           slice's contiguous range of A' (16-byte lane accesses)
//   reduce  one wavefront per wave-bin: Hw accumulators in LDS, walk the bin's S runs of P;
//           y = alpha*acc + beta*y
// HBM traffic per nonzero: 6 B + 4 B (expand) + 6 B (reduce) = 16 B vs 8 B algorithmic, all of it
// coalesced streams.
//
// Why wave-owned bins: ds_add_f32 retires ~0.33 lanes/clk/CU on gfx950 (tools/ubench/lds_atomic:
// 1e8 LDS float atomics = 509 us, 12x slower than integer atomics or a plain read-add-write),
// so accumulation must be a plain LDS read-modify-write.  That is race free iff (a) no other
// wavefront touches the rows -- each wave owns its bin -- and (b) the <= 64 entries one wave
// instruction handles hit distinct rows.  (b) is arranged at inspect time: within each
// 64-entry chunk of a run all but one entry of a repeated row carry the duplicate flag and are
// applied afterwards with the (slow, rare: ~1 % of entries) LDS atomic.
// The order of additions into a row is fixed by the plan, so results are run-to-run
// reproducible for a given plan (plans built twice may order entries differently).
#include <algorithm>
#include <vector>

#include "common.hpp"
#include "plan.hpp"
#include "scan.hpp"

#include <cstdlib>
#include <ctime>

namespace spb {

static int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return v && *v ? std::atoi(v) : dflt;
}

static constexpr int PB_THREADS = 1024;         // expand: 16 waves share one x slice
static constexpr int PB_RTHREADS = 512;         // inspect (flag kernel): 8 waves, one wave-bin each
static constexpr int PB_RWAVES = PB_RTHREADS / 64;
static constexpr int PB_RWAVES_DEFAULT = 4;     // reduce: wave-bins per workgroup (plan->rwaves)
static constexpr int PB_LDS_BYTES = 80 * 1024;  // two workgroups per CU (160 KiB LDS)

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

// hub_len > 0: entries of rows longer than hub_len are left out of the tiles (a run that repeats one
// row hundreds of times would serialise on the LDS atomic); pb_hub_rows_kernel adds those rows.
template <typename O>
__global__ __launch_bounds__(256) void pb_count_kernel(int64_t m, const O* __restrict__ rowptr,
                                                       const int32_t* __restrict__ colind, int W, int H, int S, int NB,
                                                       int32_t* __restrict__ cnt, int hub_len) {
  extern __shared__ int hist[];  // [S]
  const int wb = blockIdx.x;
  for (int i = threadIdx.x; i < S; i += 256)
    hist[i] = 0;
  __syncthreads();
  const int64_t r0 = (int64_t) wb * H, r1 = (r0 + H) < m ? (r0 + H) : m;
  if (r0 < m) {
    const O p0 = rowptr[r0], p1 = rowptr[r1];
    for (O p = p0 + threadIdx.x; p < p1; p += 256) {
      if (hub_len > 0) {
        const int64_t r = pb_row_of(rowptr, r0, r1, p);
        if (rowptr[r + 1] - rowptr[r] > (O) hub_len)
          continue;
      }
      atomicAdd(&hist[colind[p] / W], 1);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < S; i += 256)
    cnt[(int64_t) i * NB + wb] = hist[i];
}

// Same ownership for the scatter: LDS cursors start at the runs' offsets; the row of entry p is
// found by a binary search in the bin's slice of rowptr (<= 12 probes, L1/L2 resident).
template <typename T, typename O>
__global__ __launch_bounds__(256) void pb_scatter_kernel(int64_t m, const O* __restrict__ rowptr,
                                                         const int32_t* __restrict__ colind,
                                                         const T* __restrict__ values, int W, int H, int S, int NB,
                                                         const int32_t* __restrict__ seg, T* __restrict__ s_val,
                                                         uint16_t* __restrict__ s_col, uint16_t* __restrict__ s_row,
                                                         int32_t* __restrict__ perm, int hub_len) {
  extern __shared__ int cursor[];  // [S]
  const int wb = blockIdx.x;
  for (int i = threadIdx.x; i < S; i += 256)
    cursor[i] = seg[(int64_t) i * NB + wb];
  __syncthreads();
  const int64_t r0 = (int64_t) wb * H, r1 = (r0 + H) < m ? (r0 + H) : m;
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
    s_row[i] = (uint16_t) (lo - r0);
    perm[i] = (int32_t) p;
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

template <typename T, typename O>
__global__ __launch_bounds__(PB_STAGE_THREADS) void pb_scatter_staged_kernel(
    int64_t m, const O* __restrict__ rowptr, const int32_t* __restrict__ colind, const T* __restrict__ values, int W,
    int H, int S, int NB, const int32_t* __restrict__ seg, T* __restrict__ s_val, uint16_t* __restrict__ s_col,
    uint16_t* __restrict__ s_row, int32_t* __restrict__ perm, int hub_len, int cap, int rt_len) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* lcnt = reinterpret_cast<int*>(smem);  // [S] entries of this bin per slice
  int* gdst = lcnt + S;                      // [S] start of the run in A' order
  int* lcur = gdst + S;                      // [S] staging cursor (local offset, advanced by the atomics)
  int* rp = lcur + S;                        // [H + 1] the bin's row offsets relative to its first entry
  int* rt = rp + H + 1;                      // [rt_len] row of every 64th entry (narrows the row search)
  int* st = rt + rt_len;                     // [cap] staged entries: position relative to the first entry,
  T* stv = reinterpret_cast<T*>(st + cap);   // [cap] value,
  uint16_t* stc = reinterpret_cast<uint16_t*>(stv + cap);  // [cap] column inside the slice
  __shared__ int pass_end, pass_direct;
  const int wb = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t r0 = (int64_t) wb * H, r1 = (r0 + H) < m ? (r0 + H) : m;
  if (r0 >= m)
    return;
  const int nr = (int) (r1 - r0);
  const O p0 = rowptr[r0], p1 = rowptr[r1];
  const int ne = (int) (p1 - p0);
  if (ne == 0)
    return;  // nothing to place (and p0 may be the end of the arrays: the clamped gathers below need ne > 0)
  for (int i = tid; i < S; i += PB_STAGE_THREADS) {
    const int a = seg[(int64_t) i * NB + wb], b = seg[(int64_t) i * NB + wb + 1];
    gdst[i] = a;
    lcnt[i] = b - a;
  }
  for (int i = tid; i <= nr; i += PB_STAGE_THREADS)
    rp[i] = (int) (rowptr[r0 + i] - p0);
  __syncthreads();
  // row (inside the bin) of the entry at relative position q: last i in [lo, hi) with rp[i] <= q
  auto row_between = [&](int q, int lo, int hi) {
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (rp[mid] <= q)
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
      rt[k] = k < nblk ? row_between(k << 6, 0, nr) : (nr > 0 ? nr - 1 : 0);
    __syncthreads();
  }
  auto row_of = [&](int q) {
    if (!use_rt)
      return row_between(q, 0, nr);
    const int k = q >> 6;
    return row_between(q, rt[k], rt[k + 1] + 1);
  };
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
        if (hub_len > 0 && rp[r + 1] - rp[r] > hub_len)
          continue;
      }
      const int pos = atomicAdd(&lcur[sl], 1);
      if (direct) {
        const int i = gdst[sl] + pos;
        s_val[i] = values[p0 + q];
        s_col[i] = (uint16_t) (c - sl * W);
        s_row[i] = (uint16_t) r;
        perm[i] = (int32_t) (p0 + q);
      } else {
        st[pos] = q;
        stv[pos] = values[p0 + q];  // neighbouring threads: neighbouring addresses
        stc[pos] = (uint16_t) (c - sl * W);
      }
      }
    }
    __syncthreads();
    if (!direct) {
      // every wave writes whole runs from the staging area (contiguous stores, no global gathers: fetching
      // value and column again by position cost 2.4 of the kernel's 2.75 ms -- random 4-byte reads, even
      // L2 hits, run at ~10 cycles per request and CU)
      constexpr int NW = PB_STAGE_THREADS / 64;
      for (int sl = s0 + wave; sl < s1; sl += NW) {
        const int n = lcnt[sl], lo = lcur[sl] - n, g = gdst[sl];
        for (int j = lane; j < n; j += 64) {
          const int q = st[lo + j];
          s_val[g + j] = stv[lo + j];
          s_col[g + j] = stc[lo + j];
          s_row[g + j] = (uint16_t) row_of(q);
          perm[g + j] = (int32_t) (p0 + q);
        }
      }
    }
    // end of the pass: the next one reuses the LDS staging area, so LDS traffic must be complete -- but not
    // the global stores above; __syncthreads() would also wait for those (vmcnt(0)) in every wave
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    s0 = s1;
  }
}

// balance probe: entries per slice and per bin group (RW bins = one reduce workgroup).  Workgroups
// [0, S) sum one slice each (a contiguous row of NB counters), workgroups [S, S + ngroups) one bin group
// each (RW counters out of every slice's row); no atomics (the first version added every counter to its
// group with a global atomic: 0.59 ms at cfg2).
__global__ __launch_bounds__(256) void pb_balance_kernel(int S, int NB, int RW, const int32_t* __restrict__ cnt,
                                                         unsigned long long* __restrict__ slice_sum,
                                                         unsigned long long* __restrict__ slice_ne,
                                                         unsigned long long* __restrict__ group_sum) {
  __shared__ unsigned long long red[4];
  __shared__ unsigned long long red_ne[4];
  unsigned long long tot = 0, ne = 0;  // ne: non-empty (slice, bin) tiles -- the locality probe
  if ((int) blockIdx.x < S) {
    const int sl = blockIdx.x;
    for (int b = threadIdx.x; b < NB; b += 256) {
      const unsigned long long c = (unsigned long long) cnt[(int64_t) sl * NB + b];
      tot += c;
      ne += c != 0;
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
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = tot;
    red_ne[threadIdx.x >> 6] = ne;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = red[0] + red[1] + red[2] + red[3];
    if ((int) blockIdx.x < S) {
      slice_sum[blockIdx.x] = t;
      slice_ne[blockIdx.x] = red_ne[0] + red_ne[1] + red_ne[2] + red_ne[3];
    } else {
      group_sum[blockIdx.x - S] = t;
    }
  }
}

// entries of bin group g (RW bins) in slice s, from the exclusive offsets: out[g*S + s]
__global__ __launch_bounds__(256) void pb_group_slice_kernel(int S, int NB, int RW, int64_t ngroups,
                                                             const int32_t* __restrict__ seg,
                                                             int32_t* __restrict__ out) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= ngroups * S)
    return;
  const int64_t g = i / S;
  const int sl = (int) (i % S);
  const int64_t b0 = g * RW, b1 = (g + 1) * RW < NB ? (g + 1) * RW : NB;
  out[i] = seg[(int64_t) sl * NB + b1] - seg[(int64_t) sl * NB + b0];
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
struct pack4;
template <>
struct pack4<float> {
  static __device__ __forceinline__ void load(const float* p, float (&o)[4]) {
    const f32x4 v = stream_load(reinterpret_cast<const f32x4*>(p));
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&o)[4]) {
    f32x4 v;
    v.x = o[0]; v.y = o[1]; v.z = o[2]; v.w = o[3];
    *reinterpret_cast<f32x4*>(p) = v;  // plain store: 12 % faster than nt here (measured)
  }
};
template <>
struct pack4<double> {
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
};

typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

// expand: P[i] = s_val[i] * x[slice_base + s_col[i]] over the slice's contiguous range of A'.
// The x slice lives in LDS; A' and P are touched exactly once with 16-byte lane accesses.
template <typename T>
__global__ __launch_bounds__(PB_THREADS) void pb_expand_kernel(int64_t n, int W, int NB, const int32_t* __restrict__ seg,
                                                               const T* __restrict__ s_val,
                                                               const uint16_t* __restrict__ s_col,
                                                               const T* __restrict__ x, T* __restrict__ P,
                                                               const int4* __restrict__ items, int S, int share) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* xs = reinterpret_cast<T*>(smem);
  const int tid = threadIdx.x;
  auto load_x = [&](int s) {
    const int64_t c0 = (int64_t) s * W;
    const int cw = (int) ((n - c0) < W ? (n - c0) : W);
    for (int i = tid; i < cw; i += PB_THREADS)
      xs[i] = x[c0 + i];
  };
  // entries [a0, a1) of the slice whose x values are in LDS
  auto process = [&](int a0, int a1) {
    int body0 = (a0 + 3) & ~3;
    if (body0 > a1)
      body0 = a1;
    const int body1 = body0 + ((a1 - body0) & ~3);
    // unaligned head and tail (< 4 entries each)
    if (tid < body0 - a0)
      P[a0 + tid] = s_val[a0 + tid] * xs[s_col[a0 + tid]];
    if (tid < a1 - body1)
      P[body1 + tid] = s_val[body1 + tid] * xs[s_col[body1 + tid]];
    // aligned body: 4 entries (16 B of values, 8 B of columns) per lane per step, 2 steps in flight
    int i = body0 + 4 * tid;
    for (; i + 4 * PB_THREADS < body1; i += 8 * PB_THREADS) {
      T va[4], vb[4], pa[4], pb[4];
      pack4<T>::load(s_val + i, va);
      pack4<T>::load(s_val + i + 4 * PB_THREADS, vb);
      const u16x4 ca = stream_load(reinterpret_cast<const u16x4*>(s_col + i));
      const u16x4 cb = stream_load(reinterpret_cast<const u16x4*>(s_col + i + 4 * PB_THREADS));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pa[j] = va[j] * xs[ca[j]];
        pb[j] = vb[j] * xs[cb[j]];
      }
      pack4<T>::store(P + i, pa);
      pack4<T>::store(P + i + 4 * PB_THREADS, pb);
    }
    for (; i < body1; i += 4 * PB_THREADS) {
      T va[4], pa[4];
      pack4<T>::load(s_val + i, va);
      const u16x4 ca = stream_load(reinterpret_cast<const u16x4*>(s_col + i));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        pa[j] = va[j] * xs[ca[j]];
      pack4<T>::store(P + i, pa);
    }
  };
  if (items) {
    // items (column-skewed matrices): workgroup i takes entries [items[i].y, items[i].z) of slice items[i].x,
    // so that a slice holding a large share of the matrix is spread over proportionally many workgroups
    const int4 item = items[blockIdx.x];
    load_x(item.x);
    __syncthreads();
    process(item.y, item.z);
    return;
  }
  // Equal shares: workgroup b takes the entries [b*share, (b+1)*share) of A' (share is a multiple of 4) and
  // loads the x slice of every slice its range touches -- one or two for the usual case of about one slice
  // per workgroup.  The grid is exactly one wave of workgroups whatever the slice count is (slices cut for
  // LDS capacity rarely come in multiples of 512; a second, nearly empty round of whole-slice workgroups cost
  // up to 2x).
  const int total = seg[(int64_t) S * NB];
  const long long g0l = (long long) blockIdx.x * share;
  int g0 = g0l < total ? (int) g0l : total;
  const int g1 = (total - g0) < share ? total : g0 + share;
  if (g0 >= g1)
    return;
  int lo = 0, hi = S;  // last slice starting at or before g0
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seg[(int64_t) mid * NB] <= g0)
      lo = mid;
    else
      hi = mid;
  }
  for (int s = lo; g0 < g1 && s < S; ++s) {
    const int slice_end = seg[(int64_t) (s + 1) * NB];
    if (slice_end <= g0)
      continue;  // empty slice
    const int a1 = g1 < slice_end ? g1 : slice_end;
    __syncthreads();  // everyone is done with the previous x slice
    load_x(s);
    __syncthreads();
    process(g0, a1);
    g0 = a1;
  }
}

// inspect: mark duplicates.  Same walk as the reduce kernel: wave-bin -> groups of GR runs -> chunks.
// The reduce kernel issues the LDS reads of a whole group (GR runs x C 64-entry chunks) before the
// first write, so every entry whose row already occurs earlier in its group gets the duplicate
// flag (bit 15) and takes the atomic path there.  Chunks beyond the C-th of a run are applied one
// at a time and only need flags inside the chunk.
//   tag[row]:   lane id, resolves repeats inside one chunk (exactly one lane reads back its own id)
//   stamp[row]: id of the last group that claimed the row; ids wrap after 255 groups, and a stale
//               match merely sends a unique entry down the (still correct) atomic path.
__global__ __launch_bounds__(PB_RTHREADS) void pb_flag_dups_kernel(int Hw, int S, int64_t NBw, int C, int GR,
                                                                   const int2* __restrict__ segT,
                                                                   uint16_t* __restrict__ s_row) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned char* tag = smem + (size_t) wave * 2 * Hw;
  unsigned char* stamp = tag + Hw;
  const int64_t wb = (int64_t) blockIdx.x * PB_RWAVES + wave;
  if (wb >= NBw)
    return;
  for (int i = lane; i < Hw; i += 64)
    stamp[i] = 0;
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  const int2* mine = segT + wb * S;
  // one 64-entry chunk of a run: an entry whose row was already seen in this group (stamp) or is claimed by
  // another lane of the chunk (tag) gets the flag
  auto chunk = [&](int start, int ln, int base, int rw, bool have, unsigned char cur) {
    const bool grouped = base < 64 * C;
    const int o = base + lane;
    const bool ok = o < ln;
    int row = 0;
    bool seen = false;
    if (ok) {
      row = (have ? rw : (int) s_row[start + o]) & 0x7FFF;
      seen = grouped && stamp[row] == cur;
      if (!seen)
        tag[row] = (unsigned char) lane;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the tag stores have landed
    if (ok) {
      const bool won = !seen && tag[row] == (unsigned char) lane;
      if (!won)
        s_row[start + o] = (uint16_t) (row | 0x8000);
      else if (grouped)
        stamp[row] = cur;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
  };
  // Runs are taken in batches of 8 (descriptors 64 at a time, one per lane): the first two chunks of all
  // eight runs are loaded before the first is processed, so the wave waits for memory once per batch and
  // not twice per chunk (0.77 -> 0.3 ms at cfg2).  Loads past the end of a run are clamped to its last
  // entry (the arrays carry 64 entries of slack for empty trailing runs).
  constexpr int B = 8, PC = 2;
  for (int g0 = 0; g0 < S; g0 += 64) {
    const int t = g0 + lane;
    int2 d = mine[t < S ? t : S - 1];
    if (t >= S)
      d.y = 0;
    for (int b0 = 0; b0 < 64 && g0 + b0 < S; b0 += B) {
      int st[B], ln[B], rw[B][PC];
#pragma unroll
      for (int u = 0; u < B; ++u) {
        st[u] = __builtin_amdgcn_readlane(d.x, b0 + u);
        ln[u] = __builtin_amdgcn_readlane(d.y, b0 + u);
        const int last = ln[u] > 0 ? ln[u] - 1 : 0;
#pragma unroll
        for (int c = 0; c < PC; ++c) {
          const int o = lane + 64 * c;
          rw[u][c] = s_row[st[u] + (o < last ? o : last)];
        }
      }
#pragma unroll
      for (int u = 0; u < B; ++u) {
        const int sidx = g0 + b0 + u;  // slice of this run; runs past S have ln = 0
        const unsigned char cur = (unsigned char) ((sidx / GR) % 255 + 1);
#pragma unroll
        for (int c = 0; c < PC; ++c)
          if (64 * c < ln[u])
            chunk(st[u], ln[u], 64 * c, rw[u][c], true, cur);
        for (int base = 64 * PC; base < ln[u]; base += 64)
          chunk(st[u], ln[u], base, 0, false, cur);
      }
    }
  }
}

// reduce: one wavefront per wave-bin (8 per workgroup).  The wave fetches 64 run descriptors
// with one coalesced load, then handles runs four at a time: all product/row loads of the
// four runs are issued before the first LDS read-modify-write consumes them.
template <typename T>
__device__ __forceinline__ void pb_apply(T* acc, T p, int r, bool ok) {
  const int row = r & 0x7FFF;
  const bool dup = ok && (r & 0x8000);
  if (ok && !dup)
    acc[row] += p;  // plain LDS read-add-write: rows are distinct within the instruction
  if (__builtin_amdgcn_ballot_w64(dup) != 0) {
    if (dup)
      unsafeAtomicAdd(acc + row, p);
  }
}

// RW wave-bins per workgroup; C 64-entry chunks of every run held in registers.  The loop is
// software-pipelined by hand: the loads of batch k+1 (B runs) are
// issued before batch k is applied.  Every load is unconditional -- lanes past the end of a run
// re-read its last entry (same cache line, no extra traffic) -- because a load inside a divergent
// branch makes the compiler drain vmcnt(0) at the join, which would serialise the pipeline.
template <typename T, int RW, int C, int GR>
__global__ __launch_bounds__(RW * 64) void pb_reduce_kernel(int64_t m, int Hw, int S, int64_t wb_begin, int64_t NBw,
                                                            const int2* __restrict__ segT, const T* __restrict__ P,
                                                            const uint16_t* __restrict__ s_row, T* __restrict__ y,
                                                            T alpha, T beta, int s_per, T* __restrict__ partial,
                                                            int64_t pstride, T* const* __restrict__ peers,
                                                            int n_peers, int64_t peer_off,
                                                            const int4* __restrict__ ritems) {
  // blockIdx.y = k selects the slices [k*s_per, (k+1)*s_per): with few wave-bins (a row shard of
  // a multi-GPU run) the slices are split over several workgroups per bin group, each writing a
  // partial sum that pb_combine_kernel adds up in a fixed order.
  // ritems (row-skewed matrices): workgroup i reduces the slices [ritems[i].y, ritems[i].z) of bin group
  // ritems[i].x, so that a heavy group is spread over as many workgroups as its share of the entries
  // asks for; .w >= 0 is the offset of its partial sums (RW*Hw values, pb_combine_items_kernel adds
  // them up), .w < 0 means the group is not split and y is written directly.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  T* acc = reinterpret_cast<T*>(smem) + (size_t) wave * Hw;
  const int4 item = ritems ? ritems[blockIdx.x] : make_int4((int) blockIdx.x, 0, 0, 0);
  const int64_t wb = wb_begin + (int64_t) item.x * RW + wave;  // NBw = end of the bin range
  const int s_lo = ritems ? item.y : (int) blockIdx.y * s_per;
  const int s_hi = ritems ? item.z : ((s_lo + s_per) < S ? (s_lo + s_per) : S);
  if (ritems)
    partial = item.w >= 0 ? partial + item.w + (int64_t) wave * Hw - wb * (int64_t) Hw : nullptr;
  if (wb >= NBw)
    return;
  const int64_t r0 = wb * Hw;
  const int rh = (int) ((m - r0) < Hw ? (m - r0) : Hw);
  for (int i = lane; i < rh; i += 64)
    acc[i] = T(0);
  const int2* mine = segT + wb * S + s_lo;
  const int ns = s_hi - s_lo;
  constexpr int B = 8;         // runs per batch
  constexpr int GB = 64 / B;   // 64 descriptors (one per lane) make a group of GB batches
  struct batch_t {
    T p[B][C];
    int r[B][C];
    int st[B], ln[B];
  };
  // The wave issues one instruction at a time and only two waves share a SIMD (LDS bounds the occupancy),
  // so the instruction count per 64-entry chunk is what the kernel time follows (SQ counters in DESIGN 4.3):
  // 32-bit byte offsets from scalar bases for the loads, no exec-masked blocks around the stores -- lanes
  // without a plain entry store to a per-lane dummy slot behind the accumulators instead.
  T* const dummy = reinterpret_cast<T*>(smem) + (size_t) RW * Hw + wave * 64 + lane;
  constexpr int SH = sizeof(T) == 4 ? 2 : 3;
  auto fetch_group = [&](int g) -> int2 {  // descriptors of slices 64g .. 64g+63, one per lane
    const int t = 64 * g + lane;
    const int tc = t < ns ? t : (ns > 0 ? ns - 1 : 0);
    int2 d = mine[tc];
    if (t >= ns)
      d.y = 0;
    return d;
  };
  auto issue = [&](const int2& d, int j0, batch_t& q) {
#pragma unroll
    for (int u = 0; u < B; ++u) {
      q.st[u] = __builtin_amdgcn_readlane(d.x, j0 + u);
      q.ln[u] = __builtin_amdgcn_readlane(d.y, j0 + u);
      const int last = q.ln[u] > 0 ? q.ln[u] - 1 : 0;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int o = lane + 64 * c;
        // lanes past the end of the run re-read its last entry; byte offsets fit 32 bits (checked at build)
        const unsigned offP = (unsigned) (q.st[u] + (o < last ? o : last)) << SH;
        const unsigned offR = offP >> (SH - 1);
        // products: read-once stream (nt: L1 bypass); row words: plain loads -- the 128-byte row lines are
        // shared with the neighbouring chunk and hit in L1 on the second touch (reduce 147.5 -> 145 us;
        // plain product loads instead: 173 us; plain column loads in the expand: +5 us there, +9 us here)
        q.p[u][c] = stream_load(reinterpret_cast<const T*>(reinterpret_cast<const char*>(P) + offP));
        if constexpr (sizeof(T) == 4)
          q.r[u][c] = *reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(s_row) + offR);
        else  // fp64: the plain row loads measured 2 % slower (667 vs 654 us at 10M^2)
          q.r[u][c] = stream_load(reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(s_row) + offR));
      }
    }
  };
  auto consume = [&](const batch_t& q) {
#pragma unroll
    for (int g = 0; g < B; g += GR) {
      // one group: all LDS reads, then all writes.  Rows of unflagged entries are distinct inside a
      // group (pb_flag_dups_kernel), flagged ones are added atomically afterwards.
      T v[GR][C];
      T* slot[GR][C];
      bool plain[GR][C], dup[GR][C];
      unsigned long long dups = 0;
#pragma unroll
      for (int u = 0; u < GR; ++u)
#pragma unroll
        for (int c = 0; c < C; ++c) {
          int r = q.r[g + u][c];
          asm("" : "+v"(r));  // keep the row word a 32-bit value (the 16-bit forms cost extra masking)
          const bool ok = lane + 64 * c < q.ln[g + u];
          const bool flagged = (unsigned) r > 0x7FFFu;
          slot[u][c] = acc + (r & 0x7FFF);
          v[u][c] = *slot[u][c];  // clamped lanes hold a real entry too: the row is always in range
          plain[u][c] = ok && !flagged;
          dup[u][c] = ok && flagged;
          dups |= __builtin_amdgcn_ballot_w64(dup[u][c]);
        }
#pragma unroll
      for (int u = 0; u < GR; ++u)
#pragma unroll
        for (int c = 0; c < C; ++c) {
          T* w = plain[u][c] ? slot[u][c] : dummy;
          *w = v[u][c] + q.p[g + u][c];
        }
      if (dups != 0) {
#pragma unroll
        for (int u = 0; u < GR; ++u)
#pragma unroll
          for (int c = 0; c < C; ++c)
            if (dup[u][c])
              unsafeAtomicAdd(slot[u][c], q.p[g + u][c]);
      }
      // longer runs: the tail straight from memory.  Keep these loads inside the `if`: with
      // unconditional loads here the compiler loses track of the in-flight batch and waits
      // vmcnt(0) before every chunk of the main path.
#pragma unroll
      for (int u = 0; u < GR; ++u)
        for (int base = 64 * C; base < q.ln[g + u]; base += 64) {
          const int o = base + lane;
          const bool ok = o < q.ln[g + u];
          T pp = T(0);
          int rr = 0;
          if (ok) {
            pp = stream_load(P + q.st[g + u] + o);
            rr = stream_load(s_row + q.st[g + u] + o);
          }
          pb_apply<T>(acc, pp, rr, ok);
        }
    }
  };
  if (ns > 0) {
    const int nbatch = (ns + B - 1) / B;
    int2 dcur = fetch_group(0), dnext = fetch_group(1);
    // descriptors for batch kk (called with kk = 1, 2, 3, ... in order)
    auto advance = [&](int kk) {
      if ((kk % GB) == 0) {
        dcur = dnext;
        dnext = fetch_group(kk / GB + 1);
      }
    };
    batch_t qa, qb;
    issue(dcur, 0, qa);
    for (int k = 0; k < nbatch; k += 2) {
      advance(k + 1);
      issue(dcur, ((k + 1) % GB) * B, qb);  // past the last batch the descriptors have ln = 0
      consume(qa);
      advance(k + 2);
      issue(dcur, ((k + 2) % GB) * B, qa);
      consume(qb);
    }
  }
  if (partial) {
    // uniform split: slot k of the [K][m] array; work items: the item's own RW*Hw block (pointer pre-biased
    // above so that "+ r0" lands on this wave's part of it)
    T* dst = partial + (ritems ? (int64_t) 0 : (int64_t) blockIdx.y * pstride) + r0;
    for (int i = lane; i < rh; i += 64)
      dst[i] = acc[i];
    return;
  }
  if (peers) {
    // fused all-gather (multi-GPU row shards): local row r is row peer_off + r of the full y, and is
    // stored straight into every rank's copy (peers[] holds the local buffer and the IPC-mapped
    // buffers of the other ranks; stores to those travel over xGMI).  beta = 0 by contract.
    for (int p = 0; p < n_peers; ++p) {
      T* dst = peers[p] + peer_off + r0;
      for (int i = lane; i < rh; i += 64)
        dst[i] = alpha * acc[i];
    }
    return;
  }
  for (int i = lane; i < rh; i += 64) {
    const T v = alpha * acc[i];
    y[r0 + i] = beta == T(0) ? v : v + beta * y[r0 + i];
  }
}

// y = alpha * (partial[0] + partial[1] + ... in this fixed order) + beta * y
template <typename T>
__global__ __launch_bounds__(256) void pb_combine_kernel(int64_t r_lo, int64_t r_hi, int K,
                                                         const T* __restrict__ partial, int64_t pstride,
                                                         T* __restrict__ y, T alpha, T beta,
                                                         T* const* __restrict__ peers, int n_peers,
                                                         int64_t peer_off) {
  const int64_t i = r_lo + (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= r_hi)
    return;
  T s = partial[i];
  for (int k = 1; k < K; ++k)
    s += partial[(int64_t) k * pstride + i];
  if (peers) {
    for (int p = 0; p < n_peers; ++p)
      peers[p][peer_off + i] = alpha * s;
    return;
  }
  y[i] = beta == T(0) ? alpha * s : alpha * s + beta * y[i];
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
                                                            int64_t row_begin, int64_t row_end) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= n_hub)
    return;
  const int64_t r = hub_rows[i];
  if (r < row_begin || r >= row_end)
    return;
  T s = T(0);
  for (int k = 0; k < parts; ++k)
    s += part[i * parts + k];
  y[r] += alpha * s;
}

// Work-item variant of the combine: one entry of `cg` per SPLIT bin group = (group, K_g, offset of its
// first partial block, rows in the group); blockIdx.y walks the group's rows in chunks of 256.  The K_g
// partial blocks of RW*Hw values lie back to back and are added in that order.
template <typename T>
__global__ __launch_bounds__(256) void pb_combine_items_kernel(const int4* __restrict__ cg, int64_t group_rows,
                                                               int64_t m, const T* __restrict__ partial,
                                                               T* __restrict__ y, T alpha, T beta) {
  const int4 g = cg[blockIdx.x];
  const int64_t i = (int64_t) blockIdx.y * 256 + threadIdx.x;
  const int64_t row = (int64_t) g.x * group_rows + i;
  if (i >= group_rows || row >= m)
    return;
  const T* src = partial + (int64_t) g.z + i;
  T s = src[0];
  for (int k = 1; k < g.y; ++k)
    s += src[(int64_t) k * group_rows];
  y[row] = beta == T(0) ? alpha * s : alpha * s + beta * y[row];
}

template <typename T>
static const void* pb_reduce_fn(int rw, int c, int gr) {
#define SPB_RK(RW_, C_, G_) reinterpret_cast<const void*>(pb_reduce_kernel<T, RW_, C_, G_>)
#define SPB_RKG(RW_, C_) (gr == 1 ? SPB_RK(RW_, C_, 1) : (gr == 2 ? SPB_RK(RW_, C_, 2) : SPB_RK(RW_, C_, 4)))
#define SPB_RKC(RW_) (c == 1 ? SPB_RKG(RW_, 1) : (c == 2 ? SPB_RKG(RW_, 2) : SPB_RKG(RW_, 4)))
  return rw == 4 ? SPB_RKC(4) : SPB_RKC(8);
#undef SPB_RKC
#undef SPB_RKG
#undef SPB_RK
}

// ---- host -------------------------------------------------------------------------------
static int pick_ksplit(int64_t groups, int S);

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
  const int64_t m = pl->m, n = pl->n, nnz = pl->nnz;
  // the reduce addresses the product stream with 32-bit byte offsets
  if (nnz > INT32_MAX - 8 || (uint64_t) (nnz + 64) * sizeof(T) >= ((uint64_t) 1 << 32))
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  // x slice of the expand: 80 KiB for fp32 (two workgroups per CU); for fp64 the whole 160 KiB of a CU
  // (one workgroup), which halves the number of slices and doubles the run length again
  // (10M^2, 10/row: 755 -> 651 us; for fp32 the wider slice changed nothing)
  // fp32 switches to 160 KiB slices as well from n = 8 M on: twice the run length for the reduce (with the
  // equal-share expand: n = 6 M -5 %, 8 M +1 %, 10 M +2..5 %, 11 M +26 %, 40 M +17 %)
  // ... provided a slice still carries enough entries to pay for its x load (row shards of a multi-GPU run
  // have the columns of the whole matrix but a fraction of its entries: 2.5 M x 10 M rows ran 15 % slower)
  const int64_t s80 = cdiv(n, PB_LDS_BYTES / 4);
  const bool wide32 = sizeof(T) == 4 && s80 >= 390 && nnz / s80 >= 75000;
  const int xlds = env_int("SPBLAS_GFX950_PB_XLDS_KB", (sizeof(T) == 8 || wide32) ? 160 : PB_LDS_BYTES / 1024) * 1024;
  int max_cols = xlds / (int) sizeof(T);
  if (max_cols > 65536)
    max_cols = 65536;  // 16-bit local column
  // reduce shape: RW wave-bins per workgroup share the 80 KiB.  Fewer, taller bins make longer runs
  // (less cache-line over-fetch at run boundaries) but leave fewer wavefronts to hide latency.
  int RW = env_int("SPBLAS_GFX950_PB_RWAVES", PB_RWAVES_DEFAULT);
  if (RW != 4 && RW != 8)
    RW = PB_RWAVES_DEFAULT;
  pl->rwaves = RW;
  // per wave-bin; < 32768 (15-bit row + flag); 64 dummy slots per wave follow the accumulators
  int max_rows = PB_LDS_BYTES / RW / (int) sizeof(T) - 64;
  if (max_rows > 32767)
    max_rows = 32767;
  int S, W, NB, H;
  const int w_env = env_int("SPBLAS_GFX950_SLICE_COLS", 0);  // test hooks: force small tiles
  const int h_env = env_int("SPBLAS_GFX950_SLICE_ROWS", 0);
  // Slices: as few (as wide) as LDS allows -- every extra slice shortens all runs of the reduce (293 slices
  // at n = 6 M rounded up to 512 cost 10 %).  The expand gives every workgroup an equal share of A' whatever
  // the slice count is, so the count needs no rounding to waves of workgroups (SPBLAS_GFX950_PB_XROUND is a
  // test hook).
  const int xround = env_int("SPBLAS_GFX950_PB_XROUND", 1);
  pick_tiling(n, w_env > 0 && w_env < max_cols ? w_env : max_cols, xround, xround, 4, &S, &W);
  // Matrices with few slices (n of a few million) would get runs of many hundred entries with
  // full-height bins: beyond the C prefetched chunks a run is read in a latency-exposed loop, and there
  // are too few bins to fill the chip.  Shorter bins bring the average run back to ~128 entries.
  int h_want = max_rows;
  {
    const int64_t bins_full = cdiv(m, max_rows);
    const int64_t run_full = nnz / (bins_full * S > 0 ? bins_full * S : 1);
    if (run_full > 160) {
      const int64_t h = (int64_t) ((double) m * 128.0 * (double) S / (double) (nnz > 0 ? nnz : 1));
      h_want = (int) (h < 64 ? 64 : (h > max_rows ? max_rows : h));
    }
  }
  const int bround = env_int("SPBLAS_GFX950_PB_ROUND", 512) * RW;
  pick_tiling(m, h_env > 0 && h_env < max_rows ? h_env : h_want, bround, bround / 2, 1, &NB, &H);
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
  const int64_t nseg = (int64_t) S * NB;
  if (nseg > (int64_t) 64 << 20 || S > 16384)  // S ints of LDS per inspect workgroup
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  pl->n_slices = S;
  pl->slice_cols = W;
  pl->n_rblk = NB;
  pl->rows_per_blk = H;
  pl->n_ksplit = pick_ksplit(cdiv(NB, RW), S);
  {
    const int64_t avg_run = nnz / (nseg > 0 ? nseg : 1);
    int C = env_int("SPBLAS_GFX950_PB_RCHUNKS", 0);
    if (C != 1 && C != 2 && C != 4)
      C = avg_run > 112 ? 4 : (avg_run > 48 ? 2 : 1);
    pl->rchunks = C;
    int GR = env_int("SPBLAS_GFX950_PB_RGROUP", 0);
    if (GR != 1 && GR != 2 && GR != 4)
      GR = 2;
    pl->rgroup = GR;
  }

  int rc;
  pb_tracer tr(s);
  int32_t* seg = nullptr;
  long long* partials = nullptr;
  if ((rc = dev_alloc((void**) &seg, (size_t) (nseg + 1) * 4, s)))
    return rc;
  tr.mark("seg allocated");
  pl->seg_ptr = seg;
  SPB_HIP(hipMemsetAsync(seg, 0, (size_t) (nseg + 1) * 4, s));
  const O* rowptr = static_cast<const O*>(pl->rowptr);
  // rows longer than the nnz window (the plan's long_rows list) stay out of the tiles
  pl->hub_len = pl->n_long > 0 ? pl->win : 0;
  pl->values_ptr = values_p;
  if (pl->hub_len > 0) {
    // workgroups per hub row: ~16K entries each, at most 64
    int64_t parts = cdiv(pl->max_row_len, 16384);
    pl->hub_parts = (int) (parts < 1 ? 1 : (parts > 64 ? 64 : parts));
    int rc_h = dev_alloc(&pl->s_hub_part, (size_t) pl->n_long * pl->hub_parts * sizeof(T), s);
    if (rc_h)
      return rc_h;
  }
  hipLaunchKernelGGL((pb_count_kernel<O>), dim3((unsigned) NB), dim3(256), (size_t) S * 4, s, m, rowptr, pl->colind, W, H, S,
                     NB, seg, pl->hub_len);
  tr.mark("count kernel");
  // One probe pass over the counters, read back once: entries per slice, non-empty tiles per slice, entries
  // per bin group.  AUTO uses them to decline matrices the plan does not suit; the work lists below use them
  // to spot column / row skew, and their total is the number of entries placed in tiles.
  const int64_t ngroups = cdiv(NB, RW);
  std::vector<unsigned long long> h_sum((size_t) (2 * S + ngroups));
  {
    unsigned long long* d_sum = nullptr;
    if ((rc = dev_alloc((void**) &d_sum, h_sum.size() * sizeof(unsigned long long), s)))
      return rc;
    hipLaunchKernelGGL(pb_balance_kernel, dim3((unsigned) (S + ngroups)), dim3(256), 0, s, S, NB, RW, seg, d_sum,
                       d_sum + S, d_sum + 2 * S);
    SPB_HIP(hipMemcpyAsync(h_sum.data(), d_sum, h_sum.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    SPB_HIP(hipStreamSynchronize(s));
    dev_free(d_sum, s);
  }
  unsigned long long placed_total = 0, max_slice = 0, max_group = 0, ne = 0;
  for (int i = 0; i < S; ++i) {
    placed_total += h_sum[(size_t) i];
    max_slice = std::max(max_slice, h_sum[(size_t) i]);
    ne += h_sum[(size_t) (S + i)];
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
        ((double) max_slice > 6.0 * mean_slice + 65536.0 || (double) max_group > 6.0 * mean_group + 65536.0))
      return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  }
  tr.mark("probe read back");
  if ((rc = dev_alloc((void**) &partials, (size_t) (cdiv(nseg, 2048) + 2) * sizeof(long long), s)))
    return rc;
  if ((rc = dev_alloc((void**) &pl->s_values, (size_t) nnz * sizeof(T), s)))
    return rc;
  if ((rc = dev_alloc((void**) &pl->s_colind, (size_t) nnz * 2, s)))
    return rc;
  // the reduce kernel's unconditional (clamped) loads touch entry `start` of an EMPTY run, which is
  // entry nnz for empty runs at the very end: keep one cache line of slack behind both streams
  if ((rc = dev_alloc((void**) &pl->s_lrow, (size_t) (nnz + 64) * 2, s)))
    return rc;
  if ((rc = dev_alloc((void**) &pl->s_perm, (size_t) nnz * 4, s)))
    return rc;
  if ((rc = dev_alloc((void**) &pl->s_products, (size_t) (nnz + 64) * sizeof(T), s)))
    return rc;
  if ((rc = dev_alloc((void**) &pl->s_segT, (size_t) nseg * sizeof(int2), s)))
    return rc;
  pl->device_bytes += (size_t) nnz * (2 * sizeof(T) + 8) + (size_t) nseg * 16;
  tr.mark("plan arrays allocated");
  (void) scan_counts_i32(s, nseg, seg, partials);  // the total is already known from the probe
  {
    // column skew: slice sizes from the segment offsets; when one slice is far above the average the
    // expand gets an explicit work list with workgroups in proportion to the slice sizes
    // A' is slice-major: a slice starts where the entries of the slices before it end
    std::vector<int64_t> start((size_t) S + 1, 0);
    for (int i = 0; i < S; ++i)
      start[(size_t) i + 1] = start[(size_t) i] + (int64_t) h_sum[(size_t) i];
    const int64_t max_len = (int64_t) max_slice;
    const int64_t total = start[(size_t) S];
    if (total > 0 && max_len * S > 3 * total) {
      const int cus = h->num_cus > 0 ? h->num_cus : 256;
      const int64_t target = std::max<int64_t>(cdiv(total, 2 * cus), 4 * (int64_t) W);
      std::vector<int4> items;
      for (int i = 0; i < S; ++i) {
        const int64_t lo = start[(size_t) i], hi = start[(size_t) i + 1];
        const int64_t np = std::max<int64_t>(1, cdiv(hi - lo, target));
        const int64_t per = (cdiv(hi - lo, np) + 3) & ~(int64_t) 3;
        for (int64_t k = 0; k < np; ++k) {
          // cut on multiples of 4 entries (absolute), the vector body of the kernel relies on it
          int64_t a = k == 0 ? lo : ((lo + k * per + 3) & ~(int64_t) 3);
          int64_t b = k == np - 1 ? hi : ((lo + (k + 1) * per + 3) & ~(int64_t) 3);
          a = std::min(a, hi);
          b = std::min(b, hi);
          if (b > a || (k == 0 && np == 1))
            items.push_back(make_int4(i, (int) a, (int) b, 0));
        }
      }
      if ((rc = dev_alloc(&pl->s_xitems, items.size() * sizeof(int4), s)))
        return rc;
      SPB_HIP(hipMemcpyAsync(pl->s_xitems, items.data(), items.size() * sizeof(int4), hipMemcpyHostToDevice, s));
      SPB_HIP(hipStreamSynchronize(s));
      pl->n_xitems = (int64_t) items.size();
    }
  }
  {
    // row skew: entries per (bin group, slice).  When one group is far above the average the reduce gets a
    // work list: every group is cut into as many slice ranges as its share of the entries asks for.
    // (the group totals come from the probe; the per-slice breakdown is only fetched for skewed matrices)
    const int64_t cells = ngroups * S;
    std::vector<int64_t> tot((size_t) ngroups, 0);
    const int64_t total = (int64_t) placed_total, max_tot = (int64_t) max_group;
    for (int64_t g = 0; g < ngroups; ++g)
      tot[(size_t) g] = (int64_t) h_sum[(size_t) (2 * S + g)];
    std::vector<int32_t> gs;
    if (total > 0 && max_tot * ngroups > 3 * total && S >= 16) {
      int32_t* d_gs = nullptr;
      if ((rc = dev_alloc((void**) &d_gs, (size_t) cells * 4, s)))
        return rc;
      hipLaunchKernelGGL(pb_group_slice_kernel, dim3((unsigned) cdiv(cells, 256)), dim3(256), 0, s, S, NB, RW, ngroups,
                         seg, d_gs);
      gs.resize((size_t) cells);
      SPB_HIP(hipMemcpyAsync(gs.data(), d_gs, (size_t) cells * 4, hipMemcpyDeviceToHost, s));
      SPB_HIP(hipStreamSynchronize(s));
      dev_free(d_gs, s);
    }
    if (total > 0 && max_tot * ngroups > 3 * total && S >= 16) {
      const int64_t target = std::max<int64_t>(total / 768, 16384);
      const int64_t block = (int64_t) RW * H;  // values per partial block
      std::vector<int4> items, split;
      int64_t poff = 0;
      for (int64_t g = 0; g < ngroups; ++g) {
        int64_t K = (tot[(size_t) g] + target / 2) / target;
        K = std::max<int64_t>(1, std::min<int64_t>(K, S / 8));
        if (K == 1) {
          items.push_back(make_int4((int) g, 0, S, -1));
          continue;
        }
        if (poff + K * block > (int64_t) INT32_MAX) {  // offsets are 32-bit: stop splitting
          items.push_back(make_int4((int) g, 0, S, -1));
          continue;
        }
        // cut at multiples of 8 slices (the duplicate-flag groups of the reduce kernel) by cumulative count
        const int64_t first = (int64_t) items.size();
        int lo = 0;
        int64_t run = 0, done = 0;
        int made = 0;
        for (int sl = 0; sl < S; ++sl) {
          run += gs[(size_t) (g * S + sl)];
          const bool boundary = ((sl + 1) % 8 == 0) || sl + 1 == S;
          if (boundary && made + 1 < K && (done + run) * K >= (int64_t) (made + 1) * tot[(size_t) g] && sl + 1 < S) {
            items.push_back(make_int4((int) g, lo, sl + 1, 0));
            lo = sl + 1;
            done += run;
            run = 0;
            ++made;
          }
        }
        items.push_back(make_int4((int) g, lo, S, 0));
        const int64_t Kg = (int64_t) items.size() - first;
        if (Kg == 1) {
          items.back().w = -1;
          continue;
        }
        for (int64_t k = 0; k < Kg; ++k)
          items[(size_t) (first + k)].w = (int) (poff + k * block);
        split.push_back(make_int4((int) g, (int) Kg, (int) poff, 0));
        poff += Kg * block;
      }
      if (!split.empty()) {
        if ((rc = dev_alloc(&pl->s_ritems, items.size() * sizeof(int4), s)) ||
            (rc = dev_alloc(&pl->s_rsplit, split.size() * sizeof(int4), s)) ||
            (rc = dev_alloc(&pl->s_rpartial, (size_t) poff * sizeof(T), s)))
          return rc;
        SPB_HIP(hipMemcpyAsync(pl->s_ritems, items.data(), items.size() * sizeof(int4), hipMemcpyHostToDevice, s));
        SPB_HIP(hipMemcpyAsync(pl->s_rsplit, split.data(), split.size() * sizeof(int4), hipMemcpyHostToDevice, s));
        SPB_HIP(hipStreamSynchronize(s));
        pl->n_ritems = (int64_t) items.size();
        pl->n_rsplit = (int64_t) split.size();
        pl->device_bytes += (size_t) poff * sizeof(T);
      }
    }
  }
  tr.mark("scan + work lists");
  if (S <= PB_STAGE_MAX_S && env_int("SPBLAS_GFX950_PB_STAGED_SCATTER", 1)) {
    // a quarter of the staging area at most goes to the row table (one entry per 64 matrix entries of a bin)
    const int rt_len = 2048;
    const int cap = (int) (((size_t) PB_STAGE_LDS - (size_t) 12 * S - (size_t) 4 * (H + 1) - (size_t) 4 * rt_len - 128) /
                           (6 + sizeof(T))) & ~7;
    SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pb_scatter_staged_kernel<T, O>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, PB_STAGE_LDS - 64));
    hipLaunchKernelGGL((pb_scatter_staged_kernel<T, O>), dim3((unsigned) NB), dim3(PB_STAGE_THREADS),
                       (size_t) PB_STAGE_LDS - 64, s, m, rowptr, pl->colind, static_cast<const T*>(values_p), W, H, S,
                       NB, seg, static_cast<T*>(pl->s_values), reinterpret_cast<uint16_t*>(pl->s_colind), pl->s_lrow,
                       reinterpret_cast<int32_t*>(pl->s_perm), pl->hub_len, cap, rt_len);
  } else
  hipLaunchKernelGGL((pb_scatter_kernel<T, O>), dim3((unsigned) NB), dim3(256), (size_t) S * 4, s, m, rowptr,
                     pl->colind, static_cast<const T*>(values_p), W, H, S, NB, seg, static_cast<T*>(pl->s_values),
                     reinterpret_cast<uint16_t*>(pl->s_colind), pl->s_lrow, reinterpret_cast<int32_t*>(pl->s_perm),
                     pl->hub_len);
  tr.mark("scatter");
  hipLaunchKernelGGL(pb_transpose_seg_kernel, dim3((unsigned) cdiv(nseg, 256)), dim3(256), 0, s, S, NB, seg,
                     static_cast<int2*>(pl->s_segT));
  SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pb_flag_dups_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_BYTES));
  hipLaunchKernelGGL(pb_flag_dups_kernel, dim3((unsigned) cdiv(NB, PB_RWAVES)), dim3(PB_RTHREADS),
                     (size_t) PB_RWAVES * 2 * H, s, H, S, (int64_t) NB, pl->rchunks, pl->rgroup,
                     static_cast<const int2*>(pl->s_segT), pl->s_lrow);
  SPB_HIP(hipGetLastError());
  SPB_HIP(hipStreamSynchronize(s));
  tr.mark("flags");
  dev_free(partials, s);
  // both kernels may use up to 80 KiB of dynamic LDS
  SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pb_expand_kernel<T>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, xlds > PB_LDS_BYTES ? xlds : PB_LDS_BYTES));
  {
    const void* fn = pb_reduce_fn<T>(pl->rwaves, pl->rchunks, pl->rgroup);
    SPB_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_BYTES));
  }
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
  pl->values_ptr = values;  // the hub rows read the caller's array directly
  if (pl->s_placed == 0)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  hipLaunchKernelGGL((pb_update_values_kernel<T>), dim3((unsigned) cdiv(pl->s_placed, 256)), dim3(256), 0, h->stream,
                     pl->s_placed, reinterpret_cast<const int32_t*>(pl->s_perm), static_cast<const T*>(values),
                     static_cast<T*>(pl->s_values));
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spmv_sliced_update(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* values) {
  return pl->value_type == SPBLAS_GFX950_F32 ? sliced_update_typed<float>(h, pl, values)
                                             : sliced_update_typed<double>(h, pl, values);
}

static int pick_ksplit(int64_t groups, int S) {
  int K = env_int("SPBLAS_GFX950_PB_KSPLIT", 0);
  if (K <= 0) {
    K = 1;
    while (groups * K < 384 && K < 32 && S / (2 * K) >= 8)
      K *= 2;
  }
  return K > S ? S : (K < 1 ? 1 : K);
}

template <typename T>
static int sliced_expand_typed(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* x) {
  pl->last_x = x;  // the hub rows are computed in the reduce stage and gather x themselves
  const int32_t* seg = reinterpret_cast<const int32_t*>(pl->seg_ptr);
  // one wave of workgroups (2 per CU with 80 KiB slices, 1 with 160 KiB), but never shares so small that
  // re-loading the x slice dominates
  const int4* items = static_cast<const int4*>(pl->s_xitems);
  const int cus = h->num_cus > 0 ? h->num_cus : 256;
  const size_t xbytes = (size_t) pl->slice_cols * sizeof(T);
  int64_t nwg = (int64_t) cus * (xbytes > (size_t) PB_LDS_BYTES ? 1 : 2);
  const int64_t total = pl->s_placed;
  const int64_t min_share = 2 * (int64_t) pl->slice_cols;
  if (nwg * min_share > total) {
    // ... but not fewer workgroups than slices (row shards: many slices with few entries each)
    const int64_t floor_wg = pl->n_slices < nwg ? pl->n_slices : nwg;
    nwg = total / min_share;
    if (nwg < floor_wg)
      nwg = floor_wg;
  }
  if (nwg < 1)
    nwg = 1;
  const int share = (int) ((cdiv(total > 0 ? total : 1, nwg) + 3) & ~(int64_t) 3);
  const dim3 grid = items ? dim3((unsigned) pl->n_xitems) : dim3((unsigned) cdiv(total > 0 ? total : 1, share));
  hipLaunchKernelGGL((pb_expand_kernel<T>), grid, dim3(PB_THREADS), xbytes, h->stream, pl->n, pl->slice_cols,
                     (int) pl->n_rblk, seg, static_cast<const T*>(pl->s_values),
                     reinterpret_cast<const uint16_t*>(pl->s_colind), static_cast<const T*>(x),
                     static_cast<T*>(pl->s_products), items, (int) pl->n_slices, share);
  SPB_HIP(hipGetLastError());
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
  if (peers_p && pl->hub_len > 0 && pl->n_long > 0)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;  // the fused all-gather epilogue does not cover hub rows
  const T alpha = *static_cast<const T*>(alpha_p), beta = *static_cast<const T*>(beta_p);
  const int RW = pl->rwaves;
  const int64_t groups = cdiv(wb_end - wb_begin, RW);
  // slices per split: whole batches of 8 runs, so the duplicate-flag groups stay aligned
  int k_want = pick_ksplit(groups, pl->n_slices);
  if (h->max_ksplit > 0 && k_want > h->max_ksplit)
    k_want = (int) h->max_ksplit;  // striped callers run several reduces side by side
  const int s_per = (int) cdiv(cdiv(pl->n_slices, k_want), 8) * 8;
  const int K = (int) cdiv(pl->n_slices, s_per);
  const int64_t r_lo = wb_begin * pl->rows_per_blk;
  const int64_t r_hi = wb_end * pl->rows_per_blk < pl->m ? wb_end * pl->rows_per_blk : pl->m;
  bool K_used_items = false;
  const bool will_use_items = pl->s_ritems && !peers_p && wb_begin == 0 && wb_end == pl->n_rblk;
  if (!will_use_items && K > 1 && pl->s_partial_k < K) {  // grow the partial-sum workspace (stream ordered)
    dev_free(pl->s_partial, s);
    pl->s_partial = nullptr;
    int rc = dev_alloc(&pl->s_partial, (size_t) K * pl->m * sizeof(T), s);
    if (rc)
      return rc;
    pl->s_partial_k = K;
  }
  {
    const int2* segT = static_cast<const int2*>(pl->s_segT);
    const T* Pp = static_cast<const T*>(pl->s_products);
    const uint16_t* rowp = pl->s_lrow;
    T* yp = static_cast<T*>(y);
    T* part = K > 1 ? static_cast<T*>(pl->s_partial) : nullptr;
    int64_t mm = pl->m, pstride = pl->m;
    int Hw = pl->rows_per_blk, S = pl->n_slices, sp = s_per;
    T a = alpha, b = beta;
    T* const* peers = reinterpret_cast<T* const*>(peers_p);
    const int4* ritems = nullptr;
    void* args[] = {&mm, &Hw, &S, &wb_begin, &wb_end, &segT, &Pp, &rowp, &yp, &a, &b, &sp, &part, &pstride,
                    &peers, &n_peers, &peer_off, &ritems};
    if (pl->s_ritems && !peers_p && wb_begin == 0 && wb_end == pl->n_rblk) {
      // row-skewed matrix, whole range: explicit work list (built at inspect), compact partial sums
      ritems = static_cast<const int4*>(pl->s_ritems);
      part = static_cast<T*>(pl->s_rpartial);
      pstride = 0;
      SPB_HIP(hipLaunchKernel(pb_reduce_fn<T>(RW, pl->rchunks, pl->rgroup), dim3((unsigned) pl->n_ritems), dim3(RW * 64),
                              args, (size_t) RW * (pl->rows_per_blk + 64) * sizeof(T), s));
      if (pl->n_rsplit > 0)
        hipLaunchKernelGGL((pb_combine_items_kernel<T>),
                           dim3((unsigned) pl->n_rsplit, (unsigned) cdiv((int64_t) RW * pl->rows_per_blk, 256)), dim3(256),
                           0, s, static_cast<const int4*>(pl->s_rsplit), (int64_t) RW * pl->rows_per_blk, pl->m,
                           static_cast<const T*>(pl->s_rpartial), static_cast<T*>(y), alpha, beta);
      K_used_items = true;
    } else
    SPB_HIP(hipLaunchKernel(pb_reduce_fn<T>(RW, pl->rchunks, pl->rgroup), dim3((unsigned) groups, (unsigned) K), dim3(RW * 64), args,
                            (size_t) RW * (pl->rows_per_blk + 64) * sizeof(T), s));
  }
  if (!K_used_items && K > 1 && r_hi > r_lo)
    hipLaunchKernelGGL((pb_combine_kernel<T>), dim3((unsigned) cdiv(r_hi - r_lo, 256)), dim3(256), 0, s, r_lo, r_hi, K,
                       static_cast<const T*>(pl->s_partial), pl->m, static_cast<T*>(y), alpha, beta,
                       reinterpret_cast<T* const*>(peers_p), n_peers, peer_off);
  if (pl->hub_len > 0 && pl->n_long > 0) {
    // rows kept out of the tiles: y[row] += alpha * (row . x), for the rows of this bin range
    if (!pl->values_ptr || !pl->last_x)
      return SPBLAS_GFX950_STATUS_INVALID_VALUE;
    if (!pl->s_hub_part)
      return SPBLAS_GFX950_STATUS_INVALID_VALUE;
    const dim3 grid((unsigned) pl->n_long, (unsigned) pl->hub_parts);
    T* part = static_cast<T*>(pl->s_hub_part);
    if (pl->offset_type == SPBLAS_GFX950_I32)
      hipLaunchKernelGGL((pb_hub_rows_kernel<T, int32_t>), grid, dim3(256), 0, s, pl->n_long, pl->long_rows,
                         static_cast<const int32_t*>(pl->rowptr), pl->colind, static_cast<const T*>(pl->values_ptr),
                         static_cast<const T*>(pl->last_x), part, r_lo, r_hi);
    else
      hipLaunchKernelGGL((pb_hub_rows_kernel<T, int64_t>), grid, dim3(256), 0, s, pl->n_long, pl->long_rows,
                         static_cast<const int64_t*>(pl->rowptr), pl->colind, static_cast<const T*>(pl->values_ptr),
                         static_cast<const T*>(pl->last_x), part, r_lo, r_hi);
    hipLaunchKernelGGL((pb_hub_finish_kernel<T>), dim3((unsigned) cdiv(pl->n_long, 256)), dim3(256), 0, s, pl->n_long,
                       pl->hub_parts, pl->long_rows, part, static_cast<T*>(y), alpha, r_lo, r_hi);
  }
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

// slice split the reduce of the WHOLE row range would use, and the partial-sum workspace for it:
// callers that reduce stripe by stripe on several streams cap K with it (handle->max_ksplit) and
// reserve the workspace before they fork, so no stripe allocates.
int spmv_sliced_full_ksplit(spblas_gfx950_plan_s* pl) {
  return pick_ksplit(cdiv(pl->n_rblk, pl->rwaves), pl->n_slices);
}

int spmv_sliced_reserve_partial(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, int K) {
  if (K <= 1 || pl->s_partial_k >= K)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  hipStream_t s = h->stream;
  dev_free(pl->s_partial, s);
  pl->s_partial = nullptr;
  pl->s_partial_k = 0;
  const size_t tsz = pl->value_type == SPBLAS_GFX950_F32 ? 4 : 8;
  int rc = dev_alloc(&pl->s_partial, (size_t) K * pl->m * tsz, s);
  if (rc)
    return rc;
  pl->s_partial_k = K;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spmv_sliced_expand(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* x) {
  return pl->value_type == SPBLAS_GFX950_F32 ? sliced_expand_typed<float>(h, pl, x)
                                             : sliced_expand_typed<double>(h, pl, x);
}

// bins whose first row lies in [row_begin, row_end)
int spmv_sliced_reduce_rows(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* alpha, const void* beta,
                            void* y, int64_t row_begin, int64_t row_end, void* const* peers, int n_peers,
                            int64_t peer_off) {
  const int64_t H = pl->rows_per_blk;
  const int64_t wb0 = cdiv(row_begin, H);
  int64_t wb1 = cdiv(row_end, H);
  if (wb1 > pl->n_rblk)
    wb1 = pl->n_rblk;
  return pl->value_type == SPBLAS_GFX950_F32
             ? sliced_reduce_typed<float>(h, pl, alpha, beta, y, wb0, wb1, peers, n_peers, peer_off)
             : sliced_reduce_typed<double>(h, pl, alpha, beta, y, wb0, wb1, peers, n_peers, peer_off);
}

int spmv_sliced_exec(spblas_gfx950_handle_t h, spblas_gfx950_plan_s* pl, const void* alpha, const void* x,
                     const void* beta, void* y) {
  int rc = spmv_sliced_expand(h, pl, x);
  if (rc)
    return rc;
  return spmv_sliced_reduce_rows(h, pl, alpha, beta, y, 0, pl->m, nullptr, 0, 0);
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
  dev_free(pl->s_partial, s);
  dev_free(pl->s_hub_part, s);
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
  pl->seg_ptr = pl->s_colind = pl->s_values = pl->s_products = pl->s_segT = pl->s_perm = nullptr;
  pl->s_lrow = nullptr;
  pl->n_slices = 0;
}

} // namespace spb
