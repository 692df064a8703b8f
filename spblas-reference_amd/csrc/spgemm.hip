// CSR x CSR -> CSR SpGEMM for gfx950:  C = alpha * A * B  (+ beta * D with an addend).
//
// Replaces the rocsparse_spgemm stages used at
// /root/reference/include/spblas/vendor/rocsparse/multiply_spgemm.hpp:94-115 (buffer_size+nnz),
// :137-144 (compute), :167-172 (symbolic), :209-213 (numeric).  Results follow the CPU
// path include/spblas/algorithms/detail/spgemm/spgemm_gustavsons.hpp:17-89: nnz(C) is the
// STRUCTURAL count (no numeric cancellation), columns ascending within each row.
//
// Row-wise Gustavson with per-row accumulators sized from an upper bound
// ub[i] = sum_{k in A_i} len(B_k) (number of products):
//   bin 0: ub == 0                      empty row
//   bin 1: ub <= 64     LDS hash 128    16 lanes / row, 16 rows / workgroup
//   bin 2: ub <= 256    LDS hash 512    one wavefront / row, 4 rows / workgroup
//   bin 3: ub <= 1024   LDS hash 2048   two wavefronts / row, 2 rows / workgroup
//   bin 4: ub <= 4096   LDS hash 8192   one workgroup / row
//   bin 5: ub  > 4096   dense bitmap (+ dense values) per workgroup in HBM
// Symbolic counts distinct keys; numeric accumulates with LDS float atomics, compacts,
// rank-sorts the (unique) keys and writes colind/values in ascending column order.
// Integer/byte traffic bound: algorithmic bytes = A + B once + C once.
//
// Four-argument form C = alpha*A*B + beta*D (multiply_spgemm.hpp:147-214; expected result
// test/gtest/device/rocsparse/spgemm_4args_test.cpp:78-95): row i of D is fed through the same
// accumulator after the products of row i, so pattern(C) = pattern(AB) U pattern(D) and the row
// bound counts len(D_i) as well.
//
// add(a, b, c):  C = alpha*A + beta*B  (algorithms/add_impl.hpp:40-108) is the same
// walk with B := identity (b_rowptr == nullptr in the kernels: "row" kk of B is the single entry
// (kk, 1), the product is the A entry itself) and D := the second summand.  add_inspect is the
// symbolic pass, add_compute the numeric one; columns come out ascending like the CPU SPA + sort.
#include "common.hpp"
#include <type_traits>
#include "scan.hpp"

#include <cstdlib>
#include <new>

#define SPG_NBINS 6
// counters of the binning pass: the accumulator bins plus one for the rows of bin 2 that a wavefront can take in ONE round
// of vector loads (placed at the end of bin 2's range of perm[]: spg_sort_symbolic_kernel / spg_direct_kernel)
#define SPG_NCNT 7
#define SPG_SORTABLE 6
#ifndef SPG_NBK64
#define SPG_NBK64 64   // buckets of the rank sort in the numeric wave-per-row kernel (round 4, same box, cfg5 one-shot fill:
                       // 32 -> 1.62 ms, 64 -> 1.51, 128 -> 1.65, 256 -> 1.90: the bucket scan costs more than the rank loop saves)
#endif
#ifndef SPG_DIR_NBK
#define SPG_DIR_NBK 128  // buckets of the rank sort in spg_direct_kernel (64 / 128 / 256)
#endif
#ifndef SPG_DIR_READ2
#define SPG_DIR_READ2 1  // its rank loop compares two keys per iteration
#endif
#ifndef SPG_INREG
#define SPG_INREG 1    // A/B: 0 = direct rows always go through the product list
#endif

struct spblas_gfx950_spgemm_s {
  int64_t m = 0, k = 0, n = 0, a_nnz = 0, b_nnz = 0, c_nnz = -1;
  const int32_t *a_rowptr = nullptr, *a_colind = nullptr, *b_rowptr = nullptr, *b_colind = nullptr;
  const int32_t *d_rowptr = nullptr, *d_colind = nullptr;  // addend pattern (nullptr: C = alpha*A*B)
  int64_t d_nnz = 0;
  bool has_addend = false;  // pattern of D was part of the last symbolic pass
  bool identity_b = false;  // B is the identity (add(): C = alpha*A + beta*D)
  int32_t* rowptr = nullptr;  // [m+1] device copy of C's row offsets
  int32_t* perm = nullptr;    // [m] rows grouped by bin
  int64_t bin_off[SPG_NBINS + 1] = {0, 0, 0, 0, 0, 0, 0};
  int sub = 16;  // lanes cooperating on one B row
  // bin-4 workspace
  int dense_blocks = 0;
  uint32_t* dense_bits = nullptr;  // [dense_blocks * ceil(n/32)]
  void* dense_vals = nullptr;      // [dense_blocks * n] T
  int dense_vals_type = -1;
  // numeric reuse (multiply_numeric after the first fill; vendor/rocsparse/multiply_spgemm.hpp:178-214): the first
  // numeric pass of a symbolic result records, for every product of the rows in the LDS-hash bins 1-2, the rank of
  // its column in the (sorted) output row; later passes accumulate by rank -- no hash, no compaction, no sort
  int32_t* r_pbase = nullptr;   // [m + 1] first product of every row in the enumeration order
  uint8_t* r_rank = nullptr;    // [products] rank of the product's column in its output row (rows of <= 256 products)
  int32_t* r_pbase3 = nullptr;  // the same pair for the rows of bin 3 (257 .. 1024 products: two-byte ranks), e.g. the
  uint16_t* r_rank3 = nullptr;  // 27-point stencil times itself or a Galerkin product
  int32_t* r_cols = nullptr;    // [nnz(C)] the sorted column indices (the caller may pass other arrays later)
  const int32_t* r_last_colind = nullptr;  // the caller's column array the last numeric pass filled
  // (start, length) of the B row every entry of A selects, written once by the symbolic pass: every later kernel
  // streams 8 B per A entry instead of chasing a_colind -> b_rowptr[.], b_rowptr[. + 1].  Under the B-row gathers of
  // a fill b_rowptr (4 MB at cfg5) does not stay in the L2s, and those lookups were 1.35 of the 4.08 GB a fill by
  // rank read (profiles/r02e_spgemm_pmc.md).  nullptr (no B, out of memory, SPBLAS_GFX950_SPG_ADESC=0): the kernels
  // look the rows up themselves.
  int2* r_adesc = nullptr;      // [a_nnz]
  // Direct rows of bin 2 (spg_direct_kernel): rows whose product count equals their structural length (nothing to
  // accumulate) and whose A row is one round of loads, listed by the symbolic pass as (first A entry, A entries, first
  // output position, row); the other rows of the bin keep the hash kernel (dir_rest).  nullptr: everything hashes.
  int4* dir_desc = nullptr;     // [n_dir]
  int32_t* dir_rest = nullptr;  // [n_rest] rows of bin 2 that are not sortable
  int64_t n_dir = 0, n_rest = 0;
  int64_t n_nodup = 0;          // the first n_nodup descriptors: no two products share a column
  int64_t n_sortable = 0;       // rows at the end of bin 2's range that are one round of vector loads
  int2* b_pack = nullptr;       // fp32 fills with many direct rows: (column, value bits) of B interleaved, REWRITTEN BY EVERY FILL
  int2* dir_ddesc = nullptr;    // with an addend: (first entry, length) of the addend's row of every direct row, in dir_desc order
  int32_t* sym_flag = nullptr;  // symbolic pass only: [n_sortable] 1 = no two products of the row share a column
  bool r_ready = false;
  int numeric_calls = 0;        // numeric passes since the symbolic one (the SECOND records: a one-shot fill pays nothing)
};

namespace spb {

__device__ __forceinline__ int spg_bin_of(int64_t ub) {
  if (ub == 0)
    return 0;
  if (ub <= 64)
    return 1;
  if (ub <= 256)
    return 2;
  if (ub <= 1024)
    return 3;
  if (ub <= 4096)
    return 4;
  return 5;
}

// ub per row (8 lanes per row) + per-bin row counts.
__global__ __launch_bounds__(256) void spg_bound_kernel(int64_t m, const int32_t* __restrict__ a_rowptr,
                                                        const int32_t* __restrict__ a_colind,
                                                        const int32_t* __restrict__ b_rowptr,
                                                        const int32_t* __restrict__ d_rowptr,
                                                        const int2* __restrict__ adesc,
                                                        int32_t* __restrict__ bin_of_row,
                                                        unsigned long long* __restrict__ bin_count, int sub, int64_t b_nnz,
                                                        int sortable_ok, int2* __restrict__ adesc_out) {
  // grid-stride over groups of 32 rows: the bin histogram stays in LDS for the whole workgroup and reaches the
  // global counters once per workgroup -- with one workgroup per 32 rows the 31 k same-address atomics of a
  // 1 M-row matrix (~11 ns each, serialised) were the whole 0.38 ms of this kernel
  __shared__ unsigned int hist[SPG_NCNT];
  if (threadIdx.x < SPG_NCNT)
    hist[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x % 8;
  for (int64_t row = (int64_t) blockIdx.x * 32 + threadIdx.x / 8; row < m; row += (int64_t) gridDim.x * 32) {
    int64_t ub = 0;
    int bad = 0;  // a B row longer than `sub`, or one whose range padded to whole vectors of four leaves B's arrays
    const int p0 = a_rowptr[row], p1 = a_rowptr[row + 1];
    for (int p = p0 + lane; p < p1; p += 8) {
      if (adesc_out) {
        // (start, length) of the B row of every A entry, written here (round 5: a pass of its own over a_colind wrote them
        // before this kernel read them back: 128 + 54 us at cfg5 for what one pass does)
        struct __attribute__((packed, aligned(4))) pair_t {
          int lo, hi;
        };
        const pair_t r = *reinterpret_cast<const pair_t*>(b_rowptr + a_colind[p]);  // one 8-byte load, 4-byte aligned
        const int2 dd = make_int2(r.lo, r.hi - r.lo);
        adesc_out[p] = dd;
        ub += dd.y;
        bad |= (int) (dd.y > sub) | (int) ((int64_t) dd.x + ((dd.y + 3) & ~3) > b_nnz);
        continue;
      }
      if (adesc) {
        const int2 dd = adesc[p];
        ub += dd.y;
        bad |= (int) (dd.y > sub) | (int) ((int64_t) dd.x + ((dd.y + 3) & ~3) > b_nnz);
        continue;
      }
      const int kk = a_colind[p];
      ub += b_rowptr ? b_rowptr[kk + 1] - b_rowptr[kk] : 1;  // no B: identity (add(), see below)
    }
    ub = group_sum_c<8>(ub);
    bad = group_sum_c<8>(bad);
    const int64_t ub_prod = ub;
    int d_len = 0;
    if (d_rowptr) {
      d_len = d_rowptr[row + 1] - d_rowptr[row];
      ub += d_len;
    }
    if (lane == 0) {
      int b = spg_bin_of(ub);
      // sortable: the products fit one round of a wavefront's vector loads (<= 256) and the addend's row, if there is one,
      // one entry per lane (round 6: cfg5's rows with an addend of 16 entries count 272 and used to fall to the 2 048-slot
      // hash of bin 3 -- 6.3 ms per fill against 0.85 without the addend)
      if ((b == 2 || (b == 3 && d_rowptr)) && sortable_ok && !bad && p1 - p0 <= 4 * (64 / sub) && p1 - p0 <= 64 &&
          ub_prod <= 256 && d_len <= 64)
        b = SPG_SORTABLE;
      bin_of_row[row] = b;
      atomicAdd(&hist[b], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < SPG_NCNT && hist[threadIdx.x])
    atomicAdd(&bin_count[threadIdx.x], (unsigned long long) hist[threadIdx.x]);
}

__global__ __launch_bounds__(1024) void spg_fill_perm_kernel(int64_t m, const int32_t* __restrict__ bin_of_row,
                                                             unsigned long long* __restrict__ cursor,
                                                             int32_t* __restrict__ perm) {
  // workgroup-aggregated append: one global atomic per (workgroup of 1024 rows, bin).  With uniform inputs every
  // row lands in the same bin: per-row atomics on that one cursor serialise (12 ms at 1 M rows), per-wavefront
  // atomics still cost 0.19 ms (15.6 k of them at ~11 ns), per-workgroup ones 1 k.
  __shared__ unsigned s_cnt[SPG_NCNT];
  __shared__ unsigned long long s_base[SPG_NCNT];
  if (threadIdx.x < SPG_NCNT)
    s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t row = (int64_t) blockIdx.x * 1024 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int bin = row < m ? bin_of_row[row] : -1;
  unsigned my_off = 0;  // position of this row among the workgroup's rows of its bin
  for (int b = 0; b < SPG_NCNT; ++b) {
    const unsigned long long mask = __ballot(bin == b);
    if (mask == 0)
      continue;
    unsigned base = 0;
    const int leader = __builtin_ctzll(mask);
    if (lane == leader)
      base = atomicAdd(&s_cnt[b], (unsigned) __popcll(mask));
    base = __shfl(base, leader);
    if (bin == b)
      my_off = base + (unsigned) __popcll(mask & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  if (threadIdx.x < SPG_NCNT && s_cnt[threadIdx.x])
    s_base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], (unsigned long long) s_cnt[threadIdx.x]);
  __syncthreads();
  if (bin >= 0)
    perm[s_base[bin] + my_off] = (int32_t) row;
}

// LDS floating-point add through an integer compare-and-swap loop.  The native LDS float atomic
// (ds_add_f32) retires only 0.33 lanes/clk/CU on this part (tools/ubench/lds_atomic.hip: 12x slower
// than integer LDS atomics), and a hash slot is rarely contended, so read + cmpswap usually
// succeeds at the first attempt.
// (The first read goes through an explicit LDS pointer: a volatile read through the generic one was compiled as a FLAT
// load with system-scope cache bits -- a vector-memory operation that waits behind every global load in flight.)
__device__ __forceinline__ void spg_lds_add(float* addr, float v) {
  int* ai = reinterpret_cast<int*>(addr);
  int old = *(volatile __attribute__((address_space(3))) int*) ai;
  while (true) {
    const int assumed = old;
    old = atomicCAS(ai, assumed, __float_as_int(__int_as_float(assumed) + v));
    if (old == assumed)
      break;
  }
}
__device__ __forceinline__ void spg_lds_add(double* addr, double v) {
  unsigned long long* ai = reinterpret_cast<unsigned long long*>(addr);
  unsigned long long old = *(volatile __attribute__((address_space(3))) unsigned long long*) ai;
  while (true) {
    const unsigned long long assumed = old;
    old = atomicCAS(ai, assumed, (unsigned long long) __double_as_longlong(__longlong_as_double((long long) assumed) + v));
    if (old == assumed)
      break;
  }
}

__device__ __forceinline__ unsigned spg_hash(int key, int log2hs) {
  return ((unsigned) key * 0x9E3779B1u) >> (32 - log2hs);
}

// Hash kernel.  TPR threads cooperate on one row; a workgroup of 256 threads
// handles 256/TPR rows taken from perm[first .. first+count).  SUB lanes walk one
// B row together.  NUMERIC = false: count distinct columns into row_nnz[row].
// NUMERIC = true: accumulate, sort, write colind/values at c_rowptr[row].
// A team of <= 64 lanes lives inside one wavefront and only touches its own LDS region: its phases
// are ordered by program order plus a wave-level barrier (LDS operations of a wave complete in order),
// so rows of different length in one workgroup do not wait for each other.  Wider teams need the
// workgroup barrier.
template <int TPR>
__device__ __forceinline__ void spg_team_sync() {
  if constexpr (TPR <= 64) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

template <typename T, int LOG2HS, int TPR, bool NUMERIC>
__global__ __launch_bounds__(256) void spg_hash_kernel(
    int64_t count, const int32_t* __restrict__ perm, const int32_t* __restrict__ a_rowptr,
    const int32_t* __restrict__ a_colind, const T* __restrict__ a_values, const int32_t* __restrict__ b_rowptr,
    const int32_t* __restrict__ b_colind, const T* __restrict__ b_values, int32_t* __restrict__ c_rowptr,
    int32_t* __restrict__ c_colind, T* __restrict__ c_values, T alpha, int sub, unsigned bucket_mul,
    const int32_t* __restrict__ d_rowptr, const int32_t* __restrict__ d_colind, const T* __restrict__ d_values,
    T beta, int b_has_entries, const int2* __restrict__ adesc) {
  // adesc != nullptr: (start, length) of the B row of every A entry (then a_colind / b_rowptr are not read)
  constexpr int HS = 1 << LOG2HS;
  constexpr int RPB = 256 / TPR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* keys = reinterpret_cast<int*>(smem);                            // [RPB][HS]
  int* cnt = keys + RPB * HS;                                          // [RPB] (+ pad to 4)
  T* vals = reinterpret_cast<T*>(cnt + ((RPB + 3) & ~3) + (sizeof(T) == 8 ? 0 : 0));  // [RPB][HS]
  int* list = NUMERIC ? reinterpret_cast<int*>(vals + RPB * HS) : nullptr;  // [RPB][HS/2] compacted keys
  // then [RPB][HS/2] compacted values (T), then the sort workspace [RPB][2*NBK+2] ints
  int* sortws = NUMERIC ? reinterpret_cast<int*>(reinterpret_cast<T*>(list + RPB * (HS / 2)) + RPB * (HS / 2)) : nullptr;

  const int team = threadIdx.x / TPR;
  const int lt = threadIdx.x % TPR;
  const int64_t idx = (int64_t) blockIdx.x * RPB + team;
  const bool live = idx < count;
  const int row = live ? perm[idx] : 0;
  int* tkeys = keys + team * HS;
  T* tvals = NUMERIC ? vals + team * HS : nullptr;

  if (lt == 0)
    cnt[team] = 0;
  // sort workspace of the numeric phase (see below): zeroed here so that the compaction pass can count the
  // buckets on the fly
  // (one bucket per lane of the team, at most 64; SPG_NBK64 is an A/B knob for the wave-per-row numeric kernel)
  constexpr int NBK = (NUMERIC && TPR == 64) ? SPG_NBK64 : (TPR < 64 ? TPR : 64);
  int* bcnt = NUMERIC ? sortws + team * (2 * NBK + 2) : nullptr;  // [NBK+1] counts -> offsets
  int* bfill = NUMERIC ? bcnt + NBK + 1 : nullptr;                // [NBK] cursors
  // bucket of a key = floor(key * NBK / ncols), as a 32x32 -> high-32 multiply by bucket_mul = floor(NBK * 2^32 / ncols),
  // which the host computes (spg_bucket_mul: a 64-bit division per wavefront was 120 scalar instructions of a
  // 900-instruction row): monotone in the key, < NBK for key < ncols.  `sub` is a power of two: shifts, not divisions.
  const int lsub = __builtin_ctz((unsigned) sub);
  // exclusive scan of the bucket counts in place (bcnt[NBK] = total): one bucket per lane, or NBK / TPR per lane
  auto scan_buckets = [&]() {
    if constexpr (NBK > TPR) {
      constexpr int BPL = NBK / TPR;
      int c[BPL], tot = 0;
#pragma unroll
      for (int j = 0; j < BPL; ++j) {
        c[j] = bcnt[lt * BPL + j];
        tot += c[j];
      }
      int incl = tot;
#pragma unroll
      for (int o = 1; o < TPR; o <<= 1) {
        const int t = __shfl_up(incl, o, TPR);
        if (lt >= o)
          incl += t;
      }
      int run = incl - tot;
#pragma unroll
      for (int j = 0; j < BPL; ++j) {
        bcnt[lt * BPL + j] = run;
        run += c[j];
      }
      if (lt == TPR - 1)
        bcnt[NBK] = incl;
    } else {
      if (lt < NBK) {
        const int c = bcnt[lt];
        int incl = c;
        for (int o = 1; o < NBK; o <<= 1) {
          const int t = __shfl_up(incl, o, NBK);
          if (lt >= o)
            incl += t;
        }
        bcnt[lt] = incl - c;
        if (lt == NBK - 1)
          bcnt[NBK] = incl;
      }
    }
  };
  if (NUMERIC) {
    for (int i = lt; i <= NBK; i += TPR)
      bcnt[i] = 0;
    for (int i = lt; i < NBK; i += TPR)
      bfill[i] = 0;
  }
  // Direct path (round 4; numeric pass, one wavefront per row, A row of <= 64 entries, B given through adesc): when the
  // row's product count equals its structural length -- known since the symbolic pass: c_rowptr -- no two products share
  // a column, so there is nothing to accumulate: the products go straight into the compacted list at their enumeration
  // position (an exclusive scan of the B-row lengths over the lanes), no table initialisation, no compare-and-swap
  // insert, no compaction sweep.  96.6 % of the rows of BASELINE cfg5 (1 M x 1 M, 16 entries per row, uniform random).
  bool direct = false;
  int direct_d = 0;
  if constexpr (NUMERIC && TPR == 64) {
    if (adesc && b_rowptr && b_has_entries && !d_rowptr) {
      int qb = 0, qe = 0;
      T av = T(0);
      int na = 0;
      if (live) {
        const int p0 = a_rowptr[row];
        na = a_rowptr[row + 1] - p0;
        if (lt < na && na <= TPR) {
          const int2 dd = adesc[p0 + lt];
          qb = dd.x;
          qe = dd.x + dd.y;
          av = alpha * a_values[p0 + lt];
        }
      }
      const int len = qe - qb;
      int incl = len;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lt >= o)
          incl += t;
      }
      const int ub = __shfl(incl, 63, 64);
      const int dlen = live ? c_rowptr[row + 1] - c_rowptr[row] : -1;
      direct = live && na <= TPR && ub == dlen && ub <= HS / 2;
      // ... and when the whole row is one round of loads (<= 4 * 64 / sub entries in the A row, no B row longer than `sub`:
      // BASELINE cfg5 again) the products never leave the registers: count the buckets (the atomic returns the arrival
      // number inside the bucket), scan, place the KEYS in bucket order, rank each product's key inside its bucket, write.
      if (SPG_INREG && direct && na <= 4 * (TPR >> lsub) && __ballot(len > sub) == 0ull) {
        const int sg = lt >> lsub, sl = lt & (sub - 1), nsg = TPR >> lsub;
        constexpr int U = 4;
        int col[U], bk[U], ai[U];
        T pv[U];
        spg_team_sync<TPR>();  // the bucket counters are zero
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = u * nsg + sg;
          const int src = j < na ? j : 0;
          const int q0 = __shfl(qb, src, 64), q1s = __shfl(qe, src, 64);
          const T a = __shfl(av, src, 64);
          const int q = q0 + sl;
          const bool in = j < na && q < q1s;
          const int qc = in ? q : 0;  // (entry 0 exists: b_has_entries)
          col[u] = in ? b_colind[qc] : -1;
          pv[u] = a * b_values[qc];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          bk[u] = col[u] >= 0 ? (int) __umulhi((unsigned) col[u], bucket_mul) : 0;
          ai[u] = col[u] >= 0 ? atomicAdd(&bcnt[bk[u]], 1) : 0;
        }
        spg_team_sync<TPR>();
        scan_buckets();
        spg_team_sync<TPR>();
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (col[u] >= 0)
            tkeys[bcnt[bk[u]] + ai[u]] = col[u];
        spg_team_sync<TPR>();
        const int out0 = c_rowptr[row];
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (col[u] >= 0) {
            const int b0 = bcnt[bk[u]], b1 = bcnt[bk[u] + 1];
            int rank = b0;
            for (int j = b0; j < b1; ++j)
              rank += tkeys[j] < col[u];
            c_colind[out0 + rank] = col[u];
            c_values[out0 + rank] = pv[u];
          }
        return;
      }
      if (direct) {
        direct_d = ub;
        const int excl = incl - len;
        int* ckeys = list + team * (HS / 2);
        T* cvals = reinterpret_cast<T*>(list + RPB * (HS / 2)) + team * (HS / 2);
        spg_team_sync<TPR>();  // the bucket counters are zero
        const int sg = lt >> lsub, sl = lt & (sub - 1), nsg = TPR >> lsub;
        constexpr int U = 4;
        for (int j0 = 0; j0 < na; j0 += U * nsg) {
          int q0[U], q1[U], col[U], pos[U];
          T a[U], bv[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int j = j0 + u * nsg + sg;
            const int src = j < na ? j : 0;
            q0[u] = __shfl(qb, src, 64);
            const int q1s = __shfl(qe, src, 64);
            q1[u] = j < na ? q1s : q0[u];
            a[u] = __shfl(av, src, 64);
            pos[u] = __shfl(excl, src, 64) + sl;
            const int q = q0[u] + sl;
            const bool in = q < q1[u];
            const int qc = in ? q : (q1[u] > q0[u] ? q0[u] : 0);
            col[u] = in ? b_colind[qc] : -1;
            bv[u] = b_values[qc];
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (col[u] >= 0) {
              ckeys[pos[u]] = col[u];
              cvals[pos[u]] = a[u] * bv[u];
              atomicAdd(&bcnt[(int) __umulhi((unsigned) col[u], bucket_mul)], 1);
            }
            for (int q = q0[u] + sl + sub; q < q1[u]; q += sub) {  // B rows longer than `sub`
              const int key = b_colind[q];
              ckeys[pos[u] + (q - q0[u] - sl)] = key;
              cvals[pos[u] + (q - q0[u] - sl)] = a[u] * b_values[q];
              atomicAdd(&bcnt[(int) __umulhi((unsigned) key, bucket_mul)], 1);
            }
          }
        }
      }
    }
  }
  if (!direct)
    for (int i = lt; i < HS; i += TPR) {
      tkeys[i] = -1;
      if (NUMERIC)
        tvals[i] = T(0);
    }
  spg_team_sync<TPR>();

  if (live && !direct) {
    const int p0 = a_rowptr[row], p1 = a_rowptr[row + 1];
    const int sg = lt >> lsub, sl = lt & (sub - 1), nsg = TPR >> lsub;
    auto insert = [&](int col, T prod) {
      unsigned slot = spg_hash(col, LOG2HS);
      while (true) {
        const int old = atomicCAS(&tkeys[slot], -1, col);
        if (old == -1 || old == col)
          break;
        slot = (slot + 1) & (HS - 1);
      }
      if (NUMERIC)
        spg_lds_add(&tvals[slot], prod);
    };
    if constexpr (TPR <= 64) {
      // The team lives in one wavefront: every lane first fetches ONE entry of the A row (column,
      // B-row bounds, scaled value) so that the three dependent loads a_colind -> b_rowptr ->
      // b_colind are not serialised per B row; the sub-groups then pick entries up by shuffle.
      const int tbase = (threadIdx.x & 63) - lt;  // first lane of the team in its wave
      for (int pc = p0; pc < p1; pc += TPR) {
        int qb = 0, qe = 0;
        T av = T(0);
        if (pc + lt < p1) {
          if (adesc) {
            const int2 dd = adesc[pc + lt];
            qb = dd.x;
            qe = dd.x + dd.y;
          } else {
            const int kk = a_colind[pc + lt];
            qb = b_rowptr ? b_rowptr[kk] : kk;
            qe = b_rowptr ? b_rowptr[kk + 1] : kk + 1;
          }
          if (NUMERIC)
            av = alpha * a_values[pc + lt];
        }
        const int cnt = (p1 - pc) < TPR ? (p1 - pc) : TPR;
        // team-uniform trip count: a lane that left the loop would read as 0 in the shuffles.
        // U rounds are taken together: the first `sub` entries of U * nsg B rows are loaded before
        // the first insert, so the team waits for one memory round trip per U rounds, not per round
        // (the kernel is bound by this dependent-load chain, not by bandwidth or LDS).
        constexpr int U = 4;
        for (int j0 = 0; j0 < cnt; j0 += U * nsg) {
          int q0[U], q1[U], col[U];
          T a[U], bv[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int j = j0 + u * nsg + sg;
            const int src = tbase + (j < cnt ? j : 0);
            q0[u] = __shfl(qb, src);
            const int q1s = __shfl(qe, src);  // unconditional: every lane of the team must take part
            q1[u] = j < cnt ? q1s : q0[u];
            a[u] = NUMERIC ? __shfl(av, src) : T(0);
            const int q = q0[u] + sl;
            const bool in = q < q1[u];
            // clamped, unconditional loads: entry 0 exists whenever B stores anything (b_has_entries)
            const int qc = in ? q : (q1[u] > q0[u] ? q0[u] : 0);
            col[u] = b_rowptr ? (b_has_entries ? b_colind[qc] : 0) : q;
            bv[u] = (NUMERIC && b_rowptr && b_has_entries) ? b_values[qc] : T(1);
            if (!in)
              col[u] = -1;
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (col[u] >= 0)
              insert(col[u], NUMERIC ? (b_rowptr ? a[u] * bv[u] : a[u]) : T(0));
            for (int q = q0[u] + sl + sub; q < q1[u]; q += sub)  // B rows longer than `sub`
              insert(b_rowptr ? b_colind[q] : q, NUMERIC ? (b_rowptr ? a[u] * b_values[q] : a[u]) : T(0));
          }
        }
      }
    } else {
      for (int p = p0 + sg; p < p1; p += nsg) {
        const T av = NUMERIC ? alpha * a_values[p] : T(0);
        int qs, q1;
        if (adesc) {
          const int2 dd = adesc[p];
          qs = dd.x;
          q1 = dd.x + dd.y;
        } else {
          const int kk = a_colind[p];
          qs = b_rowptr ? b_rowptr[kk] : kk;
          q1 = b_rowptr ? b_rowptr[kk + 1] : kk + 1;
        }
        for (int q = qs + sl; q < q1; q += sub)
          insert(b_rowptr ? b_colind[q] : q, NUMERIC ? (b_rowptr ? av * b_values[q] : av) : T(0));
      }
    }
    if (d_rowptr) {  // addend row: + beta * D_i
      const int q1 = d_rowptr[row + 1];
      for (int q = d_rowptr[row] + lt; q < q1; q += TPR)
        insert(d_colind[q], NUMERIC ? beta * d_values[q] : T(0));
    }
  }
  spg_team_sync<TPR>();

  // Occupied slots are counted / compacted in lockstep chunks of TPR slots.  A team of <= 64 lanes
  // lives inside one wavefront, so a ballot gives every lane the team's occupancy mask and the
  // running offset stays in a register; a 256-lane team adds one LDS atomic per wave and chunk.
  const int wl = threadIdx.x & 63;                         // lane in wave
  const int tshift = TPR >= 64 ? 0 : (wl / TPR) * TPR;     // first lane of the team in its wave
  const unsigned long long tmask = TPR >= 64 ? ~0ull : (((1ull << (TPR & 63)) - 1ull) << tshift);
  int* ckeys = NUMERIC ? list + team * (HS / 2) : nullptr;  // compacted keys
  T* cvals = NUMERIC ? reinterpret_cast<T*>(list + RPB * (HS / 2)) + team * (HS / 2) : nullptr;
  int running = direct ? direct_d : 0;
  for (int i0 = 0; i0 < (direct ? 0 : HS); i0 += TPR) {
    const int i = i0 + lt;
    const int key = tkeys[i];
    const bool occ = live && key != -1;
    const unsigned long long mask = __ballot(occ) & tmask;
    int base = running;
    if (TPR > 64) {
      int wbase = 0;
      if (wl == 0 && mask)
        wbase = atomicAdd(&cnt[team], (int) __popcll(mask));
      base = __shfl(wbase, 0);
    }
    if (NUMERIC && occ) {
      const int pos = base + (int) __popcll(mask & ((1ull << wl) - 1ull));
      ckeys[pos] = key;
      cvals[pos] = tvals[i];
      atomicAdd(&bcnt[(int) __umulhi((unsigned) key, bucket_mul)], 1);  // bucket sizes for the sort below
    }
    running += (int) __popcll(mask);
  }
  if (TPR > 64)
    spg_team_sync<TPR>();
  const int d = TPR > 64 ? cnt[team] : running;
  if (!NUMERIC) {
    if (live && lt == 0)
      c_rowptr[row] = d;
    return;
  } else {
    // Sort the d unique keys: bucket them by column range (monotone, NBK buckets, LDS integer
    // atomics are fast), then rank each key inside its bucket only.  Uniform columns give buckets
    // of d/NBK keys; the worst case (one bucket) degrades to the plain O(d^2) rank sort.
    spg_team_sync<TPR>();  // the bucket counts of the compaction pass are complete
    if (live)
      scan_buckets();
    spg_team_sync<TPR>();
    int* skeys = tkeys;  // the hash table is dead after compaction: reuse it for the bucketed copy
    T* svals = tvals;
    if (live)
      for (int e = lt; e < d; e += TPR) {
        const int key = ckeys[e];
        const int bk = (int) __umulhi((unsigned) key, bucket_mul);
        const int pos = bcnt[bk] + atomicAdd(&bfill[bk], 1);
        skeys[pos] = key;
        svals[pos] = cvals[e];
      }
    spg_team_sync<TPR>();
    // the sorted row is put together in the (dead) compacted list and leaves in consecutive elements: lanes storing
    // 4 bytes each at their keys' ranks are one access per lane in the vector-memory address unit, which is what bounded
    // the numeric pass (TA_BUSY 80 - 93 %, profiles/r04_spgemm_direct.md)
    if (live)
      for (int e = lt; e < d; e += TPR) {
        const int key = skeys[e];
        const int bk = (int) __umulhi((unsigned) key, bucket_mul);
        const int b0 = bcnt[bk], b1 = bcnt[bk + 1];
        int rank = b0;
        for (int j = b0; j < b1; ++j)
          rank += skeys[j] < key;
        ckeys[rank] = key;
        cvals[rank] = svals[e];
      }
    spg_team_sync<TPR>();
    if (live) {
      const int out0 = c_rowptr[row];
      for (int e = lt; e < d; e += TPR) {
        c_colind[out0 + e] = ckeys[e];
        c_values[out0 + e] = cvals[e];
      }
    }
  }
}

// bin 4: dense accumulator per workgroup.  bits: ceil(n/32) words (all zero on
// entry and on exit), vals: n values (all zero on entry and exit).  All accesses to
// the workspace are agent-scope atomics so nothing stale is read out of the CU's L1.
template <typename T, bool NUMERIC>
__global__ __launch_bounds__(256) void spg_dense_kernel(
    int64_t count, const int32_t* __restrict__ perm, int64_t n, const int32_t* __restrict__ a_rowptr,
    const int32_t* __restrict__ a_colind, const T* __restrict__ a_values, const int32_t* __restrict__ b_rowptr,
    const int32_t* __restrict__ b_colind, const T* __restrict__ b_values, int32_t* __restrict__ c_rowptr,
    int32_t* __restrict__ c_colind, T* __restrict__ c_values, T alpha, uint32_t* __restrict__ bits_all,
    T* __restrict__ vals_all, const int32_t* __restrict__ d_rowptr, const int32_t* __restrict__ d_colind,
    const T* __restrict__ d_values, T beta) {
  __shared__ int scan[256];
  __shared__ int running;
  const int64_t nwords = (n + 31) / 32;
  uint32_t* bits = bits_all + (int64_t) blockIdx.x * nwords;
  T* vals = NUMERIC ? vals_all + (int64_t) blockIdx.x * n : nullptr;
  const int tid = threadIdx.x;
  const int wave = tid / 64, lane = tid % 64;

  for (int64_t idx = blockIdx.x; idx < count; idx += gridDim.x) {
    const int row = perm[idx];
    const int p0 = a_rowptr[row], p1 = a_rowptr[row + 1];
    // one wavefront per A entry, lanes across the B row
    for (int p = p0 + wave; p < p1; p += 4) {
      const int kk = a_colind[p];
      T av = T(0);
      if (NUMERIC)
        av = alpha * a_values[p];
      const int q1 = b_rowptr ? b_rowptr[kk + 1] : kk + 1;
      for (int q = (b_rowptr ? b_rowptr[kk] : kk) + lane; q < q1; q += 64) {
        const int col = b_rowptr ? b_colind[q] : q;
        atomicOr(&bits[col >> 5], 1u << (col & 31));
        if (NUMERIC)
          unsafeAtomicAdd(&vals[col], b_rowptr ? av * b_values[q] : av);
      }
    }
    if (d_rowptr) {  // addend row: + beta * D_i
      const int q1 = d_rowptr[row + 1];
      for (int q = d_rowptr[row] + tid; q < q1; q += 256) {
        const int col = d_colind[q];
        atomicOr(&bits[col >> 5], 1u << (col & 31));
        if (NUMERIC)
          unsafeAtomicAdd(&vals[col], beta * d_values[q]);
      }
    }
    if (tid == 0)
      running = 0;
    __syncthreads();  // all atomics of this workgroup are complete (performed at L2)
    const int out0 = NUMERIC ? c_rowptr[row] : 0;
    for (int64_t w0 = 0; w0 < nwords; w0 += 256) {
      const int64_t w = w0 + tid;
      uint32_t word = 0;
      if (w < nwords) {
        word = __hip_atomic_load(&bits[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (word)
          __hip_atomic_store(&bits[w], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const int pc = __popc(word);
      // workgroup exclusive scan of pc
      scan[tid] = pc;
      __syncthreads();
      for (int o = 1; o < 256; o <<= 1) {
        const int t = tid >= o ? scan[tid - o] : 0;
        __syncthreads();
        scan[tid] += t;
        __syncthreads();
      }
      const int excl = scan[tid] - pc;
      const int base = running;
      if (NUMERIC) {
        int o = out0 + base + excl;
        while (word) {
          const int b = __ffs((int) word) - 1;
          word &= word - 1;
          const int64_t col = w * 32 + b;
          c_colind[o] = (int32_t) col;
          c_values[o] = __hip_atomic_load(&vals[col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&vals[col], T(0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ++o;
        }
      }
      __syncthreads();
      if (tid == 255)
        running = base + scan[255];
      __syncthreads();
    }
    if (!NUMERIC && tid == 0)
      c_rowptr[row] = running;
    __syncthreads();
  }
}

// products per row, the addend's entries included (0 for rows outside the LDS-hash bins 1-2: they keep the hash
// path), scanned into r_pbase
__global__ __launch_bounds__(256) void spg_products_kernel(int64_t m, const int32_t* __restrict__ a_rowptr,
                                                           const int32_t* __restrict__ a_colind,
                                                           const int32_t* __restrict__ b_rowptr,
                                                           const int2* __restrict__ adesc,
                                                           const int32_t* __restrict__ d_rowptr,
                                                           int32_t* __restrict__ prod, int32_t* __restrict__ prod3) {
  const int64_t row = (int64_t) blockIdx.x * 32 + threadIdx.x / 8;
  const int lane = threadIdx.x % 8;
  int64_t ub = 0;
  if (row < m)
    for (int p = a_rowptr[row] + lane; p < a_rowptr[row + 1]; p += 8) {
      if (adesc) {
        ub += adesc[p].y;
        continue;
      }
      const int kk = a_colind[p];
      ub += b_rowptr ? b_rowptr[kk + 1] - b_rowptr[kk] : 1;  // no B: identity (add())
    }
  ub = group_sum_c<8>(ub);
  if (row < m && d_rowptr)
    ub += d_rowptr[row + 1] - d_rowptr[row];  // the addend's entries follow the products of the row
  if (row < m && lane == 0) {
    prod[row] = ub <= 256 ? (int32_t) ub : 0;
    prod3[row] = ub > 256 && ub <= 1024 ? (int32_t) ub : 0;
  }
}

// Numeric reuse, recording pass: the rank of every product's column in its (sorted) output row -- a binary search in
// the row's columns, held in LDS.  The products of a row are enumerated in a fixed order that the fills by rank
// (spg_ranked_fill_kernel) repeat: A entries in storage order, the entries of each B row in storage order; product
// index = r_pbase[row] + (entries of the earlier B rows) + offset.  Rows of bins 1-2 have <= 256 products, hence
// (bin 3: <= 1024 products, RT = uint16_t)
// <= 256 output entries: a rank fits one byte.
// Teams of TPR <= 64 lanes inside one wavefront, SUB lanes per B row, as in spg_hash_kernel.
template <int TPR, int CAP, typename RT>
__global__ __launch_bounds__(256) void spg_rank_record_kernel(
    int64_t count, const int32_t* __restrict__ perm, const int32_t* __restrict__ a_rowptr,
    const int32_t* __restrict__ a_colind, const int32_t* __restrict__ b_rowptr, const int2* __restrict__ adesc,
    const int32_t* __restrict__ b_colind, const int32_t* __restrict__ c_rowptr,
    const int32_t* __restrict__ cols_sorted, int sub, const int32_t* __restrict__ pbase,
    RT* __restrict__ prank, const int32_t* __restrict__ d_rowptr, const int32_t* __restrict__ d_colind) {
  // b_rowptr == nullptr: B is the identity (add()): every A entry is one product on its own column.
  // d_rowptr != nullptr: the entries of the addend's row are enumerated after the products.
  constexpr int RPB = 256 / TPR;
  __shared__ int s_keys[RPB * CAP];
  const int team = threadIdx.x / TPR, lt = threadIdx.x % TPR;
  const int64_t idx = (int64_t) blockIdx.x * RPB + team;
  const bool live = idx < count;
  const int row = live ? perm[idx] : 0;
  const int out0 = live ? c_rowptr[row] : 0;
  const int d = live ? c_rowptr[row + 1] - out0 : 0;
  int* tkeys = s_keys + team * CAP;
  for (int i = lt; i < d; i += TPR)
    tkeys[i] = cols_sorted[out0 + i];
  spg_team_sync<TPR>();
  if (!live)
    return;
  const int p0 = a_rowptr[row], p1 = a_rowptr[row + 1];
  const int lsub = __builtin_ctz((unsigned) sub);  // power of two
  const int sg = lt >> lsub, sl = lt & (sub - 1), nsg = TPR >> lsub;
  const int tbase = (threadIdx.x & 63) - lt;  // first lane of the team in its wave
  auto rank_of = [&](int col) {
    int lo = 0, hi = d;  // last position with tkeys[pos] <= col (the column is present by construction)
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tkeys[mid] <= col)
        lo = mid;
      else
        hi = mid;
    }
    return lo;
  };
  int running = pbase[row];
  for (int pc = p0; pc < p1; pc += TPR) {
    int qb = 0, len = 0;
    if (pc + lt < p1) {
      if (adesc) {
        const int2 dd = adesc[pc + lt];
        qb = dd.x;
        len = dd.y;
      } else {
        const int kk = a_colind[pc + lt];
        qb = b_rowptr ? b_rowptr[kk] : kk;
        len = b_rowptr ? b_rowptr[kk + 1] - qb : 1;
      }
    }
    // exclusive scan of the B-row lengths over the team's lanes (team-uniform trip count)
    int incl = len;
    for (int o = 1; o < TPR; o <<= 1) {
      const int t = __shfl_up(incl, o, TPR);
      if (lt >= o)
        incl += t;
    }
    const int excl = incl - len;
    const int total = __shfl(incl, tbase + TPR - 1);
    const int cnt = (p1 - pc) < TPR ? (p1 - pc) : TPR;
    for (int j0 = 0; j0 < cnt; j0 += nsg) {
      const int j = j0 + sg;
      const int src = tbase + (j < cnt ? j : 0);
      const int q0 = __shfl(qb, src);
      const int ln = __shfl(len, src);
      const int off = __shfl(excl, src);
      if (j < cnt)
        for (int q = sl; q < ln; q += sub)
          prank[running + off + q] = (RT) rank_of(b_rowptr ? b_colind[q0 + q] : q0 + q);
    }
    running += total;
  }
  if (d_rowptr) {
    const int q0 = d_rowptr[row], q1 = d_rowptr[row + 1];
    for (int q = q0 + lt; q < q1; q += TPR)
      prank[running + (q - q0)] = (RT) rank_of(d_colind[q]);
  }
}

__global__ __launch_bounds__(256) void spg_adesc_kernel(int64_t a_nnz, const int32_t* __restrict__ a_colind,
                                                        const int32_t* __restrict__ b_rowptr,
                                                        int2* __restrict__ adesc) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i < a_nnz) {
    const int kk = a_colind[i];
    struct __attribute__((packed, aligned(4))) pair_t {
      int lo, hi;
    };
    const pair_t r = *reinterpret_cast<const pair_t*>(b_rowptr + kk);  // one 8-byte load, 4-byte aligned
    adesc[i] = make_int2(r.lo, r.hi - r.lo);
  }
}
// Numeric pass by rank: vals[rank] += a * b in LDS, then one coalesced write of the row's values (and columns).
//  * U entries of the A row per sub-group are taken up together, so that the rank and B-value loads of U B rows are in
//    flight before the first LDS update;
//  * a row whose product count equals its number of output entries has no two products on the same column: every
//    product then owns its slot, so the slots are written with plain LDS stores -- no zero fill, no read-modify-write
//    (cfg5: 96.6 % of the rows).  Rows with repeated columns accumulate as before.
// Same product enumeration as the recording kernel: r_rank does not depend on TPR, SUB or U.
template <typename T, int TPR, int CAP, int U, bool GENERAL, typename RT>
__global__ __launch_bounds__(256) void spg_ranked_fill_kernel(
    int64_t count, const int32_t* __restrict__ perm, const int32_t* __restrict__ a_rowptr,
    const int32_t* __restrict__ a_colind, const T* __restrict__ a_values, const int32_t* __restrict__ b_rowptr,
    const int2* __restrict__ adesc, const T* __restrict__ b_values, const int32_t* __restrict__ c_rowptr,
    const int32_t* __restrict__ cols_sorted, int32_t* __restrict__ c_colind, T* __restrict__ c_values, T alpha,
    int sub, const int32_t* __restrict__ pbase, const RT* __restrict__ prank, int copy_cols,
    const int32_t* __restrict__ d_rowptr, const T* __restrict__ d_values, T beta) {
  // GENERAL: b_rowptr == nullptr means B is the identity (add()), d_rowptr != nullptr adds beta * (row of the addend).
  // The plain three-argument product has its own instantiation: the extra tests in the load batch cost it 10 %.
  // adesc != nullptr: (start, length) in b_values of the B row of every A entry (a_colind / b_rowptr are not read)
  constexpr int RPB = 256 / TPR;
  __shared__ T s_vals[RPB * CAP];
  const int team = threadIdx.x / TPR, lt = threadIdx.x % TPR;
  const int64_t idx = (int64_t) blockIdx.x * RPB + team;
  const bool live = idx < count;
  const int row = live ? perm[idx] : 0;
  const int out0 = live ? c_rowptr[row] : 0;
  const int d = live ? c_rowptr[row + 1] - out0 : 0;
  int running = live ? pbase[row] : 0;
  const bool plain = live && pbase[row + 1] - running == d;
  T* tvals = s_vals + team * CAP;
  if (!plain)
    for (int i = lt; i < d; i += TPR)
      tvals[i] = T(0);
  spg_team_sync<TPR>();
  if (live) {
    const int p0 = a_rowptr[row], p1 = a_rowptr[row + 1];
    const int lsub = __builtin_ctz((unsigned) sub);  // power of two
    const int sg = lt >> lsub, sl = lt & (sub - 1), nsg = TPR >> lsub;
    const int tbase = (threadIdx.x & 63) - lt;  // first lane of the team in its wave
    for (int pc = p0; pc < p1; pc += TPR) {
      int qb = 0, len = 0;
      T av = T(0);
      if (pc + lt < p1) {
        if (adesc) {
          const int2 dd = adesc[pc + lt];
          qb = dd.x;
          len = dd.y;
        } else {
          const int kk = a_colind[pc + lt];
          qb = (!GENERAL || b_rowptr) ? b_rowptr[kk] : kk;
          len = (!GENERAL || b_rowptr) ? b_rowptr[kk + 1] - qb : 1;
        }
        av = alpha * a_values[pc + lt];
      }
      int incl = len;
      for (int o = 1; o < TPR; o <<= 1) {
        const int t = __shfl_up(incl, o, TPR);
        if (lt >= o)
          incl += t;
      }
      const int excl = incl - len;
      const int total = __shfl(incl, tbase + TPR - 1);
      const int cnt = (p1 - pc) < TPR ? (p1 - pc) : TPR;
      for (int j0 = 0; j0 < cnt; j0 += U * nsg) {
        int q0[U], ln[U], pb[U], rk[U];
        T a[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = j0 + u * nsg + sg;
          const int src = tbase + (j < cnt ? j : 0);
          q0[u] = __shfl(qb, src);
          ln[u] = __shfl(len, src);
          pb[u] = running + __shfl(excl, src);
          a[u] = __shfl(av, src);
          if (j >= cnt)
            ln[u] = 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          rk[u] = 0;
          bv[u] = T(0);
          if (sl < ln[u]) {
            rk[u] = prank[pb[u] + sl];
            bv[u] = (!GENERAL || b_rowptr) ? b_values[q0[u] + sl] : T(1);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (sl < ln[u]) {
            if (plain)
              tvals[rk[u]] = a[u] * bv[u];
            else
              spg_lds_add(&tvals[rk[u]], a[u] * bv[u]);
          }
        // B rows longer than the sub-group: the rest of their entries, one round at a time
#pragma unroll
        for (int u = 0; u < U; ++u)
          for (int q = sl + sub; q < ln[u]; q += sub) {  // (never with the identity: its rows have one entry)
            const T v = a[u] * b_values[q0[u] + q];
            const int r = prank[pb[u] + q];
            if (plain)
              tvals[r] = v;
            else
              spg_lds_add(&tvals[r], v);
          }
      }
      running += total;
    }
    if (GENERAL && d_rowptr) {
      const int q0 = d_rowptr[row], q1 = d_rowptr[row + 1];
      for (int q = q0 + lt; q < q1; q += TPR) {
        const T v = beta * d_values[q];
        const int r = prank[running + (q - q0)];
        if (plain)
          tvals[r] = v;
        else
          spg_lds_add(&tvals[r], v);
      }
    }
  }
  spg_team_sync<TPR>();
  for (int i = lt; i < d; i += TPR) {
    c_values[out0 + i] = tvals[i];
    if (copy_cols)
      c_colind[out0 + i] = cols_sorted[out0 + i];
  }
}

__global__ __launch_bounds__(256) void spg_zero_rows_kernel(int64_t count, const int32_t* __restrict__ perm,
                                                            int32_t* __restrict__ c_rowptr) {
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i < count)
    c_rowptr[perm[i]] = 0;
}

template <typename T, int LOG2HS, int TPR, bool NUMERIC>
static size_t hash_smem_bytes() {
  constexpr int HS = 1 << LOG2HS;
  constexpr int RPB = 256 / TPR;
  size_t b = (size_t) RPB * HS * 4 + (size_t) ((RPB + 3) & ~3) * 4;
  if (NUMERIC)
    b += (size_t) RPB * HS * sizeof(T) + (size_t) RPB * (HS / 2) * (4 + sizeof(T)) + (size_t) RPB * (2 * ((NUMERIC && TPR == 64) ? SPG_NBK64 : (TPR < 64 ? TPR : 64)) + 2) * 4;
  else
    b += 16;
  return b;
}

// ---- direct rows (round 4) ------------------------------------------------------------------------------------------
// A row of C whose product count equals its structural length has no two products in the same column: there is nothing
// to accumulate, only to sort.  When, in addition, the A row is one round of loads (<= 4 * 64 / sub entries, no B row
// longer than `sub`: BASELINE cfg5), a wavefront holds the row's <= 256 products in registers.  The symbolic pass lists
// such rows (spg_direct_classify_kernel); spg_direct_kernel is a PERSISTENT kernel over that list: the hash kernel above
// was bound by its chain of dependent loads -- perm -> row offsets -> A entries -> B entries, one row per wavefront,
// 67 % of the wave cycles waiting (SQ counters, profiles/r04_spgemm_direct.md) -- so every wavefront here keeps three rows
// in flight: while row i is sorted it has already issued the B loads of row i + 1, the A loads of row i + 2 and the
// descriptor load of row i + 3.
// descriptors of the sortable rows for spg_direct_kernel, (first A entry, A entries, first output position, output
// length): flag[] of spg_sort_symbolic_kernel scanned -- the rows whose products all have a column of their own from the
// front of desc[] in list order, the rows with shared columns from its end
__global__ __launch_bounds__(256) void spg_direct_lists_kernel(int64_t count, const int32_t* __restrict__ rows,
                                                               const int32_t* __restrict__ a_rowptr,
                                                               const int32_t* __restrict__ c_rowptr,
                                                               const int32_t* __restrict__ flag_excl,
                                                               int4* __restrict__ desc,
                                                               const int32_t* __restrict__ d_rowptr,
                                                               int2* __restrict__ ddesc) {
  const int64_t idx = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= count)
    return;
  const int row = rows[idx];
  const int p0 = a_rowptr[row], out0 = c_rowptr[row], pos = flag_excl[idx];
  const int4 d = make_int4(p0, a_rowptr[row + 1] - p0, out0, c_rowptr[row + 1] - out0);
  const int64_t at = flag_excl[idx + 1] != pos ? (int64_t) pos : count - 1 - (idx - pos);
  desc[at] = d;
  if (ddesc)
    ddesc[at] = make_int2(d_rowptr[row], d_rowptr[row + 1] - d_rowptr[row]);
}

// Inclusive scan of one int per lane over the wavefront with DPP moves (row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes,
// row_bcast:15 / :31 across them; lanes without a source read 0): six vector instructions with no LDS round trip -- the
// shuffle form is six DEPENDENT ds_bpermute, a third of the LDS latency chain of a row in the kernels below.
template <int CTRL, int ROWS>
__device__ __forceinline__ int spg_dpp(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWS, 0xf, true);
}
__device__ __forceinline__ int spg_wave_incl_scan(int v) {
  v += spg_dpp<0x111, 0xf>(v);
  v += spg_dpp<0x112, 0xf>(v);
  v += spg_dpp<0x114, 0xf>(v);
  v += spg_dpp<0x118, 0xf>(v);
  v += spg_dpp<0x142, 0xa>(v);
  v += spg_dpp<0x143, 0xc>(v);
  return v;
}

// four consecutive elements as one 16-byte (fp64 values: 32-byte) access at 4-byte (8-byte) alignment: global memory takes
// unaligned vector accesses, and the vector-memory address unit spends its 16 cycles per INSTRUCTION, however wide
typedef int spg_i4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float spg_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef double spg_d4u __attribute__((ext_vector_type(4), aligned(8)));
template <typename T> struct spg_vec4;
template <> struct spg_vec4<float> { typedef spg_f4u type; };
template <> struct spg_vec4<double> { typedef spg_d4u type; };

// (column, value) pairs of B in one array: what limits spg_direct_kernel is the number of distinct 128-byte lines the
// memory system can fetch at random (~32 G/s measured, whether the 16 M gathers of cfg5 ask for one line each -- the symbolic
// pass, 0.56 ms -- or two -- columns and values of the numeric pass, 1.0 ms); a B row of 16 entries is ONE line here.
__global__ __launch_bounds__(256) void spg_pack_b_kernel(int64_t nnz, const int32_t* __restrict__ col,
                                                         const float* __restrict__ val, int2* __restrict__ out) {
  const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nnz)
    out[i] = make_int2(col[i], __float_as_int(val[i]));
}

// ADD (round 6, C = alpha A B + beta D): the row of the addend joins the products as a FIFTH element per lane (<= 64 entries:
// the classification; enumerated after the products, as the reference adds it after them), 320 slots instead of 256.
template <typename T, bool PACKED, bool DUP, bool ADD = false>
__global__ __launch_bounds__(256) void spg_direct_kernel(
    int n_dir, const int4* __restrict__ desc, const int2* __restrict__ adesc, const T* __restrict__ a_values,
    const int32_t* __restrict__ b_colind, const T* __restrict__ b_values, int32_t* __restrict__ c_colind,
    T* __restrict__ c_values, T alpha, int sub, unsigned bucket_mul, const int2* __restrict__ b_pack,
    const int2* __restrict__ ddesc = nullptr, const int32_t* __restrict__ d_colind = nullptr,
    const T* __restrict__ d_values = nullptr, T beta = T(0)) {
  constexpr int NBK = SPG_DIR_NBK, BPL = NBK / 64;
  constexpr int CAP = ADD ? 320 : 256;
  typedef typename spg_vec4<T>::type v4u;
  typedef typename std::conditional<ADD, unsigned short, unsigned char>::type enum_t;
  __shared__ __attribute__((aligned(16))) int s_keys[4][CAP + 4];
  __shared__ __attribute__((aligned(16))) T s_vals[4][CAP];
  __shared__ __attribute__((aligned(16))) int s_bcnt[4][NBK + 4];
  __shared__ enum_t s_enum[4][DUP ? CAP + 4 : 4];  // DUP: the enumeration number of the product behind every key
  const int wave = threadIdx.x >> 6, lt = threadIdx.x & 63;
  int* tkeys = s_keys[wave];
  enum_t* tenum = s_enum[wave];
  T* tvals = s_vals[wave];
  int* bcnt = s_bcnt[wave];
  // sub / 4 lanes per B row, four consecutive entries each: lane lt works on entries c4 .. c4 + 3 of the B row of A entry jr
  const int lshift = __builtin_ctz((unsigned) sub) - 2;  // sub is 4, 8 or 16
  const int jr = lt >> lshift, c4 = (lt & ((1 << lshift) - 1)) * 4;
  const int stride = (int) gridDim.x * 4;
  int i = __builtin_amdgcn_readfirstlane((int) blockIdx.x * 4 + wave);
  constexpr int U = ADD ? 5 : 4;
  struct arow {
    int qb, len;
    T a;
  };
  struct brow {  // four consecutive entries of a B row, loaded as vectors whether or not the row has that many left:
    spg_i4u cc;  // the classification admits a row only if reading up to the next multiple of four entries of each of its
    v4u vv;      // B rows stays inside B's arrays (spg_direct_flag_kernel); what lies beyond the row is masked out by n
    int n;       // entries of the lane: 0 .. 4
    int dc;      // ADD: the lane's entry of the addend's row: column (-1: none) and beta * value
    T dv;
  };
  auto load_ddesc = [&](int r) { return (ADD && r < n_dir) ? ddesc[r] : make_int2(0, 0); };
  auto load_desc = [&](int r) { return r < n_dir ? desc[r] : make_int4(0, 0, 0, 0); };
  auto load_a = [&](const int4& d) {
    arow a{0, 0, T(0)};
    if (lt < (d.y & 0xFFFF)) {
      const int2 dd = adesc[d.x + lt];
      a.qb = dd.x;
      a.len = dd.y;
      a.a = a_values[d.x + lt];
    }
    return a;
  };
  auto load_b = [&](const arow& a, int na_f, const int2& dd) {
    const int na = na_f & 0xFFFF;
    brow b;
    b.dc = -1;
    b.dv = T(0);
    if constexpr (ADD)
      if (lt < dd.y) {
        b.dc = d_colind[dd.x + lt];
        b.dv = beta * d_values[dd.x + lt];
      }
    const int q0 = __shfl(a.qb, jr, 64) + c4, ln = __shfl(a.len, jr, 64) - c4;  // (lanes >= na hold length 0)
    b.n = jr < na ? (ln < 0 ? 0 : ln > 4 ? 4 : ln) : 0;
    const int q = b.n > 0 ? q0 : 0;  // (entries 0 .. 3 exist: some admitted row has a B row whose padded range is inside)
    if constexpr (PACKED && sizeof(T) == 4) {
      const spg_i4u lo = *reinterpret_cast<const spg_i4u*>(b_pack + q), hi = *reinterpret_cast<const spg_i4u*>(b_pack + q + 2);
      b.cc = spg_i4u{lo[0], lo[2], hi[0], hi[2]};
      b.vv = v4u{__int_as_float(lo[1]), __int_as_float(lo[3]), __int_as_float(hi[1]), __int_as_float(hi[3])};
    } else {
      b.cc = *reinterpret_cast<const spg_i4u*>(b_colind + q);  // (non-temporal gathers were measured: 1.55 against 1.11 ms)
      b.vv = *reinterpret_cast<const v4u*>(b_values + q);
    }
    return b;
  };
  int4 d0 = load_desc(i), d1 = load_desc(i + stride), d2 = load_desc(i + 2 * stride);
  int2 e0d = load_ddesc(i), e1d = load_ddesc(i + stride), e2d = load_ddesc(i + 2 * stride);
  arow a0 = load_a(d0), a1 = load_a(d1);
  brow b0 = load_b(a0, d0.y, e0d);
  auto zero_buckets = [&]() {
#pragma unroll
    for (int j = 0; j < BPL; ++j)
      bcnt[lt * BPL + j] = 0;
    if (lt == 0)
      bcnt[NBK] = 0;
  };
  zero_buckets();
  spg_team_sync<64>();
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the loop is entered with every load complete, like its back edge
  while (i < n_dir) {
    // the next rows' loads first: B entries of row i + 1, A entries of row i + 2, descriptor of row i + 3
    const brow b1 = load_b(a1, d1.y, e1d);
    const arow a2 = load_a(d2);
    const int4 d3 = load_desc(i + 3 * stride);
    const int2 e3d = load_ddesc(i + 3 * stride);
    // row i: products, bucket counts (the atomic returns the arrival number inside the bucket), scan, keys in bucket
    // order, rank of every product's key inside its bucket, sorted row in LDS, write
    const int out0 = d0.z, dlen = d0.w;
    constexpr bool dup = DUP;  // some products of the row share a column: ties in the sort, sums in the output (the rows
                               // of that kind have a launch of their own: the extra registers cost the others a wavefront per SIMD)
    const T av = __shfl(alpha * a0.a, jr, 64);
    int col[U], bk[U], ai[U];
    T pv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      col[u] = u < 4 ? (u < b0.n ? b0.cc[u & 3] : -1) : b0.dc;
      pv[u] = u < 4 ? av * b0.vv[u & 3] : b0.dv;
      bk[u] = col[u] >= 0 ? (int) __umulhi((unsigned) col[u], bucket_mul) : 0;
      ai[u] = col[u] >= 0 ? atomicAdd(&bcnt[bk[u]], 1) : 0;
    }
    spg_team_sync<64>();
    {
      int c[BPL], tot = 0;
#pragma unroll
      for (int j = 0; j < BPL; ++j) {
        c[j] = bcnt[lt * BPL + j];
        tot += c[j];
      }
      const int incl = spg_wave_incl_scan(tot);
      int run = incl - tot;
#pragma unroll
      for (int j = 0; j < BPL; ++j) {
        bcnt[lt * BPL + j] = run;
        run += c[j];
      }
      if (lt == 63)
        bcnt[NBK] = incl;
    }
    spg_team_sync<64>();
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (col[u] >= 0) {
        tkeys[bcnt[bk[u]] + ai[u]] = col[u];
        if constexpr (dup)  // lane lt, entry u = product number 4 lt + u of the reference's enumeration (A entry, then B entry)
          tenum[bcnt[bk[u]] + ai[u]] = (enum_t) (u < 4 ? 4 * lt + u : 256 + lt);
      }
    spg_team_sync<64>();
    int rank[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      rank[u] = 0;
      if (col[u] >= 0) {
        const int e0 = bcnt[bk[u]], e1 = bcnt[bk[u] + 1];
        rank[u] = e0;
#if SPG_DIR_READ2
        if (dup) {  // equal keys in the order the reference enumerates the products: the run sums below then add in its order
          const int me = u < 4 ? 4 * lt + u : 256 + lt;
          for (int j = e0; j < e1; ++j) {
            const int k0 = tkeys[j];
            rank[u] += (int) (k0 < col[u]) + (int) ((k0 == col[u]) & ((int) tenum[j] < me));
          }
        } else
        for (int j = e0; j < e1; j += 2) {  // (a key past the bucket's end is read -- the array has the slack -- and not counted)
          const int k0 = tkeys[j], k1 = tkeys[j + 1];
          rank[u] += (int) (k0 < col[u]) + (int) ((j + 1 < e1) & (k1 < col[u]));
        }
#else
        for (int j = e0; j < e1; ++j)
          rank[u] += tkeys[j] < col[u];
#endif
      }
    }
    spg_team_sync<64>();
    int n_prod = dlen;  // products of the row (the scan's total)
    if constexpr (dup) {
      n_prod = bcnt[NBK];
      spg_team_sync<64>();
    }
    zero_buckets();
    // The sorted row is put together in LDS and leaves as whole lines.  Lanes storing 4 bytes each at their products'
    // ranks were 495 separate accesses per row in the vector-memory address unit (TCP_TOTAL_WRITE; TA_BUSY 80 - 93 %:
    // what bounded the kernel); four consecutive elements per lane are two store instructions per row.
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (col[u] >= 0) {
        tkeys[rank[u]] = col[u];
        tvals[rank[u]] = pv[u];
      }
    spg_team_sync<64>();
    if constexpr (dup) {
      // equal columns are neighbours now, in enumeration order: the first of a run takes the run's sum (added in that order:
      // the reference's sequence of += on one accumulator, multiply_impl / spgemm_gustavsons.hpp:30-41) and moves
      // to position (number of runs before it); every lane works its four elements out in registers before anything is
      // written back
      const int n = n_prod, e = U * lt;  // (U consecutive elements per lane: 4, or 5 with an addend)
      int k[U], o[U];
      T sv[U];
      bool head[U];
      const int prev = e > 0 && e <= n ? tkeys[e - 1] : -1;
      int heads = 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        k[u] = e + u < n ? tkeys[e + u] : -1;
        head[u] = e + u < n && k[u] != (u == 0 ? prev : k[u - 1]);
        heads += (int) head[u];
      }
      int before = spg_wave_incl_scan(heads) - heads;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        o[u] = before;
        sv[u] = T(0);
        if (head[u]) {
          sv[u] = tvals[e + u];
          for (int j = e + u + 1; j < n && tkeys[j] == k[u]; ++j)
            sv[u] += tvals[j];
          ++before;
        }
      }
      spg_team_sync<64>();
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (head[u]) {
          tkeys[o[u]] = k[u];
          tvals[o[u]] = sv[u];
        }
      spg_team_sync<64>();
    }
    // the loads issued at the top have had the whole sort to arrive: wait for them HERE, before this row's stores are
    // issued (vmcnt counts in order: a wait placed after the stores -- where the register rotation below would put it --
    // would wait for the stores as well)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt and lgkmcnt untouched
    {
      const int e = 4 * lt, n_out = dlen - e;
      if (n_out >= 4) {
        *reinterpret_cast<spg_i4u*>(c_colind + out0 + e) = *reinterpret_cast<const spg_i4u*>(tkeys + e);
        *reinterpret_cast<v4u*>(c_values + out0 + e) = *reinterpret_cast<const v4u*>(tvals + e);
      } else {
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
          if (u < n_out) {
            c_colind[out0 + e + u] = tkeys[e + u];
            c_values[out0 + e + u] = tvals[e + u];
          }
      }
      if constexpr (ADD)  // elements 256 .. 319: one per lane
        if (256 + lt < dlen) {
          c_colind[out0 + 256 + lt] = tkeys[256 + lt];
          c_values[out0 + 256 + lt] = tvals[256 + lt];
        }
    }
    spg_team_sync<64>();  // (the next row's keys go into the same array)
    b0 = b1;
    a0 = a1;
    a1 = a2;
    d0 = d1;
    d1 = d2;
    d2 = d3;
    e0d = e1d;
    e1d = e2d;
    e2d = e3d;
    i += stride;
  }
}

// Symbolic pass over the sortable rows (the same mapping and pipeline as spg_direct_kernel, keys only): the distinct
// columns of a row are its products minus the keys that have an equal key at a lower position of their bucket.
// row_nnz[row] = distinct columns; flag[i] = 1 when every product has a column of its own (the row is direct).  The LDS
// hash of spg_hash_kernel<.., false> was VALU-bound on these rows (308 vector instructions per row, 90 % busy).
// ADD (round 6): the row of an addend joins the products, one entry per lane (<= 64: the classification), as a fifth key.
template <bool ADD>
__global__ __launch_bounds__(256) void spg_sort_symbolic_kernel(int n_rows, const int32_t* __restrict__ rows,
                                                                const int32_t* __restrict__ a_rowptr,
                                                                const int2* __restrict__ adesc,
                                                                const int32_t* __restrict__ b_colind,
                                                                int32_t* __restrict__ row_nnz, int32_t* __restrict__ flag,
                                                                int sub, unsigned bucket_mul,
                                                                const int32_t* __restrict__ d_rowptr,
                                                                const int32_t* __restrict__ d_colind) {
  constexpr int NBK = SPG_DIR_NBK, BPL = NBK / 64;
  constexpr int CAP = ADD ? 320 : 256;
  __shared__ __attribute__((aligned(16))) int s_keys[4][CAP + 4];
  __shared__ __attribute__((aligned(16))) int s_bcnt[4][NBK + 4];
  const int wave = threadIdx.x >> 6, lt = threadIdx.x & 63;
  int* tkeys = s_keys[wave];
  int* bcnt = s_bcnt[wave];
  const int lshift = __builtin_ctz((unsigned) sub) - 2;
  const int jr = lt >> lshift, c4 = (lt & ((1 << lshift) - 1)) * 4;
  const int stride = (int) gridDim.x * 4;
  int i = __builtin_amdgcn_readfirstlane((int) blockIdx.x * 4 + wave);
  constexpr int U = ADD ? 5 : 4;
  struct rdesc {
    int row, p0, na, dp0, dn;
  };
  struct arow {
    int qb, len;
  };
  struct brow {
    spg_i4u cc;
    int n;
    int dc;  // ADD: the lane's column of the addend's row (-1: none)
  };
  auto load_row = [&](int r) { return r < n_rows ? rows[r] : -1; };
  auto load_rp = [&](int row) {
    rdesc d{row, 0, 0, 0, 0};
    if (row >= 0) {
      d.p0 = a_rowptr[row];
      d.na = a_rowptr[row + 1] - d.p0;
      if constexpr (ADD) {
        d.dp0 = d_rowptr[row];
        d.dn = d_rowptr[row + 1] - d.dp0;
      }
    }
    return d;
  };
  auto load_a = [&](const rdesc& d) {
    arow a{0, 0};
    if (lt < d.na) {
      const int2 dd = adesc[d.p0 + lt];
      a.qb = dd.x;
      a.len = dd.y;
    }
    return a;
  };
  auto load_b = [&](const arow& a, const rdesc& d) {
    const int na = d.na;
    brow b;
    const int q0 = __shfl(a.qb, jr, 64) + c4, ln = __shfl(a.len, jr, 64) - c4;
    b.n = jr < na ? (ln < 0 ? 0 : ln > 4 ? 4 : ln) : 0;
    b.cc = *reinterpret_cast<const spg_i4u*>(b_colind + (b.n > 0 ? q0 : 0));
    b.dc = -1;
    if constexpr (ADD)
      if (lt < d.dn)
        b.dc = d_colind[d.dp0 + lt];
    return b;
  };
  int r3 = load_row(i + 3 * stride);
  rdesc d0 = load_rp(load_row(i)), d1 = load_rp(load_row(i + stride)), d2 = load_rp(load_row(i + 2 * stride));
  arow a0 = load_a(d0), a1 = load_a(d1);
  brow b0 = load_b(a0, d0);
  auto zero_buckets = [&]() {
#pragma unroll
    for (int j = 0; j < BPL; ++j)
      bcnt[lt * BPL + j] = 0;
    if (lt == 0)
      bcnt[NBK] = 0;
  };
  zero_buckets();
  spg_team_sync<64>();
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): as in spg_direct_kernel
  while (i < n_rows) {
    const brow b1 = load_b(a1, d1);
    const arow a2 = load_a(d2);
    const rdesc d3 = load_rp(r3);
    const int r4 = load_row(i + 4 * stride);
    int col[U], bk[U], ai[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      col[u] = u < 4 ? (u < b0.n ? b0.cc[u & 3] : -1) : b0.dc;
      bk[u] = col[u] >= 0 ? (int) __umulhi((unsigned) col[u], bucket_mul) : 0;
      ai[u] = col[u] >= 0 ? atomicAdd(&bcnt[bk[u]], 1) : 0;
    }
    spg_team_sync<64>();
    {
      int c[BPL], tot = 0;
#pragma unroll
      for (int j = 0; j < BPL; ++j) {
        c[j] = bcnt[lt * BPL + j];
        tot += c[j];
      }
      const int incl = spg_wave_incl_scan(tot);
      int run = incl - tot;
#pragma unroll
      for (int j = 0; j < BPL; ++j) {
        bcnt[lt * BPL + j] = run;
        run += c[j];
      }
      if (lt == 63)
        bcnt[NBK] = incl;
    }
    spg_team_sync<64>();
    // positions, keys in bucket order, and for every key whether an EQUAL key arrived in its bucket before it -- the
    // arrival number ai is the count of such predecessors, which sit at e0 .. e0 + ai - 1.  All reads of a step are issued
    // before the first is used (unconditional: invalid lanes read bucket 0; the key array has four words of slack): the
    // row is a chain of LDS round trips, and a wavefront has nothing else to do while one is in flight.
    int e0[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      e0[u] = bcnt[bk[u]];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (col[u] >= 0)
        tkeys[e0[u] + ai[u]] = col[u];
    spg_team_sync<64>();
    int pk[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        pk[u][k] = tkeys[e0[u] + k];
    int products = 0, distinct = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bool first = col[u] >= 0;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        first = first && !(k < ai[u] && pk[u][k] == col[u]);
      if (__ballot(col[u] >= 0 && ai[u] > 4) != 0ull)  // (a bucket of more than five keys: rare with 128 buckets)
        for (int j = 4; j < ai[u]; ++j)
          first = first && tkeys[e0[u] + j] != col[u];
      products += (int) __popcll(__ballot(col[u] >= 0));
      distinct += (int) __popcll(__ballot(first));
    }
    spg_team_sync<64>();
    zero_buckets();
    spg_team_sync<64>();
    __builtin_amdgcn_s_waitcnt(0x0F70);  // the prefetched loads, before this row's stores (see spg_direct_kernel)
    if (lt == 0) {
      row_nnz[d0.row] = distinct;
      flag[i] = (int32_t) (distinct == products && products > 0);
    }
    b0 = b1;
    a0 = a1;
    a1 = a2;
    d0 = d1;
    d1 = d2;
    d2 = d3;
    r3 = r4;
    i += stride;
  }
}

// floor(NBK * 2^32 / ncols), clamped: the multiplier of the rank sort's bucket function (see spg_hash_kernel)
static unsigned spg_bucket_mul(int nbk, int64_t ncols) {
  const unsigned long long bm = ((unsigned long long) nbk << 32) / (unsigned long long) (ncols > 0 ? ncols : 1);
  return bm > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned) bm;
}

template <typename T, int LOG2HS, int TPR, bool NUMERIC>
static int launch_hash(hipStream_t s, const spblas_gfx950_spgemm_s* st, int bin, const T* a_values,
                       const T* b_values, int32_t* c_rowptr, int32_t* c_colind, T* c_values, T alpha,
                       const T* d_values, T beta, const int32_t* rows = nullptr, int64_t n_rows = 0) {
  // rows != nullptr: that list instead of the whole bin (the rows of bin 2 the direct kernel does not take)
  const int64_t count = rows ? n_rows : st->bin_off[bin + 1] - st->bin_off[bin];
  if (count == 0)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  constexpr int RPB = 256 / TPR;
  const size_t smem = hash_smem_bytes<T, LOG2HS, TPR, NUMERIC>();
  auto kern = spg_hash_kernel<T, LOG2HS, TPR, NUMERIC>;
  if (smem > 48 * 1024)
    SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int) smem));
  int sub = st->sub < TPR ? st->sub : TPR;
  hipLaunchKernelGGL(kern, dim3((unsigned) cdiv(count, RPB)), dim3(256), smem, s, count,
                     rows ? rows : st->perm + st->bin_off[bin], st->a_rowptr, st->a_colind, a_values, st->b_rowptr,
                     st->b_colind, b_values, c_rowptr, c_colind, c_values, alpha, sub,
                     spg_bucket_mul((NUMERIC && TPR == 64) ? SPG_NBK64 : (TPR < 64 ? TPR : 64), st->n), st->d_rowptr,
                     st->d_colind, d_values, beta, (int) (st->b_nnz > 0), st->r_adesc);
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

template <typename T, bool NUMERIC>
static int run_bins(spblas_gfx950_handle_t h, spblas_gfx950_spgemm_s* st, const T* a_values, const T* b_values,
                    int32_t* c_rowptr, int32_t* c_colind, T* c_values, T alpha, const T* d_values = nullptr,
                    T beta = T(0), int skip_upto = 0) {
  hipStream_t s = h->stream;
  int rc;
  // skip_upto = 2 / 3: bins 1-2 / 1-3 were done by the rank-based reuse pass
  if (skip_upto < 2 &&
      (rc = launch_hash<T, 7, 16, NUMERIC>(s, st, 1, a_values, b_values, c_rowptr, c_colind, c_values, alpha,
                                           d_values, beta)))
    return rc;
  // bin 2: a wave per row for the numeric pass (its rank sort works on TPR buckets: 32-lane teams take 3.2 instead of
  // 2.1 ms at cfg5), two rows per wave for the symbolic one (0.61 -> 0.56 ms: more rows' loads in flight)
  if (NUMERIC && skip_upto < 2 && st->dir_desc && (!st->d_rowptr || st->dir_ddesc) && st->b_rowptr && st->r_adesc) {
    if constexpr (NUMERIC) {
      if (st->n_dir > 0) {
        // fp32 with enough direct rows: one pass interleaves B's columns and values (3 ps per entry of B against ~30 ps
        // per line a gather no longer fetches).  SPBLAS_GFX950_SPG_PACK=0 / 1: never / whenever there is a direct row
        static const int pack_env = [] {
          const char* ev = std::getenv("SPBLAS_GFX950_SPG_PACK");
          return ev ? std::atoi(ev) : -1;
        }();
        bool packed = false;
        if constexpr (sizeof(T) == 4) {
          const double dir_entries = (double) st->n_dir * (st->m > 0 ? (double) st->a_nnz / (double) st->m : 0.0);
          if (pack_env != 0 && (pack_env == 1 || dir_entries * 8.0 >= (double) st->b_nnz) &&
              (st->b_pack || dev_alloc((void**) &st->b_pack, (size_t) (st->b_nnz + 4) * 8, s) == SPBLAS_GFX950_STATUS_SUCCESS)) {
            hipLaunchKernelGGL(spg_pack_b_kernel, dim3((unsigned) cdiv(st->b_nnz, 256)), dim3(256), 0, s, st->b_nnz,
                               st->b_colind, b_values, st->b_pack);
            packed = true;
          }
        }
        const int sub_k = st->sub < 64 ? st->sub : 64;
        const unsigned bmul = spg_bucket_mul(SPG_DIR_NBK, st->n);
        const int64_t n_dup = st->n_dir - st->n_nodup;
        const int64_t wgs1 = std::min<int64_t>(cdiv(st->n_nodup, 4), (int64_t) h->num_cus * 8);
        const int64_t wgs2 = std::min<int64_t>(cdiv(n_dup, 4), (int64_t) h->num_cus * 8);
        auto launch = [&](auto kern, int64_t g, int64_t first, int64_t cnt) {
          if (cnt > 0)
            hipLaunchKernelGGL(kern, dim3((unsigned) g), dim3(256), 0, s, (int) cnt, st->dir_desc + first, st->r_adesc, a_values,
                               st->b_colind, b_values, c_colind, c_values, alpha, sub_k, bmul, st->b_pack,
                               st->dir_ddesc ? st->dir_ddesc + first : nullptr, st->d_colind, d_values, beta);
        };
        if (st->d_rowptr) {  // with an addend: its row is a fifth element per lane
          if (packed) {
            launch(spg_direct_kernel<T, true, false, true>, wgs1, 0, st->n_nodup);
            launch(spg_direct_kernel<T, true, true, true>, wgs2, st->n_nodup, n_dup);
          } else {
            launch(spg_direct_kernel<T, false, false, true>, wgs1, 0, st->n_nodup);
            launch(spg_direct_kernel<T, false, true, true>, wgs2, st->n_nodup, n_dup);
          }
        } else if (packed) {
          launch(spg_direct_kernel<T, true, false>, wgs1, 0, st->n_nodup);
          launch(spg_direct_kernel<T, true, true>, wgs2, st->n_nodup, n_dup);
        } else {
          launch(spg_direct_kernel<T, false, false>, wgs1, 0, st->n_nodup);
          launch(spg_direct_kernel<T, false, true>, wgs2, st->n_nodup, n_dup);
        }
        SPB_HIP(hipGetLastError());
      }
      // (the hash kernel on a second stream next to the direct kernel was measured: 1.257 against 1.225 ms)
      if (st->n_rest > 0 &&
          (rc = launch_hash<T, 9, 64, true>(s, st, 2, a_values, b_values, c_rowptr, c_colind, c_values, alpha, d_values,
                                            beta, st->dir_rest, st->n_rest)))
        return rc;
    }
  } else if (!NUMERIC && st->sym_flag && st->n_sortable > 0) {
    // symbolic pass: the sortable rows (the end of the bin's range) are sorted, the others hashed
    const int64_t c2 = st->bin_off[3] - st->bin_off[2], n_other = c2 - st->n_sortable;
    if constexpr (!NUMERIC) {
      if (n_other > 0 && (rc = launch_hash<T, 9, 32, false>(s, st, 2, a_values, b_values, c_rowptr, c_colind, c_values, alpha,
                                                            d_values, beta, st->perm + st->bin_off[2], n_other)))
        return rc;
    }
    const int64_t wgs = std::min<int64_t>(cdiv(st->n_sortable, 4), (int64_t) h->num_cus * 8);
    if (st->d_rowptr)
      hipLaunchKernelGGL(spg_sort_symbolic_kernel<true>, dim3((unsigned) wgs), dim3(256), 0, s, (int) st->n_sortable,
                         st->perm + st->bin_off[3] - st->n_sortable, st->a_rowptr, st->r_adesc, st->b_colind, c_rowptr,
                         st->sym_flag, st->sub < 64 ? st->sub : 64, spg_bucket_mul(SPG_DIR_NBK, st->n), st->d_rowptr,
                         st->d_colind);
    else
      hipLaunchKernelGGL(spg_sort_symbolic_kernel<false>, dim3((unsigned) wgs), dim3(256), 0, s, (int) st->n_sortable,
                         st->perm + st->bin_off[3] - st->n_sortable, st->a_rowptr, st->r_adesc, st->b_colind, c_rowptr,
                         st->sym_flag, st->sub < 64 ? st->sub : 64, spg_bucket_mul(SPG_DIR_NBK, st->n), nullptr, nullptr);
    SPB_HIP(hipGetLastError());
  } else if (skip_upto < 2 &&
             (rc = launch_hash<T, 9, NUMERIC ? 64 : 32, NUMERIC>(s, st, 2, a_values, b_values, c_rowptr, c_colind,
                                                                 c_values, alpha, d_values, beta)))
    return rc;
  if (skip_upto < 3 &&
      (rc = launch_hash<T, 11, 128, NUMERIC>(s, st, 3, a_values, b_values, c_rowptr, c_colind, c_values, alpha,
                                           d_values, beta)))
    return rc;
  if ((rc = launch_hash<T, 13, 256, NUMERIC>(s, st, 4, a_values, b_values, c_rowptr, c_colind, c_values, alpha,
                                           d_values, beta)))
    return rc;
  const int64_t cnt4 = st->bin_off[6] - st->bin_off[5];
  if (cnt4 > 0) {
    const int64_t nwords = (st->n + 31) / 32;
    if (!st->dense_bits) {
      st->dense_blocks = (int) (cnt4 < 64 ? cnt4 : 64);
      if ((rc = dev_alloc((void**) &st->dense_bits, (size_t) st->dense_blocks * nwords * 4, s)))
        return rc;
      SPB_HIP(hipMemsetAsync(st->dense_bits, 0, (size_t) st->dense_blocks * nwords * 4, s));
    }
    if (NUMERIC && (!st->dense_vals || st->dense_vals_type != (int) sizeof(T))) {
      dev_free(st->dense_vals, s);
      st->dense_vals = nullptr;
      if ((rc = dev_alloc(&st->dense_vals, (size_t) st->dense_blocks * st->n * sizeof(T), s)))
        return rc;
      SPB_HIP(hipMemsetAsync(st->dense_vals, 0, (size_t) st->dense_blocks * st->n * sizeof(T), s));
      st->dense_vals_type = (int) sizeof(T);
    }
    hipLaunchKernelGGL((spg_dense_kernel<T, NUMERIC>), dim3((unsigned) st->dense_blocks), dim3(256), 0, s, cnt4,
                       st->perm + st->bin_off[5], st->n, st->a_rowptr, st->a_colind, a_values, st->b_rowptr,
                       st->b_colind, b_values, c_rowptr, c_colind, c_values, alpha, st->dense_bits,
                       static_cast<T*>(st->dense_vals), st->d_rowptr, st->d_colind, d_values, beta);
    SPB_HIP(hipGetLastError());
  }
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

template <typename T, int TPR, int CAP, typename RT>
static void launch_ranked_fill(hipStream_t s, const spblas_gfx950_spgemm_s* st, int64_t cnt, int64_t first,
                               const T* a_values, const T* b_values, int32_t* c_colind, T* c_values, T alpha,
                               int copy_cols, const T* d_values, T beta, const int32_t* pbase, const RT* ranks) {
  const int sub = st->identity_b ? 1 : st->sub < TPR ? st->sub : TPR;  // identity B: one lane per A entry
  constexpr int RPB = 256 / TPR;
  if (st->identity_b || st->d_rowptr)
    hipLaunchKernelGGL((spg_ranked_fill_kernel<T, TPR, CAP, 4, true, RT>), dim3((unsigned) cdiv(cnt, RPB)), dim3(256), 0,
                       s, cnt, st->perm + first, st->a_rowptr, st->a_colind, a_values, st->b_rowptr, st->r_adesc,
                       b_values, st->rowptr, st->r_cols, c_colind, c_values, alpha, sub, pbase, ranks, copy_cols,
                       st->d_rowptr, d_values, beta);
  else
    hipLaunchKernelGGL((spg_ranked_fill_kernel<T, TPR, CAP, 4, false, RT>), dim3((unsigned) cdiv(cnt, RPB)), dim3(256), 0,
                       s, cnt, st->perm + first, st->a_rowptr, st->a_colind, a_values, st->b_rowptr, st->r_adesc,
                       b_values, st->rowptr, st->r_cols, c_colind, c_values, alpha, sub, pbase, ranks, copy_cols,
                       st->d_rowptr, d_values, beta);
}

// the recording pass for the rows of bins 1-3 ...
static void launch_rank_record(hipStream_t s, const spblas_gfx950_spgemm_s* st) {
  const int64_t c1 = st->bin_off[2] - st->bin_off[1], c2 = st->bin_off[3] - st->bin_off[2];
  const int64_t c3 = st->bin_off[4] - st->bin_off[3];
  const int sub1 = st->identity_b ? 1 : st->sub < 16 ? st->sub : 16;
  const int sub2 = st->identity_b ? 1 : st->sub < 64 ? st->sub : 64;
  if (c1 > 0)
    hipLaunchKernelGGL((spg_rank_record_kernel<16, 64, uint8_t>), dim3((unsigned) cdiv(c1, 16)), dim3(256), 0, s, c1,
                       st->perm + st->bin_off[1], st->a_rowptr, st->a_colind, st->b_rowptr, st->r_adesc, st->b_colind,
                       st->rowptr, st->r_cols, sub1, st->r_pbase, st->r_rank, st->d_rowptr, st->d_colind);
  if (c2 > 0)
    hipLaunchKernelGGL((spg_rank_record_kernel<64, 256, uint8_t>), dim3((unsigned) cdiv(c2, 4)), dim3(256), 0, s, c2,
                       st->perm + st->bin_off[2], st->a_rowptr, st->a_colind, st->b_rowptr, st->r_adesc, st->b_colind,
                       st->rowptr, st->r_cols, sub2, st->r_pbase, st->r_rank, st->d_rowptr, st->d_colind);
  if (c3 > 0 && st->r_rank3)
    hipLaunchKernelGGL((spg_rank_record_kernel<64, 1024, uint16_t>), dim3((unsigned) cdiv(c3, 4)), dim3(256), 0, s, c3,
                       st->perm + st->bin_off[3], st->a_rowptr, st->a_colind, st->b_rowptr, st->r_adesc, st->b_colind,
                       st->rowptr, st->r_cols, sub2, st->r_pbase3, st->r_rank3, st->d_rowptr, st->d_colind);
}

// ... and their fills by rank.  Team width of a bin-2 row: the A entries of a row are walked TPR at a time, and a
// 64-lane team on a 16-entry row leaves the wave with one row's dependent loads in flight; narrower teams put 2-4
// rows into every wave (cfg5: 0.93 -> 0.78 ms; SPBLAS_GFX950_SPG_RANKED_TPR=16/32/64 overrides the choice).
template <typename T>
static void launch_ranked(hipStream_t s, const spblas_gfx950_spgemm_s* st, const T* a_values, const T* b_values,
                          int32_t* c_colind, T* c_values, T alpha, int copy_cols, const T* d_values, T beta) {
  const int64_t c1 = st->bin_off[2] - st->bin_off[1], c2 = st->bin_off[3] - st->bin_off[2];
  static const int tpr_env = [] {
    const char* e = std::getenv("SPBLAS_GFX950_SPG_RANKED_TPR");
    return e ? std::atoi(e) : 0;
  }();
  static const int tpr1_env = [] {
    const char* e = std::getenv("SPBLAS_GFX950_SPG_RANKED_TPR1");
    return e ? std::atoi(e) : 0;
  }();
  const double avg_a = st->m > 0 ? (double) st->a_nnz / (double) st->m : 0.0;
  const int tpr2 = tpr_env ? tpr_env : avg_a <= 16.0 ? 16 : avg_a <= 32.0 ? 32 : 64;
  // bin 1 (<= 64 products): 8-lane teams when the rows are that short (add(): 16 + 16 entries per row at the 8f size)
  const int tpr1 = tpr1_env ? tpr1_env : (st->identity_b || avg_a <= 8.0) ? 8 : 16;
  if (c1 > 0) {
    if (tpr1 == 8)
      launch_ranked_fill<T, 8, 64, uint8_t>(s, st, c1, st->bin_off[1], a_values, b_values, c_colind, c_values, alpha,
                                            copy_cols, d_values, beta, st->r_pbase, st->r_rank);
    else
      launch_ranked_fill<T, 16, 64, uint8_t>(s, st, c1, st->bin_off[1], a_values, b_values, c_colind, c_values, alpha,
                                             copy_cols, d_values, beta, st->r_pbase, st->r_rank);
  }
  if (c2 > 0) {
    if (tpr2 == 16)
      launch_ranked_fill<T, 16, 256, uint8_t>(s, st, c2, st->bin_off[2], a_values, b_values, c_colind, c_values, alpha,
                                              copy_cols, d_values, beta, st->r_pbase, st->r_rank);
    else if (tpr2 == 32)
      launch_ranked_fill<T, 32, 256, uint8_t>(s, st, c2, st->bin_off[2], a_values, b_values, c_colind, c_values, alpha,
                                              copy_cols, d_values, beta, st->r_pbase, st->r_rank);
    else
      launch_ranked_fill<T, 64, 256, uint8_t>(s, st, c2, st->bin_off[2], a_values, b_values, c_colind, c_values, alpha,
                                              copy_cols, d_values, beta, st->r_pbase, st->r_rank);
  }
  // bin 3 (257 .. 1024 products): a wave per row, 4 KiB (fp32) of LDS accumulators per row
  const int64_t c3 = st->bin_off[4] - st->bin_off[3];
  if (c3 > 0 && st->r_rank3)
    launch_ranked_fill<T, 64, 1024, uint16_t>(s, st, c3, st->bin_off[3], a_values, b_values, c_colind, c_values, alpha,
                                              copy_cols, d_values, beta, st->r_pbase3, st->r_rank3);
}

// numeric pass.  Rows that sit in the LDS-hash bins 1-2 are eligible for reuse -- three- and four-argument products and
// add() (identity B + addend) alike: the SECOND pass on a symbolic result runs the hash kernels and then records the
// rank of every product and addend entry (SPBLAS_GFX950_SPGEMM_REUSE=0 turns the recording off, =2 records in the
// first pass already); every later pass accumulates by rank.  Rows in the other bins always take the hash / dense
// kernels.
template <typename T>
static int spgemm_numeric_typed(spblas_gfx950_handle_t h, spblas_gfx950_spgemm_s* st, const T* a_values,
                                const T* b_values, int32_t* c_colind, T* c_values, T alpha, const T* d_values, T beta) {
  hipStream_t s = h->stream;
  if (st->r_ready) {
    // columns: rewritten unless the caller vouches for the array's contents (OPT_SPGEMM_KEEP_COLIND) and it is the
    // array the previous pass filled -- the address alone proves nothing, allocators hand freed addresses out again
    launch_ranked<T>(s, st, a_values, b_values, c_colind, c_values, alpha,
                     !(h->spgemm_keep_colind != 0 && c_colind == st->r_last_colind), d_values, beta);
    SPB_HIP(hipGetLastError());
    st->r_last_colind = c_colind;
    return run_bins<T, true>(h, st, a_values, b_values, st->rowptr, c_colind, c_values, alpha, d_values, beta,
                             st->r_rank3 ? 3 : 2);
  }
  int rc = run_bins<T, true>(h, st, a_values, b_values, st->rowptr, c_colind, c_values, alpha, d_values, beta);
  if (rc)
    return rc;
  // the first fill of a symbolic result stays a plain hash pass; a second one shows that the structure is being
  // reused and records the ranks (hash pass + 1.3 ms once at cfg5), the third and later ones take the rank path
  const char* env = std::getenv("SPBLAS_GFX950_SPGEMM_REUSE");
  // (sortable rows with an addend may hold up to 320 entries in bin 2's range of perm[], whose rank path has 256 slots per
  // row: such a result keeps the sort-based fill, which costs what a first fill costs)
  const bool want = !(env && env[0] == '0') && ++st->numeric_calls >= (env && env[0] == '2' ? 1 : 2) &&
                    !(st->d_rowptr && st->n_sortable > 0);
  const int64_t small_rows = st->bin_off[4] - st->bin_off[1];  // bins 1-3
  if (!want || small_rows == 0 || st->m == 0)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  // record: products per row -> r_pbase (bins 1-2) / r_pbase3 (bin 3), a copy of the sorted columns, then the ranks
  const int64_t m = st->m, nb = cdiv(m, 2048);
  long long *partials = nullptr, *partials3 = nullptr;
  auto drop = [&]() {  // out of memory for the optional fast path: keep the hash path
    dev_free(st->r_pbase, s);
    dev_free(st->r_pbase3, s);
    dev_free(st->r_rank, s);
    dev_free(st->r_rank3, s);
    dev_free(st->r_cols, s);
    st->r_pbase = st->r_pbase3 = nullptr;
    st->r_rank = nullptr;
    st->r_rank3 = nullptr;
    st->r_cols = nullptr;
    return SPBLAS_GFX950_STATUS_SUCCESS;
  };
  if (dev_alloc((void**) &st->r_pbase, (size_t) (m + 1) * 4, s) || dev_alloc((void**) &st->r_pbase3, (size_t) (m + 1) * 4, s) ||
      dev_alloc((void**) &partials, (size_t) 2 * (nb + 2) * sizeof(long long), s))
    return drop();
  partials3 = partials + (nb + 2);
  hipLaunchKernelGGL(spg_products_kernel, dim3((unsigned) cdiv(m, 32)), dim3(256), 0, s, m, st->a_rowptr, st->a_colind,
                     st->b_rowptr, st->r_adesc, st->d_rowptr, st->r_pbase, st->r_pbase3);
  long long* total_dev = scan_counts_i32(s, m, st->r_pbase, partials);
  long long* total3_dev = scan_counts_i32(s, m, st->r_pbase3, partials3);
  long long total = 0, total3 = 0;
  hipError_t e = hipMemcpyAsync(&total, total_dev, sizeof(total), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess)
    e = hipMemcpyAsync(&total3, total3_dev, sizeof(total3), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess)
    e = hipStreamSynchronize(s);
  dev_free(partials, s);
  if (e != hipSuccess)
    return hip_fail(e);
  if (total + total3 <= 0 || total > INT32_MAX || total3 > INT32_MAX ||
      (total > 0 && dev_alloc((void**) &st->r_rank, (size_t) total, s) != SPBLAS_GFX950_STATUS_SUCCESS) ||
      (total3 > 0 && dev_alloc((void**) &st->r_rank3, (size_t) total3 * 2, s) != SPBLAS_GFX950_STATUS_SUCCESS) ||
      dev_alloc((void**) &st->r_cols, (size_t) st->c_nnz * 4, s) != SPBLAS_GFX950_STATUS_SUCCESS)
    return drop();
  SPB_HIP(hipMemcpyAsync(st->r_cols, c_colind, (size_t) st->c_nnz * 4, hipMemcpyDeviceToDevice, s));
  launch_rank_record(s, st);
  SPB_HIP(hipGetLastError());
  st->r_last_colind = c_colind;
  st->r_ready = true;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

static void spgemm_release(spblas_gfx950_spgemm_s* st, hipStream_t s) {
  dev_free(st->rowptr, s);
  dev_free(st->perm, s);
  dev_free(st->dense_bits, s);
  dev_free(st->dense_vals, s);
  dev_free(st->r_pbase, s);
  dev_free(st->r_rank, s);
  dev_free(st->r_pbase3, s);
  dev_free(st->r_rank3, s);
  st->r_pbase3 = nullptr;
  st->r_rank3 = nullptr;
  dev_free(st->r_cols, s);
  dev_free(st->r_adesc, s);
  st->r_adesc = nullptr;
  dev_free(st->dir_desc, s);
  dev_free(st->dir_ddesc, s);
  st->dir_ddesc = nullptr;
  dev_free(st->dir_rest, s);
  dev_free(st->b_pack, s);
  st->dir_desc = nullptr;
  st->dir_rest = nullptr;
  st->b_pack = nullptr;
  st->n_dir = st->n_rest = st->n_nodup = 0;
  st->r_pbase = nullptr;
  st->r_rank = nullptr;
  st->r_cols = nullptr;
  st->r_last_colind = nullptr;
  st->r_ready = false;
  st->numeric_calls = 0;
  st->rowptr = nullptr;
  st->perm = nullptr;
  st->dense_bits = nullptr;
  st->dense_vals = nullptr;
  st->dense_vals_type = -1;
  st->c_nnz = -1;
}

} // namespace spb

using namespace spb;

extern "C" {

int spblas_gfx950_spgemm_create(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t* state) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!state)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  *state = new (std::nothrow) spblas_gfx950_spgemm_s();
  return *state ? SPBLAS_GFX950_STATUS_SUCCESS : SPBLAS_GFX950_STATUS_ALLOC_FAILED;
}

int spblas_gfx950_spgemm_destroy(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t state) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!state)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  spgemm_release(state, handle->stream);
  delete state;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_spgemm_info(spblas_gfx950_spgemm_t st, int64_t info[8]) {
  if (!st || !info)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  for (int i = 0; i < 8; ++i)
    info[i] = 0;
  info[0] = st->c_nnz;
  info[1] = st->bin_off[3] - st->bin_off[2];
  info[2] = st->dir_desc ? st->n_dir : 0;
  info[3] = st->r_ready ? 1 : 0;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_spgemm_set_addend(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t st, int64_t d_nnz,
                                    const int32_t* d_rowptr, const int32_t* d_colind) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!st || (d_rowptr && d_nnz > 0 && !d_colind))
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (d_nnz < 0 || d_nnz > INT32_MAX)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  st->d_rowptr = d_rowptr;
  st->d_colind = d_rowptr ? d_colind : nullptr;
  st->d_nnz = d_rowptr ? d_nnz : 0;
  st->c_nnz = -1;  // a new symbolic pass is required
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

static int spgemm_symbolic_impl(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t st, int64_t m, int64_t k,
                                int64_t n, int64_t a_nnz, const int32_t* a_rowptr, const int32_t* a_colind,
                                int64_t b_nnz, const int32_t* b_rowptr, const int32_t* b_colind,
                                int32_t* c_rowptr, int64_t* c_nnz, bool identity_b) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (stream_capturing(handle->stream))  // inspect-class call: sizes its output on the host, never part of a graph
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  if (!st || !c_nnz || !a_rowptr || (!identity_b && !b_rowptr) || !c_rowptr || (a_nnz > 0 && !a_colind) ||
      (b_nnz > 0 && !b_colind))
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (m < 0 || k < 0 || n < 0 || a_nnz < 0 || b_nnz < 0 || m >= INT32_MAX || k > INT32_MAX || n > INT32_MAX ||
      a_nnz > INT32_MAX || b_nnz > INT32_MAX)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  hipStream_t s = handle->stream;
  spgemm_release(st, s);
  st->has_addend = st->d_rowptr != nullptr;
  st->identity_b = identity_b;
  st->m = m; st->k = k; st->n = n; st->a_nnz = a_nnz; st->b_nnz = b_nnz;
  st->a_rowptr = a_rowptr; st->a_colind = a_colind; st->b_rowptr = b_rowptr; st->b_colind = b_colind;
  // lanes per B row: power of two near the average B row length
  const double avg_b = k > 0 ? (double) b_nnz / (double) k : 0.0;
  st->sub = 4;
  while (st->sub < 64 && st->sub < avg_b)
    st->sub <<= 1;
  if (st->sub > 16)
    st->sub = 16;  // bin 1 teams are 16 lanes wide

  int rc;
  if ((rc = dev_alloc((void**) &st->rowptr, (size_t) (m + 1) * 4, s)))
    return rc;
  if (m == 0) {
    SPB_HIP(hipMemsetAsync(st->rowptr, 0, 4, s));
    SPB_HIP(hipMemsetAsync(c_rowptr, 0, 4, s));
    SPB_HIP(hipStreamSynchronize(s));
    st->c_nnz = 0;
    *c_nnz = 0;
    return SPBLAS_GFX950_STATUS_SUCCESS;
  }
  if ((rc = dev_alloc((void**) &st->perm, (size_t) m * 4, s)))
    return rc;
  // temporaries of this call live in the handle's grow-only scratch: no allocation on repeated calls
  // (on the null stream every dev_alloc / dev_free is a synchronous hipMalloc / hipFree)
  const int64_t nb = cdiv(m, 2048);
  const size_t bin_bytes = (((size_t) m * 4) + 255) & ~(size_t) 255;
  const size_t cnt_bytes = 256;  // 2 * SPG_NCNT counters
  void* scratch = nullptr;
  if ((rc = handle_scratch(handle, bin_bytes + cnt_bytes + (size_t) (nb + 1) * sizeof(long long) + 256, &scratch)))
    return rc;
  int32_t* bin_of_row = static_cast<int32_t*>(scratch);
  unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(static_cast<char*>(scratch) + bin_bytes);
  long long* partials = reinterpret_cast<long long*>(static_cast<char*>(scratch) + bin_bytes + cnt_bytes);
  SPB_HIP(hipMemsetAsync(d_cnt, 0, 2 * SPG_NCNT * sizeof(unsigned long long), s));
  // (start, length) of the B row of every A entry, one coalesced pass while b_rowptr still has the L2s to itself
  {
    const char* env = std::getenv("SPBLAS_GFX950_SPG_ADESC");
    if (b_rowptr && a_nnz > 0 && !(env && env[0] == '0'))
      (void) dev_alloc((void**) &st->r_adesc, (size_t) a_nnz * sizeof(int2), s);  // (filled by spg_bound_kernel below)
  }
  readback_scope rb_scope(handle);
  const int64_t bound_wgs = cdiv(m, 32) < 8 * (int64_t) handle->num_cus ? cdiv(m, 32) : 8 * (int64_t) handle->num_cus;
  // rows of bin 2 that a wavefront can take in one round of vector loads get a counter of their own and the END of bin 2's
  // range in perm[] (SPBLAS_GFX950_SPG_DIRECT=0: none): sorted, not hashed, by both passes
  static const int dir_env = [] {
    const char* ev = std::getenv("SPBLAS_GFX950_SPG_DIRECT");
    return ev ? std::atoi(ev) : 1;
  }();
  const int sortable_ok = dir_env != 0 && !identity_b && b_rowptr && st->r_adesc && b_nnz >= 4 &&
                          (!st->d_rowptr || std::getenv("SPBLAS_GFX950_SPG_DIRECT_ADD") == nullptr ||
                           std::atoi(std::getenv("SPBLAS_GFX950_SPG_DIRECT_ADD")) != 0);
  hipLaunchKernelGGL(spg_bound_kernel, dim3((unsigned) bound_wgs), dim3(256), 0, s, m, a_rowptr, a_colind,
                     b_rowptr, st->d_rowptr, st->r_adesc, bin_of_row, d_cnt, st->sub < 64 ? st->sub : 64, b_nnz, sortable_ok,
                     st->r_adesc);
  SPB_HIP(hipGetLastError());
  // (the bin counts come back through the handle's pinned buffer -- a copy kernel, no SDMA transfer: hipMemcpyAsync to
  // pageable memory left the queue idle for 35 us on either side of it, round-5 timeline)
  unsigned long long counts[SPG_NCNT];
  if ((rc = readback_add(handle, counts, d_cnt, sizeof(counts))) || (rc = readback_flush(handle)))
    return rc;
  unsigned long long cursors[SPG_NCNT];
  st->bin_off[0] = 0;
  for (int b = 0; b < SPG_NBINS; ++b) {
    cursors[b] = (unsigned long long) st->bin_off[b];
    st->bin_off[b + 1] = st->bin_off[b] + (int64_t) counts[b] + (b == 2 ? (int64_t) counts[SPG_SORTABLE] : 0);
  }
  cursors[SPG_SORTABLE] = (unsigned long long) st->bin_off[2] + counts[2];
  st->n_sortable = (int64_t) counts[SPG_SORTABLE];
  if ((rc = upload_add(handle, d_cnt + SPG_NCNT, cursors, sizeof(cursors))))
    return rc;
  hipLaunchKernelGGL(spg_fill_perm_kernel, dim3((unsigned) cdiv(m, 1024)), dim3(1024), 0, s, m, bin_of_row,
                     d_cnt + SPG_NCNT, st->perm);
  SPB_HIP(hipGetLastError());

  // (the flags of the sortable rows go where the bin numbers were: dead once perm[] is filled, and [m + 1] ints with the
  // counter block behind them)
  st->sym_flag = st->n_sortable > 0 ? bin_of_row : nullptr;
  // distinct-column counts per row -> st->rowptr (as counts)
  if (st->bin_off[1] > 0)
    hipLaunchKernelGGL(spg_zero_rows_kernel, dim3((unsigned) cdiv(st->bin_off[1], 256)), dim3(256), 0, s,
                       st->bin_off[1], st->perm, st->rowptr);
  rc = run_bins<float, false>(handle, st, nullptr, nullptr, st->rowptr, nullptr, nullptr, 0.f);
  if (rc == SPBLAS_GFX950_STATUS_SUCCESS) {
    // counts -> offsets, written to the plan's copy and the caller's c_rowptr by the same kernel; one host
    // synchronisation for the total (the offsets are stream-ordered like everything else the caller does next)
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned) nb), dim3(256), 0, s, m, st->rowptr, partials);
    hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(256), 0, s, nb, partials);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned) nb), dim3(256), 0, s, m, st->rowptr, partials, c_rowptr);
    long long total = 0;
    hipError_t e = hipSuccess;
    if ((rc = readback_add(handle, &total, partials + nb, sizeof(total))))
      return rc;
    // sortable rows (see spg_direct_kernel): one descriptor each, with the symbolic sort's "no shared column" flag; the
    // rows of the bin that are not sortable keep the hash kernel (dir_rest)
    const int64_t c2 = st->bin_off[3] - st->bin_off[2], ns = st->n_sortable, n_other = c2 - ns;
    bool classify = e == hipSuccess && ns > 0 && st->sym_flag && ns <= INT32_MAX - 16;
    if (classify && (dev_alloc((void**) &st->dir_desc, (size_t) ns * sizeof(int4), s) != SPBLAS_GFX950_STATUS_SUCCESS ||
                     (st->d_rowptr && dev_alloc((void**) &st->dir_ddesc, (size_t) ns * sizeof(int2), s) != SPBLAS_GFX950_STATUS_SUCCESS) ||
                     (n_other > 0 && dev_alloc((void**) &st->dir_rest, (size_t) n_other * 4, s) != SPBLAS_GFX950_STATUS_SUCCESS))) {
      dev_free(st->dir_desc, s);
      dev_free(st->dir_ddesc, s);
      dev_free(st->dir_rest, s);
      st->dir_desc = nullptr;
      st->dir_ddesc = nullptr;
      st->dir_rest = nullptr;
      classify = false;  // out of memory for the optional lists: every row hashes
    }
    long long n_nodup = 0;
    if (classify) {
      if (n_other > 0)
        e = hipMemcpyAsync(st->dir_rest, st->perm + st->bin_off[2], (size_t) n_other * 4, hipMemcpyDeviceToDevice, s);
      // (the scan's partial sums reuse `partials`: the copy of the total above is ordered before these kernels)
      long long* n_nodup_dev = scan_counts_i32(s, ns, st->sym_flag, partials);
      hipLaunchKernelGGL(spg_direct_lists_kernel, dim3((unsigned) cdiv(ns, 256)), dim3(256), 0, s, ns,
                         st->perm + st->bin_off[3] - ns, a_rowptr, st->rowptr, st->sym_flag, st->dir_desc, st->d_rowptr,
                         st->dir_ddesc);
      if (e == hipSuccess && (rc = readback_add(handle, &n_nodup, n_nodup_dev, sizeof(n_nodup))))
        return rc;
    }
    st->sym_flag = nullptr;
    {
      const int rc_f = readback_flush(handle);  // the ONE wait the API asks for: nnz(C) sizes the caller's arrays
      if (rc_f)
        return rc_f;
    }
    st->n_dir = classify ? ns : 0;
    st->n_nodup = classify ? n_nodup : 0;
    st->n_rest = classify ? n_other : 0;
    if (e != hipSuccess)
      rc = hip_fail(e);
    else if (total > INT32_MAX)
      rc = SPBLAS_GFX950_STATUS_INVALID_SIZE;  // nnz(C) does not fit int32 offsets (c_rowptr holds wrapped offsets)
    else {
      st->c_nnz = total;
      *c_nnz = total;
    }
  }
  return rc;
}

int spblas_gfx950_spgemm_symbolic(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t st, int64_t m, int64_t k,
                                  int64_t n, int64_t a_nnz, const int32_t* a_rowptr, const int32_t* a_colind,
                                  int64_t b_nnz, const int32_t* b_rowptr, const int32_t* b_colind,
                                  int32_t* c_rowptr, int64_t* c_nnz) {
  return spgemm_symbolic_impl(handle, st, m, k, n, a_nnz, a_rowptr, a_colind, b_nnz, b_rowptr, b_colind, c_rowptr,
                              c_nnz, false);
}

static int spgemm_numeric_impl(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t st, const void* alpha,
                               const int32_t* a_rowptr, const int32_t* a_colind, const void* a_values,
                               const int32_t* b_rowptr, const int32_t* b_colind, const void* b_values,
                               const void* beta, const int32_t* d_rowptr, const int32_t* d_colind,
                               const void* d_values, int32_t* c_rowptr, int32_t* c_colind, void* c_values,
                               int64_t c_capacity, int value_type) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!st || !alpha || !a_rowptr || !c_rowptr)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (st->c_nnz < 0)
    return SPBLAS_GFX950_STATUS_INVALID_VALUE;  // symbolic has not run
  if (value_type != SPBLAS_GFX950_F32 && value_type != SPBLAS_GFX950_F64)
    return SPBLAS_GFX950_STATUS_INVALID_VALUE;
  if (st->identity_b != (b_rowptr == nullptr))
    return SPBLAS_GFX950_STATUS_PLAN_MISMATCH;
  if (st->has_addend != (d_rowptr != nullptr))
    return SPBLAS_GFX950_STATUS_PLAN_MISMATCH;  // the symbolic pass saw a different operand set
  if (st->has_addend && (!beta || (st->d_nnz > 0 && (!d_colind || !d_values))))
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (c_capacity < st->c_nnz)
    return SPBLAS_GFX950_STATUS_INSUFFICIENT_SPACE;
  if (st->c_nnz > 0 && (!c_colind || !c_values || (st->a_nnz > 0 && (!a_values || !a_colind)) ||
                        (!st->identity_b && st->b_nnz > 0 && (!b_values || !b_colind))))
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  hipStream_t s = handle->stream;
  // pointers may be rebound between calls as long as the pattern is unchanged
  // (multiply_spgemm.hpp:195-208 rebinds them with rocsparse_csr_set_pointers)
  st->a_rowptr = a_rowptr; st->a_colind = a_colind; st->b_rowptr = b_rowptr; st->b_colind = b_colind;
  if (st->has_addend) {
    st->d_rowptr = d_rowptr;
    st->d_colind = d_colind;
  }
  if (c_rowptr != st->rowptr)
    SPB_HIP(hipMemcpyAsync(c_rowptr, st->rowptr, (size_t) (st->m + 1) * 4, hipMemcpyDeviceToDevice, s));
  if (st->c_nnz == 0)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  if (value_type == SPBLAS_GFX950_F32)
    return spgemm_numeric_typed<float>(handle, st, static_cast<const float*>(a_values),
                                       static_cast<const float*>(b_values), c_colind, static_cast<float*>(c_values),
                                       *static_cast<const float*>(alpha), static_cast<const float*>(d_values),
                                       beta ? *static_cast<const float*>(beta) : 0.f);
  return spgemm_numeric_typed<double>(handle, st, static_cast<const double*>(a_values),
                                      static_cast<const double*>(b_values), c_colind, static_cast<double*>(c_values),
                                      *static_cast<const double*>(alpha), static_cast<const double*>(d_values),
                                      beta ? *static_cast<const double*>(beta) : 0.0);
}

int spblas_gfx950_spgemm_numeric(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t st, const void* alpha,
                                 const int32_t* a_rowptr, const int32_t* a_colind, const void* a_values,
                                 const int32_t* b_rowptr, const int32_t* b_colind, const void* b_values,
                                 int32_t* c_rowptr, int32_t* c_colind, void* c_values, int64_t c_capacity,
                                 int value_type) {
  return spgemm_numeric_impl(handle, st, alpha, a_rowptr, a_colind, a_values, b_rowptr, b_colind, b_values, nullptr,
                             nullptr, nullptr, nullptr, c_rowptr, c_colind, c_values, c_capacity, value_type);
}

int spblas_gfx950_spgemm_numeric_addend(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t st, const void* alpha,
                                        const int32_t* a_rowptr, const int32_t* a_colind, const void* a_values,
                                        const int32_t* b_rowptr, const int32_t* b_colind, const void* b_values,
                                        const void* beta, const int32_t* d_rowptr, const int32_t* d_colind,
                                        const void* d_values, int32_t* c_rowptr, int32_t* c_colind, void* c_values,
                                        int64_t c_capacity, int value_type) {
  if (!d_rowptr)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  return spgemm_numeric_impl(handle, st, alpha, a_rowptr, a_colind, a_values, b_rowptr, b_colind, b_values, beta,
                             d_rowptr, d_colind, d_values, c_rowptr, c_colind, c_values, c_capacity, value_type);
}

/* ---- add: C = alpha*A + beta*B through the same accumulators (B := identity, D := B) ---- */
int spblas_gfx950_csr_add_symbolic(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t st, int64_t m, int64_t n,
                                   int64_t a_nnz, const int32_t* a_rowptr, const int32_t* a_colind, int64_t b_nnz,
                                   const int32_t* b_rowptr, const int32_t* b_colind, int32_t* c_rowptr,
                                   int64_t* c_nnz) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!st || !b_rowptr)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  int rc = spblas_gfx950_spgemm_set_addend(handle, st, b_nnz, b_rowptr, b_colind);
  if (rc)
    return rc;
  return spgemm_symbolic_impl(handle, st, m, n, n, a_nnz, a_rowptr, a_colind, 0, nullptr, nullptr, c_rowptr, c_nnz,
                              true);
}

int spblas_gfx950_csr_add_numeric(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t st, const void* alpha,
                                  const int32_t* a_rowptr, const int32_t* a_colind, const void* a_values,
                                  const void* beta, const int32_t* b_rowptr, const int32_t* b_colind,
                                  const void* b_values, int32_t* c_rowptr, int32_t* c_colind, void* c_values,
                                  int64_t c_capacity, int value_type) {
  if (!b_rowptr)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  return spgemm_numeric_impl(handle, st, alpha, a_rowptr, a_colind, a_values, nullptr, nullptr, nullptr, beta,
                             b_rowptr, b_colind, b_values, c_rowptr, c_colind, c_values, c_capacity, value_type);
}

} // extern "C"

// Loads this file's code object (the runtime loads a code object at the first use of one of its kernels: milliseconds
// that would otherwise fall on the caller's first inspect / compute call -- handle.hip: spblas_gfx950_create).
namespace spb {
void preload_spgemm() {
  hipFuncAttributes attr;
  (void) hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&spg_bound_kernel));
  (void) hipGetLastError();
}
} // namespace spb
