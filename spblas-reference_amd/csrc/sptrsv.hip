// Sparse triangular solve  x = inv(op(A)) b  for CSR operands on gfx950  (SURVEY.md section 8f rank 4).
//
// Device counterpart of triangular_solve_inspect / triangular_solve
// (/root/reference/include/spblas/algorithms/triangular_solve_impl.hpp:13-107).  Semantics of the
// reference loop (:57-93): only the strict triangle selected by `uplo` and the diagonal entries
// of a row are read -- a general matrix may be passed, the other triangle is ignored;
//   x_i = (b_i - sum_{k in strict part of row i} a_ik x_k) / a_ii      explicit_diagonal
//   x_i =  b_i - sum_{k in strict part of row i} a_ik x_k             implicit_unit_diagonal
// with a_ii = the last stored entry of row i whose column is i.
//
// The reference walks the rows sequentially.  Here triangular_solve_inspect builds LEVEL SETS on the
// device (rows of one level depend only on rows of earlier levels) and the solve runs level by level:
//   inspect   levels by dependency polling: one self-scheduling kernel hands rows out in index order, a row's level is
//             1 + the deepest level among the rows it reads (polled until known; bounded spins), then a histogram
//             of the levels and a placement pass sort the rows by level.  Fallback (a wave gave up polling, or
//             SPBLAS_GFX950_TRSV_KAHN=1): in-degrees + transposed adjacency + Kahn's algorithm, frontier by frontier.
//   solve     ONE cooperative launch (trsv_coop_kernel): a grid barrier per wide level, a workgroup barrier per level
//             inside a run of NARROW levels (workgroup 0 alone), the loads that do not depend on x pipelined three
//             levels ahead.  Where a cooperative launch is not possible (stream capture, device attribute missing,
//             a narrow run longer than 4096 levels, SPBLAS_GFX950_TRSV_COOP=0): one launch per wide level and one
//             single-workgroup launch per narrow run.  Measured at 4 M rows / 246 levels / 36 M entries: 1.70 ms
//             cooperative, 1.74 ms launch per level; the floor is the x gather (32 M agent-scope 64-byte fetches,
//             ~0.5 ms) plus one store -> load hand-off per level.  (Rounds 2 - 4 carried a third, barrier-free form
//             behind SPBLAS_GFX950_TRSV_SELFSCHED=1 -- one self-scheduling launch per run of wide levels with {value,
//             solve number} granules as the hand-off, 2.59 ms.  Round 5's fuzzer found that its fp64 form stops making
//             progress on some 400 k-row matrices (the bounded polls then gave up with wrong values); a slower opt-in
//             path has no user, so it was removed rather than repaired.)
// Row sums are computed G lanes wide and tree-reduced, so they re-associate with respect to the
// reference's sequential loop: parity is norm-wise, not bit-wise.
#include "common.hpp"
#include "scan.hpp"

#include <algorithm>
#include <cstdlib>
#include <new>
#include <vector>

static int env_int(const char* name, int def) {  // tuning / test hook
  const char* v = std::getenv(name);
  return v && *v ? std::atoi(v) : def;
}

#define TRSV_NARROW 2048       // inspect: frontiers with fewer rows are advanced by the single-workgroup kernel
#define TRSV_BLOCK_THREADS 1024

struct spblas_gfx950_trsv_s {
  int64_t m = 0, nnz = 0;
  int uplo = 0, diag = 0;
  int32_t* order = nullptr;      // [m] rows sorted by level
  int32_t* level_ptr = nullptr;  // [n_levels + 1] device copy
  std::vector<int32_t> h_level_ptr;
  // launch groups: {first_level, last_level (exclusive), wide ? 1 : 0}
  struct group_t {
    int32_t l0, l1, wide;
  };
  std::vector<group_t> groups;
  int64_t max_width = 0;
  int lanes = 8;  // lanes per row in the solve kernels
  int narrow = 128;      // levels with fewer rows are "narrow": walked by one workgroup
  bool coop_ok = false;  // the solve is ONE cooperative launch (trsv_coop_kernel)
  int32_t* tickets = nullptr;         // device: status word and the grid barrier's counters / release lines
  // pinned, device-visible host word: a solve whose grid barrier ran into its poll bound sets it (system-scope store, only
  // on that path), the NEXT solve on this plan -- and spblas_gfx950_sptrsv_status -- reads it without touching the stream
  int* sticky = nullptr;
};

namespace spb {

__device__ __forceinline__ bool trsv_strict(int c, int r, int upper) {
  return upper ? c > r : c < r;
}

// in-degree of every row (strict entries) and out-degree of every column (how many rows read x_k)
__global__ __launch_bounds__(256) void trsv_degree_kernel(int64_t m, const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ colind, int upper,
                                                          int32_t* __restrict__ indeg, int32_t* __restrict__ outdeg) {
  const int64_t row = (int64_t) blockIdx.x * 32 + threadIdx.x / 8;
  const int lane = threadIdx.x % 8;
  int cnt = 0;
  if (row < m)
    for (int p = rowptr[row] + lane; p < rowptr[row + 1]; p += 8) {
      const int c = colind[p];
      if (c >= 0 && c < m && trsv_strict(c, (int) row, upper)) {
        ++cnt;
        atomicAdd(&outdeg[c], 1);
      }
    }
  cnt = group_sum_c<8>(cnt);
  if (row < m && lane == 0)
    indeg[row] = cnt;
}

__global__ __launch_bounds__(256) void trsv_fill_adj_kernel(int64_t m, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ colind, int upper,
                                                            const int32_t* __restrict__ adj_ptr,
                                                            int32_t* __restrict__ cursor, int32_t* __restrict__ adj) {
  const int64_t row = (int64_t) blockIdx.x * 32 + threadIdx.x / 8;
  const int lane = threadIdx.x % 8;
  if (row >= m)
    return;
  for (int p = rowptr[row] + lane; p < rowptr[row + 1]; p += 8) {
    const int c = colind[p];
    if (c >= 0 && c < m && trsv_strict(c, (int) row, upper))
      adj[adj_ptr[c] + atomicAdd(&cursor[c], 1)] = (int32_t) row;
  }
}

// ---- levels by dependency polling (the default inspect) ---------------------------------------------------
// level(r) = 1 + max level of the rows r reads (0 without dependencies).  For the lower triangle every dependency has
// a smaller row index (larger for the upper one), so ONE self-scheduling kernel that hands out row blocks in index
// order can compute all levels: a lane owns a row, polls lev[c] (level + 1; 0 = not yet known) of each strict entry
// and publishes its own when all are known -- agent-scope 4-byte atomics, no adjacency lists, no transposed graph,
// no launch per level (Kahn's algorithm above needs all three: 17 ms at 4 M rows / 246 levels).  Blocks go to
// RUNNING workgroups in dependency order, so whatever a lane waits for is held by a running lane or already done;
// lanes of one wavefront that depend on each other make progress because the wave-uniform outer loop lets the
// producer store before the consumer polls again.  stat[0] = deepest level.
__global__ __launch_bounds__(256) void trsv_levels_poll_kernel(int64_t m, const int32_t* __restrict__ rowptr,
                                                               const int32_t* __restrict__ colind, int upper,
                                                               int32_t* __restrict__ lev,
                                                               int32_t* __restrict__ ticket, int32_t* __restrict__ stat,
                                                               int spin_limit) {
  // (four lanes per row instead of one -- a shorter hand-off chain through a row -- measured 4.2 vs 3.7 ms at 4 M
  // rows: the kernel is bound by head-of-line blocking of the index-ordered tickets, not by the walk of a row)
  __shared__ int s_ticket;
  const int64_t nblk = (m + 255) / 256;
  while (true) {
    __syncthreads();
    if (threadIdx.x == 0)
      s_ticket = atomicAdd(ticket, 1);
    __syncthreads();
    const int64_t t = s_ticket;
    if (t >= nblk)
      return;
    const int64_t blk = upper ? nblk - 1 - t : t;  // upper triangle: dependencies have larger indices
    // inside a block, too, lanes take the rows in dependency order (irrelevant for correctness, shortens the waits)
    const int64_t r = upper ? blk * 256 + 255 - threadIdx.x : blk * 256 + threadIdx.x;
    const bool live = r < m;
    int p = live ? rowptr[r] : 0;
    const int p1 = live ? rowptr[r + 1] : 0;
    int mx = 0;
    bool done = !live;
    int spins = 0;
    while (__any(!done)) {
      if (!done) {
        while (p < p1) {
          const int c = colind[p];
          if (c >= 0 && c < m && trsv_strict(c, (int) r, upper)) {
            const int v = __hip_atomic_load(&lev[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v == 0)
              break;  // not known yet: poll again on the next round
            mx = v > mx ? v : mx;
          }
          ++p;
        }
        if (p >= p1) {
          __hip_atomic_store(&lev[r], mx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          done = true;
        } else if (++spins > spin_limit) {
          stat[1] = 1;  // gave up (corrupt input or a scheduling assumption broken): the host falls back to Kahn
          done = true;
        } else {
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    // deepest level: one atomic per wavefront (one per row on a single address took 20 ms at 4 M rows)
    int wmax = live ? mx : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int other = __shfl_xor(wmax, o, SPB_WAVE);
      wmax = other > wmax ? other : wmax;
    }
    if ((threadIdx.x & 63) == 0)
      atomicMax(&stat[0], wmax);
  }
}

// rows per level: LDS histogram of the first 4 096 levels per workgroup (global atomics beyond), flushed once
__global__ __launch_bounds__(256) void trsv_level_hist_kernel(int64_t m, const int32_t* __restrict__ lev,
                                                              int32_t* __restrict__ hist) {
  __shared__ int lh[4096];
  for (int i = threadIdx.x; i < 4096; i += 256)
    lh[i] = 0;
  __syncthreads();
  const int64_t per = (m + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t) blockIdx.x * per, hi = (lo + per) < m ? (lo + per) : m;
  for (int64_t r = lo + threadIdx.x; r < hi; r += 256) {
    const int l = lev[r] - 1;
    if (l < 4096)
      atomicAdd(&lh[l], 1);
    else
      atomicAdd(&hist[l], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 256)
    if (lh[i])
      atomicAdd(&hist[i], lh[i]);
}

// order[level_ptr[level(r)] + k] = r: rows grouped by level (the order inside a level is irrelevant).  Every workgroup
// counts its chunk of rows per level in LDS, reserves its share of each level with ONE global atomic, and places the
// rows with LDS cursors (one global atomic per row on 246 hot counters took 11 ms at 4 M rows); levels beyond the
// first 4 096 use the global cursors directly.
__global__ __launch_bounds__(256) void trsv_place_rows_kernel(int64_t m, const int32_t* __restrict__ lev,
                                                              const int32_t* __restrict__ level_ptr,
                                                              int32_t* __restrict__ cursor, int32_t* __restrict__ order) {
  __shared__ int lh[4096];
  __shared__ int lbase[4096];
  for (int i = threadIdx.x; i < 4096; i += 256)
    lh[i] = 0;
  __syncthreads();
  const int64_t per = (m + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t) blockIdx.x * per, hi = (lo + per) < m ? (lo + per) : m;
  for (int64_t r = lo + threadIdx.x; r < hi; r += 256) {
    const int l = lev[r] - 1;
    if (l < 4096)
      atomicAdd(&lh[l], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 256) {
    const int c = lh[i];
    if (c)
      lbase[i] = level_ptr[i] + atomicAdd(&cursor[i], c);
    lh[i] = 0;
  }
  __syncthreads();
  for (int64_t r = lo + threadIdx.x; r < hi; r += 256) {
    const int l = lev[r] - 1;
    const int pos = l < 4096 ? lbase[l] + atomicAdd(&lh[l], 1) : level_ptr[l] + atomicAdd(&cursor[l], 1);
    order[pos] = (int32_t) r;
  }
}

// level 0: rows without dependencies.  state[0] = tail of `order`.
__global__ __launch_bounds__(256) void trsv_roots_kernel(int64_t m, const int32_t* __restrict__ indeg,
                                                         int32_t* __restrict__ order, int32_t* __restrict__ state) {
  const int64_t row = (int64_t) blockIdx.x * 256 + threadIdx.x;
  const bool root = row < m && indeg[row] == 0;
  const unsigned long long mask = __ballot(root);
  if (mask == 0)
    return;
  const int lane = threadIdx.x & 63;
  const int leader = __builtin_ctzll(mask);
  int base = 0;
  if (lane == leader)
    base = atomicAdd(&state[0], (int) __popcll(mask));
  base = __shfl(base, leader);
  if (root)
    order[base + (int) __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t) row;
}

// Wide levels: rows order[f0..f1) release their dependents; newly free rows are appended.
// A batch of wide levels runs without host round trips: the frontier bounds live in
// state = {tail, f0, f1, n_levels, tickets}; every workgroup reads them (they were written by the previous
// kernel on the stream), walks the frontier with a grid stride, and the last workgroup to finish publishes
// the next level.  A narrow or empty frontier is left alone (the host hands it to the single-workgroup
// kernel below), so launches enqueued past the end of a run of wide levels are no-ops.
__global__ __launch_bounds__(256) void trsv_advance_dev_kernel(const int32_t* __restrict__ adj_ptr,
                                                               const int32_t* __restrict__ adj,
                                                               int32_t* __restrict__ indeg, int32_t* __restrict__ order,
                                                               int32_t* __restrict__ state,
                                                               int32_t* __restrict__ level_ptr, int narrow) {
  const int f0 = state[1], f1 = state[2];
  const bool wide = f1 - f0 >= narrow;
  if (wide) {
    const int lane = threadIdx.x % 8;
    for (int idx = f0 + blockIdx.x * 32 + threadIdx.x / 8; idx < f1; idx += gridDim.x * 32) {
      const int r = order[idx];
      for (int q = adj_ptr[r] + lane; q < adj_ptr[r + 1]; q += 8) {
        const int j = adj[q];
        // one reservation on the shared tail per wavefront and iteration, not one per freed row (the single
        // hot address cost ~3 ns per atomic: 50 us for a level of 16 K rows)
        const bool freed = atomicSub(&indeg[j], 1) == 1;
        const unsigned long long fm = __ballot(freed);
        if (fm) {
          const int wl = threadIdx.x & 63;
          const int leader = __builtin_ctzll(fm);
          int base = 0;
          if (wl == leader)
            base = atomicAdd(&state[0], (int) __popcll(fm));
          base = __shfl(base, leader);
          if (freed)
            order[base + (int) __popcll(fm & ((1ull << wl) - 1ull))] = j;
        }
      }
    }
  }
  // every workgroup has read f0/f1 before the state can change: the update below happens only after ALL
  // workgroups have taken their ticket, i.e. after they passed the reads above
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const int ticket = atomicAdd(&state[4], 1);
    if (ticket == (int) gridDim.x - 1) {
      state[4] = 0;
      if (wide) {
        const int nl = state[3];
        level_ptr[nl] = f0;
        state[1] = f1;
        state[2] = atomicAdd(&state[0], 0);  // the tail, read where the other workgroups' atomics landed
        state[3] = nl + 1;
      }
    }
  }
}

// Single workgroup: keeps taking levels while they are narrow.  state = {tail, f0, f1, n_levels};
// level_ptr[l] = first position of level l in `order`.  Stops when the current frontier is empty,
// wide (>= TRSV_NARROW rows: the host spreads it over the chip), or `max_levels` were taken.
__global__ __launch_bounds__(TRSV_BLOCK_THREADS) void trsv_bfs_block_kernel(const int32_t* __restrict__ adj_ptr,
                                                                          const int32_t* __restrict__ adj,
                                                                          int32_t* __restrict__ indeg,
                                                                          int32_t* __restrict__ order,
                                                                          int32_t* __restrict__ state,
                                                                          int32_t* __restrict__ level_ptr,
                                                                          int max_levels) {
  __shared__ int s_tail;
  int f0 = state[1], f1 = state[2], nl = state[3];
  if (threadIdx.x == 0)
    s_tail = state[0];
  __syncthreads();
  int taken = 0;
  while (f1 > f0 && f1 - f0 < TRSV_NARROW && taken < max_levels) {
    for (int idx = f0 + threadIdx.x / 8; idx < f1; idx += TRSV_BLOCK_THREADS / 8) {
      // entries of order[] written by this workgroup one level earlier: agent-scope accesses, so a
      // line of order[] cached before the write can never be served stale
      const int r = __hip_atomic_load(&order[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int q = adj_ptr[r] + (threadIdx.x & 7); q < adj_ptr[r + 1]; q += 8) {
        const int j = adj[q];
        if (atomicSub(&indeg[j], 1) == 1)
          __hip_atomic_store(&order[atomicAdd(&s_tail, 1)], j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0)
      level_ptr[nl] = f0;
    f0 = f1;
    f1 = s_tail;
    ++nl;
    ++taken;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    state[0] = s_tail;
    state[1] = f0;
    state[2] = f1;
    state[3] = nl;
  }
}

// x_r for one row, computed by a group of G lanes (all lanes of the group return the same values).
// Entries whose column lies outside [0, m) are ignored, as the inspect kernels ignore them; a row without a
// stored diagonal divides by alpha * 0 (the reference would reuse the previous row's diagonal: undefined input).
template <typename T, int G>
__device__ __forceinline__ void trsv_row(int r, int lane, const int32_t* __restrict__ rowptr,
                                         const int32_t* __restrict__ colind, const T* __restrict__ values, T alpha,
                                         const T* __restrict__ b, T* x, int upper, int unit, int m) {
  T dot = T(0), dval = T(0);
  int dpos = -1;
  const int p1 = rowptr[r + 1];
  const T br = b[r];  // (issued with the row's other loads, not after the reduction)
  for (int p = rowptr[r] + lane; p < p1; p += G) {
    const int c = colind[p];
    const T a = values[p];
    if (c >= 0 && c < m && trsv_strict(c, r, upper))
      dot += a * __hip_atomic_load(&x[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (c == r)
      dpos = p, dval = a;  // the last stored diagonal entry wins (triangular_solve_impl.hpp:64-66,81-83)
  }
  // the diagonal VALUE travels with its position through the reduction: a load of values[dpos] afterwards was one more
  // memory round trip on the critical path of every level
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) {
    dot += __shfl_xor(dot, o, SPB_WAVE);
    const int other = __shfl_xor(dpos, o, SPB_WAVE);
    const T oval = __shfl_xor(dval, o, SPB_WAVE);
    if (other > dpos)
      dpos = other, dval = oval;
  }
  if (lane == 0) {
    T v = br - alpha * dot;
    if (!unit)
      v = v / (alpha * (dpos >= 0 ? dval : T(0)));
    __hip_atomic_store(&x[r], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- the whole solve in ONE cooperative launch ------------------------------------------------------------
// Every level in one kernel, a grid barrier between levels (a workgroup barrier inside a run of NARROW levels,
// which workgroup 0 walks alone while the others wait at the next grid barrier).  What a launch per level pays
// per level -- drain, dispatch, then FOUR dependent loads (order -> rowptr -> colind/values -> x) -- becomes one
// barrier and ONE load latency: the three loads that do not depend on x are software-pipelined across levels
// (while the barrier after level l is pending a lane group loads colind/values for its rows of level l + 1, rowptr
// for level l + 2 and order for level l + 3; after the barrier only the x gather is left).  A lane group owns R row
// slots per level, so one pipelined pass covers gridDim * (1024 / G) * R rows; wider levels take further plain passes.
// x crosses workgroups (and XCDs, whose L2s are not coherent) through agent-scope stores and loads only.
// Launched with hipLaunchCooperativeKernel: all workgroups are resident, so the barrier cannot deadlock; the
// spin is bounded all the same (status[0] = 1 and every workgroup leaves).
// Measured, 4 M rows / 246 levels (158 grid barriers): barrier 2.3 us (two-level arrival; all-to-all flags 4.1 us,
// acq_rel arrivals or a release fence per wavefront 10-60 us: every one is an L2 write-back), pipeline loads 1.9 us
// per level when not overlapped with the barrier, x gather + store 2.2 us.
#define TRSV_COOP_THREADS 1024

// Grid barrier number n, first half.  x is written with agent-scope (write-through) stores: once a wavefront's
// stores are acknowledged (vmcnt = 0) they are at the coherent level, and the consumers read x with agent-scope
// loads, so the arrival itself needs no cache maintenance (relaxed).  Two-level arrival: 8 group counters
// (workgroup w -> group w % 8, the XCD it runs on; bar[32 * (1 + g)]), the last of a group arrives at bar[0].
__device__ __forceinline__ void trsv_barrier_arrive(unsigned* bar, unsigned n, int* s_flags) {
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned gid = blockIdx.x & 7u, ngroups = gridDim.x < 8u ? gridDim.x : 8u;
    const unsigned gsize = (gridDim.x - gid + 7u) >> 3;
    unsigned last = 0;
    if (__hip_atomic_fetch_add(&bar[32 * (1 + gid)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == n * gsize)
      last = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == n * ngroups;
    s_flags[1] = (int) last;
  }
}
// Second half: the last workgroup to arrive releases the others through one flag LINE per workgroup
// (bar[32 * (9 + w)]): 256 workgroups polling one address saturate its memory channel (measured 45 us per barrier).
__device__ __forceinline__ bool trsv_barrier_wait(unsigned* bar, unsigned n, int spin_limit, int* status, int* sticky, int* s_flags) {
  __syncthreads();
  if (s_flags[1]) {
    for (unsigned w = threadIdx.x; w < gridDim.x; w += blockDim.x)
      __hip_atomic_store(&bar[32 * (9 + w)], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(&bar[32 * (9 + blockIdx.x)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > spin_limit) {
        status[0] = 1;
        if (sticky)
          __hip_atomic_store(sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        s_flags[0] = 1;
        break;
      }
    }
  }
  __syncthreads();
  return s_flags[0] == 0;
}

template <typename T, int G, int R>
__global__ __launch_bounds__(TRSV_COOP_THREADS) void trsv_coop_kernel(int n_levels, int narrow,
                                                                      const int32_t* __restrict__ level_ptr,
                                                                      const int32_t* __restrict__ order,
                                                                      const int32_t* __restrict__ rowptr,
                                                                      const int32_t* __restrict__ colind,
                                                                      const T* __restrict__ values, T alpha,
                                                                      const T* __restrict__ b, T* x, int upper,
                                                                      int unit, int m, unsigned* bar, int* status, int* sticky,
                                                                      int spin_limit, int dbg) {
  constexpr int RPB = TRSV_COOP_THREADS / G;  // rows per workgroup and slot
  const int gl = threadIdx.x % G, grp = threadIdx.x / G;
  __shared__ int s_flags[2];  // {abort, last arrival}
  if (threadIdx.x == 0)
    s_flags[0] = 0;
  __syncthreads();
  // first row position of this lane group in level L (>= the level's end: none), the level's end and slot stride
  auto first_slot = [&](int L, int* f1, int* stride) -> int {
    if (L < 0 || L >= n_levels) {
      *f1 = 0;
      *stride = 1;
      return 0;
    }
    const int f0 = level_ptr[L];
    *f1 = level_ptr[L + 1];
    if (*f1 - f0 >= narrow) {
      *stride = (int) gridDim.x * RPB;
      return f0 + (int) blockIdx.x * RPB + grp;
    }
    *stride = RPB;
    return blockIdx.x == 0 ? f0 + grp : *f1;
  };
  // pipeline registers per slot: A = row known, B = row + entry range, C = first two entries per lane loaded
  int rA[R], rB[R], pB[R], qB[R], rC[R], pC[R], qC[R], c0[R], c1[R];
  T v0[R], v1[R], bC[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    rA[k] = rB[k] = rC[k] = -1;
    pB[k] = qB[k] = pC[k] = qC[k] = 0;
    c0[k] = c1[k] = -1;
    v0[k] = v1[k] = bC[k] = T(0);
  }
  // one pipeline step: C <- entries of B's rows, B <- entry range of A's rows, A <- rows of level L (all loads
  // independent of each other and of x)
  auto advance = [&](int L) {
    int f1, stride;
    const int slot = first_slot(L, &f1, &stride);
#pragma unroll
    for (int k = 0; k < R; ++k) {
      rC[k] = rB[k];
      pC[k] = pB[k];
      qC[k] = qB[k];
      c0[k] = c1[k] = -1;
      v0[k] = v1[k] = bC[k] = T(0);
      if (rB[k] >= 0) {
        bC[k] = b[rB[k]];  // (with the entries: not after the reduction, on the level's critical path)
        if (pB[k] < qB[k]) {
          c0[k] = colind[pB[k]];
          v0[k] = values[pB[k]];
        }
        if (pB[k] + G < qB[k]) {
          c1[k] = colind[pB[k] + G];
          v1[k] = values[pB[k] + G];
        }
      }
      rB[k] = rA[k];
      pB[k] = qB[k] = 0;
      if (rA[k] >= 0) {
        pB[k] = rowptr[rA[k]] + gl;
        qB[k] = rowptr[rA[k] + 1];
      }
      const int idx = slot + k * stride;
      rA[k] = idx < f1 ? order[idx] : -1;
    }
  };
  advance(0);
  advance(1);
  advance(2);
  unsigned n_bar = 0;
  for (int l = 0; l < n_levels; ++l) {
    // ---- level l: gather x for the pipelined rows, finish them ----
    if (!(dbg & 4)) {
      bool s0[R], s1[R];
      T x0[R], x1[R];
#pragma unroll
      for (int k = 0; k < R; ++k) {
        s0[k] = rC[k] >= 0 && c0[k] >= 0 && c0[k] < m && trsv_strict(c0[k], rC[k], upper);
        s1[k] = rC[k] >= 0 && c1[k] >= 0 && c1[k] < m && trsv_strict(c1[k], rC[k], upper);
        x0[k] = x1[k] = T(0);
        if (s0[k])
          x0[k] = __hip_atomic_load(&x[c0[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s1[k])
          x1[k] = __hip_atomic_load(&x[c1[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int k = 0; k < R; ++k) {
        T dot = T(0), dval = T(0);
        int dpos = -1;
        if (s0[k])
          dot += v0[k] * x0[k];
        else if (rC[k] >= 0 && c0[k] == rC[k])
          dpos = pC[k], dval = v0[k];
        if (s1[k])
          dot += v1[k] * x1[k];
        else if (rC[k] >= 0 && c1[k] == rC[k])
          dpos = pC[k] + G, dval = v1[k];
        for (int p = pC[k] + 2 * G; p < qC[k]; p += G) {  // rows longer than 2 G entries
          const int c = colind[p];
          const T a = values[p];
          if (c >= 0 && c < m && trsv_strict(c, rC[k], upper))
            dot += a * __hip_atomic_load(&x[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else if (c == rC[k])
            dpos = p, dval = a;
        }
        // (the diagonal value travels with its position: no load of values[dpos] after the reduction)
#pragma unroll
        for (int o = G >> 1; o > 0; o >>= 1) {
          dot += __shfl_xor(dot, o, SPB_WAVE);
          const int other = __shfl_xor(dpos, o, SPB_WAVE);
          const T oval = __shfl_xor(dval, o, SPB_WAVE);
          if (other > dpos)
            dpos = other, dval = oval;
        }
        if (rC[k] >= 0 && gl == 0) {
          T v = bC[k] - alpha * dot;
          if (!unit)
            v = v / (alpha * (dpos >= 0 ? dval : T(0)));
          __hip_atomic_store(&x[rC[k]], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    {  // rows of a level wider than the pipelined pass
      int f1, stride;
      const int slot = first_slot(l, &f1, &stride);
      if (!(dbg & 8))
      for (int idx = slot + R * stride; idx < f1; idx += stride)
        trsv_row<T, G>(order[idx], gl, rowptr, colind, values, alpha, b, x, upper, unit, m);
    }
    if (l + 1 == n_levels)
      break;
    // ---- hand level l over to level l + 1; the pipeline advances while the barrier is pending ----
    const int w0 = level_ptr[l + 1] - level_ptr[l], w1 = level_ptr[l + 2] - level_ptr[l + 1];
    if (w0 < narrow && w1 < narrow) {  // inside a narrow run: workgroup 0 alone
      advance(l + 3);
      if (blockIdx.x == 0) {
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this lane's write-through x stores are acknowledged
        __syncthreads();
      }
    } else if (dbg & 2) {
      advance(l + 3);
    } else {
      ++n_bar;
      trsv_barrier_arrive(bar, n_bar, s_flags);
      advance(l + 3);
      if (!trsv_barrier_wait(bar, n_bar, spin_limit, status, sticky, s_flags))
        return;
    }
  }
}

// one wide level: rows order[f0..f1), G lanes per row
template <typename T, int G>
__global__ __launch_bounds__(256) void trsv_level_kernel(int f0, int f1, const int32_t* __restrict__ order,
                                                         const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ colind,
                                                         const T* __restrict__ values, T alpha,
                                                         const T* __restrict__ b, T* x, int upper, int unit, int m) {
  const int idx = f0 + blockIdx.x * (256 / G) + threadIdx.x / G;
  if (idx >= f1)
    return;
  trsv_row<T, G>(order[idx], threadIdx.x % G, rowptr, colind, values, alpha, b, x, upper, unit, m);
}

// levels [l0, l1), all narrow: one workgroup, a barrier between levels
template <typename T, int G>
__global__ __launch_bounds__(TRSV_BLOCK_THREADS) void trsv_chain_kernel(int l0, int l1,
                                                                       const int32_t* __restrict__ level_ptr,
                                                                       const int32_t* __restrict__ order,
                                                                       const int32_t* __restrict__ rowptr,
                                                                       const int32_t* __restrict__ colind,
                                                                       const T* __restrict__ values, T alpha,
                                                                       const T* __restrict__ b, T* x, int upper,
                                                                       int unit, int m) {
  for (int l = l0; l < l1; ++l) {
    const int f0 = level_ptr[l], f1 = level_ptr[l + 1];
    for (int idx = f0 + threadIdx.x / G; idx < f1; idx += TRSV_BLOCK_THREADS / G)
      trsv_row<T, G>(order[idx], threadIdx.x % G, rowptr, colind, values, alpha, b, x, upper, unit, m);
    __threadfence();  // x of this level must be visible to the whole workgroup before the next one
    __syncthreads();
  }
}

template <typename T, int G>
static int trsv_solve_typed(spblas_gfx950_handle_t h, spblas_gfx950_trsv_s* pl, const int32_t* rowptr,
                            const int32_t* colind, const T* values, T alpha, const T* b, T* x) {
  hipStream_t s = h->stream;
  const int upper = pl->uplo == SPBLAS_GFX950_UPPER, unit = pl->diag == SPBLAS_GFX950_DIAG_UNIT;
  const int m = (int) pl->m;
  const size_t ng = pl->groups.size();
  const int cus = h->num_cus > 0 ? h->num_cus : 256;
  // [ng + 1] = status word (the leading words are unused since the self-scheduling form left), then (32-int aligned) the
  // grid barrier: one line for the arrival counter and one release flag line per workgroup
  const size_t bar_off = (ng + 2 + 31) / 32 * 32, ctl_ints = bar_off + 32 * (size_t) (9 + 2 * cus);
  // A solve recorded into a graph (hipStreamBeginCapture / torch.cuda.graph) is replayed with the arguments it was
  // recorded with: no cooperative launch (the kernel node does not carry the co-residency guarantee of the launch: replays
  // were seen to leave rows unsolved), no allocation (the first solve of a plan must run outside the capture).
  const bool capturing = stream_capturing(s);
  if (capturing && !pl->tickets)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  if (!pl->tickets) {
    int rc = dev_alloc((void**) &pl->tickets, ctl_ints * 4, s);
    if (rc)
      return rc;
  }
  // A solve of this plan whose grid barrier gave up (x incomplete) is reported by the NEXT call on the plan at the latest:
  // the kernel set the pinned word, no stream work is needed to see it.  The word is cleared by the report.
  if (pl->sticky && __atomic_load_n(pl->sticky, __ATOMIC_RELAXED) != 0) {
    __atomic_store_n(pl->sticky, 0, __ATOMIC_RELAXED);
    return hip_fail(hipErrorLaunchTimeOut);
  }
  if (!pl->sticky && !capturing) {
    void* hp = nullptr;
    if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess) {
      pl->sticky = static_cast<int*>(hp);
      *pl->sticky = 0;
    } else {
      (void) hipGetLastError();  // no pinned word: spblas_gfx950_sptrsv_status (device word) remains
    }
  }
  SPB_HIP(hipMemsetAsync(pl->tickets, 0, ctl_ints * 4, s));
  int* status = pl->tickets + ng + 1;
  int* sticky = pl->sticky;
  const int spin_limit = env_int("SPBLAS_GFX950_TRSV_SPIN_LIMIT", 1 << 22);  // ~ seconds of polling
  // the whole solve as one cooperative launch (default when the device offers it and no narrow run is so long
  // that the waiting workgroups could exhaust their bounded spin: 4096 levels ~ 10 ms)
  if (pl->coop_ok && ng > 1 && !capturing) {
    int n_levels = (int) pl->h_level_ptr.size() - 1, narrow = pl->narrow;
    int wgs = env_int("SPBLAS_GFX950_TRSV_COOP_WGS", 1);
    int occ = 0;
    constexpr int R = G <= 8 ? 4 : (G <= 16 ? 2 : 1);  // row slots per lane group in the pipelined pass
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, trsv_coop_kernel<T, G, R>, TRSV_COOP_THREADS, 0) != hipSuccess || occ < 1)
      occ = 0;
    if (wgs > occ)
      wgs = occ;
    if (wgs > 2)
      wgs = 2;
    if (wgs >= 1) {
      unsigned* bar = reinterpret_cast<unsigned*>(pl->tickets + bar_off);
      const int32_t* lp = pl->level_ptr;
      const int32_t* ord = pl->order;
      int mm = m, up = upper, un = unit, sl = spin_limit, bm = env_int("SPBLAS_GFX950_TRSV_DBG", 0);
      void* args[] = {&n_levels, &narrow, (void*) &lp, (void*) &ord, (void*) &rowptr, (void*) &colind, (void*) &values,
                      &alpha, (void*) &b, (void*) &x, &up, &un, &mm, &bar, &status, &sticky, &sl, &bm};
      int grid = env_int("SPBLAS_GFX950_TRSV_COOP_GRID", 0);
      if (grid < 1 || grid > cus * wgs)
        grid = cus * wgs;
      const hipError_t ce = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&trsv_coop_kernel<T, G, R>), dim3((unsigned) grid),
                                                       dim3(TRSV_COOP_THREADS), args, 0, s);
      if (ce == hipSuccess)
        return SPBLAS_GFX950_STATUS_SUCCESS;
      (void) hipGetLastError();  // not launched (e.g. the stream is being captured): one launch per group below
    }
  }
  for (size_t gi = 0; gi < ng; ++gi) {
    const auto& g = pl->groups[gi];
    if (g.wide) {
      const int f0 = pl->h_level_ptr[g.l0], f1 = pl->h_level_ptr[g.l0 + 1];
      hipLaunchKernelGGL((trsv_level_kernel<T, G>), dim3((unsigned) cdiv(f1 - f0, 256 / G)), dim3(256), 0, s, f0, f1,
                         pl->order, rowptr, colind, values, alpha, b, x, upper, unit, m);
    } else {
      hipLaunchKernelGGL((trsv_chain_kernel<T, G>), dim3(1), dim3(TRSV_BLOCK_THREADS), 0, s, g.l0, g.l1,
                         pl->level_ptr, pl->order, rowptr, colind, values, alpha, b, x, upper, unit, m);
    }
  }
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

template <typename T>
static int trsv_solve_lanes(spblas_gfx950_handle_t h, spblas_gfx950_trsv_s* pl, const int32_t* rowptr,
                            const int32_t* colind, const T* values, T alpha, const T* b, T* x) {
  switch (pl->lanes) {
    case 4: return trsv_solve_typed<T, 4>(h, pl, rowptr, colind, values, alpha, b, x);
    case 16: return trsv_solve_typed<T, 16>(h, pl, rowptr, colind, values, alpha, b, x);
    case 64: return trsv_solve_typed<T, 64>(h, pl, rowptr, colind, values, alpha, b, x);
    default: return trsv_solve_typed<T, 8>(h, pl, rowptr, colind, values, alpha, b, x);
  }
}

} // namespace spb

using namespace spb;

extern "C" {

int spblas_gfx950_sptrsv_create(spblas_gfx950_handle_t handle, spblas_gfx950_trsv_t* plan_out, int64_t m, int64_t nnz,
                                const int32_t* rowptr, const int32_t* colind, int uplo, int diag) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (stream_capturing(handle->stream))  // inspect-class call: sizes its output on the host, never part of a graph
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  if (!plan_out || !rowptr || (nnz > 0 && !colind))
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (m < 0 || nnz < 0 || m >= INT32_MAX || nnz > INT32_MAX)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  if ((uplo != SPBLAS_GFX950_LOWER && uplo != SPBLAS_GFX950_UPPER) ||
      (diag != SPBLAS_GFX950_DIAG_EXPLICIT && diag != SPBLAS_GFX950_DIAG_UNIT))
    return SPBLAS_GFX950_STATUS_INVALID_VALUE;
  auto* pl = new (std::nothrow) spblas_gfx950_trsv_s();
  if (!pl)
    return SPBLAS_GFX950_STATUS_ALLOC_FAILED;
  pl->m = m;
  pl->nnz = nnz;
  pl->uplo = uplo;
  pl->diag = diag;
  const double avg = m > 0 ? (double) nnz / (double) m : 0.0;
  pl->lanes = avg > 96 ? 64 : (avg > 24 ? 16 : (avg > 6 ? 8 : 4));
  *plan_out = pl;
  if (m == 0)
    return SPBLAS_GFX950_STATUS_SUCCESS;

  hipStream_t s = handle->stream;
  const int upper = uplo == SPBLAS_GFX950_UPPER;
  int32_t *indeg = nullptr, *adj_ptr = nullptr, *cursor = nullptr, *adj = nullptr, *state = nullptr,
          *level_ptr = nullptr;
  long long* partials = nullptr;
  int rc = SPBLAS_GFX950_STATUS_SUCCESS;
  auto fail = [&](int code) {
    (void) hipStreamSynchronize(s);
    dev_free(indeg, s);
    dev_free(adj_ptr, s);
    dev_free(cursor, s);
    dev_free(adj, s);
    dev_free(state, s);
    dev_free(partials, s);
    dev_free(level_ptr, s);
    dev_free(pl->order, s);
    dev_free(pl->level_ptr, s);
    delete pl;
    *plan_out = nullptr;
    return code;
  };
  if ((rc = dev_alloc((void**) &indeg, (size_t) m * 4, s)) || (rc = dev_alloc((void**) &adj_ptr, (size_t) (m + 1) * 4, s)) ||
      (rc = dev_alloc((void**) &cursor, (size_t) m * 4, s)) ||
      (rc = dev_alloc((void**) &adj, (size_t) (nnz > 0 ? nnz : 1) * 4, s)) ||
      (rc = dev_alloc((void**) &state, 8 * 4, s)) ||
      (rc = dev_alloc((void**) &partials, (size_t) (cdiv(m, 2048) + 1) * sizeof(long long), s)) ||
      (rc = dev_alloc((void**) &level_ptr, (size_t) (m + 1) * 4, s)) ||
      (rc = dev_alloc((void**) &pl->order, (size_t) m * 4, s)))
    return fail(rc);
  std::vector<int32_t>& lp = pl->h_level_ptr;
  int32_t n_levels = 0;
  hipError_t e = hipSuccess;
  bool have_levels = false;
  if (env_int("SPBLAS_GFX950_TRSV_KAHN", 0) == 0) {
    // default: levels by dependency polling.  indeg doubles as lev[], adj_ptr as hist[], cursor as the placement
    // cursors, state = {ticket, -, -, -, deepest level, gave-up flag}
    e = hipMemsetAsync(indeg, 0, (size_t) m * 4, s);
    if (e == hipSuccess)
      e = hipMemsetAsync(adj_ptr, 0, (size_t) (m + 1) * 4, s);
    if (e == hipSuccess)
      e = hipMemsetAsync(cursor, 0, (size_t) m * 4, s);
    if (e == hipSuccess)
      e = hipMemsetAsync(state, 0, 32, s);
    if (e != hipSuccess)
      return fail(hip_fail(e));
    const int cus = handle->num_cus > 0 ? handle->num_cus : 256;
    const int grid = (int) std::min<int64_t>((int64_t) cus * 8, cdiv(m, 256));
    hipLaunchKernelGGL(trsv_levels_poll_kernel, dim3((unsigned) grid), dim3(256), 0, s, m, rowptr, colind, upper, indeg,
                       state, state + 4, env_int("SPBLAS_GFX950_TRSV_SPIN_LIMIT", 1 << 22));
    hipLaunchKernelGGL(trsv_level_hist_kernel, dim3((unsigned) std::min<int64_t>(1024, cdiv(m, 256))), dim3(256), 0, s, m,
                       indeg, adj_ptr);
    int32_t st2[2] = {0, 0};
    if ((e = hipMemcpyAsync(st2, state + 4, 8, hipMemcpyDeviceToHost, s)) != hipSuccess ||
        (e = hipStreamSynchronize(s)) != hipSuccess)
      return fail(hip_fail(e));
    if (st2[1] == 0) {
      n_levels = st2[0] + 1;
      // hist -> level_ptr (exclusive scan over the levels), then the rows are placed level by level
      scan_counts_i32(s, n_levels, adj_ptr, partials);
      hipLaunchKernelGGL(trsv_place_rows_kernel, dim3((unsigned) std::min<int64_t>(1024, cdiv(m, 256))), dim3(256), 0, s, m,
                         indeg, adj_ptr, cursor, pl->order);
      lp.resize((size_t) n_levels + 1);
      if ((e = hipMemcpyAsync(lp.data(), adj_ptr, (size_t) (n_levels + 1) * 4, hipMemcpyDeviceToHost, s)) != hipSuccess ||
          (e = hipMemcpyAsync(level_ptr, adj_ptr, (size_t) (n_levels + 1) * 4, hipMemcpyDeviceToDevice, s)) != hipSuccess ||
          (e = hipStreamSynchronize(s)) != hipSuccess)
        return fail(hip_fail(e));
      if (lp[(size_t) n_levels] != (int32_t) m)
        return fail(SPBLAS_GFX950_STATUS_INVALID_VALUE);
      have_levels = true;
    }
  }
  if (!have_levels) {
  e = hipMemsetAsync(adj_ptr, 0, (size_t) (m + 1) * 4, s);
  if (e == hipSuccess)
    e = hipMemsetAsync(cursor, 0, (size_t) m * 4, s);
  if (e == hipSuccess)
    e = hipMemsetAsync(state, 0, 32, s);
  if (e != hipSuccess)
    return fail(hip_fail(e));
  hipLaunchKernelGGL(trsv_degree_kernel, dim3((unsigned) cdiv(m, 32)), dim3(256), 0, s, m, rowptr, colind, upper, indeg,
                     adj_ptr);
  scan_counts_i32(s, m, adj_ptr, partials);
  hipLaunchKernelGGL(trsv_fill_adj_kernel, dim3((unsigned) cdiv(m, 32)), dim3(256), 0, s, m, rowptr, colind, upper,
                     adj_ptr, cursor, adj);
  hipLaunchKernelGGL(trsv_roots_kernel, dim3((unsigned) cdiv(m, 256)), dim3(256), 0, s, m, indeg, pl->order, state);
  if ((e = hipGetLastError()) != hipSuccess)
    return fail(hip_fail(e));

  // Kahn's algorithm.  Host view of state = {tail, f0, f1, n_levels}.
  int32_t st[4] = {0, 0, 0, 0};
  if ((e = hipMemcpyAsync(st, state, 4, hipMemcpyDeviceToHost, s)) != hipSuccess ||
      (e = hipStreamSynchronize(s)) != hipSuccess)
    return fail(hip_fail(e));
  st[1] = 0;
  st[2] = st[0];
  st[3] = 0;
  // Wide levels are enqueued in batches (the kernels find the frontier in `state` themselves), narrow ones go
  // to the single-workgroup kernel; the host looks at the state once per batch instead of once per level
  // (246 levels at 4 M rows: 104 synchronisations before, 18.8 ms of inspect).
  if ((e = hipMemcpyAsync(state, st, 16, hipMemcpyHostToDevice, s)) != hipSuccess)
    return fail(hip_fail(e));
  const int adv_grid = (int) (cdiv(m, 32) < 512 ? cdiv(m, 32) : 512);  // also the number of ticket atomics per level
  const int batch = 16;
  while (st[2] > st[1]) {
    if (st[2] - st[1] >= TRSV_NARROW) {
      for (int k = 0; k < batch; ++k)
        hipLaunchKernelGGL(trsv_advance_dev_kernel, dim3((unsigned) adv_grid), dim3(256), 0, s, adj_ptr, adj, indeg,
                           pl->order, state, level_ptr, (int) TRSV_NARROW);
    } else {  // narrow levels: one workgroup takes as many as it can
      hipLaunchKernelGGL(trsv_bfs_block_kernel, dim3(1), dim3(TRSV_BLOCK_THREADS), 0, s, adj_ptr, adj, indeg, pl->order,
                         state, level_ptr, (int) (m + 1));
    }
    if ((e = hipMemcpyAsync(st, state, 16, hipMemcpyDeviceToHost, s)) != hipSuccess ||
        (e = hipStreamSynchronize(s)) != hipSuccess)
      return fail(hip_fail(e));
  }
  n_levels = st[3];
  if (st[0] != (int32_t) m)  // cannot happen for a strict triangle; guards against corrupt input
    return fail(SPBLAS_GFX950_STATUS_INVALID_VALUE);
  lp.resize((size_t) n_levels + 1);
  if (n_levels > 0 && (e = hipMemcpyAsync(lp.data(), level_ptr, (size_t) n_levels * 4, hipMemcpyDeviceToHost, s)) != hipSuccess)
    return fail(hip_fail(e));
  if ((e = hipStreamSynchronize(s)) != hipSuccess)
    return fail(hip_fail(e));
  lp[n_levels] = (int32_t) m;
  if ((e = hipMemcpyAsync(level_ptr, lp.data(), (size_t) (n_levels + 1) * 4, hipMemcpyHostToDevice, s)) != hipSuccess ||
      (e = hipStreamSynchronize(s)) != hipSuccess)
    return fail(hip_fail(e));
  }  // Kahn fallback
  pl->level_ptr = level_ptr;
  level_ptr = nullptr;
  // launch groups of the solve: every level of >= `narrow` rows is a group of its own (one grid-wide step), a run of
  // narrower levels ONE single-workgroup group
  const int narrow = env_int("SPBLAS_GFX950_TRSV_NARROW", 128);
  for (int32_t l = 0; l < n_levels;) {
    const bool wide = lp[l + 1] - lp[l] >= narrow;
    int32_t e1 = l + 1;
    if (!wide)
      while (e1 < n_levels && lp[e1 + 1] - lp[e1] < narrow)
        ++e1;
    for (int32_t q = l; q < e1; ++q)
      pl->max_width = std::max<int64_t>(pl->max_width, lp[q + 1] - lp[q]);
    pl->groups.push_back({l, e1, wide ? 1 : 0});
    l = e1;
  }
  pl->narrow = narrow;
  {
    int longest_run = 0, coop_attr = 0, dev = 0;
    for (const auto& g : pl->groups)
      if (!g.wide)
        longest_run = std::max(longest_run, g.l1 - g.l0);
    (void) hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&coop_attr, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess)
      coop_attr = 0;
    // An HSA tool that intercepts queues (rocprofv3 of ROCm 7.2) crashes in its exit handlers once the process has
    // used the cooperative queue (SIGSEGV after the tool has written its output): with such a tool loaded the
    // default falls back to one launch per level; SPBLAS_GFX950_TRSV_COOP=1 forces the cooperative kernel anyway
    // (that is how profiles/ shows it), =0 switches it off.
    const char* tool = std::getenv("ROCP_TOOL_LIBRARIES");
    const char* tool2 = std::getenv("HSA_TOOLS_LIB");
    const bool intercepted = (tool && *tool) || (tool2 && *tool2);
    pl->coop_ok = coop_attr != 0 && env_int("SPBLAS_GFX950_TRSV_COOP", intercepted ? 0 : 1) != 0 &&
                  longest_run <= env_int("SPBLAS_GFX950_TRSV_COOP_MAX_RUN", 4096);
  }
  dev_free(indeg, s);
  dev_free(adj_ptr, s);
  dev_free(cursor, s);
  dev_free(adj, s);
  dev_free(state, s);
  dev_free(partials, s);
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_sptrsv_destroy(spblas_gfx950_handle_t handle, spblas_gfx950_trsv_t plan) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!plan)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  dev_free(plan->order, handle->stream);
  dev_free(plan->level_ptr, handle->stream);
  dev_free(plan->tickets, handle->stream);
  if (plan->sticky) {
    // (the kernel of the last solve may still be running: the word must outlive it)
    (void) hipStreamSynchronize(handle->stream);
    (void) hipHostFree(plan->sticky);
  }
  delete plan;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_sptrsv_info(spblas_gfx950_trsv_t plan, int64_t info[4]) {
  if (!plan || !info)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  info[0] = plan->h_level_ptr.empty() ? 0 : (int64_t) plan->h_level_ptr.size() - 1;  // levels
  info[1] = plan->max_width;                                                         // widest level
  info[2] = plan->coop_ok && plan->groups.size() > 1 ? 1 : (int64_t) plan->groups.size();  // kernel launches per solve
  info[3] = plan->lanes;                                                             // lanes per row
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_sptrsv_status(spblas_gfx950_handle_t handle, spblas_gfx950_trsv_t plan, int* status) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!plan || !status)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (stream_capturing(handle->stream))
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  *status = 0;
  if (!plan->tickets)  // no solve yet
    return SPBLAS_GFX950_STATUS_SUCCESS;
  int st = 0;
  SPB_HIP(hipMemcpyAsync(&st, plan->tickets + plan->groups.size() + 1, sizeof(int), hipMemcpyDeviceToHost, handle->stream));
  SPB_HIP(hipStreamSynchronize(handle->stream));
  *status = st;
  if (plan->sticky)  // reported here: the next solve need not report it again
    __atomic_store_n(plan->sticky, 0, __ATOMIC_RELAXED);
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_sptrsv_solve(spblas_gfx950_handle_t handle, spblas_gfx950_trsv_t plan, int64_t m, int64_t nnz,
                               const void* alpha, const int32_t* rowptr, const int32_t* colind, const void* values,
                               const void* b, void* x, int value_type) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!plan || !alpha || !rowptr || (nnz > 0 && (!colind || !values)) || (m > 0 && (!b || !x)))
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (m != plan->m || nnz != plan->nnz)
    return SPBLAS_GFX950_STATUS_PLAN_MISMATCH;
  if (value_type != SPBLAS_GFX950_F32 && value_type != SPBLAS_GFX950_F64)
    return SPBLAS_GFX950_STATUS_INVALID_VALUE;
  if (m == 0)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  if (value_type == SPBLAS_GFX950_F32)
    return trsv_solve_lanes<float>(handle, plan, rowptr, colind, static_cast<const float*>(values),
                                   *static_cast<const float*>(alpha), static_cast<const float*>(b),
                                   static_cast<float*>(x));
  return trsv_solve_lanes<double>(handle, plan, rowptr, colind, static_cast<const double*>(values),
                                  *static_cast<const double*>(alpha), static_cast<const double*>(b),
                                  static_cast<double*>(x));
}

} // extern "C"

// Loads this file's code object (the runtime loads a code object at the first use of one of its kernels: milliseconds
// that would otherwise fall on the caller's first inspect / compute call -- handle.hip: spblas_gfx950_create).
namespace spb {
void preload_sptrsv() {
  hipFuncAttributes attr;
  (void) hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&trsv_degree_kernel));
  (void) hipGetLastError();
}
} // namespace spb
