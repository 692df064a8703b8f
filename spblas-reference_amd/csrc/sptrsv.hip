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
//   inspect   in-degree of every row + adjacency "row k -> rows that read x_k" (strict triangle
//             transposed, built with atomics: order is irrelevant), then Kahn's algorithm: the
//             frontier of rows whose in-degree dropped to zero becomes the next level.
//   solve     one launch per WIDE level (rows spread over the chip, G lanes per row), and ONE
//             single-workgroup launch per run of consecutive NARROW levels, which walks them with a
//             workgroup barrier in between -- a chain-like matrix then costs a barrier per level
//             instead of a kernel launch per level.  The same split is used inside inspect.
// Row sums are computed G lanes wide and tree-reduced, so they re-associate with respect to the
// reference's sequential loop: parity is norm-wise, not bit-wise.
#include "common.hpp"
#include "scan.hpp"

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

// x_r for one row, computed by a group of G lanes (all lanes of the group return the same values)
template <typename T, int G>
__device__ __forceinline__ void trsv_row(int r, int lane, const int32_t* __restrict__ rowptr,
                                         const int32_t* __restrict__ colind, const T* __restrict__ values, T alpha,
                                         const T* __restrict__ b, T* x, int upper, int unit) {
  T dot = T(0);
  int dpos = -1;
  const int p1 = rowptr[r + 1];
  for (int p = rowptr[r] + lane; p < p1; p += G) {
    const int c = colind[p];
    if (trsv_strict(c, r, upper))
      dot += values[p] * __hip_atomic_load(&x[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (c == r)
      dpos = p;  // the last stored diagonal entry wins (triangular_solve_impl.hpp:64-66,81-83)
  }
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) {
    dot += __shfl_xor(dot, o, SPB_WAVE);
    const int other = __shfl_xor(dpos, o, SPB_WAVE);
    dpos = other > dpos ? other : dpos;
  }
  if (lane == 0) {
    T v = b[r] - alpha * dot;
    if (!unit)
      v = v / (alpha * (dpos >= 0 ? values[dpos] : T(0)));
    __hip_atomic_store(&x[r], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// one wide level: rows order[f0..f1), G lanes per row
template <typename T, int G>
__global__ __launch_bounds__(256) void trsv_level_kernel(int f0, int f1, const int32_t* __restrict__ order,
                                                         const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ colind,
                                                         const T* __restrict__ values, T alpha,
                                                         const T* __restrict__ b, T* x, int upper, int unit) {
  const int idx = f0 + blockIdx.x * (256 / G) + threadIdx.x / G;
  if (idx >= f1)
    return;
  trsv_row<T, G>(order[idx], threadIdx.x % G, rowptr, colind, values, alpha, b, x, upper, unit);
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
                                                                       int unit) {
  for (int l = l0; l < l1; ++l) {
    const int f0 = level_ptr[l], f1 = level_ptr[l + 1];
    for (int idx = f0 + threadIdx.x / G; idx < f1; idx += TRSV_BLOCK_THREADS / G)
      trsv_row<T, G>(order[idx], threadIdx.x % G, rowptr, colind, values, alpha, b, x, upper, unit);
    __threadfence();  // x of this level must be visible to the whole workgroup before the next one
    __syncthreads();
  }
}

template <typename T, int G>
static int trsv_solve_typed(hipStream_t s, const spblas_gfx950_trsv_s* pl, const int32_t* rowptr,
                            const int32_t* colind, const T* values, T alpha, const T* b, T* x) {
  const int upper = pl->uplo == SPBLAS_GFX950_UPPER, unit = pl->diag == SPBLAS_GFX950_DIAG_UNIT;
  for (const auto& g : pl->groups) {
    if (g.wide) {
      const int f0 = pl->h_level_ptr[g.l0], f1 = pl->h_level_ptr[g.l0 + 1];
      hipLaunchKernelGGL((trsv_level_kernel<T, G>), dim3((unsigned) cdiv(f1 - f0, 256 / G)), dim3(256), 0, s, f0, f1,
                         pl->order, rowptr, colind, values, alpha, b, x, upper, unit);
    } else {
      hipLaunchKernelGGL((trsv_chain_kernel<T, G>), dim3(1), dim3(TRSV_BLOCK_THREADS), 0, s, g.l0, g.l1,
                         pl->level_ptr, pl->order, rowptr, colind, values, alpha, b, x, upper, unit);
    }
  }
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

template <typename T>
static int trsv_solve_lanes(hipStream_t s, const spblas_gfx950_trsv_s* pl, const int32_t* rowptr,
                            const int32_t* colind, const T* values, T alpha, const T* b, T* x) {
  switch (pl->lanes) {
    case 4: return trsv_solve_typed<T, 4>(s, pl, rowptr, colind, values, alpha, b, x);
    case 16: return trsv_solve_typed<T, 16>(s, pl, rowptr, colind, values, alpha, b, x);
    case 64: return trsv_solve_typed<T, 64>(s, pl, rowptr, colind, values, alpha, b, x);
    default: return trsv_solve_typed<T, 8>(s, pl, rowptr, colind, values, alpha, b, x);
  }
}

} // namespace spb

using namespace spb;

extern "C" {

int spblas_gfx950_sptrsv_create(spblas_gfx950_handle_t handle, spblas_gfx950_trsv_t* plan_out, int64_t m, int64_t nnz,
                                const int32_t* rowptr, const int32_t* colind, int uplo, int diag) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
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
  hipError_t e = hipMemsetAsync(adj_ptr, 0, (size_t) (m + 1) * 4, s);
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
  std::vector<int32_t>& lp = pl->h_level_ptr;
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
  const int32_t n_levels = st[3];
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
  pl->level_ptr = level_ptr;
  level_ptr = nullptr;
  // launch groups of the solve: a level of >= `narrow` rows gets its own chip-wide launch, runs of
  // narrower levels share one single-workgroup launch
  const int narrow = env_int("SPBLAS_GFX950_TRSV_NARROW", 128);
  for (int32_t l = 0; l < n_levels;) {
    const int64_t w = lp[l + 1] - lp[l];
    pl->max_width = w > pl->max_width ? w : pl->max_width;
    if (w >= narrow) {
      pl->groups.push_back({l, l + 1, 1});
      ++l;
    } else {
      int32_t e1 = l + 1;
      while (e1 < n_levels && lp[e1 + 1] - lp[e1] < narrow) {
        const int64_t w2 = lp[e1 + 1] - lp[e1];
        pl->max_width = w2 > pl->max_width ? w2 : pl->max_width;
        ++e1;
      }
      pl->groups.push_back({l, e1, 0});
      l = e1;
    }
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
  delete plan;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_sptrsv_info(spblas_gfx950_trsv_t plan, int64_t info[4]) {
  if (!plan || !info)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  info[0] = plan->h_level_ptr.empty() ? 0 : (int64_t) plan->h_level_ptr.size() - 1;  // levels
  info[1] = plan->max_width;                                                         // widest level
  info[2] = (int64_t) plan->groups.size();                                           // kernel launches per solve
  info[3] = plan->lanes;                                                             // lanes per row
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
    return trsv_solve_lanes<float>(handle->stream, plan, rowptr, colind, static_cast<const float*>(values),
                                   *static_cast<const float*>(alpha), static_cast<const float*>(b),
                                   static_cast<float*>(x));
  return trsv_solve_lanes<double>(handle->stream, plan, rowptr, colind, static_cast<const double*>(values),
                                  *static_cast<const double*>(alpha), static_cast<const double*>(b),
                                  static_cast<double*>(x));
}

} // extern "C"
