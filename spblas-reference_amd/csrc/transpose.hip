// Device CSR -> CSR transpose:  B = A^T  (SURVEY.md section 8f ranks 1-2).
//
// Result identical to the reference's counting sort
// (/root/reference/include/spblas/algorithms/transpose_impl.hpp:14-53): entries of every output row
// are in source order (ascending original row, ties in storage order), i.e. a STABLE sort of the
// entries by column -- done here by hand-written LSD radix passes of 8 bits (spt_count_kernel, scan.hpp,
// spt_scatter_kernel) followed by one streaming pass that turns the sorted columns into t_rowptr
// (spt_rowptr_fill_kernel).  No library primitive: round 1 and most of round 2 used rocprim::radix_sort_pairs on
// (column, (row, value)) pairs between a payload pass, a binary-search pass and a split pass (3.45 ms at 1e8
// entries); the passes below take 2.6 ms.
// Used by transpose() itself and by multiply_inspect on csc_view / transposed(csr) operands, which then run the
// regular (row-block or LDS-sliced) SpMV kernels on the materialised transpose instead of the atomic
// scatter kernel.
#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "scan.hpp"

namespace spb {

// ---------------------------------------------------------------------------------------------------------------
// Hand-written stable LSD radix sort of the entries by column, 8 bits per pass, specialised for the transpose:
//   * the first pass builds the (row, value) payload on the fly (row of entry i = binary search in the tile's slice
//     of rowptr, held in LDS) -- no payload pass;
//   * the last pass writes t_colind / t_values directly -- no split pass;
//   * keys, rows and values travel as three arrays (three coalesced streams per piece);
//   * a tile is 8 waves x SPT_ROUNDS x 64 = 4096 entries per workgroup of 512 lanes (46 / 62 KiB of LDS: three / two
//     workgroups per CU), so a tile's piece of one of the 256 buckets averages 16 entries; tiles of 8192 (one
//     workgroup per CU) and of 3072 / 2048 entries were measured and are slower or equal;
//   * every XCD works on a contiguous range of tiles, so the pieces of neighbouring tiles, adjacent in memory, are
//     merged into whole lines by one L2.
// Per pass: spt_count_kernel (digit histogram of every tile) -> exclusive scan of counts[digit][tile] (scan.hpp; the
// flattened digit-major order IS the stable output order) -> spt_scatter_kernel.
// Stability inside a tile: wave w owns the contiguous entries [(w*R)*64, (w*R+R)*64) of the tile and walks them 64 at a
// time; the rank of a lane among the lanes of its round with the same digit comes from eight ballots, the running
// count of the digit over the wave's earlier rounds from an LDS counter that only this wave touches.
constexpr int SPT_WAVES = 8;

__global__ __launch_bounds__(512) void spt_count_kernel(int64_t nnz, int shift, int tile_size,
                                                        const int32_t* __restrict__ keys, int64_t ntiles,
                                                        int32_t* __restrict__ counts, int xcd_map) {
  // (one histogram per wave instead: no faster; 16-byte key loads: 172 -> 104 us per pass at 1e8 entries; four tiles per
  // workgroup with all their keys requested up front: 109 against 100 us)
  __shared__ int hist[256];
  if (threadIdx.x < 256)
    hist[threadIdx.x] = 0;
  __syncthreads();
  // the same tile -> XCD mapping as the scatter kernel: the counters of neighbouring tiles are neighbours in counts[digit][tile],
  // and dealt round-robin they reached their 32-byte sector from eight different L2s -- eight partial writes per sector
  int64_t tile = blockIdx.x;
  if (xcd_map) {
    const int64_t per = ntiles / 8, body = per * 8;
    if (tile < body)
      tile = (tile & 7) * per + (tile >> 3);
  }
  const int64_t base = tile * tile_size;
  int* mine = hist;
  const int tile_n = (int) (nnz - base < tile_size ? nnz - base : tile_size);
  if (tile_n == tile_size && (reinterpret_cast<uintptr_t>(keys + base) & 15) == 0) {
    const int4* k4 = reinterpret_cast<const int4*>(keys + base);
    for (int q = threadIdx.x; q < tile_size / 4; q += 512) {
      const int4 k = k4[q];
      atomicAdd(&mine[(k.x >> shift) & 255], 1);
      atomicAdd(&mine[(k.y >> shift) & 255], 1);
      atomicAdd(&mine[(k.z >> shift) & 255], 1);
      atomicAdd(&mine[(k.w >> shift) & 255], 1);
    }
  } else {
    for (int li = threadIdx.x; li < tile_n; li += 512)
      atomicAdd(&mine[(keys[base + li] >> shift) & 255], 1);
  }
  __syncthreads();
  if (threadIdx.x < 256)
    counts[(int64_t) threadIdx.x * ntiles + tile] = hist[threadIdx.x];
}

// tile_row[t] = row that owns entry t * tile_size (the last r with rowptr[r] <= it), tile_row[ntiles] = the row of the
// last entry: the rows of tile t lie in [tile_row[t], tile_row[t + 1]]
__global__ __launch_bounds__(256) void spt_tile_rows_kernel(int64_t ntiles, int tile_size, int64_t nnz, int64_t m,
                                                            const int32_t* __restrict__ rowptr,
                                                            int32_t* __restrict__ tile_row) {
  const int64_t t = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (t > ntiles)
    return;
  const int64_t target = t * tile_size < nnz ? t * tile_size : nnz - 1;
  int64_t lo = 0, hi = m;  // first r in [0, m] with rowptr[r] > target
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (rowptr[mid] <= target)
      lo = mid + 1;
    else
      hi = mid;
  }
  tile_row[t] = (int32_t) (lo - 1);
}

template <typename T, int SPT_ROUNDS, bool FIRST>
__global__ __launch_bounds__(512, sizeof(T) == 4 ? 6 : 4) void spt_scatter_kernel(int64_t nnz, int shift, const int32_t* __restrict__ in_keys,
                                                          const int32_t* __restrict__ in_rows,
                                                          const T* __restrict__ in_vals, int64_t m,
                                                          const int32_t* __restrict__ rowptr, int64_t ntiles,
                                                          const int32_t* __restrict__ offsets,
                                                          int32_t* __restrict__ out_keys,
                                                          int32_t* __restrict__ out_rows, T* __restrict__ out_vals,
                                                          int xcd_map, const int32_t* __restrict__ tile_row) {
  constexpr int SPT_TILE = SPT_WAVES * SPT_ROUNDS * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char spt_smem[];
  // (explicit LDS pointers: through generic ones the VOLATILE counter accesses below were compiled as flat loads and stores
  // with system-scope cache bits, which also count as vector-memory operations -- every counter read of the ranking loop
  // then sat behind the global loads in flight: 657 -> 586 us per pass at 1e8 entries)
  typedef __attribute__((address_space(3))) T lds_T;
  typedef __attribute__((address_space(3))) int lds_int;
  lds_T* stage_val = (lds_T*) spt_smem;                              // [TILE]
  lds_int* stage_row = (lds_int*) (stage_val + SPT_TILE);            // [TILE] rows, then (second write-out) the keys
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  lds_u8* stage_dig = (lds_u8*) (stage_row + SPT_TILE);              // [TILE] digit of the staged entry
  volatile lds_int* cnt = (lds_int*) (stage_dig + SPT_TILE);         // [WAVES][256] running counts -> wave offsets
  lds_int* lstart = (lds_int*) cnt + SPT_WAVES * 256;                // [256] first local position of a bucket
  lds_int* delta = lstart + 256;                                     // [256] global - local position of a bucket
  lds_int* misc = delta + 256;                                       // [8]

  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  // workgroups are dealt round-robin to the 8 XCDs: give every XCD a contiguous range of tiles, so that the pieces
  // of neighbouring tiles (adjacent in memory, bucket by bucket) meet in ONE L2 and leave it as whole lines
  // (4.0 -> 3.3 ms at 1e8 entries; SPBLAS_GFX950_TRANSPOSE_XCD=0 switches the mapping off)
  int64_t tile = blockIdx.x;
  if (xcd_map) {
    const int64_t per = ntiles / 8, body = per * 8;
    if (tile < body)
      tile = (tile & 7) * per + (tile >> 3);
  }
  const int64_t base = tile * SPT_TILE;
  const int tile_n = (int) (nnz - base < SPT_TILE ? nnz - base : SPT_TILE);
  for (int i = tid; i < SPT_WAVES * 256; i += 512)
    cnt[i] = 0;

  __syncthreads();

  int key[SPT_ROUNDS], row[SPT_ROUNDS], rank[SPT_ROUNDS];
  T val[SPT_ROUNDS];
#pragma unroll
  for (int r = 0; r < SPT_ROUNDS; ++r) {
    const int li = (w * SPT_ROUNDS + r) * 64 + lane;
    const bool valid = li < tile_n;
    key[r] = valid ? in_keys[base + li] : 0;
    val[r] = valid ? in_vals[base + li] : T(0);
    if (!FIRST)
      row[r] = valid ? in_rows[base + li] : 0;
  }
  if (FIRST) {
    // row of every entry of the tile (the (row, value) payload is built here, no separate pass): the rows
    // r_lo + 1 .. r_hi that start inside the tile leave their number at their first entry (LDS max: of several rows
    // starting at one position -- empty ones -- the last owns the entry), and an inclusive running maximum over the
    // tile's positions turns the marks into "row of entry".  Wave scans side by side for the ROUNDS of a lane, a carry
    // along the wave's rounds, one maximum per earlier wave.  (A binary search per entry in an LDS copy of the row
    // offsets took 0.34 ms more per pass at 1e8 entries, one search at a time 0.9 ms.)
    lds_int* mark = stage_row;  // free until the staging phase
    const int r_lo = tile_row[tile], r_hi = tile_row[tile + 1];
    for (int i = tid; i < SPT_TILE; i += 512)
      mark[i] = 0;
    __syncthreads();
    for (int j = 1 + tid; j <= r_hi - r_lo; j += 512) {
      const int64_t pos = (int64_t) rowptr[r_lo + j] - base;
      if (pos >= 0 && pos < tile_n)
        __hip_atomic_fetch_max(&mark[pos], j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    int x[SPT_ROUNDS];
#pragma unroll
    for (int r = 0; r < SPT_ROUNDS; ++r)
      x[r] = mark[(w * SPT_ROUNDS + r) * 64 + lane];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
      for (int r = 0; r < SPT_ROUNDS; ++r) {
        const int t = __shfl_up(x[r], o, 64);
        if (lane >= o)
          x[r] = x[r] > t ? x[r] : t;
      }
    }
    int carry = 0;
#pragma unroll
    for (int r = 0; r < SPT_ROUNDS; ++r) {
      x[r] = x[r] > carry ? x[r] : carry;
      carry = __shfl(x[r], 63);
    }
    if (lane == 0)
      misc[2 + w] = carry;
    __syncthreads();
    int pre = 0;
    for (int q = 0; q < w; ++q)
      pre = misc[2 + q] > pre ? misc[2 + q] : pre;
#pragma unroll
    for (int r = 0; r < SPT_ROUNDS; ++r)
      row[r] = r_lo + (x[r] > pre ? x[r] : pre);
  }
  // ranks inside the wave's chunk
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < SPT_ROUNDS; ++r) {
    const int li = (w * SPT_ROUNDS + r) * 64 + lane;
    const bool valid = li < tile_n;
    const int d = (key[r] >> shift) & 255;
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1;
      const unsigned long long bal = __ballot(bit);
      same &= bit ? bal : ~bal;
    }
    const int before = (int) __popcll(same & lt_mask);
    const int prev = valid ? cnt[w * 256 + d] : 0;
    __builtin_amdgcn_wave_barrier();
    if (valid && before == 0)
      cnt[w * 256 + d] = prev + (int) __popcll(same);
    __builtin_amdgcn_wave_barrier();
    rank[r] = prev + before;
  }
  __syncthreads();
  int run = 0, incl = 0;
  if (tid < 256) {
    for (int q = 0; q < SPT_WAVES; ++q) {
      const int c = cnt[q * 256 + tid];
      cnt[q * 256 + tid] = run;
      run += c;
    }
    incl = run;
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o)
        incl += v;
    }
    if (lane == 63)
      misc[2 + w] = incl;
  }
  __syncthreads();
  if (tid < 256) {
    int ls = incl - run;
    for (int q = 0; q < w; ++q)
      ls += misc[2 + q];
    lstart[tid] = ls;
    delta[tid] = offsets[(int64_t) tid * ntiles + tile] - ls;
  }
  __syncthreads();
  // Staged in TWO rounds through the same space -- rows and values first, then the keys where the rows were, the digit of
  // every staged entry kept as a byte for the address of both write-outs: 46 KiB of LDS per workgroup instead of 59.5 (fp32),
  // THREE workgroups per CU instead of two.  The kernel lives on overlap between workgroups (with one per CU a pass took
  // 895 us instead of 570); two more barriers per tile pay for a third.
#pragma unroll
  for (int r = 0; r < SPT_ROUNDS; ++r) {
    const int d = (key[r] >> shift) & 255;
    rank[r] += lstart[d] + cnt[w * 256 + d];  // now the entry's position in the staged tile
    if ((w * SPT_ROUNDS + r) * 64 + lane < tile_n) {
      stage_row[rank[r]] = row[r];
      stage_val[rank[r]] = val[r];
      stage_dig[rank[r]] = (unsigned char) d;
    }
  }
  __syncthreads();
  for (int j = tid; j < tile_n; j += 512) {
    const int64_t g = (int64_t) j + delta[stage_dig[j]];
    out_rows[g] = stage_row[j];
    out_vals[g] = stage_val[j];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < SPT_ROUNDS; ++r)
    if ((w * SPT_ROUNDS + r) * 64 + lane < tile_n)
      stage_row[rank[r]] = key[r];
  __syncthreads();
  for (int j = tid; j < tile_n; j += 512)
    out_keys[(int64_t) j + delta[stage_dig[j]]] = stage_row[j];
}

// t_rowptr from the sorted columns, one streaming pass: entry k-1 is the last entry of its column c when the next
// entry has another column c'; then t_rowptr[c + 1 .. c'] = k.  Work item 0 covers the columns up to the first entry's.
// A lane fills short gaps itself, the wave fills gaps up to 4096 columns together, longer ones (at most n / 4096 of
// them) go to a list that spt_rowptr_long_kernel fills with the whole grid.
__global__ __launch_bounds__(256) void spt_rowptr_fill_kernel(int64_t n, int64_t nnz,
                                                              const int32_t* __restrict__ sorted_cols,
                                                              int32_t* __restrict__ t_rowptr, int4* __restrict__ longs,
                                                              unsigned* __restrict__ n_longs) {
  // four work items per lane (one 16-byte load of sorted columns): items 4q .. 4q + 3 of 0 .. nnz
  const int64_t q = (int64_t) blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int64_t t0 = 4 * q;
  int k[4] = {0, 0, 0, 0};
  int prev = -1;
  if (t0 <= nnz) {
    if (t0 > 0)
      prev = sorted_cols[t0 - 1];
    if (t0 + 4 <= nnz && (reinterpret_cast<uintptr_t>(sorted_cols + t0) & 15) == 0) {
      const int4 v4 = *reinterpret_cast<const int4*>(sorted_cols + t0);
      k[0] = v4.x, k[1] = v4.y, k[2] = v4.z, k[3] = v4.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        k[e] = t0 + e < nnz ? sorted_cols[t0 + e] : (int) n;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t t = t0 + e;
    int lo = 0, hi = -1;
    const int v = (int) t;
    if (t <= nnz) {
      lo = prev + 1;
      hi = t < nnz ? k[e] : (int) n;
      prev = k[e];
    }
    const int len = hi - lo + 1;
    if (len > 0 && len <= 8)
      for (int j = lo; j <= hi; ++j)
        t_rowptr[j] = v;
    if (len > 4096) {
      const unsigned slot = atomicAdd(n_longs, 1u);
      longs[slot] = make_int4(lo, hi, v, 0);
    }
    unsigned long long mid = __ballot(len > 8 && len <= 4096);
    while (mid) {
      const int src = __ffsll((long long) mid) - 1;
      const int slo = __shfl(lo, src), shi = __shfl(hi, src), sv = __shfl(v, src);
      for (int j = slo + lane; j <= shi; j += 64)
        t_rowptr[j] = sv;
      mid &= mid - 1;
    }
  }
}

__global__ __launch_bounds__(256) void spt_rowptr_long_kernel(const int4* __restrict__ longs,
                                                              const unsigned* __restrict__ n_longs,
                                                              int32_t* __restrict__ t_rowptr) {
  const unsigned cnt = *n_longs;
  for (unsigned it = 0; it < cnt; ++it) {
    const int4 e = longs[it];
    const int64_t len = (int64_t) e.y - e.x + 1;
    const int64_t per = (len + gridDim.x - 1) / gridDim.x;
    const int64_t j0 = e.x + (int64_t) blockIdx.x * per;
    const int64_t j1 = j0 + per - 1 < e.y ? j0 + per - 1 : e.y;
    for (int64_t j = j0 + threadIdx.x; j <= j1; j += 256)
      t_rowptr[j] = e.z;
  }
}

template <typename T>
static int transpose_radix(spblas_gfx950_handle_t handle, int64_t m, int64_t n, int64_t nnz, const int32_t* rowptr,
                           const int32_t* colind, const T* values, int32_t* t_rowptr, int32_t* t_colind,
                           T* t_values) {
  hipStream_t s = handle->stream;
  int bits = 1;
  while (bits < 32 && ((int64_t) 1 << bits) < n)
    ++bits;
  const int passes = (bits + 7) / 8;
  static const int xcd_map = [] {
    const char* e = std::getenv("SPBLAS_GFX950_TRANSPOSE_XCD");
    return e ? std::atoi(e) : 1;
  }();
  constexpr int SPT_ROUNDS = 8;
  constexpr int SPT_TILE = SPT_WAVES * SPT_ROUNDS * 64;
  const int64_t ntiles = cdiv(nnz, SPT_TILE);
  const int64_t nscan = 256 * ntiles, nb = cdiv(nscan, 2048);
  auto al = [](size_t b) { return (b + 255) & ~(size_t) 255; };
  const size_t key_b = al((size_t) nnz * 4), val_b = al((size_t) nnz * sizeof(T));
  const size_t cnt_b = al((size_t) (nscan + 1) * 4), part_b = al((size_t) (nb + 2) * sizeof(long long));
  const size_t long_b = al((size_t) (n / 4096 + 4) * sizeof(int4)) + 256;
  const size_t trow_b = al((size_t) (ntiles + 1) * 4);
  // two intermediate (key, row, value) sets -- a one- or two-pass sort needs none or one -- and the sorted keys
  const int nsets = passes >= 3 ? 2 : passes - 1;
  void* basep = nullptr;
  int rc = handle_scratch(handle, (size_t) nsets * (2 * key_b + val_b) + key_b + cnt_b + part_b + long_b + trow_b + 256,
                          &basep);
  if (rc)
    return rc;
  char* bp = static_cast<char*>(basep);
  int32_t* set_keys[2];
  int32_t* set_rows[2];
  T* set_vals[2];
  for (int i = 0; i < 2; ++i) {
    char* q = bp + (size_t) (i < nsets ? i : 0) * (2 * key_b + val_b);
    set_keys[i] = reinterpret_cast<int32_t*>(q);
    set_rows[i] = reinterpret_cast<int32_t*>(q + key_b);
    set_vals[i] = reinterpret_cast<T*>(q + 2 * key_b);
  }
  char* q = bp + (size_t) nsets * (2 * key_b + val_b);
  int32_t* sorted_cols = reinterpret_cast<int32_t*>(q);
  int32_t* counts = reinterpret_cast<int32_t*>(q + key_b);
  long long* partials = reinterpret_cast<long long*>(q + key_b + cnt_b);
  int4* longs = reinterpret_cast<int4*>(q + key_b + cnt_b + part_b);
  unsigned* n_longs = reinterpret_cast<unsigned*>(q + key_b + cnt_b + part_b + long_b - 256);
  int32_t* tile_row = reinterpret_cast<int32_t*>(q + key_b + cnt_b + part_b + long_b);

  const size_t smem = (size_t) SPT_TILE * (5 + sizeof(T)) + (size_t) (SPT_WAVES * 256 + 512 + 16) * 4;
  auto k_first = spt_scatter_kernel<T, SPT_ROUNDS, true>;
  auto k_next = spt_scatter_kernel<T, SPT_ROUNDS, false>;
  SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_first), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int) smem));
  SPB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_next), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int) smem));
  hipLaunchKernelGGL(spt_tile_rows_kernel, dim3((unsigned) cdiv(ntiles + 1, 256)), dim3(256), 0, s, ntiles, SPT_TILE, nnz,
                     m, rowptr, tile_row);
  const int32_t* in_keys = colind;
  const int32_t* in_rows = nullptr;
  const T* in_vals = values;
  for (int p = 0; p < passes; ++p) {
    const bool last = p == passes - 1;
    int32_t* out_keys = last ? sorted_cols : set_keys[p & 1];
    int32_t* out_rows = last ? t_colind : set_rows[p & 1];
    T* out_vals = last ? t_values : set_vals[p & 1];
    hipLaunchKernelGGL(spt_count_kernel, dim3((unsigned) ntiles), dim3(512), 0, s, nnz, 8 * p, SPT_TILE, in_keys, ntiles,
                       counts, xcd_map);
    scan_counts_i32(s, nscan, counts, partials);
    if (p == 0)
      hipLaunchKernelGGL(k_first, dim3((unsigned) ntiles), dim3(512), smem, s, nnz, 8 * p, in_keys, in_rows, in_vals, m,
                         rowptr, ntiles, counts, out_keys, out_rows, out_vals, xcd_map, tile_row);
    else
      hipLaunchKernelGGL(k_next, dim3((unsigned) ntiles), dim3(512), smem, s, nnz, 8 * p, in_keys, in_rows, in_vals, m,
                         rowptr, ntiles, counts, out_keys, out_rows, out_vals, xcd_map, tile_row);
    in_keys = out_keys;
    in_rows = out_rows;
    in_vals = out_vals;
  }
  SPB_HIP(hipMemsetAsync(n_longs, 0, 4, s));
  hipLaunchKernelGGL(spt_rowptr_fill_kernel, dim3((unsigned) cdiv(nnz + 1, 1024)), dim3(256), 0, s, n, nnz, sorted_cols,
                     t_rowptr, longs, n_longs);
  hipLaunchKernelGGL(spt_rowptr_long_kernel, dim3(512), dim3(256), 0, s, longs, n_longs, t_rowptr);
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

// scale(alpha, t): 16-byte accesses over the aligned body, scalar head and tail
template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(int64_t n, T alpha, T* __restrict__ v) {
  constexpr int V = 16 / (int) sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(V)));
  const int64_t head = ((16 - (reinterpret_cast<uintptr_t>(v) & 15)) & 15) / sizeof(T);
  const int64_t h = head < n ? head : n;
  const int64_t nv = (n - h) / V;
  const int64_t tid = (int64_t) blockIdx.x * 256 + threadIdx.x, stride = (int64_t) gridDim.x * 256;
  vec_t* body = reinterpret_cast<vec_t*>(v + h);
  for (int64_t i = tid; i < nv; i += stride)
    body[i] = body[i] * alpha;
  if (tid < h)
    v[tid] *= alpha;
  const int64_t t0 = h + nv * V;
  if (t0 + tid < n && tid < V)
    v[t0 + tid] *= alpha;
}

} // namespace spb

using namespace spb;

extern "C" int spblas_gfx950_csr_transpose(spblas_gfx950_handle_t handle, int64_t m, int64_t n, int64_t nnz,
                                           const int32_t* rowptr, const int32_t* colind, const void* values,
                                           int32_t* t_rowptr, int32_t* t_colind, void* t_values, int value_type) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (stream_capturing(handle->stream))  // the radix passes allocate their scratch per call
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  if (m < 0 || n < 0 || nnz < 0 || m > INT32_MAX || n >= INT32_MAX || nnz > INT32_MAX)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  if (value_type != SPBLAS_GFX950_F32 && value_type != SPBLAS_GFX950_F64)
    return SPBLAS_GFX950_STATUS_INVALID_VALUE;
  if (!rowptr || !t_rowptr || (nnz > 0 && (!colind || !values || !t_colind || !t_values)))
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (nnz == 0) {
    SPB_HIP(hipMemsetAsync(t_rowptr, 0, (size_t) (n + 1) * 4, handle->stream));
    return SPBLAS_GFX950_STATUS_SUCCESS;
  }
  if (value_type == SPBLAS_GFX950_F32)
    return transpose_radix<float>(handle, m, n, nnz, rowptr, colind, static_cast<const float*>(values), t_rowptr,
                                  t_colind, static_cast<float*>(t_values));
  return transpose_radix<double>(handle, m, n, nnz, rowptr, colind, static_cast<const double*>(values), t_rowptr,
                                 t_colind, static_cast<double*>(t_values));
}

// dst[i] = (int32) src[i]; *bad is raised when an index does not fit [0, bound)
__global__ __launch_bounds__(256) void spt_narrow_kernel(int64_t n, const int64_t* __restrict__ src, int32_t* __restrict__ dst,
                                                         int64_t bound, int* __restrict__ bad) {
  bool off = false;
  for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t) gridDim.x * 256) {
    const int64_t v = src[i];
    off |= v < 0 || v >= bound;
    dst[i] = (int32_t) v;
  }
  if (__ballot(off) != 0ull && (threadIdx.x & 63) == 0)
    atomicOr(bad, 1);
}

// 64-bit column (row) indices of a view -- the rocSPARSE slot admits them (vendor/rocsparse/types.hpp:16-24) -- narrowed into
// a 32-bit array the kernels of this library take; inspect-class (it answers on the host whether every index fitted).
extern "C" int spblas_gfx950_narrow_indices(spblas_gfx950_handle_t handle, int64_t count, const int64_t* src, int32_t* dst,
                                            int64_t bound) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (stream_capturing(handle->stream))
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  if (count < 0 || bound < 0 || bound > (int64_t) INT32_MAX)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  if (count > 0 && (!src || !dst))
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (count == 0)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  hipStream_t s = handle->stream;
  int* bad = nullptr;
  int rc = dev_alloc((void**) &bad, sizeof(int), s);
  if (rc)
    return rc;
  int h_bad = 0;
  hipError_t e = hipMemsetAsync(bad, 0, sizeof(int), s);
  if (e == hipSuccess) {
    int64_t blocks = cdiv(count, 1024);
    const int64_t cap = (int64_t) (handle->num_cus > 0 ? handle->num_cus : 256) * 16;
    blocks = blocks > cap ? cap : blocks;
    hipLaunchKernelGGL(spt_narrow_kernel, dim3((unsigned) blocks), dim3(256), 0, s, count, src, dst, bound, bad);
    e = hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, s);
  }
  if (e == hipSuccess)
    e = hipStreamSynchronize(s);
  dev_free(bad, s);
  if (e != hipSuccess)
    return hip_fail(e);
  return h_bad ? SPBLAS_GFX950_STATUS_INVALID_VALUE : SPBLAS_GFX950_STATUS_SUCCESS;
}

extern "C" int spblas_gfx950_scale(spblas_gfx950_handle_t handle, int64_t n, const void* alpha, void* values,
                                   int value_type) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (n < 0)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  if (value_type != SPBLAS_GFX950_F32 && value_type != SPBLAS_GFX950_F64)
    return SPBLAS_GFX950_STATUS_INVALID_VALUE;
  if (!alpha || (n > 0 && !values))
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (n == 0)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  const int64_t per = value_type == SPBLAS_GFX950_F32 ? 4 : 2;
  int64_t blocks = cdiv(cdiv(n, per), 256);
  const int64_t cap = (int64_t) (handle->num_cus > 0 ? handle->num_cus : 256) * 16;
  if (blocks > cap)
    blocks = cap;
  if (blocks < 1)
    blocks = 1;
  if (value_type == SPBLAS_GFX950_F32)
    hipLaunchKernelGGL(scale_kernel<float>, dim3((unsigned) blocks), dim3(256), 0, handle->stream, n,
                       *static_cast<const float*>(alpha), static_cast<float*>(values));
  else
    hipLaunchKernelGGL(scale_kernel<double>, dim3((unsigned) blocks), dim3(256), 0, handle->stream, n,
                       *static_cast<const double*>(alpha), static_cast<double*>(values));
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

// Loads this file's code object (the runtime loads a code object at the first use of one of its kernels: milliseconds
// that would otherwise fall on the caller's first inspect / compute call -- handle.hip: spblas_gfx950_create).
namespace spb {
void preload_transpose() {
  hipFuncAttributes attr;
  (void) hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&spt_count_kernel));
  (void) hipGetLastError();
}
} // namespace spb
