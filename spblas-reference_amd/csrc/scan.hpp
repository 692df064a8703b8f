// Exclusive prefix sum of int32 counts on device (three small kernels), shared by the
// SpGEMM symbolic phase (row offsets of C) and the sliced-SpMV inspect phase (segment offsets).
#pragma once

#include "common.hpp"

namespace spb {

// ---- exclusive scan of int32 counts (in place), total in int64 ----------------
// Round 5: 16-byte accesses where the array allows them and wave scans (two workgroup barriers per kernel instead of sixteen);
// the partials are scanned 4 096 per round.  At the transpose's 6.25 M tile counters: 11 + 15 + 28 us -> see DESIGN.md.
// data[0..n) counts -> offsets; block partial sums in `partials` (int64).  A block is 2 048 entries, a thread owns 8
// consecutive ones.
static __device__ __forceinline__ void scan_load8(const int32_t* __restrict__ data, int64_t i0, int64_t n, bool vec, int (&v)[8]) {
  if (vec && i0 + 8 <= n) {
    const int4 a = *reinterpret_cast<const int4*>(data + i0), b = *reinterpret_cast<const int4*>(data + i0 + 4);
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[j] = i0 + j < n ? data[i0 + j] : 0;
  }
}

static __global__ __launch_bounds__(256) void scan_block_sums_kernel(int64_t n, const int32_t* __restrict__ data,
                                                              long long* __restrict__ partials) {
  __shared__ long long red[4];
  const bool vec = (reinterpret_cast<uintptr_t>(data) & 15) == 0;
  int v[8];
  scan_load8(data, (int64_t) blockIdx.x * 2048 + threadIdx.x * 8, n, vec, v);
  long long s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    s += v[j];
  s = group_sum_c<64>(s);
  if ((threadIdx.x & 63) == 0)
    red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0)
    partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// inclusive scan over the 64 lanes of a wavefront
template <typename V>
static __device__ __forceinline__ V scan_wave_incl(V v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const V t = __shfl_up(v, o, 64);
    if (lane >= o)
      v += t;
  }
  return v;
}

// single workgroup: exclusive scan of partials[0..nb) in place, partials[nb] = total; 16 partials per thread and round
static __global__ __launch_bounds__(256) void scan_partials_kernel(int64_t nb, long long* __restrict__ partials) {
  __shared__ long long wsum[4];
  long long carry = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += 4096) {
    const int64_t i0 = b0 + (int64_t) threadIdx.x * 16;
    long long v[16], s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      v[j] = i0 + j < nb ? partials[i0 + j] : 0;
      s += v[j];
    }
    const long long incl = scan_wave_incl(s);
    if ((threadIdx.x & 63) == 63)
      wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    long long off = carry + incl - s;
    for (int q = 0; q < (int) (threadIdx.x >> 6); ++q)
      off += wsum[q];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (i0 + j < nb)
        partials[i0 + j] = off;
      off += v[j];
    }
    carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (threadIdx.x == 0)
    partials[nb] = carry;
}

// data[0..n) counts -> exclusive offsets, data[n] = total (n+1 entries written)
// (`copy`, when given, receives the same n+1 offsets: the caller's row pointer array next to the plan's own)
static __global__ __launch_bounds__(256) void scan_apply_kernel(int64_t n, int32_t* __restrict__ data,
                                                         const long long* __restrict__ partials,
                                                         int32_t* __restrict__ copy = nullptr) {
  __shared__ int wsum[4];
  const bool vec = (reinterpret_cast<uintptr_t>(data) & 15) == 0;
  const int64_t i0 = (int64_t) blockIdx.x * 2048 + threadIdx.x * 8;
  int v[8];
  scan_load8(data, i0, n, vec, v);
  int s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    s += v[j];
  const int incl = scan_wave_incl(s);
  if ((threadIdx.x & 63) == 63)
    wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  long long off = partials[blockIdx.x] + incl - s;
  for (int q = 0; q < (int) (threadIdx.x >> 6); ++q)
    off += wsum[q];
  int o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    o[j] = (int32_t) off;
    off += v[j];
  }
  if (vec && i0 + 8 <= n) {
    *reinterpret_cast<int4*>(data + i0) = make_int4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<int4*>(data + i0 + 4) = make_int4(o[4], o[5], o[6], o[7]);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (i0 + j < n)
        data[i0 + j] = o[j];
  }
  if (copy) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (i0 + j < n)
        copy[i0 + j] = o[j];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    data[n] = (int32_t) partials[gridDim.x];
    if (copy)
      copy[n] = (int32_t) partials[gridDim.x];
  }
}


// data[0..n) int32 counts -> exclusive offsets in place, data[n] = total (so data holds n+1
// entries).  `partials` must hold cdiv(n,2048)+1 long longs.  *total_dev (device pointer into
// partials) receives the int64 total; the caller decides when to synchronise.
static inline long long* scan_counts_i32(hipStream_t s, int64_t n, int32_t* data, long long* partials) {
  const int64_t nb = cdiv(n, 2048);
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned) nb), dim3(256), 0, s, n, data, partials);
  hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(256), 0, s, nb, partials);
  hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned) nb), dim3(256), 0, s, n, data, partials, (int32_t*) nullptr);
  return partials + nb;
}

} // namespace spb
