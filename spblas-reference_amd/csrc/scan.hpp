// Exclusive prefix sum of int32 counts on device (three small kernels), shared by the
// SpGEMM symbolic phase (row offsets of C) and the sliced-SpMV inspect phase (segment offsets).
#pragma once

#include "common.hpp"

namespace spb {

// ---- exclusive scan of int32 counts (in place), total in int64 ----------------
// data[0..n) counts -> offsets; block partial sums in `partials` (int64).
static __global__ __launch_bounds__(256) void scan_block_sums_kernel(int64_t n, const int32_t* __restrict__ data,
                                                              long long* __restrict__ partials) {
  __shared__ long long red[4];
  const int64_t base = (int64_t) blockIdx.x * 2048;
  long long s = 0;
  for (int i = threadIdx.x; i < 2048; i += 256)
    if (base + i < n)
      s += data[base + i];
  s = group_sum_c<64>(s);
  if ((threadIdx.x & 63) == 0)
    red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0)
    partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// single workgroup: exclusive scan of partials[0..nb) in place, partials[nb] = total
static __global__ __launch_bounds__(256) void scan_partials_kernel(int64_t nb, long long* __restrict__ partials) {
  __shared__ long long sm[256];
  __shared__ long long carry;
  if (threadIdx.x == 0)
    carry = 0;
  __syncthreads();
  for (int64_t b0 = 0; b0 < nb; b0 += 256) {
    const int64_t i = b0 + threadIdx.x;
    const long long v = i < nb ? partials[i] : 0;
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const long long t = (int) threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
      __syncthreads();
      sm[threadIdx.x] += t;
      __syncthreads();
    }
    const long long c = carry;
    if (i < nb)
      partials[i] = c + sm[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 255)
      carry = c + sm[255];
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
  __shared__ int sm[256];
  const int64_t base = (int64_t) blockIdx.x * 2048;
  // each thread owns 8 consecutive entries
  int v[8];
  int s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int64_t i = base + threadIdx.x * 8 + j;
    v[j] = i < n ? data[i] : 0;
    s += v[j];
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int t = (int) threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
    __syncthreads();
    sm[threadIdx.x] += t;
    __syncthreads();
  }
  long long off = partials[blockIdx.x] + sm[threadIdx.x] - s;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int64_t i = base + threadIdx.x * 8 + j;
    if (i < n) {
      data[i] = (int32_t) off;
      if (copy)
        copy[i] = (int32_t) off;
    }
    off += v[j];
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
