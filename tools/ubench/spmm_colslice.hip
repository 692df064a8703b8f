// Round-2 VERDICT item 8: would blocking cfg3's SpMM for the L2s pay?  (A 2M x 2M, 32 entries per row, uniform random
// columns; B 2M x 128 fp32 row-major = 1 GB; C the same.)  DESIGN.md 4.5 argues by arithmetic that slicing B by rows
// (a slice of <= 3 MB per XCD L2, C accumulated across the slices) trades every 512-byte B-row gather for a 512-byte
// read AND a 512-byte write of a C row, because with 326 slices a row of A has 0.1 entries per slice: almost every
// stored entry becomes its own (row, slice) visit.  This measures both access patterns on the real sizes:
//   gather    one wavefront per row of A: 32 B rows gathered from anywhere in the 1 GB (HBM), accumulated in registers,
//             the C row written once                                    -- what spmm_rowgroup_kernel does today
//   colslice  one wavefront per (row, slice) visit: the B row comes from a 3 MB window (L2 resident, one window per
//             XCD), the C row is read, updated and written back         -- the column-sliced variant, 64 M visits
// Columns are hashed, not stored (the 0.5 GB of A is the same for both and left out).  Output: ms and the bytes each
// variant asks the fabric for (run under rocprofv3 --pmc TCC_EA0_RDREQ... for the measured traffic).
// Build: hipcc --offload-arch=gfx950 -O3 -o spmm_colslice spmm_colslice.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// 4 rows per workgroup, a wavefront per row; lane l owns columns 2l, 2l+1 of the 128
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ B, float* __restrict__ C, unsigned m, unsigned k,
                                                     int per_row) {
  const unsigned row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= m)
    return;
  f32x2 acc = {0.f, 0.f};
  for (int j0 = 0; j0 < per_row; j0 += 4) {
    f32x2 b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned col = (unsigned) (((unsigned long long) hash32(row * 131u + j0 + u) * k) >> 32);
      b[u] = *reinterpret_cast<const f32x2*>(B + (size_t) col * 128 + 2 * lane);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      acc += 0.5f * b[u];
  }
  *reinterpret_cast<f32x2*>(C + (size_t) row * 128 + 2 * lane) = acc;
}

// visit v of slice s: row = hash (anywhere), B row inside the slice's window; 4 visits per workgroup.  Workgroup i runs
// on XCD i % 8: the eight XCDs work on eight different slices at a time (window = s * wrows .. + wrows).
__global__ __launch_bounds__(256) void colslice_kernel(const float* __restrict__ B, float* __restrict__ C, unsigned m, unsigned k,
                                                       unsigned wrows, unsigned visits_per_slice, unsigned slice0) {
  const unsigned xcd = blockIdx.x & 7, local = (blockIdx.x >> 3) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (local >= visits_per_slice)
    return;
  const unsigned slice = slice0 + xcd;
  const unsigned h = hash32(slice * 2654435761u + local);
  const unsigned row = (unsigned) (((unsigned long long) h * m) >> 32);
  const unsigned col = slice * wrows + (unsigned) (((unsigned long long) hash32(h) * wrows) >> 32);
  if (col >= k)
    return;
  const f32x2 b = *reinterpret_cast<const f32x2*>(B + (size_t) col * 128 + 2 * lane);
  f32x2* cp = reinterpret_cast<f32x2*>(C + (size_t) row * 128 + 2 * lane);
  *cp = *cp + 0.5f * b;  // (two visits of one row in one launch would race: a timing kernel, not a product)
}

int main() {
  const unsigned m = 2000000, k = 2000000;
  const int per_row = 32;
  float *B, *C;
  CHECK(hipMalloc(&B, (size_t) k * 128 * 4));
  CHECK(hipMalloc(&C, (size_t) m * 128 * 4));
  CHECK(hipMemset(B, 0, (size_t) k * 128 * 4));
  CHECK(hipMemset(C, 0, (size_t) m * 128 * 4));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(gather_kernel, dim3((m + 3) / 4), dim3(256), 0, 0, B, C, m, k, per_row);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double entries = (double) m * per_row;
  printf("gather   : %8.3f ms   asks for %6.2f GB (B rows) + %5.2f GB (C once)\n", ms, entries * 512 / 1e9, m * 512.0 / 1e9);
  for (unsigned wkb : {3072u, 1024u}) {
    const unsigned wrows = wkb * 1024u / 512u;                      // B rows per window
    const unsigned nslices = (k + wrows - 1) / wrows;
    const unsigned visits = (unsigned) (entries / nslices);          // ~ one visit per stored entry
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (unsigned s0 = 0; s0 < nslices; s0 += 8)                   // eight slices at a time, one per XCD
        hipLaunchKernelGGL(colslice_kernel, dim3(8 * ((visits + 3) / 4)), dim3(256), 0, 0, B, C, m, k, wrows, visits, s0);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("colslice : %8.3f ms   window %u KB, %u slices, %u visits each: asks for %6.2f GB (C read + write) + %5.2f GB (B once)\n", ms,
           wkb, nslices, visits, (double) visits * nslices * 1024 / 1e9, k * 512.0 / 1e9);
  }
  CHECK(hipDeviceSynchronize());
  return 0;
}
