// Is physically scattered memory faster to write than physically contiguous memory?  (profiles/r03_store_trial.md: equal
// allocations differ by up to 20 % in write speed; contiguous ones are never the fast kind.)  A 432 MB virtual range is backed
// through the virtual-memory API by chunks of `chunk` MB created one after the other and mapped (a) in creation order,
// (b) in a shuffled order, (c) every other chunk of twice as many (the rest released); against plain hipMalloc.
//   hipcc --offload-arch=gfx950 -O3 -o vmm_scatter vmm_scatter.hip && ./vmm_scatter [chunk_mb]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { std::printf("HIP error %s (%d) line %d\n", hipGetErrorString(r_), (int) r_, __LINE__); std::exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void scat_write(f4* p, size_t n_lines, size_t mul) {
  const int sub = threadIdx.x & 7;
  for (size_t l = ((size_t) blockIdx.x * 1024 + threadIdx.x) >> 3; l < n_lines; l += ((size_t) gridDim.x * 1024) >> 3) {
    const size_t d = (l * mul) % n_lines;
    p[d * 8 + sub] = f4{1.f, 2.f, 3.f, 4.f};
  }
}
__global__ __launch_bounds__(1024) void seq_write(f4* p, size_t n16) {
  for (size_t i = (size_t) blockIdx.x * 1024 + threadIdx.x; i < n16; i += (size_t) gridDim.x * 1024)
    p[i] = f4{1.f, 2.f, 3.f, 4.f};
}
__global__ __launch_bounds__(1024) void seq_read(const f4* p, size_t n16, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t) blockIdx.x * 1024 + threadIdx.x; i < n16; i += (size_t) gridDim.x * 1024) {
    const f4 v = __builtin_nontemporal_load(p + i);
    acc += v.x + v.w;
  }
  if (acc == 12345.f)
    *out = acc;
}
static float* g_out;
static hipStream_t s;
static hipEvent_t e0, e1;
static void measure(const char* what, void* p, size_t bytes) {
  const size_t n_lines = (bytes / 128 - 1) | 1;
  float best[3] = {1e30f, 1e30f, 1e30f};
  for (int rep = 0; rep < 6; ++rep)
    for (int k = 0; k < 3; ++k) {
      CK(hipEventRecord(e0, s));
      if (k == 0)
        hipLaunchKernelGGL(scat_write, dim3(2048), dim3(1024), 0, s, static_cast<f4*>(p), n_lines, (size_t) 2654435761u);
      else if (k == 1)
        hipLaunchKernelGGL(seq_write, dim3(2048), dim3(1024), 0, s, static_cast<f4*>(p), bytes / 16);
      else
        hipLaunchKernelGGL(seq_read, dim3(2048), dim3(1024), 0, s, static_cast<const f4*>(p), bytes / 16, g_out);
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best[k])
        best[k] = ms;
    }
  std::printf("%-44s scattered lines %6.0f GB/s   sequential write %6.0f   read %6.0f\n", what, bytes / best[0] * 1e-6,
              bytes / best[1] * 1e-6, bytes / best[2] * 1e-6);
}
int main(int argc, char** argv) {
  const size_t chunk = (size_t) (argc > 1 ? std::atoi(argv[1]) : 2) << 20, bytes = (size_t) 432 << 20, nc = bytes / chunk;
  int dev = 0;
  CK(hipSetDevice(dev));
  CK(hipStreamCreate(&s));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipMalloc(&g_out, 4));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  std::printf("granularity %zu KB, chunk %zu MB, %zu chunks\n", gran >> 10, chunk >> 20, nc);
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  for (int round = 0; round < 2; ++round) {
    void* plain = nullptr;
    CK(hipMalloc(&plain, bytes));
    measure("hipMalloc", plain, bytes);
    for (int mode = 0; mode < 4; ++mode) {
      const size_t csz = mode == 3 ? bytes : chunk, ncm = mode == 3 ? 1 : nc;
      const size_t make = mode == 2 ? 2 * nc : ncm;
      std::vector<hipMemGenericAllocationHandle_t> h(make);
      for (size_t i = 0; i < make; ++i)
        CK(hipMemCreate(&h[i], csz, &prop, 0));
      std::vector<size_t> order(ncm);
      if (mode == 2)
        for (size_t i = 0; i < ncm; ++i)
          order[i] = 2 * i;
      else
        std::iota(order.begin(), order.end(), (size_t) 0);
      if (mode == 1) {
        std::mt19937 g(7);
        std::shuffle(order.begin(), order.end(), g);
      }
      void* va = nullptr;
      CK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
      for (size_t i = 0; i < ncm; ++i)
        CK(hipMemMap(static_cast<char*>(va) + i * csz, csz, 0, h[order[i]], 0));
      CK(hipMemSetAccess(va, bytes, &acc, 1));
      measure(mode == 0 ? "VMM chunks, creation order" : mode == 1 ? "VMM chunks, shuffled" : mode == 2 ? "VMM every other chunk of 2x" : "VMM one handle for the whole range", va, bytes);
      CK(hipMemUnmap(va, bytes));
      CK(hipMemAddressFree(va, bytes));
      for (size_t i = 0; i < make; ++i)
        CK(hipMemRelease(h[i]));
    }
    CK(hipFree(plain));
  }
  return 0;
}
