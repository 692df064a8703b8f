// What does one random 4-byte (or 8-byte) gather cost the memory system, by load flavour?
//   plain      global_load_dword                      (what the CSR kernels issue for x[col])
//   nt         __builtin_nontemporal_load             (nt bit)
//   agent      relaxed atomic load, agent scope       (sc1: what the SpTRSV kernels use across XCDs)
//   system     relaxed atomic load, system scope      (sc0 sc1)
// 1e8 gathers from a table of `n` floats with a 64-bit LCG per lane (no index array: the stream is the gathers alone).
// Prints time and implied bytes per gather if the fabric ran at 6.3 TB/s; the rocprofv3 request-size counters
// (TCC_EA0_RDREQ_32B / _64B / _128B) say what was really fetched.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__device__ __forceinline__ float ld(const float* p) {
  if (MODE == 0) return *p;
  if (MODE == 1) return __builtin_nontemporal_load(p);
  if (MODE == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ table, unsigned long long n, int per_lane,
                                                     float* __restrict__ out) {
  unsigned long long s = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345ull;
  float acc = 0.f;
  for (int i = 0; i < per_lane; i += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      v[u] = ld<MODE>(table + (s >> 20) % n);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      acc += v[u];
  }
  if (acc == 123.456f)
    out[0] = acc;
}

int main(int argc, char** argv) {
  const unsigned long long n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ull;  // 400 MB
  float *table, *out;
  CHECK(hipMalloc(&table, n * 4));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(table, 0, n * 4));
  const int blocks = 256 * 16, per_lane = 96;  // 256*16*256*96 = 1.007e8 gathers
  const double gathers = (double) blocks * 256 * per_lane;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const char* names[4] = {"plain", "nt", "agent", "system"};
  for (int mode = 0; mode < 4; ++mode) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(gather_kernel<0>, dim3(blocks), dim3(256), 0, 0, table, n, per_lane, out);
      if (mode == 1) hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 0, 0, table, n, per_lane, out);
      if (mode == 2) hipLaunchKernelGGL(gather_kernel<2>, dim3(blocks), dim3(256), 0, 0, table, n, per_lane, out);
      if (mode == 3) hipLaunchKernelGGL(gather_kernel<3>, dim3(blocks), dim3(256), 0, 0, table, n, per_lane, out);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      if (rep > 0 && ms < best) best = ms;
    }
    printf("%-7s table %6.0f MB: %8.3f ms for %.3g gathers = %6.1f G gathers/s  (= %5.1f B per gather at 6.3 TB/s)\n", names[mode],
           n * 4 / 1e6, best, gathers, gathers / best / 1e6, 6.3e12 * best * 1e-3 / gathers);
  }
  CHECK(hipDeviceSynchronize());
  return 0;
}
