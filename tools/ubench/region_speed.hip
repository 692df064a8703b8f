// Do equal-sized device allocations differ in speed?  (round 3: the cfg2 SpMV runs 292 or 313 us depending on which memory its
// 433 MB product workspace got.)  N buffers of 433 MB, each: sequential 16-byte writes, sequential reads, and 128-byte
// lines written in a scattered order by 8-lane groups -- the expand kernel's store pattern.  Best of 5, GB/s.
//   hipcc --offload-arch=gfx950 -O3 -o region_speed region_speed.hip && ./region_speed [n_buffers] [mode: 0 hipMalloc, 1 pool,
//   2 hipDeviceMallocContiguous, 3 hipDeviceMallocUncached, 4 hipDeviceMallocFinegrained]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(r_), __LINE__); std::exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void seq_write(f4* p, size_t n16) {
  for (size_t i = (size_t) blockIdx.x * 1024 + threadIdx.x; i < n16; i += (size_t) gridDim.x * 1024)
    p[i] = f4{1.f, 2.f, 3.f, 4.f};
}
__global__ __launch_bounds__(1024) void seq_read(const f4* p, size_t n16, float* out) {
  float s = 0.f;
  for (size_t i = (size_t) blockIdx.x * 1024 + threadIdx.x; i < n16; i += (size_t) gridDim.x * 1024) {
    const f4 v = __builtin_nontemporal_load(p + i);
    s += v.x + v.w;
  }
  if (s == 12345.f)
    *out = s;
}
// line l of the pass goes to line (l * 2654435761) mod n_lines (n_lines odd-ish: a permutation when coprime)
__global__ __launch_bounds__(1024) void scat_write(f4* p, size_t n_lines, size_t mul) {
  const int sub = threadIdx.x & 7;
  for (size_t l = ((size_t) blockIdx.x * 1024 + threadIdx.x) >> 3; l < n_lines; l += ((size_t) gridDim.x * 1024) >> 3) {
    const size_t d = (l * mul) % n_lines;
    p[d * 8 + sub] = f4{1.f, 2.f, 3.f, 4.f};
  }
}

int main(int argc, char** argv) {
  const int nb = argc > 1 ? std::atoi(argv[1]) : 10;
  const int mode = argc > 2 ? std::atoi(argv[2]) : 0;  // 0 hipMalloc, 1 stream-ordered pool, 2 contiguous, 3 uncached, 4 fine-grained
  const bool pool = mode == 1;
  const size_t bytes = (size_t) 433 << 20, n16 = bytes / 16, n_lines = bytes / 128 - 1;  // (n_lines odd)
  hipStream_t s;
  CK(hipStreamCreate(&s));
  float* out;
  CK(hipMalloc(&out, 4));
  std::vector<void*> bufs;
  for (int i = 0; i < nb; ++i) {
    void* p = nullptr;
    if (pool)
      CK(hipMallocAsync(&p, bytes, s));
    else if (mode >= 2)
      CK(hipExtMallocWithFlags(&p, bytes, mode == 2 ? hipDeviceMallocContiguous : mode == 3 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained));
    else
      CK(hipMalloc(&p, bytes));
    CK(hipMemsetAsync(p, 0, bytes, s));
    bufs.push_back(p);
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto time = [&](auto&& launch) {
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0, s));
      launch();
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best)
        best = ms;
    }
    return best;
  };
  for (int pass = 0; pass < 2; ++pass)
    for (int i = 0; i < nb; ++i) {
      f4* p = static_cast<f4*>(bufs[(size_t) i]);
      const float tw = time([&] { hipLaunchKernelGGL(seq_write, dim3(2048), dim3(1024), 0, s, p, n16); });
      const float tr = time([&] { hipLaunchKernelGGL(seq_read, dim3(2048), dim3(1024), 0, s, p, n16, out); });
      const float ts = time([&] { hipLaunchKernelGGL(scat_write, dim3(2048), dim3(1024), 0, s, p, n_lines, (size_t) 2654435761u); });
      std::printf("pass %d buffer %2d at %p: seq write %6.0f GB/s  seq read %6.0f GB/s  scattered lines %6.0f GB/s\n", pass, i,
                  bufs[(size_t) i], bytes / tw * 1e-6, bytes / tr * 1e-6, bytes / ts * 1e-6);
    }
  return 0;
}
