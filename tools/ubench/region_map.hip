// At which granularity does the write speed of device memory vary?  (tools/ubench/region_speed.hip: equal 433 MB allocations take
// scattered line stores at 4.9-6.1 TB/s.)  One allocation of `total` GB is cut into windows of `win` MB; every window is written
// with scattered 128-byte lines (one pass, after 768 MB of other memory was written to push it out of the caches), best of 5.
//   hipcc --offload-arch=gfx950 -O3 -o region_map region_map.hip && ./region_map [total_gb] [win_mb]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(r_), __LINE__); std::exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void scat_write(f4* p, size_t n_lines, size_t mul) {
  const int sub = threadIdx.x & 7;
  for (size_t l = ((size_t) blockIdx.x * 1024 + threadIdx.x) >> 3; l < n_lines; l += ((size_t) gridDim.x * 1024) >> 3) {
    const size_t d = (l * mul) % n_lines;
    p[d * 8 + sub] = f4{1.f, 2.f, 3.f, 4.f};
  }
}
int main(int argc, char** argv) {
  const size_t total = (size_t) (argc > 1 ? std::atoi(argv[1]) : 8) << 30, win = (size_t) (argc > 2 ? std::atoi(argv[2]) : 128) << 20;
  const size_t flush_bytes = (size_t) 768 << 20;
  char *buf, *flush;
  CK(hipMalloc(&buf, total));
  CK(hipMalloc(&flush, flush_bytes));
  CK(hipMemset(buf, 0, total));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t nw = total / win, n_lines = (win / 128 - 1) | 1, fl = (flush_bytes / 128 - 1) | 1;
  std::vector<float> best(nw, 1e30f);
  for (int rep = 0; rep < 5; ++rep)
    for (size_t w = 0; w < nw; ++w) {
      hipLaunchKernelGGL(scat_write, dim3(2048), dim3(1024), 0, s, reinterpret_cast<f4*>(flush), fl, (size_t) 2654435761u);
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(scat_write, dim3(2048), dim3(1024), 0, s, reinterpret_cast<f4*>(buf + w * win), n_lines, (size_t) 2654435761u);
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best[w])
        best[w] = ms;
    }
  std::printf("buffer %p, %zu windows of %zu MB, GB/s of scattered line stores per window:\n", (void*) buf, nw, win >> 20);
  for (size_t w = 0; w < nw; ++w)
    std::printf("%5.0f%s", win / best[w] * 1e-6, (w % 16 == 15 || w + 1 == nw) ? "\n" : " ");
  return 0;
}
