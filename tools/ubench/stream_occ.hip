// Does the reduce's occupancy (8 wavefronts per CU: its LDS accumulators) cap its stream rate?  One wavefront per "bin"
// streams a contiguous part of P (16 B per lane) and of the row codes (4 B per lane) with UB loads in flight per array, as
// pb_reduce_kernel does; LDS per workgroup limits the wavefronts per CU.  Output: GB/s by (waves per CU, UB).
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_occ stream_occ.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int UB>
__global__ __launch_bounds__(256) void stream_kernel(const f32x4* __restrict__ P, const unsigned* __restrict__ code,
                                                     float* __restrict__ out, long groups_per_bin, int nbins) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long b = (long) blockIdx.x * 4 + wave;
  if (b >= nbins)
    return;
  const long g0 = b * groups_per_bin, g1 = g0 + groups_per_bin;
  float acc = 0.f;
  f32x4 pa[UB], pb[UB];
  unsigned ca[UB], cb[UB];
  auto issue = [&](long g, f32x4 (&p)[UB], unsigned (&c)[UB]) {
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const long gg = g + u < g1 ? g + u : g1 - 1;
      p[u] = __builtin_nontemporal_load(P + gg * 64 + lane);
      c[u] = __builtin_nontemporal_load(code + gg * 64 + lane);
    }
  };
  auto consume = [&](f32x4 (&p)[UB], unsigned (&c)[UB]) {
#pragma unroll
    for (int u = 0; u < UB; ++u)
      acc += p[u].x + p[u].y + p[u].z + p[u].w + (float) c[u];
  };
  issue(g0, pa, ca);
  for (long g = g0; g < g1; g += 2 * UB) {
    issue(g + UB, pb, cb);
    consume(pa, ca);
    issue(g + 2 * UB, pa, ca);
    consume(pb, cb);
  }
  lds[threadIdx.x] = acc;
  if (acc == 1234.5f)
    out[0] = lds[0];
}

template <int UB>
static float run(const f32x4* P, const unsigned* code, float* out, long gpb, int nbins, size_t lds) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<UB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 2; ++i)
    hipLaunchKernelGGL(stream_kernel<UB>, dim3(nbins / 4), dim3(256), lds, 0, P, code, out, gpb, nbins);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i)
    hipLaunchKernelGGL(stream_kernel<UB>, dim3(nbins / 4), dim3(256), lds, 0, P, code, out, gpb, nbins);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}

int main() {
  const long entries = 108000000L / 256 * 256;
  f32x4* P;
  unsigned* code;
  float* out;
  CHECK(hipMalloc(&P, entries * 4 + 65536));
  CHECK(hipMalloc(&code, entries + 65536));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(P, 0, entries * 4 + 65536));
  CHECK(hipMemset(code, 0, entries + 65536));
  printf("%8s %10s | %8s %8s %8s %8s   (GB/s of 5 B per entry, %ld entries)\n", "bins", "waves/CU", "UB=1", "UB=2", "UB=4", "UB=8", entries);
  for (int nbins : {2048, 4096, 8192}) {
    const long gpb = entries / 256 / nbins;
    for (size_t lds : {(size_t) 80 * 1024, (size_t) 40 * 1024, (size_t) 20 * 1024, (size_t) 1024}) {
      const int wgs = (int) (160 * 1024 / lds) < 8 ? (int) (160 * 1024 / lds) : 8;
      const double gb = (double) gpb * nbins * 256 * 5 / 1e6;
      printf("%8d %10d | %8.0f %8.0f %8.0f %8.0f\n", nbins, wgs * 4, gb / run<1>(P, code, out, gpb, nbins, lds),
             gb / run<2>(P, code, out, gpb, nbins, lds), gb / run<4>(P, code, out, gpb, nbins, lds),
             gb / run<8>(P, code, out, gpb, nbins, lds));
    }
  }
  return 0;
}
