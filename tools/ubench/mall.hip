// Does the 256 MiB Infinity Cache keep a freshly WRITTEN slab so that the next kernel reads it
// at more than HBM speed?  (Decides whether row super-blocks can keep the product slab P on chip.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void wr(f32x4* p, long n4, int nt) {
  long i = (long) blockIdx.x * 256 + threadIdx.x;
  const long stride = (long) gridDim.x * 256;
  f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (; i < n4; i += stride) { if (nt) __builtin_nontemporal_store(v, p + i); else p[i] = v; }
}
__global__ __launch_bounds__(256) void rd(const f32x4* p, long n4, float* out, int nt) {
  long i = (long) blockIdx.x * 256 + threadIdx.x;
  const long stride = (long) gridDim.x * 256;
  f32x4 a = {0, 0, 0, 0};
  for (; i < n4; i += stride) { f32x4 v = nt ? __builtin_nontemporal_load(p + i) : p[i]; a += v; }
  if (a.x + a.y + a.z + a.w == 12345.f) out[0] = 1.f;
}
static float ms(hipEvent_t a, hipEvent_t b) { float t; hipEventElapsedTime(&t, a, b); return t; }

int main() {
  const long XB = 2L << 30;
  f32x4 *X, *P; float* out;
  hipMalloc(&X, XB); hipMalloc(&P, 512L << 20); hipMalloc(&out, 64);
  hipMemset(X, 0, XB);
  hipEvent_t e0, e1, e2, e3; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3);
  const int G = 2048;
  printf("%8s %3s %3s | %9s %9s %9s %9s  (GB/s)\n", "slabMB", "wnt", "rnt", "write", "read_hot", "read_mix", "read_cold");
  for (long mb : {32L, 64L, 128L, 192L, 256L, 400L}) for (int wnt = 0; wnt < 2; ++wnt) for (int rnt = 0; rnt < 2; ++rnt) {
    const long n4 = (mb << 20) / 16;
    float tw = 0, th = 0, tm = 0, tc = 0; const int reps = 5;
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(rd, dim3(G), dim3(256), 0, 0, X, XB / 16, out, 1);           // flush
      hipEventRecord(e0); hipLaunchKernelGGL(wr, dim3(G), dim3(256), 0, 0, P, n4, wnt);
      hipEventRecord(e1); hipLaunchKernelGGL(rd, dim3(G), dim3(256), 0, 0, P, n4, out, rnt);
      hipEventRecord(e2); hipEventSynchronize(e2); tw += ms(e0, e1); th += ms(e1, e2);
      hipLaunchKernelGGL(rd, dim3(G), dim3(256), 0, 0, X, XB / 16, out, 1);           // flush
      hipLaunchKernelGGL(wr, dim3(G), dim3(256), 0, 0, P, n4, wnt);
      hipLaunchKernelGGL(rd, dim3(G), dim3(256), 0, 0, X, (3 * n4) / 2, out, 1);      // unrelated stream 1.5x slab (nt)
      hipEventRecord(e0); hipLaunchKernelGGL(rd, dim3(G), dim3(256), 0, 0, P, n4, out, rnt);
      hipEventRecord(e1); hipEventSynchronize(e1); tm += ms(e0, e1);
      hipLaunchKernelGGL(wr, dim3(G), dim3(256), 0, 0, P, n4, wnt);
      hipLaunchKernelGGL(rd, dim3(G), dim3(256), 0, 0, X, XB / 16, out, 1);           // flush
      hipEventRecord(e0); hipLaunchKernelGGL(rd, dim3(G), dim3(256), 0, 0, P, n4, out, rnt);
      hipEventRecord(e1); hipEventSynchronize(e1); tc += ms(e0, e1);
    }
    const double gb = (double) (mb << 20) / 1e9 * reps * 1e3;
    printf("%8ld %3d %3d | %9.0f %9.0f %9.0f %9.0f\n", mb, wnt, rnt, gb / tw, gb / th, gb / tm, gb / tc);
  }
  return 0;
}
