// Does the 256 MiB Infinity Cache (MALL) add bandwidth on top of HBM?  tools/ubench/mall.hip (round 1) read a
// freshly written slab with ONE 16-byte load in flight per thread and found "no bandwidth to win" -- but 32 waves x
// 1 KiB per CU in flight is itself a ~6.5 TB/s latency bound, so that test could not see a faster level.  Here every
// thread keeps 8 x 16 B in flight.
//   rd      repeated reads of a slab of B bytes (B <= MALL: hits after the first pass)
//   wr->rd  write the slab, then read it (the product round trip of the sliced SpMV when cut into row chunks)
//   mix     one kernel reads a big stream from HBM (HB bytes) AND the slab (fresh from the writer) at once
// Output: TB/s of the bytes the kernel touches.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NT>
__global__ __launch_bounds__(256) void rd(const f32x4* __restrict__ p, long n4, float* out) {
  const long per = (n4 + gridDim.x - 1) / gridDim.x;  // contiguous share per workgroup
  const long lo = (long) blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
  f32x4 a = {0, 0, 0, 0};
  for (long i = lo + threadIdx.x; i < hi; i += 8 * 256) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long q = i + u * 256 < hi ? i + u * 256 : hi - 1;
      v[u] = NT ? __builtin_nontemporal_load(p + q) : p[q];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      a += v[u];
  }
  if (a.x + a.y + a.z + a.w == 12345.f)
    out[0] = 1.f;
}

template <int NT>
__global__ __launch_bounds__(256) void wr(f32x4* __restrict__ p, long n4, float s) {
  const long per = (n4 + gridDim.x - 1) / gridDim.x;
  const long lo = (long) blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
  const f32x4 v = {s, s + 1, s + 2, s + 3};
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    if (NT)
      __builtin_nontemporal_store(v, p + i);
    else
      p[i] = v;
  }
}

// reads na f32x4 from a (big, HBM) and nb from b (slab) in the same loop, proportionally interleaved
__global__ __launch_bounds__(256) void rd2(const f32x4* __restrict__ a, long na, const f32x4* __restrict__ b, long nb,
                                           float* out) {
  const long pa = (na + gridDim.x - 1) / gridDim.x, pb = (nb + gridDim.x - 1) / gridDim.x;
  const long la = (long) blockIdx.x * pa, ha = la + pa < na ? la + pa : na;
  const long lb = (long) blockIdx.x * pb, hb = lb + pb < nb ? lb + pb : nb;
  f32x4 acc = {0, 0, 0, 0};
  const long steps = (pa + 4 * 256 - 1) / (4 * 256);
  const long bstep = (pb + steps - 1) / steps;  // slab elements per step (per workgroup)
  for (long st = 0; st < steps; ++st) {
    f32x4 v[4], w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      long q = la + st * 4 * 256 + u * 256 + threadIdx.x;
      q = q < ha ? q : ha - 1;
      v[u] = __builtin_nontemporal_load(a + q);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      long q = lb + st * bstep + u * 256 + threadIdx.x;
      const bool ok = u * 256 + threadIdx.x < bstep && q < hb;
      q = q < hb ? q : hb - 1;
      w[u] = ok ? b[q] : f32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      acc += v[u] + w[u];
  }
  if (acc.x + acc.y == 12345.f)
    out[0] = 1.f;
}

static float ms(hipEvent_t a, hipEvent_t b) {
  float t;
  (void) hipEventElapsedTime(&t, a, b);
  return t;
}

int main() {
  const long XB = 3L << 30;
  f32x4 *X, *P;
  float* out;
  CHECK(hipMalloc(&X, XB));
  CHECK(hipMalloc(&P, 1L << 30));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(X, 0, XB));
  CHECK(hipMemset(P, 0, 1L << 30));
  hipEvent_t e0, e1, e2;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventCreate(&e2);
  const int G = 2048;
  printf("%7s | %8s %8s | %8s %8s | %8s (TB/s)   [mix: HBM stream 2x the slab, read together with the fresh slab]\n", "slabMB",
         "rd_rep", "rd_rep_nt", "wr", "rd_after", "mix");
  for (long mb : {16L, 32L, 64L, 96L, 128L, 160L, 192L, 256L, 384L, 512L, 1024L}) {
    const long n4 = (mb << 20) / 16;
    const double tb = (double) (mb << 20) / 1e12;
    float t_rep = 0, t_rep_nt = 0, t_w = 0, t_r = 0, t_mix = 0;
    const int reps = 6;
    // repeated reads
    hipLaunchKernelGGL(rd<0>, dim3(G), dim3(256), 0, 0, P, n4, out);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
      hipLaunchKernelGGL(rd<0>, dim3(G), dim3(256), 0, 0, P, n4, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    t_rep = ms(e0, e1) / reps;
    hipLaunchKernelGGL(rd<1>, dim3(G), dim3(256), 0, 0, P, n4, out);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
      hipLaunchKernelGGL(rd<1>, dim3(G), dim3(256), 0, 0, P, n4, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    t_rep_nt = ms(e0, e1) / reps;
    // write then read, with an HBM flush (big nt read) before each pair so the slab starts cold
    float t_wn = 0, t_rn = 0, t_rp = 0;
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(rd<1>, dim3(G), dim3(256), 0, 0, X, XB / 16, out);
      hipEventRecord(e0);
      hipLaunchKernelGGL(wr<0>, dim3(G), dim3(256), 0, 0, P, n4, (float) r);
      hipEventRecord(e1);
      hipLaunchKernelGGL(rd<1>, dim3(G), dim3(256), 0, 0, P, n4, out);
      hipEventRecord(e2);
      hipEventSynchronize(e2);
      t_w += ms(e0, e1);
      t_r += ms(e1, e2);
      hipLaunchKernelGGL(rd<1>, dim3(G), dim3(256), 0, 0, X, XB / 16, out);
      hipEventRecord(e0);
      hipLaunchKernelGGL(wr<1>, dim3(G), dim3(256), 0, 0, P, n4, (float) r);
      hipEventRecord(e1);
      hipLaunchKernelGGL(rd<1>, dim3(G), dim3(256), 0, 0, P, n4, out);
      hipEventRecord(e2);
      hipEventSynchronize(e2);
      t_wn += ms(e0, e1);
      t_rn += ms(e1, e2);
      hipLaunchKernelGGL(rd<1>, dim3(G), dim3(256), 0, 0, X, XB / 16, out);
      hipLaunchKernelGGL(wr<0>, dim3(G), dim3(256), 0, 0, P, n4, (float) r);
      hipEventRecord(e1);
      hipLaunchKernelGGL(rd<0>, dim3(G), dim3(256), 0, 0, P, n4, out);
      hipEventRecord(e2);
      hipEventSynchronize(e2);
      t_rp += ms(e1, e2);
    }
    t_w /= reps;
    t_r /= reps;
    printf("        wr_nt %.2f rd_after_nt_wr %.2f | plain-load rd_after %.2f | pair plain %.2f pair nt %.2f TB/s\n", tb / (t_wn / reps * 1e-3),
           tb / (t_rn / reps * 1e-3), tb / (t_rp / reps * 1e-3), 2 * tb / ((t_w + t_r) * 1e-3), 2 * tb / ((t_wn + t_rn) / reps * 1e-3));
    // mixed: write the slab, then ONE kernel reads 2x slab bytes from a cold part of X plus the slab
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(rd<1>, dim3(G), dim3(256), 0, 0, X + (1L << 30) / 16, (2L << 30) / 16, out);  // flush with the far part of X
      hipLaunchKernelGGL(wr<0>, dim3(G), dim3(256), 0, 0, P, n4, (float) r);
      hipEventRecord(e0);
      hipLaunchKernelGGL(rd2, dim3(G), dim3(256), 0, 0, X, 2 * n4, P, n4, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      t_mix += ms(e0, e1);
    }
    t_mix /= reps;
    printf("%7ld | %8.2f %8.2f | %8.2f %8.2f | %8.2f\n", mb, tb / (t_rep * 1e-3), tb / (t_rep_nt * 1e-3), tb / (t_w * 1e-3),
           tb / (t_r * 1e-3), 3 * tb / (t_mix * 1e-3));
  }
  CHECK(hipDeviceSynchronize());
  return 0;
}
