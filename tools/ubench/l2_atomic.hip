// Round-2 VERDICT item 2(b): can products be added into a y window that stays resident in ONE XCD's L2 with float
// atomics, instead of making the HBM round trip through P?  (x slices in LDS, rows partitioned by XCD, a window of
// <= 3 MB of y per XCD; the scheme wins only above ~450 G atomics/s chip-wide.)
//
// Every workgroup adds N/G pseudo-random floats into the window of the XCD it runs on (XCC_ID is read in the kernel;
// the count of workgroups that did not land on XCD blockIdx % 8 is reported).  Variants:
//   scope   wg    = __HIP_MEMORY_SCOPE_WORKGROUP (no sc bits: the RMW happens in the issuing XCD's L2)
//           agent = __HIP_MEMORY_SCOPE_AGENT     (what atomicAdd() emits: device-coherent)
//   order   rand = every lane a random address of the window; sorted = a wavefront's 64 addresses ascending inside one
//                  4 KiB stretch (what a row-sorted run would look like)
//   window  bytes per XCD
// Output: G atomics/s.   Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o l2_atomic l2_atomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int SCOPE, int SORTED>
__global__ __launch_bounds__(1024) void atomics(float* __restrict__ y, unsigned win_elems, long per_thread,
                                               unsigned* __restrict__ misplaced) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 0xf;
  if (threadIdx.x == 0 && xcc != (blockIdx.x & 7))
    atomicAdd(misplaced, 1u);
  float* win = y + (size_t) xcc * win_elems;  // the window of the XCD this workgroup REALLY runs on
  unsigned s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  const unsigned lane = threadIdx.x & 63;
  for (long i = 0; i < per_thread; ++i) {
    s = s * 1664525u + 1013904223u;
    unsigned idx;
    if (SORTED) {
      // wave-uniform base of a 1 024-element stretch + lane * 16 + small jitter: ascending inside the wavefront
      const unsigned base = __builtin_amdgcn_readfirstlane(s >> 8) % (win_elems - 1024u);
      idx = base + lane * 16u + ((s >> 4) & 15u);
    } else {
      idx = (unsigned) (((unsigned long long) (s >> 4) * win_elems) >> 28);
    }
    __hip_atomic_fetch_add(win + idx, 1.0f, __ATOMIC_RELAXED, SCOPE);
  }
}

// the same loop with a plain (racy, wrong) read-modify-write: what the memory path costs without the atomic unit
__global__ __launch_bounds__(1024) void plain_rmw(float* __restrict__ y, unsigned win_elems, long per_thread) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float* win = y + (size_t) (xcc & 0xf) * win_elems;
  unsigned s = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  for (long i = 0; i < per_thread; ++i) {
    s = s * 1664525u + 1013904223u;
    const unsigned idx = (unsigned) (((unsigned long long) (s >> 4) * win_elems) >> 28);
    win[idx] += 1.0f;
  }
}

template <typename F>
static float time_ms(F f, int reps = 4) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i)
    f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  float* y;
  unsigned* mis;
  const size_t cap = 8ull * 16 * 1024 * 1024;  // up to 16 MB per XCD
  CHECK(hipMalloc(&y, cap));
  CHECK(hipMalloc(&mis, 4));
  CHECK(hipMemset(y, 0, cap));
  CHECK(hipMemset(mis, 0, 4));
  const int G = 512;  // two workgroups of 1 024 threads per CU
  const long total = 100000000L, per_thread = total / ((long) G * 1024);
  const double n = (double) per_thread * G * 1024;
  printf("%10s | %9s %9s %9s %9s %9s   (G atomics/s, %.0f M float adds chip-wide; 450 G/s needed)\n", "window/XCD", "wg rand",
         "agent rand", "wg sorted", "agent sort", "plain rmw", n / 1e6);
  for (unsigned kb : {256u, 1024u, 3072u, 4096u, 16384u}) {
    const unsigned we = kb * 256u;
    float t[5];
    t[0] = time_ms([&] { hipLaunchKernelGGL((atomics<__HIP_MEMORY_SCOPE_WORKGROUP, 0>), dim3(G), dim3(1024), 0, 0, y, we, per_thread, mis); });
    t[1] = time_ms([&] { hipLaunchKernelGGL((atomics<__HIP_MEMORY_SCOPE_AGENT, 0>), dim3(G), dim3(1024), 0, 0, y, we, per_thread, mis); });
    t[2] = time_ms([&] { hipLaunchKernelGGL((atomics<__HIP_MEMORY_SCOPE_WORKGROUP, 1>), dim3(G), dim3(1024), 0, 0, y, we, per_thread, mis); });
    t[3] = time_ms([&] { hipLaunchKernelGGL((atomics<__HIP_MEMORY_SCOPE_AGENT, 1>), dim3(G), dim3(1024), 0, 0, y, we, per_thread, mis); });
    t[4] = time_ms([&] { hipLaunchKernelGGL(plain_rmw, dim3(G), dim3(1024), 0, 0, y, we, per_thread); });
    printf("%7u KB | %9.1f %9.1f %9.1f %9.1f %9.1f   us: %.0f %.0f %.0f %.0f %.0f\n", kb, n / t[0] / 1e6, n / t[1] / 1e6, n / t[2] / 1e6,
           n / t[3] / 1e6, n / t[4] / 1e6, t[0] * 1e3, t[1] * 1e3, t[2] * 1e3, t[3] * 1e3, t[4] * 1e3);
  }
  unsigned h = 0;
  CHECK(hipMemcpy(&h, mis, 4, hipMemcpyDeviceToHost));
  printf("workgroup launches that did not run on XCD blockIdx %% 8: %u\n", h);
  return 0;
}
