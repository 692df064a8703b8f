// Microbenchmarks behind two design decisions of the sliced SpMV (results in DESIGN.md):
//  (1) LDS atomic throughput: ds_add_f32 vs ds_add_u32, random vs lane-linear addresses
//  (2) 4-byte scattered global stores: every 128-B line receives words from 32 different
//      workgroups (the "write P in CSR order" alternative)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void lds_atomic_kernel(int iters, int span, float* out) {
  extern __shared__ float acc[];
  for (int i = threadIdx.x; i < span; i += 1024) acc[i] = 0.f;
  __syncthreads();
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  for (int it = 0; it < iters; ++it) {
    h = h * 1664525u + 1013904223u;
    int idx;
    if (MODE & 1) idx = (threadIdx.x + it * 1024) % span;      // lane-linear, conflict free
    else idx = (h >> 8) % span;                                 // random
    if (MODE & 8) {                                                            // float add as read + cmpswap loop
      int* ai = reinterpret_cast<int*>(acc) + idx;
      int old = *reinterpret_cast<volatile int*>(ai);
      while (true) {
        const int assumed = old;
        old = atomicCAS(ai, assumed, __float_as_int(__int_as_float(assumed) + 1.0f));
        if (old == assumed) break;
      }
    } else if (MODE & 2) atomicAdd(reinterpret_cast<unsigned*>(acc) + idx, 1u);      // ds_add_u32
    else if (MODE & 4) acc[idx] += 1.0f;                                      // plain RMW (racy; rate only)
    else unsafeAtomicAdd(acc + idx, 1.0f);                                    // ds_add_f32
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = acc[0] + acc[span - 1];
}

__global__ __launch_bounds__(1024) void scatter_store_kernel(long n_per_block, int nblocks, float* P) {
  // word j of block b goes to P[j * nblocks + b]: each line is shared by 32 consecutive blocks
  for (long j = threadIdx.x; j < n_per_block; j += 1024)
    P[j * nblocks + blockIdx.x] = (float) j;
}

__global__ __launch_bounds__(1024) void linear_store_kernel(long n_per_block, float* P) {
  for (long j = threadIdx.x; j < n_per_block; j += 1024)
    P[(long) blockIdx.x * n_per_block + j] = (float) j;
}

template <typename F>
static float time_ms(F f, int reps = 5) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  float* out;
  CHECK(hipMalloc(&out, 4096 * 4));
  const int span = 19532, iters = 195, blocks = 512;  // = cfg2's reduce: 1e8 updates
  const double total = (double) blocks * 1024 * iters;
  const char* names[] = {"ds_add_f32 random", "ds_add_f32 linear", "ds_add_u32 random", "ds_add_u32 linear",
                         "plain rmw random", "plain rmw linear", "cas-loop f32 random", "cas-loop f32 linear"};
  hipFuncSetAttribute((const void*) lds_atomic_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipFuncSetAttribute((const void*) lds_atomic_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipFuncSetAttribute((const void*) lds_atomic_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipFuncSetAttribute((const void*) lds_atomic_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipFuncSetAttribute((const void*) lds_atomic_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipFuncSetAttribute((const void*) lds_atomic_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipFuncSetAttribute((const void*) lds_atomic_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipFuncSetAttribute((const void*) lds_atomic_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  float ms[8];
  ms[0] = time_ms([&] { hipLaunchKernelGGL(lds_atomic_kernel<0>, dim3(blocks), dim3(1024), span * 4, 0, iters, span, out); });
  ms[1] = time_ms([&] { hipLaunchKernelGGL(lds_atomic_kernel<1>, dim3(blocks), dim3(1024), span * 4, 0, iters, span, out); });
  ms[2] = time_ms([&] { hipLaunchKernelGGL(lds_atomic_kernel<2>, dim3(blocks), dim3(1024), span * 4, 0, iters, span, out); });
  ms[3] = time_ms([&] { hipLaunchKernelGGL(lds_atomic_kernel<3>, dim3(blocks), dim3(1024), span * 4, 0, iters, span, out); });
  ms[4] = time_ms([&] { hipLaunchKernelGGL(lds_atomic_kernel<4>, dim3(blocks), dim3(1024), span * 4, 0, iters, span, out); });
  ms[5] = time_ms([&] { hipLaunchKernelGGL(lds_atomic_kernel<5>, dim3(blocks), dim3(1024), span * 4, 0, iters, span, out); });
  ms[6] = time_ms([&] { hipLaunchKernelGGL(lds_atomic_kernel<8>, dim3(blocks), dim3(1024), span * 4, 0, iters, span, out); });
  ms[7] = time_ms([&] { hipLaunchKernelGGL(lds_atomic_kernel<9>, dim3(blocks), dim3(1024), span * 4, 0, iters, span, out); });
  for (int i = 0; i < 8; ++i)
    printf("%-20s %8.1f us  %7.1f Gupdates/s  (%.3f lanes/clk/CU @2.4GHz)\n", names[i], ms[i] * 1e3, total / ms[i] / 1e6,
           total / (ms[i] * 1e-3) / 256 / 2.4e9);

  float* P;
  const long n = 100000000;
  CHECK(hipMalloc(&P, n * 4));
  const long per = n / blocks;
  float t1 = time_ms([&] { hipLaunchKernelGGL(scatter_store_kernel, dim3(blocks), dim3(1024), 0, 0, per, blocks, P); });
  float t2 = time_ms([&] { hipLaunchKernelGGL(linear_store_kernel, dim3(blocks), dim3(1024), 0, 0, per, P); });
  printf("scattered 4B stores (line shared by 32 blocks): %8.1f us  %.2f TB/s useful\n", t1 * 1e3, n * 4.0 / t1 / 1e9);
  printf("linear 4B stores:                               %8.1f us  %.2f TB/s\n", t2 * 1e3, n * 4.0 / t2 / 1e9);
  return 0;
}
