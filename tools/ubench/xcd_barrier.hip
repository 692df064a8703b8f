// Round-4 VERDICT item 6: "give the tail to ONE XCD (32 CUs, one coherent L2 -> a workgroup-counter barrier in that L2 instead
// of the 8-L2 grid barrier) ... measure barrier cost per level both ways first".  This is that measurement.
//
// One cooperative launch of 256 workgroups x 1024 lanes runs N rounds of { hand one value to a neighbour, barrier }:
//   barrier  grid : all workgroups, the two-level arrival + one release line per workgroup of csrc/sptrsv.hip (agent scope)
//            xcd  : only the workgroups that run on XCC 0 (the others leave at once); the arrival counter and the release word
//                   are touched with L2-level operations only (atomics without sc1, reads as returning atomic adds of zero: they are served
//                   by the one L2 all participants share)
//   hand-off none : nothing but the barrier
//            agent: store with agent scope (write-through to memory), wait for it, barrier, load with agent scope -- what the
//                   solve does today
//            l2   : plain store (acknowledged by the L2), wait, barrier, read it back with an L2 atomic
// Output: microseconds per round, and whether every value arrived.
// Build: hipcc --offload-arch=gfx950 -O3 -o xcd_barrier xcd_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// A read that is served by the XCD's L2: a returning atomic add of zero WITHOUT sc1 (atomics are executed in the L2; a load
// with sc0 -- "workgroup scope" -- may hit in the CU's L1 and never saw the other CUs' stores: the first version of this
// program polled such loads and every round ran into its spin limit; a load with sc1 goes past the L2 to the fabric).
// (Inline assembly: the compiler turns __hip_atomic_fetch_add(p, 0, relaxed, workgroup) back into such an sc0 load.)
__device__ __forceinline__ unsigned load_l2(unsigned* p) {
  unsigned v;
  const unsigned zero = 0;
  asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(zero) : "memory");
  return v;
}

// ctl: [0] arrival counter (xcd) / top counter (grid), [32 * (1 + g)] group counters, [32 * 9] release word (xcd),
//      [32 * (10 + w)] release lines (grid), [32 * 300] ticket, [32 * 301] errors, [32 * 302] participants seen
template <int BARRIER, int HANDOFF>
__global__ __launch_bounds__(1024) void rounds(unsigned* ctl, unsigned* x, int n_rounds, int spin_limit) {
  __shared__ unsigned s_rank, s_last;
  __shared__ int s_abort;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 0xf;
  if (BARRIER == 1 && xcc != 0)
    return;
  if (threadIdx.x == 0) {
    s_rank = BARRIER == 1 ? atomicAdd(&ctl[32 * 300], 1u) : blockIdx.x;
    s_abort = 0;
    atomicAdd(&ctl[32 * 302], 1u);
  }
  __syncthreads();
  const unsigned rank = s_rank;
  const unsigned P = BARRIER == 1 ? gridDim.x / 8 : gridDim.x;
  for (int it = 1; it <= n_rounds; ++it) {
    // hand-off, first half
    if (threadIdx.x == 0) {
      if (HANDOFF == 1)
        __hip_atomic_store(&x[rank * 32], (unsigned) it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (HANDOFF == 2)
        x[rank * 32] = (unsigned) it;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();
    // barrier
    if (threadIdx.x == 0) {
      if (BARRIER == 0) {
        const unsigned gid = rank & 7u, gsize = (gridDim.x - gid + 7u) >> 3;
        unsigned last = 0;
        if (__hip_atomic_fetch_add(&ctl[32 * (1 + gid)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == (unsigned) it * gsize)
          last = __hip_atomic_fetch_add(&ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == (unsigned) it * 8u;
        s_last = last;
      } else {
        s_last = __hip_atomic_fetch_add(&ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1 == (unsigned) it * P;
      }
    }
    __syncthreads();
    if (s_last) {
      if (BARRIER == 0) {
        for (unsigned w = threadIdx.x; w < gridDim.x; w += blockDim.x)
          __hip_atomic_store(&ctl[32 * (10 + w)], (unsigned) it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (threadIdx.x == 0) {
        ctl[32 * 9] = (unsigned) it;  // plain store: through the L1 into the shared L2
      }
    } else if (threadIdx.x == 0) {
      int spins = 0;
      if (BARRIER == 0) {
        while (__hip_atomic_load(&ctl[32 * (10 + rank)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned) it) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > spin_limit) {
            s_abort = 1;
            break;
          }
        }
      } else {
        while (load_l2(&ctl[32 * 9]) < (unsigned) it) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > spin_limit) {
            s_abort = 1;
            break;
          }
        }
      }
    }
    __syncthreads();
    if (s_abort) {
      if (threadIdx.x == 0)
        atomicAdd(&ctl[32 * 301], 1000000u);
      return;
    }
    // hand-off, second half: the neighbour's value of THIS round
    if (threadIdx.x == 0 && HANDOFF != 0) {
      const unsigned nb = (rank + 1) % P;
      const unsigned v = HANDOFF == 1 ? __hip_atomic_load(&x[nb * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : load_l2(&x[nb * 32]);
      if (v < (unsigned) it)  // (the neighbour may already have stored the next round's value)
        atomicAdd(&ctl[32 * 301], 1u);
    }
  }
}

template <int BARRIER, int HANDOFF>
static int run(const char* name, unsigned* ctl, unsigned* x, int n_rounds) {
  CHECK(hipMemset(ctl, 0, 32 * 400 * 4));
  CHECK(hipMemset(x, 0, 32 * 512 * 4));
  int spin = 1 << 20;
  void* args[] = {&ctl, &x, &n_rounds, &spin};
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  CHECK(hipEventRecord(a, 0));
  CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&rounds<BARRIER, HANDOFF>), dim3(256), dim3(1024), args, 0, 0));
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  unsigned h[3];
  CHECK(hipMemcpy(&h[0], ctl + 32 * 300, 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(&h[1], ctl + 32 * 301, 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(&h[2], ctl + 32 * 302, 4, hipMemcpyDeviceToHost));
  printf("%-28s %7.3f us per round   participants %u  errors %u\n", name, ms * 1e3 / n_rounds, h[2], h[1]);
  return 0;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned *ctl, *x;
  CHECK(hipMalloc(&ctl, 32 * 400 * 4));
  CHECK(hipMalloc(&x, 32 * 512 * 4));
  for (int rep = 0; rep < 2; ++rep) {
    if (run<0, 0>("grid barrier", ctl, x, n)) return 1;
    if (run<0, 1>("grid barrier + agent value", ctl, x, n)) return 1;
    if (run<1, 0>("xcd barrier", ctl, x, n)) return 1;
    if (run<1, 1>("xcd barrier + agent value", ctl, x, n)) return 1;
    if (run<1, 2>("xcd barrier + L2 value", ctl, x, n)) return 1;
  }
  return 0;
}
