// Round 4, the bounded experiment the round-3 review asked for (its item 6b): the XCD-local producer -> consumer hand-off of
// tools/ubench/l2_handoff.hip, PIPELINED.  The first version handed one 4 KiB slot over per handshake with a drain per slot
// (0.5 us each: 1e8 products in 0.38 ms, 1.0 ms with the streams) and could not say what a real kernel would reach.  Here a
// channel is a pair of WAVEFRONTS (producer wave w of the even-seated workgroup, consumer wave w of its odd-seated partner
// on the same XCD), a slot is 1 KiB (64 lanes x 16 B), the producer keeps DEPTH slots in flight (its publish of slot i - DEPTH
// follows an s_waitcnt vmcnt(..) that only waits for that slot's stores) and the consumer polls once per BATCH slots, issues
// the BATCH L1-bypassing loads together and returns one credit per batch.  mode 1 adds the streams of the real kernels
// (producer: 6 B per product of A', non-temporal; consumer: 1.25 B per product of row codes) and an LDS gather / an LDS
// accumulate per product, so that the time is comparable with the cfg2 launch pair (expand 195 us + reduce 120 us).
// Roles come from XCC_ID: a measurement, not a product path (cdna_hip_programming.md, Guideline 16).  Every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifndef H2_THREADS
#define H2_THREADS 256
#endif
#ifndef H2_R
#define H2_R 32
#endif
#ifndef H2_DEPTH
#define H2_DEPTH 4
#endif
#ifndef H2_BATCH
#define H2_BATCH 8
#endif
static constexpr int WAVES = H2_THREADS / 64;
static constexpr int R = H2_R;    // slots per ring (1 KiB each); rings per XCD: 32 pairs x WAVES x R KiB (keep it well under the 4 MiB L2)
static constexpr int DEPTH = H2_DEPTH;   // producer: slots whose stores may still be in flight
static constexpr int BATCH = H2_BATCH;   // consumer: slots per poll / credit
static constexpr unsigned SPIN_MAX = 1u << 20;

__device__ __forceinline__ unsigned poll(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // L1-bypassing: the line lives in the shared L2
}
__device__ __forceinline__ f32x4 load_l2(const f32x4* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

__global__ __launch_bounds__(H2_THREADS) void handoff2_kernel(float* __restrict__ rings, unsigned* __restrict__ seq,
                                                       unsigned* __restrict__ ack, unsigned* __restrict__ team,
                                                       unsigned* __restrict__ fail, unsigned long long* __restrict__ errors,
                                                       int slots, int mode, const f32x4* __restrict__ a_stream,
                                                       const unsigned* __restrict__ c_stream, float* __restrict__ sink) {
  extern __shared__ float lds[];  // 150 KiB: one workgroup per CU; mode 1 gathers from / accumulates into it
  __shared__ unsigned s_seat;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 0xf;
  if (threadIdx.x == 0)
    s_seat = atomicAdd(&team[xcc], 1u);
  for (int i = threadIdx.x; i < 150 * 256; i += H2_THREADS)
    lds[i] = 1.0f;
  __syncthreads();
  const unsigned seat = s_seat;
  if (threadIdx.x == 0) {  // everybody seated?  (all 256 workgroups are resident: one per CU)
    atomicAdd(&team[8], 1u);
    unsigned spins = 0;
    while (poll(&team[8]) < gridDim.x && ++spins < SPIN_MAX)
      __builtin_amdgcn_s_sleep(8);
    if (spins >= SPIN_MAX)
      atomicExch(fail, 1u);
  }
  __syncthreads();
  if (poll(fail))
    return;
  const unsigned nteam = poll(&team[xcc]);
  if ((nteam & 1u) && seat == nteam - 1)
    return;  // odd team: the last seat has no partner
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned chan = (xcc * 32 + seat / 2) * WAVES + wave;
  f32x4* ring = reinterpret_cast<f32x4*>(rings) + (size_t) chan * R * 64;
  unsigned* cseq = seq + chan * 16;  // a line per channel: [0] = slots published
  unsigned* cack = ack + chan * 16;  // [0] = slots consumed
  const bool producer = (seat & 1u) == 0;
  float acc = 0.f;
  unsigned long long bad = 0;
  unsigned credits = 0;  // producer: slots the consumer is known to have freed
  if (producer) {
    // mode 1: the A' loads of slot it + 4 are issued while slot it is worked on (four register sets, the loop unrolled by
    // four: no register copies, so the compiler's own waits count operations)
    f32x4 pa[4] = {};
    unsigned pc[4] = {};
    auto load_a = [&](int it, f32x4& a, unsigned& c) {
      const size_t o = ((size_t) chan * slots + it) * 96;
      a = __builtin_nontemporal_load(a_stream + o + lane);
      c = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(a_stream + o + 64) + lane * 2);
    };
    if (mode == 1)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < slots)
          load_a(k, pa[k], pc[k]);
    bool stop = false;
    for (int it0 = 0; it0 < slots && !stop; it0 += 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int it = it0 + k;
        if (it >= slots || stop)
          break;
        if ((unsigned) it >= credits + R) {  // no free slot known: poll the consumer's counter (wave-uniform)
          unsigned spins = 0, a = 0;
          do {
            a = poll(cack);
            if ((unsigned) it < a + R)
              break;
            __builtin_amdgcn_s_sleep(1);
          } while (++spins < SPIN_MAX);
          if (spins >= SPIN_MAX) {
            atomicExch(fail, 2u);
            stop = true;
            break;
          }
          credits = a;
        }
        float base = (float) (it & 1023);
        if (mode == 1) {  // A' of these 256 products: 1.5 KiB = 24 B per lane; the products come from an LDS gather
          const unsigned c = pc[k];
          const unsigned i0 = (c & 0xffffu) % 38400u, i1 = (c >> 16) % 38400u;
          base += pa[k].x * 0.f + (lds[i0] + lds[i1] + lds[(i0 + 7919u) % 38400u] + lds[(i1 + 104729u) % 38400u] - 4.0f);
        }
        const f32x4 v = {base, base + 1.f, (float) lane, (float) chan};
        ring[(size_t) (it % R) * 64 + lane] = v;  // plain store: through the L1 into the L2 we share with the consumer
        const bool more = mode == 1 && it + 4 < slots;
        if (more)
          load_a(it + 4, pa[k], pc[k]);
        if (it >= DEPTH) {
          // the store of slot it - DEPTH is complete when at most the vector-memory operations issued after it are
          // outstanding (vmcnt counts in order): DEPTH stores, in mode 1 also two loads per slot (2 + 3 (DEPTH - 1) + 3)
          if (mode == 1 && more)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * DEPTH + 2) : "memory");
          else if (mode == 1)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
          if (lane == 0)
            __hip_atomic_store(cseq, (unsigned) (it - DEPTH + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if ((it0 & 255) == 252 && poll(fail))
        break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && !stop)
      __hip_atomic_store(cseq, (unsigned) slots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    unsigned published = 0;
    for (int it = 0; it < slots; it += BATCH) {
      const int nb = slots - it < BATCH ? slots - it : BATCH;
      if (published < (unsigned) (it + nb)) {
        unsigned spins = 0;
        do {
          published = poll(cseq);
          if (published >= (unsigned) (it + nb))
            break;
          __builtin_amdgcn_s_sleep(1);
        } while (++spins < SPIN_MAX);
        if (spins >= SPIN_MAX) {
          atomicExch(fail, 3u);
          break;
        }
      }
      f32x4 v[BATCH];
      unsigned code[BATCH];
#pragma unroll
      for (int b = 0; b < BATCH; ++b) {
        v[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        code[b] = 0;
        if (b < nb) {
          v[b] = load_l2(ring + (size_t) ((it + b) % R) * 64 + lane);
          if (mode == 1)  // row codes of these 256 products: 320 B = 5 B per lane (one dword here, a fifth lane-group one more)
            code[b] = __builtin_nontemporal_load(c_stream + ((size_t) chan * slots + it + b) * 80 + lane) +
                      (lane < 16 ? __builtin_nontemporal_load(c_stream + ((size_t) chan * slots + it + b) * 80 + 64 + lane) : 0u);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int b = 0; b < BATCH; ++b)
        if (b < nb) {
          const float base = (float) ((it + b) & 1023);
          bad += (v[b].x != base) + (v[b].y != base + 1.f) + (v[b].z != (float) lane) + (v[b].w != (float) chan);
          if (mode == 1) {  // four accumulations per lane into the reduce's LDS accumulators
            const unsigned r0 = (code[b] + lane * 4u) % 38400u;
            lds[r0] += v[b].x;
            lds[(r0 + 1u) % 38400u] += v[b].y;
            lds[(r0 + 2u) % 38400u] += v[b].z;
            lds[(r0 + 3u) % 38400u] += v[b].w;
          }
          acc += v[b].x + v[b].y;
        }
      if (lane == 0)
        __hip_atomic_store(cack, (unsigned) (it + nb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((it & 255) == 0 && poll(fail))
        break;
    }
  }
  if (bad)
    atomicAdd(errors, bad);
  if (acc == 12345.678f)
    sink[0] = acc + lds[threadIdx.x];
}

int main(int argc, char** argv) {
  const int slots = argc > 1 ? atoi(argv[1]) : 763 * 4 / WAVES;  // x 128 pairs x WAVES channels x 256 products ~ 1e8
  float *rings, *sink;
  unsigned *seq, *ack, *team, *fail;
  unsigned long long* errors;
  f32x4* a_stream;
  unsigned* c_stream;
  const size_t nchan = 8 * 32 * WAVES;
  CHECK(hipMalloc(&rings, nchan * R * 1024));
  CHECK(hipMalloc(&seq, nchan * 64));
  CHECK(hipMalloc(&ack, nchan * 64));
  CHECK(hipMalloc(&team, 64));
  CHECK(hipMalloc(&fail, 4));
  CHECK(hipMalloc(&errors, 8));
  CHECK(hipMalloc(&sink, 64));
  const size_t a_bytes = nchan * (size_t) slots * 96 * 16 + 65536, c_bytes = nchan * (size_t) slots * 320 + 65536;
  CHECK(hipMalloc(&a_stream, a_bytes));
  CHECK(hipMalloc(&c_stream, c_bytes));
  CHECK(hipMemset(a_stream, 0x11, a_bytes));
  CHECK(hipMemset(c_stream, 0x01, c_bytes));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(handoff2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipMemset(seq, 0, nchan * 64));
      CHECK(hipMemset(ack, 0, nchan * 64));
      CHECK(hipMemset(team, 0, 64));
      CHECK(hipMemset(fail, 0, 4));
      CHECK(hipMemset(errors, 0, 8));
      hipEventRecord(e0);
      hipLaunchKernelGGL(handoff2_kernel, dim3(256), dim3(H2_THREADS), 150 * 1024, 0, rings, seq, ack, team, fail, errors, slots, mode,
                         a_stream, c_stream, sink);
      hipEventRecord(e1);
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned h_team[16], h_fail = 0;
      unsigned long long h_err = 0;
      CHECK(hipMemcpy(h_team, team, 64, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(&h_fail, fail, 4, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(&h_err, errors, 8, hipMemcpyDeviceToHost));
      unsigned pairs = 0;
      for (int x = 0; x < 8; ++x)
        pairs += h_team[x] / 2;
      const double products = (double) pairs * WAVES * slots * 256;
      printf("mode %d (%s): %8.3f ms  %6.1f G products/s, %.3g products over %u pairs x %d waves (teams %u %u %u %u %u %u %u %u)  fail %u  wrong words %llu\n",
             mode, mode ? "6 B + 1.25 B per product streamed, LDS gather + accumulate" : "hand-off only", ms, products / ms / 1e6,
             products, pairs, WAVES, h_team[0], h_team[1], h_team[2], h_team[3], h_team[4], h_team[5], h_team[6], h_team[7], h_fail, h_err);
    }
  return 0;
}
