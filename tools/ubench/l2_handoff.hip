// What would an XCD-LOCAL producer -> consumer hand-off of the products cost?  (the P round trip is 8 of
// the 15.25 B per entry the cfg2 launch pair moves through the fabric; a fused kernel whose expanding and reducing
// workgroups sit on ONE XCD could hand the products over inside that XCD's L2.  This is NOT the placement-independent
// protocol of the programming guide (Guideline 16): it only works because both ends share an L2, so roles are taken
// from the XCC_ID the workgroup really runs on.  A measurement, not a product path.)
//
// 256 workgroups, one per CU (160 KiB of LDS requested).  Each reads XCC_ID, takes a seat in its XCD's team (atomic
// counter) and pairs up: even seat = producer, odd seat = consumer.  A pair shares a ring of R slots of 4 KiB in ordinary
// device memory: the producer writes a slot with plain 16-byte stores, drains (vmcnt(0)), publishes the slot's sequence
// number; the consumer polls the number with an L1-bypassing load, reads the slot with L1-bypassing 16-byte loads (sc1),
// checks the contents and returns a credit.  mode 1 adds the streams the real kernels would carry: the producer
// non-temporally loads 6 B per product (A'), the consumer 1.25 B (row codes).  Every spin is bounded.
// Output: products/s, errors; run under rocprofv3 --pmc TCC_EA0_* for the fabric bytes (400 MB of products pass by).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned gu32 __attribute__((address_space(1)));

static constexpr int R = 32, CHUNK = 1024;  // slots per ring, floats per slot (4 KiB)
static constexpr unsigned SPIN_MAX = 1u << 22;

__device__ __forceinline__ f32x4 load_l2(const f32x4* p) {  // bypass this CU's L1: the line lives in the shared L2
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned poll(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1 load: L2, not L1
}

__global__ __launch_bounds__(256) void handoff_kernel(float* __restrict__ rings, unsigned* __restrict__ seq,
                                                      unsigned* __restrict__ ack, unsigned* __restrict__ team,
                                                      unsigned* __restrict__ fail, unsigned long long* __restrict__ errors,
                                                      int chunks, int mode, const f32x4* __restrict__ a_stream,
                                                      const unsigned* __restrict__ c_stream, float* __restrict__ sink) {
  extern __shared__ float lds[];
  __shared__ unsigned s_seat;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 0xf;
  if (threadIdx.x == 0)
    s_seat = atomicAdd(&team[xcc], 1u);
  __syncthreads();
  const unsigned seat = s_seat;
  // everybody seated?  (a pair needs both ends; all 256 workgroups are resident: one per CU)
  if (threadIdx.x == 0) {
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
  const unsigned chan = xcc * 32 + seat / 2;
  float* ring = rings + (size_t) chan * R * CHUNK;
  unsigned* cseq = seq + chan * R;
  unsigned* cack = ack + chan * 16;  // a line per channel
  const bool producer = (seat & 1u) == 0;
  float acc = 0.f;
  unsigned long long bad = 0;
  for (int it = 0; it < chunks; ++it) {
    const int slot = it % R;
    f32x4* sp = reinterpret_cast<f32x4*>(ring + (size_t) slot * CHUNK) + threadIdx.x;
    if (producer) {
      if (threadIdx.x == 0) {  // credit: the consumer has freed this slot
        unsigned spins = 0;
        while ((int) (it - (int) poll(cack)) >= R && ++spins < SPIN_MAX)
          __builtin_amdgcn_s_sleep(2);
        if (spins >= SPIN_MAX)
          atomicExch(fail, 2u);
      }
      __syncthreads();
      float base = (float) (it & 1023);
      if (mode == 1) {  // the A' stream of these 1 024 products: 6 KiB = 1.5 x 16 B per thread
        const size_t o = ((size_t) chan * chunks + it) * 384;
        const f32x4 a = __builtin_nontemporal_load(a_stream + o + threadIdx.x);
        base += a.x * 0.f;
        if (threadIdx.x < 128) {
          const f32x4 b = __builtin_nontemporal_load(a_stream + o + 256 + threadIdx.x);
          base += b.x * 0.f;
        }
      }
      f32x4 v = {base, base + 1.f, (float) threadIdx.x, (float) chan};
      *sp = v;  // plain store: through the L1 into the L2 we share with the consumer
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0)
        __hip_atomic_store(&cseq[slot], (unsigned) it + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (poll(&cseq[slot]) != (unsigned) it + 1u && ++spins < SPIN_MAX)
          __builtin_amdgcn_s_sleep(2);
        if (spins >= SPIN_MAX)
          atomicExch(fail, 3u);
      }
      __syncthreads();
      if (mode == 1 && threadIdx.x < 80) {  // the row codes of these products: 1.25 KiB
        const unsigned c = __builtin_nontemporal_load(c_stream + ((size_t) chan * chunks + it) * 320 / 4 + threadIdx.x);
        acc += (float) c * 0.f;
      }
      const f32x4 v = load_l2(sp);
      const float base = (float) (it & 1023);
      bad += (v.x != base) + (v.y != base + 1.f) + (v.z != (float) threadIdx.x) + (v.w != (float) chan);
      acc += v.x + v.y;
      lds[threadIdx.x] += v.z;  // (the reduce's accumulators live here)
      __syncthreads();
      if (threadIdx.x == 0)
        __hip_atomic_store(cack, (unsigned) it + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if ((it & 63) == 63 && poll(fail))
      break;
  }
  if (bad)
    atomicAdd(errors, bad);
  if (acc == 12345.678f)
    sink[0] = acc + lds[0];
}

int main(int argc, char** argv) {
  const int chunks = argc > 1 ? atoi(argv[1]) : 763;  // x 128 pairs x 1 024 products ~ 1e8
  float *rings, *sink;
  unsigned *seq, *ack, *team, *fail;
  unsigned long long* errors;
  f32x4* a_stream;
  unsigned* c_stream;
  const size_t nchan = 8 * 32;
  CHECK(hipMalloc(&rings, nchan * R * CHUNK * 4));
  CHECK(hipMalloc(&seq, nchan * R * 4));
  CHECK(hipMalloc(&ack, nchan * 64));
  CHECK(hipMalloc(&team, 64));
  CHECK(hipMalloc(&fail, 4));
  CHECK(hipMalloc(&errors, 8));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMalloc(&a_stream, nchan * (size_t) chunks * 384 * 16 + 65536));
  CHECK(hipMalloc(&c_stream, nchan * (size_t) chunks * 320 + 65536));
  CHECK(hipMemset(a_stream, 0, nchan * (size_t) chunks * 384 * 16 + 65536));
  CHECK(hipMemset(c_stream, 0, nchan * (size_t) chunks * 320 + 65536));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(handoff_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipMemset(seq, 0, nchan * R * 4));
      CHECK(hipMemset(ack, 0, nchan * 64));
      CHECK(hipMemset(team, 0, 64));
      CHECK(hipMemset(fail, 0, 4));
      CHECK(hipMemset(errors, 0, 8));
      hipEventRecord(e0);
      hipLaunchKernelGGL(handoff_kernel, dim3(256), dim3(256), 150 * 1024, 0, rings, seq, ack, team, fail, errors, chunks, mode,
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
      const double products = (double) pairs * chunks * CHUNK;
      printf("mode %d (%s): %8.3f ms  %6.1f G products/s over %u pairs (teams %u %u %u %u %u %u %u %u)  fail %u  wrong words %llu\n", mode,
             mode ? "with 6 B + 1.25 B per product of streamed loads" : "hand-off only", ms, products / ms / 1e6, pairs, h_team[0],
             h_team[1], h_team[2], h_team[3], h_team[4], h_team[5], h_team[6], h_team[7], h_fail, h_err);
    }
  return 0;
}
