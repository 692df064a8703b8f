// How fast does HBM move a stream that is contiguous only in PIECES of L entries?
// (Decides the layout of the product array P of the sliced SpMV, DESIGN.md 4.3: today the expand writes P
//  linearly and the reduce reads it in ~100-entry runs; the alternative is the transposed piece order, where
//  the expand writes ~100-entry runs and the reduce reads linearly.)
//
//   lin      read 6 B + write 4 B per entry, both linear                       (the expand as it is)
//   wr_T     same reads, every piece of L products stored at the transposed position of an S x NB piece grid
//   wr_hash  same, pieces at pseudo-random positions
//   rd_lin   read 4 B + 2 B per entry linearly, nothing written                (the ideal reduce stream)
//   rd_T     read the same 6 B per entry in pieces taken in transposed order    (what the reduce does today)
// Every lane moves 4 consecutive entries (16-byte product access, 8-byte index access); pieces are multiples
// of 4 entries.  Output: GB/s of bytes moved (algorithmic bytes of the access, not counting over-fetch).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// position (in entries) of entry group q (4 entries) under the piece permutation
__device__ __forceinline__ long piece_pos(long q4, int L4, int S, int NB, int mode) {
  if (mode == 0)
    return q4;
  const long piece = q4 / L4, off = q4 - piece * L4;
  long dst;
  if (mode == 1) {  // transpose of the S x NB grid: piece (s, b) -> (b, s)
    const long s = piece / NB, b = piece - s * NB;
    dst = b * S + s;
  } else {  // bijective hash over S*NB pieces (odd multiplier modulo a power of two would need 2^k pieces:
            // use a multiplicative permutation modulo the prime-free count via 64-bit arithmetic)
    const long P = (long) S * NB;
    dst = (piece * 40503L + 12345L) % P;  // 40503 coprime with P when P has no factor 3, 23, 587
  }
  return dst * L4 + off;
}

template <int MODE_W, int XCD = 0>
__global__ __launch_bounds__(1024) void expand_like(const f32x4* __restrict__ val, const u16x4* __restrict__ col,
                                                    f32x4* __restrict__ P, long n4, int L4, int S, int NB) {
  const long stride = (long) gridDim.x * 1024;
  const long per = (n4 + gridDim.x - 1) / gridDim.x;  // contiguous share per workgroup, like the real expand
  // XCD = 1 (r03): workgroup i runs on XCD i % 8; give every XCD a CONTIGUOUS range of shares, so that the pieces of
  // neighbouring shares -- adjacent in memory under wr_T -- are written through ONE L2 at about the same time and the
  // partial lines at their ends can merge there before they reach HBM
  const int bid = XCD ? (int) ((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int) blockIdx.x;
  const long lo = (long) bid * per, hi = lo + per < n4 ? lo + per : n4;
  (void) stride;
  for (long q = lo + threadIdx.x; q < hi; q += 2048) {
    const long q2 = q + 1024 < hi ? q + 1024 : q;
    const f32x4 va = __builtin_nontemporal_load(val + q), vb = __builtin_nontemporal_load(val + q2);
    const u16x4 ca = __builtin_nontemporal_load(col + q), cb = __builtin_nontemporal_load(col + q2);
    f32x4 pa, pb;
    pa.x = va.x * (float) ca.x; pa.y = va.y * (float) ca.y; pa.z = va.z * (float) ca.z; pa.w = va.w * (float) ca.w;
    pb.x = vb.x * (float) cb.x; pb.y = vb.y * (float) cb.y; pb.z = vb.z * (float) cb.z; pb.w = vb.w * (float) cb.w;
    P[piece_pos(q, L4, S, NB, MODE_W)] = pa;
    if (q2 != q)
      P[piece_pos(q2, L4, S, NB, MODE_W)] = pb;
  }
}

// one wave per bin: walks the bin's share.  MODE_R = 0: the bin's entries are contiguous (b*S*L .. ), 1: the
// bin's S pieces are L-entry runs spread with stride NB*L (slice-major storage)
template <int MODE_R>
__global__ __launch_bounds__(256) void reduce_like(const f32x4* __restrict__ P, const u16x4* __restrict__ row,
                                                   float* __restrict__ out, int L4, int S, int NB) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long b = (long) blockIdx.x * 4 + wave;
  if (b >= NB)
    return;
  float acc = 0.f;
  if (MODE_R == 0) {
    const long lo = b * S * (long) L4, hi = lo + (long) S * L4;
    for (long q = lo + lane; q < hi; q += 256) {
      f32x4 p[4];
      u16x4 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long qq = q + 64 * u < hi ? q + 64 * u : hi - 1;
        p[u] = __builtin_nontemporal_load(P + qq);
        r[u] = __builtin_nontemporal_load(row + qq);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        acc += p[u].x + p[u].y + p[u].z + p[u].w + (float) (r[u].x ^ r[u].y ^ r[u].z ^ r[u].w);
    }
  } else {
    // runs of L4 groups: lanes [0, L4) of each run active (L4 <= 64 handled per run, longer runs in steps)
    for (int s0 = 0; s0 < S; s0 += 8) {
      f32x4 p[8];
      u16x4 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = s0 + u < S ? s0 + u : S - 1;
        const long base = ((long) s * NB + b) * L4;
        const long qq = base + (lane < L4 ? lane : L4 - 1);
        p[u] = __builtin_nontemporal_load(P + qq);
        r[u] = __builtin_nontemporal_load(row + qq);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (lane < L4 && s0 + u < S)
          acc += p[u].x + p[u].y + p[u].z + p[u].w + (float) (r[u].x ^ r[u].y ^ r[u].z ^ r[u].w);
      for (int u = 0; u < 8 && s0 + u < S; ++u)
        for (int o = 64; o < L4; o += 64) {
          const long base = ((long) (s0 + u) * NB + b) * L4;
          if (o + lane < L4) {
            const f32x4 pp = __builtin_nontemporal_load(P + base + o + lane);
            const u16x4 rr = __builtin_nontemporal_load(row + base + o + lane);
            acc += pp.x + pp.y + pp.z + pp.w + (float) (rr.x ^ rr.y ^ rr.z ^ rr.w);
          }
        }
    }
  }
  if (acc == 12345.678f)
    out[0] = acc;
}

template <typename F>
static float time_ms(F f, int reps = 6) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  f();
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
  const int NB = 4096;
  f32x4 *val, *P;
  u16x4 *col;
  float* out;
  const long cap = 110000000L;
  CHECK(hipMalloc(&val, cap * 4));
  CHECK(hipMalloc(&P, cap * 4));
  CHECK(hipMalloc(&col, cap * 2));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(val, 0, cap * 4));
  CHECK(hipMemset(col, 0, cap * 2));
  CHECK(hipMemset(P, 0, cap * 4));
  printf("%6s %5s %9s | %8s %8s %8s %8s | %8s %8s   (GB/s; expand-like moves 10 B/entry, reduce-like 6 B/entry)\n", "L", "S",
         "entries", "lin", "wr_T", "wr_Txcd", "wr_hash", "rd_lin", "rd_T");
  // S chosen so that S * NB * L ~ 1e8 entries
  for (int L : {8, 16, 32, 64, 100, 128, 200, 204, 196, 256}) {
    const int L4 = L / 4;
    int S = (int) (100000000L / ((long) NB * L));
    if (S < 1)
      S = 1;
    while ((((long) S * NB) % 3 == 0) || (((long) S * NB) % 23 == 0) || (((long) S * NB) % 587 == 0))
      ++S;
    const long n = (long) S * NB * L, n4 = n / 4;
    if (n > cap) {
      printf("skip L=%d\n", L);
      continue;
    }
    const int G = 256;
    float t0 = time_ms([&] { hipLaunchKernelGGL(expand_like<0>, dim3(G), dim3(1024), 0, 0, val, col, P, n4, L4, S, NB); });
    float t1 = time_ms([&] { hipLaunchKernelGGL(expand_like<1>, dim3(G), dim3(1024), 0, 0, val, col, P, n4, L4, S, NB); });
    float t1x = time_ms([&] { hipLaunchKernelGGL((expand_like<1, 1>), dim3(G), dim3(1024), 0, 0, val, col, P, n4, L4, S, NB); });
    float t2 = time_ms([&] { hipLaunchKernelGGL(expand_like<2>, dim3(G), dim3(1024), 0, 0, val, col, P, n4, L4, S, NB); });
    float t3 = time_ms([&] { hipLaunchKernelGGL(reduce_like<0>, dim3(NB / 4), dim3(256), 0, 0, P, col, out, L4, S, NB); });
    float t4 = time_ms([&] { hipLaunchKernelGGL(reduce_like<1>, dim3(NB / 4), dim3(256), 0, 0, P, col, out, L4, S, NB); });
    const double gb10 = n * 10.0 / 1e6, gb6 = n * 6.0 / 1e6;
    printf("%6d %5d %9ld | %8.0f %8.0f %8.0f %8.0f | %8.0f %8.0f   us: %.0f %.0f %.0f %.0f | %.0f %.0f\n", L, S, n, gb10 / t0, gb10 / t1,
           gb10 / t1x, gb10 / t2, gb6 / t3, gb6 / t4, t0 * 1e3, t1 * 1e3, t1x * 1e3, t2 * 1e3, t3 * 1e3, t4 * 1e3);
  }
  CHECK(hipDeviceSynchronize());
  return 0;
}
