// Shared host/device helpers of the gfx950 backend library (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "spblas_gfx950.h"

#define SPB_WAVE 64  // CDNA wavefront width

struct spblas_gfx950_handle_s {
  hipStream_t stream;
  int device;
  int num_cus;
  int64_t bin_row_align = 0;  // SPBLAS_GFX950_OPT_BIN_ROW_ALIGN
  int64_t max_ksplit = 0;     // SPBLAS_GFX950_OPT_MAX_KSPLIT (0 = no cap)
  int64_t value_snapshot = 0; // SPBLAS_GFX950_OPT_VALUE_SNAPSHOT: AUTO may pick a plan that copies A's values
  int64_t spgemm_keep_colind = 0;  // SPBLAS_GFX950_OPT_SPGEMM_KEEP_COLIND: same c_colind address = same contents
  int64_t store_flavour = 0;       // SPBLAS_GFX950_OPT_STORE_TRIAL: 0 plain product stores, 1 non-temporal, 2 timed trial
  int nt_choice[2] = {0, 0};       // the trial's decision per value size (fp32 / fp64): 0 unknown, 1 plain, 2 non-temporal
  // one-shot: the next reduce_rows_bcast waits (on the device) for this step barrier right before the kernel that
  // stores to the peers -- the combine kernel when the reduce is K-split, else the reduce itself (spblas_gfx950_bcast_wait_before)
  struct bcast_wait_t {
    const void* flags = nullptr;
    int n_peers = 0;
    int64_t step = 0, timeout_ms = 0;
    int* status_dev = nullptr;
  } bcast_wait;
  // one-shot: the next expand waits, slice by slice, for the chunks of the peers' previous-step y its x slice is made of
  // (spblas_gfx950_spmv_step_bcast_chunked; csrc/spmv_sliced.hip: pb_expand_kernel)
  struct chunk_wait_t {
    const long long* flags = nullptr;       // this rank's flag array [n_ranks * chunks]
    const long long* chunk_rows = nullptr;  // device [n_ranks * (chunks + 1)]: global first row of every chunk
    int n_ranks = 0, chunks = 0, rank = 0;
    long long step = 0, timeout_ticks = 0;
    int* status_dev = nullptr;              // [0] timed out, [1] longest wait of a workgroup in wall-clock ticks
    int max_wgs = 0;                        // > 0: at most this many expand workgroups (several ranks sharing one device)
  } chunk_wait;
  // one-shot: the next K-split combine of a reduce_rows_bcast publishes a flag per chunk of rows itself -- the last
  // workgroup of a chunk, after a system-scope fence -- instead of ending at a kernel boundary (pb_combine_publish_kernel)
  struct chunk_pub_t {
    void* const* flag_peers = nullptr;
    int n_peers = 0, slot0 = 0, chunks = 0;
    long long step = 0, rows_per_chunk = 0, delay_ticks = 0;
  } chunk_pub;
  int* chunk_done = nullptr;  // [64] arrival counters of the publishing combine (zero between launches)
  // second stream + fork/join events of the striped fused step (created on first use)
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // grow-only scratch for stand-alone operations (transpose): allocated with plain hipMalloc, reused
  // by stream order, released with the handle
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  hipStream_t scratch_stream = nullptr;
  // pinned, device-visible host buffer for small read-backs (spb::readback_*), and the copies staged in it
  void* pinned = nullptr;
  size_t pinned_bytes = 0, pinned_used = 0;
  struct pending_copy {
    void* dst;
    size_t off, bytes;
  };
  pending_copy pending[8];
  int n_pending = 0;
};

namespace spb {

extern thread_local int g_last_hip_error;

inline int hip_fail(hipError_t e) {
  g_last_hip_error = (int) e;
  return e == hipErrorOutOfMemory ? SPBLAS_GFX950_STATUS_ALLOC_FAILED : SPBLAS_GFX950_STATUS_HIP_ERROR;
}

#define SPB_HIP(expr)                                                                              \
  do {                                                                                             \
    hipError_t spb_e_ = (expr);                                                                    \
    if (spb_e_ != hipSuccess)                                                                      \
      return ::spb::hip_fail(spb_e_);                                                              \
  } while (0)

// True while work submitted to `s` is being recorded into a graph (hipStreamBeginCapture, torch.cuda.graph).  Execute
// paths with a plan are capturable -- launches and memsets only -- as long as they have nothing to allocate: a buffer
// allocated or freed inside a capture would belong to the graph, not to the plan.
inline bool stream_capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) {
    (void) hipGetLastError();
    return false;
  }
  return st != hipStreamCaptureStatusNone;
}

// Stream-ordered device allocation, the same primitive the reference's
// hip_allocator uses (vendor/rocsparse/hip_allocator.hpp:34-47).  Refused on a stream that is being captured.
inline int dev_alloc(void** p, size_t bytes, hipStream_t s) {
  *p = nullptr;
  if (bytes == 0)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  if (s && stream_capturing(s))
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  // On the legacy null stream plain hipMalloc/hipFree are used: pool memory released there by
  // stream order was seen to be recycled while still in use when the application mixes in ordinary
  // hipMalloc/hipFree traffic (tests/cpp/device_tests.cpp exposed it).  Allocation only happens at
  // inspect time, so the implied synchronisation is acceptable.
  static const bool no_pool = std::getenv("SPBLAS_GFX950_NO_POOL") != nullptr;  // experiment knob: plain hipMalloc
  hipError_t e = (s && !no_pool) ? hipMallocAsync(p, bytes, s) : hipErrorNotSupported;
  if (e != hipSuccess) {
    (void) hipGetLastError();
    e = hipMalloc(p, bytes);
  }
  if (e != hipSuccess) {
    g_last_hip_error = (int) e;
    return SPBLAS_GFX950_STATUS_ALLOC_FAILED;
  }
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

inline void dev_free(void* p, hipStream_t s) {
  if (!p)
    return;
  if (!s || hipFreeAsync(p, s) != hipSuccess) {
    (void) hipGetLastError();
    (void) hipFree(p);
  }
}

// `bytes` of handle-owned scratch, valid until the next handle_scratch call on this handle.  Work that
// used the previous contents is ordered before the new user by the stream; when the stream changed or
// the buffer has to grow, the old stream is drained first.
inline int handle_scratch(spblas_gfx950_handle_s* h, size_t bytes, void** out) {
  if (h->scratch && h->scratch_stream != h->stream)
    (void) hipStreamSynchronize(h->scratch_stream);
  if (bytes > h->scratch_bytes) {
    if (h->scratch) {
      (void) hipStreamSynchronize(h->scratch_stream);
      (void) hipFree(h->scratch);
      h->scratch = nullptr;
      h->scratch_bytes = 0;
    }
    const hipError_t e = hipMalloc(&h->scratch, bytes);
    if (e != hipSuccess) {
      g_last_hip_error = (int) e;
      h->scratch = nullptr;
      return SPBLAS_GFX950_STATUS_ALLOC_FAILED;
    }
    h->scratch_bytes = bytes;
  }
  h->scratch_stream = h->stream;
  *out = h->scratch;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

// Small device -> host read-backs WITHOUT the SDMA engine: a copy kernel writes into the handle's pinned host buffer
// (device-mapped), readback_flush drains the stream and hands the bytes out.  The first hipMemcpy of more than a few KB
// in a process sets up an SDMA queue (12-15 ms measured inside an inspect of 38 ms); counters and offset tables of an
// inspect are a few dozen KB.  Larger or unaligned requests take hipMemcpyAsync.
static __global__ __launch_bounds__(256) void spb_readback_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                                  size_t n) {
  for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256)
    dst[i] = src[i];
}
inline int readback_add(spblas_gfx950_handle_s* h, void* host_dst, const void* dev_src, size_t bytes) {
  constexpr size_t cap = (size_t) 1 << 20;
  if (bytes == 0)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  const bool fits = (bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(dev_src) & 3) == 0 && h->n_pending < 8 &&
                    h->pinned_used + bytes <= cap;
  if (fits && !h->pinned) {
    if (hipHostMalloc(&h->pinned, cap, hipHostMallocDefault) != hipSuccess) {
      (void) hipGetLastError();
      h->pinned = nullptr;
    } else {
      h->pinned_bytes = cap;
    }
  }
  if (!fits || !h->pinned) {
    const hipError_t e = hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, h->stream);
    return e == hipSuccess ? SPBLAS_GFX950_STATUS_SUCCESS : hip_fail(e);
  }
  uint32_t* dst = reinterpret_cast<uint32_t*>(static_cast<char*>(h->pinned) + h->pinned_used);
  const size_t n = bytes / 4;
  const unsigned grid = (unsigned) (n / 256 + 1 < 64 ? n / 256 + 1 : 64);
  hipLaunchKernelGGL(spb_readback_kernel, dim3(grid), dim3(256), 0, h->stream, static_cast<const uint32_t*>(dev_src), dst, n);
  h->pending[h->n_pending++] = {host_dst, h->pinned_used, bytes};
  h->pinned_used += (bytes + 63) & ~(size_t) 63;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}
// The mirror image for small host -> device uploads (work lists): the bytes are staged in the pinned buffer and a
// kernel copies them to their place.  The staging space is recycled by readback_flush (which drains the stream).
inline int upload_add(spblas_gfx950_handle_s* h, void* dev_dst, const void* host_src, size_t bytes) {
  constexpr size_t cap = (size_t) 1 << 20;
  if (bytes == 0)
    return SPBLAS_GFX950_STATUS_SUCCESS;
  const bool fits = (bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(dev_dst) & 3) == 0 && h->pinned_used + bytes <= cap;
  if (fits && !h->pinned) {
    if (hipHostMalloc(&h->pinned, cap, hipHostMallocDefault) != hipSuccess) {
      (void) hipGetLastError();
      h->pinned = nullptr;
    } else {
      h->pinned_bytes = cap;
    }
  }
  if (!fits || !h->pinned) {
    const hipError_t e = hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, h->stream);
    return e == hipSuccess ? SPBLAS_GFX950_STATUS_SUCCESS : hip_fail(e);
  }
  char* stage = static_cast<char*>(h->pinned) + h->pinned_used;
  std::memcpy(stage, host_src, bytes);
  const size_t n = bytes / 4;
  const unsigned grid = (unsigned) (n / 256 + 1 < 64 ? n / 256 + 1 : 64);
  hipLaunchKernelGGL(spb_readback_kernel, dim3(grid), dim3(256), 0, h->stream, reinterpret_cast<const uint32_t*>(stage),
                     static_cast<uint32_t*>(dev_dst), n);
  h->pinned_used += (bytes + 63) & ~(size_t) 63;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}
inline int readback_flush(spblas_gfx950_handle_s* h) {
  const hipError_t e = hipStreamSynchronize(h->stream);
  for (int i = 0; i < h->n_pending; ++i)
    if (e == hipSuccess)
      std::memcpy(h->pending[i].dst, static_cast<char*>(h->pinned) + h->pending[i].off, h->pending[i].bytes);
  h->n_pending = 0;
  h->pinned_used = 0;
  return e == hipSuccess ? SPBLAS_GFX950_STATUS_SUCCESS : hip_fail(e);
}

// Every function that queues read-backs holds one of these: whatever path it leaves by, nothing stays queued that
// points at its stack frame or at memory it frees (readback_flush hands bytes out to the queued destinations; an error
// return between readback_add and readback_flush used to leave them behind for the NEXT flush of the handle).
struct readback_scope {
  spblas_gfx950_handle_s* h;
  explicit readback_scope(spblas_gfx950_handle_s* handle) : h(handle) {}
  readback_scope(const readback_scope&) = delete;
  readback_scope& operator=(const readback_scope&) = delete;
  ~readback_scope() {
    if (h && (h->n_pending != 0 || h->pinned_used != 0)) {
      (void) hipStreamSynchronize(h->stream);  // the copy kernels may still be using the staging buffer
      h->n_pending = 0;
      h->pinned_used = 0;
    }
  }
};

// launches the device-side step barrier of the fused multi-GPU step (multigpu.hip)
int launch_step_wait(spblas_gfx950_handle_s* h, const void* flags, int n_peers, int64_t step, int64_t timeout_ms,
                     int* status_dev);
int launch_chunk_signal(spblas_gfx950_handle_s* h, hipStream_t s, void* const* flag_peers, int n_peers, int slot, int n_slots,
                        int64_t step, int64_t delay_us);
int wall_clock_khz(spblas_gfx950_handle_s* h);

template <typename T>
struct scalar_of;
template <>
struct scalar_of<float> {
  static constexpr int id = SPBLAS_GFX950_F32;
};
template <>
struct scalar_of<double> {
  static constexpr int id = SPBLAS_GFX950_F64;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

// Sum over the `width` (power of two <= 64) consecutive lanes a lane belongs to.
template <typename T>
__device__ __forceinline__ T group_sum(T v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1)
    v += __shfl_xor(v, o, SPB_WAVE);
  return v;
}

template <int WIDTH, typename T>
__device__ __forceinline__ T group_sum_c(T v) {
#pragma unroll
  for (int o = WIDTH >> 1; o > 0; o >>= 1)
    v += __shfl_xor(v, o, SPB_WAVE);
  return v;
}

// Streaming (read-once) loads: keep A's arrays from evicting x / B out of L2.
template <typename V>
__device__ __forceinline__ V stream_load(const V* p) {
  return __builtin_nontemporal_load(p);
}

inline int64_t cdiv(int64_t a, int64_t b) {
  return (a + b - 1) / b;
}

} // namespace spb
