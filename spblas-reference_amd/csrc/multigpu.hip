// Multi-GPU plumbing of the row-sharded SpMV ("second stage: fuse the gather into
// the kernel epilogue"): inter-process mapping of the y buffers and a device-side step barrier.
//
// The reference is single-device; there is nothing to mirror here.  One process per GPU.  Every rank
// allocates its copy of the full y with spblas_gfx950_ipc_alloc, exports it (hipIpcGetMemHandle), and
// maps the other ranks' copies (hipIpcOpenMemHandle, which also enables peer access).  The reduce /
// combine kernels of spmv_sliced.hip then store each finished row into all P copies
// (spblas_gfx950_spmv_reduce_rows_bcast): on xGMI that is the same traffic as a direct all-gather -- each
// shard crosses each link once -- without a collective launch and overlapped with the computation.
// A step ends with spblas_gfx950_step_signal (tell every rank "my stores of step k are issued"; it runs
// after the producing kernels on the same stream, so they are complete and visible system-wide) and
// spblas_gfx950_step_wait (spin, on the device, until all P ranks have signalled step k).
#include <cstring>

#include "common.hpp"

namespace spb {

__global__ void step_signal_kernel(long long* const* __restrict__ flag_peers, int n_peers, int rank, long long step) {
  const int p = threadIdx.x;
  if (p < n_peers)
    __hip_atomic_store(flag_peers[p] + rank, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// chunked step: slot (rank * chunks + c) of every rank's flag array <- step, after the peer stores of chunk c (same stream)
__global__ void chunk_signal_kernel(long long* const* __restrict__ flag_peers, int n_peers, int slot, int n_slots, long long step,
                                    long long delay_ticks) {
  if (delay_ticks > 0 && threadIdx.x == 0) {  // test hook: a deliberately late chunk
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < delay_ticks)
      __builtin_amdgcn_s_sleep(32);
  }
  __syncthreads();
  const int p = threadIdx.x;
  if (p < n_peers)
    for (int k = 0; k < n_slots; ++k)
      __hip_atomic_store(flag_peers[p] + slot + k, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// status[0] = 1 when the wait timed out (a peer died): the host checks it after synchronising.
__global__ void step_wait_kernel(const long long* __restrict__ flags, int n_peers, long long step,
                                 long long timeout_ticks, int* __restrict__ status) {
  const int p = threadIdx.x;
  if (p >= n_peers)
    return;
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(flags + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < step) {
    __builtin_amdgcn_s_sleep(8);
    if (wall_clock64() - t0 > timeout_ticks) {
      status[0] = 1;
      return;
    }
  }
}

int wall_clock_khz(spblas_gfx950_handle_s* h) {
  int rate_khz = 100000;  // wall_clock64 ticks at 100 MHz on gfx9
  (void) hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, h->device);
  return rate_khz > 0 ? rate_khz : 100000;
}

int launch_chunk_signal(spblas_gfx950_handle_s* h, hipStream_t s, void* const* flag_peers, int n_peers, int slot, int n_slots,
                        int64_t step, int64_t delay_us) {
  hipLaunchKernelGGL(chunk_signal_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<long long* const*>(flag_peers), n_peers, slot,
                     n_slots, (long long) step, (long long) (delay_us * wall_clock_khz(h) / 1000));
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int launch_step_wait(spblas_gfx950_handle_s* h, const void* flags, int n_peers, int64_t step, int64_t timeout_ms,
                     int* status_dev) {
  int rate_khz = 100000;  // wall_clock64 ticks at 100 MHz on gfx9
  (void) hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, h->device);
  if (rate_khz <= 0)
    rate_khz = 100000;
  hipLaunchKernelGGL(step_wait_kernel, dim3(1), dim3(64), 0, h->stream, static_cast<const long long*>(flags), n_peers,
                     (long long) step, (long long) timeout_ms * rate_khz, status_dev);
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

} // namespace spb

using namespace spb;

extern "C" {

int spblas_gfx950_ipc_alloc(size_t bytes, int uncached, void** ptr) {
  if (!ptr)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  *ptr = nullptr;
  // uncached (MTYPE UC) for memory that is POLLED while other devices write it (the step flags): a
  // cached line would never show the remote update.  Bulk data (y) stays ordinary device memory: it is
  // only read by kernels launched after the barrier, whose acquire makes the peers' stores visible.
  hipError_t e = uncached ? hipExtMallocWithFlags(ptr, bytes ? bytes : 1, hipDeviceMallocUncached)
                          : hipMalloc(ptr, bytes ? bytes : 1);
  if (e != hipSuccess && uncached) {
    (void) hipGetLastError();
    e = hipExtMallocWithFlags(ptr, bytes ? bytes : 1, hipDeviceMallocFinegrained);
  }
  if (e != hipSuccess) {
    g_last_hip_error = (int) e;
    return SPBLAS_GFX950_STATUS_ALLOC_FAILED;
  }
  SPB_HIP(hipMemset(*ptr, 0, bytes ? bytes : 1));
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_ipc_free(void* ptr) {
  if (ptr)
    SPB_HIP(hipFree(ptr));
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_ipc_export(void* ptr, unsigned char handle[64]) {
  if (!ptr || !handle)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is expected to be 64 bytes");
  hipIpcMemHandle_t h;
  SPB_HIP(hipIpcGetMemHandle(&h, ptr));
  std::memcpy(handle, &h, 64);
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_ipc_open(const unsigned char handle[64], void** ptr) {
  if (!handle || !ptr)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle, 64);
  *ptr = nullptr;
  SPB_HIP(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_ipc_close(void* ptr) {
  if (ptr)
    SPB_HIP(hipIpcCloseMemHandle(ptr));
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_step_signal(spblas_gfx950_handle_t handle, void* const* flag_peers, int n_peers, int rank,
                              int64_t step) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!flag_peers)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (n_peers < 1 || n_peers > 64 || rank < 0 || rank >= n_peers)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  hipLaunchKernelGGL(step_signal_kernel, dim3(1), dim3(64), 0, handle->stream,
                     reinterpret_cast<long long* const*>(flag_peers), n_peers, rank, (long long) step);
  SPB_HIP(hipGetLastError());
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_step_wait(spblas_gfx950_handle_t handle, const void* flags, int n_peers, int64_t step,
                            int64_t timeout_ms, int* status_dev) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!flags || !status_dev)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (n_peers < 1 || n_peers > 64)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  return launch_step_wait(handle, flags, n_peers, step, timeout_ms, status_dev);
}

int spblas_gfx950_wall_clock_khz(spblas_gfx950_handle_t handle, int* khz) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!khz)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  *khz = wall_clock_khz(handle);
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_bcast_wait_before(spblas_gfx950_handle_t handle, const void* flags, int n_peers, int64_t step,
                                    int64_t timeout_ms, int* status_dev) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!flags || !status_dev)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  if (n_peers < 1 || n_peers > 64)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  handle->bcast_wait.flags = flags;
  handle->bcast_wait.n_peers = n_peers;
  handle->bcast_wait.step = step;
  handle->bcast_wait.timeout_ms = timeout_ms;
  handle->bcast_wait.status_dev = status_dev;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

} // extern "C"

// Loads this file's code object (the runtime loads a code object at the first use of one of its kernels: milliseconds
// that would otherwise fall on the caller's first inspect / compute call -- handle.hip: spblas_gfx950_create).
namespace spb {
void preload_multigpu() {
  hipFuncAttributes attr;
  (void) hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&step_signal_kernel));
  (void) hipGetLastError();
}
} // namespace spb
