// Handle / status plumbing of the gfx950 backend C ABI (include/spblas_gfx950.h).
// Replaces rocsparse_create_handle / rocsparse_set_stream / rocsparse_destroy_handle
// (/root/reference/include/spblas/vendor/rocsparse/detail/abstract_operation_state.hpp:20-28,
//  vendor/rocsparse/multiply_spgemm.hpp:34-43).
#include "common.hpp"

#include <cstdlib>
#include <mutex>

#include <new>

namespace spb {
thread_local int g_last_hip_error = 0;
}

namespace spb {
void preload_spmv();
void preload_sliced();
void preload_hot();
void preload_spmm();
void preload_spgemm();
void preload_transpose();
void preload_sptrsv();
void preload_multigpu();
} // namespace spb

extern "C" {

int spblas_gfx950_version(void) {
  return 100;  // 0.1.0
}

const char* spblas_gfx950_status_string(int status) {
  switch (status) {
  case SPBLAS_GFX950_STATUS_SUCCESS: return "success";
  case SPBLAS_GFX950_STATUS_INVALID_HANDLE: return "invalid handle";
  case SPBLAS_GFX950_STATUS_INVALID_POINTER: return "invalid pointer";
  case SPBLAS_GFX950_STATUS_INVALID_SIZE: return "matrix dimensions are incompatible";
  case SPBLAS_GFX950_STATUS_INVALID_VALUE: return "invalid value";
  case SPBLAS_GFX950_STATUS_NOT_SUPPORTED: return "not supported";
  case SPBLAS_GFX950_STATUS_ALLOC_FAILED: return "device allocation failed";
  case SPBLAS_GFX950_STATUS_HIP_ERROR: return "HIP runtime error";
  case SPBLAS_GFX950_STATUS_INSUFFICIENT_SPACE: return "SpGEMM ran out of memory";
  case SPBLAS_GFX950_STATUS_PLAN_MISMATCH: return "plan does not match the matrix";
  default: return "unknown status";
  }
}

int spblas_gfx950_last_hip_error(void) {
  return spb::g_last_hip_error;
}

int spblas_gfx950_create(spblas_gfx950_handle_t* handle, void* stream) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  *handle = nullptr;
  int dev = 0;
  SPB_HIP(hipGetDevice(&dev));
  int cus = 0;
  SPB_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  auto* h = new (std::nothrow) spblas_gfx950_handle_s();
  if (!h)
    return SPBLAS_GFX950_STATUS_ALLOC_FAILED;
  h->stream = static_cast<hipStream_t>(stream);
  h->device = dev;
  h->num_cus = cus;
  *handle = h;
  // The first handle on each DEVICE loads the library's code objects there (SPBLAS_GFX950_PRELOAD=0: left to the first use,
  // as the runtime does by itself; code objects are loaded per device, so a process that drives several GPUs pays once per
  // GPU at its first handle there, not at its first inspect).  Measured on MI355X: see DESIGN.md section 2, "first call".
  static std::once_flag preload_once[64];
  static const bool preload_on = [] {
    const char* e = std::getenv("SPBLAS_GFX950_PRELOAD");
    return !(e && std::atoi(e) == 0);
  }();
  if (preload_on && dev >= 0 && dev < 64)
    std::call_once(preload_once[dev], [] {
      spb::preload_spmv();
      spb::preload_sliced();
      spb::preload_hot();
      spb::preload_spmm();
      spb::preload_spgemm();
      spb::preload_transpose();
      spb::preload_sptrsv();
      spb::preload_multigpu();
    });
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_destroy(spblas_gfx950_handle_t handle) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (handle->aux_stream) {
    (void) hipStreamSynchronize(handle->aux_stream);
    (void) hipEventDestroy(handle->ev_fork);
    (void) hipEventDestroy(handle->ev_join);
    (void) hipStreamDestroy(handle->aux_stream);
  }
  if (handle->scratch) {
    (void) hipStreamSynchronize(handle->scratch_stream);
    (void) hipFree(handle->scratch);
  }
  if (handle->pinned)
    (void) hipHostFree(handle->pinned);
  if (handle->chunk_done) {
    (void) hipDeviceSynchronize();
    (void) hipFree(handle->chunk_done);
  }
  delete handle;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_set_stream(spblas_gfx950_handle_t handle, void* stream) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  handle->stream = static_cast<hipStream_t>(stream);
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

int spblas_gfx950_set_option(spblas_gfx950_handle_t handle, int option, int64_t value) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (option == SPBLAS_GFX950_OPT_BIN_ROW_ALIGN) {
    if (value < 0)
      return SPBLAS_GFX950_STATUS_INVALID_VALUE;
    handle->bin_row_align = value;
    return SPBLAS_GFX950_STATUS_SUCCESS;
  }
  if (option == SPBLAS_GFX950_OPT_MAX_KSPLIT) {
    if (value < 0)
      return SPBLAS_GFX950_STATUS_INVALID_VALUE;
    handle->max_ksplit = value;
    return SPBLAS_GFX950_STATUS_SUCCESS;
  }
  if (option == SPBLAS_GFX950_OPT_VALUE_SNAPSHOT) {
    if (value < 0 || value > 2)
      return SPBLAS_GFX950_STATUS_INVALID_VALUE;
    handle->value_snapshot = value;
    return SPBLAS_GFX950_STATUS_SUCCESS;
  }
  if (option == SPBLAS_GFX950_OPT_SPGEMM_KEEP_COLIND) {
    if (value != 0 && value != 1)
      return SPBLAS_GFX950_STATUS_INVALID_VALUE;
    handle->spgemm_keep_colind = value;
    return SPBLAS_GFX950_STATUS_SUCCESS;
  }
  if (option == SPBLAS_GFX950_OPT_STORE_TRIAL) {
    if (value < 0 || value > 2)
      return SPBLAS_GFX950_STATUS_INVALID_VALUE;
    handle->store_flavour = value;
    handle->nt_choice[0] = handle->nt_choice[1] = 0;
    return SPBLAS_GFX950_STATUS_SUCCESS;
  }
  return SPBLAS_GFX950_STATUS_INVALID_VALUE;
}

int spblas_gfx950_get_stream(spblas_gfx950_handle_t handle, void** stream) {
  if (!handle)
    return SPBLAS_GFX950_STATUS_INVALID_HANDLE;
  if (!stream)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  *stream = handle->stream;
  return SPBLAS_GFX950_STATUS_SUCCESS;
}

} // extern "C"
