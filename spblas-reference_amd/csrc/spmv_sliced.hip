// Column-sliced SpMV (SPBLAS_GFX950_SPMV_SLICED): placeholder until the first
// measurements decide the format.
#include "common.hpp"
#include "plan.hpp"

namespace spb {

int spmv_sliced_build(spblas_gfx950_handle_t, spblas_gfx950_plan_s*, const void*) {
  return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
}
int spmv_sliced_update(spblas_gfx950_handle_t, spblas_gfx950_plan_s*, const void*) {
  return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
}
int spmv_sliced_exec(spblas_gfx950_handle_t, const spblas_gfx950_plan_s*, const void*, const void*,
                     const void*, void*) {
  return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
}
void spmv_sliced_free(spblas_gfx950_handle_t, spblas_gfx950_plan_s*) {}

} // namespace spb
