/*
 * oracle_shim.c -- the part of the C ABI (include/spblas_gfx950.h) that the drop-in headers
 * (the vendor/gfx950 headers under include/spblas/) call, implemented ON THE HOST by the CPU oracle (oracle/spblas_oracle.c).
 *
 * TEST INFRASTRUCTURE ONLY.  Purpose: pin the oracle with the reference's OWN tests.  The reference's host test files
 * (test/gtest/{spmv,spmm,spgemm,spgemm_csr_csc,add,transpose,triangular_solve}_test.cpp) hold its known answers as
 * inline comparator loops; tests/compile_check/build_dropin.py compiles them unmodified against the drop-in header layer
 * and links them to THIS library instead of libspblas_gfx950.so, so that spblas::multiply & co. end in oracle_spmv_*,
 * oracle_spmm_*, oracle_spgemm_*, oracle_add_*, oracle_transpose_*, oracle_trsv_*.  If those tests pass, the oracle
 * reproduces every known answer the reference's tests hold for this path (tests/test_oracle_reference_tests.py, CPU).
 * Nothing in the product links or loads this file.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spblas_gfx950.h"

/* oracle/spblas_oracle.c */
#define DECL_T(S, T)                                                                                                 \
  int oracle_spmv_##S(int64_t, int64_t, int64_t, int64_t, const int32_t*, const int32_t*, const T*, int, T, const T*, \
                      int, T, T*);                                                                                   \
  int oracle_spmv_##S##_o64(int64_t, int64_t, int64_t, int64_t, const int64_t*, const int32_t*, const T*, int, T,     \
                            const T*, int, T, T*);                                                                   \
  int oracle_spmv_csc_##S(int64_t, int64_t, int64_t, int64_t, const int32_t*, const int32_t*, const T*, int, T,       \
                          const T*, int, T, T*);                                                                     \
  int oracle_spmm_##S(int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, const int32_t*, const int32_t*,  \
                      const T*, int, T, const T*, int64_t, int, T, T*, int64_t, int);                                \
  int oracle_spgemm_numeric_##S(int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, const int32_t*,                 \
                                const int32_t*, const T*, int, T, const int32_t*, const int32_t*, const T*, int, T,  \
                                int32_t*, int32_t*, T*, int64_t, int64_t*);                                          \
  int oracle_spgemm_numeric_d_##S(int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,             \
                                  const int32_t*, const int32_t*, const T*, T, const int32_t*, const int32_t*,       \
                                  const T*, T, const int32_t*, const int32_t*, const T*, int32_t*, int32_t*, T*,     \
                                  int64_t, int64_t*);                                                                \
  int oracle_add_##S(int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, const int32_t*, const int32_t*, const T*,  \
                     int, T, const int32_t*, const int32_t*, const T*, int, T, int32_t*, int32_t*, T*, int64_t,      \
                     int64_t*);                                                                                      \
  int oracle_trsv_##S(int64_t, int64_t, int64_t, int64_t, const int32_t*, const int32_t*, const T*, int, T, int, int, \
                      const T*, T*);                                                                                 \
  int oracle_transpose_##S(int64_t, int64_t, int64_t, int64_t, const int32_t*, const int32_t*, const T*, int64_t,     \
                           int32_t*, int32_t*, T*);
DECL_T(f32, float)
DECL_T(f64, double)
int oracle_spgemm_symbolic(int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, const int32_t*, const int32_t*,
                           const int32_t*, const int32_t*, int64_t*, int64_t*);
int oracle_spgemm_symbolic_d(int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, const int32_t*,
                             const int32_t*, const int32_t*, const int32_t*, const int32_t*, const int32_t*, int64_t*,
                             int64_t*);

struct spblas_gfx950_handle_s {
  int unused;
};
struct spblas_gfx950_plan_s {
  int unused;
};
struct spblas_gfx950_spgemm_s {
  int64_t m, k, n, c_nnz;
  int64_t d_nnz;
  const int32_t *d_rowptr, *d_colind;
  int has_addend;
};
struct spblas_gfx950_trsv_s {
  int uplo, diag;
};

/* Column order of a result the reference's own comparators do not look at (they re-accumulate C's rows through a sparse
 * accumulator, test/gtest/spgemm_test.cpp:56-65): every row of an SpGEMM / add result must come back with strictly
 * ascending columns (spgemm_gustavsons.hpp:42 sorts).  Checked here so that the reference's tests,
 * run on the oracle, fail when the oracle stops sorting. */
static int rows_strictly_ascending(int64_t m, const int32_t* rowptr, const int32_t* colind) {
  for (int64_t r = 0; r < m; ++r)
    for (int32_t p = rowptr[r] + 1; p < rowptr[r + 1]; ++p)
      if (colind[p - 1] >= colind[p]) {
        fprintf(stderr, "oracle shim: row %lld of the result is not in ascending column order\n", (long long) r);
        return 0;
      }
  return 1;
}
static int checked(int rc, int64_t m, const int32_t* rowptr, const int32_t* colind) {
  if (rc == 0 && !rows_strictly_ascending(m, rowptr, colind))
    return 3;
  return rc;
}

static int map_rc(int rc) {
  switch (rc) {
  case 0: return SPBLAS_GFX950_STATUS_SUCCESS;
  case 1: return SPBLAS_GFX950_STATUS_INVALID_SIZE;         /* shape */
  case 2: return SPBLAS_GFX950_STATUS_INSUFFICIENT_SPACE;   /* capacity */
  default: return SPBLAS_GFX950_STATUS_ALLOC_FAILED;
  }
}

int spblas_gfx950_version(void) { return 0; }
const char* spblas_gfx950_status_string(int status) { return status == 0 ? "success" : "oracle shim error"; }
int spblas_gfx950_last_hip_error(void) { return 0; }
int spblas_gfx950_create(spblas_gfx950_handle_t* handle, void* stream) {
  (void) stream;
  *handle = (spblas_gfx950_handle_t) calloc(1, sizeof(struct spblas_gfx950_handle_s));
  return *handle ? 0 : SPBLAS_GFX950_STATUS_ALLOC_FAILED;
}
int spblas_gfx950_destroy(spblas_gfx950_handle_t handle) {
  free(handle);
  return 0;
}
int spblas_gfx950_set_stream(spblas_gfx950_handle_t handle, void* stream) {
  (void) handle;
  (void) stream;
  return 0;
}
int spblas_gfx950_set_option(spblas_gfx950_handle_t handle, int option, int64_t value) {
  (void) handle;
  (void) option;
  (void) value;
  return 0;
}
int spblas_gfx950_ipc_alloc(size_t bytes, int uncached, void** ptr) {
  (void) uncached;
  *ptr = malloc(bytes ? bytes : 1);
  return *ptr ? 0 : SPBLAS_GFX950_STATUS_ALLOC_FAILED;
}
int spblas_gfx950_ipc_free(void* ptr) {
  free(ptr);
  return 0;
}

/* ---- SpMV / SpMM: plans are empty (the CPU inspect of the reference is empty too, multiply_impl.hpp:19-29) ---- */
int spblas_gfx950_spmv_plan_create(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t* plan, int64_t m, int64_t n,
                                   int64_t nnz, const void* rowptr, const int32_t* colind, const void* values,
                                   int offset_type, int value_type, int alg) {
  (void) handle; (void) m; (void) n; (void) nnz; (void) rowptr; (void) colind; (void) values; (void) offset_type;
  (void) value_type; (void) alg;
  *plan = (spblas_gfx950_plan_t) calloc(1, sizeof(struct spblas_gfx950_plan_s));
  return *plan ? 0 : SPBLAS_GFX950_STATUS_ALLOC_FAILED;
}
int spblas_gfx950_spmv_plan_update_values(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, const void* values) {
  (void) handle; (void) plan; (void) values;
  return 0;
}
/* the shim's plans are empty, there is nothing self-contained to keep: the owner goes on holding its arrays */
int spblas_gfx950_spmv_plan_detach(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan) {
  (void) handle; (void) plan;
  return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
}
int spblas_gfx950_plan_destroy(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan) {
  (void) handle;
  free(plan);
  return 0;
}
int spblas_gfx950_narrow_indices(spblas_gfx950_handle_t handle, int64_t count, const int64_t* src, int32_t* dst,
                                 int64_t bound) {
  (void) handle;
  if (count < 0 || bound < 0 || bound > (int64_t) INT32_MAX)
    return SPBLAS_GFX950_STATUS_INVALID_SIZE;
  for (int64_t i = 0; i < count; ++i) {
    if (src[i] < 0 || src[i] >= bound)
      return SPBLAS_GFX950_STATUS_INVALID_VALUE;
    dst[i] = (int32_t) src[i];
  }
  return 0;
}
int spblas_gfx950_spmm_inspect(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan) {
  (void) handle; (void) plan;
  return 0;
}

#define SPMV_BODY(S, T)                                                                                             \
  {                                                                                                                 \
    const T a = *(const T*) alpha, b = *(const T*) beta;                                                            \
    const int64_t ylen = op == SPBLAS_GFX950_OP_N ? m : n;                                                          \
    T* out = (T*) y;                                                                                                \
    T* tmp = NULL;                                                                                                  \
    if (b != (T) 0) {                                                                                               \
      tmp = (T*) malloc(sizeof(T) * (size_t) (ylen > 0 ? ylen : 1));                                                \
      if (!tmp) return SPBLAS_GFX950_STATUS_ALLOC_FAILED;                                                           \
      out = tmp;                                                                                                    \
    }                                                                                                               \
    int rc;                                                                                                         \
    if (op == SPBLAS_GFX950_OP_N) {                                                                                 \
      rc = offset_type == SPBLAS_GFX950_I32                                                                         \
               ? oracle_spmv_##S(m, n, m, n, (const int32_t*) rowptr, colind, (const T*) values, a != (T) 1, a,     \
                                 (const T*) x, 0, (T) 0, out)                                                       \
               : oracle_spmv_##S##_o64(m, n, m, n, (const int64_t*) rowptr, colind, (const T*) values, a != (T) 1,  \
                                       a, (const T*) x, 0, (T) 0, out);                                             \
    } else { /* y = A^T x: the CSR arrays of A are the CSC arrays of A^T (n x m) */                                 \
      if (offset_type != SPBLAS_GFX950_I32) { free(tmp); return SPBLAS_GFX950_STATUS_NOT_SUPPORTED; }               \
      rc = oracle_spmv_csc_##S(n, m, n, m, (const int32_t*) rowptr, colind, (const T*) values, a != (T) 1, a,       \
                               (const T*) x, 0, (T) 0, out);                                                        \
    }                                                                                                               \
    if (tmp) {                                                                                                      \
      for (int64_t i = 0; i < ylen; ++i) ((T*) y)[i] = tmp[i] + b * ((T*) y)[i];                                     \
      free(tmp);                                                                                                    \
    }                                                                                                               \
    return map_rc(rc);                                                                                              \
  }

int spblas_gfx950_spmv(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, int op, int64_t m, int64_t n,
                       int64_t nnz, const void* alpha, const void* rowptr, const int32_t* colind, const void* values,
                       const void* x, const void* beta, void* y, int offset_type, int value_type) {
  (void) handle; (void) plan; (void) nnz;
  if (value_type == SPBLAS_GFX950_F32)
    SPMV_BODY(f32, float)
  SPMV_BODY(f64, double)
}

int spblas_gfx950_spmm(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, int64_t m, int64_t k, int64_t n,
                       int64_t nnz, const void* alpha, const void* rowptr, const int32_t* colind, const void* values,
                       const void* B, int64_t ldb, const void* beta, void* C, int64_t ldc, int offset_type,
                       int value_type) {
  (void) handle; (void) plan; (void) nnz; (void) beta; /* the header layer always passes beta = 0 */
  if (offset_type != SPBLAS_GFX950_I32)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  if (value_type == SPBLAS_GFX950_F32) {
    const float a = *(const float*) alpha;
    return map_rc(oracle_spmm_f32(m, k, n, m, k, k, n, (const int32_t*) rowptr, colind, (const float*) values, a != 1.f, a,
                                  (const float*) B, ldb, 0, 0.f, (float*) C, ldc, 1));
  }
  const double a = *(const double*) alpha;
  return map_rc(oracle_spmm_f64(m, k, n, m, k, k, n, (const int32_t*) rowptr, colind, (const double*) values, a != 1.0, a,
                                (const double*) B, ldb, 0, 0.0, (double*) C, ldc, 1));
}

/* either mdspan layout for B and C (element (i, j) at i*rs + j*cs): both row-major or both column-major go straight to
 * the oracle's two layouts; a mixed pair is staged through row-major temporaries */
int spblas_gfx950_spmm_strided(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, int64_t m, int64_t k, int64_t n,
                               int64_t nnz, const void* alpha, const void* rowptr, const int32_t* colind,
                               const void* values, const void* B, int64_t brs, int64_t bcs, const void* beta, void* C,
                               int64_t crs, int64_t ccs, int offset_type, int value_type) {
  if (bcs == 1 && ccs == 1)
    return spblas_gfx950_spmm(handle, plan, m, k, n, nnz, alpha, rowptr, colind, values, B, brs, beta, C, crs, offset_type,
                              value_type);
  if (offset_type != SPBLAS_GFX950_I32)
    return SPBLAS_GFX950_STATUS_NOT_SUPPORTED;
  const size_t tsz = value_type == SPBLAS_GFX950_F32 ? 4 : 8;
  char* bt = (char*) malloc(tsz * (size_t) (k * n > 0 ? k * n : 1));
  char* ct = (char*) malloc(tsz * (size_t) (m * n > 0 ? m * n : 1));
  if (!bt || !ct) {
    free(bt);
    free(ct);
    return SPBLAS_GFX950_STATUS_ALLOC_FAILED;
  }
  for (int64_t i = 0; i < k; ++i)
    for (int64_t j = 0; j < n; ++j)
      memcpy(bt + tsz * (size_t) (i * n + j), (const char*) B + tsz * (size_t) (i * brs + j * bcs), tsz);
  const int rc = spblas_gfx950_spmm(handle, plan, m, k, n, nnz, alpha, rowptr, colind, values, bt, n > 0 ? n : 1, beta, ct,
                                    n > 0 ? n : 1, offset_type, value_type);
  if (rc == 0)
    for (int64_t i = 0; i < m; ++i)
      for (int64_t j = 0; j < n; ++j)
        memcpy((char*) C + tsz * (size_t) (i * crs + j * ccs), ct + tsz * (size_t) (i * n + j), tsz);
  free(bt);
  free(ct);
  return rc;
}

/* ---- SpGEMM and add ---- */
int spblas_gfx950_spgemm_create(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t* state) {
  (void) handle;
  *state = (spblas_gfx950_spgemm_t) calloc(1, sizeof(struct spblas_gfx950_spgemm_s));
  if (*state)
    (*state)->c_nnz = -1;
  return *state ? 0 : SPBLAS_GFX950_STATUS_ALLOC_FAILED;
}
int spblas_gfx950_spgemm_destroy(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t state) {
  (void) handle;
  free(state);
  return 0;
}
int spblas_gfx950_spgemm_set_addend(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t state, int64_t d_nnz,
                                    const int32_t* d_rowptr, const int32_t* d_colind) {
  (void) handle;
  state->has_addend = d_rowptr != NULL;
  state->d_nnz = d_nnz;
  state->d_rowptr = d_rowptr;
  state->d_colind = d_colind;
  return 0;
}
static int counts_to_rowptr(int64_t m, const int64_t* row_nnz, int32_t* c_rowptr) {
  int64_t run = 0;
  for (int64_t i = 0; i < m; ++i) {
    c_rowptr[i] = (int32_t) run;
    run += row_nnz[i];
  }
  c_rowptr[m] = (int32_t) run;
  return 0;
}
int spblas_gfx950_spgemm_symbolic(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t state, int64_t m, int64_t k,
                                  int64_t n, int64_t a_nnz, const int32_t* a_rowptr, const int32_t* a_colind,
                                  int64_t b_nnz, const int32_t* b_rowptr, const int32_t* b_colind, int32_t* c_rowptr,
                                  int64_t* c_nnz) {
  (void) handle; (void) a_nnz; (void) b_nnz;
  int64_t* row_nnz = (int64_t*) calloc((size_t) (m > 0 ? m : 1), sizeof(int64_t));
  if (!row_nnz)
    return SPBLAS_GFX950_STATUS_ALLOC_FAILED;
  int rc = state->has_addend
               ? oracle_spgemm_symbolic_d(m, k, n, m, n, k, m, n, a_rowptr, a_colind, b_rowptr, b_colind, state->d_rowptr,
                                          state->d_colind, row_nnz, c_nnz)
               : oracle_spgemm_symbolic(m, k, n, m, n, k, a_rowptr, a_colind, b_rowptr, b_colind, row_nnz, c_nnz);
  if (rc == 0)
    counts_to_rowptr(m, row_nnz, c_rowptr);
  free(row_nnz);
  state->m = m;
  state->k = k;
  state->n = n;
  state->c_nnz = rc == 0 ? *c_nnz : -1;
  return map_rc(rc);
}
int spblas_gfx950_spgemm_numeric(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t state, const void* alpha,
                                 const int32_t* a_rowptr, const int32_t* a_colind, const void* a_values,
                                 const int32_t* b_rowptr, const int32_t* b_colind, const void* b_values,
                                 int32_t* c_rowptr, int32_t* c_colind, void* c_values, int64_t c_capacity,
                                 int value_type) {
  (void) handle;
  int64_t nnz = 0;
  const int64_t m = state->m, k = state->k, n = state->n;
  if (state->c_nnz < 0)
    return SPBLAS_GFX950_STATUS_INVALID_VALUE;
  if (c_capacity < state->c_nnz)
    return SPBLAS_GFX950_STATUS_INSUFFICIENT_SPACE;
  if (value_type == SPBLAS_GFX950_F32) {
    const float a = *(const float*) alpha;
    return map_rc(checked(oracle_spgemm_numeric_f32(m, k, n, m, n, k, a_rowptr, a_colind, (const float*) a_values, a != 1.f, a,
                                                    b_rowptr, b_colind, (const float*) b_values, 0, 0.f, c_rowptr, c_colind,
                                                    (float*) c_values, c_capacity, &nnz),
                          m, c_rowptr, c_colind));
  }
  const double a = *(const double*) alpha;
  return map_rc(checked(oracle_spgemm_numeric_f64(m, k, n, m, n, k, a_rowptr, a_colind, (const double*) a_values, a != 1.0, a,
                                                  b_rowptr, b_colind, (const double*) b_values, 0, 0.0, c_rowptr, c_colind,
                                                  (double*) c_values, c_capacity, &nnz),
                        m, c_rowptr, c_colind));
}
int spblas_gfx950_spgemm_numeric_addend(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t state, const void* alpha,
                                        const int32_t* a_rowptr, const int32_t* a_colind, const void* a_values,
                                        const int32_t* b_rowptr, const int32_t* b_colind, const void* b_values,
                                        const void* beta, const int32_t* d_rowptr, const int32_t* d_colind,
                                        const void* d_values, int32_t* c_rowptr, int32_t* c_colind, void* c_values,
                                        int64_t c_capacity, int value_type) {
  (void) handle;
  int64_t nnz = 0;
  const int64_t m = state->m, k = state->k, n = state->n;
  if (state->c_nnz < 0 || !state->has_addend)
    return SPBLAS_GFX950_STATUS_INVALID_VALUE;
  if (c_capacity < state->c_nnz)
    return SPBLAS_GFX950_STATUS_INSUFFICIENT_SPACE;
  if (value_type == SPBLAS_GFX950_F32)
    return map_rc(checked(oracle_spgemm_numeric_d_f32(m, k, n, m, n, k, m, n, a_rowptr, a_colind, (const float*) a_values,
                                                      *(const float*) alpha, b_rowptr, b_colind, (const float*) b_values,
                                                      *(const float*) beta, d_rowptr, d_colind, (const float*) d_values,
                                                      c_rowptr, c_colind, (float*) c_values, c_capacity, &nnz),
                          m, c_rowptr, c_colind));
  return map_rc(checked(oracle_spgemm_numeric_d_f64(m, k, n, m, n, k, m, n, a_rowptr, a_colind, (const double*) a_values,
                                                    *(const double*) alpha, b_rowptr, b_colind, (const double*) b_values,
                                                    *(const double*) beta, d_rowptr, d_colind, (const double*) d_values,
                                                    c_rowptr, c_colind, (double*) c_values, c_capacity, &nnz),
                        m, c_rowptr, c_colind));
}
int spblas_gfx950_csr_add_symbolic(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t state, int64_t m, int64_t n,
                                   int64_t a_nnz, const int32_t* a_rowptr, const int32_t* a_colind, int64_t b_nnz,
                                   const int32_t* b_rowptr, const int32_t* b_colind, int32_t* c_rowptr, int64_t* c_nnz) {
  (void) handle; (void) a_nnz; (void) b_nnz;
  /* structural count: values == NULL (add_inspect, add_impl.hpp:79-108) */
  const int rc = oracle_add_f32(m, n, m, n, m, n, a_rowptr, a_colind, NULL, 0, 0.f, b_rowptr, b_colind, NULL, 0, 0.f, c_rowptr,
                                NULL, NULL, 0, c_nnz);
  state->m = m;
  state->n = n;
  state->c_nnz = rc == 0 ? *c_nnz : -1;
  return map_rc(rc);
}
int spblas_gfx950_csr_add_numeric(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t state, const void* alpha,
                                  const int32_t* a_rowptr, const int32_t* a_colind, const void* a_values, const void* beta,
                                  const int32_t* b_rowptr, const int32_t* b_colind, const void* b_values, int32_t* c_rowptr,
                                  int32_t* c_colind, void* c_values, int64_t c_capacity, int value_type) {
  (void) handle;
  int64_t nnz = 0;
  const int64_t m = state->m, n = state->n;
  if (value_type == SPBLAS_GFX950_F32) {
    const float a = *(const float*) alpha, b = *(const float*) beta;
    return map_rc(checked(oracle_add_f32(m, n, m, n, m, n, a_rowptr, a_colind, (const float*) a_values, a != 1.f, a, b_rowptr,
                                         b_colind, (const float*) b_values, b != 1.f, b, c_rowptr, c_colind, (float*) c_values,
                                         c_capacity, &nnz),
                          m, c_rowptr, c_colind));
  }
  const double a = *(const double*) alpha, b = *(const double*) beta;
  return map_rc(checked(oracle_add_f64(m, n, m, n, m, n, a_rowptr, a_colind, (const double*) a_values, a != 1.0, a, b_rowptr,
                                       b_colind, (const double*) b_values, b != 1.0, b, c_rowptr, c_colind, (double*) c_values,
                                       c_capacity, &nnz),
                        m, c_rowptr, c_colind));
}

/* ---- transpose, scale, triangular solve ---- */
int spblas_gfx950_csr_transpose(spblas_gfx950_handle_t handle, int64_t m, int64_t n, int64_t nnz, const int32_t* rowptr,
                                const int32_t* colind, const void* values, int32_t* t_rowptr, int32_t* t_colind,
                                void* t_values, int value_type) {
  (void) handle;
  if (value_type == SPBLAS_GFX950_F32)
    return map_rc(oracle_transpose_f32(m, n, n, m, rowptr, colind, (const float*) values, nnz, t_rowptr, t_colind,
                                       (float*) t_values));
  return map_rc(oracle_transpose_f64(m, n, n, m, rowptr, colind, (const double*) values, nnz, t_rowptr, t_colind,
                                     (double*) t_values));
}
int spblas_gfx950_scale(spblas_gfx950_handle_t handle, int64_t n, const void* alpha, void* values, int value_type) {
  (void) handle;
  if (value_type == SPBLAS_GFX950_F32)
    for (int64_t i = 0; i < n; ++i) ((float*) values)[i] *= *(const float*) alpha;
  else
    for (int64_t i = 0; i < n; ++i) ((double*) values)[i] *= *(const double*) alpha;
  return 0;
}
int spblas_gfx950_sptrsv_create(spblas_gfx950_handle_t handle, spblas_gfx950_trsv_t* plan, int64_t m, int64_t nnz,
                                const int32_t* rowptr, const int32_t* colind, int uplo, int diag) {
  (void) handle; (void) m; (void) nnz; (void) rowptr; (void) colind;
  *plan = (spblas_gfx950_trsv_t) calloc(1, sizeof(struct spblas_gfx950_trsv_s));
  if (!*plan)
    return SPBLAS_GFX950_STATUS_ALLOC_FAILED;
  (*plan)->uplo = uplo;
  (*plan)->diag = diag;
  return 0;
}
int spblas_gfx950_sptrsv_destroy(spblas_gfx950_handle_t handle, spblas_gfx950_trsv_t plan) {
  (void) handle;
  free(plan);
  return 0;
}
int spblas_gfx950_sptrsv_status(spblas_gfx950_handle_t handle, spblas_gfx950_trsv_t plan, int* status) {
  (void) handle; (void) plan;
  if (!status)
    return SPBLAS_GFX950_STATUS_INVALID_POINTER;
  *status = 0;  /* the oracle solves on the host, in program order */
  return SPBLAS_GFX950_STATUS_SUCCESS;
}
int spblas_gfx950_sptrsv_info(spblas_gfx950_trsv_t plan, int64_t info[4]) {
  (void) plan;
  memset(info, 0, 4 * sizeof(int64_t));
  return 0;
}
int spblas_gfx950_sptrsv_solve(spblas_gfx950_handle_t handle, spblas_gfx950_trsv_t plan, int64_t m, int64_t nnz,
                               const void* alpha, const int32_t* rowptr, const int32_t* colind, const void* values,
                               const void* b, void* x, int value_type) {
  (void) handle; (void) nnz;
  const int upper = plan->uplo == SPBLAS_GFX950_UPPER, unit = plan->diag == SPBLAS_GFX950_DIAG_UNIT;
  if (value_type == SPBLAS_GFX950_F32) {
    const float a = *(const float*) alpha;
    return map_rc(oracle_trsv_f32(m, m, m, m, rowptr, colind, (const float*) values, a != 1.f, a, upper, unit, (const float*) b,
                                  (float*) x));
  }
  const double a = *(const double*) alpha;
  return map_rc(oracle_trsv_f64(m, m, m, m, rowptr, colind, (const double*) values, a != 1.0, a, upper, unit, (const double*) b,
                                (double*) x));
}
