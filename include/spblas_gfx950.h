/*
 * spblas_gfx950.h -- C ABI of the MI355X (gfx950 / CDNA4) Sparse BLAS backend for
 * the spblas-reference view/operator API.
 *
 * This is the drop-in boundary.  The reference has no runtime
 * plugin ABI of its own: a backend is a set of C++ overloads compiled in with
 * -DSPBLAS_ENABLE_<NAME> whose bodies call a vendor C library.  The entry points
 * below are exactly what our overload set (the headers under
 * include/spblas/vendor/gfx950/) binds, one for each vendor-library call the existing AMD slot makes
 * (paths relative to /root/reference/include/spblas/):
 *
 *   rocsparse_create_handle / rocsparse_set_stream / rocsparse_destroy_handle
 *       vendor/rocsparse/detail/abstract_operation_state.hpp:20-28,
 *       vendor/rocsparse/multiply_spgemm.hpp:34-43      -> spblas_gfx950_create/destroy/set_stream
 *   rocsparse_spmv(..., stage_buffer_size) + allocate_workspace
 *       vendor/rocsparse/detail/spmv_impl.hpp:60-69     -> spblas_gfx950_spmv_plan_create  (multiply_inspect)
 *   rocsparse_spmv(..., stage_compute)
 *       vendor/rocsparse/detail/spmv_impl.hpp:72-77     -> spblas_gfx950_spmv
 *   oneapi::mkl::sparse::gemm (the only device SpMM in the reference)
 *       vendor/onemkl_sycl/spmm_impl.hpp:116-120        -> spblas_gfx950_spmm
 *   rocsparse_spgemm stage_buffer_size + stage_nnz + rocsparse_spmat_get_size
 *       vendor/rocsparse/multiply_spgemm.hpp:94-115     -> spblas_gfx950_spgemm_symbolic (multiply_compute)
 *   rocsparse_spgemm stage_compute / stage_symbolic / stage_numeric
 *       vendor/rocsparse/multiply_spgemm.hpp:137-213    -> spblas_gfx950_spgemm_numeric  (multiply_fill / multiply_numeric)
 *
 * Conventions
 *   - Plain C: pointers, sizes, enums.  No C++/torch types cross this boundary.
 *   - Every array pointer is a DEVICE pointer owned by the caller (csr_view is
 *     non-owning, views/csr_view.hpp:12-77).  alpha/beta are HOST pointers to one
 *     scalar of the value type.  The library never frees caller memory.
 *   - Work is enqueued on the handle's hipStream_t and is asynchronous with respect
 *     to the host, like the rocSPARSE slot; the only host synchronisations are in
 *     plan creation and in spgemm_symbolic (it must return nnz(C), as
 *     rocsparse_spmat_get_size does at multiply_spgemm.hpp:114-115).
 *   - Every function returns a spblas_gfx950_status; the C++ header layer maps
 *     them to the reference's exception types (std::invalid_argument for shape
 *     errors, std::runtime_error otherwise, std::bad_alloc for allocation).
 *   - Zero-based indices; column indices within a row may be unsorted and may
 *     repeat (backend/generate.hpp:112-117 shuffles them).
 *   - One handle / plan per host thread; no internal locking (the reference makes
 *     no thread-safety statement either).
 */
#ifndef SPBLAS_GFX950_H
#define SPBLAS_GFX950_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spblas_gfx950_handle_s* spblas_gfx950_handle_t;
typedef struct spblas_gfx950_plan_s* spblas_gfx950_plan_t;
typedef struct spblas_gfx950_spgemm_s* spblas_gfx950_spgemm_t;
typedef struct spblas_gfx950_trsv_s* spblas_gfx950_trsv_t;

typedef enum spblas_gfx950_status {
  SPBLAS_GFX950_STATUS_SUCCESS = 0,
  SPBLAS_GFX950_STATUS_INVALID_HANDLE = 1,
  SPBLAS_GFX950_STATUS_INVALID_POINTER = 2,
  SPBLAS_GFX950_STATUS_INVALID_SIZE = 3,  /* shape mismatch -> std::invalid_argument */
  SPBLAS_GFX950_STATUS_INVALID_VALUE = 4, /* bad enum / flag */
  SPBLAS_GFX950_STATUS_NOT_SUPPORTED = 5,
  SPBLAS_GFX950_STATUS_ALLOC_FAILED = 6,       /* -> std::bad_alloc */
  SPBLAS_GFX950_STATUS_HIP_ERROR = 7,          /* see spblas_gfx950_last_hip_error */
  SPBLAS_GFX950_STATUS_INSUFFICIENT_SPACE = 8, /* "SpGEMM ran out of memory" */
  SPBLAS_GFX950_STATUS_PLAN_MISMATCH = 9       /* plan built for a different matrix */
} spblas_gfx950_status;

typedef enum spblas_gfx950_datatype {
  SPBLAS_GFX950_F32 = 0, /* float  : vendor/rocsparse/types.hpp:42-44 */
  SPBLAS_GFX950_F64 = 1  /* double : vendor/rocsparse/types.hpp:47-49 */
} spblas_gfx950_datatype;

typedef enum spblas_gfx950_indextype {
  SPBLAS_GFX950_I32 = 0, /* vendor/rocsparse/types.hpp:11-12 (index_t = offset_t = int32_t) */
  SPBLAS_GFX950_I64 = 1  /* row offsets only; column indices are always int32 */
} spblas_gfx950_indextype;

typedef enum spblas_gfx950_operation {
  SPBLAS_GFX950_OP_N = 0, /* csr_view                      (vendor/rocsparse/detail/get_transpose.hpp:25-26) */
  SPBLAS_GFX950_OP_T = 1  /* csc_view / transposed(csr)    (vendor/rocsparse/detail/get_transpose.hpp:27-28) */
} spblas_gfx950_operation;

/* SpMV algorithm selector (plan creation).  AUTO picks from the row statistics. */
typedef enum spblas_gfx950_spmv_alg {
  SPBLAS_GFX950_SPMV_AUTO = 0,
  SPBLAS_GFX950_SPMV_VECTOR = 1,   /* sub-wavefront group per row, no plan data */
  SPBLAS_GFX950_SPMV_ROWBLOCK = 2, /* nnz-window row blocks staged through LDS   */
  SPBLAS_GFX950_SPMV_SLICED = 3    /* column-sliced reorder (x slice L2-resident) */
} spblas_gfx950_spmv_alg;

/* ---- library / handle ---------------------------------------------------- */
int spblas_gfx950_version(void);
const char* spblas_gfx950_status_string(int status);
/* hipError_t of the most recent STATUS_HIP_ERROR on this thread (0 if none). */
int spblas_gfx950_last_hip_error(void);

/* stream: a hipStream_t (NULL = the null stream), cf. hip_allocator(hipStream_t),
 * vendor/rocsparse/hip_allocator.hpp:22.
 * Graph capture: the execute calls that take a plan or a state whose structure is known (spblas_gfx950_spmv,
 * spblas_gfx950_spmm, spblas_gfx950_sptrsv_solve, spblas_gfx950_spgemm_numeric after the first fill) only launch kernels
 * and memsets on this stream and may be recorded with hipStreamBeginCapture and replayed.  Nothing is allocated on a
 * capturing stream: a call that would have to (plan creation, inspect, symbolic passes, the first execute of a plan that
 * sizes a workspace) returns SPBLAS_GFX950_STATUS_NOT_SUPPORTED there -- run it once outside the capture.
 * The FIRST handle a process creates loads the library's code objects on the current device (about 16 ms on MI355X, once):
 * the runtime would otherwise load each one at the first launch of one of its kernels -- 5.5 of the 9 ms of a first
 * multiply_inspect, 2.2 of the 3.2 ms of a first multiply_compute.  SPBLAS_GFX950_PRELOAD=0 in the environment leaves it to
 * the first use. */
int spblas_gfx950_create(spblas_gfx950_handle_t* handle, void* stream);
int spblas_gfx950_destroy(spblas_gfx950_handle_t handle);
int spblas_gfx950_set_stream(spblas_gfx950_handle_t handle, void* stream);
int spblas_gfx950_get_stream(spblas_gfx950_handle_t handle, void** stream);

/* Handle options consulted by later plan creations. */
typedef enum spblas_gfx950_option {
  /* Row-sharded multi-GPU runs gather y stripe by stripe (spblas-reference_amd/sharded.py).
   * value > 0 asks the SLICED inspect to put its row-bin boundaries on divisors of `value`
   * (the stripe length), so that spblas_gfx950_spmv_reduce_rows can finish whole stripes. */
  SPBLAS_GFX950_OPT_BIN_ROW_ALIGN = 1,
  /* value > 0 caps the slice split K of spblas_gfx950_spmv_reduce_rows (0 = heuristic only).  Callers
   * that run the reduces of several stripes side by side on different streams keep K small so the
   * partial-sum traffic does not grow with the number of stripes. */
  SPBLAS_GFX950_OPT_MAX_KSPLIT = 2,
  /* value = 1 lets plan creation with alg = AUTO choose the SLICED plan, which re-tiles A and keeps a COPY
   * of its values.  Default 0: AUTO only picks plans that read the caller's value array on every multiply,
   * which is what the reference's CPU path and the rocSPARSE slot do (algorithms/multiply_impl.hpp:48-52,
   * vendor/rocsparse/detail/spmv_impl.hpp:72-77).  The host layers set it for operands wrapped in
   * matrix_opt (views/matrix_opt_impl.hpp: the view that owns a vendor-optimised form of the matrix), the
   * same opt-in oneMKL's optimize_* stage gets in vendor/onemkl_sycl/spmm_impl.hpp:48-61.  Asking for
   * alg = SLICED explicitly needs no option.
   * value = 2: as 1, and the caller announces that the values WILL change (a time-stepping solver): SLICED snapshot
   * plans -- chosen by AUTO or requested explicitly -- keep the source position of every entry from the start (+4 B per
   * entry), so that the first spblas_gfx950_spmv_plan_update_values is a gather like every later one instead of a second
   * inspect (see the value snapshot contract below). */
  SPBLAS_GFX950_OPT_VALUE_SNAPSHOT = 3,
  /* value = 1: the caller GUARANTEES that a c_colind array it passes to spblas_gfx950_spgemm_numeric[_addend] still
   * holds what the previous numeric call on the same state wrote there whenever it is the same address; repeated
   * fills of one result then leave the column indices alone (cfg5: 1.40 -> 0.94 ms per fill).  Default 0: every
   * numeric call writes c_colind -- an equal address proves nothing (allocators hand freed addresses out again:
   * the reference's SpGEMMReuseAndChangePointer test does exactly that, test/gtest/device/spgemm_reuse_test.cpp:325). */
  SPBLAS_GFX950_OPT_SPGEMM_KEEP_COLIND = 4,
  /* SLICED plans store their products with plain stores (value 0, the default) or with the non-temporal hint
   * (value 1).  value 2 = decide by a timed trial: the first plan of this HANDLE with >= 32 M placed entries per
   * value size runs eight SpMVs on a zero vector at inspect and the handle keeps the faster flavour for its later
   * plans.  Which flavour wins is a property of the machine (-1...-4 % where the reduce pays for the expand's
   * write-backs, +3 % where it does not); the trial roughly doubles the first inspect, so it is for callers who
   * will run hundreds of multiplies.  Results are identical either way.  The environment variable
   * SPBLAS_GFX950_PB_NT = 0 / 1 / -2 (trial) overrides the option for reproducible runs. */
  SPBLAS_GFX950_OPT_STORE_TRIAL = 5
} spblas_gfx950_option;
int spblas_gfx950_set_option(spblas_gfx950_handle_t handle, int option, int64_t value);

/* ---- SpMV:  y = alpha * op(A) * x + beta * y ------------------------------ */
/* multiply_inspect(A, x, y): device-side analysis of the sparsity pattern.
 * Builds the nnz-window row partition, the long-row list and (SLICED) the
 * column-sliced reorder; the result lives in device memory owned by the plan.
 * `values` may be NULL unless alg == SLICED.
 * Value snapshot contract: ROWBLOCK / VECTOR plans hold structure only and every multiply reads the
 * caller's values.  The SLICED plan holds a re-tiled COPY of the values.  On request (alg = SLICED) or by AUTO under
 * SPBLAS_GFX950_OPT_VALUE_SNAPSHOT the copy is a SNAPSHOT: after changing the values IN PLACE call
 * spblas_gfx950_spmv_plan_update_values; a multiply that passes a DIFFERENT values pointer than the one
 * the copy was taken from refreshes the copy by itself first (one extra pass over A).  A snapshot plan keeps no source
 * positions until then: the FIRST refresh builds the plan again from the caller's arrays (inspect-class work: not inside a
 * stream capture: STATUS_NOT_SUPPORTED there, also from a spblas_gfx950_spmv that meets a new values pointer) and keeps
 * them, every later refresh is a gather (option value 2 above: kept from the start).  If that second build runs out of memory
 * or declines, the plan falls back to its structure-only form (row-block windows on the caller's arrays, plan_info[0]
 * changes) and the call succeeds: a plan is never left half built.  AUTO WITHOUT the option may choose the SLICED plan
 * too (>= 16 M entries, not skewed): such a plan reads the caller's values on EVERY multiply (plan_info_sliced[9] bit 6),
 * so the caller sees the same semantics as with a structure-only plan.  Its default form is VALUE-FREE (bit 7, round 5):
 * the plan holds no values at all -- the first kernel gathers x, the second multiplies by the caller's array, staged bin by
 * bin through LDS; where that form does not apply, the copying form of round 4 (a value refresh per multiply, kept only
 * if it beats the row-block kernel by rule -- x of 32 MB or more -- or, with SPBLAS_GFX950_AUTO_TRIAL=1, in a timed trial).
 * The calls that are not given the values -- the two-stage calls (spmv_expand / spmv_reduce_rows) and the multi-GPU steps
 * (spmv_reduce_rows_bcast, spmv_step_bcast[_chunked]) -- refuse the COPYING form.  Since round 6 the value-free form is
 * accepted by spmv_reduce_rows_bcast and spmv_step_bcast: it reads the value array registered with the plan (plan_create,
 * the last spblas_gfx950_spmv, plan_update_values) as it is when the step runs -- no copy that could go stale.
 * The plan is tied to (m, n, nnz, rowptr, colind). */
int spblas_gfx950_spmv_plan_create(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t* plan,
                                   int64_t m, int64_t n, int64_t nnz, const void* rowptr,
                                   const int32_t* colind, const void* values, int offset_type,
                                   int value_type, int alg);
int spblas_gfx950_spmv_plan_update_values(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan,
                                          const void* values);
/* The owner of a plan declares that it is about to RELEASE the arrays the plan was built from -- what a host layer does
 * for an inspected csc_view / transposed(csr) operand, whose row-major form it materialised for the plan alone
 * (vendor/rocsparse/detail/get_transpose.hpp:19-29 has rocSPARSE do the transposition inside the call; here it is done
 * once, at inspect).  Succeeds only for a self-contained plan: SLICED with its own copy of the values, no hub rows, no
 * hot-column split (STATUS_NOT_SUPPORTED otherwise: keep the arrays).  Afterwards spblas_gfx950_spmv with this plan takes
 * rowptr = colind = values = NULL (op = N, same m / n / nnz / types), plan_update_values returns STATUS_NOT_SUPPORTED (the
 * values changed: inspect again), spblas_gfx950_spmm does not accept the plan.  cfg2-sized transposed operand: the inspected
 * form holds 1.43 x the matrix instead of 2.43 x. */
int spblas_gfx950_spmv_plan_detach(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan);
/* A plan carries workspaces (products, partial sums): it must not run on two streams at once.  plan_destroy frees on
 * the handle's current stream, ordered behind the last launch that used the plan on whichever stream that was. */
int spblas_gfx950_plan_destroy(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan);
/* Introspection for tests/bench: info[0]=alg, [1]=window nnz, [2]=#windows,
 * [3]=#long rows, [4]=max row length, [5]=device bytes held, [6]=#column slices,
 * [7]=#empty rows, [8]=rows per row-bin (SLICED), [9]=1 if the bins honour BIN_ROW_ALIGN,
 * [10]=#expand work items, [11]=#reduce work items (SLICED plans of skewed matrices; else 0). */
int spblas_gfx950_plan_info(spblas_gfx950_plan_t plan, int64_t info[12]);
/* More of a SLICED plan (zeros for other plans): info[0]=#row-bins, [1]=1 if the bins have variable heights (row-skewed
 * matrix: a new bin every [8] rows and every ~nnz/2048 entries), [2]=blocks of 32 entries in expand order,
 * [3]=blocks in reduce order (bins padded to groups of 8 blocks), [4]=entries placed in tiles, [5]=#rows kept out of
 * the tiles (hub rows), [6]=hub threshold (row length), [7]=reduce K split, [8]=rows the tiles are built over (fewer
 * than m when the empty rows were taken out).  For ANY plan: [9] bit 0 = AUTO decided by a timed trial, bit 1 = the
 * reduce streams one-byte row codes (runs sorted by row) instead of 16-bit rows, bit 2 = the expand stores its products
 * with the non-temporal hint, bit 3 = this plan ran the store trial that decides bit 2 (plans with >= 32 M placed
 * entries, once per process, device and value size; SPBLAS_GFX950_PB_NT=0/1 forces the flavour), bit 4 = hot-column
 * split (plan_info_hot), bit 6 = the plan reads the caller's values on every multiply (made without the snapshot opt-in),
 * bit 7 = ... and holds no copy of them (value-free tiles); [10]/[11]=time of the
 * row-block / the sliced plan in AUTO's trial, nanoseconds -- or, when only the store trial ran, of one SpMV with plain /
 * non-temporal product stores. */
int spblas_gfx950_plan_info_sliced(spblas_gfx950_plan_t plan, int64_t info[12]);
/* Hot-column split of a SLICED plan (column-skewed matrices, csrc/spmv_hot.hip): [0] hot columns, [1] entries multiplied
 * in row order with those x values in LDS, [2] rows that have such entries, [3] of them longer than a window, [4] entries
 * left to the tiles, [5] device bytes of the tiled part, [6] windows of 256 entries the hot part is cut into, [7] = 0.
 * All 0 for a plan without the split. */
int spblas_gfx950_plan_info_hot(spblas_gfx950_plan_t plan, int64_t info[8]);

/* Two-stage execution of a SLICED plan (other plans: STATUS_NOT_SUPPORTED), used to overlap the
 * multi-GPU all-gather of finished y rows with the rest of the SpMV:
 *   expand       products of every stored entry with x (x read once, LDS-resident slices)
 *   reduce_rows  y[r] = alpha * (A x)[r] + beta * y[r] for the row-bins that START in
 *                [row_begin, row_end), from the products of the last expand.  `y` is the base of
 *                the full local y (entry r at y[r]).  spblas_gfx950_spmv == expand + reduce_rows(0, m).
 *                A plan of a row-skewed matrix that cuts its long rows into pieces (plan_info_sliced[8] > m or the
 *                plan was made without SPBLAS_GFX950_OPT_BIN_ROW_ALIGN on such a matrix) reduces all rows in ONE call:
 *                a proper sub-range returns STATUS_NOT_SUPPORTED there.  Plans created under OPT_BIN_ROW_ALIGN > 1 (what
 *                the striped multi-GPU step sets) never cut rows. */
int spblas_gfx950_spmv_expand(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, const void* x);
int spblas_gfx950_spmv_reduce_rows(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, const void* alpha,
                                   const void* beta, void* y, int64_t row_begin, int64_t row_end);

/* multiply(info, A, x, y) / multiply(A, x, y).  plan may be NULL (no inspect):
 * a plan-free kernel is chosen from nnz/m.  op == OP_T computes y = alpha*A^T*x
 * for an m x n CSR A (y has n entries, x has m): the CSC / transposed case.
 * op == OP_T without a plan, from 4 M entries and 65 536 columns on: the products go through a workspace in the handle's
 * scratch (6 B per fp32 entry, 10 B per fp64 entry + 4 B per (8 192-entry tile, column slice) pair; kept by the handle until
 * spblas_gfx950_destroy), 0.66 ms at 1e8 entries against 4.8 ms for one float atomic per entry -- which is what a call
 * recorded in a stream capture, or one the scratch cannot be allocated for, still does.  Either way the additions into one
 * element of y can come in another order on the next call (reproducibility); an inspected operand
 * (multiply_inspect on the csc_view) runs the CSR kernels, whose order is fixed. */
int spblas_gfx950_spmv(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, int op, int64_t m,
                       int64_t n, int64_t nnz, const void* alpha, const void* rowptr,
                       const int32_t* colind, const void* values, const void* x, const void* beta,
                       void* y, int offset_type, int value_type);

/* ---- multi-GPU: fused all-gather of y (second stage) -------------------- */
/* One process per GPU; the reference has no multi-device path, so nothing is mirrored here.
 * ipc_alloc/export/open: every rank allocates its copy of the full y (and, with uncached = 1 because
 * it is polled while peers write it, a small flag array) with ipc_alloc, exports a 64-byte handle (hipIpcGetMemHandle) that it sends to the other ranks by any
 * host channel, and maps their buffers with ipc_open (hipIpcOpenMemHandle, peer access enabled).
 * spmv_reduce_rows_bcast: like spmv_reduce_rows with beta = 0, but local row r is stored as row
 * y_row_offset + r into ALL n_peers buffers of the DEVICE array y_peers (the local copy and the mapped
 * copies of the other ranks) -- on xGMI the traffic of a direct all-gather, issued by the reduce /
 * combine kernels themselves.  step_signal / step_wait: device-side barrier that ends a step:
 * signal stores `step` into slot `rank` of every rank's flag array (after the producing kernels on
 * the same stream), wait spins until all n_peers slots of the local array reached `step`
 * (status_dev[0] = 1 after timeout_ms without progress). */
int spblas_gfx950_ipc_alloc(size_t bytes, int uncached, void** ptr);
int spblas_gfx950_ipc_free(void* ptr);
int spblas_gfx950_ipc_export(void* ptr, unsigned char handle[64]);
int spblas_gfx950_ipc_open(const unsigned char handle[64], void** ptr);
int spblas_gfx950_ipc_close(void* ptr);
int spblas_gfx950_spmv_reduce_rows_bcast(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, const void* alpha,
                                         void* const* y_peers, int n_peers, int64_t y_row_offset,
                                         int64_t row_begin, int64_t row_end);
/* expand + reduce_rows_bcast of the whole local matrix in one call.  stripes > 1 cuts the row bins into
 * that many contiguous groups whose reduces alternate between the handle's stream and an auxiliary
 * stream owned by the handle, so that the link-bound peer stores of one stripe overlap the reduce of
 * the next; the handle's stream is joined with the auxiliary one before the call returns. */
int spblas_gfx950_spmv_step_bcast(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, const void* alpha,
                                  const void* x, void* const* y_peers, int n_peers, int64_t y_row_offset,
                                  int stripes);
/* Dependent iteration (x of step k + 1 is y of step k) with the peers' rows arriving BEHIND the next expand instead of in
 * front of it (round 4).  The reduce runs in `chunks` stripes as in spmv_step_bcast; after each stripe's peer stores a
 * small kernel on the same stream publishes slot rank * chunks + c = step in every rank's flag array (flag_peers: device
 * array of the n_peers flag arrays, int64[n_peers * chunks] each, uncached memory).  With `wait` the expand of THIS step --
 * whose x must be this rank's copy of the previous step's y -- waits, x slice by x slice, for the chunks that slice is made
 * of (own rows: stream order; bounded spin with s_sleep; a time-out sets status_dev[0] and lets the kernel finish) and reads
 * the slice with system-scope loads; status_dev[1] receives the longest wait of a workgroup in wall-clock ticks.  There is
 * no step barrier: a rank's step k + 2 overwrites copy k & 1 only after its expand has seen every rank's chunks of step
 * k + 1, which they publish after their expand of step k + 1 has read that copy.  spmv_chunk_rows: the local row
 * boundaries (chunks + 1, host) of the stripes -- every rank needs every rank's (chunk_rows of the wait descriptor: global
 * rows, device, int64[n_ranks * (chunks + 1)]).  Plans cut on the arithmetic bin grid only (row shards of a matrix with
 * uniform rows); STATUS_NOT_SUPPORTED otherwise.  Nothing here has been run across
 * devices yet: n ranks sharing one device exercise the code path (tests/mp_fused_worker.py), max_expand_workgroups
 * keeps their waiting expands from filling that one device. */
typedef struct spblas_gfx950_chunk_wait {
  const void* flags;          /* this rank's flag array */
  const int64_t* chunk_rows;  /* device */
  int n_ranks, chunks;
  int64_t step;               /* wait until the slots have reached this step */
  int64_t timeout_ms;
  int* status_dev;            /* device int[2] */
  int max_expand_workgroups;  /* 0 = one workgroup per x slice */
} spblas_gfx950_chunk_wait;
int spblas_gfx950_spmv_chunk_rows(spblas_gfx950_plan_t plan, int chunks, int64_t* rows);
int spblas_gfx950_spmv_step_bcast_chunked(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, const void* alpha,
                                          const void* x, void* const* y_peers, int n_peers, int64_t y_row_offset,
                                          int chunks, void* const* flag_peers, int rank, int64_t step,
                                          const spblas_gfx950_chunk_wait* wait);
int spblas_gfx950_step_signal(spblas_gfx950_handle_t handle, void* const* flag_peers, int n_peers, int rank,
                              int64_t step);
int spblas_gfx950_step_wait(spblas_gfx950_handle_t handle, const void* flags, int n_peers, int64_t step,
                            int64_t timeout_ms, int* status_dev);
/* Tick rate (kHz) of the device clock the waits above are bounded by and the wait statistics (status_dev[1] of a
 * chunk wait) are counted in: hipDeviceAttributeWallClockRate of the handle's device (100 MHz on gfx9).  No vendor call
 * replaced: rocSPARSE has no device-side waits. */
int spblas_gfx950_wall_clock_khz(spblas_gfx950_handle_t handle, int* khz);
/* One-shot: the NEXT spmv_reduce_rows_bcast on this handle issues step_wait(flags, n_peers, step, ...) right before the
 * kernel that stores into the peers' copies of y (the combine kernel when the reduce is K-split, else the reduce
 * itself).  Throughput form of the fused step for independent right-hand sides: expand and reduce of step k overlap
 * the peers' stores of step k-1. */
int spblas_gfx950_bcast_wait_before(spblas_gfx950_handle_t handle, const void* flags, int n_peers, int64_t step,
                                    int64_t timeout_ms, int* status_dev);

/* ---- SpMM:  C = alpha * A * B + beta * C,  B (k x n), C (m x n) row-major --- */
/* ldb/ldc are row strides in elements (mdspan layout_right, test/gtest/spmm_test.cpp).
 * multiply_inspect(A, B, C) = spmv_plan_create (alg ROWBLOCK: row statistics, long-row list) followed by
 * spmm_inspect, the counterpart of oneMKL's optimize_gemm (vendor/onemkl_sycl/spmm_impl.hpp:40-67): it probes every
 * block of 32 rows for column locality and hands the blocks whose entries fall into <= 8 aligned tiles of 128
 * columns (>= 1/5 dense: the measured crossover) to the LDS-staged matrix-core kernel (fp32; exact f32 MFMA).  spmm with such a plan
 * also cuts rows longer than the plan's nnz window into parts (hub rows of power-law matrices).  plan = NULL:
 * one lane group per row, no analysis.  spmm_plan_info: [0] inspected, [1] row blocks on the matrix cores,
 * [2] entries inside them, [3] long rows. */
int spblas_gfx950_spmm_inspect(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan);
int spblas_gfx950_spmm_plan_info(spblas_gfx950_plan_t plan, int64_t info[4]);
int spblas_gfx950_spmm(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, int64_t m, int64_t k,
                       int64_t n, int64_t nnz, const void* alpha, const void* rowptr,
                       const int32_t* colind, const void* values, const void* B, int64_t ldb,
                       const void* beta, void* C, int64_t ldc, int offset_type, int value_type);
/* The same product for dense operands of either mdspan layout: element (i, j) of B lives at B[i*b_row_stride +
 * j*b_col_stride], likewise C.  The reference's CPU path takes any layout through mdspan's operator()
 * (backend/view_customizations.hpp:230-240; mdspan_col_major is a public alias, detail/mdspan.hpp:31-36).  Both
 * column strides 1 (layout_right): identical to spblas_gfx950_spmm with ldb / ldc = the row strides, plan included.
 * Any other combination of layout_right / layout_left operands (row stride 1 and column stride >= rows, or column
 * stride 1 and row stride >= n) runs a lane-group-per-row kernel whose gathers follow the strides; the plan is only
 * checked.  Overlapping layouts return STATUS_INVALID_SIZE. */
int spblas_gfx950_spmm_strided(spblas_gfx950_handle_t handle, spblas_gfx950_plan_t plan, int64_t m, int64_t k,
                               int64_t n, int64_t nnz, const void* alpha, const void* rowptr,
                               const int32_t* colind, const void* values, const void* B, int64_t b_row_stride,
                               int64_t b_col_stride, const void* beta, void* C, int64_t c_row_stride,
                               int64_t c_col_stride, int offset_type, int value_type);

/* ---- SpGEMM:  C = alpha * A * B   (CSR x CSR -> CSR, int32 indices) -------- */
/* State object = spgemm_state_t (vendor/rocsparse/multiply_spgemm.hpp:28-230). */
int spblas_gfx950_spgemm_create(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t* state);
int spblas_gfx950_spgemm_destroy(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t state);
/* multiply_compute / multiply_symbolic_compute: structural product.  Writes
 * c_rowptr[0..m] (device, caller-allocated: test/gtest/device/spgemm_test.cpp:37-40)
 * and returns nnz(C) = sum_i |union_{k in A_i} cols(B_k)| in *c_nnz (host).
 * Synchronises the stream once. */
int spblas_gfx950_spgemm_symbolic(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t state, int64_t m,
                                  int64_t k, int64_t n, int64_t a_nnz, const int32_t* a_rowptr,
                                  const int32_t* a_colind, int64_t b_nnz, const int32_t* b_rowptr,
                                  const int32_t* b_colind, int32_t* c_rowptr, int64_t* c_nnz);
/* multiply_fill / multiply_symbolic_fill + multiply_numeric: fills c_colind
 * (ascending within each row, as spgemm_gustavsons.hpp:42 sorts them) and c_values.
 * c_capacity = entries available in c_colind / c_values; fewer than nnz(C) ->
 * STATUS_INSUFFICIENT_SPACE (csr_builder.hpp:18-22).  c_rowptr is rewritten from
 * the state's copy, so a different buffer than the one passed to symbolic is fine
 * (test/gtest/device/spgemm_reuse_test.cpp).  a_values/b_values may change between
 * calls; the pattern may not.
 * Memory held by the state: 8 bytes per entry of A from the symbolic pass on (the bounds of the B row
 * every A entry selects) and 16 bytes per row of 65..256 products that a wavefront can sort in one round
 * (spblas_gfx950_spgemm_info: "direct" rows); fp32 fills with many such rows keep an interleaved
 * (column, value) copy of B, 8 bytes per entry, rewritten by every fill; from the SECOND numeric pass on -- a one-shot fill pays nothing -- also one
 * byte per product of the rows with at most 256 products, two per product of the rows with 257..1024,
 * and 4 bytes per entry of C: later passes accumulate by recorded rank (SPBLAS_GFX950_SPGEMM_REUSE=0
 * in the environment keeps every pass on the hash kernels).  All of it is optional: an allocation
 * failure leaves the hash path in place. */
int spblas_gfx950_spgemm_numeric(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t state, const void* alpha,
                                 const int32_t* a_rowptr, const int32_t* a_colind, const void* a_values,
                                 const int32_t* b_rowptr, const int32_t* b_colind, const void* b_values,
                                 int32_t* c_rowptr, int32_t* c_colind, void* c_values, int64_t c_capacity,
                                 int value_type);
/* Introspection of a state after spgemm_symbolic (no reference counterpart; the tests and bench.py use it):
 * info[0] = nnz(C), info[1] = rows with 65..256 products (the wave-per-row bin), info[2] = those of them that are
 * "direct" -- product count == structural length and an A row of one round of loads: sorted in registers by the
 * persistent kernel, no hash --, info[3] = 1 when later fills accumulate by recorded rank, info[4..7] = 0. */
int spblas_gfx950_spgemm_info(spblas_gfx950_spgemm_t state, int64_t info[8]);

/* Four-argument SpGEMM  C = alpha*A*B + beta*D  (the reference
 * surface is multiply_compute, multiply_fill, multiply_symbolic_compute, multiply_symbolic_fill and
 * multiply_numeric taking (state, a, b, c, d),
 * vendor/rocsparse/multiply_spgemm.hpp:118-214,237-274, with alpha = scale(a)*scale(b) and
 * beta = scale(d); expected values test/gtest/device/rocsparse/spgemm_4args_test.cpp:78-95).
 * set_addend registers D's pattern (m x n, int32) BEFORE spgemm_symbolic, which then counts
 * pattern(A*B) U pattern(D); passing d_rowptr = NULL removes it again.  After such a symbolic
 * pass the numeric step must be spgemm_numeric_addend (plain spgemm_numeric returns
 * STATUS_PLAN_MISMATCH, and vice versa).  beta is a HOST pointer like alpha. */
int spblas_gfx950_spgemm_set_addend(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t state, int64_t d_nnz,
                                    const int32_t* d_rowptr, const int32_t* d_colind);
int spblas_gfx950_spgemm_numeric_addend(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t state, const void* alpha,
                                        const int32_t* a_rowptr, const int32_t* a_colind, const void* a_values,
                                        const int32_t* b_rowptr, const int32_t* b_colind, const void* b_values,
                                        const void* beta, const int32_t* d_rowptr, const int32_t* d_colind,
                                        const void* d_values, int32_t* c_rowptr, int32_t* c_colind, void* c_values,
                                        int64_t c_capacity, int value_type);

/* ---- add:  C = alpha*A + beta*B  (CSR + CSR -> CSR, int32 indices) ------------------------ */
/* Device counterpart of add(a, b, c) / add_inspect / add_compute (algorithms/add_impl.hpp:40-115;
 * SURVEY.md section 8f rank 2).  Uses a spgemm state object (spblas_gfx950_spgemm_create): the
 * symbolic call is add_inspect -- it writes c_rowptr (m+1) and returns nnz(C) = structural size of
 * the union, add_impl.hpp:79-108 -- the numeric call is add_compute and may be repeated with new
 * values.  Columns of every output row are ascending (SPA + sort, add_impl.hpp:57-65).  alpha/beta
 * carry the scaled_view factors of a and b (HOST pointers); capacity < nnz(C) gives
 * STATUS_INSUFFICIENT_SPACE ("add: ran out of memory", add_impl.hpp:67-72). */
int spblas_gfx950_csr_add_symbolic(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t state, int64_t m, int64_t n,
                                   int64_t a_nnz, const int32_t* a_rowptr, const int32_t* a_colind, int64_t b_nnz,
                                   const int32_t* b_rowptr, const int32_t* b_colind, int32_t* c_rowptr,
                                   int64_t* c_nnz);
int spblas_gfx950_csr_add_numeric(spblas_gfx950_handle_t handle, spblas_gfx950_spgemm_t state, const void* alpha,
                                  const int32_t* a_rowptr, const int32_t* a_colind, const void* a_values,
                                  const void* beta, const int32_t* b_rowptr, const int32_t* b_colind,
                                  const void* b_values, int32_t* c_rowptr, int32_t* c_colind, void* c_values,
                                  int64_t c_capacity, int value_type);

/* ---- sparse triangular solve  x = inv(A) b  (CSR, int32 indices) --------------------------- */
/* Device counterpart of triangular_solve_inspect / triangular_solve
 * (algorithms/triangular_solve_impl.hpp:13-107).  As in the reference
 * loop (:57-93) only the strict triangle named by uplo and the diagonal entries are read, so a
 * general square matrix may be passed; with DIAG_UNIT stored diagonal entries are ignored.
 *   sptrsv_create  = triangular_solve_inspect: level sets of the dependency graph, built on the
 *                    device from rowptr/colind only (values may change between solves).
 *   sptrsv_solve   = triangular_solve: one launch per wide level, one single-workgroup launch per
 *                    run of narrow levels.  alpha (HOST pointer) is the scaled_view factor of A
 *                    (x = inv(alpha*A) b); b and x are device vectors of m entries (b == x is fine).
 *   sptrsv_info    : info[0] levels, [1] widest level, [2] kernel launches per solve, [3] lanes/row.
 *   sptrsv_status  : synchronises the handle's stream and reports how the LAST solve of the plan ended: 0 = complete,
 *                    1 = a device-side wait of the cooperative kernel (its grid barrier: bounded polls,
 *                    SPBLAS_GFX950_TRSV_SPIN_LIMIT) gave up -- x is not valid.  Nothing in a correct program makes
 *                    that happen; a caller that wants certainty asks once after the solves it cares about.  A caller that
 *                    never asks still hears of it: the kernel raises a pinned host word on that path and the NEXT
 *                    sptrsv_solve on the plan returns STATUS_HIP_ERROR (last_hip_error = hipErrorLaunchTimeOut) once,
 *                    before it launches anything, without synchronising the stream.  The x of a solve that no later
 *                    call follows is unverified unless sptrsv_status is asked. */
enum spblas_gfx950_uplo {
  SPBLAS_GFX950_LOWER = 0, /* lower_triangle_t  (detail/triangular_types.hpp:10-13) */
  SPBLAS_GFX950_UPPER = 1  /* upper_triangle_t  (detail/triangular_types.hpp:5-8)   */
};
enum spblas_gfx950_diag {
  SPBLAS_GFX950_DIAG_EXPLICIT = 0, /* explicit_diagonal_t       (detail/triangular_types.hpp:20-23) */
  SPBLAS_GFX950_DIAG_UNIT = 1      /* implicit_unit_diagonal_t  (detail/triangular_types.hpp:15-18) */
};
int spblas_gfx950_sptrsv_create(spblas_gfx950_handle_t handle, spblas_gfx950_trsv_t* plan, int64_t m, int64_t nnz,
                                const int32_t* rowptr, const int32_t* colind, int uplo, int diag);
int spblas_gfx950_sptrsv_destroy(spblas_gfx950_handle_t handle, spblas_gfx950_trsv_t plan);
int spblas_gfx950_sptrsv_info(spblas_gfx950_trsv_t plan, int64_t info[4]);
int spblas_gfx950_sptrsv_status(spblas_gfx950_handle_t handle, spblas_gfx950_trsv_t plan, int* status);
int spblas_gfx950_sptrsv_solve(spblas_gfx950_handle_t handle, spblas_gfx950_trsv_t plan, int64_t m, int64_t nnz,
                               const void* alpha, const int32_t* rowptr, const int32_t* colind, const void* values,
                               const void* b, void* x, int value_type);

/* ---- scale:  values[i] *= alpha  (algorithms/scale_impl.hpp:13-19) --------------------------- */
/* In-place scaling of a matrix's value array or of a dense vector (n elements, device memory).  A SLICED
 * SpMV plan holds a snapshot of the values: refresh it with spblas_gfx950_spmv_plan_update_values. */
int spblas_gfx950_scale(spblas_gfx950_handle_t handle, int64_t n, const void* alpha, void* values, int value_type);

/* ---- 64-bit indices ---------------------------------------------------------------------- */
/* The kernels of this library take int32 column (row) indices; the slot it replaces also admits int64 ones
 * (vendor/rocsparse/types.hpp:16-24: rocsparse_indextype_i64).  dst[i] = (int32) src[i] for count device elements;
 * STATUS_INVALID_VALUE when an index lies outside [0, bound) (bound <= 2^31 - 1: the number of columns), in which case
 * dst is not to be used.  Inspect-class: it synchronises the handle's stream (STATUS_NOT_SUPPORTED inside a capture).
 * The host layers call it once per index array (multiply_inspect, or the first multiply) and keep the narrowed copy. */
int spblas_gfx950_narrow_indices(spblas_gfx950_handle_t handle, int64_t count, const int64_t* src, int32_t* dst,
                                 int64_t bound);

/* ---- transpose:  B = A^T  (CSR -> CSR, int32 indices) ------------------------------------ */
/* Device counterpart of transpose(a, b) (algorithms/transpose_impl.hpp:14-53): stable counting
 * sort by column, so every output row lists its entries in source order.  t_rowptr has n+1
 * entries, t_colind / t_values nnz entries, all caller-allocated device memory.  Also what
 * multiply_inspect uses to serve csc_view / transposed(csr) operands with the regular kernels
 * (vendor/rocsparse/detail/get_transpose.hpp:19-29 maps CSC to a transposed CSR operation). */
int spblas_gfx950_csr_transpose(spblas_gfx950_handle_t handle, int64_t m, int64_t n, int64_t nnz,
                                const int32_t* rowptr, const int32_t* colind, const void* values,
                                int32_t* t_rowptr, int32_t* t_colind, void* t_values, int value_type);

#ifdef __cplusplus
}
#endif
#endif /* SPBLAS_GFX950_H */
