// SpMV/SpMM plan: the product of multiply_inspect (device-side analysis).
#pragma once

#include "common.hpp"

struct spblas_gfx950_plan_s {
  // identity of the matrix the plan was built for
  int64_t m = 0, n = 0, nnz = 0;
  const void* rowptr = nullptr;
  const int32_t* colind = nullptr;
  int offset_type = 0, value_type = 0;

  int alg = SPBLAS_GFX950_SPMV_VECTOR;
  int vector_lpr = 8;  // lanes per row for the plan-free kernel

  // ROWBLOCK: nnz windows of `win` entries; win_row[w] = first row whose start
  // offset is >= w*win (w = 0..nwin), win_row[nwin] = m.
  int win = 0;
  int win_req = 0;  // > 0: window length asked for by the builder (the hot-column part of a split plan: 256)
  int64_t nwin = 0;
  int32_t* win_row = nullptr;
  // rows longer than `win` are split across the windows they cover
  int64_t n_long = 0;
  int32_t* long_rows = nullptr;
  void* part_head = nullptr;  // T[nwin]: partial of the long row entering window w
  void* part_tail = nullptr;  // T[nwin]: partial of the long row starting in window w

  // statistics
  int64_t max_row_len = 0;
  int64_t empty_rows = 0;
  int64_t long_nnz = 0;  // entries stored in rows longer than the window

  // SLICED: column-sliced re-tiling of A (see spmv_sliced.hip).  A run = the entries of one (slice, wave-bin)
  // tile, padded to whole blocks of 32 entries.  A' order = blocks by (slice, bin): what the expand streams;
  // P order = the same blocks by (bin, slice), every bin padded to whole groups of 8 blocks: what the reduce streams.
  int n_slices = 0;      // S column slices of slice_cols columns
  int slice_cols = 0;
  int rows_per_blk = 0;  // H rows per wave-bin
  int64_t n_rblk = 0;    // NB wave-bins
  void* seg_ptr = nullptr;     // int32[S*NB]: entries per run, key = s*NB + b (inspect only; kept for introspection)
  void* s_sliceblk = nullptr;  // int32[S + 1]: first A'-order block of every slice
  void* s_binblk = nullptr;    // int32[NB + 1]: first P-order block of every wave-bin (multiples of 8)
  void* s_blkdst = nullptr;    // int32[a_blocks]: P-order block of every A'-order block
  void* s_eoff = nullptr;      // int2[NB*S]: (first entry, padded length) of every (slice, bin) run in the compact A' arrays,
                               // BIN-major: key = b*S + s (end of round 4: the value refresh walks the runs bin by bin)
  void* s_blksrc = nullptr;    // int32[a_blocks]: first entry of the block in the COMPACT A' arrays (round 3: runs occupy
                               // their count rounded up to 4 entries there, not whole blocks)
  int64_t a_entries = 0;       // entries the A' arrays hold (compact stream + one block of slack)
  void* s_colind = nullptr;    // uint16[a_entries] column inside the slice (A' order, compact; pads 0)
  void* s_values = nullptr;    // T[a_entries] (A' order, compact; pads 0)
  void* s_perm = nullptr;      // int32[a_entries] source position in the caller's CSR arrays (pads -1)
  uint16_t* s_lrow = nullptr;  // uint16[p_blocks*32] row inside the bin | bit 15 = duplicate flag (P order; pads = H)
  // one-byte row codes instead of s_lrow (spmv_sliced.hip, "enc8"): runs sorted by row, the row advance per entry
  int enc8 = 0;
  unsigned char* s_code = nullptr;  // uint8[p_blocks*BLK]: advance since the previous entry of the block; 255 = left out
  void* s_hdr = nullptr;            // per P block: base row + duplicate flags (8 B per 32 fp32 / 4 B per 16 fp64 entries)
  unsigned* s_exc_idx = nullptr;    // [NB * exc_cap] P index of the entries the codes cannot reach ...
  uint16_t* s_exc_row = nullptr;    // [NB * exc_cap] ... and their row inside the bin
  int32_t* s_exc_cnt = nullptr;     // [NB] exceptions per wave-bin; [NB] = "encoding failed" flag of the build
  int exc_cap = 0;
  void* s_products = nullptr;  // T[p_blocks*32] workspace: expanded products (P order)
  size_t s_products_bytes = 0;
  int64_t a_blocks = 0, p_blocks = 0;
  int n_ksplit = 1;            // reduce workgroups per bin group (every wave-bin's stream cut into K parts)
  int rwaves = 4;              // reduce: wave-bins (= wavefronts) per workgroup
  int hub_len = 0;             // > 0: rows longer than this are NOT in the tiles (pb_hub_rows_kernel does them)
  int64_t s_placed = 0;        // entries in the tiles (nnz minus the hub rows' entries)
  const void* values_ptr = nullptr;  // caller's values array the copy was taken from (inspect / last update)
  int keep_src = 0;                  // 1: keep the source position of every entry (s_perm / hot_src / rest_src): value refreshes
  size_t base_device_bytes = 0;      // device_bytes of the structures plan_build made (before any SLICED arrays)
  int refresh_each_call = 0;         // 1: the plan was made WITHOUT the snapshot opt-in: every multiply takes the values again
  // Value-free tiles (round 5, spmv_sliced.hip "vfree"): the plan holds NO copy of A's values.  The expand writes the gathered
  // x[col] to the product stream; the reduce of a bin stages the bin's window of the CALLER's values (contiguous:
  // rowptr[r0] .. rowptr[r1]) in LDS and forms values[src] * xg with a 16-bit in-window source offset per entry.
  int vfree = 0;
  int vf_win_cap = 0;                // entries the LDS window area of the reduce holds
  int vf_waves = 0;                  // wavefronts of a reduce workgroup (one bin per WORKGROUP; private accumulators per wave)
  uint16_t* s_src = nullptr;         // uint16[p_blocks*BLK] source position inside the bin's window (P order; pads 0)
  const void* last_x = nullptr;      // x of the last expand
  void* s_hub_part = nullptr;        // T[n_long * hub_parts] partial sums of the hub rows
  int hub_parts = 1;                 // workgroups per hub row
  void* s_xitems = nullptr;          // int4[n_xitems] expand work list (slice, first block, last block) for column-skewed matrices
  int64_t n_xitems = 0;
  // reduce work list for row-skewed matrices: int4 (group, part, parts, partial offset or -1),
  // the split groups (group, K, offset of the first partial block) and the partial blocks themselves
  void *s_ritems = nullptr, *s_rsplit = nullptr, *s_rpartial = nullptr;
  int64_t n_ritems = 0, n_rsplit = 0;
  void* s_partial = nullptr;   // T[s_partial_k][m] partial sums (grown on demand)
  int s_partial_k = 0;
  int bin_aligned = 0;         // wave-bin height divides the handle's bin_row_align option
  // row-skewed matrices: wave-bins of variable height (<= rows_per_blk rows, ~equal entries); nullptr = bin b
  // holds the rows [b * rows_per_blk, (b + 1) * rows_per_blk)
  void* s_binrow = nullptr;    // int32[NB + 1] first row of every wave-bin (device)
  int32_t* h_binrow = nullptr; // host copy (malloc; fetched on first partial-range reduce)
  // matrices with many empty rows (graphs): the tiles are built over the NON-EMPTY rows only (same colind / values, a
  // compacted row pointer array); s_m = rows of that compacted matrix, nzrow[i] = original row of compact row i,
  // zrow[] = the empty rows (y = beta * y there)
  int64_t s_m = 0;
  void* s_rowptr_c = nullptr;  // O[s_m + 1]
  void* s_nzrow = nullptr;     // int32[s_m]; nullptr = no compaction (s_m == m)
  void* s_zrow = nullptr;      // int32[n_zero]
  int64_t n_zero = 0;
  int32_t* h_binrow_orig = nullptr;  // host: ORIGINAL first row of every wave-bin (compaction; fetched lazily)
  // rows longer than split_len entries are cut into PIECES of split_len entries, each a row of the compacted matrix
  // (bit 31 of its nzrow entry set): no tile then sees one row hundreds of times.  A piece's sum goes to
  // piece_out[compact row]; pb_split_finish_kernel adds the pieces of a row in order.
  int split_len = 0;
  void* s_piece_out = nullptr;   // T[s_m]
  void* s_split_rows = nullptr;  // int4[n_split]: (row of y, first compact row, pieces, 0)
  int64_t n_split = 0;
  void* s_hub_rows = nullptr;  // int32[n_hub] rows kept out of the tiles (== long_rows unless variable bins raise the threshold)
  int64_t n_hub = 0;
  bool hub_rows_owned = false;
  // the stream of the last launch that used the plan's workspaces: plan_destroy orders its frees behind it.  (A plan
  // carries mutable workspaces -- products, partial sums, long-row partials -- and must not run on two streams at once.)
  hipStream_t last_stream = nullptr;
  bool used = false;
  // spblas_gfx950_spmv_plan_detach: the plan is self-contained (SLICED snapshot, no hub rows, no hot split) and its owner has
  // released the structure / value arrays it was built from -- multiplies take no array arguments, the values cannot be refreshed
  int detached = 0;
  // AUTO: the static rules could not tell which plan is faster (hot columns / heavy rows): plan_create times both
  int s_uncertain = 0;
  float trial_ms[2] = {0.f, 0.f};  // {row-block, sliced} when the trial ran
  int nt_products = 0;             // SLICED: the expand stores its products with the non-temporal hint (spmv.hip: store_trial)
  float store_trial_ms[2] = {0.f, 0.f};  // {plain, non-temporal} when THIS plan ran the per-device store trial

  // SpMM inspect (spblas_gfx950_spmm_inspect, spmm.hip): row blocks of 32 rows whose entries fall into at most 16
  // aligned tiles of 128 columns, densely enough, are multiplied from LDS-staged B tiles on the matrix cores
  int mm_ready = 0;
  int64_t mm_nblk = 0;              // row blocks of 32 rows
  int64_t mm_npanel = 0;            // blocks taken by the panel kernel
  int64_t mm_panel_nnz = 0;         // entries inside those blocks
  unsigned char* mm_is_panel = nullptr;  // [mm_nblk] 1 = the panel kernel owns the block (the row kernel skips it)
  int32_t* mm_panel_blocks = nullptr;    // [mm_npanel] block ids
  int32_t* mm_tiles = nullptr;           // [mm_nblk * 9]: count + up to 8 ascending column-tile ids per block
  void* mm_win = nullptr;                // int2[mm_nblk]: smallest / largest column of every qualifying block (spmm_band_kernel)
  int32_t* mm_vec_blocks = nullptr;      // band path: the qualifying blocks for the entry-loop kernel, ascending
  int32_t* mm_dense_blocks = nullptr;    // ... and those dense in their window (spmm_band_mfma_kernel)
  int64_t mm_ndense = 0;
  int mm_band = 1;                       // qualifying blocks go to the band kernels (0: the tile kernel of rounds 3 - 5)
  int mm_band_ch = 128, mm_band_waves = 8;  // B rows staged per chunk, wavefronts per workgroup
  int mm_band_dense_pm = 1001;           // window density (per mille) from which a block's contraction runs on the matrix cores
  void* mm_long_part = nullptr;          // T[n_long * mm_long_parts * n] partial rows of the long rows (grown on demand)
  int64_t mm_long_cap = 0;               // elements held by mm_long_part
  int mm_long_parts = 1;

  // Hot-column split (spmv_hot.hip): the entries in the K most referenced columns live in A_hot (row order, 16-bit index
  // into hot_cols[], over the rows that have such entries) and are multiplied straight out of LDS; the others form the CSR
  // matrix A_rest, which carries the tiled plan.  Both are plans of their own (is_child: never split again, never handed out).
  spblas_gfx950_plan_s* hot_plan = nullptr;   // window partition / long rows / partials of A_hot (ROWBLOCK structures)
  spblas_gfx950_plan_s* rest_plan = nullptr;  // SLICED plan of A_rest
  int is_child = 0;              // > 0: the remainder of that many splits; < 0: the window structures of a hot part
  int hot_k = 0;                 // hot columns
  int64_t hot_nnz = 0, hot_m = 0;  // entries / rows of A_hot
  int32_t* hot_cols = nullptr;   // [hot_k] their numbers, ascending
  uint16_t* hot_col = nullptr;   // [hot_nnz] index into hot_cols
  void* hot_val = nullptr;       // T[hot_nnz]
  int32_t* hot_src = nullptr;    // [hot_nnz] position in the caller's arrays
  void* hot_rowptr = nullptr;    // O[hot_m + 1]
  int32_t* hot_rows = nullptr;   // [hot_m] row of y
  void* hot_part = nullptr;      // T[2 * nwin]: per window of A_hot the piece of the row that runs in / the row that runs on
  int32_t* hot_cross = nullptr;  // [hot_ncross] rows of A_hot that lie in more than one window
  int64_t hot_ncross = 0;
  void* rest_rowptr = nullptr;   // O[m + 1]
  int32_t* rest_col = nullptr;   // [nnz - hot_nnz]
  void* rest_val = nullptr;      // T[nnz - hot_nnz]
  int32_t* rest_src = nullptr;   // [nnz - hot_nnz] position in the caller's arrays

  size_t device_bytes = 0;
};
