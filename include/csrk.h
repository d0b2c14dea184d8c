/*
 * csrk.h -- C ABI of libcsrk, the MI355X (gfx950) kernel backend for the lenskit/csr
 * `csr.kernels` hot path.
 *
 * This is the drop-in boundary: plain C, opaque integer handles, raw pointers and
 * sizes, `int` status returns (0 = CSRK_OK; on failure csrk_last_error() describes the
 * error and nothing aborts the process).  Its shape follows the reference's one native
 * interface, the MKL helper (csr/kernels/mkl/mkl_ops.h:1-31: lk_mkl_spcreate / spfree /
 * spexport / sporder / spmv / spmab / spmabt), extended by the operations the reference
 * runs outside its kernel protocol but which belong to the same hot path
 * (transpose, row extents, row normalisation).
 *
 * Each entry point names the reference interface it replaces (paths relative to the
 * reference checkout).  The Python kernel module csr_amd/kernels/hip.py binds these
 * with ctypes and implements the reference's kernel-module protocol
 * (csr/kernel.py:9-16, docs/kernels.rst:61-104) on top of them; INTEGRATION.md shows the
 * binding a reference maintainer would add.
 *
 * Memory model.  A handle owns a device-resident (HBM) copy of one CSR matrix:
 *   rowptrs[nrows+1]  int32, or int64 when ptr_is_64  (csr/csr.py:88-93)
 *   colinds[nnz]      int32                           (csr/csr.py:89)
 *   values[nnz]       float64 / float32 / absent      (csr/csr.py:94-95)
 * "host" entry points take host pointers and copy across PCIe; "_device" entry points
 * take device pointers (e.g. torch tensors' data_ptr()) and a hipStream_t passed as
 * void* (NULL = the default stream) and never synchronise the host.
 *
 * Threading: handles are immutable after creation except for the in-place operations
 * that say so; all entry points may be called concurrently from several host threads
 * (calls on the SAME handle serialise on a per-handle lock).  ctypes releases the GIL
 * around every call, matching the reference's `nogil=True` kernels.
 */
#ifndef CSRK_H
#define CSRK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSRK_API __attribute__((visibility("default")))

typedef intptr_t csrk_handle_t;      /* 0 is never a valid handle (cf. lk_mh_t, mkl_ops.h:1) */

enum {
    CSRK_OK = 0,
    CSRK_ERR_INVALID = -1,           /* bad argument / shape mismatch */
    CSRK_ERR_HIP = -2,               /* a HIP runtime call failed (no device, OOM, ...) */
    CSRK_ERR_UNSUPPORTED = -3,
    CSRK_ERR_OVERFLOW = -4           /* result does not fit the reference's int32 fields */
};

enum { CSRK_VAL_NONE = 0, CSRK_VAL_F32 = 1, CSRK_VAL_F64 = 2 };

/* SpMV algorithm selector (csrk_set_spmv_algo).  AUTO picks per matrix. */
enum {
    CSRK_SPMV_AUTO = 0,
    CSRK_SPMV_MERGE = 1,             /* merge-path tiles, LDS-staged products            */
    CSRK_SPMV_VECTOR = 2,            /* one wavefront per row segment, shfl reduction    */
    CSRK_SPMV_SCALAR = 3             /* one lane per row                                 */
};

/* ---- library / device ------------------------------------------------------------ */
CSRK_API int csrk_version(void);
/* Thread-local description of the calling thread's most recent failure. */
CSRK_API const char *csrk_last_error(void);
CSRK_API int csrk_device_count(int *count);
/* Select the HIP device used by subsequent calls from this thread (one process per
 * GPU normally passes LOCAL_RANK). */
CSRK_API int csrk_set_device(int device);
CSRK_API int csrk_synchronize(void *stream);
/* Several GPUs from one compiled host (the reference's sequential analogue is _shard_rows,
 * csr/csr.py:599-621): cut the rows into `parts` contiguous ranges balanced by entries -- bounds[g] = the first row whose
 * row pointer is >= g * nnz / parts (searchsorted(rowptrs, g * nnz / parts), the primitive of csr/csr.py:609), bounds[0] = 0,
 * bounds[parts] = nrows, never descending -- from HOST row pointers; no device is touched.  Range g then becomes a handle
 * of its own on device g (csrk_set_device(g); csrk_create with rowptrs + bounds[g] rebased by the caller, or
 * csrk_create_device on a slice already there), x is replicated, every device runs csrk_spmv_device on its range and the
 * caller's collective (RCCL all-gather of the disjoint slices, or the all-reduce north_star names) completes y: what
 * csr_amd/dist.py does with one process per GPU. */
CSRK_API int csrk_partition_rows(int32_t nrows, const void *rowptrs, int ptr_is_64, int32_t parts, int32_t *bounds);
/* Return the library's cached (free) device memory to the driver.  libcsrk keeps freed temporaries
 * and released handles' arrays in a size-bucketed pool (at most 16 GiB) to avoid hipMalloc/hipFree. */
CSRK_API int csrk_trim_cache(void);

/* ---- handles: replaces to_handle / from_handle / release_handle ---------------------
 * csr/kernels/numba/__init__.py:16-44; csr/kernels/mkl/handle.py:61-70, 95-148;
 * lk_mkl_spcreate / lk_mkl_spexport / lk_mkl_spfree (mkl_ops.h:11-14).               */

/* Copy a host CSR to the device.  `values` may be NULL iff val_type == CSRK_VAL_NONE.
 * nnz must equal rowptrs[nrows]; nnz == 0 and nrows == 0 are valid (the reference's
 * kernel fixture creates a 1x1 empty matrix, conftest.py:33-35). */
CSRK_API int csrk_create(int32_t nrows, int32_t ncols, int64_t nnz,
                         const void *rowptrs, int ptr_is_64,
                         const int32_t *colinds,
                         const void *values, int val_type,
                         csrk_handle_t *out);
/* Wrap arrays that already live in HBM.  Nothing is copied or owned: the caller keeps
 * them alive AND UNCHANGED for the handle's lifetime (same contract as to_handle,
 * docs/kernels.rst): the SpMV / SpMM plans built on the second call keep re-ordered copies of
 * colinds and values.  (libcsrk's own in-place operations -- csrk_unit_rows, csrk_center_rows,
 * csrk_order_columns -- drop those plans themselves.) */
CSRK_API int csrk_create_device(int32_t nrows, int32_t ncols, int64_t nnz,
                                const void *d_rowptrs, int ptr_is_64,
                                const int32_t *d_colinds,
                                const void *d_values, int val_type,
                                csrk_handle_t *out);
/* Idempotent on 0. */
CSRK_API int csrk_free(csrk_handle_t h);
CSRK_API int csrk_info(csrk_handle_t h, int32_t *nrows, int32_t *ncols, int64_t *nnz,
                       int *ptr_is_64, int *val_type);
/* Device memory the handle holds right now: its three arrays plus whatever SpMV / SpMM plans it has built (private
 * re-ordered streams, tables, scratch).  Builds nothing.  (The host side's handle cache budgets with it; the
 * reference's handles are host objects and have no counterpart.) */
CSRK_API int csrk_device_bytes(csrk_handle_t h, int64_t *bytes);
/* Copy the matrix back to caller-allocated host arrays sized from csrk_info
 * (any of the three may be NULL to skip it). */
CSRK_API int csrk_export(csrk_handle_t h, void *rowptrs, int32_t *colinds, void *values);
/* Device pointers of the handle's arrays (for zero-copy torch interop). */
CSRK_API int csrk_device_ptrs(csrk_handle_t h, void **d_rowptrs, void **d_colinds,
                              void **d_values);

/* ---- mult_vec: y = A x ---------------------------------------------------------------
 * csr/kernels/numba/__init__.py:55-67; lk_mkl_spmv (mkl_ops.h:29).  x has ncols float64
 * entries, y receives nrows float64 entries (every entry is written; empty rows get 0).
 * Structure-only matrices multiply with implicit 1.0 (csr/csr.py:254-262).            */
CSRK_API int csrk_spmv(csrk_handle_t h, const double *x, double *y);
/* The same with x given as float32 (host pointers).  Numba types the reference's loop by its operands
 * (csr/kernels/numba/__init__.py:55-67): float32 values times float32 x is a float32 product -- one rounding -- added to the
 * float64 accumulator; with float64 or absent values x is widened and the product is float64 (= csrk_spmv). */
CSRK_API int csrk_spmv_f32x(csrk_handle_t h, const float *x, double *y);
/* ... and with x (float32) and y (float64) already on the device; `stream` as for csrk_spmv_device: stream-ordered, nothing
 * allocated or waited for per call.  From a handle's second product on, the planned kernels widen x as they load it (the
 * copy pass, tier 0's windows, tier 1's gathers); a first product and the plan-less forms widen it once into a buffer
 * the plan keeps.  Merge algorithm only (CSRK_ERR_INVALID under `vector` / `scalar` with float32 values). */
CSRK_API int csrk_spmv_f32x_device(csrk_handle_t h, const float *d_x, double *d_y, void *stream);
CSRK_API int csrk_spmv_device(csrk_handle_t h, const double *d_x, double *d_y, void *stream);
/* The same product in two parts, for callers that ship y elsewhere while it is being completed (csr_amd/dist.py:
 * the row-partitioned multi-GPU form of csr/csr.py:584-590, where a rank's slice travels to its peers):
 *   part 1  every row of the row-major path; rows the plan cut out for its tiers (csrk_spmv_cut_rows) get 0.0
 *   part 2  the cut rows: their sums overwrite those zeros
 *   part 3  both (= csrk_spmv_device).
 * Part 1 then part 2 on one stream give bit for bit what part 3 gives. */
CSRK_API int csrk_spmv_device_part(csrk_handle_t h, const double *d_x, double *d_y, void *stream, int part);
/* The rows (ascending indices into this handle's rows) whose y entries part 2 writes: *n_rows of them, copied to the
 * device buffer d_rows if it is not NULL and holds `capacity` >= *n_rows entries.  Builds the SpMV plan if needed. */
CSRK_API int csrk_spmv_cut_rows(csrk_handle_t h, int32_t *d_rows, int64_t capacity, int64_t *n_rows);
CSRK_API int csrk_set_spmv_algo(csrk_handle_t h, int algo);
/* Name of the kernel the handle's plan resolved to, e.g. "merge" (after first use). */
CSRK_API const char *csrk_spmv_algo_name(csrk_handle_t h);
/* Launch geometry of the dominant SpMV kernel (for roofline accounting in bench.py). */
CSRK_API int csrk_spmv_plan_info(csrk_handle_t h, int64_t *n_tiles, int32_t *tile_items);

/* out[0..n) <- {0 tiles (or segments), 1 items per tile, 2 rows cut out of the tile path, 3 entries on
 * the tile path, 4 tier-0 tiles, 5 tier-0 column blocks, 6 tier-0 row threshold, 7 tier-0 block width,
 * 8 split mode (0 none, 2 panels), 9 tier-0 rows,
 * 10 tier-0 entries, 11 tier-1 rows, 12 tier-1 pairs, 13 tier-1 entries, 14 tier-1 row threshold,
 * 15 tier-1 block width, 16 columns in the hot-column pack (0: none), 17 sampled share of the row-major
 * path's entries on packed columns (ppm), 18 1 (tier 0 in accumulator form: the only one), 19 pack slots,
 * 20 short rows on the light stream (1) or on the merge-path tile kernel (0), 21 light-stream tiles,
 * 22 non-empty rows of the light stream, 23 its workgroups, 24 light-stream entries whose x values are staged
 * per call (cold staging), 25 bytes of device memory the plan holds, 26 tiles per staging round held in LDS (0: none),
 * 27 workgroups of the tier-0 accumulator kernel, 28 0 (reserved), 29..33 the plan's bytes by part: tier 0's accumulator
 * stream, tier 1's pair panel, the light stream, cold staging + pack, tables};  n <= 34. */
CSRK_API int csrk_spmv_plan_stats(csrk_handle_t h, int64_t *out, int n);

/* Kernel timing for roofline accounting: between begin and end every csrk_spmv_device call on
 * this handle brackets its streaming kernels -- [0] the tile / segment / row kernel, [1] and [2]
 * the panel kernels of tier 0 and tier 1 (merge algorithm only) -- with hipEvent pairs recorded on
 * the launch stream; nothing synchronises until csrk_spmv_profile_end, which returns the number of
 * recorded launches and mean_ms[3], their mean durations in milliseconds (0 if a kernel did not run).
 * At most `max_records` launches are recorded. */
CSRK_API int csrk_spmv_profile_begin(csrk_handle_t h, int max_records);
/* Time only every n-th csrk_spmv_device call between begin and end (default 1: all of them).  The event
 * pairs sit on the launch stream between the kernels and cost ~3 us apiece -- 20 us per SpMV with three
 * timed kernels, 3 % of the headline step; call before csrk_spmv_profile_begin. */
CSRK_API int csrk_spmv_profile_every(csrk_handle_t h, int every_n);
/* Which of the four kernels get event pairs: bit c of `mask` = channel c of csrk_spmv_profile_end4 (default 0xf: all).
 * A timed region that needs one kernel's duration pays for one event pair per timed call instead of four. */
CSRK_API int csrk_spmv_profile_channels(csrk_handle_t h, int mask);
CSRK_API int csrk_spmv_profile_end(csrk_handle_t h, int *n_records, float *mean_ms);
/* The same with mean_ms[4]: [3] = the cold-staging pass that feeds the light stream (ls_stage_kernel; 0 if the plan
 * has none). */
CSRK_API int csrk_spmv_profile_end4(csrk_handle_t h, int *n_records, float *mean_ms);

/* ---- mult_ab / mult_abt: sparse x sparse -> sparse ------------------------------------
 * csr/kernels/numba/multiply.py:13-57; lk_mkl_spmab / lk_mkl_spmabt (mkl_ops.h:30-31).
 * The product is a NEW handle owned by the caller (rowptrs int32 as in multiply.py:28,
 * values float64).  Structural zeros produced by cancellation are KEPT (the caller
 * filters them, csr/csr.py:555).  Columns inside a product row come in the reference's
 * order (reverse order of first discovery: csrk_spgemm_set_order below).
 * Both operands need values (multiply.py:115,120).  CSRK_ERR_OVERFLOW if the product
 * has more than INT32_MAX entries.                                                     */
CSRK_API int csrk_spgemm_ab(csrk_handle_t a, csrk_handle_t b, csrk_handle_t *c);
CSRK_API int csrk_spgemm_abt(csrk_handle_t a, csrk_handle_t b, csrk_handle_t *c);
/* Column order inside the rows of a product: 1 = the reference's (default) -- _sym_mm pushes a newly discovered column
 * onto the front of the row's list (csr/kernels/numba/multiply.py:79-82) and copies the list out front to back (:94-97):
 * reverse order of first discovery, so colinds and values are the reference's arrays bit for bit --, 0 = ascending (what
 * the product kernels emit; saves the ordering pass: a second walk over the products, csrc/spgemm_order.hip), -1 = follow
 * the environment variable CSRK_SPGEMM_ORDER ("ascending" selects 0; unset or anything else: 1).  Process-wide; every
 * value has the same bits either way.  csrk_spgemm_get_order: the order in force (0 or 1). */
CSRK_API int csrk_spgemm_set_order(int order);
CSRK_API int csrk_spgemm_get_order(int *order);
/* A x dense B through the reference's own entry (BASELINE configs[2]: the reference has no dense-panel call; a caller hands
 * B over as a fully populated CSR, csr/csr.py:524-567 -> multiply.py:13-38).  csrk_spgemm_ab / _abt recognise such a B on the
 * device -- every row holds all k columns 0 .. k - 1 in ascending order, so that its values ARE the row-major panel -- and
 * run the dense-panel kernels (csrk_spmm_dense's), writing C as the reference returns it: int32 row pointers with k entries
 * per row of C whose row of A holds an entry and none otherwise, columns k - 1 .. 0 (reverse order of first discovery,
 * multiply.py:79-82, 94-97), explicit zeros kept, values = the panel's sums (work[c] += a * b over the row's entries in
 * storage order, :110-122: bit for bit for rows of A of at most 64 entries, the dense-panel kernels' fixed order of
 * partial sums beyond -- within 1e-12 of sum |a b|).  A B whose rows are short of a column or in another order, and
 * float32 values on BOTH operands (float32 products, multiply.py:120), take the general product.  CSRK_SPGEMM_DENSE=0
 * switches the route off.  csrk_spgemm_last_route: what the calling thread's last product took -- 0 the general product,
 * 1 the dense-panel route. */
CSRK_API int csrk_spgemm_last_route(int *route);

/* ---- dense-panel SpMM: C = A B, B dense row-major [ncols x k] --------------------------
 * Not a reference entry point (the reference's mult_ab is sparse x sparse only); serves
 * BASELINE.json configs[2].  Equals mult_ab(A, CSR(B)) densified.  ldb/ldc in elements. */
CSRK_API int csrk_spmm_dense(csrk_handle_t a, const double *B, int32_t k, int64_t ldb,
                             double *C, int64_t ldc);
CSRK_API int csrk_spmm_dense_device(csrk_handle_t a, const double *d_B, int32_t k, int64_t ldb,
                                    double *d_C, int64_t ldc, void *stream);
/* Diagnostics: the dense-panel plan of a handle after its first csrk_spmm_dense*: out[0] = 1 when the longest rows run in
 * the register-accumulator form (csrc/spmm_dense.hip), [1] their row-length threshold, [2] their number, [3] row groups,
 * [4] column ranges, [5] their entries, [6] column tiles, [7] segments of the other rows, [8] of which partial (split rows). */
CSRK_API int csrk_spmm_plan_stats(csrk_handle_t a, int64_t *out, int n);

/* ---- transpose ------------------------------------------------------------------------
 * csr/structure.py:172-247 (_transpose_values / _transpose_structure / transpose).
 * Bit-exact with the reference's stable counting sort: output rowptrs keep the input
 * pointer width, output colinds are the source row ids in ascending source position,
 * output values are float64 whatever the input dtype (:177).  with_values == 0, or a
 * structure-only input, gives a structure-only result (:241-242).                      */
CSRK_API int csrk_transpose(csrk_handle_t h, int with_values, csrk_handle_t *out);

/* ---- COO ingest -----------------------------------------------------------------------------
 * csr/structure.py:11-67 (_from_coo_structure / _from_coo_values / from_coo), the ingest behind
 * CSR.from_coo (csr/csr.py:138-169).  Host COO arrays -> a NEW device handle.  Entries of a row
 * keep their input order (the reference's stable counting sort); values keep their dtype; row
 * pointers are int32 unless nnz > INT32_MAX.  rows[] must lie in [0, nrows), cols[] in [0, ncols). */
CSRK_API int csrk_from_coo(int32_t nrows, int32_t ncols, int64_t nnz, const int32_t *rows,
                           const int32_t *cols, const void *values, int val_type, csrk_handle_t *out);

/* ---- row extents / counts ---------------------------------------------------------------
 * csr/_rows.py:9-13 (extent), csr/csr.py:432-441 (row_nnzs = diff(rowptrs)).
 * `out` has nrows entries of the handle's pointer width (int32 or int64).              */
CSRK_API int csrk_row_nnzs(csrk_handle_t h, void *out);
CSRK_API int csrk_row_extent(csrk_handle_t h, int32_t row, int64_t *start, int64_t *end);

/* ---- row normalisation (IN PLACE on the handle's values) --------------------------------
 * csr/transform.py:29-66 (unit_rows) and :13-26 (center_rows).  `out` receives nrows
 * norms / means in the VALUES' dtype (float32 or float64), host memory.  Requires values. */
CSRK_API int csrk_unit_rows(csrk_handle_t h, void *norms);
CSRK_API int csrk_center_rows(csrk_handle_t h, void *means);
/* The same with the norms / means left in DEVICE memory (nrows entries of the values' dtype): the caller copies them out
 * when and how it likes (a pinned buffer, another stream) -- 80 MB for 10^7 rows is 3 ms of pageable copy otherwise. */
CSRK_API int csrk_unit_rows_device(csrk_handle_t h, void *d_norms);
CSRK_API int csrk_center_rows_device(csrk_handle_t h, void *d_means);

/* ---- order_columns (IN PLACE) -------------------------------------------------------------
 * csr/kernels/numba/__init__.py:47-52 -> csr/structure.py:156-169; lk_mkl_sporder.
 * Stable sort of every row by column index; values follow.                              */
CSRK_API int csrk_order_columns(csrk_handle_t h);

/* ---- pick_rows ---------------------------------------------------------------------------
 * csr/csr.py:347-364 -> csr/structure.py:84-149 (_pick_rows, _pick_rows_nvs).  NEW handle with the
 * rows rows[0..n_rows) of h, in that order (a row may appear more than once); `rows` is a host
 * array.  with_values = 0 drops the values; otherwise they keep their dtype.  Row pointers are
 * int32 (as the reference's) unless the result has more than 2^31 - 1 entries.  An index outside
 * [0, nrows) is CSRK_ERR_INVALID (the reference raises IndexError).                        */
CSRK_API int csrk_pick_rows(csrk_handle_t h, const int32_t *rows, int64_t n_rows, int with_values,
                            csrk_handle_t *out);

/* ---- _filter_zeros ----------------------------------------------------------------------
 * csr/_struct.py:61-76.  Returns a NEW handle without the entries whose value is
 * exactly 0.0 (NaN is kept).  Requires float64 values.                                  */
CSRK_API int csrk_filter_zeros(csrk_handle_t h, csrk_handle_t *out);

#ifdef __cplusplus
}
#endif
#endif /* CSRK_H */
