/*
 * csr_oracle.c -- CPU restatement of the lenskit/csr `csr.kernels` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load it.  The
 * product (csr_amd/, libcsrk.so) never links, imports or calls anything here.
 *
 * Every function restates, in plain sequential C, the loop the reference runs under
 * Numba (single thread, `@njit(nogil=True)`, no `parallel=True`), citing the reference
 * file:line it follows (paths relative to the reference checkout).  Parity is PINNED:
 * tests/test_oracle_golden.py checks each function against golden vectors captured by
 * oracle/gen/gen_golden.py from the reference's own code (imported in its
 * NUMBA_DISABLE_JIT mode) and against the reference tests' fixed known-answer cases.
 *
 * Conventions: row pointers are int64 here (the Python wrapper widens int32 ones; an
 * int32 entry point exists for mult_vec because it is the timed CPU baseline);
 * column indices int32; values double unless the name says f32; `values == NULL`
 * means a structure-only matrix whose entries are implicitly 1.0
 * (csr/csr.py:254-262 `_e_value`).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EXPORT __attribute__((visibility("default")))

ORC_EXPORT int orc_version(void) { return 1; }

ORC_EXPORT void orc_free(void *p) { free(p); }

/* ---------------------------------------------------------------------------------
 * mult_vec: csr/kernels/numba/__init__.py:55-67.
 * One pass over the nnz entries with a moving row cursor; the cursor skips empty rows
 * (:62-63); y is a fresh zero vector (:57) accumulated in float64 in storage order.
 * ------------------------------------------------------------------------------- */
ORC_EXPORT void orc_mult_vec_i64(int32_t nrows, int64_t nnz, const int64_t *rowptrs,
                                 const int32_t *colinds, const double *values,
                                 const double *x, double *y)
{
    memset(y, 0, sizeof(double) * (size_t)nrows);
    int64_t row = 0;
    for (int64_t i = 0; i < nnz; i++) {
        while (i == rowptrs[row + 1])
            row++;
        double a = values ? values[i] : 1.0;
        y[row] += x[colinds[i]] * a;
    }
}

ORC_EXPORT void orc_mult_vec_i32(int32_t nrows, int64_t nnz, const int32_t *rowptrs,
                                 const int32_t *colinds, const double *values,
                                 const double *x, double *y)
{
    memset(y, 0, sizeof(double) * (size_t)nrows);
    int64_t row = 0;
    for (int64_t i = 0; i < nnz; i++) {
        while (i == (int64_t)rowptrs[row + 1])
            row++;
        double a = values ? values[i] : 1.0;
        y[row] += x[colinds[i]] * a;
    }
}

/* NOT the reference: the same per-row arithmetic (products rounded on their own, added in storage order, so the
 * result is bit-identical to orc_mult_vec_i32) with the rows shared out over OpenMP threads.  The reference's
 * kernel is single-threaded (@njit(nogil=True), no parallel=True: csr/kernels/numba/__init__.py:55); bench.py
 * reports this variant next to the port so that the GPU/CPU ratio is not flattered by a one-core baseline
 * (SURVEY.md section 8d). */
ORC_EXPORT void orc_mult_vec_i32_rows_omp(int32_t nrows, const int32_t *rowptrs, const int32_t *colinds,
                                          const double *values, const double *x, double *y, int32_t nthreads)
{
#pragma omp parallel for schedule(dynamic, 2048) num_threads(nthreads)
    for (int32_t r = 0; r < nrows; r++) {
        double acc = 0.0;
        for (int64_t i = rowptrs[r]; i < (int64_t)rowptrs[r + 1]; i++) {
            double a = values ? values[i] : 1.0;
            acc += x[colinds[i]] * a;
        }
        y[r] = acc;
    }
}

/* float32 values: the product v[col] * values[i] is taken in float64 when v is float64
 * (NumPy/Numba promotion), which is what every reference test feeds (test_utils.py:22-27). */
ORC_EXPORT void orc_mult_vec_f32vals(int32_t nrows, int64_t nnz, const int64_t *rowptrs,
                                     const int32_t *colinds, const float *values,
                                     const double *x, double *y)
{
    memset(y, 0, sizeof(double) * (size_t)nrows);
    int64_t row = 0;
    for (int64_t i = 0; i < nnz; i++) {
        while (i == rowptrs[row + 1])
            row++;
        y[row] += x[colinds[i]] * (double)values[i];
    }
}

/* float32 values AND float32 x: NumPy/Numba take the product in float32 (one rounding)
 * before it is added to the float64 accumulator. */
ORC_EXPORT void orc_mult_vec_f32f32(int32_t nrows, int64_t nnz, const int64_t *rowptrs,
                                    const int32_t *colinds, const float *values,
                                    const float *x, double *y)
{
    memset(y, 0, sizeof(double) * (size_t)nrows);
    int64_t row = 0;
    for (int64_t i = 0; i < nnz; i++) {
        while (i == rowptrs[row + 1])
            row++;
        float p = x[colinds[i]] * values[i];
        y[row] += (double)p;
    }
}

/* ---------------------------------------------------------------------------------
 * mult_ab: csr/kernels/numba/multiply.py:13-38 (driver), :60-100 (_sym_mm),
 * :103-129 (_num_mm).  Bank-Douglas SMMP.
 *
 * Symbolic pass: `index` is a marker array that doubles as a linked list; a newly seen
 * column k is pushed on the FRONT of the list (:79-82), so a row's columns come out in
 * reverse discovery order when the list is walked (:94-97).  c_ci starts at
 * max(A.nnz, B.nnz) entries and grows by half when short (:85-90).  c_rp is int32 (:28).
 * Numeric pass: dense work row, `work[k] += a*b` then gather+reset along c_ci (:110-127).
 * Explicit zeros are kept (they are filtered by the caller, csr/csr.py:555).
 *
 * Returns the product nnz (or -1 on allocation failure); *c_ci_out / *c_vs_out are
 * malloc'd and must be released with orc_free.
 * ------------------------------------------------------------------------------- */
ORC_EXPORT int64_t orc_sym_mm(int32_t a_nrows, int32_t a_ncols, int64_t a_nnz,
                              const int64_t *a_rp, const int32_t *a_ci,
                              int32_t b_ncols, int64_t b_nnz,
                              const int64_t *b_rp, const int32_t *b_ci,
                              int32_t *c_rp, int32_t **c_ci_out)
{
    int64_t wlen = a_nrows;
    if (a_ncols > wlen) wlen = a_ncols;
    if (b_ncols > wlen) wlen = b_ncols;
    int32_t *index = (int32_t *)malloc(sizeof(int32_t) * (size_t)(wlen > 0 ? wlen : 1));
    int64_t c_len = a_nnz > b_nnz ? a_nnz : b_nnz;
    int32_t *c_ci = (int32_t *)calloc((size_t)(c_len > 0 ? c_len : 1), sizeof(int32_t));
    if (!index || !c_ci) { free(index); free(c_ci); return -1; }
    for (int64_t k = 0; k < wlen; k++) index[k] = -1;
    int64_t c_pos = 0;
    c_rp[0] = 0;

    for (int32_t i = 0; i < a_nrows; i++) {
        int64_t istart = wlen;      /* list terminator (:64) */
        int64_t length = 0;
        for (int64_t jj = a_rp[i]; jj < a_rp[i + 1]; jj++) {
            int32_t j = a_ci[jj];
            for (int64_t kk = b_rp[j]; kk < b_rp[j + 1]; kk++) {
                int32_t k = b_ci[kk];
                if (index[k] < 0) {
                    index[k] = (int32_t)istart;
                    istart = k;
                    length++;
                }
            }
        }
        while (c_pos + length > c_len) {
            int64_t grown = c_len + c_len / 2;
            if (grown <= c_len) grown = c_len + 1;  /* c_len 0/1 cannot grow by half */
            int32_t *c2 = (int32_t *)realloc(c_ci, sizeof(int32_t) * (size_t)grown);
            if (!c2) { free(index); free(c_ci); return -1; }
            c_ci = c2;
            c_len = grown;
        }
        c_rp[i + 1] = (int32_t)(c_rp[i] + length);
        for (int64_t j = c_rp[i]; j < c_rp[i + 1]; j++) {
            c_ci[j] = (int32_t)istart;
            istart = index[istart];
            index[c_ci[j]] = -1;
        }
        c_pos += length;
    }
    free(index);
    *c_ci_out = c_ci;
    return c_pos;
}

ORC_EXPORT int orc_num_mm(int32_t a_nrows, int32_t a_ncols,
                          const int64_t *a_rp, const int32_t *a_ci, const double *a_vs,
                          int32_t b_ncols,
                          const int64_t *b_rp, const int32_t *b_ci, const double *b_vs,
                          const int32_t *c_rp, const int32_t *c_ci, double *c_vs, int round32)
{
    /* round32: both operands' values are float32 (widened exactly by the caller): `av * b_h.values[kk]`
     * (multiply.py:120) is then a float32 product -- one rounding -- added to the float64 work array */
    int64_t wlen = a_nrows;
    if (a_ncols > wlen) wlen = a_ncols;
    if (b_ncols > wlen) wlen = b_ncols;
    double *work = (double *)calloc((size_t)(wlen > 0 ? wlen : 1), sizeof(double));
    if (!work) return -1;
    for (int32_t i = 0; i < a_nrows; i++) {
        for (int64_t jj = a_rp[i]; jj < a_rp[i + 1]; jj++) {
            int32_t j = a_ci[jj];
            double av = a_vs[jj];
            for (int64_t kk = b_rp[j]; kk < b_rp[j + 1]; kk++)
                work[b_ci[kk]] += round32 ? (double)((float)av * (float)b_vs[kk]) : av * b_vs[kk];
        }
        for (int32_t jj = c_rp[i]; jj < c_rp[i + 1]; jj++) {
            int32_t j = c_ci[jj];
            c_vs[jj] = work[j];
            work[j] = 0.0;
        }
    }
    free(work);
    return 0;
}

ORC_EXPORT int64_t orc_mult_ab(int32_t a_nrows, int32_t a_ncols, int64_t a_nnz,
                               const int64_t *a_rp, const int32_t *a_ci, const double *a_vs,
                               int32_t b_ncols, int64_t b_nnz,
                               const int64_t *b_rp, const int32_t *b_ci, const double *b_vs,
                               int32_t *c_rp, int32_t **c_ci_out, double **c_vs_out, int round32)
{
    int32_t *c_ci = NULL;
    int64_t c_nnz = orc_sym_mm(a_nrows, a_ncols, a_nnz, a_rp, a_ci, b_ncols, b_nnz, b_rp, b_ci,
                               c_rp, &c_ci);
    if (c_nnz < 0) return -1;
    double *c_vs = (double *)calloc((size_t)(c_nnz > 0 ? c_nnz : 1), sizeof(double));
    if (!c_vs) { free(c_ci); return -1; }
    if (orc_num_mm(a_nrows, a_ncols, a_rp, a_ci, a_vs, b_ncols, b_rp, b_ci, b_vs,
                   c_rp, c_ci, c_vs, round32) != 0) {
        free(c_ci); free(c_vs); return -1;
    }
    *c_ci_out = c_ci;
    *c_vs_out = c_vs;
    return c_nnz;
}

/* ---------------------------------------------------------------------------------
 * transpose: csr/structure.py:172-204 (_transpose_values), :207-237
 * (_transpose_structure).  Counting sort by column: histogram into brp[col+1]
 * (:180-184), running sum (:187-188), scatter in row-major order using brp[col] as the
 * cursor (:191-197) -- hence stable: inside an output row the entries keep source
 * order -- and finally shift the cursors back by one slot (:200-202).
 * Output values are always float64 (:177); bvs may be NULL for structure-only.
 * `vs_f32` selects float32 input values (they are widened on copy).
 * ------------------------------------------------------------------------------- */
ORC_EXPORT void orc_transpose(int32_t nrows, int32_t ncols, int64_t nnz,
                              const int64_t *rp, const int32_t *ci,
                              const void *vs, int vs_f32,
                              int64_t *brp, int32_t *bci, double *bvs)
{
    (void)nnz;
    for (int64_t j = 0; j <= ncols; j++) brp[j] = 0;
    for (int32_t i = 0; i < nrows; i++)
        for (int64_t jj = rp[i]; jj < rp[i + 1]; jj++)
            brp[ci[jj] + 1] += 1;
    for (int32_t j = 0; j < ncols; j++)
        brp[j + 1] = brp[j] + brp[j + 1];
    for (int32_t i = 0; i < nrows; i++) {
        for (int64_t jj = rp[i]; jj < rp[i + 1]; jj++) {
            int32_t j = ci[jj];
            int64_t pos = brp[j];
            bci[pos] = i;
            if (bvs)
                bvs[pos] = vs_f32 ? (double)((const float *)vs)[jj] : ((const double *)vs)[jj];
            brp[j] = pos + 1;
        }
    }
    for (int32_t j = ncols - 1; j > 0; j--)
        brp[j] = brp[j - 1];
    if (ncols >= 0) brp[0] = 0;
}

/* ---------------------------------------------------------------------------------
 * unit_rows: csr/transform.py:29-66.  In place, per non-empty row: vmax = max|v| (:52);
 * (m, e) = frexp(vmax) (:55); pnexp = clamp(-e, minexp, maxexp-1) (:56-57);
 * prenorm = 2^pnexp (:58); v *= prenorm (:59); inorm = ||v||_2 (:62);
 * norms[i] = inorm / prenorm (:63); v /= inorm (:64).  An all-zero row gives norm 0 and
 * NaN values; an empty row gives norm 0 (:36-38).  finfo: f64 maxexp 1024, minexp -1022;
 * f32 maxexp 128, minexp -126.  Arithmetic is done in the values' dtype.
 * ------------------------------------------------------------------------------- */
ORC_EXPORT void orc_unit_rows_f64(int32_t nrows, const int64_t *rp, double *vs, double *norms)
{
    for (int32_t i = 0; i < nrows; i++) {
        int64_t sp = rp[i], ep = rp[i + 1];
        norms[i] = 0.0;
        if (sp == ep) continue;
        double vmax = 0.0;
        int has_nan = 0;
        for (int64_t k = sp; k < ep; k++) {
            double a = fabs(vs[k]);
            if (a != a) has_nan = 1;
            if (a > vmax) vmax = a;
        }
        if (has_nan) vmax = NAN;   /* np.max propagates NaN */
        int ve = 0;
        (void)frexp(vmax, &ve);
        if (vmax != vmax || isinf(vmax)) ve = 0;   /* math.frexp(nan|inf) -> (x, 0) */
        int pnexp = -ve;
        if (pnexp > 1024 - 1) pnexp = 1024 - 1;
        if (pnexp < -1022) pnexp = -1022;
        double prenorm = ldexp(1.0, pnexp);
        double ss = 0.0;
        for (int64_t k = sp; k < ep; k++) {
            vs[k] *= prenorm;
            ss += vs[k] * vs[k];
        }
        double inorm = sqrt(ss);
        norms[i] = inorm / prenorm;
        for (int64_t k = sp; k < ep; k++)
            vs[k] /= inorm;
    }
}

ORC_EXPORT void orc_unit_rows_f32(int32_t nrows, const int64_t *rp, float *vs, float *norms)
{
    for (int32_t i = 0; i < nrows; i++) {
        int64_t sp = rp[i], ep = rp[i + 1];
        norms[i] = 0.0f;
        if (sp == ep) continue;
        float vmax = 0.0f;
        int has_nan = 0;
        for (int64_t k = sp; k < ep; k++) {
            float a = fabsf(vs[k]);
            if (a != a) has_nan = 1;
            if (a > vmax) vmax = a;
        }
        if (has_nan) vmax = NAN;
        int ve = 0;
        (void)frexp((double)vmax, &ve);
        if (vmax != vmax || isinf(vmax)) ve = 0;
        int pnexp = -ve;
        if (pnexp > 128 - 1) pnexp = 128 - 1;
        if (pnexp < -126) pnexp = -126;
        float prenorm = (float)ldexp(1.0, pnexp);
        float ss = 0.0f;
        for (int64_t k = sp; k < ep; k++) {
            vs[k] *= prenorm;
            ss += vs[k] * vs[k];
        }
        float inorm = sqrtf(ss);
        norms[i] = inorm / prenorm;
        for (int64_t k = sp; k < ep; k++)
            vs[k] /= inorm;
    }
}

/* ---------------------------------------------------------------------------------
 * center_rows: csr/transform.py:13-26.  Per non-empty row: m = mean(v); means[i] = m;
 * v -= m.  Empty rows keep mean 0.
 * ------------------------------------------------------------------------------- */
ORC_EXPORT void orc_center_rows_f64(int32_t nrows, const int64_t *rp, double *vs, double *means)
{
    for (int32_t i = 0; i < nrows; i++) {
        int64_t sp = rp[i], ep = rp[i + 1];
        means[i] = 0.0;
        if (sp == ep) continue;
        double s = 0.0;
        for (int64_t k = sp; k < ep; k++) s += vs[k];
        double m = s / (double)(ep - sp);
        means[i] = m;
        for (int64_t k = sp; k < ep; k++) vs[k] -= m;
    }
}

ORC_EXPORT void orc_center_rows_f32(int32_t nrows, const int64_t *rp, float *vs, float *means)
{
    for (int32_t i = 0; i < nrows; i++) {
        int64_t sp = rp[i], ep = rp[i + 1];
        means[i] = 0.0f;
        if (sp == ep) continue;
        float s = 0.0f;
        for (int64_t k = sp; k < ep; k++) s += vs[k];
        float m = s / (float)(ep - sp);
        means[i] = m;
        for (int64_t k = sp; k < ep; k++) vs[k] -= m;
    }
}

/* ---------------------------------------------------------------------------------
 * _filter_zeros: csr/_struct.py:61-76.  In-place forward compaction of entries whose
 * value is exactly 0.0 (NaN is kept: NaN != 0).  Rewrites rp; returns the new nnz.
 * ------------------------------------------------------------------------------- */
ORC_EXPORT int64_t orc_filter_zeros(int32_t nrows, int64_t *rp, int32_t *ci, double *vs)
{
    int64_t nnz = 0;
    for (int32_t i = 0; i < nrows; i++) {
        int64_t sp = rp[i], ep = rp[i + 1];
        rp[i] = nnz;
        for (int64_t jp = sp; jp < ep; jp++) {
            if (vs[jp] != 0) {
                ci[nnz] = ci[jp];
                vs[nnz] = vs[jp];
                nnz++;
            }
        }
        /* rp[i+1] is still the original pointer here: only rp[0..i] were rewritten. */
    }
    rp[nrows] = nnz;
    return nnz;
}

/* ---------------------------------------------------------------------------------
 * sort_rows / order_columns: csr/structure.py:156-169 (bubble sort of each row by
 * column, values follow).  Swapping only strictly out-of-order neighbours makes it a
 * STABLE sort, so the result is the unique stable ordering; restated as an insertion
 * sort, which produces that same ordering.  vs may be NULL.
 * ------------------------------------------------------------------------------- */
ORC_EXPORT void orc_sort_rows(int32_t nrows, const int64_t *rp, int32_t *ci, double *vs)
{
    for (int32_t i = 0; i < nrows; i++) {
        for (int64_t a = rp[i] + 1; a < rp[i + 1]; a++) {
            int32_t c = ci[a];
            double v = vs ? vs[a] : 0.0;
            int64_t b = a - 1;
            while (b >= rp[i] && ci[b] > c) {
                ci[b + 1] = ci[b];
                if (vs) vs[b + 1] = vs[b];
                b--;
            }
            ci[b + 1] = c;
            if (vs) vs[b + 1] = v;
        }
    }
}

/* ---------------------------------------------------------------------------------
 * Dense-panel SpMM, C = A * B with B dense row-major [a_ncols x k].  The reference has
 * no dense-B entry point; this is the reference's numeric
 * recurrence (multiply.py:110-122: for each A entry (i,j,av): C[i,:] += av * B[j,:]) with
 * B taken as a fully populated matrix, and is what `mult_ab(A, CSR(B))` computes.
 * ------------------------------------------------------------------------------- */
ORC_EXPORT void orc_spmm_dense(int32_t nrows, const int64_t *rp, const int32_t *ci,
                               const double *vs, const double *B, int32_t k, int64_t ldb,
                               double *C, int64_t ldc)
{
    for (int32_t i = 0; i < nrows; i++) {
        double *c = C + (int64_t)i * ldc;
        for (int32_t t = 0; t < k; t++) c[t] = 0.0;
        for (int64_t jj = rp[i]; jj < rp[i + 1]; jj++) {
            double av = vs ? vs[jj] : 1.0;
            const double *b = B + (int64_t)ci[jj] * ldb;
            for (int32_t t = 0; t < k; t++) c[t] += av * b[t];
        }
    }
}

/* ---------------------------------------------------------------------------------
 * from_coo: csr/structure.py:11-32 (_from_coo_structure), :35-58 (_from_coo_values), the
 * ingest behind CSR.from_coo (csr/csr.py:138-169).  Counting sort by row: histogram
 * (:15-17), running sum into rowptrs (:19-21), then one pass over the entries IN INPUT
 * ORDER with a per-row cursor (:23-30) -- so the entries of a row keep their input order
 * (stable), duplicates included.  Values are copied bit for bit whatever their element
 * size `vsize` (vs == NULL: structure only).  rowptrs come out int64 here, as in the
 * reference (:19); the CSR constructor then narrows them to int32 when nnz fits
 * (csr/csr.py:88-93) -- the Python wrapper does the same.
 * ------------------------------------------------------------------------------- */
ORC_EXPORT int orc_from_coo(int32_t nrows, int64_t nnz, const int32_t *rows, const int32_t *cols, const void *vs,
                            int32_t vsize, int64_t *rowptrs, int32_t *out_cols, void *out_vs)
{
    int64_t *rpos = (int64_t *)calloc((size_t)nrows + 1, sizeof(int64_t));
    if (!rpos) return -1;
    for (int64_t i = 0; i <= nrows; i++) rowptrs[i] = 0;
    for (int64_t i = 0; i < nnz; i++) rowptrs[rows[i] + 1] += 1;                 /* counts[r] += 1 */
    for (int32_t i = 0; i < nrows; i++) rowptrs[i + 1] += rowptrs[i];            /* rowptrs[i+1] = rowptrs[i] + counts[i] */
    for (int32_t i = 0; i < nrows; i++) rpos[i] = rowptrs[i];
    for (int64_t i = 0; i < nnz; i++) {
        const int32_t row = rows[i];
        const int64_t pos = rpos[row];
        out_cols[pos] = cols[i];
        if (vs) memcpy((char *)out_vs + (size_t)pos * vsize, (const char *)vs + (size_t)i * vsize, (size_t)vsize);
        rpos[row] += 1;
    }
    free(rpos);
    return 0;
}

/* ---------------------------------------------------------------------------------
 * pick_rows: csr/structure.py:84-117 (_pick_rows_nvs) and :120-149 (_pick_rows).
 * First pass sums the picked rows' lengths (:89-94 / :125-130); second pass copies each
 * picked row's colinds (and values, any element size `vsize`; vs == NULL: structure
 * only) to the running position and records it in the new row pointers (:103-115 /
 * :139-147).  A row may be picked more than once.  Call with out_* == NULL to get the
 * result's nnz only.  Returns nnz.
 * ------------------------------------------------------------------------------- */
ORC_EXPORT int64_t orc_pick_rows(const int64_t *rp, const int32_t *ci, const void *vs, int32_t vsize,
                                 const int32_t *rows, int64_t nr, int64_t *out_rp, int32_t *out_ci, void *out_vs)
{
    int64_t nnz = 0;
    for (int64_t ii = 0; ii < nr; ii++) nnz += rp[rows[ii] + 1] - rp[rows[ii]];
    if (!out_rp) return nnz;
    int64_t pos = 0;
    for (int64_t ii = 0; ii < nr; ii++) {
        int64_t sp = rp[rows[ii]], ep = rp[rows[ii] + 1];
        int64_t itc = ep - sp;
        memcpy(out_ci + pos, ci + sp, (size_t)itc * sizeof(int32_t));
        if (vs) memcpy((char *)out_vs + (size_t)pos * vsize, (const char *)vs + (size_t)sp * vsize, (size_t)itc * vsize);
        out_rp[ii] = pos;
        pos += itc;
    }
    out_rp[nr] = pos;
    return nnz;
}
