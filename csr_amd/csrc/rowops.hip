// Row-wise operations of libcsrk on gfx950:
//   unit_rows / center_rows   csr/transform.py:29-66, :13-26   (in place on the values)
//   order_columns             csr/kernels/numba/__init__.py:47-52 -> csr/structure.py:156-169
//   filter_zeros              csr/_struct.py:61-76
//
// unit_rows / center_rows: one wavefront per row (rows are independent); lanes stride the
// row coalesced, reductions are __shfl_down trees accumulated in float64 whatever the
// storage dtype (so float32 results are within one rounding of exact, inside the
// reference tests' rel 1e-6), and the element updates are done in the storage dtype with
// the same two roundings as the reference (`v *= prenorm`, then `v /= inorm`).
// HBM traffic: values read 3x (max, norm, scale; the 2nd/3rd hit L2 for rows <= 4 MiB) and
// written once; row pointers once; norms once.
#include "common.h"

namespace csrk {

int transpose_matrix(Matrix *a, int with_values, Matrix **out, hipStream_t s);   // transpose.hip

__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return __shfl(v, 0, WAVE);
}

__device__ __forceinline__ double wmax_nan(double v, bool nan)
{
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) {
        double o = __shfl_down(v, off, WAVE);
        v = o > v ? o : v;
    }
    v = __shfl(v, 0, WAVE);
    return __any(nan) ? __builtin_nan("") : v;   // np.max propagates NaN (transform.py:52)
}

// reductions over aligned groups of LPR lanes (LPR a power of two <= 64); every lane gets the result
template <int LPR>
__device__ __forceinline__ double gsum(double v)
{
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}
template <int LPR>
__device__ __forceinline__ double gmax_nan(double v, bool nan)
{
    int f = nan ? 1 : 0;
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) {
        const double o = __shfl_xor(v, off, WAVE);
        v = o > v ? o : v;
        f |= __shfl_xor(f, off, WAVE);
    }
    return f ? __builtin_nan("") : v;   // np.max propagates NaN (transform.py:52)
}

constexpr int ROW_LONG = 8192;     // rows longer than this get a whole workgroup

template <class T> struct FInfo;
template <> struct FInfo<double> { static constexpr int maxexp = 1024, minexp = -1022; };
template <> struct FInfo<float> { static constexpr int maxexp = 128, minexp = -126; };

// LPR lanes per row: a whole wavefront (64) for ordinary rows, 8 when the average row has < 16 entries (a
// power-law matrix with millions of 1-entry rows kept 63 of 64 lanes idle: 6 ms for the headline matrix)
template <class P, class T, int LPR>
__global__ __launch_bounds__(256) void unit_rows_kernel(const P *__restrict__ rp, T *__restrict__ vs,
                                                       T *__restrict__ norms, int32_t nrows,
                                                       int32_t *__restrict__ long_rows, int32_t *__restrict__ n_long,
                                                       const int32_t *__restrict__ row_list, int32_t med_min,
                                                       int32_t *__restrict__ med_rows, int32_t *__restrict__ n_med)
{
    // row_list (optional): the rows to process; med_min > 0: rows with more entries than that (and <= ROW_LONG)
    // are left to a second, wavefront-per-row launch over med_rows
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPR;
    const int lane = threadIdx.x & (LPR - 1);
    if (q >= nrows) return;
    const int64_t r = row_list ? row_list[q] : q;
    const int64_t sp = rp[r], ep = rp[r + 1];
    if (ep - sp > ROW_LONG) {             // left to the workgroup-per-row kernel
        if (lane == 0) long_rows[atomicAdd(n_long, 1)] = (int32_t)r;
        return;
    }
    if (med_min > 0 && ep - sp > med_min) {
        if (lane == 0) med_rows[atomicAdd(n_med, 1)] = (int32_t)r;
        return;
    }
    if (sp == ep) {                       // empty row: norm 0 (transform.py:36-38)
        if (lane == 0) norms[r] = (T)0;
        return;
    }
    double vmax = 0.0;
    bool nan = false;
    for (int64_t k = sp + lane; k < ep; k += LPR) {
        double a = fabs((double)vs[k]);
        nan |= a != a;
        vmax = a > vmax ? a : vmax;
    }
    vmax = gmax_nan<LPR>(vmax, nan);
    // (m, e) = frexp(vmax); pnexp = clamp(-e, minexp, maxexp - 1); prenorm = 2^pnexp (:55-58)
    int ve = 0;
    if (vmax == vmax && !isinf(vmax)) (void)frexp(vmax, &ve);
    int pnexp = -ve;
    pnexp = pnexp > FInfo<T>::maxexp - 1 ? FInfo<T>::maxexp - 1 : pnexp;
    pnexp = pnexp < FInfo<T>::minexp ? FInfo<T>::minexp : pnexp;
    const T prenorm = (T)ldexp(1.0, pnexp);
    double ss = 0.0;
    for (int64_t k = sp + lane; k < ep; k += LPR) {
        T v = vs[k] * prenorm;            // :59
        ss += (double)v * (double)v;
    }
    const T inorm = (T)sqrt(gsum<LPR>(ss));    // :62
    if (lane == 0) norms[r] = inorm / prenorm;   // :63
    for (int64_t k = sp + lane; k < ep; k += LPR) {
        T v = vs[k] * prenorm;
        vs[k] = v / inorm;                // :64
    }
}

template <class P, class T, int LPR>
__global__ __launch_bounds__(256) void center_rows_kernel(const P *__restrict__ rp, T *__restrict__ vs,
                                                         T *__restrict__ means, int32_t nrows,
                                                         int32_t *__restrict__ long_rows, int32_t *__restrict__ n_long,
                                                         const int32_t *__restrict__ row_list, int32_t med_min,
                                                         int32_t *__restrict__ med_rows, int32_t *__restrict__ n_med)
{
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPR;
    const int lane = threadIdx.x & (LPR - 1);
    if (q >= nrows) return;
    const int64_t r = row_list ? row_list[q] : q;
    const int64_t sp = rp[r], ep = rp[r + 1];
    if (ep - sp > ROW_LONG) {
        if (lane == 0) long_rows[atomicAdd(n_long, 1)] = (int32_t)r;
        return;
    }
    if (med_min > 0 && ep - sp > med_min) {
        if (lane == 0) med_rows[atomicAdd(n_med, 1)] = (int32_t)r;
        return;
    }
    if (sp == ep) {
        if (lane == 0) means[r] = (T)0;
        return;
    }
    double s = 0.0;
    for (int64_t k = sp + lane; k < ep; k += LPR) s += (double)vs[k];
    const T m = (T)(gsum<LPR>(s) / (double)(ep - sp));
    if (lane == 0) means[r] = m;
    for (int64_t k = sp + lane; k < ep; k += LPR) vs[k] = vs[k] - m;
}

// Rows longer than ROW_LONG entries: one 1024-thread workgroup per row (a single wavefront would need
// ~10^4 serial iterations for the 10^6-entry rows of a power-law matrix).  Block-wide reductions =
// wavefront shuffles + one LDS stage, in a fixed order: deterministic.
constexpr int ROW_LONG_THREADS = 1024;

__device__ __forceinline__ double block_sum(double v, double *s_red)
{
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    __syncthreads();
    if (lane == 0) s_red[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int k = 0; k < ROW_LONG_THREADS / WAVE; k++) t += s_red[k];
    return t;
}

template <class P, class T, bool UNIT>
__global__ __launch_bounds__(ROW_LONG_THREADS) void row_stat_long_kernel(const P *__restrict__ rp, T *__restrict__ vs,
                                                                        T *__restrict__ out,
                                                                        const int32_t *__restrict__ long_rows)
{
    __shared__ double s_red[ROW_LONG_THREADS / WAVE];
    __shared__ int s_nan;
    const int32_t r = long_rows[blockIdx.x];
    const int64_t sp = rp[r], ep = rp[r + 1];
    const int tid = threadIdx.x;
    if (!UNIT) {
        double s = 0.0;
        for (int64_t k = sp + tid; k < ep; k += ROW_LONG_THREADS) s += (double)vs[k];
        const T m = (T)(block_sum(s, s_red) / (double)(ep - sp));
        if (tid == 0) out[r] = m;
        for (int64_t k = sp + tid; k < ep; k += ROW_LONG_THREADS) vs[k] = vs[k] - m;
        return;
    }
    if (tid == 0) s_nan = 0;
    __syncthreads();
    double vmax = 0.0;
    bool nan = false;
    for (int64_t k = sp + tid; k < ep; k += ROW_LONG_THREADS) {
        double a = fabs((double)vs[k]);
        nan |= a != a;
        vmax = a > vmax ? a : vmax;
    }
    if (nan) s_nan = 1;
    // block maximum through the same LDS stage (max is order independent)
    {
        const int lane = tid & (WAVE - 1), w = tid / WAVE;
#pragma unroll
        for (int off = WAVE / 2; off > 0; off >>= 1) {
            double o = __shfl_down(vmax, off, WAVE);
            vmax = o > vmax ? o : vmax;
        }
        __syncthreads();
        if (lane == 0) s_red[w] = vmax;
        __syncthreads();
        vmax = 0.0;
        for (int k = 0; k < ROW_LONG_THREADS / WAVE; k++) vmax = s_red[k] > vmax ? s_red[k] : vmax;
        if (s_nan) vmax = __builtin_nan("");
    }
    int ve = 0;
    if (vmax == vmax && !isinf(vmax)) (void)frexp(vmax, &ve);
    int pnexp = -ve;
    pnexp = pnexp > FInfo<T>::maxexp - 1 ? FInfo<T>::maxexp - 1 : pnexp;
    pnexp = pnexp < FInfo<T>::minexp ? FInfo<T>::minexp : pnexp;
    const T prenorm = (T)ldexp(1.0, pnexp);
    double ss = 0.0;
    for (int64_t k = sp + tid; k < ep; k += ROW_LONG_THREADS) {
        T v = vs[k] * prenorm;
        ss += (double)v * (double)v;
    }
    const T inorm = (T)sqrt(block_sum(ss, s_red));
    if (tid == 0) out[r] = inorm / prenorm;
    for (int64_t k = sp + tid; k < ep; k += ROW_LONG_THREADS) {
        T v = vs[k] * prenorm;
        vs[k] = v / inorm;
    }
}

template <bool UNIT>
static int row_stat(Matrix *m, void *out_host)
{
    CSRK_REQUIRE(m->val_type != CSRK_VAL_NONE, "matrix has no values");
    CSRK_REQUIRE(out_host || m->nrows == 0, "output is NULL");
    if (m->nrows == 0) return CSRK_OK;
    std::lock_guard<std::mutex> lk(m->mu);
    invalidate_plans(m);          // the values change in place; the plans hold copies of them
    DevBuf d, longs, nl;
    CSRK_TRY(d.alloc((size_t)m->nrows * m->val_bytes()));
    const int64_t max_long = m->nnz / ROW_LONG + 1;           // at most this many rows can be that long
    CSRK_TRY(longs.alloc((size_t)max_long * 4));
    CSRK_TRY(nl.alloc(4));
    CSRK_HIP(hipMemset(nl.p, 0, 4));
    // Mostly short rows (power-law matrices: half the rows of the headline matrix have <= 1 entry): 8 lanes per
    // row for rows of <= 64 entries, the others collected and given a wavefront each in a second launch.
    const bool short_rows = m->nnz / (int64_t)m->nrows < 64;
    constexpr int MED_MIN = 64;
    DevBuf meds, nm;
    if (short_rows) {
        CSRK_TRY(meds.alloc((size_t)(m->nnz / MED_MIN + 1) * 4));
        CSRK_TRY(nm.alloc(4));
        CSRK_HIP(hipMemset(nm.p, 0, 4));
    }
    unsigned grid = (unsigned)ceil_div((int64_t)m->nrows * (short_rows ? 8 : WAVE), 256);
#define ROW_ARGS(T, LIST, N, MEDMIN)                                                                      \
    (const P_ *)m->d_rowptrs, (T *)m->d_values, d.as<T>(), N, longs.as<int32_t>(), nl.as<int32_t>(), LIST, MEDMIN,    \
        meds.as<int32_t>(), nm.as<int32_t>()
#define GO(P, T)                                                                                          \
    do {                                                                                                  \
        typedef P P_;                                                                                     \
        if (UNIT && short_rows)                                                                           \
            unit_rows_kernel<P, T, 8><<<grid, 256>>>(ROW_ARGS(T, (const int32_t *)nullptr, m->nrows, MED_MIN));      \
        else if (UNIT)                                                                                    \
            unit_rows_kernel<P, T, WAVE><<<grid, 256>>>(ROW_ARGS(T, (const int32_t *)nullptr, m->nrows, 0));         \
        else if (short_rows)                                                                              \
            center_rows_kernel<P, T, 8><<<grid, 256>>>(ROW_ARGS(T, (const int32_t *)nullptr, m->nrows, MED_MIN));    \
        else                                                                                              \
            center_rows_kernel<P, T, WAVE><<<grid, 256>>>(ROW_ARGS(T, (const int32_t *)nullptr, m->nrows, 0));       \
        CSRK_LAUNCH_CHECK();                                                                              \
        if (short_rows) {                                                                                 \
            int32_t n_med = 0;                                                                            \
            CSRK_HIP(hipMemcpy(&n_med, nm.p, 4, hipMemcpyDeviceToHost));                                  \
            if (n_med > 0) {                                                                              \
                const unsigned gm = (unsigned)ceil_div((int64_t)n_med * WAVE, 256);                       \
                if (UNIT)                                                                                 \
                    unit_rows_kernel<P, T, WAVE><<<gm, 256>>>(ROW_ARGS(T, meds.as<int32_t>(), n_med, 0)); \
                else                                                                                      \
                    center_rows_kernel<P, T, WAVE><<<gm, 256>>>(ROW_ARGS(T, meds.as<int32_t>(), n_med, 0));           \
            }                                                                                             \
        }                                                                                                 \
        CSRK_LAUNCH_CHECK();                                                                              \
        int32_t n_long = 0;                                                                               \
        CSRK_HIP(hipMemcpy(&n_long, nl.p, 4, hipMemcpyDeviceToHost));                                     \
        if (n_long > 0)                                                                                   \
            row_stat_long_kernel<P, T, UNIT><<<(unsigned)n_long, ROW_LONG_THREADS>>>(                     \
                (const P *)m->d_rowptrs, (T *)m->d_values, d.as<T>(), longs.as<int32_t>());               \
    } while (0)
    if (m->ptr64) {
        if (m->val_type == CSRK_VAL_F64) GO(int64_t, double); else GO(int64_t, float);
    } else {
        if (m->val_type == CSRK_VAL_F64) GO(int32_t, double); else GO(int32_t, float);
    }
#undef GO
#undef ROW_ARGS
    CSRK_LAUNCH_CHECK();
    CSRK_HIP(hipMemcpy(out_host, d.p, (size_t)m->nrows * m->val_bytes(), hipMemcpyDeviceToHost));
    return CSRK_OK;
}

// ---- filter_zeros ---------------------------------------------------------------------------
template <class C>
__global__ void nz_flag_kernel(const double *__restrict__ vs, int64_t nnz, C *__restrict__ flags)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nnz) flags[i] = vs[i] != 0.0 ? 1 : 0;   // NaN != 0 is true: NaN is kept (_struct.py:68)
}

template <class C>
__global__ void nz_compact_kernel(const int32_t *__restrict__ ci, const double *__restrict__ vs, int64_t nnz,
                                  const C *__restrict__ pos, int32_t *__restrict__ oci, double *__restrict__ ovs)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    double v = vs[i];
    if (v != 0.0) {
        C o = pos[i];
        oci[o] = ci[i];
        ovs[o] = v;
    }
}

template <class P, class C>
__global__ void nz_rowptr_kernel(const P *__restrict__ rp, int32_t nrows, const C *__restrict__ pos, P *__restrict__ orp)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= nrows) orp[r] = (P)pos[rp[r]];
}

template <class P, class C>
static int filter_impl(Matrix *m, Matrix **out)
{
    const int64_t nnz = m->nnz;
    DevBuf pos;
    CSRK_TRY(pos.alloc((size_t)(nnz + 1) * sizeof(C)));
    if (nnz > 0) {
        nz_flag_kernel<C><<<(unsigned)ceil_div(nnz, 256), 256>>>((const double *)m->d_values, nnz, pos.as<C>());
        CSRK_LAUNCH_CHECK();
    }
    if (sizeof(C) == 8)
        CSRK_TRY(exclusive_scan_i64((const int64_t *)pos.p, (int64_t *)pos.p, nnz, nullptr));
    else
        CSRK_TRY(exclusive_scan_i32((const int32_t *)pos.p, (int32_t *)pos.p, nnz, nullptr));
    C total = 0;
    CSRK_HIP(hipMemcpy(&total, pos.as<C>() + nnz, sizeof(C), hipMemcpyDeviceToHost));
    Matrix *f = nullptr;
    CSRK_TRY(new_matrix(m->nrows, m->ncols, (int64_t)total, m->ptr64, CSRK_VAL_F64, &f));
    if (nnz > 0) {
        nz_compact_kernel<C><<<(unsigned)ceil_div(nnz, 256), 256>>>(m->d_colinds, (const double *)m->d_values, nnz,
                                                                    pos.as<C>(), f->d_colinds, (double *)f->d_values);
    }
    nz_rowptr_kernel<P, C><<<(unsigned)ceil_div((int64_t)m->nrows + 1, 256), 256>>>((const P *)m->d_rowptrs, m->nrows,
                                                                                    pos.as<C>(), (P *)f->d_rowptrs);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("filter_zeros kernels failed: %s", hipGetErrorString(e));
        delete f;
        return CSRK_ERR_HIP;
    }
    *out = f;
    return CSRK_OK;
}

// ---- order_columns ----------------------------------------------------------------------------
__global__ void cast_f64_to_f32_kernel(const double *__restrict__ in, float *__restrict__ out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];   // exact: the doubles were widened floats
}

// ---- pick_rows (csr/csr.py:347-364 -> csr/structure.py:84-149) ---------------------------------------
// The reference walks the requested rows twice: sum their lengths, then copy each row's colinds (and
// values) to a running position that becomes the new row pointer.  Here: lengths -> exclusive scan ->
// one wavefront per picked row copies it.  A row may be picked more than once.  HBM-bound byte moving.
template <class P>
__global__ void pick_len_kernel(const P *__restrict__ rp, const int32_t *__restrict__ rows, int64_t nr, int32_t nrows,
                                int64_t *__restrict__ len, int32_t *__restrict__ bad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nr) return;
    int64_t l = 0;
    if (i < nr) {
        const int32_t r = rows[i];
        if (r < 0 || r >= nrows)
            atomicOr(bad, 1);
        else
            l = (int64_t)rp[r + 1] - (int64_t)rp[r];
    }
    len[i] = l;
}

// One thread per OUTPUT entry (consecutive lanes write consecutive addresses whatever the row lengths -- a
// wavefront per picked row needed 13 ms for 2M rows of the headline matrix, whose 10^6-entry rows each kept one
// wavefront busy): the workgroup's 2048 entries span the picked rows [i0, i1], found by two binary searches
// over the output offsets; each thread then searches only that span.
constexpr int PICK_EPT = 8;
template <class P, class PO, class T>
__global__ __launch_bounds__(256) void pick_copy_kernel(const P *__restrict__ rp, const int32_t *__restrict__ ci,
                                                       const T *__restrict__ vs, const int32_t *__restrict__ rows,
                                                       int64_t nr, const int64_t *__restrict__ off, int64_t nnz,
                                                       int32_t *__restrict__ oci, T *__restrict__ ovs)
{
    __shared__ int64_t s_span[2];
    const int64_t base = (int64_t)blockIdx.x * (256 * PICK_EPT);
    if (threadIdx.x < 2) {
        // last picked row whose offset is <= the workgroup's first (threadIdx 0) / last (1) entry
        int64_t o = threadIdx.x == 0 ? base : base + 256 * PICK_EPT - 1;
        o = o < nnz - 1 ? o : nnz - 1;
        int64_t lo = 0, hi = nr - 1;
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) >> 1;
            if (off[mid] <= o)
                lo = mid;
            else
                hi = mid - 1;
        }
        s_span[threadIdx.x] = lo;
    }
    __syncthreads();
    const int64_t i0 = s_span[0], i1 = s_span[1];
#pragma unroll
    for (int k = 0; k < PICK_EPT; k++) {
        const int64_t o = base + k * 256 + threadIdx.x;
        if (o >= nnz) break;
        int64_t lo = i0, hi = i1;
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) >> 1;
            if (off[mid] <= o)
                lo = mid;
            else
                hi = mid - 1;
        }
        const int64_t src = (int64_t)rp[rows[lo]] + (o - off[lo]);
        oci[o] = ci[src];
        if (vs) ovs[o] = vs[src];
    }
}

template <class PO>
__global__ void pick_ptr_kernel(const int64_t *__restrict__ off, int64_t nr, PO *__restrict__ orp)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= nr) orp[i] = (PO)off[i];
}

template <class P, class T>
static int pick_impl(Matrix *m, const int32_t *d_rows, int64_t nr, int vt, Matrix **out)
{
    DevBuf len, bad;
    CSRK_TRY(len.alloc((size_t)(nr + 2) * 8));
    CSRK_TRY(bad.alloc(4));
    CSRK_HIP(hipMemsetAsync(bad.p, 0, 4, nullptr));
    pick_len_kernel<P><<<(unsigned)ceil_div(nr + 1, 256), 256>>>((const P *)m->d_rowptrs, d_rows, nr, m->nrows,
                                                                len.as<int64_t>(), bad.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(exclusive_scan_i64(len.as<int64_t>(), len.as<int64_t>(), nr, nullptr));
    int64_t nnz = 0;
    int32_t is_bad = 0;
    CSRK_HIP(hipMemcpy(&nnz, len.as<int64_t>() + nr, 8, hipMemcpyDeviceToHost));
    CSRK_HIP(hipMemcpy(&is_bad, bad.p, 4, hipMemcpyDeviceToHost));
    CSRK_REQUIRE(!is_bad, "pick_rows: row index out of range [0, %d)", m->nrows);
    const int p64 = nnz > INT32_MAX;
    Matrix *t = nullptr;
    CSRK_TRY(new_matrix((int32_t)nr, m->ncols, nnz, p64, vt, &t));
    const unsigned gp = (unsigned)ceil_div(nr + 1, 256), grid = (unsigned)ceil_div(nnz, 256 * PICK_EPT);
    const T *vs = vt == CSRK_VAL_NONE ? (const T *)nullptr : (const T *)m->d_values;
    if (p64)
        pick_ptr_kernel<int64_t><<<gp, 256>>>(len.as<int64_t>(), nr, (int64_t *)t->d_rowptrs);
    else
        pick_ptr_kernel<int32_t><<<gp, 256>>>(len.as<int64_t>(), nr, (int32_t *)t->d_rowptrs);
    if (nnz > 0) {
        if (p64)
            pick_copy_kernel<P, int64_t, T><<<grid, 256>>>((const P *)m->d_rowptrs, m->d_colinds, vs, d_rows, nr,
                                                          len.as<int64_t>(), nnz, t->d_colinds, (T *)t->d_values);
        else
            pick_copy_kernel<P, int32_t, T><<<grid, 256>>>((const P *)m->d_rowptrs, m->d_colinds, vs, d_rows, nr,
                                                          len.as<int64_t>(), nnz, t->d_colinds, (T *)t->d_values);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();      // `len` goes back to the pool on return
    if (e != hipSuccess) {
        set_error("pick_rows failed: %s", hipGetErrorString(e));
        delete t;
        return CSRK_ERR_HIP;
    }
    *out = t;
    return CSRK_OK;
}

}  // namespace csrk

using namespace csrk;

extern "C" {

int csrk_unit_rows(csrk_handle_t h, void *norms)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    return row_stat<true>(m, norms);
}

int csrk_center_rows(csrk_handle_t h, void *means)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    return row_stat<false>(m, means);
}

int csrk_pick_rows(csrk_handle_t h, const int32_t *rows, int64_t n_rows, int with_values, csrk_handle_t *out)
{
    CSRK_REQUIRE(out, "out is NULL");
    *out = 0;
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE(n_rows >= 0 && n_rows <= INT32_MAX - 1, "bad row count %lld", (long long)n_rows);
    CSRK_REQUIRE(rows || n_rows == 0, "rows is NULL");
    DevBuf d_rows;
    CSRK_TRY(d_rows.alloc((size_t)(n_rows ? n_rows : 1) * 4));
    if (n_rows) CSRK_HIP(hipMemcpy(d_rows.p, rows, (size_t)n_rows * 4, hipMemcpyHostToDevice));
    const int vt = (with_values && m->val_type != CSRK_VAL_NONE) ? m->val_type : CSRK_VAL_NONE;
    std::lock_guard<std::mutex> lk(m->mu);
    Matrix *t = nullptr;
    int rc;
    if (m->ptr64)
        rc = vt == CSRK_VAL_F32 ? pick_impl<int64_t, float>(m, d_rows.as<int32_t>(), n_rows, vt, &t)
                                : pick_impl<int64_t, double>(m, d_rows.as<int32_t>(), n_rows, vt, &t);
    else
        rc = vt == CSRK_VAL_F32 ? pick_impl<int32_t, float>(m, d_rows.as<int32_t>(), n_rows, vt, &t)
                                : pick_impl<int32_t, double>(m, d_rows.as<int32_t>(), n_rows, vt, &t);
    if (rc != CSRK_OK) return rc;
    *out = to_handle(t);
    return CSRK_OK;
}

int csrk_filter_zeros(csrk_handle_t h, csrk_handle_t *out)
{
    CSRK_REQUIRE(out, "out is NULL");
    *out = 0;
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE(m->val_type == CSRK_VAL_F64, "filter_zeros needs float64 values");
    std::lock_guard<std::mutex> lk(m->mu);
    Matrix *f = nullptr;
    int rc;
    if (m->ptr64)
        rc = filter_impl<int64_t, int64_t>(m, &f);
    else
        rc = filter_impl<int32_t, int32_t>(m, &f);
    if (rc != CSRK_OK) return rc;
    *out = to_handle(f);
    return CSRK_OK;
}

// Sorting every row by column, stably, is what two stable transposes do: the first orders
// entries by (col, source position), the second by (row, col, source position).
int csrk_order_columns(csrk_handle_t h)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    if (m->nnz == 0) return CSRK_OK;
    std::lock_guard<std::mutex> lk(m->mu);
    Matrix *t1 = nullptr, *t2 = nullptr;
    CSRK_TRY(transpose_matrix(m, 1, &t1, nullptr));
    int rc = transpose_matrix(t1, 1, &t2, nullptr);
    delete t1;
    if (rc != CSRK_OK) return rc;
    hipError_t e = hipMemcpy(m->d_colinds, t2->d_colinds, (size_t)m->nnz * 4, hipMemcpyDeviceToDevice);
    if (e == hipSuccess && m->val_type == CSRK_VAL_F64)
        e = hipMemcpy(m->d_values, t2->d_values, (size_t)m->nnz * 8, hipMemcpyDeviceToDevice);
    if (e == hipSuccess && m->val_type == CSRK_VAL_F32) {
        cast_f64_to_f32_kernel<<<(unsigned)ceil_div(m->nnz, 256), 256>>>((const double *)t2->d_values,
                                                                         (float *)m->d_values, m->nnz);
        e = hipDeviceSynchronize();
    }
    delete t2;
    if (e != hipSuccess) {
        set_error("order_columns copy-back failed: %s", hipGetErrorString(e));
        return CSRK_ERR_HIP;
    }
    invalidate_plans(m);      // the SpMV / SpMM plans hold re-ordered copies of colinds and values
    return CSRK_OK;
}

}  // extern "C"
