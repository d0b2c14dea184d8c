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

constexpr int ROW_LONG = 8192;     // rows longer than this get a whole workgroup

template <class T> struct FInfo;
template <> struct FInfo<double> { static constexpr int maxexp = 1024, minexp = -1022; };
template <> struct FInfo<float> { static constexpr int maxexp = 128, minexp = -126; };

template <class P, class T>
__global__ __launch_bounds__(256) void unit_rows_kernel(const P *__restrict__ rp, T *__restrict__ vs,
                                                       T *__restrict__ norms, int32_t nrows,
                                                       int32_t *__restrict__ long_rows, int32_t *__restrict__ n_long)
{
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (r >= nrows) return;
    const int64_t sp = rp[r], ep = rp[r + 1];
    if (ep - sp > ROW_LONG) {             // left to the workgroup-per-row kernel
        if (lane == 0) long_rows[atomicAdd(n_long, 1)] = (int32_t)r;
        return;
    }
    if (sp == ep) {                       // empty row: norm 0 (transform.py:36-38)
        if (lane == 0) norms[r] = (T)0;
        return;
    }
    double vmax = 0.0;
    bool nan = false;
    for (int64_t k = sp + lane; k < ep; k += WAVE) {
        double a = fabs((double)vs[k]);
        nan |= a != a;
        vmax = a > vmax ? a : vmax;
    }
    vmax = wmax_nan(vmax, nan);
    // (m, e) = frexp(vmax); pnexp = clamp(-e, minexp, maxexp - 1); prenorm = 2^pnexp (:55-58)
    int ve = 0;
    if (vmax == vmax && !isinf(vmax)) (void)frexp(vmax, &ve);
    int pnexp = -ve;
    pnexp = pnexp > FInfo<T>::maxexp - 1 ? FInfo<T>::maxexp - 1 : pnexp;
    pnexp = pnexp < FInfo<T>::minexp ? FInfo<T>::minexp : pnexp;
    const T prenorm = (T)ldexp(1.0, pnexp);
    double ss = 0.0;
    for (int64_t k = sp + lane; k < ep; k += WAVE) {
        T v = vs[k] * prenorm;            // :59
        ss += (double)v * (double)v;
    }
    const T inorm = (T)sqrt(wsum(ss));    // :62
    if (lane == 0) norms[r] = inorm / prenorm;   // :63
    for (int64_t k = sp + lane; k < ep; k += WAVE) {
        T v = vs[k] * prenorm;
        vs[k] = v / inorm;                // :64
    }
}

template <class P, class T>
__global__ __launch_bounds__(256) void center_rows_kernel(const P *__restrict__ rp, T *__restrict__ vs,
                                                         T *__restrict__ means, int32_t nrows,
                                                         int32_t *__restrict__ long_rows, int32_t *__restrict__ n_long)
{
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (r >= nrows) return;
    const int64_t sp = rp[r], ep = rp[r + 1];
    if (ep - sp > ROW_LONG) {
        if (lane == 0) long_rows[atomicAdd(n_long, 1)] = (int32_t)r;
        return;
    }
    if (sp == ep) {
        if (lane == 0) means[r] = (T)0;
        return;
    }
    double s = 0.0;
    for (int64_t k = sp + lane; k < ep; k += WAVE) s += (double)vs[k];
    const T m = (T)(wsum(s) / (double)(ep - sp));
    if (lane == 0) means[r] = m;
    for (int64_t k = sp + lane; k < ep; k += WAVE) vs[k] = vs[k] - m;
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
    DevBuf d, longs, nl;
    CSRK_TRY(d.alloc((size_t)m->nrows * m->val_bytes()));
    const int64_t max_long = m->nnz / ROW_LONG + 1;           // at most this many rows can be that long
    CSRK_TRY(longs.alloc((size_t)max_long * 4));
    CSRK_TRY(nl.alloc(4));
    CSRK_HIP(hipMemset(nl.p, 0, 4));
    unsigned grid = (unsigned)ceil_div((int64_t)m->nrows * WAVE, 256);
#define GO(P, T)                                                                                          \
    do {                                                                                                  \
        if (UNIT)                                                                                         \
            unit_rows_kernel<P, T><<<grid, 256>>>((const P *)m->d_rowptrs, (T *)m->d_values, d.as<T>(), m->nrows,     \
                                                  longs.as<int32_t>(), nl.as<int32_t>());                 \
        else                                                                                              \
            center_rows_kernel<P, T><<<grid, 256>>>((const P *)m->d_rowptrs, (T *)m->d_values, d.as<T>(), m->nrows,   \
                                                    longs.as<int32_t>(), nl.as<int32_t>());               \
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
    // the SpMV plan depends only on rowptrs, which are unchanged
    return CSRK_OK;
}

}  // extern "C"
