// Dense-panel SpMM for libcsrk on gfx950: C = A B with B dense row-major [ncols x k]
// (BASELINE.json configs[2]: A 2M x 2M, nnz 5e7, k = 64, float64).  The reference has no
// dense-B entry point -- its mult_ab is sparse x sparse (csr/kernels/numba/multiply.py:13-38) --
// so this is the numeric recurrence of _num_mm (multiply.py:110-122: for every A entry
// (i, j, a): C[i, :] += a * B[j, :]) applied to a fully populated B.
//
// Shape of the work: per A entry one 8*k-byte row of B is read (k = 64: 512 B, one
// coalesced wave-wide load, lane = panel column) and FMA'd into the lane's accumulator;
// arithmetic intensity is 2k flop per (12 + 8k) bytes = 0.24 flop/B at k = 64, two orders
// below the float64 ridge, so the kernel is bound by the B-row gather stream and MFMA has
// nothing to offer (a sparse row shares no B rows with its neighbours, so a 16x16x4 tile
// would be 1/16 populated).  DESIGN.md section "SpMM" has the numbers.
//
// One wavefront per row segment (rows longer than 256 entries are split so the 10^5-entry
// rows of a power-law matrix do not serialise on one wave); segment partial panels are
// combined in segment order by a second kernel -- no float atomics, bitwise reproducible.
#include "common.h"

namespace csrk {

constexpr int MM_SEG = 256;

struct SpmmPlan {
    int64_t n_segs = 0;
    int64_t n_multi = 0;       // segments belonging to split rows (need a partial panel)
    DevBuf seg_off;            // int64[nrows + 1]
    DevBuf seg_row;            // int32[n_segs]
    DevBuf part;               // double[n_segs * k] (allocated on demand)
    int32_t part_k = 0;
};

void free_spmm_plan(SpmmPlan *p) { delete p; }

template <class P>
__global__ void mm_count_kernel(const P *__restrict__ rp, int32_t nrows, int64_t *__restrict__ cnt)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    int64_t len = (int64_t)rp[r + 1] - (int64_t)rp[r];
    cnt[r] = len <= MM_SEG ? 1 : (len + MM_SEG - 1) / MM_SEG;
}

__global__ void mm_fill_kernel(const int64_t *__restrict__ seg_off, int32_t nrows, int32_t *__restrict__ seg_row)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    for (int64_t q = seg_off[r]; q < seg_off[r + 1]; q++) seg_row[q] = (int32_t)r;
}

template <int VT>
__device__ __forceinline__ double mm_val(const void *v, int64_t k)
{
    if (VT == CSRK_VAL_F64) return ((const double *)v)[k];
    if (VT == CSRK_VAL_F32) return (double)((const float *)v)[k];
    return 1.0;
}

template <class P, int VT>
__global__ __launch_bounds__(256) void spmm_seg_kernel(const P *__restrict__ rp, const int32_t *__restrict__ ci,
                                                      const void *__restrict__ vs, const double *__restrict__ B,
                                                      int32_t k, int64_t ldb, double *__restrict__ C, int64_t ldc,
                                                      const int64_t *__restrict__ seg_off,
                                                      const int32_t *__restrict__ seg_row, int64_t n_segs,
                                                      double *__restrict__ part)
{
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (q >= n_segs) return;
    const int32_t r = seg_row[q];
    const int64_t first = seg_off[r], nseg = seg_off[r + 1] - first;
    const int64_t s = (int64_t)rp[r] + (q - first) * MM_SEG;
    int64_t e = (int64_t)rp[r + 1];
    if (nseg > 1 && e > s + MM_SEG) e = s + MM_SEG;
    double *dst = nseg == 1 ? C + (int64_t)r * ldc : part + q * (int64_t)k;
    for (int32_t c0 = 0; c0 < k; c0 += WAVE) {
        const int32_t c = c0 + lane;
        double acc = 0.0;
        if (c < k) {
            int64_t jj = s;
            // two B rows in flight per iteration
            for (; jj + 1 < e; jj += 2) {
                const int32_t j0 = ci[jj], j1 = ci[jj + 1];
                const double a0 = mm_val<VT>(vs, jj), a1 = mm_val<VT>(vs, jj + 1);
                const double b0 = B[(int64_t)j0 * ldb + c], b1 = B[(int64_t)j1 * ldb + c];
                acc += a0 * b0;
                acc += a1 * b1;
            }
            if (jj < e) acc += mm_val<VT>(vs, jj) * B[(int64_t)ci[jj] * ldb + c];
            dst[c] = acc;
        }
    }
}

__global__ __launch_bounds__(256) void spmm_fixup_kernel(const int64_t *__restrict__ seg_off, int32_t nrows, int32_t k,
                                                        const double *__restrict__ part, double *__restrict__ C,
                                                        int64_t ldc)
{
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (r >= nrows) return;
    const int64_t a = seg_off[r], b = seg_off[r + 1];
    if (b - a <= 1) return;
    for (int32_t c = lane; c < k; c += WAVE) {
        double acc = 0.0;
        for (int64_t q = a; q < b; q++) acc += part[q * (int64_t)k + c];
        C[(int64_t)r * ldc + c] = acc;
    }
}

template <class P>
static int build_mm_plan(Matrix *m, SpmmPlan *p, hipStream_t s)
{
    CSRK_TRY(p->seg_off.alloc((size_t)(m->nrows + 1) * 8));
    if (m->nrows > 0) {
        mm_count_kernel<P><<<(unsigned)ceil_div(m->nrows, 256), 256, 0, s>>>((const P *)m->d_rowptrs, m->nrows,
                                                                            p->seg_off.as<int64_t>());
        CSRK_LAUNCH_CHECK();
    }
    CSRK_TRY(exclusive_scan_i64(p->seg_off.as<int64_t>(), p->seg_off.as<int64_t>(), m->nrows, s));
    int64_t n = 0;
    CSRK_HIP(hipMemcpyAsync(&n, p->seg_off.as<int64_t>() + m->nrows, 8, hipMemcpyDeviceToHost, s));
    CSRK_HIP(hipStreamSynchronize(s));
    p->n_segs = n;
    CSRK_TRY(p->seg_row.alloc((size_t)n * 4));
    if (m->nrows > 0) {
        mm_fill_kernel<<<(unsigned)ceil_div(m->nrows, 256), 256, 0, s>>>(p->seg_off.as<int64_t>(), m->nrows,
                                                                        p->seg_row.as<int32_t>());
        CSRK_LAUNCH_CHECK();
    }
    return CSRK_OK;
}

static int spmm_device(Matrix *m, const double *dB, int32_t k, int64_t ldb, double *dC, int64_t ldc, hipStream_t s)
{
    CSRK_REQUIRE(k >= 0 && ldb >= k && ldc >= k, "bad panel geometry k=%d ldb=%lld ldc=%lld", k, (long long)ldb, (long long)ldc);
    if (m->nrows == 0 || k == 0) return CSRK_OK;
    SpmmPlan *p;
    {
        std::lock_guard<std::mutex> lk(m->mu);
        if (!m->spmm_plan) {
            SpmmPlan *np = new (std::nothrow) SpmmPlan();
            CSRK_REQUIRE(np, "out of host memory");
            // default stream + completion before use: see the caching allocator's contract (common.h)
            int rc = m->ptr64 ? build_mm_plan<int64_t>(m, np, nullptr) : build_mm_plan<int32_t>(m, np, nullptr);
            if (rc == CSRK_OK && hipDeviceSynchronize() != hipSuccess) rc = CSRK_ERR_HIP;
            if (rc != CSRK_OK) {
                delete np;
                return rc;
            }
            m->spmm_plan = np;
        }
        p = m->spmm_plan;
        if (p->n_segs > m->nrows && p->part_k < k) {     // some row is split: partial panels needed
            CSRK_TRY(p->part.alloc((size_t)p->n_segs * k * 8));
            p->part_k = k;
        }
    }
    const unsigned grid = (unsigned)ceil_div(p->n_segs * WAVE, 256);
#define GO(P, VT)                                                                                                     \
    spmm_seg_kernel<P, VT><<<grid, 256, 0, s>>>((const P *)m->d_rowptrs, m->d_colinds, m->d_values, dB, k, ldb, dC, ldc, \
                                                p->seg_off.as<int64_t>(), p->seg_row.as<int32_t>(), p->n_segs,         \
                                                p->part.as<double>())
    if (m->ptr64) {
        if (m->val_type == CSRK_VAL_F64) GO(int64_t, CSRK_VAL_F64);
        else if (m->val_type == CSRK_VAL_F32) GO(int64_t, CSRK_VAL_F32);
        else GO(int64_t, CSRK_VAL_NONE);
    } else {
        if (m->val_type == CSRK_VAL_F64) GO(int32_t, CSRK_VAL_F64);
        else if (m->val_type == CSRK_VAL_F32) GO(int32_t, CSRK_VAL_F32);
        else GO(int32_t, CSRK_VAL_NONE);
    }
#undef GO
    CSRK_LAUNCH_CHECK();
    if (p->n_segs > m->nrows) {
        spmm_fixup_kernel<<<(unsigned)ceil_div((int64_t)m->nrows * WAVE, 256), 256, 0, s>>>(
            p->seg_off.as<int64_t>(), m->nrows, k, p->part.as<double>(), dC, ldc);
        CSRK_LAUNCH_CHECK();
    }
    return CSRK_OK;
}

}  // namespace csrk

using namespace csrk;

extern "C" {

int csrk_spmm_dense_device(csrk_handle_t h, const double *d_B, int32_t k, int64_t ldb, double *d_C, int64_t ldc,
                           void *stream)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE(d_B && d_C, "B or C is NULL");
    return spmm_device(m, d_B, k, ldb, d_C, ldc, (hipStream_t)stream);
}

int csrk_spmm_dense(csrk_handle_t h, const double *B, int32_t k, int64_t ldb, double *C, int64_t ldc)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE(B && C, "B or C is NULL");
    CSRK_REQUIRE(k >= 0 && ldb >= k && ldc >= k, "bad panel geometry");
    if (m->nrows == 0 || k == 0) return CSRK_OK;
    DevBuf dB, dC;
    CSRK_TRY(dB.alloc((size_t)m->ncols * k * 8));
    CSRK_TRY(dC.alloc((size_t)m->nrows * k * 8));
    if (m->ncols)
        CSRK_HIP(hipMemcpy2D(dB.p, (size_t)k * 8, B, (size_t)ldb * 8, (size_t)k * 8, m->ncols, hipMemcpyHostToDevice));
    CSRK_TRY(spmm_device(m, dB.as<double>(), k, k, dC.as<double>(), k, nullptr));
    CSRK_HIP(hipMemcpy2D(C, (size_t)ldc * 8, dC.p, (size_t)k * 8, (size_t)k * 8, m->nrows, hipMemcpyDeviceToHost));
    return CSRK_OK;
}

}  // extern "C"
