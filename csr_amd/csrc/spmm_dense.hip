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
    Tier0View heavy;           // heavy rows come from the SpMV plan's column-block-major panel (or .on == false)
    DevBuf hpart;              // double[pairs * k]: per-(block, row) partial panels
    int32_t hpart_k = 0;
    int64_t n_segs = 0;
    int64_t n_multi = 0;       // segments belonging to split rows (need a partial panel)
    DevBuf part_off;           // int64[nrows + 1]: first partial slot of each split row
    DevBuf split_rows;         // int32[n_split]: rows with more than one segment
    int32_t n_split = 0;
    DevBuf seg_off;            // int64[nrows + 1]
    DevBuf seg_row;            // int32[n_segs]
    DevBuf part;               // double[n_segs * k] (allocated on demand)
    int32_t part_k = 0;
};

void free_spmm_plan(SpmmPlan *p) { delete p; }

template <int VT>
__device__ __forceinline__ double mm_val(const void *v, int64_t k)
{
    if (VT == CSRK_VAL_F64) return ((const double *)v)[k];
    if (VT == CSRK_VAL_F32) return (double)((const float *)v)[k];
    return 1.0;
}


// acc += sum over entries [s, e) of val * B[col, c].  The wavefront first loads 64 entries' (col, val)
// with one coalesced load per array, then walks them 8 at a time with the indices broadcast by
// shuffles, so 8 independent B-row loads are in flight per wavefront.  (Loading an index, then its B
// row, then the next index made both SpMM kernels latency-bound: ~100 ps per entry.)  `live` = this lane
// owns a panel column; dead lanes still take part in the shuffles.
constexpr int MM_UNROLL = 8;
template <int VT>
__device__ __forceinline__ double mm_accumulate(const int32_t *__restrict__ ci, const void *__restrict__ vs, int64_t s,
                                                int64_t e, const double *__restrict__ B, int64_t ldb, int32_t c,
                                                bool live, int lane)
{
    double acc = 0.0;
    for (int64_t base = s; base < e; base += WAVE) {
        const int n = (int)(e - base < WAVE ? e - base : WAVE);
        const int32_t mycol = lane < n ? ci[base + lane] : 0;
        const double myval = lane < n ? mm_val<VT>(vs, base + lane) : 0.0;
        for (int j = 0; j < n; j += MM_UNROLL) {
            double bv[MM_UNROLL], av[MM_UNROLL];
#pragma unroll
            for (int u = 0; u < MM_UNROLL; u++) {
                const int src = j + u < n ? j + u : n - 1;          // clamped: every load is a real entry's row
                const int32_t cu = __shfl(mycol, src, WAVE);
                av[u] = __shfl(myval, src, WAVE);
                bv[u] = live ? B[(int64_t)cu * ldb + c] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < MM_UNROLL; u++) {
                const double t = av[u] * bv[u];
                acc += j + u < n ? t : 0.0;                          // masked after the multiply (0 * inf)
            }
        }
    }
    return acc;
}

// heavy_min > 0: rows with at least that many entries are served by the heavy-row kernels (0 segments here)
template <class P>
__global__ void mm_count_kernel(const P *__restrict__ rp, int32_t nrows, int64_t *__restrict__ cnt,
                                int64_t *__restrict__ pcnt, int32_t heavy_min)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    int64_t len = (int64_t)rp[r + 1] - (int64_t)rp[r];
    const int64_t n = (heavy_min > 0 && len >= heavy_min) ? 0 : (len <= MM_SEG ? 1 : (len + MM_SEG - 1) / MM_SEG);
    cnt[r] = n;
    pcnt[r] = n > 1 ? n : 0;         // only split rows need partial panels
}

// ---- heavy rows: column-block-major, B window resident in L2 -------------------------------------------
// A row with thousands of entries reads thousands of different B rows; row after row that is a pure
// HBM stream (B does not fit the Infinity Cache).  The SpMV plan already holds the heavy rows re-sorted
// into (column block, row) pairs of 4096 columns; per block the B rows it needs are 4096 * k * 8 B = 2 MiB
// at k = 64, which stays in an XCD's L2 when the block is served by one XCD (workgroups with
// blockIdx % 8 == block % 8; a speed assumption only).  One wavefront per pair, lane = panel column,
// partial panels per pair, summed per row in block order: deterministic.
constexpr int MM_STREAMS = 8;

template <class PP>
__global__ __launch_bounds__(256) void spmm_heavy_kernel(const PP *__restrict__ prp, const int32_t *__restrict__ pci,
                                                        const double *__restrict__ pvs, const double *__restrict__ B,
                                                        int32_t k, int64_t ldb, int32_t n_rows, int32_t n_blocks,
                                                        double *__restrict__ part)
{
    const int g = blockIdx.x % MM_STREAMS;
    const int64_t j = (int64_t)(blockIdx.x / MM_STREAMS) * (256 / WAVE) + threadIdx.x / WAVE;   // pair index inside the stream
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t b = g + (int64_t)MM_STREAMS * (j / n_rows);
    if (b >= n_blocks) return;
    const int64_t q = b * n_rows + j % n_rows;
    const int64_t s = prp[q], e = prp[q + 1];
    if (s == e) return;                                  // empty pair: the reduce skips it too
    for (int32_t c0 = 0; c0 < k; c0 += WAVE) {
        const int32_t c = c0 + lane;
        const bool live = c < k;
        const double acc = mm_accumulate<CSRK_VAL_F64>(pci, pvs, s, e, B, ldb, live ? c : 0, live, lane);
        if (live) part[q * (int64_t)k + c] = acc;
    }
}

template <class PP>
__global__ __launch_bounds__(256) void spmm_heavy_reduce_kernel(const PP *__restrict__ prp, const int32_t *__restrict__ row_list,
                                                               int32_t n_rows, int32_t n_blocks, int32_t k,
                                                               const double *__restrict__ part, double *__restrict__ C,
                                                               int64_t ldc)
{
    const int64_t h = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (h >= n_rows) return;
    const int64_t r = row_list[h];
    for (int32_t c = lane; c < k; c += WAVE) {
        double acc = 0.0;
        for (int64_t b0 = 0; b0 < n_blocks; b0 += MM_UNROLL) {
            double v[MM_UNROLL];
#pragma unroll
            for (int u = 0; u < MM_UNROLL; u++) {                 // 8 partial rows in flight, added in block order
                const int64_t b = b0 + u < n_blocks ? b0 + u : n_blocks - 1;
                const int64_t q = b * n_rows + h;
                const bool have = b0 + u < n_blocks && prp[q] != prp[q + 1];
                const double t = part[q * (int64_t)k + c];        // unconditional load (valid memory), select after
                v[u] = have ? t : 0.0;
            }
#pragma unroll
            for (int u = 0; u < MM_UNROLL; u++) acc += v[u];
        }
        C[r * ldc + c] = acc;
    }
}

__global__ void mm_fill_kernel(const int64_t *__restrict__ seg_off, int32_t nrows, int32_t *__restrict__ seg_row)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    for (int64_t q = seg_off[r]; q < seg_off[r + 1]; q++) seg_row[q] = (int32_t)r;
}

// One wavefront per row segment of the rows that are not served by the heavy-row kernels.
template <class P, int VT>
__global__ __launch_bounds__(256) void spmm_seg_kernel(const P *__restrict__ rp, const int32_t *__restrict__ ci,
                                                      const void *__restrict__ vs, const double *__restrict__ B,
                                                      int32_t k, int64_t ldb, double *__restrict__ C, int64_t ldc,
                                                      const int64_t *__restrict__ seg_off,
                                                      const int32_t *__restrict__ seg_row, int64_t n_segs,
                                                      const int64_t *__restrict__ part_off, double *__restrict__ part)
{
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (q >= n_segs) return;
    const int32_t r = seg_row[q];
    const int64_t first = seg_off[r], nseg = seg_off[r + 1] - first;
    const int64_t s = (int64_t)rp[r] + (q - first) * MM_SEG;
    int64_t e = (int64_t)rp[r + 1];
    if (nseg > 1 && e > s + MM_SEG) e = s + MM_SEG;
    double *dst = nseg == 1 ? C + (int64_t)r * ldc : part + (part_off[r] + (q - first)) * (int64_t)k;
    for (int32_t c0 = 0; c0 < k; c0 += WAVE) {
        const int32_t c = c0 + lane;
        const bool live = c < k;
        const double acc = mm_accumulate<VT>(ci, vs, s, e, B, ldb, live ? c : 0, live, lane);
        if (live) dst[c] = acc;
    }
}

__global__ void mm_list_split_kernel(const int64_t *__restrict__ seg_off, int32_t nrows, int32_t *__restrict__ list,
                                     int32_t *__restrict__ n_list)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nrows && seg_off[r + 1] - seg_off[r] > 1) list[atomicAdd(n_list, 1)] = (int32_t)r;
}

// one wavefront per SPLIT row (listed at plan time; a wavefront per row of a 2M-row matrix cost 0.15 ms);
// segment partials are added in segment order
__global__ __launch_bounds__(256) void spmm_fixup_kernel(const int64_t *__restrict__ seg_off,
                                                        const int64_t *__restrict__ part_off,
                                                        const int32_t *__restrict__ split_rows, int32_t n_split, int32_t k,
                                                        const double *__restrict__ part, double *__restrict__ C,
                                                        int64_t ldc)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (i >= n_split) return;
    const int64_t r = split_rows[i];
    const int64_t n = seg_off[r + 1] - seg_off[r];
    const int64_t a = part_off[r];
    for (int32_t c = lane; c < k; c += WAVE) {
        double acc = 0.0;
        for (int64_t q = a; q < a + n; q++) acc += part[q * (int64_t)k + c];
        C[r * ldc + c] = acc;
    }
}

template <class P>
static int build_mm_plan(Matrix *m, SpmmPlan *p, hipStream_t s)
{
    CSRK_TRY(p->seg_off.alloc((size_t)(m->nrows + 1) * 8));
    CSRK_TRY(p->part_off.alloc((size_t)(m->nrows + 1) * 8));
    if (m->nrows > 0) {
        mm_count_kernel<P><<<(unsigned)ceil_div(m->nrows, 256), 256, 0, s>>>((const P *)m->d_rowptrs, m->nrows,
                                                                            p->seg_off.as<int64_t>(),
                                                                            p->part_off.as<int64_t>(),
                                                                            p->heavy.on ? p->heavy.min_entries : 0);
        CSRK_LAUNCH_CHECK();
    }
    CSRK_TRY(exclusive_scan_i64(p->seg_off.as<int64_t>(), p->seg_off.as<int64_t>(), m->nrows, s));
    CSRK_TRY(exclusive_scan_i64(p->part_off.as<int64_t>(), p->part_off.as<int64_t>(), m->nrows, s));
    int64_t n = 0, nm = 0;
    CSRK_HIP(hipMemcpyAsync(&n, p->seg_off.as<int64_t>() + m->nrows, 8, hipMemcpyDeviceToHost, s));
    CSRK_HIP(hipMemcpyAsync(&nm, p->part_off.as<int64_t>() + m->nrows, 8, hipMemcpyDeviceToHost, s));
    CSRK_HIP(hipStreamSynchronize(s));
    p->n_segs = n;
    p->n_multi = nm;
    if (nm > 0) {
        DevBuf cnt;
        CSRK_TRY(cnt.alloc(4));
        CSRK_HIP(hipMemsetAsync(cnt.p, 0, 4, s));
        CSRK_TRY(p->split_rows.alloc((size_t)(nm / 2 + 1) * 4));      // every split row has >= 2 segments
        mm_list_split_kernel<<<(unsigned)ceil_div(m->nrows, 256), 256, 0, s>>>(p->seg_off.as<int64_t>(), m->nrows,
                                                                              p->split_rows.as<int32_t>(), cnt.as<int32_t>());
        CSRK_LAUNCH_CHECK();
        CSRK_HIP(hipMemcpyAsync(&p->n_split, cnt.p, 4, hipMemcpyDeviceToHost, s));
        CSRK_HIP(hipStreamSynchronize(s));
    }
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
    Tier0View hv;
    const char *env = getenv("CSRK_SPMM_HEAVY");
    // heavy-row blocking pays when the B rows a block needs fit in L2 but B as a whole does not
    if (!(env && env[0] == '0') && !m->spmm_plan && ((int64_t)m->ncols * k * 8 > (64ll << 20) || (env && env[0] == '1')))
        CSRK_TRY(spmv_tier0_view(m, &hv));
    // The launch group below shares the plan's partial-panel buffers (hpart, part): the per-handle lock is held
    // across it, as spmv_dispatch does, so that concurrent callers are ordered by the stream instead of interleaving
    // (csrk.h: calls on the same handle serialise).
    std::lock_guard<std::mutex> lk(m->mu);
    if (s) m->used_user_stream = true;
    if (!m->spmm_plan) {
        SpmmPlan *np = new (std::nothrow) SpmmPlan();
        CSRK_REQUIRE(np, "out of host memory");
        if (hv.on && hv.pairs * (int64_t)k * 8 <= (4ll << 30)) np->heavy = hv;
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
    // a wider panel than any before needs larger partial buffers: earlier launches (any stream) may still be using the
    // old blocks, which DevBuf::alloc returns to the pool at once -- wait for the device first
    const bool grow_h = p->heavy.on && p->hpart_k < k, grow_p = p->n_multi > 0 && p->part_k < k;
    if ((grow_h && p->hpart.p) || (grow_p && p->part.p)) CSRK_HIP(hipDeviceSynchronize());
    if (grow_h) {
        if (p->heavy.pairs * (int64_t)k * 8 > (4ll << 30)) {
            set_error("panel width %d too large for the heavy-row partial buffer of this plan", k);
            return CSRK_ERR_UNSUPPORTED;
        }
        CSRK_TRY(p->hpart.alloc((size_t)p->heavy.pairs * k * 8));
        p->hpart_k = k;
    }
    if (grow_p) {           // some row is split: partial panels needed
        CSRK_TRY(p->part.alloc((size_t)p->n_multi * k * 8));
        p->part_k = k;
    }
    if (p->heavy.on) {
        const Tier0View &hvw = p->heavy;
        const int64_t per_stream_blocks = ceil_div(hvw.n_blocks, MM_STREAMS);
        const int64_t wgs = MM_STREAMS * ceil_div(per_stream_blocks * hvw.n_rows, 256 / WAVE);
        const unsigned rgrid = (unsigned)ceil_div((int64_t)hvw.n_rows * WAVE, 256);
        if (hvw.p64) {
            spmm_heavy_kernel<int64_t><<<(unsigned)wgs, 256, 0, s>>>((const int64_t *)hvw.rp, hvw.ci, hvw.vs, dB, k, ldb,
                                                                    hvw.n_rows, hvw.n_blocks, p->hpart.as<double>());
            spmm_heavy_reduce_kernel<int64_t><<<rgrid, 256, 0, s>>>((const int64_t *)hvw.rp, hvw.row_list, hvw.n_rows,
                                                                   hvw.n_blocks, k, p->hpart.as<double>(), dC, ldc);
        } else {
            spmm_heavy_kernel<int32_t><<<(unsigned)wgs, 256, 0, s>>>((const int32_t *)hvw.rp, hvw.ci, hvw.vs, dB, k, ldb,
                                                                    hvw.n_rows, hvw.n_blocks, p->hpart.as<double>());
            spmm_heavy_reduce_kernel<int32_t><<<rgrid, 256, 0, s>>>((const int32_t *)hvw.rp, hvw.row_list, hvw.n_rows,
                                                                   hvw.n_blocks, k, p->hpart.as<double>(), dC, ldc);
        }
        CSRK_LAUNCH_CHECK();
    }
    if (p->n_segs == 0) return CSRK_OK;
    const unsigned grid = (unsigned)ceil_div(p->n_segs * WAVE, 256);
#define GO(P, VT)                                                                                                     \
    spmm_seg_kernel<P, VT><<<grid, 256, 0, s>>>((const P *)m->d_rowptrs, m->d_colinds, m->d_values, dB, k, ldb, dC, ldc, \
                                                p->seg_off.as<int64_t>(), p->seg_row.as<int32_t>(), p->n_segs,         \
                                                p->part_off.as<int64_t>(), p->part.as<double>())
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
    if (p->n_multi > 0) {
        spmm_fixup_kernel<<<(unsigned)ceil_div((int64_t)p->n_split * WAVE, 256), 256, 0, s>>>(
            p->seg_off.as<int64_t>(), p->part_off.as<int64_t>(), p->split_rows.as<int32_t>(), p->n_split, k,
            p->part.as<double>(), dC, ldc);
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
