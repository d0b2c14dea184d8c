// The reference's column order inside the rows of a sparse product, on request.
//
// _sym_mm (csr/kernels/numba/multiply.py:60-100) walks row i of A entry by entry and, for each, row j of B entry by
// entry; a column seen for the first time is pushed onto the FRONT of the row's linked list (:79-82), and the list is
// copied out front to back (:94-97).  So a row of the reference's product holds its columns in REVERSE order of first
// discovery.  libcsrk's SpGEMM kernels emit ascending columns; no test of the reference pins the
// order (every one densifies or sorts), but its raw output is what a caller gets from from_handle before sort_rows().
// With CSRK_SPGEMM_ORDER=reference (or csrk_spgemm_set_order(1)) this pass re-orders a finished product:
//
//   1. key[e] for every entry e = (i, k) of C: the smallest (position of the A entry inside row i, position inside
//      the row of B) over the products that land on it -- the product that discovers k.  One workgroup per row of A
//      walks the products; the entry of C is found by bisection (C's rows ascend), the key kept by a 64-bit atomic min
//      (an integer minimum: any order gives the same result).
//   2. entries sorted by (row, key descending): three stable radix sorts of an index permutation (transpose.hip's
//      passes) -- by B position, by A position, by row --, then one gather of columns and values.
//
// Values are not touched: each keeps the bits the product kernels gave it.  tests/test_gpu_ops.py compares the result
// with the reference's own raw arrays (tests/golden/spgemm.npz, c*_raw_*) bit for bit.
#include "common.h"

#include <atomic>
#include <mutex>

namespace csrk {

int stable_sort_payload_by_key(const int32_t *keys, const int32_t *payload, int64_t n, int32_t key_range,
                               int64_t payload_range, int32_t *out_payload, hipStream_t s);      // transpose.hip

static std::atomic<int> g_spgemm_order{-1};      // -1: follow CSRK_SPGEMM_ORDER; 0 ascending; 1 reference

bool spgemm_reference_order_wanted()
{
    const int o = g_spgemm_order.load();
    if (o >= 0) return o == 1;
    const char *e = getenv("CSRK_SPGEMM_ORDER");
    return e && (e[0] == 'r' || e[0] == 'R' || e[0] == '1');
}

template <class P>
__device__ __forceinline__ int64_t so_rp(const void *rp, int64_t i)
{
    return (int64_t)((const P *)rp)[i];
}

// row of every entry of a CSR (one thread per row writes its extent)
template <class P>
__global__ void so_row_of_kernel(const P *__restrict__ rp, int32_t nrows, int32_t *__restrict__ row_of)
{
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (w >= nrows) return;
    const int64_t s = rp[w], e = rp[w + 1];
    for (int64_t k = s + lane; k < e; k += WAVE) row_of[k] = (int32_t)w;
}

// One workgroup per ROW of A, that row of C's columns held in LDS (bisections out of LDS instead of 15 dependent L2 loads
// per product; one wavefront per A entry bisecting in global memory with an atomic per product took 27 ms on the
// MovieLens block A[2000] B[20000]^T: 3.9e8 products).
constexpr int SO_THREADS = 1024;
constexpr int SO_LDS_COLS = 24576;      // 96 KiB of columns + two bitmaps of 3 KiB
constexpr int SO_BATCH = 32;            // entries of A's row walked between two barriers (<= WAVE)
template <class PA, class PB>
__global__ __launch_bounds__(SO_THREADS) void so_first_row_kernel(const PA *__restrict__ a_rp, const int32_t *__restrict__ a_ci,
                                                                 int32_t a_nrows, const PB *__restrict__ b_rp,
                                                                 const int32_t *__restrict__ b_ci,
                                                                 const int32_t *__restrict__ c_rp,
                                                                 const int32_t *__restrict__ c_ci,
                                                                 unsigned long long *__restrict__ key)
{
    extern __shared__ int32_t so_cols[];                                  // SO_LDS_COLS columns, then the bitmaps
    unsigned int *seen = (unsigned int *)(so_cols + SO_LDS_COLS);         // discovered by an EARLIER entry of A's row
    unsigned int *fresh = seen + SO_LDS_COLS / 32;                        // discovered by the entry being walked
    const int32_t i = blockIdx.x;
    if (i >= a_nrows) return;
    const int32_t c0 = c_rp[i], nc = c_rp[i + 1] - c0;
    const int64_t a0 = a_rp[i], a1 = a_rp[i + 1];
    if (nc == 0 || a1 == a0) return;
    const int tid = threadIdx.x;
    if (nc <= SO_LDS_COLS) {
        // The workgroup walks the row's entries of A IN ORDER, a batch at a time, with a bit per entry of C: a product on
        // a column an EARLIER batch discovered stops at the bit test; the others (each column about once) send their key
        // to the atomic min, which settles the order inside a batch.  The bits of the batch being walked are kept apart
        // until it is finished: inside a batch a later product may run first.
        for (int32_t q = tid; q < nc; q += SO_THREADS) so_cols[q] = c_ci[c0 + q];
        for (int32_t q = tid; q < (nc + 31) / 32; q += SO_THREADS) seen[q] = 0u, fresh[q] = 0u;
        __syncthreads();
        // SO_BATCH entries of A at a time, their products FLATTENED over the workgroup's threads: one dependent chain
        // (column of A -> extent of the B row -> its columns) per batch instead of one per entry, and a 20 000-entry B
        // row next to fifteen short ones costs every thread the same (a wavefront per entry: 26 ms on the block above,
        // the popular items' rows setting the pace of every batch)
        __shared__ int64_t s_bs[SO_BATCH];
        __shared__ int64_t s_off[SO_BATCH + 1];      // (32 B rows can hold more than 2^31 entries between them)
        for (int64_t e0 = a0; e0 < a1; e0 += SO_BATCH) {
            if (tid < SO_BATCH) {
                const int64_t e = e0 + tid;
                int64_t bs = 0, len = 0;
                if (e < a1) {
                    const int32_t j = a_ci[e];
                    bs = b_rp[j];
                    len = (int64_t)b_rp[j + 1] - bs;
                }
                s_bs[tid] = bs;
                // inclusive scan of the lengths over the first SO_BATCH lanes (SO_BATCH <= WAVE)
                int64_t inc = len;
#pragma unroll
                for (int off = 1; off < SO_BATCH; off <<= 1) {
                    const int64_t o = __shfl_up(inc, off, WAVE);
                    if (tid >= off) inc += o;
                }
                s_off[tid + 1] = inc;
                if (tid == 0) s_off[0] = 0;
            }
            __syncthreads();
            const int64_t total = s_off[SO_BATCH];
            for (int64_t pidx = tid; pidx < total; pidx += SO_THREADS) {
                int q = 0;                               // entry of the batch this product belongs to
#pragma unroll
                for (int u = 1; u < SO_BATCH; u++) q += s_off[u] <= pidx;
                const int32_t tb = (int32_t)(pidx - s_off[q]);
                const int32_t k = b_ci[s_bs[q] + tb];
                int32_t lo = 0, hi = nc;                 // first position with column >= k
                while (lo < hi) {
                    const int32_t mid = lo + ((hi - lo) >> 1);
                    if (so_cols[mid] < k) lo = mid + 1;
                    else hi = mid;
                }
                if (lo < nc && so_cols[lo] == k && !((seen[lo >> 5] >> (lo & 31)) & 1u)) {
                    atomicOr(&fresh[lo >> 5], 1u << (lo & 31));
                    atomicMin(&key[c0 + lo], ((unsigned long long)(e0 + q - a0) << 32) | (unsigned long long)tb);
                }
            }
            __syncthreads();
            for (int32_t q = tid; q < (nc + 31) / 32; q += SO_THREADS) {
                const unsigned int f = fresh[q];
                if (f) seen[q] |= f, fresh[q] = 0u;
            }
            __syncthreads();
        }
        return;
    }
    // a row of C beyond the LDS budget: wavefronts take the entries of A in turn, bisection in global memory, a read of
    // the key as a filter before the atomic
    const int lane = tid & (WAVE - 1), wv = tid / WAVE;
    for (int64_t e = a0 + wv; e < a1; e += SO_THREADS / WAVE) {
        const int32_t j = a_ci[e];
        const unsigned long long ea = (unsigned long long)(e - a0) << 32;
        const int64_t bs = b_rp[j], be = b_rp[j + 1];
        for (int64_t t = bs + lane; t < be; t += WAVE) {
            const int32_t k = b_ci[t];
            int32_t lo = 0, hi = nc;
            while (lo < hi) {
                const int32_t mid = lo + ((hi - lo) >> 1);
                if (c_ci[c0 + mid] < k) lo = mid + 1;
                else hi = mid;
            }
            if (lo < nc && c_ci[c0 + lo] == k) {
                const unsigned long long cand = ea | (unsigned long long)(t - bs);
                if (key[c0 + lo] > cand) atomicMin(&key[c0 + lo], cand);
            }
        }
    }
}

// longest row of a CSR (upper bound of a position inside a row): block maximum, one atomic per workgroup
template <class P>
__global__ __launch_bounds__(256) void so_maxlen_kernel(const P *__restrict__ rp, int32_t nrows, unsigned int *__restrict__ out)
{
    __shared__ unsigned int s_m[256 / WAVE];
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long len = r < nrows ? (unsigned long long)((int64_t)rp[r + 1] - (int64_t)rp[r]) : 0ull;
    unsigned int m = len > 0xffffffffull ? 0xffffffffu : (unsigned int)len;
    for (int off = WAVE / 2; off > 0; off >>= 1) {
        const unsigned int o = __shfl_xor(m, off, WAVE);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & (WAVE - 1)) == 0) s_m[threadIdx.x / WAVE] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 256 / WAVE; w++) m = s_m[w] > m ? s_m[w] : m;
        if (m > *out) atomicMax(out, m);
    }
}

// the two halves of the keys and the identity permutation; *bad is raised if an entry of C was never discovered
__global__ void so_split_kernel(const unsigned long long *__restrict__ key, int64_t n, int32_t *__restrict__ hi_part,
                                int32_t *__restrict__ lo_part, int32_t *__restrict__ ident, unsigned int *__restrict__ bad)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const unsigned long long k = key[e];
    if (k == ~0ull) atomicMax(bad, 1u);
    hi_part[e] = (int32_t)(unsigned int)(k >> 32);
    lo_part[e] = (int32_t)(unsigned int)k;
    ident[e] = (int32_t)e;
}

// out[q] = range - 1 - src[perm[q]]  (descending order through an ascending sort), or src[perm[q]] itself
__global__ void so_gather_key_kernel(const int32_t *__restrict__ src, const int32_t *__restrict__ perm, int64_t n, int32_t flip,
                                     int32_t *__restrict__ out)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int32_t v = src[perm ? perm[q] : q];
    out[q] = flip >= 0 ? flip - v : v;
}

__global__ void so_apply_kernel(const int32_t *__restrict__ perm, int64_t n, const int32_t *__restrict__ ci,
                                const double *__restrict__ vs, int32_t *__restrict__ oci, double *__restrict__ ovs)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int32_t e = perm[q];
    oci[q] = ci[e];
    ovs[q] = vs[e];
}

// c = a b as the product kernels left it (ascending columns, int32 row pointers, float64 values): re-ordered in place
int spgemm_apply_reference_order(Matrix *a, Matrix *b, Matrix *c)
{
    const int64_t n = c->nnz;
    if (n <= 1 || c->nrows == 0) return CSRK_OK;
    CSRK_REQUIRE(!c->ptr64 && c->val_type == CSRK_VAL_F64, "product has an unexpected layout");
    DevBuf key, hi_part, lo_part, permA, permB, keys, maxes, oci, ovs, c_row;
    CSRK_TRY(key.alloc((size_t)n * 8));
    CSRK_TRY(hi_part.alloc((size_t)n * 4));
    CSRK_TRY(lo_part.alloc((size_t)n * 4));
    CSRK_TRY(permA.alloc((size_t)n * 4));
    CSRK_TRY(permB.alloc((size_t)n * 4));
    CSRK_TRY(keys.alloc((size_t)n * 4));
    CSRK_TRY(c_row.alloc((size_t)n * 4));
    CSRK_TRY(maxes.alloc(16));
    CSRK_TRY(oci.alloc((size_t)n * 4));
    CSRK_TRY(ovs.alloc((size_t)n * 8));
    CSRK_HIP(hipMemsetAsync(key.p, 0xff, (size_t)n * 8, nullptr));
    CSRK_HIP(hipMemsetAsync(maxes.p, 0, 16, nullptr));
    so_row_of_kernel<int32_t><<<(unsigned)ceil_div((int64_t)c->nrows * WAVE, 256), 256>>>((const int32_t *)c->d_rowptrs, c->nrows,
                                                                                         c_row.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    if (a->nnz > 0) {
        static std::once_flag once;
        static hipError_t attr_err = hipSuccess;
        std::call_once(once, [] {
            for (const void *f : {(const void *)so_first_row_kernel<int32_t, int32_t>, (const void *)so_first_row_kernel<int32_t, int64_t>,
                                  (const void *)so_first_row_kernel<int64_t, int32_t>, (const void *)so_first_row_kernel<int64_t, int64_t>}) {
                const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, SO_LDS_COLS * 4 + SO_LDS_COLS / 4);
                if (e != hipSuccess) attr_err = e;
            }
        });
        CSRK_HIP(attr_err);
#define FIRST(PA, PB)                                                                                                  \
    so_first_row_kernel<PA, PB><<<(unsigned)a->nrows, SO_THREADS, (size_t)SO_LDS_COLS * 4 + SO_LDS_COLS / 4>>>(                          \
        (const PA *)a->d_rowptrs, a->d_colinds, a->nrows, (const PB *)b->d_rowptrs, b->d_colinds,                      \
        (const int32_t *)c->d_rowptrs, c->d_colinds, key.as<unsigned long long>())
        if (a->ptr64) {
            if (b->ptr64) FIRST(int64_t, int64_t);
            else FIRST(int64_t, int32_t);
        } else {
            if (b->ptr64) FIRST(int32_t, int64_t);
            else FIRST(int32_t, int32_t);
        }
#undef FIRST
        CSRK_LAUNCH_CHECK();
    }
    const unsigned gn = (unsigned)ceil_div(n, 256);
    so_split_kernel<<<gn, 256>>>(key.as<unsigned long long>(), n, hi_part.as<int32_t>(), lo_part.as<int32_t>(),
                                 permA.as<int32_t>(), maxes.as<unsigned int>() + 2);
    CSRK_LAUNCH_CHECK();
    // positions inside a row of A / of B are below the longest row: the key ranges of the two sorts
    if (a->ptr64) so_maxlen_kernel<int64_t><<<(unsigned)ceil_div(a->nrows, 256), 256>>>((const int64_t *)a->d_rowptrs, a->nrows, maxes.as<unsigned int>());
    else so_maxlen_kernel<int32_t><<<(unsigned)ceil_div(a->nrows, 256), 256>>>((const int32_t *)a->d_rowptrs, a->nrows, maxes.as<unsigned int>());
    CSRK_LAUNCH_CHECK();
    if (b->ptr64) so_maxlen_kernel<int64_t><<<(unsigned)ceil_div(b->nrows, 256), 256>>>((const int64_t *)b->d_rowptrs, b->nrows, maxes.as<unsigned int>() + 1);
    else so_maxlen_kernel<int32_t><<<(unsigned)ceil_div(b->nrows, 256), 256>>>((const int32_t *)b->d_rowptrs, b->nrows, maxes.as<unsigned int>() + 1);
    CSRK_LAUNCH_CHECK();
    unsigned int mx[4] = {0, 0, 0, 0};
    CSRK_HIP(hipMemcpy(mx, maxes.p, 16, hipMemcpyDeviceToHost));
    CSRK_REQUIRE(mx[2] == 0, "an entry of the product has no product landing on it (internal error)");
    CSRK_REQUIRE(mx[0] < 0x7fffffffu && mx[1] < 0x7fffffffu, "row of A or B too long for the reference-order pass");
    mx[0] = mx[0] ? mx[0] - 1 : 0;      // largest position = longest row - 1
    mx[1] = mx[1] ? mx[1] - 1 : 0;
    // least significant key first: position inside the row of B (descending), position inside the row of A (descending), row
    so_gather_key_kernel<<<gn, 256>>>(lo_part.as<int32_t>(), nullptr, n, (int32_t)mx[1], keys.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(stable_sort_payload_by_key(keys.as<int32_t>(), permA.as<int32_t>(), n, (int32_t)mx[1] + 1, n, permB.as<int32_t>(), nullptr));
    so_gather_key_kernel<<<gn, 256>>>(hi_part.as<int32_t>(), permB.as<int32_t>(), n, (int32_t)mx[0], keys.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(stable_sort_payload_by_key(keys.as<int32_t>(), permB.as<int32_t>(), n, (int32_t)mx[0] + 1, n, permA.as<int32_t>(), nullptr));
    so_gather_key_kernel<<<gn, 256>>>(c_row.as<int32_t>(), permA.as<int32_t>(), n, -1, keys.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(stable_sort_payload_by_key(keys.as<int32_t>(), permA.as<int32_t>(), n, c->nrows, n, permB.as<int32_t>(), nullptr));
    so_apply_kernel<<<gn, 256>>>(permB.as<int32_t>(), n, c->d_colinds, (const double *)c->d_values, oci.as<int32_t>(),
                                 ovs.as<double>());
    CSRK_LAUNCH_CHECK();
    CSRK_HIP(hipMemcpyAsync(c->d_colinds, oci.p, (size_t)n * 4, hipMemcpyDeviceToDevice, nullptr));
    CSRK_HIP(hipMemcpyAsync(c->d_values, ovs.p, (size_t)n * 8, hipMemcpyDeviceToDevice, nullptr));
    CSRK_HIP(hipDeviceSynchronize());
    return CSRK_OK;
}

}  // namespace csrk

using namespace csrk;

extern "C" int csrk_spgemm_set_order(int order)
{
    CSRK_REQUIRE(order >= -1 && order <= 1, "order must be -1 (environment), 0 (ascending) or 1 (reference)");
    g_spgemm_order.store(order);
    return CSRK_OK;
}
