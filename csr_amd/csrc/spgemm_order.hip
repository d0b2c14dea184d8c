// The reference's column order inside the rows of a sparse product, on request.
//
// _sym_mm (csr/kernels/numba/multiply.py:60-100) walks row i of A entry by entry and, for each, row j of B entry by
// entry; a column seen for the first time is pushed onto the FRONT of the row's linked list (:79-82), and the list is
// copied out front to back (:94-97).  So a row of the reference's product holds its columns in REVERSE order of first
// discovery.  libcsrk's SpGEMM kernels emit ascending columns; no test of the reference pins the
// order (every one densifies or sorts), but its raw output is what a caller gets from from_handle before sort_rows().
// With CSRK_SPGEMM_ORDER=reference (or csrk_spgemm_set_order(1)) this pass re-orders a finished product:
//
//   1. key[e] for every entry e = (i, k) of C: the index, in the reference's walk of row i's products, of the first
//      product that lands on it -- the product that discovers k (so_discover_kernel).
//   2. entries sorted by (row, key descending): the two are folded into one number below the product count of A B
//      (so_sortkey_kernel), and an index permutation is radix-sorted by its 30-bit digits (transpose.hip's passes: one
//      sort for up to 10^9 products), then one gather of columns and values.
//
// Values are not touched: each keeps the bits the product kernels gave it.  tests/test_gpu_ops.py compares the result
// with the reference's own raw arrays (tests/golden/spgemm.npz, c*_raw_*) bit for bit.
#include "common.h"

#include <algorithm>
#include <atomic>

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

// row of every entry of a CSR (one wavefront per row writes its extent)
template <class P>
__global__ void so_row_of_kernel(const P *__restrict__ rp, int32_t nrows, int32_t *__restrict__ row_of)
{
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (w >= nrows) return;
    const int64_t s = rp[w], e = rp[w + 1];
    for (int64_t k = s + lane; k < e; k += WAVE) row_of[k] = (int32_t)w;
}

// products of every row of A B: tp[i] = sum over the entries (i, j) of A of |B_j| (one wavefront per row)
template <class PA, class PB>
__global__ __launch_bounds__(256) void so_row_products_kernel(const PA *__restrict__ a_rp, const int32_t *__restrict__ a_ci,
                                                             int32_t a_nrows, const PB *__restrict__ b_rp, int64_t *__restrict__ tp)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (i >= a_nrows) return;
    int64_t n = 0;
    for (int64_t e = (int64_t)a_rp[i] + lane; e < (int64_t)a_rp[i + 1]; e += WAVE) {
        const int32_t j = a_ci[e];
        n += (int64_t)b_rp[j + 1] - (int64_t)b_rp[j];
    }
    for (int off = WAVE / 2; off; off >>= 1) n += __shfl_down(n, off, WAVE);
    if (lane == 0) tp[i] = n;
}

// key[e] = the index, in the reference's walk of row i's products (entries of A's row in order, for each the row of B in
// order: multiply.py:69-83), of the FIRST product that lands on entry e of C -- the product that discovers its column.
// One workgroup per row of A walks that row's products IN ORDER, SO_BATCH entries of A at a time, their products
// flattened over the threads (a 20 000-entry row of B next to fifteen short ones costs every thread the same).
//   COLS (the product has few enough columns for 6.25 B of LDS each): a bit per COLUMN of C says "discovered by an earlier
//   batch" -- such a product, nearly all of them, ends at the bit test; the others mark their column in the batch's own
//   bitmap and take an LDS atomic min of their index inside the batch (an integer minimum: any order gives the same
//   result).  After the batch the marked columns are final: each is discovered exactly once, so its key is a plain store
//   (the entry of C through a column -> position map built at the start of the row), no atomic on memory at all.
//   otherwise: every product finds its entry of C by bisection (that row of C's columns, in LDS when they fit) and sends
//   its index to a 64-bit atomic min on memory, with a read of the key as a filter.
constexpr int SO_THREADS = 1024;
constexpr int SO_BATCH = 32;                 // entries of A's row walked between two barriers (<= WAVE)
constexpr int SO_UNROLL = 4;                 // products a thread has in flight
constexpr int SO_LDS_BYTES = 150 * 1024;
template <class PA, class PB, bool COLS>
__global__ __launch_bounds__(SO_THREADS) void so_discover_kernel(const PA *__restrict__ a_rp, const int32_t *__restrict__ a_ci,
                                                                int32_t a_nrows, const PB *__restrict__ b_rp,
                                                                const int32_t *__restrict__ b_ci, const int32_t *__restrict__ c_rp,
                                                                const int32_t *__restrict__ c_ci, int32_t ncols, int32_t cols_cap,
                                                                const int32_t *__restrict__ row_list,
                                                                unsigned long long *__restrict__ key)
{
    extern __shared__ unsigned int so_lds[];
    // COLS: [seen: words][fresh: words][mn: ncols u32][pos: ncols u16]; else [cols: cols_cap int32]
    const int32_t words = COLS ? (ncols + 31) / 32 : 0;
    unsigned int *seen = so_lds, *fresh = seen + words, *mn = fresh + words;
    unsigned short *pos = (unsigned short *)(mn + (COLS ? ncols : 0));
    int32_t *cols = (int32_t *)so_lds;
    __shared__ int64_t s_bs[SO_BATCH];
    __shared__ int64_t s_off[SO_BATCH + 1];          // (32 rows of B can hold more than 2^31 entries between them)
    const int32_t i = row_list[blockIdx.x];          // (the rows of more than SO_SHORT products)
    if (i >= a_nrows) return;
    const int32_t c0 = c_rp[i], nc = c_rp[i + 1] - c0;
    const int64_t a0 = a_rp[i], a1 = a_rp[i + 1];
    if (nc == 0 || a1 == a0) return;
    const int tid = threadIdx.x;
    const int32_t *crow = c_ci + c0;
    const bool in_lds = !COLS && nc <= cols_cap;
    if (COLS) {
        for (int32_t q = tid; q < 2 * words; q += SO_THREADS) seen[q] = 0u;      // (both bitmaps)
        for (int32_t q = tid; q < ncols; q += SO_THREADS) mn[q] = 0xffffffffu;
        for (int32_t q = tid; q < nc; q += SO_THREADS) pos[crow[q]] = (unsigned short)q;
    } else if (in_lds) {
        for (int32_t q = tid; q < nc; q += SO_THREADS) cols[q] = crow[q];
    }
    __syncthreads();
    int64_t base = 0;                                // products before this batch
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
            int64_t inc = len;                       // inclusive scan of the lengths over the first SO_BATCH lanes
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
        // SO_UNROLL products per thread in flight: the walk is a chain of dependent loads (column of B -> bitmap word), and
        // one at a time the kernel waits out a memory latency per product (2.0 ms on the MovieLens block, 1.7 with four)
        int q = 0;                                   // entry of the batch the thread's product belongs to: only ever grows
        for (int64_t p0 = tid; p0 < total; p0 += (int64_t)SO_UNROLL * SO_THREADS) {
            int32_t kk[SO_UNROLL];
#pragma unroll
            for (int u = 0; u < SO_UNROLL; u++) {
                const int64_t pidx = p0 + (int64_t)u * SO_THREADS;
                kk[u] = 0;
                if (pidx < total) {
                    while (s_off[q + 1] <= pidx) q++;
                    kk[u] = b_ci[s_bs[q] + (pidx - s_off[q])];
                }
            }
#pragma unroll
            for (int u = 0; u < SO_UNROLL; u++) {
                const int64_t pidx = p0 + (int64_t)u * SO_THREADS;
                if (pidx >= total) break;
                const int32_t k = kk[u];
                if (COLS) {
                    if ((seen[k >> 5] >> (k & 31)) & 1u) continue;
                    atomicOr(&fresh[k >> 5], 1u << (k & 31));
                    atomicMin(&mn[k], (unsigned int)pidx);      // (a batch holds fewer than 2^32 products: checked by the host)
                    continue;
                }
                int32_t lo = 0, hi = nc;             // first position with column >= k
                if (in_lds) {
                    while (lo < hi) {
                        const int32_t mid = lo + ((hi - lo) >> 1);
                        if (cols[mid] < k) lo = mid + 1;
                        else hi = mid;
                    }
                } else {
                    while (lo < hi) {
                        const int32_t mid = lo + ((hi - lo) >> 1);
                        if (crow[mid] < k) lo = mid + 1;
                        else hi = mid;
                    }
                }
                const unsigned long long cand = (unsigned long long)(base + pidx);
                if (key[c0 + lo] > cand) atomicMin(&key[c0 + lo], cand);      // (a read as a filter before the atomic)
            }
        }
        if (COLS) {
            __syncthreads();
            for (int32_t w = tid; w < words; w += SO_THREADS) {      // the batch's discoveries are final
                unsigned int f = fresh[w];
                if (!f) continue;
                seen[w] |= f;
                fresh[w] = 0u;
                while (f) {
                    const int32_t k = 32 * w + __builtin_ctz(f);
                    f &= f - 1;
                    key[c0 + pos[k]] = (unsigned long long)(base + (int64_t)mn[k]);
                }
            }
        }
        base += total;
        __syncthreads();
    }
}

// The rows of at most SO_SHORT products -- most rows of a sparse product: a workgroup each would be sixteen wavefronts
// to start and three barriers to pass for a handful of products (12 ms of a 14.7-ms call on a power-law 1M x 1M product).
// SO_SUB lanes per row instead (four rows per wavefront: the walk is a chain of dependent loads -- entry of A, extent of the
// row of B, its columns, the bisection -- and only rows in flight hide it: a wavefront per row 4.2 ms on that product):
// A's entries one after the other, the row of B over the lanes, bisection in that row of C, the 64-bit atomic min with a
// read as a filter.
constexpr int SO_SHORT = 4096;
constexpr int SO_SUB = 16;
template <class PA, class PB>
__global__ __launch_bounds__(256) void so_discover_short_kernel(const PA *__restrict__ a_rp, const int32_t *__restrict__ a_ci,
                                                               int32_t a_nrows, const PB *__restrict__ b_rp,
                                                               const int32_t *__restrict__ b_ci, const int32_t *__restrict__ c_rp,
                                                               const int32_t *__restrict__ c_ci, const int64_t *__restrict__ tp,
                                                               unsigned long long *__restrict__ key)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SO_SUB;
    const int lane = threadIdx.x & (SO_SUB - 1);
    if (i >= a_nrows || tp[i] > SO_SHORT || tp[i] == 0) return;
    const int32_t c0 = c_rp[i], nc = c_rp[i + 1] - c0;
    const int32_t *crow = c_ci + c0;
    int64_t base = 0;
    for (int64_t e = a_rp[i]; e < (int64_t)a_rp[i + 1]; e++) {
        const int32_t j = a_ci[e];
        const int64_t bs = b_rp[j], len = (int64_t)b_rp[j + 1] - bs;
        for (int64_t t = lane; t < len; t += SO_SUB) {
            const int32_t k = b_ci[bs + t];
            int32_t lo = 0, hi = nc;
            while (lo < hi) {
                const int32_t mid = lo + ((hi - lo) >> 1);
                if (crow[mid] < k) lo = mid + 1;
                else hi = mid;
            }
            const unsigned long long cand = (unsigned long long)(base + t);
            if (key[c0 + lo] > cand) atomicMin(&key[c0 + lo], cand);
        }
        base += len;
    }
}

// the rows of more than SO_SHORT products, in any order (one atomic per workgroup of 256 rows)
__global__ __launch_bounds__(256) void so_long_rows_kernel(const int64_t *__restrict__ tp, int32_t nrows, int32_t *__restrict__ list,
                                                          int32_t *__restrict__ count)
{
    __shared__ int32_t s_n, s_base;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool is_long = r < nrows && tp[r] > SO_SHORT;
    int32_t at = 0;
    if (is_long) at = atomicAdd(&s_n, 1);
    __syncthreads();
    if (threadIdx.x == 0 && s_n) s_base = atomicAdd(count, s_n);
    __syncthreads();
    if (is_long) list[s_base + at] = (int32_t)r;
}

// g[e] = products of the rows before e's row + (products of its row - 1 - key[e]): ascending g = rows in order, inside a
// row the reference's order (last discovered first).  *bad is raised if an entry of C was never discovered.
__global__ void so_sortkey_kernel(const unsigned long long *__restrict__ key, const int32_t *__restrict__ row_of,
                                  const int64_t *__restrict__ pbase, int64_t n, unsigned long long *__restrict__ g,
                                  unsigned int *__restrict__ bad)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const unsigned long long k = key[e];
    const int32_t r = row_of[e];
    const int64_t t = pbase[r + 1] - pbase[r];
    if (k == ~0ull || (int64_t)k >= t) {
        atomicMax(bad, 1u);
        g[e] = 0;
        return;
    }
    g[e] = (unsigned long long)(pbase[r] + (t - 1 - (int64_t)k));
}

// keys[q] = 30-bit digit `d` of g[perm[q]] (perm == nullptr: the identity); ident (optional) receives the identity
__global__ void so_digit_kernel(const unsigned long long *__restrict__ g, const int32_t *__restrict__ perm, int64_t n, int d,
                                int32_t *__restrict__ keys, int32_t *__restrict__ ident)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    keys[q] = (int32_t)((g[perm ? perm[q] : q] >> (30 * d)) & 0x3fffffffull);
    if (ident) ident[q] = (int32_t)q;
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
    if (n <= 1 || c->nrows == 0 || a->nnz == 0) return CSRK_OK;
    CSRK_REQUIRE(!c->ptr64 && c->val_type == CSRK_VAL_F64, "product has an unexpected layout");
    DevBuf key, g, tp, permA, permB, keys, flag, oci, ovs, c_row, long_rows, n_long_d;
    CSRK_TRY(key.alloc((size_t)n * 8));
    CSRK_TRY(g.alloc((size_t)n * 8));
    CSRK_TRY(tp.alloc((size_t)(a->nrows + 1) * 8));
    CSRK_TRY(permA.alloc((size_t)n * 4));
    CSRK_TRY(permB.alloc((size_t)n * 4));
    CSRK_TRY(keys.alloc((size_t)n * 4));
    CSRK_TRY(c_row.alloc((size_t)n * 4));
    CSRK_TRY(flag.alloc(4));
    CSRK_TRY(long_rows.alloc((size_t)a->nrows * 4 + 4));
    CSRK_TRY(n_long_d.alloc(4));
    CSRK_HIP(hipMemsetAsync(n_long_d.p, 0, 4, nullptr));
    CSRK_TRY(oci.alloc((size_t)n * 4));
    CSRK_TRY(ovs.alloc((size_t)n * 8));
    CSRK_HIP(hipMemsetAsync(key.p, 0xff, (size_t)n * 8, nullptr));
    CSRK_HIP(hipMemsetAsync(flag.p, 0, 4, nullptr));
    so_row_of_kernel<int32_t><<<(unsigned)ceil_div((int64_t)c->nrows * WAVE, 256), 256>>>((const int32_t *)c->d_rowptrs, c->nrows,
                                                                                         c_row.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    // LDS of the discovery kernel (the device's own limit decides).  By column -- 6.25 B per column of C -- when that fits,
    // every row of C has fewer than 65536 entries (16-bit positions; a row has at most ncols) and no 32 rows of B hold 2^32
    // entries between them (32-bit indices inside a batch); else that row of C's columns, as many as fit.
    int lds_max = 0, dev = 0;
    CSRK_HIP(hipGetDevice(&dev));
    CSRK_HIP(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    const int64_t budget = std::min<int64_t>(SO_LDS_BYTES, (int64_t)lds_max - 1024);
    const int64_t by_col = ((int64_t)c->ncols + 31) / 32 * 8 + (int64_t)c->ncols * 6 + 64;
    const bool cols_mode = c->ncols < 65536 && by_col <= budget && b->nnz < (1ll << 32) / SO_BATCH;
    const int32_t cols_cap = cols_mode ? 0 : (int32_t)std::max<int64_t>(0, budget / 4);
    const size_t lds = cols_mode ? (size_t)by_col : (size_t)cols_cap * 4;
    const unsigned ga = (unsigned)ceil_div((int64_t)a->nrows * WAVE, 256);
#define DISCOVER_GO(PA, PB, COLS)                                                                                      \
    do {                                                                                                               \
        CSRK_HIP(hipFuncSetAttribute((const void *)so_discover_kernel<PA, PB, COLS>,                                   \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
        so_discover_kernel<PA, PB, COLS><<<(unsigned)n_long, SO_THREADS, lds>>>(                                       \
            (const PA *)a->d_rowptrs, a->d_colinds, a->nrows, (const PB *)b->d_rowptrs, b->d_colinds,                  \
            (const int32_t *)c->d_rowptrs, c->d_colinds, c->ncols, cols_cap, long_rows.as<int32_t>(),                  \
            key.as<unsigned long long>());                                                                             \
    } while (0)
#define DISCOVER(PA, PB)                                                                                               \
    do {                                                                                                               \
        so_row_products_kernel<PA, PB><<<ga, 256>>>((const PA *)a->d_rowptrs, a->d_colinds, a->nrows,                  \
                                                    (const PB *)b->d_rowptrs, tp.as<int64_t>());                       \
        CSRK_LAUNCH_CHECK();                                                                                           \
        so_long_rows_kernel<<<(unsigned)ceil_div(a->nrows, 256), 256>>>(tp.as<int64_t>(), a->nrows, long_rows.as<int32_t>(), \
                                                                      n_long_d.as<int32_t>());                        \
        CSRK_LAUNCH_CHECK();                                                                                           \
        so_discover_short_kernel<PA, PB><<<(unsigned)ceil_div((int64_t)a->nrows * SO_SUB, 256), 256>>>(               \
                                                      (const PA *)a->d_rowptrs, a->d_colinds, a->nrows,                \
                                                      (const PB *)b->d_rowptrs, b->d_colinds, (const int32_t *)c->d_rowptrs, \
                                                      c->d_colinds, tp.as<int64_t>(), key.as<unsigned long long>());   \
        CSRK_LAUNCH_CHECK();                                                                                           \
        CSRK_HIP(hipMemcpy(&n_long, n_long_d.p, 4, hipMemcpyDeviceToHost));                                            \
        if (n_long > 0) {                                                                                              \
            if (cols_mode) DISCOVER_GO(PA, PB, true);                                                                  \
            else DISCOVER_GO(PA, PB, false);                                                                           \
            CSRK_LAUNCH_CHECK();                                                                                       \
        }                                                                                                              \
    } while (0)
    int32_t n_long = 0;
    if (a->ptr64) {
        if (b->ptr64) DISCOVER(int64_t, int64_t);
        else DISCOVER(int64_t, int32_t);
    } else {
        if (b->ptr64) DISCOVER(int32_t, int64_t);
        else DISCOVER(int32_t, int32_t);
    }
#undef DISCOVER
#undef DISCOVER_GO
    CSRK_TRY(exclusive_scan_i64(tp.as<int64_t>(), tp.as<int64_t>(), a->nrows, nullptr));
    const unsigned gn = (unsigned)ceil_div(n, 256);
    so_sortkey_kernel<<<gn, 256>>>(key.as<unsigned long long>(), c_row.as<int32_t>(), tp.as<int64_t>(), n,
                                   g.as<unsigned long long>(), flag.as<unsigned int>());
    CSRK_LAUNCH_CHECK();
    int64_t total = 0;
    unsigned int bad = 0;
    CSRK_HIP(hipMemcpy(&total, tp.as<int64_t>() + a->nrows, 8, hipMemcpyDeviceToHost));
    CSRK_HIP(hipMemcpy(&bad, flag.p, 4, hipMemcpyDeviceToHost));
    CSRK_REQUIRE(bad == 0, "an entry of the product has no product landing on it (internal error)");
    // one stable sort of an index permutation per 30-bit digit of the sort key, least significant first (a block of
    // 4 * 10^8 products: one digit)
    int32_t *perm = nullptr;
    for (int d = 0; d == 0 || (total - 1) >> (30 * d) > 0; d++) {
        const int64_t top = (total - 1) >> (30 * d);                 // largest value of this and the higher digits
        const int32_t range = (int32_t)(top >= (1ll << 30) ? (1ll << 30) : top + 1);
        int32_t *ident = perm ? nullptr : permB.as<int32_t>();
        so_digit_kernel<<<gn, 256>>>(g.as<unsigned long long>(), perm, n, d, keys.as<int32_t>(), ident);
        CSRK_LAUNCH_CHECK();
        const int32_t *src = perm ? perm : permB.as<int32_t>();
        int32_t *dst = src == permA.as<int32_t>() ? permB.as<int32_t>() : permA.as<int32_t>();
        CSRK_TRY(stable_sort_payload_by_key(keys.as<int32_t>(), src, n, range, n, dst, nullptr));
        perm = dst;
    }
    so_apply_kernel<<<gn, 256>>>(perm, n, c->d_colinds, (const double *)c->d_values, oci.as<int32_t>(), ovs.as<double>());
    CSRK_LAUNCH_CHECK();
    CSRK_HIP(hipDeviceSynchronize());
    // the re-ordered arrays become the product's own (pool blocks both: the old ones go back with the DevBufs)
    if (c->owns) {
        void *old_ci = c->d_colinds, *old_vs = c->d_values;
        c->d_colinds = (int32_t *)oci.take();
        c->d_values = ovs.take();
        pool_free(old_ci);
        pool_free(old_vs);
    } else {
        CSRK_HIP(hipMemcpy(c->d_colinds, oci.p, (size_t)n * 4, hipMemcpyDeviceToDevice));
        CSRK_HIP(hipMemcpy(c->d_values, ovs.p, (size_t)n * 8, hipMemcpyDeviceToDevice));
    }
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
