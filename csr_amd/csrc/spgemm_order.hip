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
//   2. every entry's place from its key, without a sort: the keys of a row are distinct numbers below the row's product
//      count, so a bitmap over product indices (in LDS, a window of 2^19 at a time) with a running population count gives
//      each key its rank among the row's keys -- a counting sort that never moves a key -- and the entry (column, value)
//      goes straight to row end - 1 - rank (so_place_kernel; so_place_short_kernel for the rows of few products).
//
// Values are not touched: each keeps the bits the product kernels gave it.  tests/test_gpu_ops.py compares the result
// with the reference's own raw arrays (tests/golden/spgemm.npz, c*_raw_*) bit for bit.
#include "common.h"

#include <algorithm>
#include <atomic>

namespace csrk {

static std::atomic<int> g_spgemm_order{-1};      // -1: follow CSRK_SPGEMM_ORDER; 0 ascending; 1 reference

bool spgemm_reference_order_wanted()
{
    const int o = g_spgemm_order.load();
    if (o >= 0) return o == 1;
    const char *e = getenv("CSRK_SPGEMM_ORDER");
    return e && (e[0] == 'r' || e[0] == 'R' || e[0] == '1');
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
//   COLS (the product has few enough columns for 6.125 B of LDS each): a bit per COLUMN of C says "discovered by an earlier
//   batch" -- such a product, nearly all of them, ends at the bit test; the others take an LDS atomic min of their index
//   inside the batch (an integer minimum: any order gives the same result; a read first, so that a column's later products
//   in the batch skip the atomic).  After the batch the columns with a minimum and no bit are final: each is discovered
//   exactly once, so its key is a plain store (the entry of C through a column -> position map built at the start of the
//   row), no atomic on memory at all.  The walk stops when every entry of the row has its key: a long row of a dense product
//   (the rows the kernel's time hangs on) has found all its columns 40 % of the way.
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
    // COLS: [seen: words][mn: ncols u32][pos: ncols u16]; else [cols: cols_cap int32]
    const int32_t words = COLS ? (ncols + 31) / 32 : 0;
    unsigned int *seen = so_lds, *mn = seen + words;
    unsigned short *pos = (unsigned short *)(mn + (COLS ? ncols : 0));
    int32_t *cols = (int32_t *)so_lds;
    __shared__ int64_t s_bs[SO_BATCH];
    __shared__ int64_t s_off[SO_BATCH + 1];          // (32 rows of B can hold more than 2^31 entries between them)
    __shared__ int32_t s_found;                      // COLS: entries of the row that have their key
    const int32_t i = row_list[blockIdx.x];          // (the rows of more than SO_SHORT products)
    if (i >= a_nrows) return;
    const int32_t c0 = c_rp[i], nc = c_rp[i + 1] - c0;
    const int64_t a0 = a_rp[i], a1 = a_rp[i + 1];
    if (nc == 0 || a1 == a0) return;
    const int tid = threadIdx.x;
    const int32_t *crow = c_ci + c0;
    const bool in_lds = !COLS && nc <= cols_cap;
    if (COLS) {
        if (tid == 0) s_found = 0;
        for (int32_t q = tid; q < words; q += SO_THREADS) seen[q] = q * 32 + 32 <= ncols ? 0u : ~0u << (ncols & 31);   // (no column past the last)
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
                    if (mn[k] > (unsigned int)pidx) atomicMin(&mn[k], (unsigned int)pidx);      // (a batch holds fewer than 2^32 products: checked by the host)
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
            int32_t found = 0;
            for (int32_t w = tid; w < words; w += SO_THREADS) {      // the batch's discoveries are final
                unsigned int open = ~seen[w], f = 0u;
                while (open) {
                    const int b = __builtin_ctz(open);
                    open &= open - 1;
                    const unsigned int m = mn[32 * w + b];
                    if (m == 0xffffffffu) continue;
                    f |= 1u << b;
                    key[c0 + pos[32 * w + b]] = (unsigned long long)(base + (int64_t)m);
                }
                if (f) {
                    seen[w] |= f;
                    found += __popc(f);
                }
            }
            if (found) atomicAdd(&s_found, found);
        }
        base += total;
        __syncthreads();
        if (COLS && s_found == nc) break;            // (every thread reads it after the barrier, and the next write is two barriers on)
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

// The rows of more than SO_SHORT products, longest first: the workgroup kernels take a row each, and a 1.2 M-product row
// started late is the kernel's whole tail (rows by the octave of their product count: `fill` false counts the octaves,
// true places the rows behind the cursors the host made of the counts; inside an octave any order).
constexpr int SO_OCTAVES = 64;
__global__ __launch_bounds__(256) void so_long_rows_kernel(const int64_t *__restrict__ tp, int32_t nrows, bool fill,
                                                          int32_t *__restrict__ cursor, int32_t *__restrict__ list)
{
    __shared__ int32_t s_n[SO_OCTAVES], s_base[SO_OCTAVES];
    if (threadIdx.x < SO_OCTAVES) s_n[threadIdx.x] = 0;
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool is_long = r < nrows && tp[r] > SO_SHORT;
    const int oct = is_long ? __clzll((long long)tp[r]) : 0;      // (longest rows: smallest count of leading zeros)
    int32_t at = 0;
    if (is_long) at = atomicAdd(&s_n[oct], 1);
    __syncthreads();
    if (threadIdx.x < SO_OCTAVES && s_n[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], s_n[threadIdx.x]);
    if (!fill) return;
    __syncthreads();
    if (is_long) list[s_base[oct] + at] = (int32_t)r;
}

// Every entry's place from its key.  The keys of row i are distinct numbers in [0, tp[i]): one bit per product index in LDS,
// SO_WIN of them at a time, each word's population count before it, and a key's rank among the row's keys is the count of
// bits below its own.  The entry goes to row end - 1 - rank: the reference's order (last discovered first).  One workgroup
// per listed row.  *bad is raised by a key outside the row's products (an entry never discovered) or two equal keys.
constexpr int SO_WIN_WORDS = 16384;          // 2^19 product indices per window: 64 KB of bits + 64 KB of counts
constexpr int SO_WIN = SO_WIN_WORDS * 32;
constexpr int SO_OWN = SO_WIN_WORDS / SO_THREADS;      // words a thread counts
__global__ __launch_bounds__(SO_THREADS) void so_place_kernel(const int32_t *__restrict__ c_rp, const int32_t *__restrict__ c_ci,
                                                             const double *__restrict__ c_vs, const int64_t *__restrict__ tp,
                                                             const int32_t *__restrict__ row_list,
                                                             const unsigned long long *__restrict__ key,
                                                             int32_t *__restrict__ oci, double *__restrict__ ovs,
                                                             unsigned int *__restrict__ bad)
{
    extern __shared__ unsigned int so_lds[];
    unsigned int *bits = so_lds, *before = so_lds + SO_WIN_WORDS;
    __shared__ int32_t s_wave[SO_THREADS / WAVE];
    const int32_t i = row_list[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    const int32_t c0 = c_rp[i], nc = c_rp[i + 1] - c0;
    const int64_t t = tp[i];
    int32_t base = 0;                                // keys below the window
    for (int64_t w0 = 0; w0 < t; w0 += SO_WIN) {
        const int32_t used = (int32_t)(((t - w0 < SO_WIN ? t - w0 : (int64_t)SO_WIN) + 31) / 32);
        for (int32_t w = tid; w < used; w += SO_THREADS) bits[w] = 0u;
        __syncthreads();
        for (int32_t q = tid; q < nc; q += SO_THREADS) {
            const unsigned long long k = key[c0 + q];
            if (k >= (unsigned long long)t) {
                if (w0 == 0) atomicMax(bad, 1u);
                continue;
            }
            if (k - (unsigned long long)w0 < (unsigned long long)SO_WIN) atomicOr(&bits[(k - w0) >> 5], 1u << ((k - w0) & 31));
        }
        __syncthreads();
        // counts: a thread's SO_OWN consecutive words, then the threads of a wavefront, then the wavefronts
        int32_t mine = 0;
        const int32_t w_first = tid * SO_OWN;
#pragma unroll
        for (int u = 0; u < SO_OWN; u++)
            if (w_first + u < used) mine += __popc(bits[w_first + u]);
        int32_t inc = mine;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const int32_t o = __shfl_up(inc, off, WAVE);
            if (lane >= off) inc += o;
        }
        if (lane == WAVE - 1) s_wave[wv] = inc;
        __syncthreads();
        int32_t run = base + inc - mine;
        int32_t all = 0;
#pragma unroll
        for (int v = 0; v < SO_THREADS / WAVE; v++) {
            if (v < wv) run += s_wave[v];
            all += s_wave[v];
        }
#pragma unroll
        for (int u = 0; u < SO_OWN; u++)
            if (w_first + u < used) {
                before[w_first + u] = (unsigned int)run;
                run += __popc(bits[w_first + u]);
            }
        __syncthreads();
        for (int32_t q = tid; q < nc; q += SO_THREADS) {
            const unsigned long long k = key[c0 + q];
            if (k >= (unsigned long long)t || k - (unsigned long long)w0 >= (unsigned long long)SO_WIN) continue;
            const unsigned int d = (unsigned int)(k - w0);
            const int32_t rank = (int32_t)before[d >> 5] + __popc(bits[d >> 5] & ((1u << (d & 31)) - 1u));
            const int64_t to = (int64_t)c0 + (nc - 1 - rank);
            oci[to] = c_ci[c0 + q];
            ovs[to] = c_vs[c0 + q];
        }
        base += all;
        __syncthreads();                             // (s_wave and the bitmap are written again)
    }
    if (tid == 0 && base != nc) atomicMax(bad, 1u);
}

// the rows of at most SO_SHORT products (and the rows of none: nothing to place): SO_SUB lanes per row, the whole bitmap of
// a row (128 words at most) and its counts in the group's own kilobyte of LDS, no barrier -- a group's lanes are lanes of
// one wavefront, whose LDS operations complete in program order
__global__ __launch_bounds__(256) void so_place_short_kernel(const int32_t *__restrict__ c_rp, const int32_t *__restrict__ c_ci,
                                                            const double *__restrict__ c_vs, const int64_t *__restrict__ tp,
                                                            int32_t nrows, const unsigned long long *__restrict__ key,
                                                            int32_t *__restrict__ oci, double *__restrict__ ovs,
                                                            unsigned int *__restrict__ bad)
{
    constexpr int WORDS = SO_SHORT / 32, OWN = WORDS / SO_SUB;
    __shared__ unsigned int s_bits[256 / SO_SUB][WORDS], s_before[256 / SO_SUB][WORDS];
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SO_SUB;
    const int lane = threadIdx.x & (SO_SUB - 1), grp = threadIdx.x / SO_SUB;
    if (i >= nrows) return;
    const int64_t t = tp[i];
    if (t > SO_SHORT) return;
    const int32_t c0 = c_rp[i], nc = c_rp[i + 1] - c0;
    if (nc == 0) return;
    unsigned int *bits = s_bits[grp], *before = s_before[grp];
    const int32_t used = (int32_t)((t + 31) / 32);
    for (int32_t w = lane; w < used; w += SO_SUB) bits[w] = 0u;
    __builtin_amdgcn_wave_barrier();
    for (int32_t q = lane; q < nc; q += SO_SUB) {
        const unsigned long long k = key[c0 + q];
        if (k >= (unsigned long long)t) {
            atomicMax(bad, 1u);
            continue;
        }
        atomicOr(&bits[k >> 5], 1u << (k & 31));
    }
    __builtin_amdgcn_wave_barrier();
    int32_t mine = 0;
#pragma unroll
    for (int u = 0; u < OWN; u++)
        if (lane * OWN + u < used) mine += __popc(bits[lane * OWN + u]);
    int32_t inc = mine;
#pragma unroll
    for (int off = 1; off < SO_SUB; off <<= 1) {
        const int32_t o = __shfl_up(inc, off, SO_SUB);
        if (lane >= off) inc += o;
    }
    const int32_t all = __shfl(inc, SO_SUB - 1, SO_SUB);
    int32_t run = inc - mine;
#pragma unroll
    for (int u = 0; u < OWN; u++)
        if (lane * OWN + u < used) {
            before[lane * OWN + u] = (unsigned int)run;
            run += __popc(bits[lane * OWN + u]);
        }
    __builtin_amdgcn_wave_barrier();
    if (all != nc) {                                 // (two equal keys, or a key out of range)
        if (lane == 0) atomicMax(bad, 1u);
        return;
    }
    for (int32_t q = lane; q < nc; q += SO_SUB) {
        const unsigned int d = (unsigned int)key[c0 + q];
        const int32_t rank = (int32_t)before[d >> 5] + __popc(bits[d >> 5] & ((1u << (d & 31)) - 1u));
        const int64_t to = (int64_t)c0 + (nc - 1 - rank);
        oci[to] = c_ci[c0 + q];
        ovs[to] = c_vs[c0 + q];
    }
}

// c = a b as the product kernels left it (ascending columns, int32 row pointers, float64 values): re-ordered in place
int spgemm_apply_reference_order(Matrix *a, Matrix *b, Matrix *c)
{
    const int64_t n = c->nnz;
    if (n <= 1 || c->nrows == 0 || a->nnz == 0) return CSRK_OK;
    CSRK_REQUIRE(!c->ptr64 && c->val_type == CSRK_VAL_F64, "product has an unexpected layout");
    DevBuf key, tp, flag, oci, ovs, long_rows, cursor;
    CSRK_TRY(key.alloc((size_t)n * 8));
    CSRK_TRY(tp.alloc((size_t)(a->nrows + 1) * 8));
    CSRK_TRY(flag.alloc(4));
    CSRK_TRY(long_rows.alloc((size_t)a->nrows * 4 + 4));
    CSRK_TRY(cursor.alloc(SO_OCTAVES * 4));
    CSRK_TRY(oci.alloc((size_t)n * 4));
    CSRK_TRY(ovs.alloc((size_t)n * 8));
    CSRK_HIP(hipMemsetAsync(cursor.p, 0, SO_OCTAVES * 4, nullptr));
    CSRK_HIP(hipMemsetAsync(key.p, 0xff, (size_t)n * 8, nullptr));
    CSRK_HIP(hipMemsetAsync(flag.p, 0, 4, nullptr));
    // LDS of the discovery kernel (the device's own limit decides).  By column -- 6.125 B per column of C -- when that fits,
    // every row of C has fewer than 65536 entries (16-bit positions; a row has at most ncols) and no 32 rows of B hold 2^32
    // entries between them (32-bit indices inside a batch); else that row of C's columns, as many as fit.
    int lds_max = 0, dev = 0;
    CSRK_HIP(hipGetDevice(&dev));
    CSRK_HIP(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    const int64_t budget = std::min<int64_t>(SO_LDS_BYTES, (int64_t)lds_max - 1024);
    const int64_t by_col = ((int64_t)c->ncols + 31) / 32 * 4 + (int64_t)c->ncols * 6 + 64;
    const bool cols_mode = c->ncols < 65536 && by_col <= budget && b->nnz < (1ll << 32) / SO_BATCH;
    const int32_t cols_cap = cols_mode ? 0 : (int32_t)std::max<int64_t>(0, budget / 4);
    const size_t lds = cols_mode ? (size_t)by_col : (size_t)cols_cap * 4;
    const size_t lds_place = (size_t)SO_WIN_WORDS * 8;
    CSRK_REQUIRE((int64_t)lds_place + 1024 <= lds_max, "device has too little LDS for the ordering pass");
    const unsigned ga = (unsigned)ceil_div((int64_t)a->nrows * WAVE, 256);
    const unsigned gr = (unsigned)ceil_div(a->nrows, 256);
    const unsigned gs = (unsigned)ceil_div((int64_t)a->nrows * SO_SUB, 256);
    int32_t n_long = 0;
    int32_t octaves[SO_OCTAVES];
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
        so_long_rows_kernel<<<gr, 256>>>(tp.as<int64_t>(), a->nrows, false, cursor.as<int32_t>(), nullptr);            \
        CSRK_LAUNCH_CHECK();                                                                                           \
        so_discover_short_kernel<PA, PB><<<gs, 256>>>((const PA *)a->d_rowptrs, a->d_colinds, a->nrows,                \
                                                      (const PB *)b->d_rowptrs, b->d_colinds, (const int32_t *)c->d_rowptrs, \
                                                      c->d_colinds, tp.as<int64_t>(), key.as<unsigned long long>());   \
        CSRK_LAUNCH_CHECK();                                                                                           \
        CSRK_HIP(hipMemcpy(octaves, cursor.p, sizeof octaves, hipMemcpyDeviceToHost));                                 \
        for (int o = 0; o < SO_OCTAVES; o++) {                                                                         \
            const int32_t cnt = octaves[o];                                                                            \
            octaves[o] = n_long;                                                                                       \
            n_long += cnt;                                                                                             \
        }                                                                                                              \
        if (n_long > 0) {                                                                                              \
            CSRK_HIP(hipMemcpyAsync(cursor.p, octaves, sizeof octaves, hipMemcpyHostToDevice, nullptr));               \
            so_long_rows_kernel<<<gr, 256>>>(tp.as<int64_t>(), a->nrows, true, cursor.as<int32_t>(),                   \
                                             long_rows.as<int32_t>());                                                 \
            CSRK_LAUNCH_CHECK();                                                                                       \
            if (cols_mode) DISCOVER_GO(PA, PB, true);                                                                  \
            else DISCOVER_GO(PA, PB, false);                                                                           \
            CSRK_LAUNCH_CHECK();                                                                                       \
        }                                                                                                              \
    } while (0)
    if (a->ptr64) {
        if (b->ptr64) DISCOVER(int64_t, int64_t);
        else DISCOVER(int64_t, int32_t);
    } else {
        if (b->ptr64) DISCOVER(int32_t, int64_t);
        else DISCOVER(int32_t, int32_t);
    }
#undef DISCOVER
#undef DISCOVER_GO
    // every entry to its place (rows of no entries or no products: nothing to move)
    so_place_short_kernel<<<gs, 256>>>((const int32_t *)c->d_rowptrs, c->d_colinds, (const double *)c->d_values, tp.as<int64_t>(),
                                       a->nrows, key.as<unsigned long long>(), oci.as<int32_t>(), ovs.as<double>(),
                                       flag.as<unsigned int>());
    CSRK_LAUNCH_CHECK();
    if (n_long > 0) {
        CSRK_HIP(hipFuncSetAttribute((const void *)so_place_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_place));
        so_place_kernel<<<(unsigned)n_long, SO_THREADS, lds_place>>>((const int32_t *)c->d_rowptrs, c->d_colinds,
                                                                     (const double *)c->d_values, tp.as<int64_t>(),
                                                                     long_rows.as<int32_t>(), key.as<unsigned long long>(),
                                                                     oci.as<int32_t>(), ovs.as<double>(), flag.as<unsigned int>());
        CSRK_LAUNCH_CHECK();
    }
    unsigned int bad = 0;
    CSRK_HIP(hipMemcpy(&bad, flag.p, 4, hipMemcpyDeviceToHost));      // (waits for the kernels)
    CSRK_REQUIRE(bad == 0, "an entry of the product has no product landing on it (internal error)");
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
