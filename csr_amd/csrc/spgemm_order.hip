// The reference's column order inside the rows of a sparse product, on request.
//
// _sym_mm (csr/kernels/numba/multiply.py:60-100) walks row i of A entry by entry and, for each, row j of B entry by
// entry; a column seen for the first time is pushed onto the FRONT of the row's linked list (:79-82), and the list is
// copied out front to back (:94-97).  So a row of the reference's product holds its columns in REVERSE order of first
// discovery.  libcsrk's SpGEMM kernels emit ascending columns; no test of the reference pins the
// order (every one densifies or sorts), but its raw output is what a caller gets from from_handle before sort_rows().
// With CSRK_SPGEMM_ORDER=reference (or csrk_spgemm_set_order(1)) this pass re-orders a finished product:
//
//   1. for every entry (i, k) of C its KEY: the index, in the reference's walk of row i's products, of the first product
//      that lands on it -- the product that discovers k.
//   2. every entry's place from its key, without a sort: the keys of a row are distinct numbers below the row's product
//      count, so a bitmap over product indices with a running population count gives each key its rank among the row's
//      keys -- a counting sort that never moves a key -- and the entry (column, value) goes straight to row end - 1 - rank.
//   Both in one kernel per row, the keys in LDS (so_tiny_kernel: sixteen lanes per row for the rows of few products;
//   so_walk_kernel: a workgroup per row), longest rows first.
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

constexpr int SO_TINY = 256;                 // rows of fewer products: SO_SUB lanes each
constexpr int SO_MID = 4096;                 // (2^12) rows of fewer products: a 256-thread workgroup each; the others 1024 threads
constexpr int SO_SUB = 16;
constexpr int SO_BATCH = 32;                 // entries of A's row walked between two barriers (<= WAVE)
constexpr int SO_UNROLL = 4;                 // products a thread has in flight
constexpr int SO_LDS_BYTES = 150 * 1024;
constexpr unsigned int SO_NONE = 0xffffffffu;

// The rows of at least SO_TINY products, longest first: the workgroup kernels take a row each, and a 1.2 M-product row
// started late is the kernel's whole tail (rows by the octave of their product count: `fill` false counts the octaves,
// true places the rows behind the cursors the host made of the counts; inside an octave any order).
constexpr int SO_OCTAVES = 64;
__global__ __launch_bounds__(256) void so_list_rows_kernel(const int64_t *__restrict__ tp, int32_t nrows, bool fill,
                                                          int32_t *__restrict__ cursor, int32_t *__restrict__ list)
{
    __shared__ int32_t s_n[SO_OCTAVES], s_base[SO_OCTAVES];
    if (threadIdx.x < SO_OCTAVES) s_n[threadIdx.x] = 0;
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool listed = r < nrows && tp[r] >= SO_TINY;
    const int oct = listed ? __clzll((long long)tp[r]) : 0;      // (longest rows: fewest leading zeros)
    int32_t at = 0;
    if (listed) at = atomicAdd(&s_n[oct], 1);
    __syncthreads();
    if (threadIdx.x < SO_OCTAVES && s_n[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], s_n[threadIdx.x]);
    if (!fill) return;
    __syncthreads();
    if (listed) list[s_base[oct] + at] = (int32_t)r;
}

// first position of the ascending `cols[0..n)` holding a column >= k (k is there: every product lands on an entry of C)
template <class T>
__device__ __forceinline__ int32_t so_find(const T *cols, int32_t n, int32_t k)
{
    int32_t lo = 0, hi = n;
    while (lo < hi) {
        const int32_t mid = lo + ((hi - lo) >> 1);
        if (cols[mid] < k) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// The rows of fewer than SO_TINY products -- 99 % of the rows of a sparse product: SO_SUB lanes per row (four rows per
// wavefront: the walk is a chain of dependent loads -- entry of A, extent of the row of B, its columns -- and only rows in
// flight hide it), the row of C's columns, a minimum per entry and the bitmap over its product indices in the group's own
// 2 KB of LDS: A's entries one after the other, the row of B over the lanes, bisection and an LDS atomic min; then the
// ranks and the entries to their places.  No barrier: a group's lanes are lanes of one wavefront, whose LDS operations
// complete in program order.  Nothing goes through memory but the product itself.
template <class PA, class PB>
__global__ __launch_bounds__(256) void so_tiny_kernel(const PA *__restrict__ a_rp, const int32_t *__restrict__ a_ci,
                                                     int32_t a_nrows, const PB *__restrict__ b_rp,
                                                     const int32_t *__restrict__ b_ci, const int32_t *__restrict__ c_rp,
                                                     const int32_t *__restrict__ c_ci, const double *__restrict__ c_vs,
                                                     const int64_t *__restrict__ tp, int32_t *__restrict__ oci,
                                                     double *__restrict__ ovs, unsigned int *__restrict__ bad)
{
    constexpr int GROUPS = 256 / SO_SUB, WORDS = SO_TINY / 32;
    __shared__ int32_t s_cols[GROUPS][SO_TINY];
    __shared__ unsigned int s_mn[GROUPS][SO_TINY], s_bits[GROUPS][WORDS], s_before[GROUPS][WORDS];
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SO_SUB;
    const int lane = threadIdx.x & (SO_SUB - 1), grp = threadIdx.x / SO_SUB;
    if (i >= a_nrows) return;
    const int64_t t = tp[i];
    if (t >= SO_TINY || t == 0) return;
    const int32_t c0 = c_rp[i], nc = c_rp[i + 1] - c0;
    if (nc == 0 || nc > t) {
        if (lane == 0) atomicMax(bad, 1u);           // (products and no entry, or more entries than products)
        return;
    }
    int32_t *cols = s_cols[grp];
    unsigned int *mn = s_mn[grp], *bits = s_bits[grp], *before = s_before[grp];
    for (int32_t q = lane; q < nc; q += SO_SUB) {
        cols[q] = c_ci[c0 + q];
        mn[q] = SO_NONE;
    }
    if (lane < WORDS) bits[lane] = 0u;
    __builtin_amdgcn_wave_barrier();
    int32_t base = 0;
    for (int64_t e = a_rp[i]; e < (int64_t)a_rp[i + 1]; e++) {
        const int32_t j = a_ci[e];
        const int64_t bs = b_rp[j];
        const int32_t len = (int32_t)((int64_t)b_rp[j + 1] - bs);
        for (int32_t tt = lane; tt < len; tt += SO_SUB) {
            const int32_t lo = so_find(cols, nc, b_ci[bs + tt]);
            const unsigned int cand = (unsigned int)(base + tt);
            if (mn[lo] > cand) atomicMin(&mn[lo], cand);
        }
        base += len;
    }
    __builtin_amdgcn_wave_barrier();
    for (int32_t q = lane; q < nc; q += SO_SUB) {
        const unsigned int m = mn[q];
        if (m < (unsigned int)t) atomicOr(&bits[m >> 5], 1u << (m & 31));
    }
    __builtin_amdgcn_wave_barrier();
    const int32_t mine = lane < WORDS ? __popc(bits[lane]) : 0;
    int32_t inc = mine;
#pragma unroll
    for (int off = 1; off < SO_SUB; off <<= 1) {
        const int32_t o = __shfl_up(inc, off, SO_SUB);
        if (lane >= off) inc += o;
    }
    const int32_t all = __shfl(inc, SO_SUB - 1, SO_SUB);
    if (lane < WORDS) before[lane] = (unsigned int)(inc - mine);
    __builtin_amdgcn_wave_barrier();
    if (all != nc) {                                 // (an entry no product lands on)
        if (lane == 0) atomicMax(bad, 1u);
        return;
    }
    for (int32_t q = lane; q < nc; q += SO_SUB) {
        const unsigned int m = mn[q];
        const int32_t rank = (int32_t)before[m >> 5] + __popc(bits[m >> 5] & ((1u << (m & 31)) - 1u));
        const int64_t to = (int64_t)c0 + (nc - 1 - rank);
        oci[to] = cols[q];
        ovs[to] = c_vs[c0 + q];
    }
}

// The other rows: one workgroup per row of A walks that row's products IN ORDER (entries of A's row in order, for each the
// row of B in order: multiply.py:69-83), SO_BATCH entries of A at a time, their products flattened over the threads (a
// 20 000-entry row of B next to fifteen short ones costs every thread the same), and finds for every entry of the row of C
// its KEY: the index in that walk of the first product that lands on it -- the product that discovers its column.
//   COLS (the product has few enough columns for 6 B of LDS each): a bit per COLUMN of C says "discovered by an earlier
//   batch" -- such a product, nearly all of them, ends at the bit test; the others take an LDS atomic min of their index
//   inside the batch (an integer minimum: any order gives the same result; a read first, so that a column's later products
//   in the batch skip the atomic).  After the batch the columns with a minimum and no bit are final: each is discovered
//   exactly once.  The walk stops when every entry of the row has its key: a long row of a dense product (the rows the
//   kernel's time hangs on) has found all its columns 40 % of the way.
//   otherwise, the row of C's columns in LDS (`cap` of them fit) and a minimum per ENTRY: every product finds its entry by
//   bisection and takes the LDS atomic min of its index in the whole walk.
//   a row of more than `cap` entries, or of 2^32 products: bisection in memory and a 64-bit atomic min on key[] (initialised
//   here); so_place_kernel finishes the row.
// Then the places, without a sort: the keys of the row are distinct numbers below its product count, so a bitmap over
// product indices (win_words * 32 at a time) with the population count before each word gives a key its rank among the
// row's keys -- a counting sort that never moves a key -- and the entry (column, value) goes to row end - 1 - rank: the
// reference's order, last discovered first.  Keys never leave LDS.
template <class PA, class PB, int THREADS, bool COLS>
__global__ __launch_bounds__(THREADS) void so_walk_kernel(const PA *__restrict__ a_rp, const int32_t *__restrict__ a_ci,
                                                         const PB *__restrict__ b_rp, const int32_t *__restrict__ b_ci,
                                                         const int32_t *__restrict__ c_rp, const int32_t *__restrict__ c_ci,
                                                         const double *__restrict__ c_vs, const int64_t *__restrict__ tp,
                                                         int32_t ncols, int32_t cap, int32_t win_words,
                                                         const int32_t *__restrict__ row_list,
                                                         unsigned long long *__restrict__ key, int32_t *__restrict__ oci,
                                                         double *__restrict__ ovs, unsigned int *__restrict__ bad)
{
    extern __shared__ unsigned int so_lds[];
    // COLS: [seen: words][mn: ncols u32][pos: ncols u16]; else [cols: cap int32][mn: cap u32]; then [bits][before]: win_words each
    const int32_t words = COLS ? (ncols + 31) / 32 : 0;
    unsigned int *seen = so_lds;
    unsigned int *mn = COLS ? seen + words : so_lds + cap;
    int32_t *cols = (int32_t *)so_lds;
    unsigned short *pos = (unsigned short *)(mn + ncols);
    unsigned int *bits = COLS ? mn + ncols + (ncols + 1) / 2 : mn + cap;
    unsigned int *before = bits + win_words;
    __shared__ int64_t s_bs[SO_BATCH];
    __shared__ int64_t s_off[SO_BATCH + 1];          // (32 rows of B can hold more than 2^31 entries between them)
    __shared__ int32_t s_found;                      // COLS: entries of the row that have their key
    __shared__ int32_t s_wave[THREADS / WAVE];
    const int32_t i = row_list[blockIdx.x];
    const int32_t c0 = c_rp[i], nc = c_rp[i + 1] - c0;
    const int64_t a0 = a_rp[i], a1 = a_rp[i + 1];
    const int64_t t = tp[i];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    if (nc == 0) {
        if (tid == 0) atomicMax(bad, 1u);            // (a listed row has products)
        return;
    }
    const int32_t *crow = c_ci + c0;
    const bool in_memory = t >= 0xffffffffll || (!COLS && nc > cap);
    unsigned long long *krow = key + c0;
    if (in_memory) {
        for (int32_t q = tid; q < nc; q += THREADS) krow[q] = ~0ull;
    } else if (COLS) {
        if (tid == 0) s_found = 0;
        for (int32_t q = tid; q < words; q += THREADS) seen[q] = q * 32 + 32 <= ncols ? 0u : ~0u << (ncols & 31);   // (no column past the last)
        for (int32_t q = tid; q < ncols; q += THREADS) mn[q] = SO_NONE;
        for (int32_t q = tid; q < nc; q += THREADS) pos[crow[q]] = (unsigned short)q;
    } else {
        for (int32_t q = tid; q < nc; q += THREADS) {
            cols[q] = crow[q];
            mn[q] = SO_NONE;
        }
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
        // one at a time the kernel waits out a memory latency per product
        int q = 0;                                   // entry of the batch the thread's product belongs to: only ever grows
        for (int64_t p0 = tid; p0 < total; p0 += (int64_t)SO_UNROLL * THREADS) {
            int32_t kk[SO_UNROLL];
#pragma unroll
            for (int u = 0; u < SO_UNROLL; u++) {
                const int64_t pidx = p0 + (int64_t)u * THREADS;
                kk[u] = 0;
                if (pidx < total) {
                    while (s_off[q + 1] <= pidx) q++;
                    kk[u] = b_ci[s_bs[q] + (pidx - s_off[q])];
                }
            }
#pragma unroll
            for (int u = 0; u < SO_UNROLL; u++) {
                const int64_t pidx = p0 + (int64_t)u * THREADS;
                if (pidx >= total) break;
                const int32_t k = kk[u];
                if (in_memory) {
                    const int32_t lo = so_find(crow, nc, k);
                    const unsigned long long cand = (unsigned long long)(base + pidx);
                    if (krow[lo] > cand) atomicMin(&krow[lo], cand);      // (a read as a filter before the atomic)
                } else if (COLS) {
                    if ((seen[k >> 5] >> (k & 31)) & 1u) continue;
                    if (mn[k] > (unsigned int)pidx) atomicMin(&mn[k], (unsigned int)pidx);      // (a batch holds fewer than 2^32 products: checked by the host)
                } else {
                    const int32_t lo = so_find(cols, nc, k);
                    const unsigned int cand = (unsigned int)(base + pidx);
                    if (mn[lo] > cand) atomicMin(&mn[lo], cand);
                }
            }
        }
        if (COLS && !in_memory) {
            __syncthreads();
            int32_t found = 0;
            for (int32_t w = tid; w < words; w += THREADS) {      // the batch's discoveries are final
                unsigned int open = ~seen[w], f = 0u;
                while (open) {
                    const int b = __builtin_ctz(open);
                    open &= open - 1;
                    const unsigned int m = mn[32 * w + b];
                    if (m == SO_NONE) continue;
                    f |= 1u << b;
                    mn[32 * w + b] = (unsigned int)(base + (int64_t)m);      // (its index in the whole walk)
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
        if (COLS && !in_memory && s_found == nc) break;      // (every thread reads it after the barrier, and the next write is two barriers on)
    }
    if (in_memory) return;
    // ---- the places
    const int64_t win = (int64_t)win_words * 32;
    const int own = (win_words + THREADS - 1) / THREADS;      // words a thread counts
    int32_t placed = 0;                              // keys below the window
    for (int64_t w0 = 0; w0 < t; w0 += win) {
        const int32_t used = (int32_t)(((t - w0 < win ? t - w0 : win) + 31) / 32);
        for (int32_t w = tid; w < used; w += THREADS) bits[w] = 0u;
        __syncthreads();
        for (int32_t q = tid; q < nc; q += THREADS) {
            const unsigned int k = COLS ? mn[crow[q]] : mn[q];
            if ((int64_t)k - w0 >= 0 && (int64_t)k - w0 < win && (int64_t)k < t) atomicOr(&bits[(k - w0) >> 5], 1u << ((k - w0) & 31));
        }
        __syncthreads();
        // counts: a thread's `own` consecutive words, then the threads of a wavefront, then the wavefronts
        int32_t mine = 0;
        const int32_t w_first = tid * own;
        for (int u = 0; u < own; u++)
            if (w_first + u < used) mine += __popc(bits[w_first + u]);
        int32_t inc = mine;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const int32_t o = __shfl_up(inc, off, WAVE);
            if (lane >= off) inc += o;
        }
        if (lane == WAVE - 1) s_wave[wv] = inc;
        __syncthreads();
        int32_t run = placed + inc - mine;
        int32_t all = 0;
#pragma unroll
        for (int v = 0; v < THREADS / WAVE; v++) {
            if (v < wv) run += s_wave[v];
            all += s_wave[v];
        }
        for (int u = 0; u < own; u++)
            if (w_first + u < used) {
                before[w_first + u] = (unsigned int)run;
                run += __popc(bits[w_first + u]);
            }
        __syncthreads();
        for (int32_t q = tid; q < nc; q += THREADS) {
            const int32_t col = crow[q];
            const unsigned int k = COLS ? mn[col] : mn[q];
            if ((int64_t)k - w0 < 0 || (int64_t)k - w0 >= win || (int64_t)k >= t) continue;
            const unsigned int d = (unsigned int)(k - w0);
            const int32_t rank = (int32_t)before[d >> 5] + __popc(bits[d >> 5] & ((1u << (d & 31)) - 1u));
            const int64_t to = (int64_t)c0 + (nc - 1 - rank);
            oci[to] = col;
            ovs[to] = c_vs[c0 + q];
        }
        placed += all;
        if (placed >= nc) break;                     // (the same for every thread)
        __syncthreads();                             // (s_wave and the bitmap are written again)
    }
    if (tid == 0 && placed != nc) atomicMax(bad, 1u);      // (an entry no product lands on)
}

// The rows so_walk_kernel left in key[] (more entries than its LDS holds, or 2^32 products): every entry's place from its
// key as there, the bitmap a window of 2^19 product indices.  *bad is raised by a key outside the row's products (an entry
// never discovered) or two equal keys.
constexpr int SO_WIN_WORDS = 16384;          // 64 KB of bits + 64 KB of counts
constexpr int SO_WIN = SO_WIN_WORDS * 32;
constexpr int SO_PLACE_THREADS = 1024;
constexpr int SO_OWN = SO_WIN_WORDS / SO_PLACE_THREADS;      // words a thread counts
__global__ __launch_bounds__(SO_PLACE_THREADS) void so_place_kernel(const int32_t *__restrict__ c_rp, const int32_t *__restrict__ c_ci,
                                                             const double *__restrict__ c_vs, const int64_t *__restrict__ tp,
                                                             const int32_t *__restrict__ row_list, int32_t cap,
                                                             const unsigned long long *__restrict__ key,
                                                             int32_t *__restrict__ oci, double *__restrict__ ovs,
                                                             unsigned int *__restrict__ bad)
{
    constexpr int SO_THREADS = SO_PLACE_THREADS;
    extern __shared__ unsigned int so_lds[];
    unsigned int *bits = so_lds, *before = so_lds + SO_WIN_WORDS;
    __shared__ int32_t s_wave[SO_THREADS / WAVE];
    const int32_t i = row_list[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    const int32_t c0 = c_rp[i], nc = c_rp[i + 1] - c0;
    const int64_t t = tp[i];
    if (!(t >= 0xffffffffll || nc > cap)) return;    // (so_walk_kernel's own test: the row is in its place already)
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

// c = a b as the product kernels left it (ascending columns, int32 row pointers, float64 values): re-ordered in place
int spgemm_apply_reference_order(Matrix *a, Matrix *b, Matrix *c)
{
    const int64_t n = c->nnz;
    if (n <= 1 || c->nrows == 0 || a->nnz == 0) return CSRK_OK;
    CSRK_REQUIRE(!c->ptr64 && c->val_type == CSRK_VAL_F64, "product has an unexpected layout");
    DevBuf key, tp, flag, oci, ovs, rows, cursor;
    CSRK_TRY(key.alloc((size_t)n * 8));              // (written only by the rows that do not fit LDS)
    CSRK_TRY(tp.alloc((size_t)(a->nrows + 1) * 8));
    CSRK_TRY(flag.alloc(4));
    CSRK_TRY(rows.alloc((size_t)a->nrows * 4 + 4));
    CSRK_TRY(cursor.alloc(SO_OCTAVES * 4));
    CSRK_TRY(oci.alloc((size_t)n * 4));
    CSRK_TRY(ovs.alloc((size_t)n * 8));
    CSRK_HIP(hipMemsetAsync(cursor.p, 0, SO_OCTAVES * 4, nullptr));
    CSRK_HIP(hipMemsetAsync(flag.p, 0, 4, nullptr));
    // LDS of the 1024-thread walk (the device's own limit decides).  By column -- 6 B per column of C and a bit -- when that
    // leaves a window of 2^15 product indices at least, every row of C has fewer than 65536 entries (16-bit positions; a row
    // has at most ncols) and no 32 rows of B hold 2^32 entries between them (32-bit indices inside a batch); else by entry,
    // 8 B each, with a window of 2^16.  The 256-thread walk: rows of fewer than SO_MID products (and entries), by entry.
    int lds_max = 0, dev = 0;
    CSRK_HIP(hipGetDevice(&dev));
    CSRK_HIP(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    const int64_t budget = std::min<int64_t>(SO_LDS_BYTES, (int64_t)lds_max - 2048);
    const int64_t by_col = (((int64_t)c->ncols + 31) / 32 + c->ncols + ((int64_t)c->ncols + 1) / 2) * 4;
    int32_t win_cols = 0;
    for (int32_t w = 16384; w >= 1024; w >>= 1)
        if (by_col + (int64_t)w * 8 <= budget) {
            win_cols = w;
            break;
        }
    const bool cols_mode = c->ncols < 65536 && win_cols > 0 && b->nnz < (1ll << 32) / SO_BATCH;
    const int32_t win_long = cols_mode ? win_cols : 2048;
    const int32_t cap_long = cols_mode ? INT32_MAX : (int32_t)std::max<int64_t>(0, (budget - (int64_t)win_long * 8) / 8);
    const size_t lds_long = cols_mode ? (size_t)(by_col + (int64_t)win_long * 8) : (size_t)cap_long * 8 + (size_t)win_long * 8;
    const int32_t win_mid = SO_MID / 32, cap_mid = SO_MID;
    const size_t lds_mid = (size_t)cap_mid * 8 + (size_t)win_mid * 8;
    const size_t lds_place = (size_t)SO_WIN_WORDS * 8;
    CSRK_REQUIRE((int64_t)lds_place + 2048 <= lds_max && cap_long >= SO_MID, "device has too little LDS for the ordering pass");
    const unsigned ga = (unsigned)ceil_div((int64_t)a->nrows * WAVE, 256);
    const unsigned gr = (unsigned)ceil_div(a->nrows, 256);
    const unsigned gs = (unsigned)ceil_div((int64_t)a->nrows * SO_SUB, 256);
    int32_t n_long = 0, n_mid = 0;                   // listed rows of at least SO_MID products (first in the list), of fewer
    int32_t octaves[SO_OCTAVES];
    unsigned long long *const key_p = key.as<unsigned long long>();
    int32_t *const oci_p = oci.as<int32_t>();
    double *const ovs_p = ovs.as<double>();
    unsigned int *const bad_p = flag.as<unsigned int>();
    const int32_t *const c_rp = (const int32_t *)c->d_rowptrs;
    const double *const c_vs = (const double *)c->d_values;
#define WALK_GO(PA, PB, THREADS, COLS, GRID, LDS, CAP, WIN, LIST)                                                      \
    do {                                                                                                               \
        CSRK_HIP(hipFuncSetAttribute((const void *)so_walk_kernel<PA, PB, THREADS, COLS>,                              \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS)));                         \
        so_walk_kernel<PA, PB, THREADS, COLS><<<(unsigned)(GRID), THREADS, LDS>>>(                                     \
            (const PA *)a->d_rowptrs, a->d_colinds, (const PB *)b->d_rowptrs, b->d_colinds, c_rp, c->d_colinds, c_vs,  \
            tp.as<int64_t>(), c->ncols, CAP, WIN, LIST, key_p, oci_p, ovs_p, bad_p);                                   \
        CSRK_LAUNCH_CHECK();                                                                                           \
    } while (0)
#define ORDER(PA, PB)                                                                                                  \
    do {                                                                                                               \
        so_row_products_kernel<PA, PB><<<ga, 256>>>((const PA *)a->d_rowptrs, a->d_colinds, a->nrows,                  \
                                                    (const PB *)b->d_rowptrs, tp.as<int64_t>());                       \
        CSRK_LAUNCH_CHECK();                                                                                           \
        so_list_rows_kernel<<<gr, 256>>>(tp.as<int64_t>(), a->nrows, false, cursor.as<int32_t>(), nullptr);            \
        CSRK_LAUNCH_CHECK();                                                                                           \
        so_tiny_kernel<PA, PB><<<gs, 256>>>((const PA *)a->d_rowptrs, a->d_colinds, a->nrows, (const PB *)b->d_rowptrs, \
                                            b->d_colinds, c_rp, c->d_colinds, c_vs, tp.as<int64_t>(), oci_p, ovs_p, bad_p); \
        CSRK_LAUNCH_CHECK();                                                                                           \
        CSRK_HIP(hipMemcpy(octaves, cursor.p, sizeof octaves, hipMemcpyDeviceToHost));                                 \
        int32_t listed = 0;                                                                                            \
        for (int o = 0; o < SO_OCTAVES; o++) {       /* (63 - o = the octave: rows of [2^(63-o), 2^(64-o)) products) */ \
            const int32_t cnt = octaves[o];                                                                            \
            octaves[o] = listed;                                                                                       \
            listed += cnt;                                                                                             \
            if (63 - o >= 12) n_long = listed;       /* (SO_MID = 2^12) */                                             \
        }                                                                                                              \
        n_mid = listed - n_long;                                                                                       \
        if (listed > 0) {                                                                                              \
            CSRK_HIP(hipMemcpyAsync(cursor.p, octaves, sizeof octaves, hipMemcpyHostToDevice, nullptr));               \
            so_list_rows_kernel<<<gr, 256>>>(tp.as<int64_t>(), a->nrows, true, cursor.as<int32_t>(), rows.as<int32_t>()); \
            CSRK_LAUNCH_CHECK();                                                                                       \
        }                                                                                                              \
        if (n_long > 0) {                                                                                              \
            if (cols_mode) WALK_GO(PA, PB, 1024, true, n_long, lds_long, cap_long, win_long, rows.as<int32_t>());      \
            else WALK_GO(PA, PB, 1024, false, n_long, lds_long, cap_long, win_long, rows.as<int32_t>());               \
        }                                                                                                              \
        if (n_mid > 0) WALK_GO(PA, PB, 256, false, n_mid, lds_mid, cap_mid, win_mid, rows.as<int32_t>() + n_long);     \
    } while (0)
    if (a->ptr64) {
        if (b->ptr64) ORDER(int64_t, int64_t);
        else ORDER(int64_t, int32_t);
    } else {
        if (b->ptr64) ORDER(int32_t, int64_t);
        else ORDER(int32_t, int32_t);
    }
#undef ORDER
#undef WALK_GO
    if (n_long > 0) {                                // (the rows the walk left in key[]: the others leave at once)
        CSRK_HIP(hipFuncSetAttribute((const void *)so_place_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_place));
        so_place_kernel<<<(unsigned)n_long, SO_PLACE_THREADS, lds_place>>>(c_rp, c->d_colinds, c_vs, tp.as<int64_t>(),
                                                                           rows.as<int32_t>(), cap_long, key_p, oci_p, ovs_p, bad_p);
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
