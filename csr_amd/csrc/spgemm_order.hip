// The reference's column order inside the rows of a sparse product (the default since round 5).
//
// _sym_mm (csr/kernels/numba/multiply.py:60-100) walks row i of A entry by entry and, for each, row j of B entry by
// entry; a column seen for the first time is pushed onto the FRONT of the row's linked list (:79-82), and the list is
// copied out front to back (:94-97).  So a row of the reference's product holds its columns in REVERSE order of first
// discovery.  libcsrk's SpGEMM kernels emit ascending columns; no test of the reference pins the
// order (every one densifies or sorts), but its raw output is what a caller gets from from_handle before sort_rows().
// Unless CSRK_SPGEMM_ORDER=ascending (or csrk_spgemm_set_order(0)) asks for the kernels' own order, this pass re-orders a
// finished product:
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
#include <type_traits>

namespace csrk {

static std::atomic<int> g_spgemm_order{-1};      // -1: follow CSRK_SPGEMM_ORDER; 0 ascending; 1 reference

// the reference's order unless the caller (csrk_spgemm_set_order(0)) or the environment (CSRK_SPGEMM_ORDER=ascending) asks
// for ascending columns
bool spgemm_reference_order_wanted()
{
    const int o = g_spgemm_order.load();
    if (o >= 0) return o == 1;
    const char *e = getenv("CSRK_SPGEMM_ORDER");
    return !(e && (e[0] == 'a' || e[0] == 'A' || e[0] == '0'));
}

// products of every row of A B: tp[i] = sum over the entries (i, j) of A of |B_j| (SUB lanes per row: sixteen when A's rows
// are short, as most rows of a sparse matrix are; a wavefront for a ratings matrix's)
template <class PA, class PB, int SUB>
__global__ __launch_bounds__(256) void so_row_products_kernel(const PA *__restrict__ a_rp, const int32_t *__restrict__ a_ci,
                                                             int32_t a_nrows, const PB *__restrict__ b_rp, int64_t *__restrict__ tp)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SUB;
    const int lane = threadIdx.x & (SUB - 1);
    if (i >= a_nrows) return;
    int64_t n = 0;
    const int64_t e1 = a_rp[i + 1];
    for (int64_t e = (int64_t)a_rp[i] + lane; e < e1; e += 4 * SUB) {      // (four entries in flight: a 7000-entry row is a chain of
        int32_t j[4];                                                      // dependent loads otherwise, 46 us for a 500-row block)
#pragma unroll
        for (int u = 0; u < 4; u++) j[u] = e + u * SUB < e1 ? a_ci[e + u * SUB] : -1;
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (j[u] >= 0) n += (int64_t)b_rp[j[u] + 1] - (int64_t)b_rp[j[u]];
    }
    for (int off = SUB / 2; off; off >>= 1) n += __shfl_down(n, off, SUB);
    if (lane == 0) tp[i] = n;
}

constexpr int SO_LEAST = 64;                 // (2^6) rows of fewer products: SO_SUB lanes each, straight from the row number
constexpr int SO_TINY = 256;                 // (2^8) rows of fewer products: SO_SUB lanes each, from the list
constexpr int SO_WAVE = 1024;                // (2^10) rows of fewer products: a wavefront each, from the list
constexpr int SO_MID = 4096;                 // (2^12) rows of fewer products: a 256-thread workgroup each; the others 1024 threads
constexpr int SO_SUB = 16;
constexpr int SO_UNROLL = 8;                 // products a thread has in flight
constexpr int SO_CHUNK = WAVE * SO_UNROLL;     // consecutive products a wavefront takes at a time
constexpr int SO_PLACE_UNROLL = 8;           // entries a thread has in flight when they are placed
constexpr int SO_LDS_BYTES = 150 * 1024;
constexpr unsigned int SO_NONE = 0xffffffffu;

// Diagnostic build only (-DCSRK_SO_STAMPS): clocks of thread 0 of every so_walk_kernel workgroup, summed per phase
// (0 tables, 1 batch tables, 2 walk, 3 a batch's discoveries, 4 places: bitmap, 5 counts, 6 moves; 7 workgroups)
#ifdef CSRK_SO_STAMPS
__device__ unsigned long long g_so_stamps[8];
#define SO_STAMP(I) { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_so_stamps[I], now_ - st_last); st_last = now_; } }
#else
#define SO_STAMP(I)
#endif

// The rows of at least SO_LEAST products, longest first: the workgroup kernels take a row each, and a 1.2 M-product row
// started late is the kernel's whole tail (rows by the octave of their product count: `fill` false counts the octaves,
// true places the rows behind the cursors the host made of the counts; inside an octave any order).  The count also
// raises cursor[SO_OCTAVES] if some row will need its keys in memory (more than `cap` entries, or 2^32 products: only then
// does the host allocate the key array).
constexpr int SO_OCTAVES = 64;
__global__ __launch_bounds__(256) void so_list_rows_kernel(const int64_t *__restrict__ tp, int32_t nrows, bool fill,
                                                          const int32_t *__restrict__ c_rp, int32_t cap,
                                                          int32_t *__restrict__ cursor, int32_t *__restrict__ list)
{
    __shared__ int32_t s_n[SO_OCTAVES], s_base[SO_OCTAVES];
    if (threadIdx.x < SO_OCTAVES) s_n[threadIdx.x] = 0;
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool listed = r < nrows && tp[r] >= SO_LEAST;
    const int oct = listed ? __clzll((long long)tp[r]) : 0;      // (longest rows: fewest leading zeros)
    if (!fill && listed && (tp[r] >= 0xffffffffll || c_rp[r + 1] - c_rp[r] > cap)) cursor[SO_OCTAVES] = 1;
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

// The rows of fewer than SO_WAVE products -- 99.6 % of the rows of a sparse product: SUB lanes per row (sixteen: four rows per
// wavefront; rows of SO_TINY products and more a wavefront each), the row of C's columns, a minimum per entry and the bitmap over its product indices in the group's own LDS
// (CAPT products at most: the kernel runs once for the rows of [1, SO_LEAST) products -- 95 % of the rows, half a kilobyte
// each, a full complement of wavefronts --, once for the listed rows of [SO_LEAST, SO_TINY) and once, a wavefront per row, for
// those of [SO_TINY, SO_WAVE): a workgroup per row left four such rows in flight on a CU, each waiting out its own barriers).  The walk is a chain of dependent loads -- entry of A,
// extent of the row of B, its columns -- so SUB entries of A are taken at a time, one per lane, and their products
// flattened over the lanes: every load of a step is in flight together.  A product finds its entry by bisection and takes
// an LDS atomic min; then the ranks, and the entries to their places.  No barrier: a group's lanes are lanes of one
// wavefront, whose LDS operations complete in program order.  Nothing goes through memory but the product itself.
template <class PA, class PB, int CAPT, int SUB>
__global__ __launch_bounds__(256) void so_tiny_kernel(const PA *__restrict__ a_rp, const int32_t *__restrict__ a_ci,
                                                     int32_t a_nrows, const PB *__restrict__ b_rp,
                                                     const int32_t *__restrict__ b_ci, const int32_t *__restrict__ c_rp,
                                                     const int32_t *__restrict__ c_ci, const double *__restrict__ c_vs,
                                                     const int64_t *__restrict__ tp, const int32_t *__restrict__ row_list,
                                                     int32_t n_list, int32_t *__restrict__ oci, double *__restrict__ ovs,
                                                     unsigned int *__restrict__ bad)
{
    constexpr int GROUPS = 256 / SUB, WORDS = CAPT / 32;
    static_assert(WORDS >= 1 && WORDS <= SUB, "a lane counts one word of the bitmap");
    __shared__ int32_t s_cols[GROUPS][CAPT];
    __shared__ unsigned int s_mn[GROUPS][CAPT], s_bits[GROUPS][WORDS], s_before[GROUPS][WORDS];
    __shared__ int32_t s_end[GROUPS][SUB];        // products up to and including each entry of the step
    __shared__ int64_t s_bs[GROUPS][SUB];
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / SUB;
    const int lane = threadIdx.x & (SUB - 1), grp = threadIdx.x / SUB;
    if (g >= (row_list ? n_list : a_nrows)) return;
    const int64_t i = row_list ? row_list[g] : g;    // (no list: every row, the kernel takes those of [1, CAPT) products)
    const int64_t t = tp[i];
    if (t >= CAPT || t == 0) return;
    const int32_t c0 = c_rp[i], nc = c_rp[i + 1] - c0;
    if (nc == 0 || nc > t) {
        if (lane == 0) atomicMax(bad, 1u);           // (products and no entry, or more entries than products)
        return;
    }
    int32_t *cols = s_cols[grp], *ends = s_end[grp];
    unsigned int *mn = s_mn[grp], *bits = s_bits[grp], *before = s_before[grp];
    for (int32_t q = lane; q < nc; q += SUB) {
        cols[q] = c_ci[c0 + q];
        mn[q] = SO_NONE;
    }
    if (lane < WORDS) bits[lane] = 0u;
    int32_t base = 0;
    const int64_t a1 = a_rp[i + 1];
    for (int64_t e0 = a_rp[i]; e0 < a1; e0 += SUB) {
        int64_t bs = 0;
        int32_t len = 0;
        if (e0 + lane < a1) {
            const int32_t j = a_ci[e0 + lane];
            bs = b_rp[j];
            len = (int32_t)((int64_t)b_rp[j + 1] - bs);
        }
        int32_t inc = len;
#pragma unroll
        for (int off = 1; off < SUB; off <<= 1) {
            const int32_t o = __shfl_up(inc, off, SUB);
            if (lane >= off) inc += o;
        }
        const int32_t total = __shfl(inc, SUB - 1, SUB);
        __builtin_amdgcn_wave_barrier();             // (the step before has read ends[] and s_bs[])
        ends[lane] = inc;
        s_bs[grp][lane] = bs;
        __builtin_amdgcn_wave_barrier();
        for (int32_t p = lane; p < total; p += SUB) {
            int q = 0;                               // first entry whose products end after p
#pragma unroll
            for (int step = SUB / 2; step; step >>= 1)
                if (ends[q + step - 1] <= p) q += step;
            const int32_t first = q ? ends[q - 1] : 0;
            const int32_t lo = so_find(cols, nc, b_ci[s_bs[grp][q] + (p - first)]);
            const unsigned int cand = (unsigned int)(base + p);
            if (mn[lo] > cand) atomicMin(&mn[lo], cand);
        }
        base += total;
    }
    __builtin_amdgcn_wave_barrier();
    for (int32_t q = lane; q < nc; q += SUB) {
        const unsigned int m = mn[q];
        if (m < (unsigned int)t) atomicOr(&bits[m >> 5], 1u << (m & 31));
    }
    __builtin_amdgcn_wave_barrier();
    const int32_t mine = lane < WORDS ? __popc(bits[lane]) : 0;
    int32_t inc = mine;
#pragma unroll
    for (int off = 1; off < SUB; off <<= 1) {
        const int32_t o = __shfl_up(inc, off, SUB);
        if (lane >= off) inc += o;
    }
    const int32_t all = __shfl(inc, SUB - 1, SUB);
    if (lane < WORDS) before[lane] = (unsigned int)(inc - mine);
    __builtin_amdgcn_wave_barrier();
    if (all != nc) {                                 // (an entry no product lands on)
        if (lane == 0) atomicMax(bad, 1u);
        return;
    }
    for (int32_t q = lane; q < nc; q += SUB) {
        const unsigned int m = mn[q];
        const int32_t rank = (int32_t)before[m >> 5] + __popc(bits[m >> 5] & ((1u << (m & 31)) - 1u));
        const int64_t to = (int64_t)c0 + (nc - 1 - rank);
        oci[to] = cols[q];
        ovs[to] = c_vs[c0 + q];
    }
}

// The other rows: one workgroup per row of A walks that row's products IN ORDER (entries of A's row in order, for each the
// row of B in order: multiply.py:69-83), a batch of one entry of A per thread at a time, their products flattened over the threads (a
// 20 000-entry row of B next to fifteen short ones costs every thread the same), and finds for every entry of the row of C
// its KEY: the index in that walk of the first product that lands on it -- the product that discovers its column.
//   COLS (the product has few enough columns for 4 B of LDS each): a bit per COLUMN of C says "discovered by an earlier
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
    // COLS: [seen: words][mn: ncols u32]; else [cols: cap int32][mn: cap u32]; then [bits][before]: win_words each; COLS: [spre: words]
    const int32_t words = COLS ? (ncols + 31) / 32 : 0;
    unsigned int *seen = so_lds;
    unsigned int *mn = COLS ? seen + words : so_lds + cap;
    int32_t *cols = (int32_t *)so_lds;
    unsigned int *bits = COLS ? mn + ncols : mn + cap;
    unsigned int *before = bits + win_words;
    unsigned int *spre = before + win_words;         // COLS: entries of the row before each word of `seen`
    __shared__ int64_t s_bs[THREADS];                // the batch: where each entry's row of B starts,
    __shared__ int64_t s_off[THREADS + 1];           // and the products before it
    __shared__ int64_t s_wave64[THREADS / WAVE];
    __shared__ int32_t s_take;                       // entries of the batch (fewer than 2^32 products between them)
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
        for (int32_t q = tid; q < words; q += THREADS) seen[q] = 0u;
        for (int32_t q = tid; q < ncols; q += THREADS) mn[q] = SO_NONE;
    } else {
        for (int32_t q = tid; q < nc; q += THREADS) {
            cols[q] = crow[q];
            mn[q] = SO_NONE;
        }
    }
#ifdef CSRK_SO_STAMPS
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
    if (tid == 0) atomicAdd(&g_so_stamps[7], 1ull);
#endif
    int64_t base = 0;                                // products before this batch
    int64_t e0 = a0;
    while (e0 < a1) {
        // the batch: up to THREADS entries of A, one per thread, and the running count of their products (a batch ends
        // before the entry that would take it to 2^32: indices inside a batch are 32-bit)
        if (tid == 0) s_take = THREADS;
        int64_t bs = 0, len = 0;
        if (e0 + tid < a1) {
            const int32_t j = a_ci[e0 + tid];
            bs = b_rp[j];
            len = (int64_t)b_rp[j + 1] - bs;
        }
        s_bs[tid] = bs;
        int64_t inc = len;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const int64_t o = __shfl_up(inc, off, WAVE);
            if (lane >= off) inc += o;
        }
        if (lane == WAVE - 1) s_wave64[wv] = inc;
        __syncthreads();                             // (also: the tables of the row are ready, the batch before is done with s_off)
#pragma unroll
        for (int v = 0; v < THREADS / WAVE; v++)
            if (v < wv) inc += s_wave64[v];
        s_off[tid + 1] = inc;
        if (tid == 0) s_off[0] = 0;
        if (!in_memory && inc >= 0xffffffffll) atomicMin(&s_take, tid);      // (in memory: 64-bit indices, no limit)
        __syncthreads();
        const int32_t take = (int32_t)(a1 - e0 < s_take ? a1 - e0 : s_take);
        if (take == 0) {                             // (cannot be: a row of B of 2^32 entries gives this row as many products)
            if (tid == 0) atomicMax(bad, 1u);
            return;
        }
        const int64_t total = s_off[take];
        SO_STAMP(1)
        // A wavefront takes SO_CHUNK = 64 * SO_UNROLL consecutive positions of the batch at a time, SO_UNROLL loads per lane
        // in flight (the walk is a chain of dependent loads -- column of B -> bitmap word -- and one at a time the kernel waits
        // out a memory latency per product).  The entry of the chunk's first position is looked for ONCE, by scalars; when
        // the whole chunk lies in that one row of B -- long rows: nearly always -- the lanes compute no address but base +
        // lane.  (Searching lane by lane, product by product, the walk of a ratings block was bound by the LDS round trips
        // of the search: 164 of a row's 270 k clocks.)
        auto seek = [&](int from, int64_t pidx) {    // the entry that holds pidx, from `from` on: gallop, then bisect
            if (s_off[from + 1] > pidx) return from;
            int lo = from + 1, step = 1;             // s_off[lo] <= pidx
            while (lo + step < take && s_off[lo + step] <= pidx) {
                lo += step;
                step <<= 1;
            }
            int hi = lo + step < take ? lo + step : take;      // s_off[hi] > pidx
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_off[mid] <= pidx) lo = mid;
                else hi = mid;
            }
            return lo;
        };
        int qw = 0;                                  // entry of the chunk's first position: only ever grows
        const int wave_u = __builtin_amdgcn_readfirstlane(wv);
        constexpr int64_t STRIDE = (int64_t)SO_CHUNK * (THREADS / WAVE);
        auto fetch = [&](int64_t pf, int32_t (&kk)[SO_UNROLL]) {
            qw = __builtin_amdgcn_readfirstlane(seek(qw, pf));
            const int64_t last = pf + SO_CHUNK - 1 < total ? pf + SO_CHUNK - 1 : total - 1;
            if (s_off[qw + 1] > last) {
                const int32_t *row = b_ci + (s_bs[qw] + (pf - s_off[qw])) + lane;
#pragma unroll
                for (int u = 0; u < SO_UNROLL; u++) kk[u] = pf + u * WAVE + lane <= last ? row[u * WAVE] : 0;
            } else {
                int q = qw;
#pragma unroll
                for (int u = 0; u < SO_UNROLL; u++) {
                    const int64_t pidx = pf + u * WAVE + lane;
                    kk[u] = 0;
                    if (pidx <= last) {
                        q = seek(q, pidx);
                        kk[u] = b_ci[s_bs[q] + (pidx - s_off[q])];
                    }
                }
            }
        };
        // MODE 0: by column (the LDS reads of a chunk's SO_UNROLL steps are issued together: step by step, each waited out its
        // own LDS round trip); 1: by entry, bisection in LDS; 2: bisection and atomic min in memory
        auto walk_chunks = [&](auto mode) {
            constexpr int MODE = decltype(mode)::value;
            for (int64_t pf = (int64_t)wave_u * SO_CHUNK; pf < total; pf += STRIDE) {
                int32_t kk[SO_UNROLL];
                fetch(pf, kk);
                const int64_t last = pf + SO_CHUNK - 1 < total ? pf + SO_CHUNK - 1 : total - 1;
                if constexpr (MODE == 0) {
                    const unsigned int p_first = (unsigned int)pf + lane, n_here = (unsigned int)(last - pf) + 1u;
                    unsigned int word[SO_UNROLL], least[SO_UNROLL];
                    bool open[SO_UNROLL];
#pragma unroll
                    for (int u = 0; u < SO_UNROLL; u++) word[u] = seen[kk[u] >> 5];
#pragma unroll
                    for (int u = 0; u < SO_UNROLL; u++)
                        open[u] = (unsigned int)(u * WAVE + lane) < n_here && !((word[u] >> (kk[u] & 31)) & 1u);
#pragma unroll
                    for (int u = 0; u < SO_UNROLL; u++) least[u] = open[u] ? mn[kk[u]] : 0u;
#pragma unroll
                    for (int u = 0; u < SO_UNROLL; u++)
                        if (open[u] && least[u] > p_first + u * WAVE) atomicMin(&mn[kk[u]], p_first + u * WAVE);
                } else if constexpr (MODE == 1) {
                    // the chunk's SO_UNROLL bisections step together: each step's LDS reads are in flight at once (one
                    // bisection after the other waited out fourteen round trips per product)
                    int32_t lo[SO_UNROLL], hi[SO_UNROLL];
#pragma unroll
                    for (int u = 0; u < SO_UNROLL; u++) {
                        lo[u] = 0;
                        hi[u] = pf + u * WAVE + lane <= last ? nc : 0;
                    }
                    for (int32_t span = nc; span > 0; span >>= 1) {      // (the same count of steps for every lane: ceil(log2(nc + 1)))
                        int32_t at[SO_UNROLL];
#pragma unroll
                        for (int u = 0; u < SO_UNROLL; u++) at[u] = lo[u] < hi[u] ? cols[lo[u] + ((hi[u] - lo[u]) >> 1)] : 0;
#pragma unroll
                        for (int u = 0; u < SO_UNROLL; u++)
                            if (lo[u] < hi[u]) {
                                const int32_t mid = lo[u] + ((hi[u] - lo[u]) >> 1);
                                if (at[u] < kk[u]) lo[u] = mid + 1;
                                else hi[u] = mid;
                            }
                    }
                    unsigned int least[SO_UNROLL];
#pragma unroll
                    for (int u = 0; u < SO_UNROLL; u++) least[u] = pf + u * WAVE + lane <= last ? mn[lo[u]] : 0u;
#pragma unroll
                    for (int u = 0; u < SO_UNROLL; u++) {
                        const unsigned int cand = (unsigned int)(base + pf + u * WAVE + lane);
                        if (pf + u * WAVE + lane <= last && least[u] > cand) atomicMin(&mn[lo[u]], cand);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < SO_UNROLL; u++) {
                        const int64_t pidx = pf + u * WAVE + lane;
                        if (pidx > last) break;
                        const int32_t lo = so_find(crow, nc, kk[u]);
                        const unsigned long long cand = (unsigned long long)(base + pidx);
                        if (krow[lo] > cand) atomicMin(&krow[lo], cand);      // (a read as a filter before the atomic)
                    }
                }
            }
        };
        if (in_memory) walk_chunks(std::integral_constant<int, 2>{});
        else walk_chunks(std::integral_constant<int, COLS ? 0 : 1>{});
        if (COLS && !in_memory) {
            __syncthreads();
            SO_STAMP(2)
            // the batch's discoveries are final: the columns with a minimum and no bit (a thread per column, a wavefront
            // 64 columns from a multiple of 64: their two words of `seen` are its own)
            int32_t found = 0;
            for (int32_t k0 = 0; k0 < ncols; k0 += THREADS) {
                const int32_t k = k0 + tid;
                unsigned int m = SO_NONE;
                if (k < ncols && !((seen[k >> 5] >> (k & 31)) & 1u)) m = mn[k];
                const bool fresh = m != SO_NONE;
                const unsigned long long f = __ballot(fresh);
                if (fresh) mn[k] = (unsigned int)(base + (int64_t)m);      // (its index in the whole walk)
                if (lane == 0 && (unsigned int)f) seen[k >> 5] |= (unsigned int)f;
                if (lane == 32 && (unsigned int)(f >> 32)) seen[k >> 5] |= (unsigned int)(f >> 32);
                if (lane == 0) found += __popcll(f);
            }
            if (found) atomicAdd(&s_found, found);
        }
        base += total;
        e0 += take;
        __syncthreads();
        SO_STAMP(3)
        if (COLS && !in_memory && s_found == nc) break;      // (every thread reads it after the barrier, and the next write is two barriers on)
    }
    if (in_memory) return;
    // ---- the places
    // population counts before each of n words: a thread's consecutive words, then the threads of a wavefront, then the
    // wavefronts; returns the count of them all (two barriers, the second after dst is written)
    auto counts_before = [&](const unsigned int *src, int32_t n, unsigned int *dst, int32_t start) {
        const int own = (n + THREADS - 1) / THREADS;
        const int32_t w_first = tid * own;
        int32_t mine = 0;
        for (int u = 0; u < own; u++)
            if (w_first + u < n) mine += __popc(src[w_first + u]);
        int32_t inc = mine;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const int32_t o = __shfl_up(inc, off, WAVE);
            if (lane >= off) inc += o;
        }
        if (lane == WAVE - 1) s_wave[wv] = inc;
        __syncthreads();
        int32_t run = start + inc - mine;
        int32_t all = 0;
#pragma unroll
        for (int v = 0; v < THREADS / WAVE; v++) {
            if (v < wv) run += s_wave[v];
            all += s_wave[v];
        }
        for (int u = 0; u < own; u++)
            if (w_first + u < n) {
                dst[w_first + u] = (unsigned int)run;
                run += __popc(src[w_first + u]);
            }
        __syncthreads();
        return all;
    };
    // COLS: the bits of `seen` are the row's columns, so an entry's position in the (ascending) row of C is the count of
    // bits before its column's -- the keys are walked by COLUMN, in LDS, and the row's columns are not read again
    if (COLS && counts_before(seen, words, spre, 0) != nc) {
        if (tid == 0) atomicMax(bad, 1u);            // (an entry no product lands on)
        return;
    }
    const int32_t span = COLS ? ncols : nc;          // keys: mn[column] or mn[entry]
    const int64_t win = (int64_t)win_words * 32;
    int32_t placed = 0;                              // keys below the window
    for (int64_t w0 = 0; w0 < t; w0 += win) {
        const int32_t used = (int32_t)(((t - w0 < win ? t - w0 : win) + 31) / 32);
        for (int32_t w = tid; w < used; w += THREADS) bits[w] = 0u;
        __syncthreads();
        // (here and below: the LDS reads and the loads of SO_PLACE_UNROLL keys are issued together, then used -- one at a
        // time each waited out its own round trip)
        for (int32_t x0 = tid; x0 < span; x0 += SO_PLACE_UNROLL * THREADS) {
            unsigned int mm[SO_PLACE_UNROLL];
#pragma unroll
            for (int u = 0; u < SO_PLACE_UNROLL; u++) mm[u] = x0 + u * THREADS < span ? mn[x0 + u * THREADS] : SO_NONE;
#pragma unroll
            for (int u = 0; u < SO_PLACE_UNROLL; u++) {
                const int64_t d = (int64_t)mm[u] - w0;      // (no key: 2^32 - 1, past every window of a row of fewer products)
                if (d >= 0 && d < win && d + w0 < t) atomicOr(&bits[d >> 5], 1u << (d & 31));
            }
        }
        __syncthreads();
        SO_STAMP(4)
        const int32_t all = counts_before(bits, used, before, placed);
        SO_STAMP(5)
        for (int32_t x0 = tid; x0 < span; x0 += SO_PLACE_UNROLL * THREADS) {
            unsigned int mm[SO_PLACE_UNROLL], sp[SO_PLACE_UNROLL], sw[SO_PLACE_UNROLL], bf[SO_PLACE_UNROLL], bw[SO_PLACE_UNROLL];
            int32_t cc[SO_PLACE_UNROLL];
            double vv[SO_PLACE_UNROLL];
            bool here[SO_PLACE_UNROLL];
#pragma unroll
            for (int u = 0; u < SO_PLACE_UNROLL; u++) {
                const int32_t x = x0 + u * THREADS < span ? x0 + u * THREADS : span - 1;      // (clamped: the reads stay unconditional)
                mm[u] = mn[x];
                if (COLS) {
                    sp[u] = spre[x >> 5];
                    sw[u] = seen[x >> 5];
                    cc[u] = x;
                } else {
                    cc[u] = cols[x];
                }
            }
#pragma unroll
            for (int u = 0; u < SO_PLACE_UNROLL; u++) {
                const int32_t x = x0 + u * THREADS;
                const int64_t d = (int64_t)mm[u] - w0;
                here[u] = x < span && d >= 0 && d < win && d + w0 < t;
                mm[u] = here[u] ? (unsigned int)d : 0u;
                const int32_t q = COLS ? (int32_t)sp[u] + __popc(sw[u] & ((1u << (x & 31)) - 1u)) : x;
                vv[u] = here[u] ? c_vs[c0 + q] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < SO_PLACE_UNROLL; u++) {
                bf[u] = before[mm[u] >> 5];
                bw[u] = bits[mm[u] >> 5];
            }
#pragma unroll
            for (int u = 0; u < SO_PLACE_UNROLL; u++) {
                if (!here[u]) continue;
                const int32_t rank = (int32_t)bf[u] + __popc(bw[u] & ((1u << (mm[u] & 31)) - 1u));
                const int64_t to = (int64_t)c0 + (nc - 1 - rank);
                oci[to] = cc[u];
                ovs[to] = vv[u];
            }
        }
        placed += all;
        SO_STAMP(6)
        if (placed >= nc) break;                     // (the same for every thread)
        __syncthreads();                             // (the bitmap is written again)
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
int spgemm_apply_reference_order(Matrix *a, Matrix *b, Matrix *c, const DevBuf *products)
{
    const int64_t n = c->nnz;
    if (n <= 1 || c->nrows == 0 || a->nnz == 0) return CSRK_OK;
    CSRK_REQUIRE(!c->ptr64 && c->val_type == CSRK_VAL_F64, "product has an unexpected layout");
    DevBuf key, tp_own, flag, oci, ovs, rows, cursor;
    // products of every row: the product kernels' own count when they made one (so_row_products_kernel: 33-60 us)
    const bool counted = products && products->p && products->bytes >= (size_t)(a->nrows + 1) * 8;
    if (!counted) CSRK_TRY(tp_own.alloc((size_t)(a->nrows + 1) * 8));
    int64_t *const tp_p = counted ? products->as<int64_t>() : tp_own.as<int64_t>();
    CSRK_TRY(flag.alloc(4));
    CSRK_TRY(rows.alloc((size_t)a->nrows * 4 + 4));
    CSRK_TRY(cursor.alloc((SO_OCTAVES + 1) * 4));
    CSRK_TRY(oci.alloc((size_t)n * 4));
    CSRK_TRY(ovs.alloc((size_t)n * 8));
    CSRK_HIP(hipMemsetAsync(cursor.p, 0, (SO_OCTAVES + 1) * 4, nullptr));
    CSRK_HIP(hipMemsetAsync(flag.p, 0, 4, nullptr));
    // LDS of the 1024-thread walk (the device's own limit decides; 18 KB of it are the kernel's batch tables).  By column --
    // 4 B per column of C and a bit -- when that leaves a window of 2^15 product indices at least and no row of B can hold
    // 2^32 entries (32-bit indices inside a batch); else by entry, 8 B each, with a window of 2^16.  The 256-thread walk:
    // rows of fewer than SO_MID products (and entries), by entry.
    int lds_max = 0, dev = 0;
    CSRK_HIP(hipGetDevice(&dev));
    CSRK_HIP(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    const int64_t budget = std::min<int64_t>(SO_LDS_BYTES, (int64_t)lds_max - 18 * 1024);
    const int64_t by_col = (((int64_t)c->ncols + 31) / 32 * 2 + c->ncols) * 4;
    int32_t win_cols = 0;
    for (int32_t w = 16384; w >= 1024; w >>= 1)
        if (by_col + (int64_t)w * 8 <= budget) {
            win_cols = w;
            break;
        }
    const bool cols_mode = win_cols > 0 && b->nnz < 0xffffffffll;
    const int32_t win_long = cols_mode ? win_cols : 2048;
    const int32_t cap_long = cols_mode ? INT32_MAX : (int32_t)std::max<int64_t>(0, (budget - (int64_t)win_long * 8) / 8);
    const size_t lds_long = cols_mode ? (size_t)(by_col + (int64_t)win_long * 8) : (size_t)cap_long * 8 + (size_t)win_long * 8;
    const int32_t win_mid = SO_MID / 32, cap_mid = SO_MID;
    const size_t lds_mid = (size_t)cap_mid * 8 + (size_t)win_mid * 8;
    const size_t lds_place = (size_t)SO_WIN_WORDS * 8;
    CSRK_REQUIRE((int64_t)lds_place + 2048 <= lds_max && cap_long >= SO_MID, "device has too little LDS for the ordering pass");
    const unsigned gr = (unsigned)ceil_div(a->nrows, 256);
    const unsigned gs = (unsigned)ceil_div((int64_t)a->nrows * SO_SUB, 256);
    int32_t n_long = 0, n_mid = 0, n_wave = 0, n_small = 0;      // listed rows of at least SO_MID products (first in the list), SO_WAVE, SO_TINY, SO_LEAST
    int32_t octaves[SO_OCTAVES + 1];                 // ([SO_OCTAVES]: some row needs its keys in memory)
    unsigned long long *key_p = nullptr;
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
            tp_p, c->ncols, CAP, WIN, LIST, key_p, oci_p, ovs_p, bad_p);                                   \
        CSRK_LAUNCH_CHECK();                                                                                           \
    } while (0)
#define ORDER(PA, PB)                                                                                                  \
    do {                                                                                                               \
        if (counted) {                                                                                                 \
        } else if (a->nnz / a->nrows >= 32)                                                                            \
            so_row_products_kernel<PA, PB, WAVE><<<(unsigned)ceil_div((int64_t)a->nrows * WAVE, 256), 256>>>(          \
                (const PA *)a->d_rowptrs, a->d_colinds, a->nrows, (const PB *)b->d_rowptrs, tp_p);         \
        else                                                                                                           \
            so_row_products_kernel<PA, PB, 16><<<(unsigned)ceil_div((int64_t)a->nrows * 16, 256), 256>>>(              \
                (const PA *)a->d_rowptrs, a->d_colinds, a->nrows, (const PB *)b->d_rowptrs, tp_p);         \
        CSRK_LAUNCH_CHECK();                                                                                           \
        so_list_rows_kernel<<<gr, 256>>>(tp_p, a->nrows, false, c_rp, cap_long, cursor.as<int32_t>(), nullptr); \
        CSRK_LAUNCH_CHECK();                                                                                           \
        so_tiny_kernel<PA, PB, SO_LEAST, SO_SUB><<<gs, 256>>>((const PA *)a->d_rowptrs, a->d_colinds, a->nrows,                \
                                                      (const PB *)b->d_rowptrs, b->d_colinds, c_rp, c->d_colinds, c_vs, \
                                                      tp_p, nullptr, 0, oci_p, ovs_p, bad_p);              \
        CSRK_LAUNCH_CHECK();                                                                                           \
        CSRK_HIP(hipMemcpy(octaves, cursor.p, sizeof octaves, hipMemcpyDeviceToHost));                                 \
        int32_t listed = 0;                                                                                            \
        for (int o = 0; o < SO_OCTAVES; o++) {       /* (63 - o = the octave: rows of [2^(63-o), 2^(64-o)) products) */ \
            const int32_t cnt = octaves[o];                                                                            \
            octaves[o] = listed;                                                                                       \
            listed += cnt;                                                                                             \
            if (63 - o >= 12) n_long = listed;       /* (SO_MID = 2^12) */                                             \
            if (63 - o >= 10) n_mid = listed - n_long; /* (SO_WAVE = 2^10) */                                          \
            if (63 - o >= 8) n_wave = listed - n_long - n_mid; /* (SO_TINY = 2^8) */                                   \
        }                                                                                                              \
        n_small = listed - n_long - n_mid - n_wave;                                                                    \
        if (octaves[SO_OCTAVES]) {                   /* (8 B per entry of C: only for the rows that do not fit LDS) */  \
            CSRK_TRY(key.alloc((size_t)n * 8));                                                                        \
            key_p = key.as<unsigned long long>();                                                                      \
        }                                                                                                              \
        if (listed > 0) {                                                                                              \
            CSRK_HIP(hipMemcpyAsync(cursor.p, octaves, sizeof octaves, hipMemcpyHostToDevice, nullptr));               \
            so_list_rows_kernel<<<gr, 256>>>(tp_p, a->nrows, true, c_rp, cap_long, cursor.as<int32_t>(),   \
                                             rows.as<int32_t>());                                                      \
            CSRK_LAUNCH_CHECK();                                                                                       \
        }                                                                                                              \
        if (n_long > 0) {                                                                                              \
            if (cols_mode) WALK_GO(PA, PB, 1024, true, n_long, lds_long, cap_long, win_long, rows.as<int32_t>());      \
            else WALK_GO(PA, PB, 1024, false, n_long, lds_long, cap_long, win_long, rows.as<int32_t>());               \
        }                                                                                                              \
        if (n_mid > 0) WALK_GO(PA, PB, 256, false, n_mid, lds_mid, cap_mid, win_mid, rows.as<int32_t>() + n_long);     \
        if (n_wave > 0) {                                                                                              \
            so_tiny_kernel<PA, PB, SO_WAVE, WAVE><<<(unsigned)ceil_div((int64_t)n_wave * WAVE, 256), 256>>>(           \
                (const PA *)a->d_rowptrs, a->d_colinds, a->nrows, (const PB *)b->d_rowptrs, b->d_colinds, c_rp,        \
                c->d_colinds, c_vs, tp_p, rows.as<int32_t>() + n_long + n_mid, n_wave, oci_p, ovs_p, bad_p);           \
            CSRK_LAUNCH_CHECK();                                                                                       \
        }                                                                                                              \
        if (n_small > 0) {                                                                                             \
            so_tiny_kernel<PA, PB, SO_TINY, SO_SUB><<<(unsigned)ceil_div((int64_t)n_small * SO_SUB, 256), 256>>>(      \
                (const PA *)a->d_rowptrs, a->d_colinds, a->nrows, (const PB *)b->d_rowptrs, b->d_colinds, c_rp,        \
                c->d_colinds, c_vs, tp_p, rows.as<int32_t>() + n_long + n_mid + n_wave, n_small, oci_p, ovs_p, bad_p); \
            CSRK_LAUNCH_CHECK();                                                                                       \
        }                                                                                                              \
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
    if (n_long > 0 && key_p) {                       // (the rows the walk left in key[]: the others leave at once)
        CSRK_HIP(hipFuncSetAttribute((const void *)so_place_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_place));
        so_place_kernel<<<(unsigned)n_long, SO_PLACE_THREADS, lds_place>>>(c_rp, c->d_colinds, c_vs, tp_p,
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

#ifdef CSRK_SO_STAMPS
extern "C" CSRK_API int csrk_debug_so_stamps(unsigned long long *out, int reset)
{
    CSRK_HIP(hipDeviceSynchronize());
    CSRK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_so_stamps), sizeof g_so_stamps));
    if (reset) {
        unsigned long long z[8] = {};
        CSRK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_so_stamps), z, sizeof z));
    }
    return CSRK_OK;
}
#endif

extern "C" int csrk_spgemm_set_order(int order)
{
    CSRK_REQUIRE(order >= -1 && order <= 1, "order must be -1 (environment), 0 (ascending) or 1 (reference)");
    g_spgemm_order.store(order);
    return CSRK_OK;
}

extern "C" int csrk_spgemm_get_order(int *order)
{
    CSRK_REQUIRE(order, "order is NULL");
    *order = spgemm_reference_order_wanted() ? 1 : 0;
    return CSRK_OK;
}
