// SpMV plans for libcsrk on gfx950: everything that is computed ONCE per handle (on its second product: see
// get_plan_locked in spmv.hip) -- the long-row split and its two tiers, the hot-column pack, the light stream and
// its cold staging lists.  The kernels of a product and their launches are in spmv.hip; DESIGN.md section 4 describes
// the forms.  Reference: csr/kernels/numba/__init__.py:55-67 has no plan (one sequential loop).
#include "spmv_plan.h"

#include <algorithm>
#include <cstring>
#include <ctime>

namespace csrk {
// tile_row[t] = number of row ends consumed before merge-path diagonal d = min(t*ITEMS, nrows+nnz).
// Row end r (= rp[r+1]) is consumed once all its nnz are: it lies before diagonal d iff
// rp[r+1] + r + 1 <= d.
// CSRK_PLAN_TRACE: laps of the plan builders on stderr (each lap waits for the device)
struct PlanTrace {
    bool on;
    double t0;
    static double now()
    {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    }
    PlanTrace() : on(getenv("CSRK_PLAN_TRACE") != nullptr), t0(0.0)
    {
        if (on) {
            (void)hipDeviceSynchronize();
            t0 = now();
        }
    }
    void lap(const char *what)
    {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const double t = now();
        fprintf(stderr, "[csrk plan] %-28s %8.3f ms\n", what, t - t0);
        t0 = t;
    }
};

template <class P>
__global__ void merge_plan_kernel(const P *__restrict__ rp, int32_t nrows, int64_t nnz, int items,
                                  int64_t n_tiles, int32_t *__restrict__ tile_row)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    int64_t total = (int64_t)nrows + nnz;
    int64_t d = t * items;
    if (d > total) d = total;
    int64_t lo = d - nnz > 0 ? d - nnz : 0;
    int64_t hi = d < nrows ? d : nrows;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)rp[mid + 1] <= d - mid - 1)
            lo = mid + 1;
        else
            hi = mid;
    }
    tile_row[t] = (int32_t)lo;
}

// 4-byte-aligned pair types: tile starts fall on arbitrary nnz indices, and gfx950 global loads
// only need dword alignment, so two consecutive colinds / values are fetched with one
// dwordx2 / dwordx4 load per lane (512 B / 1 KiB per wave-instruction).
// ---- hot-column cache: plan-time kernels ------------------------------------------------------------
// Column reference counts over the rows the tile kernel serves, from every `row_stride`-th GROUP of 64 consecutive rows
// and at most 128 entries of a row (a sample is enough to rank popularity); total[0] = entries counted.  One wavefront per
// group: the rows' kept entries are one run of colinds (but for the rows cut out to the tiers), walked 64 at a time by
// consecutive lanes -- an entry finds its row by bisection in the wavefront's table of row starts.  (A thread per row
// walking its entries one after the other waited out a memory round trip per entry with two workgroups on a CU: 2.8 ms of
// the headline matrix's plan, its largest item.)
// Popular columns collect millions of these increments: as global atomics they serialise (14 ms for the 4*10^7
// sampled entries of the headline matrix).  Each persistent workgroup therefore counts into an LDS hash table
// first (a column that finds a slot within HOT_PROBE probes stays there; the others go straight to memory) and
// flushes its <= HOT_TABLE distinct columns once at the end.
constexpr int HOT_TABLE = 8192, HOT_PROBE = 4;
constexpr int HOT_THREADS = 512;
template <class P>
__global__ __launch_bounds__(HOT_THREADS) void hot_count_kernel(const P *__restrict__ rp, const P *__restrict__ rp_light,
                                                               const int32_t *__restrict__ ci, int32_t nrows, int64_t row_stride,
                                                               int32_t *__restrict__ cnt, unsigned long long *__restrict__ total)
{
    __shared__ int32_t s_key[HOT_TABLE];
    __shared__ int32_t s_cnt[HOT_TABLE];
    __shared__ int32_t s_start[HOT_THREADS / WAVE][WAVE];
    for (int k = threadIdx.x; k < HOT_TABLE; k += HOT_THREADS) {
        s_key[k] = -1;
        s_cnt[k] = 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    int32_t *starts = s_start[wv];
    const int64_t n_groups = ((int64_t)nrows + WAVE * row_stride - 1) / (WAVE * row_stride);
    unsigned long long n = 0;
    for (int64_t q = (int64_t)blockIdx.x * (HOT_THREADS / WAVE) + wv; q < n_groups; q += (int64_t)gridDim.x * (HOT_THREADS / WAVE)) {
        const int64_t r = q * WAVE * row_stride + lane;
        int64_t s = 0;
        int32_t len = 0;
        if (r < nrows && !(rp_light && rp_light[r + 1] == rp_light[r])) {      // (not a row cut out to the tiers)
            s = rp[r];
            const int64_t l = (int64_t)rp[r + 1] - s;
            len = (int32_t)(l > 128 ? 128 : l);
        }
        const int32_t start = wave_exscan_i32(len, lane);
        const int32_t all = __builtin_amdgcn_readlane(start + len, WAVE - 1);
        __builtin_amdgcn_wave_barrier();             // (the group before has read its starts)
        starts[lane] = start;
        __builtin_amdgcn_wave_barrier();
        for (int32_t d0 = 0; d0 < all; d0 += WAVE) {
            const int32_t d = d0 + lane;
            const bool in = d < all;
            int p = 0;                               // the last row that starts at or before d (the one with entries there)
#pragma unroll
            for (int step = WAVE / 2; step; step >>= 1)
                if (in && starts[p + step] <= d) p += step;
            const int64_t k = __shfl(s, p, WAVE) + (d - __shfl(start, p, WAVE));
            if (!in) continue;
            const int32_t c = ci[k];
            uint32_t slot = ((uint32_t)c * 2654435761u) >> 19;      // 13 bits
            bool done = false;
#pragma unroll
            for (int pr = 0; pr < HOT_PROBE && !done; pr++) {
                const int32_t old = atomicCAS(&s_key[slot], -1, c);
                if (old == -1 || old == c) {
                    atomicAdd(&s_cnt[slot], 1);
                    done = true;
                }
                slot = (slot + 1) & (HOT_TABLE - 1);
            }
            if (!done) atomicAdd(&cnt[c], 1);
        }
        if (lane == 0) n += (unsigned long long)all;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < HOT_TABLE; k += HOT_THREADS)
        if (s_key[k] >= 0) atomicAdd(&cnt[s_key[k]], s_cnt[k]);
    for (int off = WAVE / 2; off; off >>= 1) n += __shfl_down(n, off, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0 && n) atomicAdd(total, n);
}

// out[0] = #columns with count >= thr, out[1] = sum of their counts
__global__ __launch_bounds__(256) void hot_census_kernel(const int32_t *__restrict__ cnt, int32_t ncols, int32_t thr,
                                                        unsigned long long *__restrict__ out)
{
    unsigned long long n = 0, sum = 0;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < ncols; c += (int64_t)gridDim.x * 256) {
        const int32_t v = cnt[c];
        if (v >= thr) {
            n++;
            sum += (unsigned long long)v;
        }
    }
    for (int off = WAVE / 2; off; off >>= 1) {
        n += __shfl_down(n, off, WAVE);
        sum += __shfl_down(sum, off, WAVE);
    }
    if ((threadIdx.x & (WAVE - 1)) == 0 && n) {
        atomicAdd(&out[0], n);
        atomicAdd(&out[1], sum);
    }
}

// The whole census in one pass: hist[v] = {#columns with count == v, sum of their counts} for v < HOT_HIST (counts at or
// above it share the last bin), so that the host finds the threshold for any slot budget from ONE copy instead of a
// binary search of ~27 launches and round trips (2.5 ms of the headline matrix's plan).  Small counts -- nearly all
// columns -- go through an LDS histogram.
constexpr int HOT_HIST = 65536, HOT_HIST_LDS = 2048;
__global__ __launch_bounds__(256) void hot_hist_kernel(const int32_t *__restrict__ cnt, int32_t ncols,
                                                      unsigned long long *__restrict__ hist_n,
                                                      unsigned long long *__restrict__ hist_sum)
{
    __shared__ unsigned int s_n[HOT_HIST_LDS];
    for (int k = threadIdx.x; k < HOT_HIST_LDS; k += 256) s_n[k] = 0u;
    __syncthreads();
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < ncols; c += (int64_t)gridDim.x * 256) {
        const int32_t v = cnt[c];
        if (v < HOT_HIST_LDS) {
            atomicAdd(&s_n[v], 1u);
        } else {
            const int b = v < HOT_HIST ? v : HOT_HIST;
            atomicAdd(&hist_n[b], 1ull);
            atomicAdd(&hist_sum[b], (unsigned long long)v);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < HOT_HIST_LDS; k += 256)
        if (s_n[k]) {
            atomicAdd(&hist_n[k], (unsigned long long)s_n[k]);
            atomicAdd(&hist_sum[k], (unsigned long long)s_n[k] * (unsigned long long)k);
        }
}

__global__ void hot_flag_kernel(const int32_t *__restrict__ cnt, int32_t ncols, int32_t thr, int32_t *__restrict__ flag)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c <= ncols) flag[c] = (c < ncols && cnt[c] >= thr) ? 1 : 0;
}

// pos[c] = exclusive scan of the flags: the packed columns in column order, with their counts
__global__ void hot_list_kernel(const int32_t *__restrict__ cnt, const int32_t *__restrict__ pos, int32_t ncols,
                                int32_t thr, int32_t *__restrict__ hot_cols, int32_t *__restrict__ hot_cnt)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < ncols && cnt[c] >= thr) {
        hot_cols[pos[c]] = (int32_t)c;
        hot_cnt[pos[c]] = cnt[c];
    }
}

// The packed columns in order of popularity: a stable LSD radix sort of (count, column) by count DESCENDING (digits of the
// complemented count), 8 bits a pass, one wavefront per chunk of 1024 -- digit counts per chunk, one scan over
// (digit, chunk), then the chunk places its elements in order (rank among the lanes of the same digit by ballots).  (On the
// host -- two 16-bit passes plus the copies both ways -- this was 1.5 ms of the headline matrix's plan.)
constexpr int HS_CHUNK = 1024;
__global__ __launch_bounds__(WAVE) void hot_sort_count_kernel(const uint32_t *__restrict__ key, int32_t n, int shift, int32_t nchunk,
                                                             int32_t *__restrict__ cnt)
{
    __shared__ int32_t h[256];
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += WAVE) h[d] = 0;
    __syncthreads();
    for (int r = 0; r < HS_CHUNK / WAVE; r++) {
        const int64_t i = (int64_t)blockIdx.x * HS_CHUNK + r * WAVE + lane;
        if (i < n) atomicAdd(&h[(~key[i] >> shift) & 255u], 1);
    }
    __syncthreads();
    for (int d = lane; d < 256; d += WAVE) cnt[(int64_t)d * nchunk + blockIdx.x] = h[d];
}
__global__ __launch_bounds__(WAVE) void hot_sort_place_kernel(const uint32_t *__restrict__ key, const int32_t *__restrict__ val, int32_t n,
                                                             int shift, int32_t nchunk, const int32_t *__restrict__ base,
                                                             uint32_t *__restrict__ key_out, int32_t *__restrict__ val_out)
{
    __shared__ int32_t b[256];
    const int lane = threadIdx.x;
    const unsigned long long below = lane ? (~0ull >> (WAVE - lane)) : 0ull;
    for (int d = lane; d < 256; d += WAVE) b[d] = base[(int64_t)d * nchunk + blockIdx.x];
    __syncthreads();
    for (int r = 0; r < HS_CHUNK / WAVE; r++) {
        const int64_t i = (int64_t)blockIdx.x * HS_CHUNK + r * WAVE + lane;
        const bool ok = i < n;
        const uint32_t k = ok ? key[i] : 0u;
        const uint32_t d = (~k >> shift) & 255u;
        unsigned long long same = __ballot(ok);      // the lanes that hold the same digit
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const unsigned long long bb = __ballot((d >> bit) & 1u);
            same &= ((d >> bit) & 1u) ? bb : ~bb;
        }
        const int rank = __popcll(same & below), group = __popcll(same);
        if (ok) {
            const int32_t pos = b[d] + rank;
            key_out[pos] = k;
            val_out[pos] = val[i];
        }
        __syncthreads();
        if (ok && rank == group - 1) b[d] += group;
        __syncthreads();
    }
}

// slot[hot_cols[k]] = k (hot_cols in its final, popularity order)
__global__ void hot_slot_kernel(const int32_t *__restrict__ hot_cols, int32_t n_hot, int32_t *__restrict__ slot)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_hot) slot[hot_cols[k]] = k;
}

static int HEAVY_MIN = 2048;      // tier 0 threshold (CSRK_HEAVY_MIN)
static int TIERB_MIN = 128;       // tier 1 threshold (CSRK_TIERB_MIN; 0 disables tier 1)

template <class P>
__global__ void heavy_flag_kernel(const P *__restrict__ rp, int32_t nrows, int32_t *__restrict__ flag,
                                  int64_t *__restrict__ hlen, int cut_min)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nrows) return;
    int64_t len = r < nrows ? (int64_t)rp[r + 1] - (int64_t)rp[r] : 0;
    bool h = len >= cut_min;
    flag[r] = h ? 1 : 0;
    hlen[r] = h ? len : 0;
}

template <class P>
__global__ void heavy_view_kernel(const P *__restrict__ rp, int32_t nrows, const int32_t *__restrict__ hidx,
                                  const int64_t *__restrict__ hbefore, P *__restrict__ rp_light,
                                  int32_t *__restrict__ heavy_row, int64_t *__restrict__ cut_pos,
                                  int64_t *__restrict__ cut_cum, int64_t *__restrict__ cut_len, int32_t n_heavy)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nrows) return;
    const int64_t light = (int64_t)rp[r] - hbefore[r];
    rp_light[r] = (P)light;
    if (r < nrows && hidx[r + 1] != hidx[r]) {     // row r is cut out
        const int32_t c = hidx[r];
        heavy_row[c] = (int32_t)r;
        cut_pos[c] = light;
        cut_cum[c] = hbefore[r];
        cut_len[c] = (int64_t)rp[r + 1] - (int64_t)rp[r];
    }
    if (r == nrows) cut_cum[n_heavy] = hbefore[nrows];
}

template <class P>
__global__ void heavy_sorted_kernel(const P *__restrict__ rp, const int32_t *__restrict__ ci,
                                    const int32_t *__restrict__ heavy_row, int32_t n_heavy, int32_t *__restrict__ bad)
{
    const int c = blockIdx.x;
    if (c >= n_heavy) return;
    const int32_t r = heavy_row[c];
    const int64_t s = rp[r], e = rp[r + 1];
    // (a 10^6-entry row is one workgroup's: 1024 threads with four comparisons in flight each; 256 threads one at a time
    // made this check 1.8 ms of the headline matrix's plan; the rows cut into pieces over a second grid dimension: 2.8 ms
    // -- a million workgroups that mostly have nothing to do)
    bool b = false;
    const int64_t step = (int64_t)blockDim.x * 4;
    for (int64_t k0 = s + threadIdx.x; k0 + 1 < e; k0 += step) {
        int32_t a[4], n[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t k = k0 + (int64_t)u * blockDim.x;
            const bool in = k + 1 < e;
            a[u] = in ? ci[k] : 0;
            n[u] = in ? ci[k + 1] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) b |= a[u] > n[u];
    }
    if (b) atomicOr(bad, 1);
}

__device__ __forceinline__ int64_t lower_bound_col(const int32_t *__restrict__ ci, int64_t lo, int64_t hi, int64_t col)
{
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)ci[mid] < col)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// cuts at or before each tile start: tile_cut[t] = #{c : cut_pos[c] <= j0(t)}
__global__ void heavy_tilecut_kernel(const int32_t *__restrict__ tile_row, int64_t n_tiles, int items, int64_t total,
                                     const int64_t *__restrict__ cut_pos, int32_t n_heavy, int32_t *__restrict__ tile_cut)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    int64_t d = t * items;
    if (d > total) d = total;
    const int64_t j0 = d - tile_row[t];
    int32_t lo = 0, hi = n_heavy;
    while (lo < hi) {
        int32_t mid = (lo + hi) >> 1;
        if (cut_pos[mid] <= j0)
            lo = mid + 1;
        else
            hi = mid;
    }
    tile_cut[t] = lo;
}

// XCD-aware walk over (column block, row chunk): workgroup g of a grid of xcd_block_chunk_grid() works on block b and
// chunk q, and the workgroups one XCD receives (g % 8) are ITS chunks (q % 8 == g % 8), block after block.  A row's entries
// in consecutive column blocks are neighbours in memory (a few entries each): walked this way the lines that hold them are
// fetched into one L2 once instead of once per block by whichever XCD the flat index fell on (tier 1's count + fill: 1.8 ->
// 1.1 ms of the headline matrix's plan; tier 0's acc_fill_kernel, a wavefront per 64 pairs, did not gain: 2.6 -> 3.0 ms).
__device__ __forceinline__ bool xcd_block_chunk(int64_t g, int32_t chunks, int32_t n_blocks, int32_t &b, int32_t &q)
{
    const int32_t per = (chunks + 7) / 8;
    const int64_t idx = g / 8;
    b = (int32_t)(idx / per);
    q = (int32_t)(g % 8) + 8 * (int32_t)(idx % per);
    return b < n_blocks && q < chunks;
}
static inline int64_t xcd_block_chunk_grid(int64_t chunks, int64_t n_blocks) { return 8 * ceil_div(chunks, 8) * n_blocks; }

template <class P>
__global__ void panel_count_kernel(const P *__restrict__ rp, const int32_t *__restrict__ ci,
                                   const int32_t *__restrict__ heavy_row, int32_t n_heavy, int32_t n_blocks,
                                   int32_t cb, int64_t *__restrict__ cnt)
{
    int32_t b, q;
    if (!xcd_block_chunk(blockIdx.x, (n_heavy + (int)blockDim.x - 1) / (int)blockDim.x, n_blocks, b, q)) return;
    const int32_t c = q * (int)blockDim.x + (int)threadIdx.x;
    if (c >= n_heavy) return;
    const int64_t i = (int64_t)b * n_heavy + c;      // index = b * H + c
    const int32_t r = heavy_row[c];
    const int64_t s = rp[r], e = rp[r + 1];
    const int64_t lo = lower_bound_col(ci, s, e, (int64_t)b * cb);
    const int64_t hi = lower_bound_col(ci, lo, e, (int64_t)(b + 1) * cb);
    cnt[i] = hi - lo;
}

template <class P, int VT, class PP>
__global__ void panel_fill_kernel(const P *__restrict__ rp, const int32_t *__restrict__ ci, const void *__restrict__ vs,
                                  const int32_t *__restrict__ heavy_row, int32_t n_heavy, int32_t n_blocks,
                                  int32_t cb, const int64_t *__restrict__ off, PP *__restrict__ prp,
                                  int32_t *__restrict__ pci, double *__restrict__ pvs)
{
    int32_t b, q;
    if (!xcd_block_chunk(blockIdx.x, (n_heavy + (int)blockDim.x - 1) / (int)blockDim.x, n_blocks, b, q)) return;
    const int32_t c = q * (int)blockDim.x + (int)threadIdx.x;
    if (c >= n_heavy) return;
    const int64_t i = (int64_t)b * n_heavy + c, pairs = (int64_t)n_heavy * n_blocks;
    prp[i] = (PP)off[i];
    if (i == pairs - 1) prp[pairs] = (PP)off[pairs];
    const int32_t r = heavy_row[c];
    const int64_t s = rp[r], e = rp[r + 1];
    const int64_t lo = lower_bound_col(ci, s, e, (int64_t)b * cb);
    const int64_t n = off[i + 1] - off[i];
    int64_t o = off[i];
    for (int64_t k = lo; k < lo + n; k++, o++) {
        pci[o] = ci[k];
        pvs[o] = ValLoad<VT>::at(vs, k);
    }
}

// one thread per tile: merge-path coordinates inside the tile's block
template <class PP>
__global__ void panel_plan_kernel(const PP *__restrict__ prp, int32_t n_heavy, int32_t n_blocks,
                                  const int64_t *__restrict__ blk_tile0, int64_t n_tiles, int items,
                                  PanelTile *__restrict__ tiles)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    int32_t lo = 0, hi = n_blocks;                 // block with blk_tile0[b] <= t < blk_tile0[b+1]
    while (hi - lo > 1) {
        int32_t mid = (lo + hi) >> 1;
        if (blk_tile0[mid] <= t)
            lo = mid;
        else
            hi = mid;
    }
    const int32_t b = lo;
    const int64_t r0 = (int64_t)b * n_heavy;                    // first row of the block in M'
    const int64_t e0 = (int64_t)prp[r0], e1 = (int64_t)prp[r0 + n_heavy];
    const int64_t total = (int64_t)n_heavy + (e1 - e0);
    const int64_t lt = t - blk_tile0[b];
    int64_t coord[2];
    for (int q = 0; q < 2; q++) {
        int64_t d = (lt + q) * items;
        if (d > total) d = total;
        int64_t a = d - (e1 - e0) > 0 ? d - (e1 - e0) : 0, z = d < n_heavy ? d : n_heavy;
        while (a < z) {
            int64_t mid = (a + z) >> 1;
            if ((int64_t)prp[r0 + mid + 1] - e0 <= d - mid - 1)
                a = mid + 1;
            else
                z = mid;
        }
        coord[q] = a;                                           // rows of the block consumed before d
    }
    int64_t d0 = lt * items, d1 = (lt + 1) * items;
    if (d0 > total) d0 = total;
    if (d1 > total) d1 = total;
    PanelTile pt;
    pt.i0 = (int32_t)(r0 + coord[0]);
    pt.i1 = (int32_t)(r0 + coord[1]);
    pt.j0 = e0 + (d0 - coord[0]);
    pt.nn = (int32_t)((d1 - coord[1]) - (d0 - coord[0]));
    pt.blk = b;
    pt.cslot = -1;
    tiles[t] = pt;
}

// Column order inside a tile (tier 1).  A tile of M' holds ~250 (block, row) pairs of ~7 entries each; in row-major order
// the repeats of a popular column are hundreds of gathers apart and L1 has dropped the line by then (0.83 line fills per
// entry).  The tile's entries are therefore STORED sorted by column: the index word becomes {column - block start : 21
// bits | position of the entry in the tile's row-major order : 11 bits}, the kernel gathers x[block start + column] with
// neighbouring lanes on neighbouring (often the same) columns and puts each product back at its row-major position in
// LDS, so the row sums see the very same addends in the very same order.  Inside a whole chunk of 128 ranks the even
// slots hold ranks 0..63 and the odd slots ranks 64..127: a lane loads the pair (2q, 2q + 1), so each of its two gather
// instructions then covers 64 CONSECUTIVE ranks.  One workgroup per tile: bitonic sort of the 2048 words in LDS.
__global__ __launch_bounds__(256) void panel_colsort_kernel(const PanelTile *__restrict__ tiles, int32_t cb,
                                                           int32_t *__restrict__ pci, double *__restrict__ pvs)
{
    __shared__ uint32_t s_key[MERGE_ITEMS];
    __shared__ double s_val[MERGE_ITEMS];
    const PanelTile pt = tiles[blockIdx.x];
    const int nn = pt.nn;
    const int64_t base = (int64_t)pt.blk * cb;
    for (int k = threadIdx.x; k < MERGE_ITEMS; k += 256) {
        uint32_t w = 0xffffffffu;
        if (k < nn) {
            w = ((uint32_t)((int64_t)pci[pt.j0 + k] - base) << PANEL_POS_BITS) | (uint32_t)k;
            s_val[k] = pvs[pt.j0 + k];
        }
        s_key[k] = w;
    }
    __syncthreads();
    for (int size = 2; size <= MERGE_ITEMS; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < MERGE_ITEMS / 2; i += 256) {
                const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint32_t a = s_key[lo], b = s_key[hi];
                if ((a > b) == up) {
                    s_key[lo] = b;
                    s_key[hi] = a;
                }
            }
            __syncthreads();
        }
    for (int k = threadIdx.x; k < nn; k += 256) {
        const uint32_t w = s_key[k];
        pci[pt.j0 + panel_slot_of_rank(k, nn)] = (int32_t)w;
        pvs[pt.j0 + panel_slot_of_rank(k, nn)] = s_val[w & ((1u << PANEL_POS_BITS) - 1)];
    }
}

__global__ void panel_blockends_kernel(const int64_t *__restrict__ off, int32_t n_heavy, int32_t n_blocks,
                                       int64_t *__restrict__ out)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b <= n_blocks) out[b] = off[(int64_t)b * n_heavy];
}

// Where each column block starts inside each heavy row, from ONE pass over the rows' (ascending) columns:
// pstart[b * H + c] = entries of heavy row c in blocks < b, for b = 0 .. n_blocks (the last = the row's length).
// One wavefront per 4096-entry piece of a row; an entry whose block differs from its predecessor's opens that
// block and every empty block skipped in between.  (A binary search per (block, row) pair -- 3.7 * 10^7 pairs on
// the headline matrix, twice -- was 14 ms of the plan.)
constexpr int ACC_PIECE = 4096;
template <class P>
__global__ __launch_bounds__(256) void acc_pairstart_kernel(const P *__restrict__ rp, const int32_t *__restrict__ ci,
                                                           const int32_t *__restrict__ heavy_row, int32_t n_heavy,
                                                           int32_t n_blocks, int32_t cb, const int32_t *__restrict__ task_row,
                                                           const int32_t *__restrict__ task_piece, int64_t n_tasks,
                                                           int32_t *__restrict__ pstart)
{
    // (an XCD, blockIdx % 8, takes a contiguous range of the tasks: neighbouring rows share the lines of pstart they write)
    const int64_t wpg = blockDim.x / WAVE, per = ((int64_t)gridDim.x + 7) / 8;
    const int64_t q = (((int64_t)blockIdx.x % 8) * per + (int64_t)blockIdx.x / 8) * wpg + threadIdx.x / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (q >= n_tasks) return;
    const int32_t c = task_row[q];
    const int64_t s = rp[heavy_row[c]], e = rp[heavy_row[c] + 1];
    const int64_t k0 = s + (int64_t)task_piece[q] * ACC_PIECE;
    const int64_t k1 = k0 + ACC_PIECE < e ? k0 + ACC_PIECE : e;
    for (int64_t k = k0 + lane; k < k1; k += WAVE) {
        const int32_t b = ci[k] / cb;
        const int32_t bp = k > s ? ci[k - 1] / cb : -1;
        for (int32_t bb = bp + 1; bb <= b; bb++) pstart[(int64_t)bb * n_heavy + c] = (int32_t)(k - s);
        if (k == e - 1)
            for (int32_t bb = b + 1; bb <= n_blocks; bb++) pstart[(int64_t)bb * n_heavy + c] = (int32_t)(e - s);
    }
}

__global__ void acc_paircount_kernel(const int32_t *__restrict__ pstart, int64_t pairs, int32_t n_heavy,
                                     int64_t *__restrict__ cnt)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < pairs) cnt[i] = (int64_t)pstart[i + n_heavy] - (int64_t)pstart[i];
}

// One wavefront per column block: gap[b * H + c] = distance from heavy row c to the previous heavy row with entries in
// block b (0 for the block's first one and for absent pairs), and the padding entries a gap over ACC_MAXSTEP needs
// are added to the pair's count.
__global__ __launch_bounds__(256) void acc_gap_kernel(int64_t *__restrict__ cnt, int32_t n_heavy, int32_t n_blocks,
                                                     int32_t *__restrict__ gap)
{
    const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (b >= n_blocks) return;
    int32_t last = -1;
    for (int32_t c0 = 0; c0 < n_heavy; c0 += WAVE) {
        const int32_t c = c0 + lane;
        const bool present = c < n_heavy && cnt[b * n_heavy + c] > 0;
        const unsigned long long mask = __ballot(present);
        const unsigned long long below = lane ? (mask & (~0ull >> (WAVE - lane))) : 0ull;
        const int32_t prev = below ? c0 + 63 - __clzll((long long)below) : last;
        if (c < n_heavy) {
            const int32_t g = present && prev >= 0 ? c - prev : 0;
            gap[b * n_heavy + c] = g;
            if (g > ACC_MAXSTEP) cnt[b * n_heavy + c] += (g - 1) / ACC_MAXSTEP;
        }
        if (mask) last = c0 + 63 - __clzll((long long)mask);
    }
}

// Where logical tile t is stored.  Workgroup w owns the logical tiles [wg_t0[w], wg_t0[w + 1]); its k-th tile is stored
// at k * n_wg + w, i.e. the workgroups' streams are interleaved tile by tile: the persistent workgroups advance at
// about the same pace, so at any moment they read one contiguous ~1 MB window of the array, spread over all HBM
// channels.  (Contiguous per-workgroup ranges put 256 concurrent streams at a fixed stride: when that stride
// resonates with the channel interleave the kernel loses up to 38 % -- 0.129 -> 0.177 ms measured on a 2-way partition's
// shard at 580 tiles per workgroup, 0.218 -> 0.224 on the headline matrix; any other workgroup count restored the rate.)
__device__ __forceinline__ int64_t acc_phys_tile(int64_t t, const int64_t *__restrict__ wg_t0, int32_t n_wg)
{
    // the workgroup with wg_t0[w] <= t < wg_t0[w + 1], wg_t0[w] = floor(n_tiles w / n_wg) (build_acc_panel): w is
    // floor(t n_wg / n_tiles) or a neighbour -- an estimate and two looks at the table instead of a bisection per entry
    // (eight dependent loads each, for every entry of tier 0)
    const int64_t n_tiles = wg_t0[n_wg];
    int64_t w = (int64_t)((double)t * (double)n_wg / (double)(n_tiles > 0 ? n_tiles : 1));      // (an estimate: the table decides)
    w = w > n_wg - 1 ? n_wg - 1 : (w < 0 ? 0 : w);
    while (w > 0 && wg_t0[w] > t) w--;
    while (w + 1 < n_wg && wg_t0[w + 1] <= t) w++;
    return (t - wg_t0[w]) * n_wg + w;
}

// One wavefront per 64 consecutive (block, heavy row) pairs of ONE block: their slots are one contiguous range of the
// block's logical stream, walked 64 slots at a time -- a slot finds its pair by bisection in the wavefront's own table of
// pair starts, a padding slot (the first ones of a pair whose row lies more than ACC_MAXSTEP rows after the previous one)
// writes (0.0, zero slot, step 7), the others copy the pair's entries -- so the stream is written by consecutive lanes and
// every lane works.  (A thread per pair walking its 3.6 entries one after the other wrote and read 64 scattered lines
// per instruction: 6.9 ms of the headline matrix's plan, its largest item; this form 2.7.)
template <class P, int VT, class SV>
__global__ __launch_bounds__(256) void acc_fill_kernel(const P *__restrict__ rp, const int32_t *__restrict__ ci,
                                                      const void *__restrict__ vs, const int32_t *__restrict__ heavy_row,
                                                      int32_t n_heavy, int32_t n_blocks, int32_t cb,
                                                      const int64_t *__restrict__ off, const int64_t *__restrict__ blk_tile0,
                                                      const int32_t *__restrict__ pstart, const int32_t *__restrict__ gap,
                                                      SV *__restrict__ pvals, uint16_t *__restrict__ pidx,
                                                      int32_t *__restrict__ tile_row0, const int64_t *__restrict__ wg_t0,
                                                      int32_t n_wg, int32_t waves_per_block)
{
    __shared__ int32_t s_start[256 / WAVE][WAVE];
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    const int32_t b = (int32_t)(w / waves_per_block);
    if (b >= n_blocks) return;
    const int32_t c0 = (int32_t)(w % waves_per_block) * WAVE, c = c0 + lane;
    const bool valid = c < n_heavy;
    const int64_t i = (int64_t)b * n_heavy + (valid ? c : c0);
    const int64_t off_i = off[i];
    int32_t n_all = 0, g = 0;
    int64_t lo = 0;
    if (valid) {
        n_all = (int32_t)(off[i + 1] - off_i);
        g = gap[i];
        lo = (int64_t)rp[heavy_row[c]] + pstart[i];
    }
    const int32_t npad = g > ACC_MAXSTEP ? (g - 1) / ACC_MAXSTEP : 0;
    const int32_t start = wave_exscan_i32(n_all, lane);
    const int32_t total = __builtin_amdgcn_readlane(start + n_all, WAVE - 1);
    if (total == 0) return;
    // logical position of the wavefront's first slot (lane 0's pair: c0 < n_heavy)
    const int64_t L0 = blk_tile0[b] * ACC_TILE + (__shfl(off_i, 0, WAVE) - off[(int64_t)b * n_heavy]);
    int32_t *starts = s_start[wv];
    starts[lane] = start;
    __builtin_amdgcn_wave_barrier();
    for (int32_t d0 = 0; d0 < total; d0 += WAVE) {
        const int32_t d = d0 + lane;
        const bool in = d < total;
        int p = 0;                                   // the last pair that starts at or before slot d (an empty pair shares its
#pragma unroll                                       // start with the next one: the last of them is the one with entries)
        for (int step = WAVE / 2; step; step >>= 1)
            if (in && starts[p + step] <= d) p += step;
        const int32_t at = d - __shfl(start, p, WAVE);      // slot inside the pair
        const int32_t npad_p = __shfl(npad, p, WAVE), g_p = __shfl(g, p, WAVE);
        const int64_t lo_p = __shfl(lo, p, WAVE);
        if (!in) continue;
        const int64_t L = L0 + d;
        const int64_t t = L / ACC_TILE;
        const int el = (int)(L % ACC_TILE);
        const int64_t pt = acc_phys_tile(t, wg_t0, n_wg);
        if (at < npad_p) {                           // padding entry k = at + 1 stands on heavy row c - g + 7k
            pvals[pt * ACC_TILE + acc_slot_of<SV>(el)] = (SV)0.0;
            pidx[pt * ACC_TILE + el] = (uint16_t)((uint32_t)cb | ((el ? (uint32_t)ACC_MAXSTEP : 0u) << ACC_ROW_SHIFT));
            if (el == 0) tile_row0[t] = c0 + p - g_p + ACC_MAXSTEP * (at + 1);
        } else {
            const int64_t k = lo_p + (at - npad_p);
            // the pair's first entry: the step from the previous row (or the last padding entry) to this row
            const uint32_t step_in = at == npad_p ? (uint32_t)(g_p - ACC_MAXSTEP * npad_p) : 0u;
            pvals[pt * ACC_TILE + acc_slot_of<SV>(el)] = (SV)ValLoad<VT>::at(vs, k);
            pidx[pt * ACC_TILE + el] = (uint16_t)((uint32_t)(ci[k] - b * cb) | ((el ? step_in : 0u) << ACC_ROW_SHIFT));
            if (el == 0) tile_row0[t] = c0 + p;
        }
    }
}

// one workgroup per block: pads the block's last tile with (0.0, column slot ACC_CB (a zero in LDS), step 0 = the
// block's last heavy row) -- a padding entry adds 0.0 * 0.0 to an accumulator
template <class SV>
__global__ __launch_bounds__(256) void acc_pad_kernel(const int64_t *__restrict__ off, int32_t n_heavy, int32_t n_blocks,
                                                     const int64_t *__restrict__ blk_tile0, SV *__restrict__ pvals,
                                                     uint16_t *__restrict__ pidx, const int64_t *__restrict__ wg_t0, int32_t n_wg)
{
    const int32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const int64_t cnt = off[(int64_t)(b + 1) * n_heavy] - off[(int64_t)b * n_heavy];
    const int64_t L0 = blk_tile0[b] * ACC_TILE + cnt, L1 = blk_tile0[b + 1] * ACC_TILE;
    for (int64_t L = L0 + threadIdx.x; L < L1; L += blockDim.x) {      // (the tail of the block's last tile: one tile)
        const int64_t pt = acc_phys_tile(L / ACC_TILE, wg_t0, n_wg);
        const int el = (int)(L % ACC_TILE);
        pvals[pt * ACC_TILE + acc_slot_of<SV>(el)] = (SV)0.0;
        pidx[pt * ACC_TILE + el] = (uint16_t)ACC_CB;
    }
}

// (wave_exscan_i32 / wave_segscan: wave.h)

// smallest r in [0, nrows) with rpv[r + 1] > L (the row holding view entry L); L < rpv[nrows]
template <class P>
__device__ __forceinline__ int32_t ls_row_of(const P *__restrict__ rpv, int32_t nrows, int64_t L)
{
    int32_t lo = 0, hi = nrows - 1;
    while (lo < hi) {
        const int32_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)rpv[mid + 1] > L)
            hi = mid;
        else
            lo = mid + 1;
    }
    return lo;
}

// tile_row[t] = the row that holds a tile's first entry, tile_row[n_tiles] = the last row: the fill kernel then looks for an
// entry's row between its tile's bounds (9 steps over neighbouring pointers instead of 23 over the whole array: 1.07 ms of
// the headline matrix's plan)
template <class P>
__global__ void ls_tilerow_kernel(const P *__restrict__ rpv, int32_t nrows, int64_t n_ent, int64_t n_tiles, int32_t *__restrict__ tile_row)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    const int64_t e0 = t * ACC_TILE;
    tile_row[t] = t < n_tiles && e0 < n_ent ? ls_row_of(rpv, nrows, e0) : nrows - 1;
}

template <class P>
__global__ void ls_rowflag_kernel(const P *__restrict__ rpv, int32_t nrows, int32_t *__restrict__ flag)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= nrows) flag[r] = (r < nrows && rpv[r + 1] > rpv[r]) ? 1 : 0;
}

// dense rows: rpd[r] = rpv[r] + (empty rows before r) -- every empty row of the view gets one slot; nz = exclusive scan of
// the non-empty flags (so r - nz[r] = empty rows before r)
template <class P>
__global__ void ls_dense_ptr_kernel(const P *__restrict__ rpv, const int32_t *__restrict__ nz, int32_t nrows, P *__restrict__ rpd)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= nrows) rpd[r] = (P)((int64_t)rpv[r] + (r - (int64_t)nz[r]));
}

template <class P>
__global__ void ls_rowids_kernel(const P *__restrict__ rpv, int32_t nrows, const int32_t *__restrict__ ridx,
                                 int32_t *__restrict__ rowids)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nrows && rpv[r + 1] > rpv[r]) rowids[ridx[r]] = (int32_t)r;
}

// One thread per slot of the stream (n_ent entries, then padding to a whole tile): view entry L of view row r is the
// source entry src[r] + (L - rpv[r]) (a view row is a whole row of the source or empty).
template <class P, int VT, class SV>
__global__ __launch_bounds__(256) void ls_fill_kernel(const P *__restrict__ src, const P *__restrict__ rpv, int32_t nrows,
                                                     const int32_t *__restrict__ ci, const void *__restrict__ vs,
                                                     int64_t n_ent, int64_t n_slots, const int32_t *__restrict__ slot_map, SV *__restrict__ svals,
                                                     uint32_t *__restrict__ sidx, const P *__restrict__ rp_len,
                                                     const int32_t *__restrict__ tile_row)
{
    // rp_len (dense rows): the view's own row pointers; rpv then gives every row at least one slot, and a row that is
    // empty in rp_len becomes one padding entry that opens (and is) its run
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots) return;
    const int64_t t = slot / ACC_TILE;
    const int el = (int)(slot % ACC_TILE);
    const int64_t L = slot;
    double v = 0.0;
    uint32_t ix = LS_PAD;
    if (L < n_ent) {
        int32_t lo = tile_row[t], hi = tile_row[t + 1];      // ls_row_of between the tile's bounds
        while (lo < hi) {
            const int32_t mid = lo + ((hi - lo) >> 1);
            if ((int64_t)rpv[mid + 1] > L)
                hi = mid;
            else
                lo = mid + 1;
        }
        const int32_t r = lo;
        const int64_t first = (int64_t)rpv[r];
        if (rp_len && rp_len[r + 1] == rp_len[r]) {
            ix = LS_PAD | LS_START_BIT;
        } else {
            const int64_t a = (int64_t)src[r] + (L - first);
            v = ValLoad<VT>::at(vs, a);
            const int32_t c = ci[a];
            const int32_t sl = slot_map ? slot_map[c] : -1;         // slot of a packed column
            ix = sl >= 0 ? (LS_HOT_BIT | (uint32_t)sl) : (uint32_t)c;
            if (L == first) ix |= LS_START_BIT;
        }
    }
    svals[t * ACC_TILE + acc_slot_of<SV>(el)] = (SV)v;
    sidx[t * ACC_TILE + acc_idx_slot(el)] = ix;
}

// per tile: run numbering base
template <class P>
__global__ void ls_tilebase_kernel(const P *__restrict__ rpv, int32_t nrows, const int32_t *__restrict__ ridx, int64_t n_tiles,
                                   const int32_t *__restrict__ tile_row, int32_t *__restrict__ tile_base)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const int64_t e0 = t * ACC_TILE;      // (< the entry count: the stream has no empty tile)
    const int32_t r = tile_row[t];
    tile_base[t] = ridx[r] + ((int64_t)rpv[r] == e0 ? 0 : 1);
}

template <class P>
__global__ void vec_count_kernel(const P *__restrict__ rp, int32_t nrows, int64_t *__restrict__ cnt)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    int64_t len = (int64_t)rp[r + 1] - (int64_t)rp[r];
    cnt[r] = len <= VEC_SEG ? 1 : (len + VEC_SEG - 1) / VEC_SEG;
}

__global__ void vec_fill_kernel(const int64_t *__restrict__ seg_off, int32_t nrows, int32_t *__restrict__ seg_row)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    for (int64_t q = seg_off[r]; q < seg_off[r + 1]; q++) seg_row[q] = (int32_t)r;
}

// ---- host side ----------------------------------------------------------------------------------
constexpr int HEAVY_STREAMS = 8;   // XCDs: blockIdx % 8 labels the XCD group (speed assumption only)

// Build one panel tier: M' (column-block-major copy of the listed rows, float64 values), its tiles and
// the workgroup list.  `xcd_streams`: order the groups so that column block b is served by workgroups
// with blockIdx % 8 == b % 8.
template <class P, int VT>
static int build_panel(Matrix *m, Panel *pn, const std::vector<int32_t> &rows, int64_t nnz_rows, int32_t cb,
                       int tpw, bool xcd_streams, hipStream_t s)
{
    PlanTrace tr;
    const P *rp = (const P *)m->d_rowptrs;
    const int32_t n = (int32_t)rows.size();
    const int32_t nb = (int32_t)ceil_div(m->ncols > 0 ? m->ncols : 1, cb);
    const int64_t pairs = (int64_t)n * nb;
    CSRK_TRY(pn->row_list.alloc((size_t)n * 4));
    CSRK_TRY(stage_h2d(pn->row_list.p, rows.data(), (size_t)n * 4, s));
    DevBuf off, bends;
    CSRK_TRY(off.alloc((size_t)(pairs + 1) * 8));
    const unsigned g = (unsigned)xcd_block_chunk_grid(ceil_div(n, 256), nb);
    panel_count_kernel<P><<<g, 256, 0, s>>>(rp, m->d_colinds, pn->row_list.as<int32_t>(), n, nb, cb, off.as<int64_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(exclusive_scan_i64(off.as<int64_t>(), off.as<int64_t>(), pairs, s));
    pn->p64 = nnz_rows > INT32_MAX;
    CSRK_TRY(pn->rp.alloc((size_t)(pairs + 1) * (pn->p64 ? 8 : 4)));
    CSRK_TRY(pn->ci.alloc((size_t)nnz_rows * 4));
    CSRK_TRY(pn->vs.alloc((size_t)nnz_rows * 8));
    if (pn->p64)
        panel_fill_kernel<P, VT, int64_t><<<g, 256, 0, s>>>(rp, m->d_colinds, m->d_values, pn->row_list.as<int32_t>(), n, nb,
                                                          cb, off.as<int64_t>(), pn->rp.as<int64_t>(),
                                                          pn->ci.as<int32_t>(), pn->vs.as<double>());
    else
        panel_fill_kernel<P, VT, int32_t><<<g, 256, 0, s>>>(rp, m->d_colinds, m->d_values, pn->row_list.as<int32_t>(), n, nb,
                                                          cb, off.as<int64_t>(), pn->rp.as<int32_t>(),
                                                          pn->ci.as<int32_t>(), pn->vs.as<double>());
    CSRK_LAUNCH_CHECK();
    // tiles per block (host: nb is at most a few thousand)
    CSRK_TRY(bends.alloc((size_t)(nb + 1) * 8));
    panel_blockends_kernel<<<(unsigned)ceil_div(nb + 1, 256), 256, 0, s>>>(off.as<int64_t>(), n, nb, bends.as<int64_t>());
    CSRK_LAUNCH_CHECK();
    std::vector<int64_t> be((size_t)nb + 1), t0((size_t)nb + 1);
    CSRK_TRY(stage_d2h(be.data(), bends.p, (size_t)(nb + 1) * 8, s));
    CSRK_HIP(hipStreamSynchronize(s));
    t0[0] = 0;
    for (int32_t b = 0; b < nb; b++) t0[b + 1] = t0[b] + ceil_div((int64_t)n + be[b + 1] - be[b], MERGE_ITEMS);
    const int64_t n_tiles = t0[nb];
    CSRK_TRY(stage_h2d(bends.p, t0.data(), (size_t)(nb + 1) * 8, s));
    tr.lap("  tier 1: count, scan, fill");
    CSRK_TRY(pn->tile.alloc((size_t)n_tiles * sizeof(PanelTile)));
    if (pn->p64)
        panel_plan_kernel<int64_t><<<(unsigned)ceil_div(n_tiles, 256), 256, 0, s>>>(
            pn->rp.as<int64_t>(), n, nb, bends.as<int64_t>(), n_tiles, MERGE_ITEMS, pn->tile.as<PanelTile>());
    else
        panel_plan_kernel<int32_t><<<(unsigned)ceil_div(n_tiles, 256), 256, 0, s>>>(
            pn->rp.as<int32_t>(), n, nb, bends.as<int64_t>(), n_tiles, MERGE_ITEMS, pn->tile.as<PanelTile>());
    CSRK_LAUNCH_CHECK();
    CSRK_REQUIRE(((int64_t)cb << PANEL_POS_BITS) <= (1ll << 32), "column block too wide for the tile's index word");
    if (n_tiles > 0) {
        panel_colsort_kernel<<<(unsigned)n_tiles, 256, 0, s>>>(pn->tile.as<PanelTile>(), cb, pn->ci.as<int32_t>(), pn->vs.as<double>());
        CSRK_LAUNCH_CHECK();
    }
    tr.lap("  tier 1: tiles, column order");

    // workgroup list: `tpw` consecutive tiles of one block per workgroup
    std::vector<PanelGroup> groups;
    auto block_groups = [&](int32_t b, std::vector<PanelGroup> &out) {
        for (int64_t t = t0[b]; t < t0[b + 1]; t += tpw) {
            PanelGroup gq;
            gq.t0 = t;
            gq.nt = (int32_t)(t0[b + 1] - t < tpw ? t0[b + 1] - t : tpw);
            gq.blk = b;
            out.push_back(gq);
        }
    };
    if (!xcd_streams || nb < 2 * HEAVY_STREAMS) {      // too few blocks to keep all 8 XCDs busy per stream
        for (int32_t b = 0; b < nb; b++) block_groups(b, groups);
    } else {
        std::vector<PanelGroup> st[HEAVY_STREAMS];
        size_t longest = 0;
        for (int32_t b = 0; b < nb; b++) block_groups(b, st[b % HEAVY_STREAMS]);
        for (int q = 0; q < HEAVY_STREAMS; q++) longest = st[q].size() > longest ? st[q].size() : longest;
        PanelGroup pad;
        pad.t0 = 0;
        pad.nt = 0;
        pad.blk = 0;
        groups.reserve(longest * HEAVY_STREAMS);
        for (size_t i = 0; i < longest; i++)
            for (int q = 0; q < HEAVY_STREAMS; q++) groups.push_back(i < st[q].size() ? st[q][i] : pad);
    }
    // the tiles' carries per long row (a tile's carry belongs to the pair holding its last, unfinished row end: static): the
    // first `ncs` of a row go to carry rows behind the block partials (Panel::ncs), the others are listed
    int32_t ncs = 0;
    {
        std::vector<PanelTile> ht((size_t)n_tiles);
        CSRK_TRY(stage_d2h(ht.data(), pn->tile.p, (size_t)n_tiles * sizeof(PanelTile), nullptr));
        std::vector<int32_t> cnt((size_t)n, 0);
        for (int64_t t = 0; t < n_tiles; t++)
            if ((int64_t)ht[(size_t)t].i1 < pairs) {
                const int32_t c = ++cnt[(size_t)(ht[(size_t)t].i1 % n)];
                ncs = c > ncs ? c : ncs;
            }
        ncs = ncs < PANEL_CARRY_ROWS ? ncs : PANEL_CARRY_ROWS;
        std::vector<int32_t> crp((size_t)n + 1, 0), cidx;
        bool listed = false;
        for (int32_t h = 0; h < n; h++) {
            crp[(size_t)h + 1] = crp[(size_t)h] + (cnt[(size_t)h] > ncs ? cnt[(size_t)h] - ncs : 0);
            listed = listed || cnt[(size_t)h] > ncs;
        }
        cidx.resize((size_t)crp[(size_t)n] + 1);
        std::vector<int32_t> seen((size_t)n, 0);
        for (int64_t t = 0; t < n_tiles; t++) {      // ascending tiles: each row's carries come out in tile order
            PanelTile &T = ht[(size_t)t];
            T.cslot = -1;
            if ((int64_t)T.i1 >= pairs) continue;
            const int32_t h = (int32_t)(T.i1 % n), j = seen[(size_t)h]++;
            if (j < ncs) T.cslot = pairs + (int64_t)j * n + h;
            else cidx[(size_t)(crp[(size_t)h] + j - ncs)] = (int32_t)t;
        }
        CSRK_TRY(stage_h2d(pn->tile.p, ht.data(), (size_t)n_tiles * sizeof(PanelTile), nullptr));
        if (listed) {
            CSRK_TRY(pn->crp.alloc(crp.size() * 4));
            CSRK_TRY(pn->cidx.alloc(cidx.size() * 4));
            CSRK_TRY(stage_h2d(pn->crp.p, crp.data(), crp.size() * 4, nullptr));
            CSRK_TRY(stage_h2d(pn->cidx.p, cidx.data(), cidx.size() * 4, nullptr));
        }
    }
    pn->groups = (int64_t)groups.size();
    CSRK_TRY(pn->group.alloc(groups.size() * sizeof(PanelGroup)));
    CSRK_TRY(stage_h2d(pn->group.p, groups.data(), groups.size() * sizeof(PanelGroup), s));
    CSRK_TRY(pn->carry_row.alloc((size_t)n_tiles * 4));
    CSRK_TRY(pn->carry_val.alloc((size_t)n_tiles * 8));
    CSRK_TRY(pn->y.alloc((size_t)(pairs + (int64_t)ncs * n) * 8));
    if (ncs) {      // -0.0 where no tile writes (x + -0.0 = x for every x)
        std::vector<double> neg0((size_t)ncs * n, -0.0);
        CSRK_TRY(stage_h2d(pn->y.as<double>() + pairs, neg0.data(), neg0.size() * 8, s));
    }
    pn->ncs = ncs;
    CSRK_HIP(hipStreamSynchronize(s));     // `groups`, `t0` are host temporaries of async copies
    tr.lap("  tier 1: groups, carries (host)");
    pn->on = true;
    pn->cb = cb;
    pn->nb = nb;
    pn->nrow = n;
    pn->rows = pairs;
    pn->tiles = n_tiles;
    pn->nnz = nnz_rows;
    return CSRK_OK;
}

// Build one accumulator-form group: the listed heavy rows (<= ACC_MAXROWS, ascending) as a tiled,
// column-block-major (value, packed index) stream plus the persistent workgroups' segment lists.
template <class P, int VT>
static int build_acc_panel(Matrix *m, AccPanel *ap, const int32_t *rows, const int64_t *lens, int32_t n, int64_t nnz_rows,
                           hipStream_t s)
{
    const P *rp = (const P *)m->d_rowptrs;
    const int32_t nb = (int32_t)ceil_div(m->ncols > 0 ? m->ncols : 1, ACC_CB);
    PlanTrace tr;
    const int64_t pairs = (int64_t)n * nb;
    CSRK_TRY(ap->row_list.alloc((size_t)n * 4));
    CSRK_TRY(stage_h2d(ap->row_list.p, rows, (size_t)n * 4, s));
    DevBuf off, bends, pstart, d_trow, d_tpiece;
    CSRK_TRY(off.alloc((size_t)(pairs + 1) * 8));
    // block starts inside every row (one pass over the rows' entries), then the pair counts
    std::vector<int32_t> trow, tpiece;
    for (int32_t c = 0; c < n; c++)
        for (int64_t pc = 0; pc * ACC_PIECE < lens[c]; pc++) {
            trow.push_back(c);
            tpiece.push_back((int32_t)pc);
        }
    const int64_t n_tasks = (int64_t)trow.size();
    CSRK_TRY(pstart.alloc((size_t)(pairs + n) * 4));
    CSRK_TRY(d_trow.alloc((size_t)(n_tasks ? n_tasks : 1) * 4));
    CSRK_TRY(d_tpiece.alloc((size_t)(n_tasks ? n_tasks : 1) * 4));
    CSRK_TRY(stage_h2d(d_trow.p, trow.data(), (size_t)n_tasks * 4, s));
    CSRK_TRY(stage_h2d(d_tpiece.p, tpiece.data(), (size_t)n_tasks * 4, s));
    acc_pairstart_kernel<P><<<(unsigned)(ceil_div(ceil_div(n_tasks * WAVE, 256), 8) * 8), 256, 0, s>>>(
        rp, m->d_colinds, ap->row_list.as<int32_t>(), n, nb, ACC_CB, d_trow.as<int32_t>(), d_tpiece.as<int32_t>(), n_tasks,
        pstart.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    tr.lap("  tier 0: pair starts");
    acc_paircount_kernel<<<(unsigned)ceil_div(pairs, 256), 256, 0, s>>>(pstart.as<int32_t>(), pairs, n, off.as<int64_t>());
    CSRK_LAUNCH_CHECK();
    DevBuf gap;
    CSRK_TRY(gap.alloc((size_t)pairs * 4));
    acc_gap_kernel<<<(unsigned)ceil_div((int64_t)nb * WAVE, 256), 256, 0, s>>>(off.as<int64_t>(), n, nb, gap.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(exclusive_scan_i64(off.as<int64_t>(), off.as<int64_t>(), pairs, s));
    CSRK_TRY(bends.alloc((size_t)(nb + 1) * 8));
    panel_blockends_kernel<<<(unsigned)ceil_div(nb + 1, 256), 256, 0, s>>>(off.as<int64_t>(), n, nb, bends.as<int64_t>());
    CSRK_LAUNCH_CHECK();
    std::vector<int64_t> be((size_t)nb + 1), t0((size_t)nb + 1);
    CSRK_TRY(stage_d2h(be.data(), bends.p, (size_t)(nb + 1) * 8, s));
    CSRK_HIP(hipStreamSynchronize(s));
    t0[0] = 0;
    for (int32_t b = 0; b < nb; b++) t0[b + 1] = t0[b] + ceil_div(be[b + 1] - be[b], ACC_TILE);
    const int64_t n_tiles = t0[nb];
    CSRK_TRY(stage_h2d(bends.p, t0.data(), (size_t)(nb + 1) * 8, s));
    tr.lap("  tier 0: counts, gaps, scan");
    // persistent workgroups: one per CU, equal shares of the tiles (stored interleaved: acc_phys_tile), cut into one-block
    // segments
    int cus = 0;
    CSRK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->device));
    int64_t n_wg = cus > 0 ? cus : 256;
    if (n_wg > n_tiles) n_wg = n_tiles;
    if (n_wg < 1) n_wg = 1;
    std::vector<int64_t> wg_t0((size_t)n_wg + 1);
    int64_t share_max = 0;
    for (int64_t w = 0; w <= n_wg; w++) wg_t0[(size_t)w] = n_tiles * w / n_wg;
    for (int64_t w = 0; w < n_wg; w++) share_max = std::max(share_max, wg_t0[(size_t)w + 1] - wg_t0[(size_t)w]);
    const int64_t n_phys = share_max * n_wg;      // stored tiles (the last sweep has holes where a share is one tile shorter)
    DevBuf d_wg_t0;
    CSRK_TRY(d_wg_t0.alloc((size_t)(n_wg + 1) * 8));
    CSRK_TRY(stage_h2d(d_wg_t0.p, wg_t0.data(), (size_t)(n_wg + 1) * 8, s));
    // a float32 matrix keeps float32 values in the stream (6 B per entry; widened in the kernel, exactly)
    ap->f32 = VT == CSRK_VAL_F32;
    CSRK_TRY(ap->vals.alloc((size_t)n_phys * ACC_TILE * (ap->f32 ? 4 : 8)));
    CSRK_TRY(ap->idx.alloc((size_t)n_phys * ACC_TILE * 2));
    CSRK_TRY(ap->tile_row0.alloc((size_t)(n_tiles ? n_tiles : 1) * 4));
    const int32_t fill_waves = (int32_t)ceil_div(n, WAVE);      // wavefronts per block: 64 pairs each
#define ACC_FILL(SV)                                                                                                   \
    do {                                                                                                               \
        acc_fill_kernel<P, VT, SV><<<(unsigned)ceil_div((int64_t)nb * fill_waves * WAVE, 256), 256, 0, s>>>(           \
            rp, m->d_colinds, m->d_values, ap->row_list.as<int32_t>(), n, nb, ACC_CB, off.as<int64_t>(),               \
            bends.as<int64_t>(), pstart.as<int32_t>(), gap.as<int32_t>(), ap->vals.as<SV>(), ap->idx.as<uint16_t>(),   \
            ap->tile_row0.as<int32_t>(), d_wg_t0.as<int64_t>(), (int32_t)n_wg, fill_waves);                            \
        CSRK_LAUNCH_CHECK();                                                                                           \
        acc_pad_kernel<SV><<<(unsigned)nb, 256, 0, s>>>(off.as<int64_t>(), n, nb, bends.as<int64_t>(), ap->vals.as<SV>(), \
                                                        ap->idx.as<uint16_t>(), d_wg_t0.as<int64_t>(), (int32_t)n_wg); \
        CSRK_LAUNCH_CHECK();                                                                                           \
    } while (0)
    if (ap->f32) ACC_FILL(float);
    else ACC_FILL(double);
#undef ACC_FILL
    tr.lap("  tier 0: fill");
    std::vector<AccSeg> segs;
    std::vector<int32_t> wg_seg((size_t)n_wg + 1);
    int32_t b = 0;
    for (int64_t w = 0; w < n_wg; w++) {
        wg_seg[(size_t)w] = (int32_t)segs.size();
        int64_t t = wg_t0[(size_t)w];
        const int64_t t_end = wg_t0[(size_t)w + 1];
        while (t < t_end) {
            while (t0[b + 1] <= t) b++;
            int64_t e = t_end < t0[b + 1] ? t_end : t0[b + 1];
            if (e - t > ACC_SEG_TILES) e = t + ACC_SEG_TILES;
            AccSeg sg;
            sg.tile0 = t;
            sg.ptile0 = (t - wg_t0[(size_t)w]) * n_wg + w;
            sg.ntiles = (int32_t)(e - t);
            sg.blk = b;
            segs.push_back(sg);
            t = e;
        }
    }
    wg_seg[(size_t)n_wg] = (int32_t)segs.size();
    CSRK_TRY(ap->segs.alloc(segs.size() * sizeof(AccSeg)));
    CSRK_TRY(ap->wg_seg.alloc(wg_seg.size() * 4));
    CSRK_TRY(stage_h2d(ap->segs.p, segs.data(), segs.size() * sizeof(AccSeg), s));
    CSRK_TRY(stage_h2d(ap->wg_seg.p, wg_seg.data(), wg_seg.size() * 4, s));
    CSRK_TRY(ap->partial.alloc((size_t)n_wg * n * 8));
    CSRK_TRY(ap->z.alloc((size_t)n * 8));
    ap->lds = (size_t)(ACC_CB + 2) * 8 + (size_t)((n + 1) & ~1) * 8 + (size_t)ACC_SEG_TILES * 12;
    CSRK_TRY(spmv_kernel_attributes());
    CSRK_HIP(hipStreamSynchronize(s));     // `segs`, `wg_seg`, `t0` are host temporaries of async copies
    ap->nrow = n;
    ap->nb = nb;
    ap->n_wg = (int32_t)n_wg;
    ap->tiles = n_tiles;
    ap->nnz = nnz_rows;
    ap->n_segs = (int64_t)segs.size();
    tr.lap("  tier 0: segments");
    return CSRK_OK;
}

// Cut the long rows out of the merge path and build their panel tiers.

template <class P>
static int build_heavy_split(Matrix *m, SpmvPlan *p, hipStream_t s, bool allow_tier1 = true)
{
    p->n_heavy = 0;
    const char *env = getenv("CSRK_SPMV_HEAVY_SPLIT");
    if (env && env[0] == '0') return CSRK_OK;
    HEAVY_MIN = 2048;
    TIERB_MIN = 128;
    // (test hooks, not tuning switches: a small matrix gets more "heavy" rows than one accumulator group holds, or no tier 1)
    if (const char *e = getenv("CSRK_HEAVY_MIN")) HEAVY_MIN = atoi(e) > 64 ? atoi(e) : 64;
    if (const char *e = getenv("CSRK_TIERB_MIN")) TIERB_MIN = atoi(e) >= 0 ? atoi(e) : 0;
    const bool tier1 = allow_tier1 && TIERB_MIN > 0 && TIERB_MIN < HEAVY_MIN;
    const int cut_min = tier1 ? TIERB_MIN : HEAVY_MIN;
    if (m->nrows == 0 || m->nnz < cut_min) return CSRK_OK;
    // The split pays for itself only when x does not fit in an XCD's 4 MiB L2: otherwise every gather
    // is an L2 hit already and the panels only add (block, row) overhead (MovieLens-25M shape, x = 472 KB:
    // 0.146 ms on the single merge path against 0.176-0.53 ms split; measured).  CSRK_SPMV_HEAVY_SPLIT=1 forces it.
    if ((int64_t)m->ncols * 8 <= (4ll << 20) && !(env && env[0] == '1')) return CSRK_OK;
    const P *rp = (const P *)m->d_rowptrs;
    const int32_t nr = m->nrows;
    const unsigned g1 = (unsigned)ceil_div((int64_t)nr + 1, 256);
    PlanTrace tr;
    DevBuf flag, hlen, bad, clen;
    CSRK_TRY(flag.alloc((size_t)(nr + 2) * 4));
    CSRK_TRY(hlen.alloc((size_t)(nr + 2) * 8));
    heavy_flag_kernel<P><<<g1, 256, 0, s>>>(rp, nr, flag.as<int32_t>(), hlen.as<int64_t>(), cut_min);
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(exclusive_scan_i32(flag.as<int32_t>(), flag.as<int32_t>(), nr, s));      // -> cut-row index
    CSRK_TRY(exclusive_scan_i64(hlen.as<int64_t>(), hlen.as<int64_t>(), nr, s));      // -> cut entries before
    int32_t n_cut = 0;
    int64_t nnz_cut = 0;
    CSRK_TRY(stage_d2h(&n_cut, flag.as<int32_t>() + nr, 4, s));
    CSRK_TRY(stage_d2h(&nnz_cut, hlen.as<int64_t>() + nr, 8, s));
    CSRK_HIP(hipStreamSynchronize(s));
    tr.lap("  split: flags + scans");
    if (n_cut == 0) return CSRK_OK;

    CSRK_TRY(p->rp_light.alloc((size_t)(nr + 1) * sizeof(P)));
    CSRK_TRY(p->heavy_row.alloc((size_t)n_cut * 4));
    CSRK_TRY(p->cut_pos.alloc((size_t)n_cut * 8));
    CSRK_TRY(p->cut_cum.alloc((size_t)(n_cut + 1) * 8));
    CSRK_TRY(clen.alloc((size_t)n_cut * 8));
    heavy_view_kernel<P><<<g1, 256, 0, s>>>(rp, nr, flag.as<int32_t>(), hlen.as<int64_t>(), p->rp_light.as<P>(),
                                          p->heavy_row.as<int32_t>(), p->cut_pos.as<int64_t>(),
                                          p->cut_cum.as<int64_t>(), clen.as<int64_t>(), n_cut);
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(bad.alloc(4));
    CSRK_HIP(hipMemsetAsync(bad.p, 0, 4, s));
    heavy_sorted_kernel<P><<<(unsigned)n_cut, 1024, 0, s>>>(rp, m->d_colinds, p->heavy_row.as<int32_t>(), n_cut,
                                                         bad.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    tr.lap("  split: view + sorted kernels");
    int32_t is_bad = 0;
    std::vector<int32_t> rows((size_t)n_cut);
    std::vector<int64_t> lens((size_t)n_cut);
    CSRK_TRY(stage_d2h(&is_bad, bad.p, 4, s));
    CSRK_TRY(stage_d2h(rows.data(), p->heavy_row.p, (size_t)n_cut * 4, s));
    CSRK_TRY(stage_d2h(lens.data(), clen.p, (size_t)n_cut * 8, s));
    CSRK_HIP(hipStreamSynchronize(s));
    if (tr.on) fprintf(stderr, "[csrk plan]   n_cut %d\n", n_cut);
    tr.lap("  split: copies to the host");
    if (is_bad) return CSRK_OK;      // unsorted columns in a long row: column blocking needs order

    // tier 0: accumulator form (groups of <= ACC_MAXROWS rows).  The accumulator form costs 1.8 ps per entry against 4.8 for tier 1 (measured, headline matrix), and
    // one group holds up to ACC_MAXROWS rows at no extra window traffic: when fewer rows than that reach
    // HEAVY_MIN, tier 0 is extended downwards to the ACC_MAXROWS longest rows (not below ACC_FLOOR).
    if (tier1 && !getenv("CSRK_HEAVY_MIN")) {
        int64_t n_min = 0;
        for (int32_t c = 0; c < n_cut; c++) n_min += lens[c] >= HEAVY_MIN;
        if (n_min < ACC_MAXROWS && n_cut > n_min) {
            const size_t kth = (size_t)(n_cut < ACC_MAXROWS ? n_cut : ACC_MAXROWS) - 1;
            // the kth longest row's length: a histogram of the lengths below 2^16 decides it (nth_element over 10^5 rows was
            // 0.4 ms of the plan), unless the kth row is longer than that
            int64_t thr = -1;
            {
                constexpr int64_t HB = 1 << 16;
                std::vector<int32_t> hist((size_t)HB + 1, 0);
                for (int32_t c = 0; c < n_cut; c++) hist[(size_t)(lens[c] < HB ? lens[c] : HB)]++;
                int64_t seen = hist[(size_t)HB];
                if (seen <= (int64_t)kth)
                    for (int64_t v = HB - 1; v >= 0; v--) {
                        seen += hist[(size_t)v];
                        if (seen > (int64_t)kth) {
                            thr = v;
                            break;
                        }
                    }
            }
            if (thr < 0) {
                std::vector<int64_t> sl(lens);
                std::nth_element(sl.begin(), sl.begin() + kth, sl.end(), [](int64_t a, int64_t b) { return a > b; });
                thr = sl[kth];
            }
            // rows tied with the kth must not push the group over its capacity
            int64_t n_ge = 0;
            for (int32_t c = 0; c < n_cut; c++) n_ge += lens[c] >= thr;
            if (n_ge > ACC_MAXROWS) thr++;
            thr = thr < ACC_FLOOR ? ACC_FLOOR : thr;
            if (thr < HEAVY_MIN) HEAVY_MIN = (int)thr;
        }
    }
    p->heavy_min = HEAVY_MIN;
    p->tier1_min = TIERB_MIN;
    std::vector<int32_t> r0, r1;     // tier 0: >= HEAVY_MIN entries; tier 1: the rest of the cut rows
    std::vector<int64_t> len0;
    int64_t nnz1 = 0;
    r0.reserve((size_t)n_cut);
    r1.reserve((size_t)n_cut);
    len0.reserve((size_t)n_cut);
    for (int32_t c = 0; c < n_cut; c++) {
        if (lens[c] >= HEAVY_MIN) {
            r0.push_back(rows[c]);
            len0.push_back(lens[c]);
        } else {
            r1.push_back(rows[c]);
            nnz1 += lens[c];
        }
    }
    const int64_t pair_cap = 256ll << 20;
    const int64_t pairs1 = (int64_t)r1.size() * ceil_div(m->ncols > 0 ? m->ncols : 1, PANEL_CB1);
    if (pairs1 > pair_cap && tier1) return build_heavy_split<P>(m, p, s, false);

    p->n_heavy = n_cut;
    p->nnz_light = m->nnz - nnz_cut;
    p->t0_rows = std::move(r0);
    p->t0_lens = std::move(len0);
    p->t1_rows = std::move(r1);
    p->t1_nnz = nnz1;
    tr.lap("  split: host lists");
    return CSRK_OK;
}

// Build the tiers of the rows build_heavy_split cut out.
template <class P>
static int build_tiers(Matrix *m, SpmvPlan *p, hipStream_t s)
{
    if (!p->n_heavy) return CSRK_OK;
    const std::vector<int32_t> &r0 = p->t0_rows, &r1 = p->t1_rows;
    const std::vector<int64_t> &len0 = p->t0_lens;
    const int64_t nnz1 = p->t1_nnz;
#define BUILD(VT)                                                                                                  \
    do {                                                                                                           \
        for (size_t g0 = 0; g0 < r0.size(); g0 += ACC_MAXROWS) {                                                   \
            const size_t g1 = g0 + ACC_MAXROWS < r0.size() ? g0 + ACC_MAXROWS : r0.size();                         \
            int64_t gn = 0;                                                                                        \
            for (size_t c = g0; c < g1; c++) gn += len0[c];                                                        \
            AccPanel *ap = new (std::nothrow) AccPanel();                                                          \
            CSRK_REQUIRE(ap, "out of host memory");                                                                \
            p->acc.push_back(ap);                                                                                  \
            CSRK_TRY((build_acc_panel<P, VT>(m, ap, r0.data() + g0, len0.data() + g0, (int32_t)(g1 - g0), gn, s)));  \
        }                                                                                                          \
        if (!r1.empty()) CSRK_TRY((build_panel<P, VT>(m, &p->tier1, r1, nnz1, PANEL_CB1, 1, true, s)));            \
    } while (0)
    if (m->val_type == CSRK_VAL_F64) BUILD(CSRK_VAL_F64);
    else if (m->val_type == CSRK_VAL_F32) BUILD(CSRK_VAL_F32);
    else BUILD(CSRK_VAL_NONE);
#undef BUILD
    return CSRK_OK;
}

// Pick the (at most HOT_SLOTS) most referenced columns and renumber them in a copy of colinds.  Built
// with the lazy plan (second launch on a handle).  Skipped when x is small enough to live in L1/L2
// next to the streams anyway, when the matrix is small, or when the cached columns would carry less
// than a fifth of the entries (no popularity skew: nothing to gain, and the persistent grid has fewer
// wavefronts in flight than the plain one).  CSRK_SPMV_HOT=0 disables, =1 forces.
template <class P>
static int build_hot_cache(Matrix *m, SpmvPlan *p, hipStream_t s)
{
    p->n_hot = 0;
    const char *env = getenv("CSRK_SPMV_HOT");
    if (env && env[0] == '0') return CSRK_OK;
    const bool force = env && env[0] == '1';
    if (m->nnz < 2 || m->ncols < 1) return CSRK_OK;
    PlanTrace tr;
    int64_t HOT_SLOTS = 524288;   // 4 MiB of packed x, most popular first (light stream, LDS for the first 8192: 64k 0.340,
                                  // 256k 0.308, 512k 0.302, 1M 0.297, 2M 0.297 ms, but the per-call pack costs more than that gains past 512k)
                                  // earlier sweep, tile kernel, 512 KiB of packed x (measured on the headline matrix: 16k 0.431, 64k 0.422,
                                  // 256k 0.430, 1M 0.439, 4M 0.460 ms for the tile kernel; none 0.481)
    p->hot_slots = (int32_t)HOT_SLOTS;
    // x that fits in L2 whole needs no packing
    if (!force && (m->nnz < (1 << 20) || (int64_t)m->ncols * 8 <= (4ll << 20))) return CSRK_OK;
    const int32_t nc = m->ncols;
    DevBuf cnt, slot, census;
    CSRK_TRY(cnt.alloc((size_t)(nc + 1) * 4));
    CSRK_TRY(slot.alloc((size_t)(nc + 2) * 4));
    CSRK_TRY(census.alloc(16));
    CSRK_HIP(hipMemsetAsync(cnt.p, 0, (size_t)(nc + 1) * 4, s));
    CSRK_HIP(hipMemsetAsync(census.p, 0, 16, s));
    const int64_t row_stride = p->nnz_light > (1ll << 25) ? p->nnz_light >> 25 : 1;
    const int64_t hc_need = ceil_div(ceil_div(m->nrows, WAVE * row_stride), HOT_THREADS / WAVE);
    hot_count_kernel<P><<<(unsigned)(hc_need < 1024 ? hc_need : 1024), HOT_THREADS, 0, s>>>(
        (const P *)m->d_rowptrs, p->n_heavy ? p->rp_light.as<P>() : (const P *)nullptr, m->d_colinds, m->nrows,
        row_stride, cnt.as<int32_t>(), census.as<unsigned long long>());
    CSRK_LAUNCH_CHECK();
    unsigned long long n_samples_u = 0;
    CSRK_TRY(stage_d2h(&n_samples_u, census.p, 8, s));
    CSRK_HIP(hipStreamSynchronize(s));
    const int64_t n_samples = (int64_t)n_samples_u;
    tr.lap("  hot: column counts");
    if (n_samples == 0) return CSRK_OK;
    // smallest threshold (>= 2 references in the sample) that leaves at most HOT_SLOTS columns
    auto census_at = [&](int32_t thr, unsigned long long out[2]) -> int {
        CSRK_HIP(hipMemsetAsync(census.p, 0, 16, s));
        hot_census_kernel<<<1024, 256, 0, s>>>(cnt.as<int32_t>(), nc, thr, census.as<unsigned long long>());
        CSRK_LAUNCH_CHECK();
        CSRK_TRY(stage_d2h(out, census.p, 16, s));
        CSRK_HIP(hipStreamSynchronize(s));
        return CSRK_OK;
    };
    unsigned long long c[2];
    int64_t lo = 2, hi = n_samples + 1;     // invariant: census(hi).n <= HOT_SLOTS
    // one histogram pass decides it unless more than HOT_SLOTS columns sit in the shared top bin (then: the search below)
    bool decided = false;
    {
        DevBuf hist;
        CSRK_TRY(hist.alloc((size_t)2 * (HOT_HIST + 1) * 8));
        CSRK_HIP(hipMemsetAsync(hist.p, 0, (size_t)2 * (HOT_HIST + 1) * 8, s));
        hot_hist_kernel<<<1024, 256, 0, s>>>(cnt.as<int32_t>(), nc, hist.as<unsigned long long>(),
                                             hist.as<unsigned long long>() + (HOT_HIST + 1));
        CSRK_LAUNCH_CHECK();
        std::vector<unsigned long long> hh((size_t)2 * (HOT_HIST + 1));
        CSRK_TRY(stage_d2h(hh.data(), hist.p, hh.size() * 8, s));
        CSRK_HIP(hipStreamSynchronize(s));
        const unsigned long long *hn = hh.data(), *hs = hh.data() + (HOT_HIST + 1);
        if (hn[HOT_HIST] <= (unsigned long long)HOT_SLOTS) {
            unsigned long long n_ge = 0, s_ge = 0;      // columns / references with count >= v
            int64_t thr0 = HOT_HIST;                     // smallest threshold >= 2 that leaves at most HOT_SLOTS columns
            n_ge = hn[HOT_HIST];
            s_ge = hs[HOT_HIST];
            for (int64_t v = HOT_HIST - 1; v >= 2; v--) {
                if (n_ge + hn[v] > (unsigned long long)HOT_SLOTS) break;
                n_ge += hn[v];
                s_ge += hs[v];
                thr0 = v;
            }
            lo = thr0;
            c[0] = n_ge;
            c[1] = s_ge;
            decided = true;
        }
    }
    if (!decided) CSRK_TRY(census_at((int32_t)lo, c));
    if (!decided && c[0] > (unsigned long long)HOT_SLOTS) {
        while (lo + 1 < hi) {
            const int64_t mid = lo + (hi - lo) / 2;
            CSRK_TRY(census_at((int32_t)(mid > INT32_MAX ? INT32_MAX : mid), c));
            if (c[0] <= (unsigned long long)HOT_SLOTS)
                hi = mid;
            else
                lo = mid;
        }
        CSRK_TRY(census_at((int32_t)(hi > INT32_MAX ? INT32_MAX : hi), c));
        lo = hi;
    }
    const int32_t thr = (int32_t)(lo > INT32_MAX ? INT32_MAX : lo);
    const int32_t n_hot = (int32_t)c[0];
    tr.lap("  hot: threshold");
    p->hot_cover = n_samples ? (double)c[1] / (double)n_samples : 0.0;
    if (n_hot == 0 || (!force && p->hot_cover < 0.2)) return CSRK_OK;

    // Slots in order of popularity (count descending, column ascending among equals): the first
    // LS_HOT_LDS slots are the ones the light stream keeps in LDS, and the packed lines that follow
    // are referenced less and less often, so what L2 fails to retain is the pack's tail.
    const unsigned gc = (unsigned)ceil_div((int64_t)nc + 1, 256);
    hot_flag_kernel<<<gc, 256, 0, s>>>(cnt.as<int32_t>(), nc, thr, slot.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(exclusive_scan_i32(slot.as<int32_t>(), slot.as<int32_t>(), nc, s));
    CSRK_TRY(p->hot_cols.alloc((size_t)n_hot * 4));
    DevBuf hcnt;
    CSRK_TRY(hcnt.alloc((size_t)n_hot * 4));
    hot_list_kernel<<<gc, 256, 0, s>>>(cnt.as<int32_t>(), slot.as<int32_t>(), nc, thr, p->hot_cols.as<int32_t>(),
                                      hcnt.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    {
        // stable, count descending (hot_sort_*_kernel): four 8-bit passes, ping-pong between (hcnt, hot_cols) and (k2, v2)
        const int32_t nchunk = (int32_t)ceil_div(n_hot, HS_CHUNK);
        DevBuf k2, v2, dc;
        CSRK_TRY(k2.alloc((size_t)n_hot * 4));
        CSRK_TRY(v2.alloc((size_t)n_hot * 4));
        CSRK_TRY(dc.alloc((size_t)(256 * (int64_t)nchunk + 1) * 4));
        uint32_t *ka = hcnt.as<uint32_t>(), *kb = k2.as<uint32_t>();
        int32_t *va = p->hot_cols.as<int32_t>(), *vb = v2.as<int32_t>();
        for (int pass = 0; pass < 4; pass++) {
            hot_sort_count_kernel<<<(unsigned)nchunk, WAVE, 0, s>>>(ka, n_hot, 8 * pass, nchunk, dc.as<int32_t>());
            CSRK_LAUNCH_CHECK();
            CSRK_TRY(exclusive_scan_i32(dc.as<int32_t>(), dc.as<int32_t>(), 256 * (int64_t)nchunk, s));
            hot_sort_place_kernel<<<(unsigned)nchunk, WAVE, 0, s>>>(ka, va, n_hot, 8 * pass, nchunk, dc.as<int32_t>(), kb, vb);
            CSRK_LAUNCH_CHECK();
            std::swap(ka, kb);
            std::swap(va, vb);
        }
        CSRK_HIP(hipStreamSynchronize(s));      // k2, v2, dc are freed here (the sorted columns are back in hot_cols)
    }
    tr.lap("  hot: list, order by count");
    // column -> slot map (-1: not packed); the light stream's fill reads it, as does the renumbered colinds copy
    // the tile kernel needs when no stream is built
    CSRK_TRY(p->hot_slot.alloc((size_t)nc * 4));
    CSRK_HIP(hipMemsetAsync(p->hot_slot.p, 0xff, (size_t)nc * 4, s));
    hot_slot_kernel<<<(unsigned)ceil_div(n_hot, 256), 256, 0, s>>>(p->hot_cols.as<int32_t>(), n_hot, p->hot_slot.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(p->xh.alloc((size_t)n_hot * 8));
    const int32_t n_top = n_hot < LS_HOT_LDS ? n_hot : LS_HOT_LDS;
    p->n_hot = n_hot;
    p->n_hot_lds = n_top;
    CSRK_HIP(hipStreamSynchronize(s));     // hcnt is freed on return
    return CSRK_OK;
}

// Build the arrays of the light stream from a view: view row r (r < nrows_view) is the source entries
// src[r] .. src[r] + (rpv[r+1] - rpv[r]) of (ci, vs); n_ent entries in n_tiles tiles.
template <class P, int VT>
static int build_stream(Matrix *m, LightStream *ls, const P *src, const P *rpv, int32_t nrows_view, const int32_t *ci,
                        const void *vs, int64_t n_ent, int64_t n_tiles, int32_t n_out, const int32_t *slot_map, hipStream_t s,
                        const P *rp_len = nullptr)
{
    // rp_len (dense rows): rpv gives every row of the view at least one slot; a row that is empty in rp_len is one padding
    // entry.  Run k is then row k, and no row-id table is built.
    ls->on = false;
    ls->dense = rp_len != nullptr;
    PlanTrace tr;
    DevBuf ridx;
    CSRK_TRY(ridx.alloc((size_t)(nrows_view + 2) * 4));
    const unsigned gr = (unsigned)ceil_div((int64_t)nrows_view + 1, 256);
    ls_rowflag_kernel<P><<<gr, 256, 0, s>>>(rpv, nrows_view, ridx.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(exclusive_scan_i32(ridx.as<int32_t>(), ridx.as<int32_t>(), nrows_view, s));
    int32_t n_runs = 0;
    CSRK_TRY(stage_d2h(&n_runs, ridx.as<int32_t>() + nrows_view, 4, s));
    CSRK_HIP(hipStreamSynchronize(s));
    if (n_runs < 1) return CSRK_OK;
    if (!ls->dense) {
        CSRK_TRY(ls->rowids.alloc((size_t)n_runs * 4));
        ls_rowids_kernel<P><<<gr, 256, 0, s>>>(rpv, nrows_view, ridx.as<int32_t>(), ls->rowids.as<int32_t>());
        CSRK_LAUNCH_CHECK();
    }
    ls->f32 = VT == CSRK_VAL_F32;      // a float32 matrix keeps float32 values in the stream (widened in the kernel, exactly)
    CSRK_TRY(ls->vals.alloc((size_t)n_tiles * ACC_TILE * (ls->f32 ? 4 : 8)));
    CSRK_TRY(ls->idx.alloc((size_t)n_tiles * ACC_TILE * 4));
    ls->idx24 = false;
    DevBuf tile_row;
    CSRK_TRY(tile_row.alloc((size_t)(n_tiles + 1) * 4));
    ls_tilerow_kernel<P><<<(unsigned)ceil_div(n_tiles + 1, 256), 256, 0, s>>>(rpv, nrows_view, n_ent, n_tiles, tile_row.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    if (ls->f32)
        ls_fill_kernel<P, VT, float><<<(unsigned)ceil_div(n_tiles * ACC_TILE, 256), 256, 0, s>>>(
            src, rpv, nrows_view, ci, vs, n_ent, n_tiles * ACC_TILE, slot_map, ls->vals.as<float>(), ls->idx.as<uint32_t>(), rp_len,
            tile_row.as<int32_t>());
    else
        ls_fill_kernel<P, VT, double><<<(unsigned)ceil_div(n_tiles * ACC_TILE, 256), 256, 0, s>>>(
            src, rpv, nrows_view, ci, vs, n_ent, n_tiles * ACC_TILE, slot_map, ls->vals.as<double>(), ls->idx.as<uint32_t>(), rp_len,
            tile_row.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(ls->tile_base.alloc((size_t)n_tiles * 4));
    ls_tilebase_kernel<P><<<(unsigned)ceil_div(n_tiles, 256), 256, 0, s>>>(
        rpv, nrows_view, ridx.as<int32_t>(), n_tiles, tile_row.as<int32_t>(), ls->tile_base.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(ls->carry_row.alloc((size_t)n_tiles * 4));
    CSRK_TRY(ls->carry_val.alloc((size_t)n_tiles * 8));
    int cus = 0;
    CSRK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->device));
    int64_t wgs = (int64_t)(cus > 0 ? cus : 256);
    CSRK_TRY(spmv_kernel_attributes());
    const int64_t need = ceil_div(n_tiles, LS_THREADS / WAVE);
    ls->grid = (unsigned)(wgs < need ? wgs : need);
    ls->n_tiles = n_tiles;
    ls->n_runs = n_runs;
    ls->n_out = n_out;
    CSRK_HIP(hipStreamSynchronize(s));      // ridx, dphys are freed on return; *phys is a host temporary
    tr.lap("  light: runs, fill");
    ls->on = true;
    return CSRK_OK;
}
// per index word: count into the (round, block) bucket; the old count is the entry's place inside the bucket
__global__ __launch_bounds__(256) void ls_cold_count_kernel(const uint32_t *__restrict__ sidx, int64_t n_words, int32_t nblk, int32_t W,
                                                           const int32_t *__restrict__ tile_round, int32_t *__restrict__ cnt,
                                                           int32_t *__restrict__ off)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    const uint32_t ix = sidx[w];
    if (!ls_is_cold(ix)) return;
    const int64_t r = tile_round[w / ACC_TILE];
    const int32_t b = (int32_t)((ix & LS_COL_MASK) / (uint32_t)W);
    off[w] = atomicAdd(&cnt[r * nblk + b], 1);
}

__global__ void ls_cold_transpose_kernel(const int32_t *__restrict__ cnt, int32_t nround, int32_t nblk, int32_t *__restrict__ cntT)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)nround * nblk) return;
    const int64_t r = i / nblk, b = i % nblk;
    cntT[b * nround + r] = cnt[i];
}

// position in xg = bucket base in (round, block) order + place; position in the copy list = bucket base in
// (block, round) order + place
// (rel: the index word keeps the position relative to the round's start -- the round-in-LDS form)
__global__ __launch_bounds__(256) void ls_cold_place_kernel(uint32_t *__restrict__ sidx, int64_t n_words, int32_t nround,
                                                           int32_t nblk, int32_t W, const int32_t *__restrict__ tile_round, int rel,
                                                           const int32_t *__restrict__ base_rb,
                                                           const int32_t *__restrict__ base_br, const int32_t *__restrict__ off,
                                                           uint16_t *__restrict__ a_col, int32_t *__restrict__ a_dst)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    const uint32_t ix = sidx[w];
    if (!ls_is_cold(ix)) return;
    const int64_t r = tile_round[w / ACC_TILE];
    const uint32_t c = ix & LS_COL_MASK;
    const int32_t b = (int32_t)(c / (uint32_t)W);
    const int32_t pos = base_rb[r * nblk + b] + off[w];
    const int32_t pa = base_br[(int64_t)b * nround + r] + off[w];
    a_col[pa] = (uint16_t)(c - (uint32_t)b * (uint32_t)W);      // offset inside the block's window
    a_dst[pa] = pos;
    sidx[w] = (ix & LS_START_BIT) | (uint32_t)(rel ? pos - base_rb[r * nblk] : pos);
}

// round_start[r] = position in xg of round r's first staged value, r = 0 .. nround_ls (the last = n_cold)
__global__ void ls_round_start_kernel(const int32_t *__restrict__ base_rb, int32_t nround_ls, int32_t nblk,
                                      int32_t *__restrict__ round_start)
{
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= nround_ls) round_start[r] = base_rb[(int64_t)r * nblk];
}

__global__ void ls_stage_starts_kernel(const int32_t *__restrict__ base_br, int32_t nround, int32_t nblk, int32_t *__restrict__ blk_start)
{
    const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b <= nblk) blk_start[b] = base_br[(int64_t)b * nround];      // base_br has nround * nblk + 1 entries
}

// The packed columns ride along: slot k of the pack is entry (column hot_cols[k], position n_cold + k) of a virtual
// last round, so xh = xg + n_cold is filled by the same pass and hot_pack_kernel (10^5.6 gathers of 128 B) goes.
__global__ void ls_pack_count_kernel(const int32_t *__restrict__ hot_cols, int32_t n_hot, int32_t W, int32_t *__restrict__ cnt_last,
                                     int32_t *__restrict__ offp)
{
    const int32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_hot) offp[k] = atomicAdd(&cnt_last[hot_cols[k] / W], 1);
}

__global__ void ls_pack_place_kernel(const int32_t *__restrict__ hot_cols, int32_t n_hot, int32_t W, int32_t nround_all,
                                     const int32_t *__restrict__ base_br, const int32_t *__restrict__ offp, int32_t n_cold,
                                     uint16_t *__restrict__ a_col, int32_t *__restrict__ a_dst)
{
    const int32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_hot) return;
    const int32_t c = hot_cols[k];
    const int32_t pa = base_br[(int64_t)(c / W) * nround_all + (nround_all - 1)] + offp[k];
    a_col[pa] = (uint16_t)(c % W);
    a_dst[pa] = n_cold + k;
}

// Round-in-LDS form (default; CSRK_LS_RND=0 for the round-major form read by gathers): the round is the largest number
// of tiles (a multiple of the workgroup's wavefronts, at most LS_RND_MAXTILES) whose staged values fit LS_RND_CAP in
// every round.

// tile_round[t] = the round that holds tile t (round r = tiles round_tile0[r] .. round_tile0[r + 1])
__global__ void ls_tile_round_kernel(const int32_t *__restrict__ round_tile0, int32_t n_rounds, int64_t n_tiles,
                                     int32_t *__restrict__ tile_round)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    int32_t lo = 0, hi = n_rounds - 1;
    while (lo < hi) {
        const int32_t mid = lo + ((hi - lo + 1) >> 1);
        if ((int64_t)round_tile0[mid] <= t)
            lo = mid;
        else
            hi = mid - 1;
    }
    tile_round[t] = lo;
}

__global__ void ls_round_total_kernel(const int64_t *__restrict__ tot, int32_t nround_ls, int32_t nblk, int64_t *__restrict__ out)
{
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= nround_ls) out[r] = tot[(int64_t)r * nblk];
}

// tile_cold[t] = cold entries of tile t (one wavefront per tile): with their prefix sums the host finds the largest round
// size whose rounds all fit the LDS round buffer without counting buckets once per candidate size
__global__ __launch_bounds__(256) void ls_tile_cold_kernel(const uint32_t *__restrict__ sidx, int64_t n_tiles, int32_t *__restrict__ tile_cold)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t t = g / WAVE;
    const int lane = (int)(g % WAVE);
    if (t >= n_tiles) return;
    const u32x4_t *ip = (const u32x4_t *)(sidx + t * ACC_TILE);
    const u32x4_t a = ip[lane], b = ip[WAVE + lane];
    int c = (int)ls_is_cold(a.x) + (int)ls_is_cold(a.y) + (int)ls_is_cold(a.z) + (int)ls_is_cold(a.w) + (int)ls_is_cold(b.x) +
            (int)ls_is_cold(b.y) + (int)ls_is_cold(b.z) + (int)ls_is_cold(b.w);
    c = wave_exscan_i32(c, lane) + c;
    if (lane == WAVE - 1) tile_cold[t] = c;
}

// One thread per (tile, lane): the lane's eight uint32 index words -> their 3-byte form (LS24_*: spmv_plan.h).
__global__ __launch_bounds__(256) void ls_idx24_kernel(const uint32_t *__restrict__ sidx, int64_t n_tiles, unsigned char *__restrict__ out)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t t = g / WAVE;
    const int lane = (int)(g % WAVE);
    if (t >= n_tiles) return;
    const u32x4_t *ip = (const u32x4_t *)(sidx + t * ACC_TILE);
    const u32x4_t a = ip[lane], b = ip[WAVE + lane];
    const uint32_t w[ACC_K] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t n[ACC_K];
#pragma unroll
    for (int j = 0; j < ACC_K; j++) {
        const uint32_t c = w[j] & LS_COL_MASK;
        n[j] = (w[j] & LS_HOT_BIT ? LS24_HOT_BIT : 0u) | (w[j] & LS_START_BIT ? 1u << LS24_START_SHIFT : 0u) |
               (c == LS_PAD ? LS24_COL_MASK : c);
    }
    u32x4_t lo;
    lo.x = (n[0] & 0xffffu) | (n[1] << 16);
    lo.y = (n[2] & 0xffffu) | (n[3] << 16);
    lo.z = (n[4] & 0xffffu) | (n[5] << 16);
    lo.w = (n[6] & 0xffffu) | (n[7] << 16);
    u32x2_t hi;
    hi.x = (n[0] >> 16) | ((n[1] >> 16) << 8) | ((n[2] >> 16) << 16) | ((n[3] >> 16) << 24);
    hi.y = (n[4] >> 16) | ((n[5] >> 16) << 8) | ((n[6] >> 16) << 16) | ((n[7] >> 16) << 24);
    unsigned char *tp = out + t * LS24_TILE_BYTES;
    ((u32x4_t *)tp)[lane] = lo;
    ((u32x2_t *)(tp + ACC_TILE * 2))[lane] = hi;
}

static int build_cold_stage(Matrix *m, LightStream *ls, const int32_t *hot_cols, int32_t n_hot, hipStream_t s)
{
    ls->idx24 = false;
    ls->n_cold = 0;
    ls->stage_tiles = 0;
    const char *env = getenv("CSRK_LS_STAGE");
    if (env && env[0] == '0') return CSRK_OK;
    const int64_t n_words = ls->n_tiles * ACC_TILE;
    // a number of column blocks that fills the chip a whole number of times (two workgroups per CU), each window
    // at most LS_STAGE_WMAX columns
    int cus = 0;
    CSRK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->device));
    const int64_t wave_of_wgs = 2 * (int64_t)(cus > 0 ? cus : 256);
    const int64_t nblk_goal = wave_of_wgs * ceil_div((int64_t)m->ncols, wave_of_wgs * LS_STAGE_WMAX);
    const int64_t W = ceil_div(ceil_div((int64_t)m->ncols, nblk_goal), 16) * 16;
    const int64_t nblk = ceil_div((int64_t)m->ncols, W);
    // (the smallest rounds make the most buckets; a balanced layout adds at most one round per workgroup)
    const int64_t nb_max = (ceil_div(ls->n_tiles, LS_STAGE_TILES < LS_THREADS / WAVE ? LS_STAGE_TILES : LS_THREADS / WAVE) + (int64_t)ls->grid + 2) * nblk;
    if (nb_max < 1 || nb_max > (int64_t)1 << 26) return CSRK_OK;
    size_t mfree = 0, mtotal = 0;
    CSRK_HIP(hipMemGetInfo(&mfree, &mtotal));
    if ((size_t)n_words * 20 + (size_t)nb_max * 16 + (64u << 20) > mfree) return CSRK_OK;
    if (ls->n_tiles >= INT32_MAX) return CSRK_OK;
    PlanTrace tr;
    DevBuf cnt, cntT, off, offp, tot, tile_round, d_rt0;
    CSRK_TRY(tile_round.alloc((size_t)ls->n_tiles * 4));
    CSRK_TRY(d_rt0.alloc((size_t)(ls->n_tiles + 2) * 4));
    CSRK_TRY(offp.alloc((size_t)n_hot * 4));
    CSRK_TRY(cnt.alloc((size_t)(nb_max + 1) * 4));
    CSRK_TRY(cntT.alloc((size_t)(nb_max + 1) * 4));
    CSRK_TRY(off.alloc((size_t)n_words * 4));
    CSRK_TRY(tot.alloc((size_t)(nb_max + 1) * 8));
    const unsigned gw = (unsigned)ceil_div(n_words, 256);
    int stage_tiles = LS_STAGE_TILES;
    int64_t nround_ls = 0, nround = 0, nb = 0, n_cold = 0;      // n_cold = where the virtual round starts
    // counts per (round, block) bucket for rounds of `nt` tiles; the old count is an entry's place inside its bucket
    // Rounds of at most `nt` tiles: every workgroup of the stream's persistent grid gets an equal share of the tiles, cut
    // into equal rounds (whole tiles per wavefront).
    std::vector<int32_t> h_rt0, h_wr0;
    constexpr int NW_ = LS_THREADS / WAVE;
    auto make_rounds = [&](int nt) {
        h_rt0.clear();
        h_wr0.clear();
        const int64_t G = ls->grid;
        for (int64_t w = 0; w < G; w++) {
            const int64_t tb = ls->n_tiles * w / G, te = ls->n_tiles * (w + 1) / G;
            h_wr0.push_back((int32_t)h_rt0.size());
            if (te > tb) {
                const int64_t k = ceil_div(te - tb, nt);
                const int64_t per = ceil_div(ceil_div(te - tb, k), NW_) * NW_;
                for (int64_t t = tb; t < te; t += per) h_rt0.push_back((int32_t)t);
            }
        }
        h_wr0.push_back((int32_t)h_rt0.size());
        nround_ls = (int64_t)h_rt0.size();
        h_rt0.push_back((int32_t)ls->n_tiles);
    };
    auto count_pass = [&](int nt, int64_t *max_round) -> int {
        stage_tiles = nt;
        make_rounds(nt);
        nround = nround_ls + 1;      // + the virtual round of the packed columns
        nb = nround * nblk;
        if (nb > nb_max) return CSRK_ERR_INVALID;      // (cannot happen: a round holds at least LS_STAGE_TILES or NW tiles)
        CSRK_TRY(stage_h2d(d_rt0.p, h_rt0.data(), h_rt0.size() * 4, s));
        ls_tile_round_kernel<<<(unsigned)ceil_div(ls->n_tiles, 256), 256, 0, s>>>(d_rt0.as<int32_t>(), (int32_t)nround_ls, ls->n_tiles,
                                                                                tile_round.as<int32_t>());
        CSRK_LAUNCH_CHECK();
        CSRK_HIP(hipMemsetAsync(cnt.p, 0, (size_t)(nb + 1) * 4, s));
        ls_cold_count_kernel<<<gw, 256, 0, s>>>(ls->idx.as<uint32_t>(), n_words, (int32_t)nblk, (int32_t)W, tile_round.as<int32_t>(),
                                               cnt.as<int32_t>(), off.as<int32_t>());
        CSRK_LAUNCH_CHECK();
        // the counts are 32-bit: total them in 64 bits before trusting the 32-bit scans
        CSRK_TRY(exclusive_scan_i32_to_i64(cnt.as<int32_t>(), tot.as<int64_t>(), nround_ls * nblk, s));
        std::vector<int64_t> rs((size_t)nround_ls + 1);
        DevBuf drs;
        CSRK_TRY(drs.alloc((size_t)(nround_ls + 1) * 8));
        ls_round_total_kernel<<<(unsigned)ceil_div(nround_ls + 1, 256), 256, 0, s>>>(tot.as<int64_t>(), (int32_t)nround_ls, (int32_t)nblk,
                                                                                   drs.as<int64_t>());
        CSRK_LAUNCH_CHECK();
        CSRK_TRY(stage_d2h(rs.data(), drs.p, (size_t)(nround_ls + 1) * 8, s));
        CSRK_HIP(hipStreamSynchronize(s));
        n_cold = rs[(size_t)nround_ls];
        *max_round = 0;
        for (int64_t r = 0; r < nround_ls; r++) *max_round = std::max(*max_round, rs[(size_t)r + 1] - rs[(size_t)r]);
        return CSRK_OK;
    };
    // the largest rounds whose staged values fit the LDS round buffer (a round of one tile per wavefront always does)
    int64_t max_round = 0;
    {
        constexpr int NW = LS_THREADS / WAVE;      // a round is a whole number of tiles per wavefront
        static_assert(NW * ACC_TILE <= LS_RND_CAP, "the smallest round must fit the LDS round buffer");
        bool fits = false;
        // where to start: the largest size that fits by the tiles' own cold counts (then one count pass confirms it)
        int nt_first = LS_RND_MAXTILES / NW * NW;
        {
            DevBuf tcold, tpre;
            CSRK_TRY(tcold.alloc((size_t)(ls->n_tiles + 1) * 4));
            CSRK_TRY(tpre.alloc((size_t)(ls->n_tiles + 2) * 8));
            ls_tile_cold_kernel<<<(unsigned)ceil_div(ls->n_tiles * WAVE, 256), 256, 0, s>>>(ls->idx.as<uint32_t>(), ls->n_tiles,
                                                                                          tcold.as<int32_t>());
            CSRK_LAUNCH_CHECK();
            CSRK_TRY(exclusive_scan_i32_to_i64(tcold.as<int32_t>(), tpre.as<int64_t>(), ls->n_tiles, s));
            std::vector<int64_t> pre((size_t)ls->n_tiles + 1);
            CSRK_TRY(stage_d2h(pre.data(), tpre.p, pre.size() * 8, s));
            CSRK_HIP(hipStreamSynchronize(s));
            for (; nt_first > NW; nt_first -= NW) {
                make_rounds(nt_first);
                int64_t mx = 0;
                for (int64_t r = 0; r < nround_ls; r++)
                    mx = std::max(mx, pre[(size_t)h_rt0[(size_t)r + 1]] - pre[(size_t)h_rt0[(size_t)r]]);
                if (mx <= LS_RND_CAP) break;
            }
        }
        for (int nt = nt_first; nt >= NW;) {
            CSRK_TRY(count_pass(nt, &max_round));
            if (max_round <= LS_RND_CAP) {
                fits = true;
                break;
            }
            // the fullest round scales with the round's size: jump to the size that would just fit, then step down
            int next = (int)((double)nt * LS_RND_CAP / (double)max_round) / NW * NW;
            nt = next < nt - NW ? next : nt - NW;
        }
        if (!fits) return CSRK_OK;
    }
    tr.lap("  cold: round size (count passes)");
    if (nround > INT32_MAX) return CSRK_OK;
    CSRK_HIP(hipMemsetAsync(cnt.as<int32_t>() + nround_ls * nblk, 0, (size_t)(nblk + 1) * 4, s));
    ls_pack_count_kernel<<<(unsigned)ceil_div(n_hot, 256), 256, 0, s>>>(hot_cols, n_hot, (int32_t)W,
                                                                       cnt.as<int32_t>() + nround_ls * nblk, offp.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_HIP(hipMemsetAsync(cntT.p, 0, (size_t)(nb + 1) * 4, s));
    ls_cold_transpose_kernel<<<(unsigned)ceil_div(nb, 256), 256, 0, s>>>(cnt.as<int32_t>(), (int32_t)nround, (int32_t)nblk,
                                                                       cntT.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    const int64_t n_all = n_cold + n_hot;
    // worth a pass of its own only when the cold columns cannot live in L2 anyway and there are enough of them
    if (n_cold < 1 || n_all >= (int64_t)LS_PAD || (!(env && env[0] == '1') && n_cold * 16 < n_words)) return CSRK_OK;
    CSRK_TRY(exclusive_scan_i32(cnt.as<int32_t>(), cnt.as<int32_t>(), nb + 1, s));
    CSRK_TRY(exclusive_scan_i32(cntT.as<int32_t>(), cntT.as<int32_t>(), nb + 1, s));
    CSRK_TRY(ls->xg.alloc((size_t)(n_all + WAVE) * 8));      // (+ padding: the stream kernel's last pair of an odd count)
    CSRK_TRY(ls->a_col.alloc((size_t)n_all * 2));
    CSRK_TRY(ls->a_dst.alloc((size_t)n_all * 4));
    CSRK_TRY(ls->round_start.alloc((size_t)(nround_ls + 1) * 4));
    ls_round_start_kernel<<<(unsigned)ceil_div(nround_ls + 1, 256), 256, 0, s>>>(cnt.as<int32_t>(), (int32_t)nround_ls, (int32_t)nblk,
                                                                               ls->round_start.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(ls->round_tile0.alloc(h_rt0.size() * 4));
    CSRK_TRY(ls->wg_round0.alloc(h_wr0.size() * 4));
    CSRK_TRY(stage_h2d(ls->round_tile0.p, h_rt0.data(), h_rt0.size() * 4, s));
    CSRK_TRY(stage_h2d(ls->wg_round0.p, h_wr0.data(), h_wr0.size() * 4, s));
    ls_cold_place_kernel<<<gw, 256, 0, s>>>(ls->idx.as<uint32_t>(), n_words, (int32_t)nround, (int32_t)nblk, (int32_t)W,
                                           tile_round.as<int32_t>(), 1, cnt.as<int32_t>(), cntT.as<int32_t>(), off.as<int32_t>(),
                                           ls->a_col.as<uint16_t>(), ls->a_dst.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    ls_pack_place_kernel<<<(unsigned)ceil_div(n_hot, 256), 256, 0, s>>>(hot_cols, n_hot, (int32_t)W, (int32_t)nround,
                                                                       cntT.as<int32_t>(), offp.as<int32_t>(), (int32_t)n_cold,
                                                                       ls->a_col.as<uint16_t>(), ls->a_dst.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(ls->blk_start.alloc((size_t)(nblk + 1) * 4));
    ls_stage_starts_kernel<<<(unsigned)ceil_div(nblk + 1, 256), 256, 0, s>>>(cntT.as<int32_t>(), (int32_t)nround, (int32_t)nblk,
                                                                            ls->blk_start.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    ls->n_stage_blk = (int32_t)nblk;
    ls->stage_w = (int32_t)W;
    CSRK_TRY(spmv_kernel_attributes());
    // every column field is now a pack slot (< n_hot) or an offset into a round's staged values (< LS_RND_CAP): 3-byte words
    DevBuf idx24;
    const bool narrow = CSRK_LS_IDX24 && (int64_t)n_hot < (int64_t)LS24_COL_MASK && (int64_t)LS_RND_CAP < (int64_t)LS24_COL_MASK;
    if (narrow) {
        CSRK_TRY(idx24.alloc((size_t)ls->n_tiles * LS24_TILE_BYTES));
        ls_idx24_kernel<<<(unsigned)ceil_div(ls->n_tiles * WAVE, 256), 256, 0, s>>>(ls->idx.as<uint32_t>(), ls->n_tiles,
                                                                                   idx24.as<unsigned char>());
        CSRK_LAUNCH_CHECK();
    }
    CSRK_HIP(hipStreamSynchronize(s));      // the temporaries are freed on return
    if (narrow) {
        ls->idx.release();
        ls->idx.p = idx24.take();
        ls->idx.bytes = (size_t)ls->n_tiles * LS24_TILE_BYTES;
        ls->idx24 = true;
    }
    ls->n_cold = n_cold;
    ls->stage_tiles = stage_tiles;
    tr.lap("  cold: place, 3-byte words");
    return CSRK_OK;
}

// Copy the rows of the row-major path (the light view, or the whole matrix when nothing was cut out) into
// the light stream.  Built with the lazy plan.  Skipped (the tile kernel stays in charge) when ncols needs
// the two flag bits, when the copy does not fit in device memory, or with CSRK_SPMV_STREAM=0.
template <class P, int VT>
static int build_light_stream(Matrix *m, SpmvPlan *p, hipStream_t s)
{
    p->ls.on = false;
    const char *env = getenv("CSRK_SPMV_STREAM");
    if (env && env[0] == '0') return CSRK_OK;
    const int64_t n_view = p->n_heavy ? p->nnz_light : m->nnz;
    if (m->nrows == 0 || n_view < 1 || (int64_t)m->ncols > (int64_t)LS_COL_MASK) return CSRK_OK;
    // Without long rows cut out and without a popularity skew worth packing, the gathers are either local
    // (banded: the tile kernel's entry-per-lane order coalesces them better: 0.526 vs 0.638 ms measured) or
    // all equally cold (uniform random columns: both kernels run at the 128-B-per-gather fabric rate), and
    // the stream's copy of the matrix buys nothing.
    if (!p->n_heavy && !p->n_hot && !(env && env[0] == '1')) return CSRK_OK;
    const P *rp = (const P *)m->d_rowptrs;
    const P *rpv = p->n_heavy ? p->rp_light.as<P>() : rp;
    const int32_t *slot_map = p->n_hot ? p->hot_slot.as<int32_t>() : (const int32_t *)nullptr;
    const int64_t n_tiles = ceil_div(n_view, ACC_TILE);
    size_t mfree = 0, mtotal = 0;
    CSRK_HIP(hipMemGetInfo(&mfree, &mtotal));
    if ((size_t)n_tiles * ACC_TILE * 12 + ((size_t)m->nrows + n_tiles) * 8 + (64u << 20) > mfree && !(env && env[0] == '1'))
        return CSRK_OK;
    // Dense rows: when few rows of the view are empty (rows without entries, rows cut out for the tiers: 7 % on the
    // headline matrix) each of them gets ONE padding entry, so every row has a run, run k IS row k, and the stream kernel
    // neither loads row ids (4 loads per tile: 12 of its 170 us) nor clears gaps.  rpd = the view's pointers with empty
    // rows widened to one slot.  Not when the padding would add more than an eighth to the stream, nor past P's range.
    DevBuf rpd_buf;
    const P *rp_dense = nullptr;
    int64_t n_view_d = n_view, n_tiles_d = n_tiles;
    {
        DevBuf nz;
        CSRK_TRY(nz.alloc((size_t)(m->nrows + 2) * 4));
        ls_rowflag_kernel<P><<<(unsigned)ceil_div((int64_t)m->nrows + 1, 256), 256, 0, s>>>(rpv, m->nrows, nz.as<int32_t>());
        CSRK_LAUNCH_CHECK();
        CSRK_TRY(exclusive_scan_i32(nz.as<int32_t>(), nz.as<int32_t>(), m->nrows, s));
        int32_t n_nonempty = 0;
        CSRK_TRY(stage_d2h(&n_nonempty, nz.as<int32_t>() + m->nrows, 4, s));
        CSRK_HIP(hipStreamSynchronize(s));
        const int64_t n_pad = (int64_t)m->nrows - n_nonempty;
        const bool fits = sizeof(P) == 8 || n_view + n_pad <= (int64_t)INT32_MAX;
        if (fits && n_pad * 8 <= n_view) {
            CSRK_TRY(rpd_buf.alloc((size_t)(m->nrows + 1) * sizeof(P)));
            ls_dense_ptr_kernel<P><<<(unsigned)ceil_div((int64_t)m->nrows + 1, 256), 256, 0, s>>>(rpv, nz.as<int32_t>(), m->nrows,
                                                                                                rpd_buf.as<P>());
            CSRK_LAUNCH_CHECK();
            rp_dense = rpd_buf.as<P>();
            n_view_d = n_view + n_pad;
            n_tiles_d = ceil_div(n_view_d, ACC_TILE);
        }
        CSRK_HIP(hipStreamSynchronize(s));      // nz is released here
    }
    CSRK_TRY((build_stream<P, VT>(m, &p->ls, rp, rp_dense ? rp_dense : rpv, m->nrows, m->d_colinds, m->d_values, n_view_d, n_tiles_d,
                                  m->nrows, slot_map, s, rp_dense ? rpv : (const P *)nullptr)));
    if (p->ls.on && p->n_hot) {
        CSRK_TRY(build_cold_stage(m, &p->ls, p->hot_cols.as<int32_t>(), p->n_hot, s));
        // (round-in-LDS form: the round's staged values take the place of the hot window's tail)
        if (p->ls.n_cold && p->ls.round_start.p && p->n_hot_lds > LS_RND_HOT) p->n_hot_lds = LS_RND_HOT;
    }
    return CSRK_OK;
}

template <class P>
static int build_plan(Matrix *m, SpmvPlan *p, hipStream_t s, bool allow_split)
{
    const P *rp = (const P *)m->d_rowptrs;
    if (p->algo == CSRK_SPMV_MERGE) {
        p->tile_items = MERGE_ITEMS;
        p->nnz_light = m->nnz;
        p->split_considered = allow_split;
        PlanTrace tr;
        if (allow_split) CSRK_TRY(build_heavy_split<P>(m, p, s));
        tr.lap("heavy split");
        const P *rp_path = p->n_heavy ? p->rp_light.as<P>() : rp;
        int64_t total = (int64_t)m->nrows + p->nnz_light;
        p->n_tiles = ceil_div(total, MERGE_ITEMS);
        CSRK_TRY(p->tile_row.alloc((size_t)(p->n_tiles + 1) * 4));
        CSRK_TRY(p->carry_row.alloc((size_t)p->n_tiles * 4));
        CSRK_TRY(p->carry_val.alloc((size_t)p->n_tiles * 8));
        int64_t nthr = p->n_tiles + 1;
        merge_plan_kernel<P><<<(unsigned)ceil_div(nthr, 256), 256, 0, s>>>(rp_path, m->nrows, p->nnz_light, MERGE_ITEMS,
                                                                          p->n_tiles, p->tile_row.as<int32_t>());
        CSRK_LAUNCH_CHECK();
        if (p->n_heavy) {
            CSRK_TRY(p->tile_cut.alloc((size_t)(p->n_tiles + 1) * 4));
            heavy_tilecut_kernel<<<(unsigned)ceil_div(nthr, 256), 256, 0, s>>>(
                p->tile_row.as<int32_t>(), p->n_tiles, MERGE_ITEMS, total, p->cut_pos.as<int64_t>(), p->n_heavy,
                p->tile_cut.as<int32_t>());
            CSRK_LAUNCH_CHECK();
        }
        tr.lap("merge-path tables");
        if (allow_split) {
            CSRK_TRY(build_hot_cache<P>(m, p, s));
            tr.lap("hot-column census + pack");
            CSRK_TRY(build_tiers<P>(m, p, s));
            tr.lap("tiers");
            if (m->val_type == CSRK_VAL_F64) CSRK_TRY((build_light_stream<P, CSRK_VAL_F64>(m, p, s)));
            else if (m->val_type == CSRK_VAL_F32) CSRK_TRY((build_light_stream<P, CSRK_VAL_F32>(m, p, s)));
            else CSRK_TRY((build_light_stream<P, CSRK_VAL_NONE>(m, p, s)));
            tr.lap("light stream + cold staging");
            if (p->n_hot && !p->ls.on) {        // no stream (no memory for it, CSRK_SPMV_STREAM=0): the tile kernel reads x itself
                p->n_hot = 0;
                p->hot_cols.release();
                p->xh.release();
            }
            p->hot_slot.release();
        }
    } else if (p->algo == CSRK_SPMV_VECTOR) {
        CSRK_TRY(p->seg_off.alloc((size_t)(m->nrows + 1) * 8));
        if (m->nrows > 0) {
            vec_count_kernel<P><<<(unsigned)ceil_div(m->nrows, 256), 256, 0, s>>>(rp, m->nrows, p->seg_off.as<int64_t>());
            CSRK_LAUNCH_CHECK();
        }
        CSRK_TRY(exclusive_scan_i64(p->seg_off.as<int64_t>(), p->seg_off.as<int64_t>(), m->nrows, s));
        int64_t n_segs = 0;
        CSRK_TRY(stage_d2h(&n_segs, p->seg_off.as<int64_t>() + m->nrows, 8, s));
        CSRK_HIP(hipStreamSynchronize(s));
        p->n_segs = n_segs;
        CSRK_TRY(p->seg_row.alloc((size_t)n_segs * 4));
        CSRK_TRY(p->seg_part.alloc((size_t)n_segs * 8));
        if (m->nrows > 0) {
            vec_fill_kernel<<<(unsigned)ceil_div(m->nrows, 256), 256, 0, s>>>(p->seg_off.as<int64_t>(), m->nrows,
                                                                             p->seg_row.as<int32_t>());
            CSRK_LAUNCH_CHECK();
        }
    }
    return CSRK_OK;
}

int build_spmv_plan(Matrix *m, SpmvPlan *p, bool allow_split)
{
    return m->ptr64 ? build_plan<int64_t>(m, p, nullptr, allow_split) : build_plan<int32_t>(m, p, nullptr, allow_split);
}

}  // namespace csrk
