// SpMV (y = A x) for libcsrk on gfx950.  Replaces the reference's mult_vec
// (csr/kernels/numba/__init__.py:55-67: one sequential pass over nnz with a moving row
// cursor) and lk_mkl_spmv (csr/kernels/mkl/mkl_ops.c:116-126).
//
// Algorithms, selected per handle (csrk_set_spmv_algo; AUTO = merge):
//
//  merge   The production path.  FIRST call on a handle: the merge-path tile kernel on the CSR arrays as they
//          are (no plan beyond the tile coordinates) -- the path over (row ends, nnz) is cut into tiles of a
//          fixed number of ITEMS = rows + nnz, so a tile's work is bounded whatever the row-length
//          distribution; one 256-thread workgroup per tile stages products in LDS, one lane per row sums its
//          row in storage order, rows cut by a tile boundary leave a carry that a tiny kernel adds in tile
//          order.  From the SECOND call on, a plan built once per handle (see "plan" below and DESIGN.md
//          section 4) splits the rows three ways, each with a private, pre-decoded stream:
//            tier 0  the (up to 15360) longest rows: column-block-major, x window AND one accumulator per
//                    row in LDS, 10 B per entry: f64 value + 16-bit (column, row step) word
//                    (spmv_acc_kernel, "long rows, accumulator form");
//            tier 1  rows of 128 .. tier-0 threshold: (column block, row) pairs over 1 MiB x windows kept in
//                    one XCD's L2 (spmv_panel_kernel, "long rows, panel form");
//            light   everything else: one wavefront per 512-entry tile, hot columns from LDS / a packed,
//                    L2-resident copy of x, the x values of the unpacked columns copied beforehand into the
//                    order the stream reads them (ls_stage_kernel, "cold staging"), short rows bit-identical
//                    to the sequential loop (spmv_lstream_kernel, "short rows: the light stream").
//          No float atomics decide an order anywhere: results are bitwise reproducible run to run.
//          csrk_spmv_device_part runs the light part and the tiers' part separately (multi-GPU exchange
//          hidden behind the tiers: csr_amd/dist.py).
//  vector  One wavefront per row segment (rows longer than 4096 entries are split);
//          coalesced 64-lane strides over colinds/values, __shfl_down reduction, ordered
//          partial combine.  The classic CSR-vector shape; kept as an A/B baseline.
//  scalar  One lane per row.  Kept as an A/B baseline.
//
// All kernels accumulate in float64 whatever the storage dtype, like the reference
// (float32 values are widened on load; structure-only matrices multiply by 1.0,
// csr/csr.py:254-262).
#include "common.h"
#include "wave.h"

#include <algorithm>
#include <ctime>
#include <vector>

namespace csrk {

// ---- value loads ------------------------------------------------------------------------
template <int VT> struct ValLoad;
template <> struct ValLoad<CSRK_VAL_F64> {
    static __device__ __forceinline__ double at(const void *v, int64_t k) { return ((const double *)v)[k]; }
};
template <> struct ValLoad<CSRK_VAL_F32> {
    static __device__ __forceinline__ double at(const void *v, int64_t k) { return (double)((const float *)v)[k]; }
};
template <> struct ValLoad<CSRK_VAL_NONE> {
    static __device__ __forceinline__ double at(const void *, int64_t) { return 1.0; }
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return v;
}

// Sum p[s..e) strictly left to right (the reference's order), four LDS loads in flight at a time: a
// plain `for (k) acc += p[k]` is a load -> wait -> add chain of ~100 cycles per entry.
__device__ __forceinline__ double ordered_sum(const double *p, int s, int e)
{
    double acc = 0.0;
    int k = s;
    for (; k + 4 <= e; k += 4) {
        const double a = p[k], b = p[k + 1], c = p[k + 2], d = p[k + 3];
        acc += a;
        acc += b;
        acc += c;
        acc += d;
    }
    for (; k < e; k++) acc += p[k];
    return acc;
}

// ---- plan ---------------------------------------------------------------------------------
// One column-blocked panel: the entries of a set of long rows re-sorted block-major into a matrix
// M' whose rows are (column block, long row) pairs; see "long rows, panel form" below.
struct Panel {
    bool on = false, p64 = false;
    int32_t cb = 0, nb = 0, nrow = 0;            // block width (columns), blocks, long rows in this tier
    int64_t rows = 0, tiles = 0, nnz = 0, groups = 0;
    DevBuf row_list;                             // int32[nrow]: original row ids, ascending
    DevBuf rp, ci, vs, tile, group, carry_row, carry_val, y;
    // the tiles' carries, listed per long row (static: a tile's carry belongs to the pair its last row end falls in):
    // carries of long row h = carry_val[cidx[crp[h] .. crp[h + 1])], in tile order -- added by the tier's ordered reduce
    DevBuf crp, cidx;
};

// one segment of an accumulator-form workgroup's tile range (see "long rows, accumulator form")
struct AccSeg {
    int64_t tile0;    // first tile of the segment (logical: block-major order)
    int64_t ptile0;   // ... and where it is stored; the segment's tile t is stored at ptile0 + t * n_wg
    int32_t ntiles;   // <= ACC_SEG_TILES, all in one column block
    int32_t blk;
};

// Tier 0 in accumulator form: one group of <= ACC_MAXROWS heavy rows (see "long rows, accumulator form")
struct AccPanel {
    int32_t nrow = 0, nb = 0, n_wg = 0;
    int64_t tiles = 0, nnz = 0, n_segs = 0;
    size_t lds = 0;
    DevBuf row_list, vals, idx, tile_row0, segs, wg_seg, partial;      // idx: 16-bit words (column, row step)
};

// A light stream (see "short rows: the light stream"): a private tiled copy of a set of rows ("runs") plus
// the tables the one-wavefront-per-tile kernel needs.
struct LightStream {
    bool on = false;
    int64_t n_tiles = 0;
    int32_t n_runs = 0, n_out = 0;      // non-empty runs; length of the output vector (rows, or pairs)
    unsigned grid = 0;
    DevBuf vals, idx, rowids, tile_base, carry_row, carry_val;
    // dense rows (build_light_stream): EVERY row of the view has a run -- a row without entries holds one padding entry --
    // so run k is row k: no row-id table (`rowids` stays empty), no gaps to clear
    bool dense = false;
    // cold staging (build_cold_stage): the x values of the stream's unpacked columns, copied per call into the
    // order the stream reads them
    int64_t n_cold = 0;
    int32_t n_stage_blk = 0, stage_w = 0;
    DevBuf xg, a_col, a_dst, blk_start;
    // round-in-LDS staging (build_cold_stage, LS_RND): workgroup b walks rounds wg_round0[b] .. wg_round0[b + 1]; round r =
    // tiles round_tile0[r] .. round_tile0[r + 1] (at most `stage_tiles`), whose staged values xg[round_start[r] ..
    // round_start[r + 1]) the workgroup copies into LDS; a cold entry's index word holds its offset inside that range
    int32_t stage_tiles = 0;
    DevBuf round_start, round_tile0, wg_round0;
};

struct SpmvPlan {
    int algo = CSRK_SPMV_MERGE;
    // merge
    int tile_items = 0;
    int64_t n_tiles = 0;
    DevBuf tile_row;    // int32[n_tiles + 1]: rows completed before each tile boundary
    DevBuf carry_row;   // int32[n_tiles]
    DevBuf carry_val;   // double[n_tiles]
    // merge, long-row split: rows >= the cut threshold are taken out of the merge path (light view)
    // and served by one or two column-blocked panels
    bool split_considered = false; // false: built without looking at the long-row split (first call)
    int32_t n_heavy = 0;          // rows cut out
    int32_t heavy_min = 0;        // tier 0 holds the cut rows with at least this many entries
    int64_t nnz_light = 0;
    DevBuf rp_light;    // P[nrows + 1]: row pointers with the cut rows collapsed to length 0
    DevBuf cut_pos;     // int64[n_heavy]: light-index position of each cut row
    DevBuf cut_cum;     // int64[n_heavy + 1]: cut entries before each cut row (shift table)
    DevBuf tile_cut;    // int32[n_tiles + 1]: cuts at or before each tile start
    DevBuf heavy_row;   // int32[n_heavy]
    // merge, hot-column pack: the HOT_SLOTS most referenced columns are renumbered to -1 - slot in a
    // copy of colinds; their x values are packed into xh before every tile-kernel launch
    int32_t n_hot = 0;            // 0: no pack
    int32_t n_hot_lds = 0;        // slots [0, n_hot_lds) hold the most referenced columns (kept in LDS by the light stream)
    int32_t hot_slots = 0;
    double hot_cover = 0.0;       // sampled fraction of the tile kernel's entries on packed columns
    DevBuf hot_slot;    // int32[ncols]: slot of a packed column, -1 otherwise (kept until the light stream is built)
    DevBuf hot_cols;    // int32[n_hot]: column of each slot
    DevBuf xh;          // double[n_hot]
    Panel tier1;        // mid rows: (column block, row) pairs over 262144-column blocks, the x window kept in L2 by
                        // block-major, XCD-aware scheduling
    LightStream ls;                       // the rows that stay on the row-major path
    std::vector<int32_t> t1_rows;         // tier-1 rows (ascending) and their entries: build_tiers
    int64_t t1_nnz = 0;
    std::vector<AccPanel *> acc;          // tier 0: accumulator form, groups of <= ACC_MAXROWS rows
    std::vector<int32_t> t0_rows;         // tier-0 rows (ascending) and their lengths
    std::vector<int64_t> t0_lens;
    // vector
    int64_t n_segs = 0;
    DevBuf seg_off;     // P-agnostic: int64[nrows + 1] segment offsets per row
    DevBuf seg_row;     // int32[n_segs]
    DevBuf seg_part;    // double[n_segs]
    // kernel timing (csrk_spmv_profile_begin/end)
    std::vector<hipEvent_t> ev;   // start/stop pairs
    std::vector<int> ev_chan;     // channel of each pair: 0 = tile/segment/row kernel, 1/2 = panel tier 0/1
    int ev_used = 0;
    bool profiling = false;
    int prof_every = 1;           // profile every n-th launch group only (an event pair costs ~3 us on the stream)
    int prof_calls = 0;
    int prof_mask = 0xf;          // channels that get event pairs (csrk_spmv_profile_channels)
    bool prof_this = false;       // the launch group in progress is being timed
    ~SpmvPlan()
    {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        for (AccPanel *a : acc) delete a;
    }
};

struct KernelTimer {   // records an event pair around one launch when the plan is profiling
    SpmvPlan *p;
    hipStream_t s;
    int slot = -1;
    KernelTimer(SpmvPlan *p_, hipStream_t s_, int chan = 0) : p(p_), s(s_)
    {
        if (p->profiling && p->prof_this && ((p->prof_mask >> chan) & 1) && p->ev_used + 2 <= (int)p->ev.size()) {
            slot = p->ev_used;
            p->ev_used += 2;
            p->ev_chan[slot / 2] = chan;
            (void)hipEventRecord(p->ev[slot], s);
        }
    }
    void stop()
    {
        if (slot >= 0) (void)hipEventRecord(p->ev[slot + 1], s);
    }
};

void free_spmv_plan(SpmvPlan *p) { delete p; }

// device memory the plan holds (private streams, tables, scratch)
int64_t spmv_plan_bytes(const SpmvPlan *p)
{
    int64_t plan_bytes = 0;
    {
        auto add = [&](const DevBuf &b) { plan_bytes += (int64_t)b.bytes; };
        for (const DevBuf *b : {&p->tile_row, &p->carry_row, &p->carry_val, &p->rp_light, &p->cut_pos, &p->cut_cum, &p->tile_cut,
                                &p->heavy_row, &p->hot_slot, &p->hot_cols, &p->xh, &p->seg_off, &p->seg_row, &p->seg_part})
            add(*b);
        {
            const Panel &t = p->tier1;
            for (const DevBuf *b : {&t.row_list, &t.rp, &t.ci, &t.vs, &t.tile, &t.group, &t.carry_row, &t.carry_val, &t.y, &t.crp, &t.cidx})
                add(*b);
        }
        for (const AccPanel *ap : p->acc)
            for (const DevBuf *b : {&ap->row_list, &ap->vals, &ap->idx, &ap->tile_row0, &ap->segs, &ap->wg_seg, &ap->partial}) add(*b);
        {
            const LightStream *l = &p->ls;
            for (const DevBuf *b : {&l->vals, &l->idx, &l->rowids, &l->tile_base, &l->carry_row, &l->carry_val, &l->xg, &l->a_col,
                                    &l->a_dst, &l->blk_start, &l->round_start, &l->round_tile0, &l->wg_round0})
                add(*b);
        }
    }
    return plan_bytes;
}


constexpr int MERGE_THREADS = 256;
constexpr int MERGE_IPT = 8;
constexpr int MERGE_ITEMS = MERGE_THREADS * MERGE_IPT;   // 2048 path items per tile
constexpr int MERGE_LONG = 64;                           // rows this long get a whole wave
constexpr int MERGE_MAXLONG = MERGE_ITEMS / MERGE_LONG + 2;

// tile_row[t] = number of row ends consumed before merge-path diagonal d = min(t*ITEMS, nrows+nnz).
// Row end r (= rp[r+1]) is consumed once all its nnz are: it lies before diagonal d iff
// rp[r+1] + r + 1 <= d.
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
typedef int32_t i32x2_t __attribute__((ext_vector_type(2)));
typedef double f64x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef i32x2_t I32x2 __attribute__((aligned(4)));
typedef f64x2_t F64x2 __attribute__((aligned(4)));
typedef f32x2_t F32x2 __attribute__((aligned(4)));

template <int VT>
__device__ __forceinline__ void load_val_pair(const void *vs, int64_t k, bool two, double &v0, double &v1)
{
    if (VT == CSRK_VAL_F64) {
        const double *p = (const double *)vs + k;
        if (two) {
            f64x2_t t = __builtin_nontemporal_load((const F64x2 *)p);
            v0 = t.x;
            v1 = t.y;
        } else {
            v0 = *p;
            v1 = 0.0;
        }
    } else if (VT == CSRK_VAL_F32) {
        const float *p = (const float *)vs + k;
        if (two) {
            f32x2_t t = *(const F32x2 *)p;
            v0 = t.x;
            v1 = t.y;
        } else {
            v0 = *p;
            v1 = 0.0;
        }
    } else {
        v0 = 1.0;
        v1 = two ? 1.0 : 0.0;
    }
}

// Branch-free tile loads.  hipcc turns `if (k < nn) v = p[k];` inside an unrolled loop into a branch
// around each load followed by s_waitcnt vmcnt(0), which serialises the loads (one memory round trip
// per element).  These helpers always load -- from an address clamped into the array -- and mask
// afterwards, so all of a lane's loads are in flight together.  `last_pair` = n_total - 2 (the last
// index at which a 2-entry load is in bounds; requires n_total >= 2).  A lane whose pair would start
// at the array's final entry loads the pair one entry earlier and takes its second half.
template <int VT>
__device__ __forceinline__ void load_pair_clamped(const int32_t *__restrict__ ci, const void *__restrict__ vs,
                                                  int64_t a, int64_t tile_first, int nn, int64_t last_pair,
                                                  int32_t &c0, int32_t &c1, double &v0, double &v1)
{
    // Lanes past the tile's end re-read the tile's own last pair (same cache lines as their
    // neighbours: no extra traffic, and their x gather is the neighbours' address); the tile's final
    // odd entry is taken from the second half of the pair that starts one entry earlier.
    int64_t q = tile_first + (nn >= 2 ? nn - 2 : 0);
    q = a < q ? a : q;
    q = q < last_pair ? q : last_pair;
    q = q > 0 ? q : 0;
    const bool second = q != a;
    const i32x2_t cc = __builtin_nontemporal_load((const I32x2 *)(ci + q));
    c0 = second ? cc.y : cc.x;
    c1 = cc.y;
    if (VT == CSRK_VAL_F64) {
        const f64x2_t t = __builtin_nontemporal_load((const F64x2 *)((const double *)vs + q));
        v0 = second ? t.y : t.x;
        v1 = t.y;
    } else if (VT == CSRK_VAL_F32) {
        const f32x2_t t = *(const F32x2 *)((const float *)vs + q);
        v0 = second ? t.y : t.x;
        v1 = t.y;
    } else {
        v0 = v1 = 1.0;
    }
}

constexpr int MERGE_PAIRS = MERGE_IPT / 2;
#ifndef MERGE_GATHER_PAIRS
#define MERGE_GATHER_PAIRS 4
#endif

// HEAVY: the path runs over the light view (rp = rp_light, nnz = nnz_light); a light entry index
// jl maps to the actual entry jl + cut_cum[#cuts with cut_pos <= jl].
template <class P, int VT, bool HEAVY>
__global__ __launch_bounds__(MERGE_THREADS) void spmv_merge_kernel(
    const P *__restrict__ rp, const int32_t *__restrict__ ci, const void *__restrict__ vs,
    const double *__restrict__ x, double *__restrict__ y, const int32_t *__restrict__ tile_row,
    int32_t nrows, int64_t nnz, int32_t *__restrict__ carry_row, double *__restrict__ carry_val,
    const int32_t *__restrict__ tile_cut, const int64_t *__restrict__ cut_pos,
    const int64_t *__restrict__ cut_cum, int64_t nnz_total)
{
    // One LDS buffer: nn products (8 B each) followed by nr + 1 tile-relative row ends (4 B each);
    // nn + nr <= MERGE_ITEMS, so MERGE_ITEMS * 8 + 8 bytes always suffice (16.4 KB -> 8 tiles per CU).
    __shared__ double s_buf[MERGE_ITEMS + 1];
    __shared__ int32_t s_long[MERGE_MAXLONG];
    __shared__ int32_t s_nlong;
    __shared__ double s_wpart[MERGE_THREADS / WAVE];

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1), wv = tid / WAVE;
    const int64_t t = blockIdx.x;
    const int32_t i0 = tile_row[t], i1 = tile_row[t + 1];
    const int64_t total = (int64_t)nrows + nnz;
    const int64_t d0 = t * MERGE_ITEMS;
    const int64_t d1 = d0 + MERGE_ITEMS < total ? d0 + MERGE_ITEMS : total;
    const int64_t j0 = d0 - i0;
    const int nn = (int)((d1 - i1) - j0);   // nnz in this tile
    const int nr = i1 - i0;                 // rows completed in this tile

    // light index -> actual entry index
    int64_t ja = j0;                        // actual index of the tile's first entry
    int32_t cb = 0, ce = 0;                 // cuts strictly inside the tile: [cb, ce)
    if (HEAVY) {
        cb = tile_cut[t];
        ce = tile_cut[t + 1];
        ja = j0 + cut_cum[cb];
    }

    // phase 1a: products.  Lane owns the consecutive pair (2q, 2q+1), q = tid + u*THREADS; all
    // colind / value / row-end loads are issued before the dependent x gathers.
    int32_t c0[MERGE_PAIRS], c1[MERGE_PAIRS];
    double p0[MERGE_PAIRS], p1[MERGE_PAIRS];
    constexpr int RPT = MERGE_ITEMS / MERGE_THREADS;
    int32_t rv[RPT];
#pragma unroll
    for (int u = 0; u < RPT; u++) {
        const int r = tid + u * MERGE_THREADS;
        const int rc = r < nr ? r : (nr > 0 ? nr - 1 : 0);       // clamped: always a valid row pointer
        rv[u] = (int32_t)((int64_t)rp[i0 + rc + 1] - j0);
    }
    if (!HEAVY || cb == ce) {
#pragma unroll
        for (int u = 0; u < MERGE_PAIRS; u++) {
            const int k = 2 * (tid + u * MERGE_THREADS);
            load_pair_clamped<VT>(ci, vs, ja + k, ja, nn, nnz_total - 2, c0[u], c1[u], p0[u], p1[u]);
        }
    } else {
        // a heavy row was cut out somewhere inside this tile: per-entry shift (rare tiles)
#pragma unroll
        for (int u = 0; u < MERGE_PAIRS; u++) {
            const int k = 2 * (tid + u * MERGE_THREADS);
            c0[u] = c1[u] = 0;
            p0[u] = p1[u] = 0.0;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (k + h < nn) {
                    const int64_t jl = j0 + k + h;
                    int32_t lo = cb, hi = ce;
                    while (lo < hi) {
                        int32_t mid = (lo + hi) >> 1;
                        if (cut_pos[mid] <= jl)
                            lo = mid + 1;
                        else
                            hi = mid;
                    }
                    const int64_t a = jl + cut_cum[lo];
                    double v0, v1;
                    load_val_pair<VT>(vs, a, false, v0, v1);
                    if (h == 0) {
                        c0[u] = ci[a];
                        p0[u] = v0;
                    } else {
                        c1[u] = ci[a];
                        p1[u] = v0;
                    }
                }
            }
        }
    }
    // x gathers, MERGE_GATHER_PAIRS pairs (2 gathers each) in flight per lane at a time: with all 8
    // in flight a wavefront has 512 lines outstanding, twice the 256-line L1, and the popular x
    // entries that would hit in L1 are evicted between uses (measured: tools/probe notes in DESIGN.md).
    // Lanes past the tile's end hold a valid (neighbouring) entry; they are zeroed AFTER the
    // multiply -- 0 * x[c] would be NaN for a non-finite x[c] -- with a select, not a branch.
#pragma unroll
    for (int u = 0; u < MERGE_PAIRS; u++) {
        const int k = 2 * (tid + u * MERGE_THREADS);
        const double t0 = p0[u] * x[c0[u]], t1 = p1[u] * x[c1[u]];
        p0[u] = k < nn ? t0 : 0.0;
        p1[u] = k + 1 < nn ? t1 : 0.0;
        if ((u + 1) % MERGE_GATHER_PAIRS == 0 && u + 1 < MERGE_PAIRS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    if (nr == 0) {
        // The whole tile lies inside one row (a row longer than the tile): no LDS staging, each
        // lane sums its products, wavefront __shfl_down tree, four wave partials in fixed order.
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < MERGE_PAIRS; u++) acc += p0[u] + p1[u];
        acc = wave_sum(acc);
        if (lane == 0) s_wpart[wv] = acc;
        __syncthreads();
        if (tid == 0) {
            double tot = s_wpart[0];
#pragma unroll
            for (int w = 1; w < MERGE_THREADS / WAVE; w++) tot += s_wpart[w];
            carry_row[t] = i1 < nrows ? i1 : -1;
            carry_val[t] = tot;
        }
        return;
    }

    double *s_prod = s_buf;
    int32_t *s_rend = (int32_t *)(s_buf + nn);
    if (tid == 0) s_nlong = 0;
#pragma unroll
    for (int u = 0; u < MERGE_PAIRS; u++) {
        const int k = 2 * (tid + u * MERGE_THREADS);
        if (k < nn) s_prod[k] = p0[u];
        if (k + 1 < nn) s_prod[k + 1] = p1[u];
    }
    // phase 1b: tile-relative row ends; the tail segment (row i1, not completed here) ends at nn
#pragma unroll
    for (int u = 0; u < RPT; u++) {
        const int r = tid + u * MERGE_THREADS;
        if (r < nr) s_rend[r] = rv[u];
    }
    if (tid == 0) s_rend[nr] = nn;
    __syncthreads();

    // phase 2a: one lane per row, in storage order
    for (int r = tid; r <= nr; r += MERGE_THREADS) {
        int s = r ? s_rend[r - 1] : 0;
        int e = s_rend[r];
        if (e - s >= MERGE_LONG) {
            int q = atomicAdd(&s_nlong, 1);
            s_long[q] = r;
            continue;
        }
        double acc = ordered_sum(s_prod, s, e);
        if (r < nr) {
            y[i0 + r] = acc;
        } else {
            carry_row[t] = i1 < nrows ? i1 : -1;
            carry_val[t] = acc;
        }
    }
    __syncthreads();

    // phase 2b: long rows, one wavefront each
    const int nlong = s_nlong;
    for (int q = wv; q < nlong; q += MERGE_THREADS / WAVE) {
        int r = s_long[q];
        int s = r ? s_rend[r - 1] : 0;
        int e = s_rend[r];
        double acc = 0.0;
        for (int k = s + lane; k < e; k += WAVE) acc += s_prod[k];
        acc = wave_sum(acc);
        if (lane == 0) {
            if (r < nr) {
                y[i0 + r] = acc;
            } else {
                carry_row[t] = i1 < nrows ? i1 : -1;
                carry_val[t] = acc;
            }
        }
    }
}

// ---- hot-column cache: plan-time kernels ------------------------------------------------------------
// Column reference counts over the rows the tile kernel serves, from every `row_stride`-th row and at
// most 128 entries of it (a sample is enough to rank popularity); total[0] = entries counted.
// Popular columns collect millions of these increments: as global atomics they serialise (14 ms for the 4*10^7
// sampled entries of the headline matrix).  Each persistent workgroup therefore counts into an LDS hash table
// first (a column that finds a slot within HOT_PROBE probes stays there; the others go straight to memory) and
// flushes its <= HOT_TABLE distinct columns once at the end.
constexpr int HOT_TABLE = 8192, HOT_PROBE = 4;
template <class P>
__global__ __launch_bounds__(256) void hot_count_kernel(const P *__restrict__ rp, const P *__restrict__ rp_light,
                                                       const int32_t *__restrict__ ci, int32_t nrows, int64_t row_stride,
                                                       int32_t *__restrict__ cnt, unsigned long long *__restrict__ total)
{
    __shared__ int32_t s_key[HOT_TABLE];
    __shared__ int32_t s_cnt[HOT_TABLE];
    for (int k = threadIdx.x; k < HOT_TABLE; k += 256) {
        s_key[k] = -1;
        s_cnt[k] = 0;
    }
    __syncthreads();
    const int64_t n_sampled = (nrows + row_stride - 1) / row_stride;
    unsigned long long n = 0;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n_sampled; q += (int64_t)gridDim.x * 256) {
        const int64_t r = q * row_stride;
        if (rp_light && rp_light[r + 1] == rp_light[r]) continue;      // a row cut out to the tiers
        const int64_t s = rp[r];
        int64_t e = rp[r + 1];
        e = e - s > 128 ? s + 128 : e;
        for (int64_t k = s; k < e; k++) {
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
        n += (unsigned long long)(e - s);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < HOT_TABLE; k += 256)
        if (s_key[k] >= 0) atomicAdd(&cnt[s_key[k]], s_cnt[k]);
    for (int off = WAVE / 2; off; off >>= 1) n += __shfl_down(n, off, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0 && n) atomicAdd(total, n);
}

__global__ void hot_pack_kernel(const double *__restrict__ x, const int32_t *__restrict__ hot_cols, int32_t n_hot,
                                double *__restrict__ xh)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_hot) xh[i] = x[hot_cols[i]];
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
    // made this check 1.8 ms of the headline matrix's plan)
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

// ---- mid rows, pair form (tier 1) ---------------------------------------------------------------
// At plan time the rows' entries are re-sorted column-block-major into a panel matrix M' whose rows are (column block
// b, row h) pairs; values are widened to float64.  Per call the merge-tile algorithm runs over M' with tiles confined
// to one block; row sums of M' are the per-(block, row) partials y'[b][h], reduced over b in block order by the
// epilogue.  No float atomics: deterministic.  Rows of 128 .. tier-0 threshold entries: a (row, block) pair of 4096
// columns would hold < 1 entry, so blocks are 262144 columns (2 MiB of x) and x is gathered from global memory; tiles
// run block-major and block b is served only by workgroups with blockIdx % 8 == b % 8 (one XCD, so ONE L2 holds the
// window -- a speed assumption only), which turns Infinity-Cache gathers into L2 hits.  (The longest rows had this form
// too, with the x window in LDS, until the accumulator form below replaced it.)
constexpr int PANEL_CB1 = 262144;      // (2 MiB of x per block: half the (block, row) pairs of 131072 at the same kernel time, -9 us of partials)
#ifndef PANEL_T1
#define PANEL_T1 256
#endif

template <class P>
__global__ void panel_count_kernel(const P *__restrict__ rp, const int32_t *__restrict__ ci,
                                   const int32_t *__restrict__ heavy_row, int32_t n_heavy, int32_t n_blocks,
                                   int32_t cb, int64_t *__restrict__ cnt)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_heavy * n_blocks) return;
    const int32_t b = (int32_t)(i / n_heavy), c = (int32_t)(i % n_heavy);      // index = b * H + c
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
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t pairs = (int64_t)n_heavy * n_blocks;
    if (i > pairs) return;
    prp[i] = (PP)off[i];
    if (i == pairs) return;
    const int32_t b = (int32_t)(i / n_heavy), c = (int32_t)(i % n_heavy);
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

struct PanelTile {
    int64_t j0;      // first entry of the tile in M'
    int32_t i0, i1;  // rows of M' completed before the tile start / end
    int32_t nn;      // entries in the tile
    int32_t blk;     // column block
};

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
    tiles[t] = pt;
}

struct PanelGroup {
    int64_t t0;      // first tile
    int32_t nt;      // tiles handled by this workgroup (all in one column block)
    int32_t blk;
};
template <class PP, int PT>
__global__ __launch_bounds__(PT) void spmv_panel_kernel(
    const PP *__restrict__ prp, const int32_t *__restrict__ pci, const double *__restrict__ pvs,
    const double *__restrict__ x, int32_t ncols, double *__restrict__ yp, const PanelTile *__restrict__ tiles,
    const PanelGroup *__restrict__ groups, int64_t n_prows, int32_t *__restrict__ carry_row,
    double *__restrict__ carry_val, int64_t pnnz)
{
    __shared__ double s_buf[MERGE_ITEMS + 1];
    __shared__ int32_t s_long[MERGE_MAXLONG];
    __shared__ int32_t s_nlong;
    __shared__ double s_wpart[PT / WAVE];

    constexpr int PPAIRS = MERGE_ITEMS / PT / 2;      // consecutive pairs per lane
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1), wv = tid / WAVE;
    const PanelGroup grp = groups[blockIdx.x];
    if (grp.nt == 0) return;                      // padding group of an XCD stream
    // Software pipeline over the group's tiles: the entries (and row ends) of tile it+1 are loaded
    // into registers while tile it is reduced out of LDS, so one global-load latency is exposed per
    // group instead of two per tile.
    constexpr int RPT = MERGE_ITEMS / PT;     // row ends a lane may have to fetch
    int32_t c0[PPAIRS], c1[PPAIRS], rv[RPT];
    double p0[PPAIRS], p1[PPAIRS];
    PanelTile pt = tiles[grp.t0];

#define PANEL_LOAD_TILE(T)                                                                            \
    {                                                                                                 \
        const int64_t j0_ = (T).j0;                                                                   \
        const int nn_ = (T).nn, nr_ = (T).i1 - (T).i0;                                                \
        _Pragma("unroll") for (int u = 0; u < PPAIRS; u++)                                       \
        {                                                                                             \
            const int k = 2 * (tid + u * PT);                                              \
            load_pair_clamped<CSRK_VAL_F64>(pci, pvs, j0_ + k, j0_, nn_, pnnz - 2, c0[u], c1[u], p0[u], p1[u]); \
        }                                                                                             \
        _Pragma("unroll") for (int u = 0; u < RPT; u++)                                               \
        {                                                                                             \
            const int r = tid + u * PT;                                                    \
            const int rc = r < nr_ ? r : (nr_ > 0 ? nr_ - 1 : 0);                                     \
            rv[u] = (int32_t)((int64_t)prp[(T).i0 + rc + 1] - j0_);                                   \
        }                                                                                             \
    }

    PANEL_LOAD_TILE(pt);
    for (int it = 0; it < grp.nt; it++) {
        const int64_t t = grp.t0 + it;
        const int32_t i0 = pt.i0, i1 = pt.i1;
        const int nn = pt.nn;
        const int nr = i1 - i0;
        PanelTile nx = pt;
        const bool more = it + 1 < grp.nt;
        if (more) nx = tiles[t + 1];

        __syncthreads();      // previous tile's LDS reads finished
#pragma unroll
        for (int u = 0; u < PPAIRS; u++) {
            const int k = 2 * (tid + u * PT);
            const double t0 = p0[u] * x[c0[u]];
            const double t1 = p1[u] * x[c1[u]];
            p0[u] = k < nn ? t0 : 0.0;       // masked after the multiply: 0 * inf would be NaN
            p1[u] = k + 1 < nn ? t1 : 0.0;
        }

        if (nr == 0) {
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < PPAIRS; u++) acc += p0[u] + p1[u];
            if (more) PANEL_LOAD_TILE(nx);
            acc = wave_sum(acc);
            if (lane == 0) s_wpart[wv] = acc;
            __syncthreads();
            if (tid == 0) {
                double tot = s_wpart[0];
#pragma unroll
                for (int w = 1; w < PT / WAVE; w++) tot += s_wpart[w];
                carry_row[t] = i1 < n_prows ? i1 : -1;
                carry_val[t] = tot;
            }
            pt = nx;
            continue;         // the barrier at the top of the next pass orders the s_wpart reuse
        }

        double *s_prod = s_buf;
        int32_t *s_rend = (int32_t *)(s_buf + nn);
        if (tid == 0) s_nlong = 0;
#pragma unroll
        for (int u = 0; u < PPAIRS; u++) {
            const int k = 2 * (tid + u * PT);
            if (k < nn) s_prod[k] = p0[u];
            if (k + 1 < nn) s_prod[k + 1] = p1[u];
        }
#pragma unroll
        for (int u = 0; u < RPT; u++) {
            const int r = tid + u * PT;
            if (r < nr) s_rend[r] = rv[u];
        }
        if (tid == 0) s_rend[nr] = nn;
        if (more) PANEL_LOAD_TILE(nx);       // registers are free again: next tile's loads in flight
        __syncthreads();

        for (int r = tid; r <= nr; r += PT) {
            int s = r ? s_rend[r - 1] : 0;
            int e = s_rend[r];
            if (e - s >= MERGE_LONG) {
                int q = atomicAdd(&s_nlong, 1);
                s_long[q] = r;
                continue;
            }
            double acc = ordered_sum(s_prod, s, e);
            if (r < nr) {
                yp[i0 + r] = acc;
            } else {
                carry_row[t] = i1 < n_prows ? i1 : -1;
                carry_val[t] = acc;
            }
        }
        __syncthreads();
        const int nlong = s_nlong;
        for (int q = wv; q < nlong; q += PT / WAVE) {
            int r = s_long[q];
            int s = r ? s_rend[r - 1] : 0;
            int e = s_rend[r];
            double acc = 0.0;
            for (int k = s + lane; k < e; k += WAVE) acc += s_prod[k];
            acc = wave_sum(acc);
            if (lane == 0) {
                if (r < nr) {
                    yp[i0 + r] = acc;
                } else {
                    carry_row[t] = i1 < n_prows ? i1 : -1;
                    carry_val[t] = acc;
                }
            }
        }
        pt = nx;
    }
#undef PANEL_LOAD_TILE
}

__global__ void panel_blockends_kernel(const int64_t *__restrict__ off, int32_t n_heavy, int32_t n_blocks,
                                       int64_t *__restrict__ out)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b <= n_blocks) out[b] = off[(int64_t)b * n_heavy];
}

// ---- long rows, accumulator form (tier 0, default) ---------------------------------------------------
// The pair form above spends a quarter of the tier-0 traffic on bookkeeping: a row pointer and a partial
// per (column block, row) pair (avg. 7.8 entries), the partials re-read by the reduce, plus four
// workgroup barriers per 2048-item tile.  The heavy rows are FEW (thousands), so one accumulator per heavy
// row fits in LDS next to the x window: 8192 rows * 8 B = 64 KiB + 32 KiB.  The accumulator form is a pure
// stream:
//   * Heavy rows are taken in groups of <= ACC_MAXROWS.  A group's entries are stored column-block-major
//     (block = ACC_CB columns), inside a block by heavy row, as (float64 value, packed uint32
//     {column - block start : 13 bits, heavy-row index : 13 bits}) = 12 B per entry, nothing else.
//   * A block's entries are padded to whole TILES of 512 = 64 lanes x 8 entries; a tile is stored lane-
//     interleaved so that one wavefront reads it with 16-B-per-lane coalesced loads and every lane
//     receives 8 CONSECUTIVE entries (a run of one row is then mostly inside one lane).
//   * A persistent workgroup (one per CU, 16 wavefronts) owns a contiguous range of tiles, cut into
//     SEGMENTS (tiles of one column block, <= 256).  Per segment: the block's x window -> LDS; each
//     wavefront walks tiles: lane-local ordered sums per row, a segmented scan over the lanes (__shfl_up)
//     joins the runs that cross lanes, and the finished row sums are added to the LDS accumulators.
//   * Determinism: within a segment a row's run is owned by the tile it starts in; the leading run of a
//     tile (which may belong to the previous tile's last row) is parked in a per-tile head slot instead and
//     the heads are folded in, in tile order, by one wavefront after the segment's barrier.  So every
//     accumulator sees its addends in a fixed order whichever wavefront took which tile: results are
//     bitwise reproducible, although ds_add_f64 is used for the adds.
//   * At the end the workgroup stores its accumulators (H * 8 B) and acc_reduce_multi_kernel sums the
//     workgroups' partials in workgroup order into y.
// HBM traffic: 12 B per entry + one 32 KiB window per segment + n_wg * H * 8 B of partials (14 MB on the
// headline matrix) -- against 12 B + 20 B per pair + windows for the pair form.
constexpr int ACC_CB = 4096;
constexpr int ACC_K = 8;                      // consecutive entries per lane
constexpr int ACC_TILE = WAVE * ACC_K;        // 512
constexpr int ACC_MAXROWS = 15936;            // heavy rows per group: 124.5 KiB of accumulators + 32 KiB window + 3 KiB of head slots <= 160 KiB
constexpr int ACC_FLOOR = 128;                // tier 0 is never extended to rows shorter than this (512 before the 10-B stream: a rank of an 8-way split ran 0.129 ms, 0.119 with 128)
constexpr int ACC_SEG_TILES = 256;            // head slots per segment
constexpr int ACC_THREADS = 1024;
// index word of the accumulator stream, 16 bits: column - block start in the low 13 (ACC_CB = the zero slot of the
// window, for padding), and in the high 3 the STEP from the previous entry's heavy-row index to this one's (rows
// ascend inside a block; 0 = same row).  A tile's first entry has step 0 and its row in tile_row0[]; a step over 7
// is bridged by padding entries (0.0 * zero slot) of step 7.  10 B per entry instead of 12: the kernel runs at the
// fabric's rate, so bytes are its time (16-bit columns + a row id per run, fetched by a dependent load, had not paid).
constexpr int ACC_ROW_SHIFT = 13;
constexpr uint32_t ACC_COL_MASK = (1u << ACC_ROW_SHIFT) - 1;
constexpr int ACC_MAXSTEP = 7;

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// physical slot of logical entry e (0..511) of a tile: lane = e / 8, j = e % 8
__host__ __device__ __forceinline__ int acc_val_slot(int e)
{
    const int lane = e >> 3, j = e & 7;
    return (j >> 1) * (2 * WAVE) + lane * 2 + (j & 1);        // four 16-B loads per lane
}
__host__ __device__ __forceinline__ int acc_idx_slot(int e)
{
    const int lane = e >> 3, j = e & 7;
    return (j >> 2) * (4 * WAVE) + lane * 4 + (j & 3);        // two 16-B loads per lane
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
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
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
    int32_t lo = 0, hi = n_wg - 1;      // the workgroup with wg_t0[w] <= t < wg_t0[w + 1]
    while (lo < hi) {
        const int32_t mid = lo + ((hi - lo + 1) >> 1);
        if (wg_t0[mid] <= t)
            lo = mid;
        else
            hi = mid - 1;
    }
    return (t - wg_t0[lo]) * n_wg + lo;
}

// one thread per (block, heavy row) pair: copies the pair's entries into the tiled stream (after the padding
// entries that bridge a long step)
template <class P, int VT>
__global__ void acc_fill_kernel(const P *__restrict__ rp, const int32_t *__restrict__ ci, const void *__restrict__ vs,
                                const int32_t *__restrict__ heavy_row, int32_t n_heavy, int32_t n_blocks, int32_t cb,
                                const int64_t *__restrict__ off, const int64_t *__restrict__ blk_tile0,
                                const int32_t *__restrict__ pstart, const int32_t *__restrict__ gap,
                                double *__restrict__ pvals, uint16_t *__restrict__ pidx, int32_t *__restrict__ tile_row0,
                                const int64_t *__restrict__ wg_t0, int32_t n_wg)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_heavy * n_blocks) return;
    const int32_t b = (int32_t)(i / n_heavy), c = (int32_t)(i % n_heavy);
    const int64_t n_all = off[i + 1] - off[i];
    if (n_all == 0) return;
    const int32_t g = gap[i];
    const int32_t npad = g > ACC_MAXSTEP ? (g - 1) / ACC_MAXSTEP : 0;
    const int64_t n = n_all - npad;
    const int32_t r = heavy_row[c];
    const int64_t lo = (int64_t)rp[r] + pstart[i];
    int64_t L = blk_tile0[b] * ACC_TILE + (off[i] - off[(int64_t)b * n_heavy]);      // logical position
    int64_t t_of = -1, pt = 0;                                 // the logical tile last looked up and where it is stored
    for (int32_t k = 1; k <= npad; k++, L++) {                 // padding entry k stands on heavy row c - g + 7k
        const int64_t t = L / ACC_TILE;
        const int el = (int)(L % ACC_TILE);
        if (t != t_of) pt = acc_phys_tile(t_of = t, wg_t0, n_wg);
        pvals[pt * ACC_TILE + acc_val_slot(el)] = 0.0;
        pidx[pt * ACC_TILE + el] = (uint16_t)((uint32_t)cb | ((el ? (uint32_t)ACC_MAXSTEP : 0u) << ACC_ROW_SHIFT));
        if (el == 0) tile_row0[t] = c - g + ACC_MAXSTEP * k;
    }
    int32_t step = g - ACC_MAXSTEP * npad;                     // first entry: from the previous row (or padding) to c
    for (int64_t k = lo; k < lo + n; k++, L++) {
        const int64_t t = L / ACC_TILE;
        const int el = (int)(L % ACC_TILE);
        if (t != t_of) pt = acc_phys_tile(t_of = t, wg_t0, n_wg);
        pvals[pt * ACC_TILE + acc_val_slot(el)] = ValLoad<VT>::at(vs, k);
        pidx[pt * ACC_TILE + el] = (uint16_t)((uint32_t)(ci[k] - b * cb) | ((el ? (uint32_t)step : 0u) << ACC_ROW_SHIFT));
        if (el == 0) tile_row0[t] = c;
        step = 0;
    }
}

// one workgroup per block: pads the block's last tile with (0.0, column slot ACC_CB (a zero in LDS), step 0 = the
// block's last heavy row) -- a padding entry adds 0.0 * 0.0 to an accumulator
__global__ __launch_bounds__(256) void acc_pad_kernel(const int64_t *__restrict__ off, int32_t n_heavy, int32_t n_blocks,
                                                     const int64_t *__restrict__ blk_tile0, double *__restrict__ pvals,
                                                     uint16_t *__restrict__ pidx, const int64_t *__restrict__ wg_t0, int32_t n_wg)
{
    const int32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const int64_t cnt = off[(int64_t)(b + 1) * n_heavy] - off[(int64_t)b * n_heavy];
    const int64_t L0 = blk_tile0[b] * ACC_TILE + cnt, L1 = blk_tile0[b + 1] * ACC_TILE;
    for (int64_t L = L0 + threadIdx.x; L < L1; L += blockDim.x) {      // (the tail of the block's last tile: one tile)
        const int64_t pt = acc_phys_tile(L / ACC_TILE, wg_t0, n_wg);
        const int el = (int)(L % ACC_TILE);
        pvals[pt * ACC_TILE + acc_val_slot(el)] = 0.0;
        pidx[pt * ACC_TILE + el] = (uint16_t)ACC_CB;
    }
}

// (wave_exscan_i32 / wave_segscan: wave.h)

template <int CB, int PT>
__global__ __launch_bounds__(PT) void spmv_acc_kernel(const double *__restrict__ pvals, const uint16_t *__restrict__ pidx,
                                                     const int32_t *__restrict__ tile_row0,
                                                     const double *__restrict__ x, int32_t ncols,
                                                     const AccSeg *__restrict__ segs, const int32_t *__restrict__ wg_seg,
                                                     int32_t H, double *__restrict__ partial)
{
    extern __shared__ __align__(16) unsigned char acc_smem[];
    double *s_x = (double *)acc_smem;                     // CB + 2 (slot CB = 0.0 for padding entries)
    double *s_acc = s_x + CB + 2;                         // Hpad
    const int Hpad = (H + 1) & ~1;
    double *s_hval = s_acc + Hpad;                        // ACC_SEG_TILES
    int32_t *s_hrow = (int32_t *)(s_hval + ACC_SEG_TILES);
    constexpr int NW = PT / WAVE;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;

    for (int h = tid; h < Hpad; h += PT) s_acc[h] = 0.0;
    if (tid < 2) s_x[CB + tid] = 0.0;
    const unsigned long long below = lane ? (~0ull >> (WAVE - lane)) : 0ull;      // lanes < lane
    const unsigned long long upto = below | (1ull << lane);                       // lanes <= lane

    int cur_blk = -1;
    const int sb = wg_seg[blockIdx.x], se = wg_seg[blockIdx.x + 1];
    for (int si = sb; si < se; si++) {
        const AccSeg sg = segs[si];
        if (sg.blk != cur_blk) {
            // every wavefront is past barrier B of the previous segment: nobody reads s_x any more
            cur_blk = sg.blk;
            const int32_t w0 = sg.blk * CB;
            const int wlen = ncols - w0 < CB ? ncols - w0 : CB;
            if (wlen == CB) {
                constexpr int WL = CB / 2 / PT;
                f64x2_t v[WL];
#pragma unroll
                for (int u = 0; u < WL; u++) v[u] = *((const F64x2 *)(x + w0) + tid + u * PT);
#pragma unroll
                for (int u = 0; u < WL; u++) ((f64x2_t *)s_x)[tid + u * PT] = v[u];
            } else {
                for (int k = tid; k < wlen; k += PT) s_x[k] = x[w0 + k];
            }
        }
        __syncthreads();      // A: window stored; heads of the previous segment folded in; accumulators zeroed

        const int nt = sg.ntiles;
        f64x2_t v[4], vn[4];
        u32x4_t ix, ixn;                  // eight 16-bit index words per lane
        int32_t tr0 = 0, tr0n = 0;        // heavy-row index of the tile's first entry
        int t = wv;
        const int64_t pstep = (int64_t)gridDim.x;      // the workgroups' tiles are interleaved (acc_phys_tile)
        if (t < nt) {
            const f64x2_t *vp = (const f64x2_t *)(pvals + (sg.ptile0 + t * pstep) * ACC_TILE);
            const u32x4_t *ip = (const u32x4_t *)(pidx + (sg.ptile0 + t * pstep) * ACC_TILE);
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = __builtin_nontemporal_load(vp + q * WAVE + lane);
            ix = __builtin_nontemporal_load(ip + lane);
            tr0 = tile_row0[sg.tile0 + t];
        }
        for (; t < nt; t += NW) {
            const bool more = t + NW < nt;
            if (more) {      // next tile's loads are in flight while this one is reduced
                const f64x2_t *vp = (const f64x2_t *)(pvals + (sg.ptile0 + (t + NW) * pstep) * ACC_TILE);
                const u32x4_t *ip = (const u32x4_t *)(pidx + (sg.ptile0 + (t + NW) * pstep) * ACC_TILE);
#pragma unroll
                for (int q = 0; q < 4; q++) vn[q] = __builtin_nontemporal_load(vp + q * WAVE + lane);
                ixn = __builtin_nontemporal_load(ip + lane);
                tr0n = tile_row0[sg.tile0 + t + NW];
            }
            const uint32_t e[ACC_K] = {ix.x & 0xffffu, ix.x >> 16, ix.y & 0xffffu, ix.y >> 16,
                                       ix.z & 0xffffu, ix.z >> 16, ix.w & 0xffffu, ix.w >> 16};
            const double a[ACC_K] = {v[0].x, v[0].y, v[1].x, v[1].y, v[2].x, v[2].y, v[3].x, v[3].y};
            double xv[ACC_K];
#pragma unroll
            for (int j = 0; j < ACC_K; j++) xv[j] = s_x[e[j] & ACC_COL_MASK];
            // heavy-row index of every entry: the tile's first row + the running sum of the steps
            int rw[ACC_K];
            {
                int lsum = 0;
#pragma unroll
                for (int j = 0; j < ACC_K; j++) lsum += (int)(e[j] >> ACC_ROW_SHIFT);
                int run = tr0 + wave_exscan_i32(lsum, lane);
#pragma unroll
                for (int j = 0; j < ACC_K; j++) {
                    run += (int)(e[j] >> ACC_ROW_SHIFT);
                    rw[j] = run;
                }
            }
            // lane-local: ordered sum per row; the first run is the lane's head, the last its tail, runs in
            // between start and end inside this lane and go straight to their accumulators
            const int hr = rw[0];
            int cur = hr;
            double acc = 0.0, hs = 0.0;
            bool nb = false;
#pragma unroll
            for (int j = 0; j < ACC_K; j++) {
                const int r = rw[j];
                if (r != cur) {
                    if (!nb) {
                        hs = acc;
                        nb = true;
                    } else {
                        atomicAdd(&s_acc[cur], acc);
                    }
                    acc = 0.0;
                    cur = r;
                }
                acc += a[j] * xv[j];
            }
            const int tr = cur;
            const double ts = acc;
            if (!nb) hs = acc;
            // join the runs that cross lanes
            const int tr_prev = wave_up1_i32(tr, -1);
            const bool ne = lane > 0 && tr_prev != hr;           // a run ends between lane - 1 and this lane
            const double T = wave_segscan(ts, nb || ne, lane);    // running sum of this lane's tail run
            const double T_prev = wave_up1_f64(T, 0.0);
            const double X = (lane > 0 && !ne) ? T_prev : 0.0;    // what earlier lanes carry into this lane's head
            const unsigned long long m_nb = __ballot(nb), m_ne = __ballot(ne);
            const bool before = ((m_nb & below) | (m_ne & upto)) != 0;   // some run ended before this lane's head
            const int ne_next = wave_down1_i32((int)ne, 1);
            const bool tail_done = lane == WAVE - 1 || ne_next != 0;
            if (nb) {                                            // the head run ends inside this lane
                const double hv = hs + X;
                if (before) {
                    atomicAdd(&s_acc[hr], hv);
                } else {                                         // it is the tile's leading run
                    s_hrow[t] = hr;
                    s_hval[t] = hv;
                }
            }
            if (tail_done) {
                if (nb || before) {
                    atomicAdd(&s_acc[tr], T);
                } else {                                         // the whole tile up to here is one run
                    s_hrow[t] = tr;
                    s_hval[t] = T;
                }
            }
            if (more) {
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = vn[q];
                ix = ixn;
                tr0 = tr0n;
            }
        }
        __syncthreads();      // B: every tile's accumulator adds and head slot are in LDS
        if (wv == 0) {
            // fold the heads in, in tile order (rows ascend inside a block, so equal rows are adjacent)
            for (int base = 0; base < nt; base += WAVE) {
                const int i = base + lane;
                const bool ok = i < nt;
                const int row = ok ? s_hrow[i] : -1 - lane;
                const double val = ok ? s_hval[i] : 0.0;
                const int row_prev = wave_up1_i32(row, -2 - WAVE);
                const double S = wave_segscan(val, lane == 0 || row_prev != row, lane);
                const int row_next = wave_down1_i32(row, -2 - WAVE);
                if (ok && (lane == WAVE - 1 || row_next != row)) atomicAdd(&s_acc[row], S);
            }
        }
    }
    __syncthreads();
    for (int h = tid; h < H; h += PT) partial[(int64_t)blockIdx.x * H + h] = s_acc[h];
}

// The per-SpMV epilogue -- the carry fix-up of the light stream, the ordered reduces of tier 0 (over the accumulator
// kernel's workgroups) and of tier 1 (over the column blocks, plus the pair kernel's carries) -- is ONE launch covering up
// to six jobs (a job = a contiguous range of 1024-thread workgroups).  As separate launches the two small kernels took
// 5.4 + 14.7 us of a 0.553 ms SpMV, mostly launch latency and exposed round trips.
//   fix:    y[row] += the carries of the tiles that end inside `row`, in tile order (one thread per tile; the first tile of a
//           run of equal carry_row adds the whole run).
//   reduce: y[row_list[h]] = sum over w < n_wg of partial[w][h], in order, then the row's listed carries (crp/cidx: the
//           tiles of the pair kernel whose last row end falls in a pair of long row h), in tile order.  A workgroup takes
//           1024 / (64 G) sets of 64 rows; the G wavefronts of a set each sum a contiguous range of w, joined in order
//           through LDS.  G = 4 for tier 0 (256 partials per row: 64 per wavefront), 2 for tier 1 (38 blocks).  Measured
//           on the headline matrix (tier 0 / tier 1): 16 / 2 -> 21.1 us, 8 / 2 -> 19.1, 4 / 2 -> 17.4, 2 / 2 -> 23.7,
//           1 / 2 -> 38.2, 4 / 4 -> 18.5, 16 / 4 (588 workgroups: two rounds) -> 21.9.
struct EpiJob {
    int32_t kind, blocks;      // 0 = fix, 1 = reduce
    // fix
    const int32_t *carry_row;
    const double *carry_val;
    int64_t n;
    double *fy;
    // reduce
    const double *partial;
    const int32_t *row_list;
    int32_t H, n_wg, G;
    const int32_t *crp, *cidx;      // optional (nullptr: no carries to add)
    const double *cval;
};
struct EpiJobs {
    EpiJob j[6];
    int32_t n;
};
constexpr int EPI_THREADS = 1024;

__global__ __launch_bounds__(EPI_THREADS) void spmv_epilogue_kernel(EpiJobs jobs, double *__restrict__ y)
{
    __shared__ double s_p[EPI_THREADS / WAVE][WAVE];
    int b = blockIdx.x, q = 0;
    while (q + 1 < jobs.n && b >= jobs.j[q].blocks) b -= jobs.j[q++].blocks;
    const EpiJob &J = jobs.j[q];
    if (J.kind == 0) {
        const int64_t t = (int64_t)b * EPI_THREADS + threadIdx.x;
        if (t >= J.n) return;
        const int32_t row = J.carry_row[t];
        if (row < 0) return;
        if (t > 0 && J.carry_row[t - 1] == row) return;
        double acc = J.carry_val[t];
        for (int64_t u = t + 1; u < J.n && J.carry_row[u] == row; u++) acc += J.carry_val[u];
        J.fy[row] = acc + J.fy[row];
        return;
    }
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    const int G = J.G, set = wv / G, g = wv % G;                      // G divides 16
    const int h = (b * (EPI_THREADS / WAVE / G) + set) * WAVE + lane;
    const int per = (J.n_wg + G - 1) / G;
    const int w0 = g * per, w1 = w0 + per < J.n_wg ? w0 + per : J.n_wg;
    double acc = 0.0;
    if (h < J.H) {
#pragma unroll 8
        for (int w = w0; w < w1; w++) acc += J.partial[(int64_t)w * J.H + h];
    }
    s_p[wv][lane] = acc;
    __syncthreads();
    if (g == 0 && h < J.H) {
        double tot = s_p[wv][lane];
        for (int u = 1; u < G; u++) tot += s_p[wv + u][lane];
        if (J.crp)
            for (int32_t k = J.crp[h]; k < J.crp[h + 1]; k++) tot += J.cval[J.cidx[k]];
        y[J.row_list[h]] = tot;
    }
}

// ---- short rows: the light stream ---------------------------------------------------------------------
// The merge-path tile kernel spends ~600 vector instructions per wavefront-tile on index arithmetic (clamped
// 64-bit addresses, merge coordinates, the cut table) and four dependent memory round trips per 2048-item
// tile; with every x gather served from L1 it still took 0.25-0.28 ms on the headline matrix against
// 0.09 ms for its 0.61 GB at streaming rate (measured: SQ counters, gather ablation).  The light stream is
// the same idea as the accumulator form, for the rows that stay on the row-major path: at plan time their
// entries are copied into a private stream of (float64 value, uint32 index) tiles of 512 = 64 lanes x 8
// consecutive entries, lane-interleaved for 16-B coalesced loads, with
//     index bit 31  hot column (low bits = slot in the packed xh), else low bits = column
//     index bit 30  first entry of its row
// plus rowids[k] = k-th non-empty row of the view and tile_base[t] (run numbering, below).  One wavefront
// per tile, no LDS, no workgroup barrier:
//   * every lane sums its 8 entries run by run in storage order; a run that starts and ends inside the lane
//     is stored to y at once;
//   * a segmented scan over the lanes joins runs that cross lanes (fixed tree order: deterministic);
//   * the tile's leading run, when it continues a row of the previous tile, goes to carry[] and the
//     existing fix-up kernel adds it to y in tile order.
// Rows without a run (no entries, or served by the tiers) are cleared by the run that follows them.
// Run numbering: run(e) = tile_base[t] - 1 + #{row starts in the tile up to and including e}, with
// tile_base[t] = index of the row holding the tile's first entry among the non-empty rows, + 1 if that entry
// is not the row's first.  Needs ncols < 2^30 (two flag bits); otherwise the tile kernel stays in charge.
// One persistent workgroup per CU: 8 wavefronts and the 15360 most referenced packed columns in LDS (120 KiB + 8 staging
// buffers = 152 KiB).  Measured on the headline matrix (stream kernel alone; threads / LDS slots): 1024 / 8192 -> 0.204 ms,
// 512 / 8192 -> 0.195, 640 / 14336 -> 0.200, 512 / 15360 -> 0.190, 448 / 15872 -> 0.196, 384 / 16384 -> 0.200, 256 / 17408 ->
// 0.230: the kernel queues on the CU's vector memory path, and eight wavefronts keep it as busy
// as sixteen while leaving LDS for twice the columns.
#ifndef CSRK_LS_THREADS
#define CSRK_LS_THREADS 512
#endif
constexpr int LS_THREADS = CSRK_LS_THREADS;
#ifndef CSRK_LS_HOT_LDS
#define CSRK_LS_HOT_LDS 15360
#endif
constexpr int LS_HOT_LDS = CSRK_LS_HOT_LDS;
// round-in-LDS form of the stream kernel (LS_RND): LDS = LS_RND_HOT hot slots + the LS_RND_CAP staged values of the
// workgroup's current round + the run-sum buffers
#ifndef CSRK_LS_RND_CAP
#define CSRK_LS_RND_CAP 8192
#endif
constexpr int LS_RND_CAP = CSRK_LS_RND_CAP, LS_RND_MAXTILES = 128;
constexpr int LS_RND_HOT = (160 * 1024 - (CSRK_LS_THREADS / 64) * (512 + 2) * 8) / 8 - LS_RND_CAP;      // 8176 with the defaults
constexpr int LS_RID = 4;        // batches of 64 run-slot row ids fetched ahead per tile (2 / 3 / 4: 0.557 / 0.553 / 0.553 ms)
#ifndef CSRK_LS_SEQ
#define CSRK_LS_SEQ 3
#endif
constexpr int LS_SEQ = CSRK_LS_SEQ;        // rounds of in-order carry hand-over (runs over <= LS_SEQ + 1 lanes are exact)
constexpr uint32_t LS_HOT_BIT = 1u << 31, LS_START_BIT = 1u << 30, LS_COL_MASK = (1u << 30) - 1;
constexpr uint32_t LS_PAD = LS_COL_MASK;      // a padding slot: value 0.0, "column" 2^30 - 1 (never a real one), no flags

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
template <class P, int VT>
__global__ __launch_bounds__(256) void ls_fill_kernel(const P *__restrict__ src, const P *__restrict__ rpv, int32_t nrows,
                                                     const int32_t *__restrict__ ci, const void *__restrict__ vs,
                                                     int64_t n_ent, int64_t n_slots, const int32_t *__restrict__ slot_map, double *__restrict__ svals,
                                                     uint32_t *__restrict__ sidx, const P *__restrict__ rp_len)
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
        const int32_t r = ls_row_of(rpv, nrows, L);
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
    svals[t * ACC_TILE + acc_val_slot(el)] = v;
    sidx[t * ACC_TILE + acc_idx_slot(el)] = ix;
}

// per tile: run numbering base
template <class P>
__global__ void ls_tilebase_kernel(const P *__restrict__ rpv, int32_t nrows, const int32_t *__restrict__ ridx, int64_t n_tiles,
                                   int32_t *__restrict__ tile_base)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const int64_t e0 = t * ACC_TILE;      // (< the entry count: the stream has no empty tile)
    const int32_t r = ls_row_of(rpv, nrows, e0);
    tile_base[t] = ridx[r] + ((int64_t)rpv[r] == e0 ? 0 : 1);
}

// LDS: [0, LS_HOT_LDS) the x values of the most popular packed columns (slots below n_lds are read from
// here instead of gathered), then one staging buffer of ACC_TILE + 2 run sums per wavefront.
// MODE: LS_PLAIN = the unpacked columns' x values are gathered from x (nothing was staged);
// LS_RND = round-in-LDS staging: the workgroup walks ROUNDS of `stage_tiles` consecutive tiles (wavefront w takes tiles
// w, w + 8, ... of the round); `x` is xg and round_start[r] the start of round r's staged values in it, which the workgroup
// copies into LDS with coalesced loads (requested one round ahead, into registers) -- a cold entry's index word holds its
// offset there.  The copy pass can then use rounds of 64 tiles (its store transactions are per (round, column block)
// bucket: 0.049 ms against 0.065 at 8 tiles) without the stream side paying for it in L1 lines (0.257 ms at 64 tiles when
// the round's range is read by gathers).  Two workgroup barriers per round.
// DENSE: run k is row k (LightStream::dense): row ids are not loaded and there are no gaps between runs to clear.
constexpr int LS_PLAIN = 0, LS_RND = 2;
template <int MODE, bool DENSE = false>
__global__ __launch_bounds__(LS_THREADS) void spmv_lstream_kernel(
    const double *__restrict__ svals, const uint32_t *__restrict__ sidx, const int32_t *__restrict__ rowids,
    const int32_t *__restrict__ tile_base, const double *__restrict__ x,
    const double *__restrict__ xh, int32_t n_lds, int64_t n_tiles, int32_t n_runs, int32_t nrows,
    double *__restrict__ y, int32_t *__restrict__ carry_row, double *__restrict__ carry_val,
    const int32_t *__restrict__ round_start, const int32_t *__restrict__ round_tile0, const int32_t *__restrict__ wg_round0)
{
    // No FMA contraction in this kernel: the reference rounds every product before adding it.  (HIP's rounding
    // intrinsics for multiply and add are plain * and + inside inline functions compiled with
    // -ffp-contract=fast and fuse after inlining -- measured: 2041 instead of 75 rows of BASELINE configs[0]
    // differed in the last bits; the pragma governs the operators written in this body.)
#pragma clang fp contract(off)
    constexpr bool RND = MODE == LS_RND;
    extern __shared__ __align__(16) unsigned char ls_smem[];
    double *s_hot = (double *)ls_smem;
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);      // wave-uniform: tile numbers and bases stay scalar
    double *s_rnd = s_hot + LS_RND_HOT;                    // RND: the round's staged values
    double *s_out = s_hot + (RND ? LS_RND_HOT + LS_RND_CAP : LS_HOT_LDS) + wv * (ACC_TILE + 2);
    for (int i = threadIdx.x; i < n_lds; i += LS_THREADS) s_hot[i] = xh[i];
    __syncthreads();

    // A workgroup walks ROUNDS; wavefront w takes tiles w, w + NW, ... of a round.  RND: the workgroup's rounds are
    // wg_round0[b] .. wg_round0[b + 1], round R = tiles round_tile0[R] .. round_tile0[R + 1] (equal shares of the tiles per
    // workgroup, cut into rounds: build_cold_stage).  Otherwise a round is one tile per wavefront and workgroup b takes
    // rounds b, b + grid, ...
    constexpr int NW = LS_THREADS / WAVE;
    const int64_t R_begin = RND ? (int64_t)wg_round0[blockIdx.x] : (int64_t)blockIdx.x;
    const int64_t R_end = RND ? (int64_t)wg_round0[blockIdx.x + 1] : (n_tiles + NW - 1) / NW;
    const int64_t R_step = RND ? 1 : (int64_t)gridDim.x;
    auto round_t0 = [&](int64_t R_) -> int64_t { return RND ? (int64_t)round_tile0[R_] : R_ * NW; };
    auto round_t1 = [&](int64_t R_) -> int64_t {
        if (RND) return (int64_t)round_tile0[R_ + 1];
        return (R_ + 1) * NW < n_tiles ? (R_ + 1) * NW : n_tiles;
    };
    const int64_t wave0 = R_begin < R_end ? round_t0(R_begin) + wv : n_tiles;
    f64x2_t v[4], vn[4];
    u32x4_t ix[2], ixn[2];
    int32_t tb = 0, tbn = 0;
    constexpr int RQ = LS_RND_CAP / 2 / LS_THREADS;      // RND: 16-B loads per thread that cover a round's staged values
    f64x2_t rv[RND ? RQ : 1];
    auto round_request = [&](int64_t R_) {      // the staged values of round R_ -> registers (pairs past its count re-read its first)
        const int32_t r0 = round_start[R_];
        const int32_t rn = round_start[R_ + 1] - r0;
#pragma unroll
        for (int q = 0; q < RQ; q++) {
            const int k = 2 * (q * LS_THREADS + (int)threadIdx.x);
            // (non-temporal, like the y stores below: read / written once per product -- kept out of the caches they leave
            // x there for the next product's copy pass: 48 -> 42 us, the step 0.547 -> 0.538 ms)
            rv[q] = __builtin_nontemporal_load((const F64x2 *)(x + r0 + (k < rn ? k : 0)));
        }
    };
    if (RND && R_begin < R_end) round_request(R_begin);
    int64_t t = wave0;
    if (R_begin < R_end && t < round_t1(R_begin)) {
        const f64x2_t *vp = (const f64x2_t *)(svals + t * ACC_TILE);
        const u32x4_t *ip = (const u32x4_t *)(sidx + t * ACC_TILE);
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = __builtin_nontemporal_load(vp + q * WAVE + lane);
#pragma unroll
        for (int q = 0; q < 2; q++) ix[q] = __builtin_nontemporal_load(ip + q * WAVE + lane);
        tb = __builtin_amdgcn_readfirstlane(tile_base[t]);
    }
    for (int64_t R = R_begin; R < R_end; R += R_step) {
    if (RND) {
        __syncthreads();      // every wavefront is done with the previous round's values
#pragma unroll
        for (int q = 0; q < RQ; q++) ((f64x2_t *)s_rnd)[q * LS_THREADS + threadIdx.x] = rv[q];
        __syncthreads();
        if (R + R_step < R_end) round_request(R + R_step);      // in flight across this round's tiles
    }
    const int64_t rt1 = round_t1(R);
    for (t = round_t0(R) + wv; t < rt1; t += NW) {
        // the wavefront's next tile: in this round, else in the workgroup's next round (every round but a workgroup's last
        // is a whole number of tiles per wavefront), else none (itself)
        int64_t t_next = t + NW;
        if (t_next >= rt1) {
            t_next = t;
            if (R + R_step < R_end) {
                const int64_t tf = round_t0(R + R_step) + wv;
                if (tf < round_t1(R + R_step)) t_next = tf;
            }
        }
        const uint32_t e[ACC_K] = {ix[0].x, ix[0].y, ix[0].z, ix[0].w, ix[1].x, ix[1].y, ix[1].z, ix[1].w};
        const double a[ACC_K] = {v[0].x, v[0].y, v[1].x, v[1].y, v[2].x, v[2].y, v[3].x, v[3].y};
        // Issue order matters: vmcnt retires loads in issue order, so whatever is requested BEFORE the loads this tile
        // waits for is waited for too.  This tile's own loads (staged values / gathers, row ids) therefore go first and
        // the next tile's stream loads -- HBM latency -- are requested after them and stay in flight across the whole
        // tile.  (Requested first, as they used to be, every tile waited for the next tile's HBM loads before its first
        // multiply.)
        double gv[ACC_K], lv[ACC_K];
        bool inl[ACC_K];
        if (RND) {
            // packed columns beyond the LDS slots: gathered from the pack; everything else is in LDS
#pragma unroll
            for (int j = 0; j < ACC_K; j++) {
                const uint32_t c = e[j] & LS_COL_MASK;
                const bool hot = (e[j] & LS_HOT_BIT) != 0;
                inl[j] = c != LS_PAD && (!hot || (int32_t)c < n_lds);      // (a padding slot multiplies 0 * 0)
                gv[j] = 0.0;
                if (hot && (int32_t)c >= n_lds) gv[j] = xh[c];
            }
        } else {
            // x values: the most popular packed columns from LDS, the others gathered (lanes served from LDS
            // are masked out of the gather, which is what the texture path charges for); all eight in flight
#pragma unroll
            for (int j = 0; j < ACC_K; j++) {
                const uint32_t c = e[j] & LS_COL_MASK;
                const bool hot = (e[j] & LS_HOT_BIT) != 0;
                inl[j] = hot && (int32_t)c < n_lds;
                gv[j] = 0.0;
                const double *g = hot ? xh + c : x + c;
                if (!inl[j] && c != LS_PAD) gv[j] = *g;          // padding slots gather nothing and multiply 0 * 0
            }
        }
        // row ids of the tile's first run slots (slot k <-> run tb - 1 + k): requested now, used at the end
        int32_t rid[LS_RID];
#pragma unroll
        for (int i = 0; i < LS_RID; i++) {
            const int run = tb - 1 + lane + i * WAVE;
            const int runc = run < 0 ? 0 : (run > n_runs - 1 ? n_runs - 1 : run);
            rid[i] = DENSE ? runc : rowids[runc];
        }
        asm volatile("" ::: "memory");      // (compiler-level: keep the two groups of loads in this order)
        {
            // The next tile's stream loads stay in flight while this one is gathered and reduced.  UNCONDITIONAL (the
            // last tile re-requests itself): behind an `if (more)` the compiler cannot count the loads in flight at the
            // join and waits for all of them (s_waitcnt vmcnt(0)) before this tile's first multiply.
            const int64_t tn = t_next;
            const f64x2_t *vp = (const f64x2_t *)(svals + tn * ACC_TILE);
            const u32x4_t *ip = (const u32x4_t *)(sidx + tn * ACC_TILE);
            // (index words first: the next tile's gathers need them at its very top, the values only at its multiplies)
#pragma unroll
            for (int q = 0; q < 2; q++) ixn[q] = __builtin_nontemporal_load(ip + q * WAVE + lane);
#pragma unroll
            for (int q = 0; q < 4; q++) vn[q] = __builtin_nontemporal_load(vp + q * WAVE + lane);
            tbn = tile_base[tn];
        }
        asm volatile("" ::: "memory");
        // row starts: bit j of st = entry j opens a row (index words only: this runs while the tile's x values arrive)
        uint32_t st = 0;
#pragma unroll
        for (int j = 0; j < ACC_K; j++) st |= ((e[j] >> 30) & 1u) << j;
        const int cnt = __popc(st);
        const int S = wave_exscan_i32(cnt, lane);          // row starts in the lanes below
        const int total = wave_last_i32(S + cnt);            // row starts in the tile
        if (RND) {
#pragma unroll
            for (int j = 0; j < ACC_K; j++) {
                const uint32_t c = e[j] & LS_COL_MASK;
                const bool cold = !(e[j] & LS_HOT_BIT) && c != LS_PAD;
                const double *src = cold ? s_rnd + c : s_hot + (inl[j] ? c : 0);
                lv[j] = *src;
            }
        } else {
#pragma unroll
            for (int j = 0; j < ACC_K; j++) lv[j] = s_hot[inl[j] ? (e[j] & LS_COL_MASK) : 0];
        }
        // products, rounded on their own like the reference's `v * x` (contraction is off in this kernel:
        // short rows are to come out bit-identical to the sequential loop)
        double pr[ACC_K];
#pragma unroll
        for (int j = 0; j < ACC_K; j++) {
            pr[j] = a[j] * (inl[j] ? lv[j] : gv[j]);
        }
        // Run sums go to the wavefront's staging buffer: slot 0 = the tile's leading run (the part of a row
        // begun in an earlier tile; 0.0 if the tile opens a row), slot k = the run opened by the tile's k-th
        // row start.  Lane-local pass: runs that start and end inside the lane.
        double acc = 0.0, hs = 0.0;
        bool started = false;
#pragma unroll
        for (int j = 0; j < ACC_K; j++) {
            if ((st >> j) & 1u) {
                if (started) {
                    s_out[S + __popc(st & ((1u << j) - 1))] = acc;      // the run opened by the previous start
                } else {
                    hs = acc;
                    started = true;
                }
                acc = 0.0;
            }
            acc = acc + pr[j];
        }
        const bool has_start = st != 0;
        if (!has_start) hs = acc;
        const double ts = acc;                                   // the lane's last run so far
        // runs that cross lanes, tree order (any length)
        const double T = wave_segscan(ts, has_start, lane);
        const double X = wave_up1_f64(T, 0.0);               // what the lanes below carry into this lane's head
        // ... and in the reference's own order for runs over at most LS_SEQ + 1 lanes: round k hands the
        // exact running sum of the lane below to a lane that has not got its carry yet, which then re-adds
        // its head entries one by one on top of it
        const int first = has_start ? __ffs(st) - 1 : ACC_K;    // entries before the lane's first row start
        double Tq = ts, Hq = hs;
        bool okq = has_start || lane == 0, hok = lane == 0;
#pragma unroll
        for (int it = 0; it < LS_SEQ; it++) {
            const double Xq = wave_up1_f64(Tq, 0.0);
            const int xok = wave_up1_i32((int)okq, 0);
            const bool take = lane > 0 && xok && !hok;
            double sum = Xq;
#pragma unroll
            for (int j = 0; j < ACC_K; j++) sum = sum + (j < first ? pr[j] : -0.0);
            Hq = take ? sum : Hq;
            hok = hok || take;
            if (!has_start) {
                Tq = take ? sum : Tq;
                okq = okq || take;
            }
        }
        if (has_start) s_out[S] = hok ? Hq : hs + X;             // the head run ends in this lane
        if (lane == WAVE - 1) s_out[S + cnt] = okq ? Tq : T;     // the tile's last run (continued by the next tile's slot 0)
        // out: slot k -> the row of run tile_base - 1 + k; consecutive lanes write ascending (mostly
        // consecutive) rows.  LDS operations of one wavefront complete in order: no barrier needed.
        // Rows without a run -- empty rows, rows served by the tiers (their reduce kernels overwrite y later
        // in the stream) -- get their zero from the run that follows them: slot k also clears the rows
        // between the previous run's row and its own.
        // (The row ids of the first LS_RID * 64 slots were requested with the gathers -- the dependent
        // rowids -> store round trips per 64 runs were a third of a tile's latency; the id of the previous run's
        // row is the lane below's.)
        // The batches whose row ids are in registers run as straight-line code with NO load inside: a load in this
        // loop (the row ids of batch LS_RID and beyond, a tile with more than LS_RID * 64 runs) makes the compiler drain
        // the wavefront's memory queue -- every y store of the batch before, s_waitcnt vmcnt(0) -- once per batch
        // (measured with in-kernel stamps: a third of a tile's cycles went there).
        int32_t r_last = -1;                                   // row of the slot before this batch of 64
        auto out_batch = [&](const int k0, const int32_t r) {
            const int k = k0 + lane;
            const int run = tb - 1 + k;
            const int32_t r_prev = wave_up1_i32(r, r_last);
            r_last = wave_last_i32(r);
            int64_t g0 = 0, g1 = 0;                             // rows [g0, g1) to clear
            if (k <= total) {
                const double val = s_out[k];
                if (k == 0) {
                    const bool opens = (st & 1u) != 0;           // lane 0: the tile's first entry opens a row
                    carry_val[t] = val;
                    carry_row[t] = opens ? -1 : r;              // r = rowids[tb - 1] (clamped when tb == 0: then it opens)
                } else {
                    __builtin_nontemporal_store(val, y + r);
                    if (!DENSE) {
                        g0 = run > 0 ? (int64_t)r_prev + 1 : 0;
                        g1 = r;
                    }
                }
            }
            if (DENSE) return;
            const int64_t gap = g1 - g0;
            if (gap > 0 && gap <= 4) {
                for (int64_t q = g0; q < g1; q++) y[q] = 0.0;
            }
            // long gaps: the whole wavefront clears them, one after the other
            unsigned long long big = __ballot(gap > 4);
            while (big) {
                const int src = __ffsll((long long)big) - 1;
                big &= big - 1;
                const int64_t b0 = __shfl(g0, src, WAVE), b1 = __shfl(g1, src, WAVE);
                for (int64_t q = b0 + lane; q < b1; q += WAVE) y[q] = 0.0;
            }
        };
#pragma unroll
        for (int it = 0; it < LS_RID; it++)
            if (it * WAVE <= total) out_batch(it * WAVE, rid[it]);
#pragma unroll 1
        for (int k0 = LS_RID * WAVE; k0 <= total; k0 += WAVE) {
            const int run = tb - 1 + k0 + lane;
            const int runc = run < 0 ? 0 : (run > n_runs - 1 ? n_runs - 1 : run);
            out_batch(k0, DENSE ? runc : rowids[runc]);
        }
        if (!DENSE && total >= 1 && tb - 1 + total == n_runs - 1) {       // the matrix's last run: the rows after it are this tile's too
            for (int64_t q = (int64_t)rowids[n_runs - 1] + 1 + lane; q < nrows; q += WAVE) y[q] = 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = vn[q];
#pragma unroll
        for (int q = 0; q < 2; q++) ix[q] = ixn[q];
        tb = __builtin_amdgcn_readfirstlane(tbn);
    }
    }
}

// One wavefront per tile: the first tile of each run of equal carry_row adds the whole run,
// in tile order, onto the y entry written by the tile that completed the row.
__global__ __launch_bounds__(256) void spmv_merge_fixup_kernel(const int32_t *__restrict__ carry_row,
                                                              const double *__restrict__ carry_val,
                                                              int64_t n_tiles, double *__restrict__ y)
{
    int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    int lane = threadIdx.x & (WAVE - 1);
    if (t >= n_tiles) return;
    int32_t row = carry_row[t];
    if (row < 0) return;
    if (t > 0 && carry_row[t - 1] == row) return;
    double acc = 0.0;
    for (int64_t u0 = t; u0 < n_tiles; u0 += WAVE) {
        int64_t u = u0 + lane;
        bool ok = u < n_tiles && carry_row[u] == row;
        if (ok) acc += carry_val[u];
        if (__ballot(ok) != ~0ull) break;
    }
    acc = wave_sum(acc);
    if (lane == 0) y[row] = acc + y[row];
}

// Thread-per-tile form of the fix-up, used when runs of equal carry_row are known to be short: with the
// long-row split active no row on a tile path exceeds a few tiles (light rows < 2048 entries, panel
// rows <= one column block), so a serial loop is cheaper than 64 lanes per tile (13 us -> 4 us).
__global__ __launch_bounds__(256) void spmv_merge_fixup_short_kernel(const int32_t *__restrict__ carry_row,
                                                                    const double *__restrict__ carry_val,
                                                                    int64_t n_tiles, double *__restrict__ y)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const int32_t row = carry_row[t];
    if (row < 0) return;
    if (t > 0 && carry_row[t - 1] == row) return;
    double acc = carry_val[t];
    for (int64_t u = t + 1; u < n_tiles && carry_row[u] == row; u++) acc += carry_val[u];
    y[row] = acc + y[row];
}

// ---- vector: one wavefront per row segment ------------------------------------------------
constexpr int VEC_SEG = 4096;

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

template <class P, int VT>
__global__ __launch_bounds__(256) void spmv_vector_kernel(const P *__restrict__ rp, const int32_t *__restrict__ ci,
                                                         const void *__restrict__ vs, const double *__restrict__ x,
                                                         double *__restrict__ y, const int64_t *__restrict__ seg_off,
                                                         const int32_t *__restrict__ seg_row, int64_t n_segs,
                                                         double *__restrict__ seg_part)
{
    int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    int lane = threadIdx.x & (WAVE - 1);
    if (q >= n_segs) return;
    int32_t r = seg_row[q];
    int64_t first = seg_off[r], nseg = seg_off[r + 1] - first;
    int64_t s = (int64_t)rp[r] + (q - first) * VEC_SEG;
    int64_t e = (int64_t)rp[r + 1];
    if (e > s + VEC_SEG && nseg > 1) e = s + VEC_SEG;
    double acc = 0.0;
    for (int64_t k = s + lane; k < e; k += WAVE) acc += x[ci[k]] * ValLoad<VT>::at(vs, k);
    acc = wave_sum(acc);
    if (lane == 0) {
        if (nseg == 1)
            y[r] = acc;
        else
            seg_part[q] = acc;
    }
}

__global__ void spmv_vector_fixup_kernel(const int64_t *__restrict__ seg_off, int32_t nrows,
                                         const double *__restrict__ seg_part, double *__restrict__ y)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    int64_t a = seg_off[r], b = seg_off[r + 1];
    if (b - a <= 1) return;
    double acc = 0.0;
    for (int64_t q = a; q < b; q++) acc += seg_part[q];
    y[r] = acc;
}

// ---- scalar: one lane per row ----------------------------------------------------------------
template <class P, int VT>
__global__ __launch_bounds__(256) void spmv_scalar_kernel(const P *__restrict__ rp, const int32_t *__restrict__ ci,
                                                         const void *__restrict__ vs, const double *__restrict__ x,
                                                         double *__restrict__ y, int32_t nrows)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    int64_t s = rp[r], e = rp[r + 1];
    double acc = 0.0;
    for (int64_t k = s; k < e; k++) acc += x[ci[k]] * ValLoad<VT>::at(vs, k);
    y[r] = acc;
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
    const P *rp = (const P *)m->d_rowptrs;
    const int32_t n = (int32_t)rows.size();
    const int32_t nb = (int32_t)ceil_div(m->ncols > 0 ? m->ncols : 1, cb);
    const int64_t pairs = (int64_t)n * nb;
    CSRK_TRY(pn->row_list.alloc((size_t)n * 4));
    CSRK_HIP(hipMemcpyAsync(pn->row_list.p, rows.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    DevBuf off, bends;
    CSRK_TRY(off.alloc((size_t)(pairs + 1) * 8));
    const unsigned g = (unsigned)ceil_div(pairs + 1, 256);
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
    CSRK_HIP(hipMemcpyAsync(be.data(), bends.p, (size_t)(nb + 1) * 8, hipMemcpyDeviceToHost, s));
    CSRK_HIP(hipStreamSynchronize(s));
    t0[0] = 0;
    for (int32_t b = 0; b < nb; b++) t0[b + 1] = t0[b] + ceil_div((int64_t)n + be[b + 1] - be[b], MERGE_ITEMS);
    const int64_t n_tiles = t0[nb];
    CSRK_HIP(hipMemcpyAsync(bends.p, t0.data(), (size_t)(nb + 1) * 8, hipMemcpyHostToDevice, s));
    CSRK_TRY(pn->tile.alloc((size_t)n_tiles * sizeof(PanelTile)));
    if (pn->p64)
        panel_plan_kernel<int64_t><<<(unsigned)ceil_div(n_tiles, 256), 256, 0, s>>>(
            pn->rp.as<int64_t>(), n, nb, bends.as<int64_t>(), n_tiles, MERGE_ITEMS, pn->tile.as<PanelTile>());
    else
        panel_plan_kernel<int32_t><<<(unsigned)ceil_div(n_tiles, 256), 256, 0, s>>>(
            pn->rp.as<int32_t>(), n, nb, bends.as<int64_t>(), n_tiles, MERGE_ITEMS, pn->tile.as<PanelTile>());
    CSRK_LAUNCH_CHECK();

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
    // the tiles' carries per long row (a tile's carry belongs to the pair holding its last, unfinished row end: static)
    {
        std::vector<PanelTile> ht((size_t)n_tiles);
        CSRK_HIP(hipMemcpy(ht.data(), pn->tile.p, (size_t)n_tiles * sizeof(PanelTile), hipMemcpyDeviceToHost));
        std::vector<int32_t> crp((size_t)n + 1, 0), cidx;
        for (int64_t t = 0; t < n_tiles; t++)
            if ((int64_t)ht[(size_t)t].i1 < pairs) crp[(size_t)(ht[(size_t)t].i1 % n) + 1]++;
        for (int32_t h = 0; h < n; h++) crp[(size_t)h + 1] += crp[(size_t)h];
        cidx.resize((size_t)crp[(size_t)n] + 1);
        std::vector<int32_t> cur(crp.begin(), crp.end() - 1);
        for (int64_t t = 0; t < n_tiles; t++)      // ascending tiles: each row's list comes out in tile order
            if ((int64_t)ht[(size_t)t].i1 < pairs) cidx[(size_t)cur[(size_t)(ht[(size_t)t].i1 % n)]++] = (int32_t)t;
        CSRK_TRY(pn->crp.alloc(crp.size() * 4));
        CSRK_TRY(pn->cidx.alloc(cidx.size() * 4));
        CSRK_HIP(hipMemcpy(pn->crp.p, crp.data(), crp.size() * 4, hipMemcpyHostToDevice));
        CSRK_HIP(hipMemcpy(pn->cidx.p, cidx.data(), cidx.size() * 4, hipMemcpyHostToDevice));
    }
    pn->groups = (int64_t)groups.size();
    CSRK_TRY(pn->group.alloc(groups.size() * sizeof(PanelGroup)));
    CSRK_HIP(hipMemcpyAsync(pn->group.p, groups.data(), groups.size() * sizeof(PanelGroup), hipMemcpyHostToDevice, s));
    CSRK_TRY(pn->carry_row.alloc((size_t)n_tiles * 4));
    CSRK_TRY(pn->carry_val.alloc((size_t)n_tiles * 8));
    CSRK_TRY(pn->y.alloc((size_t)pairs * 8));
    CSRK_HIP(hipStreamSynchronize(s));     // `groups`, `t0` are host temporaries of async copies
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
    const int64_t pairs = (int64_t)n * nb;
    CSRK_TRY(ap->row_list.alloc((size_t)n * 4));
    CSRK_HIP(hipMemcpyAsync(ap->row_list.p, rows, (size_t)n * 4, hipMemcpyHostToDevice, s));
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
    CSRK_HIP(hipMemcpyAsync(d_trow.p, trow.data(), (size_t)n_tasks * 4, hipMemcpyHostToDevice, s));
    CSRK_HIP(hipMemcpyAsync(d_tpiece.p, tpiece.data(), (size_t)n_tasks * 4, hipMemcpyHostToDevice, s));
    acc_pairstart_kernel<P><<<(unsigned)ceil_div(n_tasks * WAVE, 256), 256, 0, s>>>(
        rp, m->d_colinds, ap->row_list.as<int32_t>(), n, nb, ACC_CB, d_trow.as<int32_t>(), d_tpiece.as<int32_t>(), n_tasks,
        pstart.as<int32_t>());
    CSRK_LAUNCH_CHECK();
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
    CSRK_HIP(hipMemcpyAsync(be.data(), bends.p, (size_t)(nb + 1) * 8, hipMemcpyDeviceToHost, s));
    CSRK_HIP(hipStreamSynchronize(s));
    t0[0] = 0;
    for (int32_t b = 0; b < nb; b++) t0[b + 1] = t0[b] + ceil_div(be[b + 1] - be[b], ACC_TILE);
    const int64_t n_tiles = t0[nb];
    CSRK_HIP(hipMemcpyAsync(bends.p, t0.data(), (size_t)(nb + 1) * 8, hipMemcpyHostToDevice, s));
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
    CSRK_HIP(hipMemcpyAsync(d_wg_t0.p, wg_t0.data(), (size_t)(n_wg + 1) * 8, hipMemcpyHostToDevice, s));
    CSRK_TRY(ap->vals.alloc((size_t)n_phys * ACC_TILE * 8));
    CSRK_TRY(ap->idx.alloc((size_t)n_phys * ACC_TILE * 2));
    CSRK_TRY(ap->tile_row0.alloc((size_t)(n_tiles ? n_tiles : 1) * 4));
    acc_fill_kernel<P, VT><<<(unsigned)ceil_div(pairs, 256), 256, 0, s>>>(
        rp, m->d_colinds, m->d_values, ap->row_list.as<int32_t>(), n, nb, ACC_CB, off.as<int64_t>(), bends.as<int64_t>(),
        pstart.as<int32_t>(), gap.as<int32_t>(), ap->vals.as<double>(), ap->idx.as<uint16_t>(), ap->tile_row0.as<int32_t>(),
        d_wg_t0.as<int64_t>(), (int32_t)n_wg);
    CSRK_LAUNCH_CHECK();
    acc_pad_kernel<<<(unsigned)nb, 256, 0, s>>>(off.as<int64_t>(), n, nb, bends.as<int64_t>(), ap->vals.as<double>(),
                                               ap->idx.as<uint16_t>(), d_wg_t0.as<int64_t>(), (int32_t)n_wg);
    CSRK_LAUNCH_CHECK();
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
    CSRK_HIP(hipMemcpyAsync(ap->segs.p, segs.data(), segs.size() * sizeof(AccSeg), hipMemcpyHostToDevice, s));
    CSRK_HIP(hipMemcpyAsync(ap->wg_seg.p, wg_seg.data(), wg_seg.size() * 4, hipMemcpyHostToDevice, s));
    CSRK_TRY(ap->partial.alloc((size_t)n_wg * n * 8));
    ap->lds = (size_t)(ACC_CB + 2) * 8 + (size_t)((n + 1) & ~1) * 8 + (size_t)ACC_SEG_TILES * 12;
    CSRK_HIP(hipFuncSetAttribute((const void *)spmv_acc_kernel<ACC_CB, ACC_THREADS>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    CSRK_HIP(hipStreamSynchronize(s));     // `segs`, `wg_seg`, `t0` are host temporaries of async copies
    ap->nrow = n;
    ap->nb = nb;
    ap->n_wg = (int32_t)n_wg;
    ap->tiles = n_tiles;
    ap->nnz = nnz_rows;
    ap->n_segs = (int64_t)segs.size();
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
    DevBuf flag, hlen, bad, clen;
    CSRK_TRY(flag.alloc((size_t)(nr + 2) * 4));
    CSRK_TRY(hlen.alloc((size_t)(nr + 2) * 8));
    heavy_flag_kernel<P><<<g1, 256, 0, s>>>(rp, nr, flag.as<int32_t>(), hlen.as<int64_t>(), cut_min);
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(exclusive_scan_i32(flag.as<int32_t>(), flag.as<int32_t>(), nr, s));      // -> cut-row index
    CSRK_TRY(exclusive_scan_i64(hlen.as<int64_t>(), hlen.as<int64_t>(), nr, s));      // -> cut entries before
    int32_t n_cut = 0;
    int64_t nnz_cut = 0;
    CSRK_HIP(hipMemcpyAsync(&n_cut, flag.as<int32_t>() + nr, 4, hipMemcpyDeviceToHost, s));
    CSRK_HIP(hipMemcpyAsync(&nnz_cut, hlen.as<int64_t>() + nr, 8, hipMemcpyDeviceToHost, s));
    CSRK_HIP(hipStreamSynchronize(s));
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
    int32_t is_bad = 0;
    std::vector<int32_t> rows((size_t)n_cut);
    std::vector<int64_t> lens((size_t)n_cut);
    CSRK_HIP(hipMemcpyAsync(&is_bad, bad.p, 4, hipMemcpyDeviceToHost, s));
    CSRK_HIP(hipMemcpyAsync(rows.data(), p->heavy_row.p, (size_t)n_cut * 4, hipMemcpyDeviceToHost, s));
    CSRK_HIP(hipMemcpyAsync(lens.data(), clen.p, (size_t)n_cut * 8, hipMemcpyDeviceToHost, s));
    CSRK_HIP(hipStreamSynchronize(s));
    if (is_bad) return CSRK_OK;      // unsorted columns in a long row: column blocking needs order

    // tier 0: accumulator form (groups of <= ACC_MAXROWS rows).  The accumulator form costs 1.8 ps per entry against 4.8 for tier 1 (measured, headline matrix), and
    // one group holds up to ACC_MAXROWS rows at no extra window traffic: when fewer rows than that reach
    // HEAVY_MIN, tier 0 is extended downwards to the ACC_MAXROWS longest rows (not below ACC_FLOOR).
    if (tier1 && !getenv("CSRK_HEAVY_MIN")) {
        int64_t n_min = 0;
        for (int32_t c = 0; c < n_cut; c++) n_min += lens[c] >= HEAVY_MIN;
        if (n_min < ACC_MAXROWS && n_cut > n_min) {
            std::vector<int64_t> sl(lens);
            const size_t kth = (size_t)(n_cut < ACC_MAXROWS ? n_cut : ACC_MAXROWS) - 1;
            std::nth_element(sl.begin(), sl.begin() + kth, sl.end(), [](int64_t a, int64_t b) { return a > b; });
            int64_t thr = sl[kth];
            // rows tied with the kth must not push the group over its capacity
            int64_t n_ge = 0;
            for (int32_t c = 0; c < n_cut; c++) n_ge += lens[c] >= thr;
            if (n_ge > ACC_MAXROWS) thr++;
            thr = thr < ACC_FLOOR ? ACC_FLOOR : thr;
            if (thr < HEAVY_MIN) HEAVY_MIN = (int)thr;
        }
    }
    p->heavy_min = HEAVY_MIN;
    std::vector<int32_t> r0, r1;     // tier 0: >= HEAVY_MIN entries; tier 1: the rest of the cut rows
    std::vector<int64_t> len0;
    int64_t nnz1 = 0;
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
    p->t0_rows = r0;
    p->t0_lens = len0;
    p->t1_rows = r1;
    p->t1_nnz = nnz1;
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
    const int64_t hc_need = ceil_div(ceil_div(m->nrows, row_stride), 256);
    hot_count_kernel<P><<<(unsigned)(hc_need < 2048 ? hc_need : 2048), 256, 0, s>>>(
        (const P *)m->d_rowptrs, p->n_heavy ? p->rp_light.as<P>() : (const P *)nullptr, m->d_colinds, m->nrows,
        row_stride, cnt.as<int32_t>(), census.as<unsigned long long>());
    CSRK_LAUNCH_CHECK();
    unsigned long long n_samples_u = 0;
    CSRK_HIP(hipMemcpyAsync(&n_samples_u, census.p, 8, hipMemcpyDeviceToHost, s));
    CSRK_HIP(hipStreamSynchronize(s));
    const int64_t n_samples = (int64_t)n_samples_u;
    if (n_samples == 0) return CSRK_OK;
    // smallest threshold (>= 2 references in the sample) that leaves at most HOT_SLOTS columns
    auto census_at = [&](int32_t thr, unsigned long long out[2]) -> int {
        CSRK_HIP(hipMemsetAsync(census.p, 0, 16, s));
        hot_census_kernel<<<1024, 256, 0, s>>>(cnt.as<int32_t>(), nc, thr, census.as<unsigned long long>());
        CSRK_LAUNCH_CHECK();
        CSRK_HIP(hipMemcpyAsync(out, census.p, 16, hipMemcpyDeviceToHost, s));
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
        CSRK_HIP(hipMemcpyAsync(hh.data(), hist.p, hh.size() * 8, hipMemcpyDeviceToHost, s));
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
        std::vector<int32_t> hc((size_t)n_hot), hn((size_t)n_hot), ord((size_t)n_hot), sorted((size_t)n_hot);
        CSRK_HIP(hipMemcpyAsync(hc.data(), p->hot_cols.p, (size_t)n_hot * 4, hipMemcpyDeviceToHost, s));
        CSRK_HIP(hipMemcpyAsync(hn.data(), hcnt.p, (size_t)n_hot * 4, hipMemcpyDeviceToHost, s));
        CSRK_HIP(hipStreamSynchronize(s));
        // stable, count descending: two 16-bit LSD radix passes over the complemented count (a comparison sort of
        // 4 * 10^5 indices through a lambda took tens of ms of the plan)
        {
            std::vector<int32_t> tmp((size_t)n_hot);
            std::vector<uint32_t> bucket(65537);
            for (int32_t i = 0; i < n_hot; i++) ord[(size_t)i] = i;
            for (int pass = 0; pass < 2; pass++) {
                const int sh = 16 * pass;
                std::fill(bucket.begin(), bucket.end(), 0u);
                for (int32_t i = 0; i < n_hot; i++) bucket[((~(uint32_t)hn[(size_t)i] >> sh) & 0xffffu) + 1]++;
                for (size_t b = 0; b < 65536; b++) bucket[b + 1] += bucket[b];
                for (int32_t i = 0; i < n_hot; i++) {
                    const int32_t o = ord[(size_t)i];
                    tmp[bucket[(~(uint32_t)hn[(size_t)o] >> sh) & 0xffffu]++] = o;
                }
                ord.swap(tmp);
            }
        }
        for (int32_t i = 0; i < n_hot; i++) sorted[(size_t)i] = hc[(size_t)ord[(size_t)i]];
        CSRK_HIP(hipMemcpyAsync(p->hot_cols.p, sorted.data(), (size_t)n_hot * 4, hipMemcpyHostToDevice, s));
        CSRK_HIP(hipStreamSynchronize(s));      // `sorted` is a host temporary
    }
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
    DevBuf ridx;
    CSRK_TRY(ridx.alloc((size_t)(nrows_view + 2) * 4));
    const unsigned gr = (unsigned)ceil_div((int64_t)nrows_view + 1, 256);
    ls_rowflag_kernel<P><<<gr, 256, 0, s>>>(rpv, nrows_view, ridx.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(exclusive_scan_i32(ridx.as<int32_t>(), ridx.as<int32_t>(), nrows_view, s));
    int32_t n_runs = 0;
    CSRK_HIP(hipMemcpyAsync(&n_runs, ridx.as<int32_t>() + nrows_view, 4, hipMemcpyDeviceToHost, s));
    CSRK_HIP(hipStreamSynchronize(s));
    if (n_runs < 1) return CSRK_OK;
    if (!ls->dense) {
        CSRK_TRY(ls->rowids.alloc((size_t)n_runs * 4));
        ls_rowids_kernel<P><<<gr, 256, 0, s>>>(rpv, nrows_view, ridx.as<int32_t>(), ls->rowids.as<int32_t>());
        CSRK_LAUNCH_CHECK();
    }
    CSRK_TRY(ls->vals.alloc((size_t)n_tiles * ACC_TILE * 8));
    CSRK_TRY(ls->idx.alloc((size_t)n_tiles * ACC_TILE * 4));
    ls_fill_kernel<P, VT><<<(unsigned)ceil_div(n_tiles * ACC_TILE, 256), 256, 0, s>>>(
        src, rpv, nrows_view, ci, vs, n_ent, n_tiles * ACC_TILE, slot_map, ls->vals.as<double>(), ls->idx.as<uint32_t>(), rp_len);
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(ls->tile_base.alloc((size_t)n_tiles * 4));
    ls_tilebase_kernel<P><<<(unsigned)ceil_div(n_tiles, 256), 256, 0, s>>>(
        rpv, nrows_view, ridx.as<int32_t>(), n_tiles, ls->tile_base.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(ls->carry_row.alloc((size_t)n_tiles * 4));
    CSRK_TRY(ls->carry_val.alloc((size_t)n_tiles * 8));
    int cus = 0;
    CSRK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->device));
    int64_t wgs = (int64_t)(cus > 0 ? cus : 256);
    CSRK_HIP(hipFuncSetAttribute((const void *)spmv_lstream_kernel<LS_PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(160 * 1024)));
    CSRK_HIP(hipFuncSetAttribute((const void *)spmv_lstream_kernel<LS_RND>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(160 * 1024)));
    CSRK_HIP(hipFuncSetAttribute((const void *)spmv_lstream_kernel<LS_PLAIN, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(160 * 1024)));
    CSRK_HIP(hipFuncSetAttribute((const void *)spmv_lstream_kernel<LS_RND, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(160 * 1024)));
    const int64_t need = ceil_div(n_tiles, LS_THREADS / WAVE);
    ls->grid = (unsigned)(wgs < need ? wgs : need);
    ls->n_tiles = n_tiles;
    ls->n_runs = n_runs;
    ls->n_out = n_out;
    CSRK_HIP(hipStreamSynchronize(s));      // ridx, dphys are freed on return; *phys is a host temporary
    ls->on = true;
    return CSRK_OK;
}

// ---- cold staging ----------------------------------------------------------------------------------------------
// A gather of x[col] that misses L2 moves a 128-B line over the fabric for 8 useful bytes, and on a power-law
// matrix the light stream's unpacked ("cold") columns nearly all miss: 1.4 of the 2.2 GB the kernel moved.  The
// whole SpMV runs at the fabric's rate, so those bytes are its time.  Instead, before each light-stream launch one
// pass copies the cold entries' x values into `xg`, in an order that is cheap on BOTH sides:
//   * the stream side reads xg[pos]; the positions of the cold entries of one workgroup round (LS_STAGE_TILES
//     consecutive tiles = the 16 wavefronts of a workgroup, one tile each) form one contiguous range of xg, so
//     every line of xg is fetched by one workgroup within one round and used completely;
//   * the copy side (ls_stage_kernel) walks the cold entries sorted by (column block, position): one workgroup per
//     block of columns, whose x window it holds in LDS (x crosses the fabric once, coalesced), and its
//     writes land in runs: inside a round the positions are ordered by column block, so the entries of one
//     (round, block) bucket are neighbours on both sides and neighbouring blocks fill neighbouring pieces of a line.
// A cold entry's index word then holds its position in xg instead of its column (flags unchanged) and the stream
// kernel is given xg as the base of its cold gathers: the kernel itself does not change, nor does any result bit.
// Tiles whose staged values share one contiguous range of xg ("round").  The copy pass pays per store transaction (~13 ps
// chip-wide; a (round, column block) bucket of several values is one transaction), the stream kernel per line its
// gathers pull into L1 (the round's range is shared by the wavefronts that process it together): measured on the
// headline matrix, copy + stream = 0.125 + 0.178 ms at 1 tile (tile-major), 0.071 + 0.203 at 8, 0.064 + 0.215 at 16,
// 0.055 + 0.229 at 32, 0.049 + 0.257 at 64.
#ifndef CSRK_STAGE_TILES
#define CSRK_STAGE_TILES 8
#endif
constexpr int LS_STAGE_TILES = CSRK_STAGE_TILES;     // tiles per staging round
constexpr int LS_STAGE_WMAX = 9984;                   // columns per block at most: a 78-KiB window of x in LDS, two per CU (one per CU with 156 KiB: 58 vs 46 us; three: 49)
constexpr int LS_STAGE_THREADS = 1024, LS_STAGE_IPT = 8;

__device__ __forceinline__ bool ls_is_cold(uint32_t ix) { return !(ix & LS_HOT_BIT) && (ix & LS_COL_MASK) != LS_PAD; }

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

// xg[a_dst[k]] = x[a_col[k]] for the entries of one column block: the block's window of x is copied into LDS with
// coalesced loads and the gathers are LDS reads (a gather that misses L1 costs the CU ~4 clocks per lane to pull its
// 128-B line in, wherever the line comes from: as L2 gathers this pass took 78 us for 9 * 10^6 entries).
// Workgroup i takes block (i % 8) * (blocks / 8) + i / 8: the workgroups of one XCD (i % 8) walk consecutive blocks,
// so the short runs that neighbouring blocks write into one line of xg meet in one L2 before they are written back.
__global__ __launch_bounds__(LS_STAGE_THREADS) void ls_stage_kernel(const double *__restrict__ x, int32_t ncols, int32_t W,
                                                                   const uint16_t *__restrict__ a_col,
                                                                   const int32_t *__restrict__ a_dst,
                                                                   const int32_t *__restrict__ blk_start, int32_t nblk,
                                                                   double *__restrict__ xg)
{
    extern __shared__ __align__(16) double s_x[];
    const int32_t per = (nblk + 7) / 8;
    const int32_t b = (int32_t)(blockIdx.x % 8) * per + (int32_t)(blockIdx.x / 8);
    if (b >= nblk) return;
    const int32_t k0 = blk_start[b], k1 = blk_start[b + 1];
    if (k0 == k1) return;
    const int64_t c0 = (int64_t)b * W;
    // the first batch of (column, position) pairs is requested before the window, so both are in flight together
    int32_t c[LS_STAGE_IPT], d[LS_STAGE_IPT];
#pragma unroll
    for (int q = 0; q < LS_STAGE_IPT; q++) {
        const int32_t k = k0 + (int32_t)threadIdx.x + q * LS_STAGE_THREADS;
        c[q] = k < k1 ? (int32_t)__builtin_nontemporal_load(a_col + k) : -1;
        d[q] = k < k1 ? __builtin_nontemporal_load(a_dst + k) : 0;
    }
    for (int i = threadIdx.x; i < W; i += LS_STAGE_THREADS) s_x[i] = c0 + i < ncols ? x[c0 + i] : 0.0;
    __syncthreads();
    for (int32_t kb = k0; kb < k1; kb += LS_STAGE_THREADS * LS_STAGE_IPT) {
        int32_t cn[LS_STAGE_IPT], dn[LS_STAGE_IPT];
        const int32_t kn = kb + LS_STAGE_THREADS * LS_STAGE_IPT;
        if (kn < k1) {
#pragma unroll
            for (int q = 0; q < LS_STAGE_IPT; q++) {
                const int32_t k = kn + (int32_t)threadIdx.x + q * LS_STAGE_THREADS;
                cn[q] = k < k1 ? (int32_t)__builtin_nontemporal_load(a_col + k) : -1;
                dn[q] = k < k1 ? __builtin_nontemporal_load(a_dst + k) : 0;
            }
        }
#pragma unroll
        for (int q = 0; q < LS_STAGE_IPT; q++)
            if (c[q] >= 0) xg[d[q]] = s_x[c[q]];      // (non-temporal stores: 0.299 instead of 0.073 ms -- the runs no longer merge in L2)
        if (kn < k1) {
#pragma unroll
            for (int q = 0; q < LS_STAGE_IPT; q++) c[q] = cn[q], d[q] = dn[q];
        }
    }
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

static int build_cold_stage(Matrix *m, LightStream *ls, const int32_t *hot_cols, int32_t n_hot, hipStream_t s)
{
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
    auto count_pass = [&](int nt, int64_t *max_round) -> int {
        stage_tiles = nt;
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
        nround = nround_ls + 1;      // + the virtual round of the packed columns
        nb = nround * nblk;
        if (nb > nb_max) return CSRK_ERR_INVALID;      // (cannot happen: a round holds at least LS_STAGE_TILES or NW tiles)
        CSRK_HIP(hipMemcpyAsync(d_rt0.p, h_rt0.data(), h_rt0.size() * 4, hipMemcpyHostToDevice, s));
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
        CSRK_HIP(hipMemcpyAsync(rs.data(), drs.p, (size_t)(nround_ls + 1) * 8, hipMemcpyDeviceToHost, s));
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
        for (int nt = LS_RND_MAXTILES / NW * NW; nt >= NW;) {
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
    CSRK_HIP(hipMemcpyAsync(ls->round_tile0.p, h_rt0.data(), h_rt0.size() * 4, hipMemcpyHostToDevice, s));
    CSRK_HIP(hipMemcpyAsync(ls->wg_round0.p, h_wr0.data(), h_wr0.size() * 4, hipMemcpyHostToDevice, s));
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
    CSRK_HIP(hipFuncSetAttribute((const void *)ls_stage_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LS_STAGE_WMAX * 8)));
    CSRK_HIP(hipStreamSynchronize(s));      // the temporaries are freed on return
    ls->n_cold = n_cold;
    ls->stage_tiles = stage_tiles;
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
        CSRK_HIP(hipMemcpyAsync(&n_nonempty, nz.as<int32_t>() + m->nrows, 4, hipMemcpyDeviceToHost, s));
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
        CSRK_HIP(hipMemcpyAsync(&n_segs, p->seg_off.as<int64_t>() + m->nrows, 8, hipMemcpyDeviceToHost, s));
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

// `launching`: the call is an SpMV launch (counts towards the lazy split), not a query.
// Caller holds m->mu.
static int get_plan_locked(Matrix *m, hipStream_t s, SpmvPlan **out, bool launching)
{
    if (launching) m->spmv_calls++;
    // The long-row split costs ~25 ms and 2.4 GB on the headline matrix and pays back ~1.4 ms per
    // SpMV, so it is built on the SECOND launch on a handle: the reference's CSR.mult_vec makes a
    // handle per call (csr/csr.py:582) and must not pay for a plan it uses once.  Forcing the split
    // (CSRK_SPMV_HEAVY_SPLIT=1) or profiling builds it at once.
    const char *env = getenv("CSRK_SPMV_HEAVY_SPLIT"), *env_hot = getenv("CSRK_SPMV_HOT"), *env_ls = getenv("CSRK_SPMV_STREAM");
    const bool eager = (env && env[0] == '1') || (env_hot && env_hot[0] == '1') || (env_ls && env_ls[0] == '1') || !launching;
    const bool want_split = eager || m->spmv_calls >= 2;
    if (m->spmv_plan && !m->spmv_plan->split_considered && want_split && m->spmv_plan->algo == CSRK_SPMV_MERGE &&
        !m->spmv_plan->profiling) {
        if (hipDeviceSynchronize() != hipSuccess) {
            set_error("device synchronisation failed: %s", hipGetErrorString(hipGetLastError()));
            return CSRK_ERR_HIP;
        }
        free_spmv_plan(m->spmv_plan);
        m->spmv_plan = nullptr;
    }
    if (!m->spmv_plan) {
        SpmvPlan *p = new (std::nothrow) SpmvPlan();
        CSRK_REQUIRE(p, "out of host memory");
        p->algo = m->spmv_algo == CSRK_SPMV_AUTO ? CSRK_SPMV_MERGE : m->spmv_algo;
        // Plans are built on the default stream and completed before use: their temporaries come from
        // the caching allocator, whose recycling is safe only in default-stream order.
        (void)s;
        int rc = m->ptr64 ? build_plan<int64_t>(m, p, nullptr, want_split) : build_plan<int32_t>(m, p, nullptr, want_split);
        if (rc == CSRK_OK && hipDeviceSynchronize() != hipSuccess) {
            set_error("SpMV plan construction failed: %s", hipGetErrorString(hipGetLastError()));
            rc = CSRK_ERR_HIP;
        }
        if (rc != CSRK_OK) {
            delete p;
            return rc;
        }
        m->spmv_plan = p;
    }
    *out = m->spmv_plan;
    return CSRK_OK;
}

static int get_plan(Matrix *m, hipStream_t s, SpmvPlan **out)
{
    std::lock_guard<std::mutex> lk(m->mu);
    return get_plan_locked(m, s, out, false);
}

template <class P, int VT>
static int launch_spmv(Matrix *m, SpmvPlan *p, const double *d_x, double *d_y, hipStream_t s, int part)
{
    const P *rp = (const P *)m->d_rowptrs;
    if (m->nrows == 0) return CSRK_OK;
    int algo = p->algo;
    if (algo == CSRK_SPMV_MERGE && m->nnz < 2) algo = CSRK_SPMV_SCALAR;   // the tile kernel's pair loads need >= 2 entries
    // part (csrk_spmv_device_part): bit 0 = the rows of the row-major path (every row gets a value: the rows cut out
    // for the tiers get 0.0), bit 1 = the tiers' rows (their reduces overwrite those zeros).  1 then 2 = 3.
    const bool do_light = (part & 1) != 0, do_heavy = (part & 2) != 0;
    if (algo != CSRK_SPMV_MERGE && !do_light) return CSRK_OK;      // no tiers outside the merge algorithm
    switch (algo) {
    case CSRK_SPMV_MERGE: {
        // epilogue jobs, one launch at the end
        EpiJobs epi;
        epi.n = 0;
        auto flush_epi = [&]() -> int {
            if (epi.n == 0) return CSRK_OK;
            unsigned g = 0;
            for (int i = 0; i < epi.n; i++) g += (unsigned)epi.j[i].blocks;
            spmv_epilogue_kernel<<<g, EPI_THREADS, 0, s>>>(epi, d_y);
            CSRK_LAUNCH_CHECK();
            epi.n = 0;
            return CSRK_OK;
        };
        auto add_fix = [&](const int32_t *cr, const double *cv, int64_t n, double *yy) -> int {
            if (n <= 0) return CSRK_OK;
            if (epi.n == 6) CSRK_TRY(flush_epi());
            EpiJob J = {};
            J.kind = 0;
            J.blocks = (int32_t)ceil_div(n, EPI_THREADS);
            J.carry_row = cr;
            J.carry_val = cv;
            J.n = n;
            J.fy = yy;
            epi.j[epi.n++] = J;
            return CSRK_OK;
        };
        auto add_red = [&](const double *part, const int32_t *rows, int32_t H, int32_t n_wg, int G, const int32_t *crp,
                           const int32_t *cidx, const double *cval) -> int {
            if (H <= 0) return CSRK_OK;
            if (epi.n == 6) CSRK_TRY(flush_epi());
            EpiJob J = {};
            J.kind = 1;
            J.blocks = (int32_t)ceil_div(H, WAVE * (EPI_THREADS / WAVE / G));
            J.partial = part;
            J.row_list = rows;
            J.H = H;
            J.n_wg = n_wg;
            J.G = G;
            J.crp = crp;
            J.cidx = cidx;
            J.cval = cval;
            epi.j[epi.n++] = J;
            return CSRK_OK;
        };
        // cold staging (ls_stage_kernel) fills xg and the pack; it runs first, so that the x it has just read is still in
        // the Infinity Cache when the accumulator kernel fetches its windows
        if (do_light && p->ls.on && p->ls.n_cold) {
            KernelTimer ks(p, s, 3);
            const unsigned gs = (unsigned)(ceil_div(p->ls.n_stage_blk, 8) * 8);
            ls_stage_kernel<<<gs, LS_STAGE_THREADS, (size_t)p->ls.stage_w * 8, s>>>(
                d_x, m->ncols, p->ls.stage_w, p->ls.a_col.as<uint16_t>(), p->ls.a_dst.as<int32_t>(),
                p->ls.blk_start.as<int32_t>(), p->ls.n_stage_blk, p->ls.xg.as<double>());
            ks.stop();
            CSRK_LAUNCH_CHECK();
        }
        if (do_light && p->n_hot && !(p->ls.on && p->ls.n_cold)) {      // (with cold staging the pack is filled by ls_stage_kernel)
            hot_pack_kernel<<<(unsigned)ceil_div(p->n_hot, 256), 256, 0, s>>>(d_x, p->hot_cols.as<int32_t>(), p->n_hot,
                                                                            p->xh.as<double>());
            CSRK_LAUNCH_CHECK();
        }
        if (do_heavy && p->n_heavy && !p->acc.empty()) {        // tier 0, accumulator form
            KernelTimer kh(p, s, 1);
            for (AccPanel *ap : p->acc) {
                spmv_acc_kernel<ACC_CB, ACC_THREADS><<<(unsigned)ap->n_wg, ACC_THREADS, ap->lds, s>>>(
                    ap->vals.as<double>(), ap->idx.as<uint16_t>(), ap->tile_row0.as<int32_t>(), d_x, m->ncols, ap->segs.as<AccSeg>(),
                    ap->wg_seg.as<int32_t>(), ap->nrow, ap->partial.as<double>());
                CSRK_LAUNCH_CHECK();
            }
            kh.stop();
        }
        if (do_heavy && p->n_heavy && p->tier1.on) {            // tier 1, pair form
            Panel *pn = &p->tier1;
            KernelTimer kh(p, s, 2);
#define PANEL_ARGS(PP)                                                                                              \
    pn->rp.as<PP>(), pn->ci.as<int32_t>(), pn->vs.as<double>(), d_x, m->ncols, pn->y.as<double>(),                    \
        pn->tile.as<PanelTile>(), pn->group.as<PanelGroup>(), pn->rows, pn->carry_row.as<int32_t>(),                  \
        pn->carry_val.as<double>(), pn->nnz
            const unsigned grid = (unsigned)pn->groups;
            if (pn->p64) spmv_panel_kernel<int64_t, PANEL_T1><<<grid, PANEL_T1, 0, s>>>(PANEL_ARGS(int64_t));
            else spmv_panel_kernel<int32_t, PANEL_T1><<<grid, PANEL_T1, 0, s>>>(PANEL_ARGS(int32_t));
#undef PANEL_ARGS
            kh.stop();
            CSRK_LAUNCH_CHECK();
            // (the pair kernel's carries are added by the tier's reduce: Panel::crp)
        }
        if (do_light) {
#define MERGE_ARGS_LIGHT(CI)                                                                                        \
    p->rp_light.as<P>(), CI, m->d_values, d_x, d_y, p->tile_row.as<int32_t>(), m->nrows, p->nnz_light,                \
        p->carry_row.as<int32_t>(), p->carry_val.as<double>(), p->tile_cut.as<int32_t>(), p->cut_pos.as<int64_t>(),  \
        p->cut_cum.as<int64_t>(), m->nnz
#define MERGE_ARGS_FULL(CI)                                                                                         \
    rp, CI, m->d_values, d_x, d_y, p->tile_row.as<int32_t>(), m->nrows, m->nnz, p->carry_row.as<int32_t>(),          \
        p->carry_val.as<double>(), nullptr, nullptr, nullptr, m->nnz
            const unsigned grid = (unsigned)p->n_tiles;
            if (p->ls.on) {
                KernelTimer kl(p, s);
                constexpr size_t ls_lds = ((size_t)LS_HOT_LDS + (size_t)(LS_THREADS / WAVE) * (ACC_TILE + 2)) * 8;
                const double *x_cold = d_x, *x_pack = p->xh.as<double>();
                if (p->ls.n_cold) {      // cold staging: the unpacked columns' x values, in the stream's order
                    x_cold = p->ls.xg.as<double>();
                    x_pack = x_cold + p->ls.n_cold;
                }
#define LS_ARGS                                                                                                       \
    p->ls.vals.as<double>(), p->ls.idx.as<uint32_t>(), p->ls.rowids.as<int32_t>(), p->ls.tile_base.as<int32_t>(),      \
        x_cold, x_pack, p->n_hot_lds, p->ls.n_tiles, p->ls.n_runs, p->ls.n_out, d_y,         \
        p->ls.carry_row.as<int32_t>(), p->ls.carry_val.as<double>()
                constexpr size_t rnd_lds = ((size_t)LS_RND_HOT + LS_RND_CAP + (size_t)(LS_THREADS / WAVE) * (ACC_TILE + 2)) * 8;
                const int32_t *nil = nullptr;
#define LS_GO(D)                                                                                                       \
    do {                                                                                                               \
        if (p->ls.n_cold && p->ls.round_start.p)                                                                       \
            spmv_lstream_kernel<LS_RND, D><<<p->ls.grid, LS_THREADS, rnd_lds, s>>>(                                    \
                LS_ARGS, p->ls.round_start.as<int32_t>(), p->ls.round_tile0.as<int32_t>(), p->ls.wg_round0.as<int32_t>()); \
        else                                                                                                           \
            spmv_lstream_kernel<LS_PLAIN, D><<<p->ls.grid, LS_THREADS, ls_lds, s>>>(LS_ARGS, nil, nil, nil);           \
    } while (0)
                if (p->ls.dense) LS_GO(true);
                else LS_GO(false);
#undef LS_GO
#undef LS_ARGS
                kl.stop();
                CSRK_LAUNCH_CHECK();
                if (p->n_heavy) {
                    CSRK_TRY(add_fix(p->ls.carry_row.as<int32_t>(), p->ls.carry_val.as<double>(), p->ls.n_tiles, d_y));
                } else {
                    spmv_merge_fixup_kernel<<<(unsigned)ceil_div(p->ls.n_tiles * WAVE, 256), 256, 0, s>>>(
                        p->ls.carry_row.as<int32_t>(), p->ls.carry_val.as<double>(), p->ls.n_tiles, d_y);
                    CSRK_LAUNCH_CHECK();
                }
            } else {
            KernelTimer kt(p, s);
            if (p->n_heavy)
                spmv_merge_kernel<P, VT, true><<<grid, MERGE_THREADS, 0, s>>>(MERGE_ARGS_LIGHT(m->d_colinds));
            else
                spmv_merge_kernel<P, VT, false><<<grid, MERGE_THREADS, 0, s>>>(MERGE_ARGS_FULL(m->d_colinds));
#undef MERGE_ARGS_LIGHT
#undef MERGE_ARGS_FULL
            kt.stop();
            CSRK_LAUNCH_CHECK();
            if (p->n_heavy)
                spmv_merge_fixup_short_kernel<<<(unsigned)ceil_div(p->n_tiles, 256), 256, 0, s>>>(
                    p->carry_row.as<int32_t>(), p->carry_val.as<double>(), p->n_tiles, d_y);
            else
                spmv_merge_fixup_kernel<<<(unsigned)ceil_div(p->n_tiles * WAVE, 256), 256, 0, s>>>(
                    p->carry_row.as<int32_t>(), p->carry_val.as<double>(), p->n_tiles, d_y);
            CSRK_LAUNCH_CHECK();
            }
        }
        if (do_heavy && p->n_heavy && !p->acc.empty())
            for (AccPanel *ap : p->acc)
                CSRK_TRY(add_red(ap->partial.as<double>(), ap->row_list.as<int32_t>(), ap->nrow, ap->n_wg, 4, nullptr, nullptr, nullptr));
        if (do_heavy && p->n_heavy && p->tier1.on) {
            // y[row] = sum over column blocks of the (block, row) partials, in block order, then the row's listed carries
            Panel *pn = &p->tier1;
            CSRK_TRY(add_red(pn->y.as<double>(), pn->row_list.as<int32_t>(), pn->nrow, pn->nb, pn->nb > 64 ? 16 : 2,
                             pn->crp.as<int32_t>(), pn->cidx.as<int32_t>(), pn->carry_val.as<double>()));
        }
        CSRK_TRY(flush_epi());      // the light stream's carries and the ordered reduces of the tiers, one launch
        break;
    }
    case CSRK_SPMV_VECTOR: {
        if (p->n_segs > 0) {
            KernelTimer kt(p, s);
            spmv_vector_kernel<P, VT><<<(unsigned)ceil_div(p->n_segs * WAVE, 256), 256, 0, s>>>(
                rp, m->d_colinds, m->d_values, d_x, d_y, p->seg_off.as<int64_t>(), p->seg_row.as<int32_t>(),
                p->n_segs, p->seg_part.as<double>());
            kt.stop();
            CSRK_LAUNCH_CHECK();
            spmv_vector_fixup_kernel<<<(unsigned)ceil_div(m->nrows, 256), 256, 0, s>>>(
                p->seg_off.as<int64_t>(), m->nrows, p->seg_part.as<double>(), d_y);
            CSRK_LAUNCH_CHECK();
        }
        break;
    }
    case CSRK_SPMV_SCALAR: {
        KernelTimer kt(p, s);
        spmv_scalar_kernel<P, VT><<<(unsigned)ceil_div(m->nrows, 256), 256, 0, s>>>(rp, m->d_colinds, m->d_values, d_x,
                                                                                  d_y, m->nrows);
        kt.stop();
        CSRK_LAUNCH_CHECK();
        break;
    }
    default:
        set_error("unknown spmv algo %d", p->algo);
        return CSRK_ERR_INVALID;
    }
    return CSRK_OK;
}

static int spmv_dispatch(Matrix *m, const double *d_x, double *d_y, hipStream_t s, int part = 3)
{
    // One SpMV = several kernels that share the plan's carry / partial arrays.  The per-handle lock keeps
    // the launch group together so that concurrent callers (the reference's kernels are nogil) are
    // ordered by the stream instead of interleaving.  Calls on one handle with DIFFERENT streams must
    // not overlap in time (same contract as any plan-owning library).
    std::lock_guard<std::mutex> lk(m->mu);
    SpmvPlan *p = nullptr;
    CSRK_TRY(get_plan_locked(m, s, &p, true));
    if (s) m->used_user_stream = true;
    // every n-th PRODUCT is timed: a product issued in two parts (part 1, then part 2) is one step, so part 2 keeps
    // the decision taken for its part 1 instead of counting as a call of its own
    if (part != 2) p->prof_this = p->profiling && (p->prof_calls++ % (p->prof_every > 0 ? p->prof_every : 1)) == 0;
#define GO(P, VT) return launch_spmv<P, VT>(m, p, d_x, d_y, s, part)
    if (m->ptr64) {
        if (m->val_type == CSRK_VAL_F64) GO(int64_t, CSRK_VAL_F64);
        if (m->val_type == CSRK_VAL_F32) GO(int64_t, CSRK_VAL_F32);
        GO(int64_t, CSRK_VAL_NONE);
    } else {
        if (m->val_type == CSRK_VAL_F64) GO(int32_t, CSRK_VAL_F64);
        if (m->val_type == CSRK_VAL_F32) GO(int32_t, CSRK_VAL_F32);
        GO(int32_t, CSRK_VAL_NONE);
    }
#undef GO
    return CSRK_ERR_INVALID;   // not reached
}

}  // namespace csrk

namespace csrk {

// float32 values times a float32 x: the reference's loop (csr/kernels/numba/__init__.py:55-67) is typed by Numba with a
// float32 product -- ONE rounding -- that is then added to the float64 accumulator.  One wavefront per row, lanes take the
// entries 64 apart, ordered tree over the lanes.  (A parity path: float32 matrices are the reference's test inputs, not the
// headline workload; every other dtype combination multiplies in float64, which the planned kernels do.)
template <class P>
__global__ __launch_bounds__(256) void spmv_f32x_kernel(const P *__restrict__ rp, const int32_t *__restrict__ ci,
                                                       const float *__restrict__ vs, const float *__restrict__ x, int32_t nrows,
                                                       double *__restrict__ y)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (i >= nrows) return;
    double acc = 0.0;
    for (int64_t k = (int64_t)rp[i] + lane, e = (int64_t)rp[i + 1]; k < e; k += WAVE) acc += (double)__fmul_rn(x[ci[k]], vs[k]);
    for (int off = WAVE / 2; off; off >>= 1) acc += __shfl_down(acc, off, WAVE);
    if (lane == 0) y[i] = acc;
}

__global__ void widen_f32_kernel(const float *__restrict__ in, double *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}

}  // namespace csrk

using namespace csrk;

extern "C" {

int csrk_spmv_device(csrk_handle_t h, const double *d_x, double *d_y, void *stream)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE((d_x || m->ncols == 0) && (d_y || m->nrows == 0), "x or y is NULL");
    return spmv_dispatch(m, d_x, d_y, (hipStream_t)stream);
}

int csrk_spmv_device_part(csrk_handle_t h, const double *d_x, double *d_y, void *stream, int part)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE((d_x || m->ncols == 0) && (d_y || m->nrows == 0), "x or y is NULL");
    CSRK_REQUIRE(part >= 1 && part <= 3, "part must be 1 (row-major path), 2 (tiers) or 3 (both), not %d", part);
    return spmv_dispatch(m, d_x, d_y, (hipStream_t)stream, part);
}

int csrk_spmv_cut_rows(csrk_handle_t h, int32_t *d_rows, int64_t capacity, int64_t *n_rows)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE(n_rows, "n_rows is NULL");
    SpmvPlan *p = nullptr;
    CSRK_TRY(get_plan(m, nullptr, &p));              // a query: builds the split eagerly
    std::lock_guard<std::mutex> lk(m->mu);
    const int64_t n = p->algo == CSRK_SPMV_MERGE ? p->n_heavy : 0;
    *n_rows = n;
    if (n == 0 || !d_rows) return CSRK_OK;
    CSRK_REQUIRE(capacity >= n, "the buffer holds %lld rows, the plan cut %lld out", (long long)capacity, (long long)n);
    CSRK_HIP(hipMemcpy(d_rows, p->heavy_row.p, (size_t)n * 4, hipMemcpyDeviceToDevice));
    return CSRK_OK;
}

int csrk_spmv(csrk_handle_t h, const double *x, double *y)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE((x || m->ncols == 0) && (y || m->nrows == 0), "x or y is NULL");
    if (m->nrows == 0) return CSRK_OK;
    // staging vectors come from the caching allocator per call, so concurrent calls on one handle
    // (the reference's kernels are nogil) do not share scratch space
    DevBuf dx, dy;
    CSRK_TRY(dx.alloc((size_t)m->ncols * 8));
    CSRK_TRY(dy.alloc((size_t)m->nrows * 8));
    if (m->ncols) CSRK_HIP(hipMemcpy(dx.p, x, (size_t)m->ncols * 8, hipMemcpyHostToDevice));
    CSRK_TRY(spmv_dispatch(m, dx.as<double>(), dy.as<double>(), nullptr));
    CSRK_HIP(hipMemcpy(y, dy.p, (size_t)m->nrows * 8, hipMemcpyDeviceToHost));
    return CSRK_OK;
}

int csrk_spmv_f32x(csrk_handle_t h, const float *x, double *y)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE((x || m->ncols == 0) && (y || m->nrows == 0), "x or y is NULL");
    if (m->nrows == 0) return CSRK_OK;
    DevBuf dx32, dx, dy;
    CSRK_TRY(dx32.alloc((size_t)m->ncols * 4 + 4));
    CSRK_TRY(dy.alloc((size_t)m->nrows * 8));
    if (m->ncols) CSRK_HIP(hipMemcpy(dx32.p, x, (size_t)m->ncols * 4, hipMemcpyHostToDevice));
    if (m->val_type == CSRK_VAL_F32) {
        // float32 x float32: the product is rounded to float32 (the reference's arithmetic), outside the planned kernels.
        // Reads the handle's arrays directly: under the handle's lock like every other product (unit_rows, center_rows
        // and order_columns rewrite them under it), held until the kernel has finished.
        std::lock_guard<std::mutex> lk(m->mu);
        const unsigned g = (unsigned)ceil_div((int64_t)m->nrows * WAVE, 256);
        if (m->ptr64)
            spmv_f32x_kernel<int64_t><<<g, 256>>>((const int64_t *)m->d_rowptrs, m->d_colinds, (const float *)m->d_values,
                                                  dx32.as<float>(), m->nrows, dy.as<double>());
        else
            spmv_f32x_kernel<int32_t><<<g, 256>>>((const int32_t *)m->d_rowptrs, m->d_colinds, (const float *)m->d_values,
                                                  dx32.as<float>(), m->nrows, dy.as<double>());
        CSRK_LAUNCH_CHECK();
        CSRK_HIP(hipDeviceSynchronize());
    } else {
        // float64 (or absent) values: Numba widens x, the product is float64 -- the usual kernels on the widened vector
        CSRK_TRY(dx.alloc((size_t)m->ncols * 8 + 8));
        if (m->ncols) {
            widen_f32_kernel<<<(unsigned)ceil_div(m->ncols, 256), 256>>>(dx32.as<float>(), dx.as<double>(), m->ncols);
            CSRK_LAUNCH_CHECK();
        }
        CSRK_TRY(spmv_dispatch(m, dx.as<double>(), dy.as<double>(), nullptr));
    }
    CSRK_HIP(hipMemcpy(y, dy.p, (size_t)m->nrows * 8, hipMemcpyDeviceToHost));
    return CSRK_OK;
}

int csrk_set_spmv_algo(csrk_handle_t h, int algo)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE(algo >= CSRK_SPMV_AUTO && algo <= CSRK_SPMV_SCALAR, "unknown spmv algo %d", algo);
    std::lock_guard<std::mutex> lk(m->mu);
    if (m->spmv_plan) {
        CSRK_HIP(hipDeviceSynchronize());
        free_spmv_plan(m->spmv_plan);
        m->spmv_plan = nullptr;
        if (m->spmm_plan) {               // it may hold a view of the SpMV plan's tier-0 panel
            free_spmm_plan(m->spmm_plan);
            m->spmm_plan = nullptr;
        }
    }
    m->spmv_algo = algo;
    m->spmv_calls = 0;
    return CSRK_OK;
}

const char *csrk_spmv_algo_name(csrk_handle_t h)
{
    Matrix *m = from_handle(h);
    if (!m) return "invalid";
    int a = m->spmv_plan ? m->spmv_plan->algo : m->spmv_algo;
    switch (a) {
    case CSRK_SPMV_MERGE: return "merge";
    case CSRK_SPMV_VECTOR: return "vector";
    case CSRK_SPMV_SCALAR: return "scalar";
    default: return "auto";
    }
}

int csrk_spmv_profile_begin(csrk_handle_t h, int max_records)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE(max_records > 0 && max_records <= 100000, "max_records out of range");
    SpmvPlan *p = nullptr;
    CSRK_TRY(get_plan(m, nullptr, &p));
    std::lock_guard<std::mutex> lk(m->mu);
    while ((int)p->ev.size() < 8 * max_records) {      // up to four timed kernels per launch
        hipEvent_t e;
        CSRK_HIP(hipEventCreate(&e));
        p->ev.push_back(e);
    }
    p->ev_chan.assign(p->ev.size() / 2, 0);
    p->ev_used = 0;
    p->prof_calls = 0;
    p->profiling = true;
    return CSRK_OK;
}

int csrk_spmv_profile_every(csrk_handle_t h, int every_n)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE(every_n >= 1, "every_n must be >= 1");
    SpmvPlan *p = nullptr;
    CSRK_TRY(get_plan(m, nullptr, &p));
    std::lock_guard<std::mutex> lk(m->mu);
    p->prof_every = every_n;
    return CSRK_OK;
}

int csrk_spmv_profile_channels(csrk_handle_t h, int mask)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE(mask > 0 && mask <= 0xf, "mask must name at least one of the four channels");
    SpmvPlan *p = nullptr;
    CSRK_TRY(get_plan(m, nullptr, &p));
    std::lock_guard<std::mutex> lk(m->mu);
    p->prof_mask = mask;
    return CSRK_OK;
}

int csrk_spmv_profile_end4(csrk_handle_t h, int *n_records, float *mean_ms)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE(n_records && mean_ms, "output is NULL");
    std::lock_guard<std::mutex> lk(m->mu);
    SpmvPlan *p = m->spmv_plan;
    CSRK_REQUIRE(p && p->profiling, "profiling was not started on this handle");
    p->profiling = false;
    double tot[4] = {0.0, 0.0, 0.0, 0.0};      // channel 3: the cold-staging pass
    int cnt[4] = {0, 0, 0, 0};
    int n = p->ev_used / 2;
    for (int i = 0; i < n; i++) {
        float ms = 0.f;
        CSRK_HIP(hipEventSynchronize(p->ev[2 * i + 1]));
        CSRK_HIP(hipEventElapsedTime(&ms, p->ev[2 * i], p->ev[2 * i + 1]));
        tot[p->ev_chan[i]] += ms;
        cnt[p->ev_chan[i]]++;
    }
    *n_records = 0;
    for (int c = 0; c < 4; c++) *n_records = cnt[c] > *n_records ? cnt[c] : *n_records;      // (launch groups timed)
    for (int c = 0; c < 4; c++) mean_ms[c] = cnt[c] ? (float)(tot[c] / cnt[c]) : 0.f;
    p->ev_used = 0;
    return CSRK_OK;
}

int csrk_spmv_profile_end(csrk_handle_t h, int *n_records, float *mean_ms)
{
    float ms4[4] = {0.f, 0.f, 0.f, 0.f};
    CSRK_REQUIRE(n_records && mean_ms, "output is NULL");
    CSRK_TRY(csrk_spmv_profile_end4(h, n_records, ms4));
    for (int c = 0; c < 3; c++) mean_ms[c] = ms4[c];
    return CSRK_OK;
}

int csrk_spmv_plan_stats(csrk_handle_t h, int64_t *out, int n)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE(out && n >= 0, "out is NULL");
    SpmvPlan *p = nullptr;
    CSRK_TRY(get_plan(m, nullptr, &p));
    const Panel &t1 = p->tier1;
    // tier 0: [4] tiles, [5] column blocks, [7] block width, [9] rows, [10] entries, [18] 1 (accumulator form)
    int64_t a_tiles = 0, a_rows = 0, a_nnz = 0, a_nb = 0;
    for (const AccPanel *ap : p->acc) {
        a_tiles += ap->tiles;
        a_rows += ap->nrow;
        a_nnz += ap->nnz;
        a_nb = ap->nb;
    }
    const bool af = !p->acc.empty();
    const int64_t plan_bytes = spmv_plan_bytes(p);      // [25]
    // [26] tiles per staging round (0: nothing staged); [27] workgroups of the accumulator kernel; [28] 0 (was: tier 1 on a side stream)
    const int64_t v[29] = {p->algo == CSRK_SPMV_MERGE ? p->n_tiles : p->n_segs,
                           p->algo == CSRK_SPMV_MERGE ? p->tile_items : VEC_SEG,
                           p->n_heavy, p->algo == CSRK_SPMV_MERGE ? p->nnz_light : m->nnz,
                           a_tiles, a_nb, p->heavy_min, af ? ACC_CB : 0, p->n_heavy ? 2 : 0, a_rows, a_nnz,
                           t1.nrow, t1.rows, t1.nnz, TIERB_MIN, t1.cb,
                           p->n_hot, (int64_t)(p->hot_cover * 1e6), af ? 1 : 0, p->hot_slots,
                           p->ls.on ? 1 : 0, p->ls.n_tiles, p->ls.n_runs, p->ls.grid, p->ls.n_cold, plan_bytes,
                           p->ls.round_start.p ? p->ls.stage_tiles : 0,
                           af ? (int64_t)p->acc[0]->n_wg : 0, 0};
    for (int i = 0; i < n && i < 29; i++) out[i] = v[i];
    return CSRK_OK;
}


int csrk_spmv_plan_info(csrk_handle_t h, int64_t *n_tiles, int32_t *tile_items)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    SpmvPlan *p = nullptr;
    CSRK_TRY(get_plan(m, nullptr, &p));
    if (n_tiles) *n_tiles = p->algo == CSRK_SPMV_MERGE ? p->n_tiles : p->n_segs;
    if (tile_items) *tile_items = p->algo == CSRK_SPMV_MERGE ? p->tile_items : VEC_SEG;
    return CSRK_OK;
}

}  // extern "C"
