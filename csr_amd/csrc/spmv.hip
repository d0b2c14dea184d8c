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
//            tier 0  the (up to 15936) longest rows: column-block-major, x window AND one accumulator per
//                    row in LDS, 10 B per entry: f64 value + 16-bit (column, row step) word
//                    (spmv_acc_kernel, "long rows, accumulator form");
//            tier 1  rows of 128 .. tier-0 threshold: (column block, row) pairs over 2 MiB x windows kept in
//                    one XCD's L2, a tile's entries stored in column order and the products put back in row
//                    order through LDS (spmv_panel_kernel, "mid rows, pair form"); tier 0's ordered reduce
//                    rides in this kernel's launch (PanelRider);
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
#include "spmv_plan.h"

#include <algorithm>
#include <ctime>

namespace csrk {
// R32: float32 values times a float32 vector.  Numba types the reference's loop by its operands (csr/kernels/numba/
// __init__.py:55-67): the PRODUCT is float32, rounded once, then added to the float64 sum.  The plans hold the values
// widened to float64 and the vector is widened too, both exactly; their float64 product is exact (24 + 24 significant
// bits), so rounding it to float32 IS the float32 product.
template <bool R32>
__device__ __forceinline__ double spmv_prod(double a, double x)
{
    const double t = a * x;
    return R32 ? (double)(float)t : t;
}

void free_spmv_plan(SpmvPlan *p) { delete p; }

// device memory the plan holds (private streams, tables, scratch)
// by part: [0] tier 0 (accumulator stream, segment tables, partials), [1] tier 1 (pair panel, tiles, partials), [2] the light
// stream (values, index words, run tables), [3] cold staging (copy list, staged values, round tables) and the pack,
// [4] everything else (merge-path tables, the cut view, vector segments)
void spmv_plan_bytes_by_part(const SpmvPlan *p, int64_t out[5])
{
    for (int i = 0; i < 5; i++) out[i] = 0;
    auto add = [&](int part, std::initializer_list<const DevBuf *> bufs) {
        for (const DevBuf *b : bufs) out[part] += (int64_t)b->bytes;
    };
    for (const AccPanel *ap : p->acc) add(0, {&ap->row_list, &ap->vals, &ap->idx, &ap->tile_row0, &ap->segs, &ap->wg_seg, &ap->partial, &ap->z});
    const Panel &t = p->tier1;
    add(1, {&t.row_list, &t.rp, &t.ci, &t.vs, &t.tile, &t.group, &t.carry_row, &t.carry_val, &t.y, &t.crp, &t.cidx});
    const LightStream *l = &p->ls;
    add(2, {&l->vals, &l->idx, &l->rowids, &l->tile_base, &l->carry_row, &l->carry_val});
    add(3, {&l->xg, &l->a_col, &l->a_dst, &l->blk_start, &l->round_start, &l->round_tile0, &l->wg_round0, &p->hot_slot, &p->hot_cols, &p->xh});
    add(4, {&p->tile_row, &p->carry_row, &p->carry_val, &p->rp_light, &p->cut_pos, &p->cut_cum, &p->tile_cut, &p->heavy_row, &p->seg_off,
            &p->seg_row, &p->seg_part, &p->xwide});
}

int64_t spmv_plan_bytes(const SpmvPlan *p)
{
    int64_t part[5];
    spmv_plan_bytes_by_part(p, part);
    return part[0] + part[1] + part[2] + part[3] + part[4];
}


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
template <class P, int VT, bool HEAVY, bool R32 = false>
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
        const double t0 = spmv_prod<R32>(p0[u], x[c0[u]]), t1 = spmv_prod<R32>(p1[u], x[c1[u]]);
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

// XT: the type of the caller's x -- double, or float for csrk_spmv_f32x[_device]: the kernels that read x itself (the copy
// pass, the accumulator kernel's windows, the pair kernel's gathers, this pack) widen it as they load it, exactly; the
// light stream reads what the copy pass left (float64).
template <class XT>
__global__ void hot_pack_kernel(const XT *__restrict__ x, const int32_t *__restrict__ hot_cols, int32_t n_hot,
                                double *__restrict__ xh)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_hot) xh[i] = (double)x[hot_cols[i]];
}

// Tier 0's reduce as extra workgroups of the pair kernel's launch.  The accumulator kernel has finished when the pair kernel
// starts (stream order) and the pair kernel waits on L1 line fills, not on HBM: the 33 MB of partials are read beside
// it, the epilogue only scatters the 15 936 sums (z -> y[row_list]) instead of reading them.  Workgroups
// [first, first + ceil(H / 64)) of the launch; first = the launch's size when nothing rides along.
struct PanelRider {
    int64_t first;
    const double *partial;      // [n_wg][H]: the accumulator kernel's per-workgroup sums
    double *z;                  // [H]
    int32_t H, n_wg;
};

template <class PP, int PT, bool R32 = false, class XT = double>
__global__ __launch_bounds__(PT) void spmv_panel_kernel(
    const PP *__restrict__ prp, const int32_t *__restrict__ pci, const double *__restrict__ pvs,
    const XT *__restrict__ x, int32_t ncols, double *__restrict__ yp, const PanelTile *__restrict__ tiles,
    const PanelGroup *__restrict__ groups, int64_t n_prows, int32_t *__restrict__ carry_row,
    double *__restrict__ carry_val, int64_t pnnz, int32_t cb, PanelRider rider)
{
    // (a tile's entries are stored in column order; an index word = {column - block start | row-major position}: spmv_plan.h)
    constexpr uint32_t POS_MASK = (1u << PANEL_POS_BITS) - 1;
    __shared__ double s_buf[MERGE_ITEMS + 1];
    __shared__ int32_t s_long[MERGE_MAXLONG];
    __shared__ int32_t s_nlong;
    __shared__ double s_wpart[PT / WAVE];

    constexpr int PPAIRS = MERGE_ITEMS / PT / 2;      // consecutive pairs per lane
    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1), wv = tid / WAVE;
    if ((int64_t)blockIdx.x >= rider.first) {
        // Tier 0's ordered reduce rides along (PanelRider): this workgroup sums the accumulator kernel's partials of 64
        // heavy rows -- wavefront g the workgroups [g * per, (g + 1) * per), joined in order through LDS -- into z[h].
        constexpr int G = PT / WAVE;
        const int h = (int)((int64_t)blockIdx.x - rider.first) * WAVE + lane;
        const int per = (rider.n_wg + G - 1) / G;
        const int w0 = wv * per, w1 = w0 + per < rider.n_wg ? w0 + per : rider.n_wg;
        double acc = 0.0;
        if (h < rider.H) {
#pragma unroll 8
            for (int w = w0; w < w1; w++) acc += rider.partial[(int64_t)w * rider.H + h];
        }
        s_buf[wv * WAVE + lane] = acc;
        __syncthreads();
        if (wv == 0 && h < rider.H) {
            double tot = s_buf[lane];
            for (int u = 1; u < G; u++) tot += s_buf[u * WAVE + lane];
            rider.z[h] = tot;
        }
        return;
    }
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
        const XT *xw = x + (int64_t)pt.blk * cb;
#pragma unroll
        for (int u = 0; u < PPAIRS; u++) {
            const int k = 2 * (tid + u * PT);
            const double t0 = spmv_prod<R32>(p0[u], (double)xw[(uint32_t)c0[u] >> PANEL_POS_BITS]);
            const double t1 = spmv_prod<R32>(p1[u], (double)xw[(uint32_t)c1[u] >> PANEL_POS_BITS]);
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
                *(pt.cslot >= 0 ? yp + pt.cslot : carry_val + t) = tot;
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
            if (k < nn) s_prod[(uint32_t)c0[u] & POS_MASK] = p0[u];
            if (k + 1 < nn) s_prod[(uint32_t)c1[u] & POS_MASK] = p1[u];
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
                *(pt.cslot >= 0 ? yp + pt.cslot : carry_val + t) = acc;
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
                    *(pt.cslot >= 0 ? yp + pt.cslot : carry_val + t) = acc;
                }
            }
        }
        pt = nx;
    }
#undef PANEL_LOAD_TILE
}

// a lane's eight values of a tile: float64 four 16-B loads, float32 (a float32 matrix's stream) two, widened on use
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <class SV> struct AccVals;
template <> struct AccVals<double> {
    f64x2_t v[4];
    __device__ __forceinline__ void load(const double *tile, int lane)
    {
        const f64x2_t *vp = (const f64x2_t *)tile;
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = __builtin_nontemporal_load(vp + q * WAVE + lane);
    }
    __device__ __forceinline__ void get(double (&a)[ACC_K]) const
    {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            a[2 * q] = v[q].x;
            a[2 * q + 1] = v[q].y;
        }
    }
};
template <> struct AccVals<float> {
    f32x4_t v[2];
    __device__ __forceinline__ void load(const float *tile, int lane)
    {
        const f32x4_t *vp = (const f32x4_t *)tile;
#pragma unroll
        for (int q = 0; q < 2; q++) v[q] = __builtin_nontemporal_load(vp + q * WAVE + lane);
    }
    __device__ __forceinline__ void get(double (&a)[ACC_K]) const
    {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            a[4 * q] = (double)v[q].x;
            a[4 * q + 1] = (double)v[q].y;
            a[4 * q + 2] = (double)v[q].z;
            a[4 * q + 3] = (double)v[q].w;
        }
    }
};

template <int CB, int PT, bool R32 = false, class SV = double, class XT = double>
__global__ __launch_bounds__(PT) void spmv_acc_kernel(const SV *__restrict__ pvals, const uint16_t *__restrict__ pidx,
                                                     const int32_t *__restrict__ tile_row0,
                                                     const XT *__restrict__ x, int32_t ncols,
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
                for (int u = 0; u < WL; u++) {
                    if (sizeof(XT) == 8) {
                        v[u] = *((const F64x2 *)(x + w0) + tid + u * PT);
                    } else {
                        const f32x2_t t = *((const F32x2 *)(x + w0) + tid + u * PT);
                        v[u].x = (double)t.x;
                        v[u].y = (double)t.y;
                    }
                }
#pragma unroll
                for (int u = 0; u < WL; u++) ((f64x2_t *)s_x)[tid + u * PT] = v[u];
            } else {
                for (int k = tid; k < wlen; k += PT) s_x[k] = (double)x[w0 + k];
            }
        }
        __syncthreads();      // A: window stored; heads of the previous segment folded in; accumulators zeroed

        const int nt = sg.ntiles;
        AccVals<SV> v, vn;
        u32x4_t ix, ixn;                  // eight 16-bit index words per lane
        int32_t tr0 = 0, tr0n = 0;        // heavy-row index of the tile's first entry
        int t = wv;
        const int64_t pstep = (int64_t)gridDim.x;      // the workgroups' tiles are interleaved (acc_phys_tile)
        if (t < nt) {
            const u32x4_t *ip = (const u32x4_t *)(pidx + (sg.ptile0 + t * pstep) * ACC_TILE);
            v.load(pvals + (sg.ptile0 + t * pstep) * ACC_TILE, lane);
            ix = __builtin_nontemporal_load(ip + lane);
            tr0 = tile_row0[sg.tile0 + t];
        }
        for (; t < nt; t += NW) {
            const bool more = t + NW < nt;
            if (more) {      // next tile's loads are in flight while this one is reduced
                const u32x4_t *ip = (const u32x4_t *)(pidx + (sg.ptile0 + (t + NW) * pstep) * ACC_TILE);
                vn.load(pvals + (sg.ptile0 + (t + NW) * pstep) * ACC_TILE, lane);
                ixn = __builtin_nontemporal_load(ip + lane);
                tr0n = tile_row0[sg.tile0 + t + NW];
            }
            const uint32_t e[ACC_K] = {ix.x & 0xffffu, ix.x >> 16, ix.y & 0xffffu, ix.y >> 16,
                                       ix.z & 0xffffu, ix.z >> 16, ix.w & 0xffffu, ix.w >> 16};
            double a[ACC_K];
            v.get(a);
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
                if (R32) acc += spmv_prod<true>(a[j], xv[j]);
                else acc += a[j] * xv[j];
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
                v = vn;
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

// The per-SpMV epilogue -- the carry fix-up of the light stream, the ordered reduce of tier 1 (over the column blocks, plus
// the pair kernel's carries) and what is left of tier 0's -- is ONE launch covering up to six jobs (a job = a contiguous
// range of 1024-thread workgroups).
//   fix:     y[row] += the carries of the tiles that end inside `row`, in tile order (one thread per tile; the first tile of
//            a run of equal carry_row adds the whole run).
//   reduce:  y[row_list[h]] = sum over w < n_wg of partial[w][h], in order, then the `extra` carry rows partial[n_wg + j][h]
//            (Panel::ncs), then the row's listed carries (crp / cidx), in tile order.  A workgroup takes 1024 / (64 G) sets of
//            64 rows; the G wavefronts of a set each sum a contiguous range of w, joined in order through LDS.
//   scatter: y[row_list[h]] = partial[h] -- tier 0's sums when its reduce rode in the pair kernel's launch (PanelRider).
// What the kernel costs is its longest chain of dependent round trips, not its 55 MB (rocprofv3, headline matrix, round 6):
// 17.5 us with tier 0's reduce (63 workgroups of 256 partials per row: 17 us on its own) beside tier 1's whose listed carries
// were chased lane by lane (crp -> cidx -> carry_val per carry of the longest list: a few rows hold a carry per column
// block); 11.5 us with tier 0's reduce riding in the pair kernel's launch (+3.4 us there), the first carries of a row in
// carry rows behind the partials and the listed rest loaded a row at a time by the whole wavefront; 5 us is the floor of a
// launch that only scatters.  Measured and left out (profiles/r06_experiments.json): sixteen partials in flight per lane,
// row ids and carry ends requested at the top, G = 16 for tier 0 (25 us), 256-thread workgroups (12.5).
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
    int32_t extra;                  // rows of partial[] behind the n_wg: added last, in order (tier 1's carry rows, Panel::ncs)
    const int32_t *crp, *cidx;      // optional (nullptr: no listed carries to add)
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
    if (J.kind == 2) {      // scatter: the sums a rider of the pair kernel's launch left in `partial`
        const int h = b * EPI_THREADS + (int)threadIdx.x;
        if (h < J.H) y[J.row_list[h]] = J.partial[h];
        return;
    }
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    const int G = J.G, set = wv / G, g = wv % G;                      // G divides EPI_THREADS / 64
    const int h = (b * (EPI_THREADS / WAVE / G) + set) * WAVE + lane;
    const bool live = h < J.H;
    const int hc = live ? h : J.H - 1;
    const int per = (J.n_wg + G - 1) / G;
    const int w0 = g * per, w1 = w0 + per < J.n_wg ? w0 + per : J.n_wg;
    // the carry rows and the ends of the listed carries: requested with the partials (no branch around them: behind one
    // the compiler drains the queue before the loop below starts)
    double ex[PANEL_CARRY_ROWS];
#pragma unroll
    for (int j = 0; j < PANEL_CARRY_ROWS; j++) ex[j] = J.partial[(int64_t)(j < J.extra ? J.n_wg + j : 0) * J.H + hc];
    int32_t k0 = 0, k1 = 0;
    if (J.crp) {
        k0 = J.crp[hc];
        k1 = J.crp[hc + 1];
    }
    double acc = 0.0;
    if (live) {
#pragma unroll 8
        for (int w = w0; w < w1; w++) acc += J.partial[(int64_t)w * J.H + h];
    }
    s_p[wv][lane] = acc;
    __syncthreads();
    if (g != 0) return;
    double tot = s_p[wv][lane];
    for (int u = 1; u < G; u++) tot += s_p[wv + u][lane];
#pragma unroll
    for (int j = 0; j < PANEL_CARRY_ROWS; j++) tot += j < J.extra ? ex[j] : -0.0;
    // Listed carries.  Tile boundaries fall at about the same rows in every column block, so a few rows hold a carry per
    // block (dozens) and most hold none: lane by lane that was a chain of two dependent loads per carry of the longest
    // list.  The wavefront takes its rows that have some one after the other: a row's carries are loaded together, one per
    // lane, and added in list order.
    unsigned long long todo = __ballot(live && k1 > k0);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int32_t s0 = __shfl(k0, src, WAVE), s1 = __shfl(k1, src, WAVE);
        double T = __shfl(tot, src, WAVE);
        for (int32_t kb = s0; kb < s1; kb += WAVE) {
            const int32_t k = kb + lane;
            double c = -0.0;
            if (k < s1) c = J.cval[J.cidx[k]];
            const int cnt = s1 - kb < WAVE ? s1 - kb : WAVE;
            for (int u = 0; u < cnt; u++) T += __shfl(c, u, WAVE);
        }
        if (lane == src) tot = T;
    }
    if (live) y[J.row_list[h]] = tot;
}

// ---- short rows: the light stream ---------------------------------------------------------------------
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
// LS_RND24 = LS_RND with the index words stored in 3 bytes (LightStream::idx24): per tile a 16-bit plane (a lane's eight
// low halves: one 16-B load) and an 8-bit plane (its eight high bytes: one 8-B load) -- 11 B per entry instead of 12; the
// kernel's time follows its bytes (2 B per entry less, ablated: 161 -> 151 us).
constexpr int LS_PLAIN = 0, LS_RND = 2, LS_RND24 = 3;
template <int MODE, bool DENSE = false, bool R32 = false, class SV = double>
__global__ __launch_bounds__(LS_THREADS) void spmv_lstream_kernel(
    const SV *__restrict__ svals, const uint32_t *__restrict__ sidx, const int32_t *__restrict__ rowids,
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
    constexpr bool RND = MODE == LS_RND || MODE == LS_RND24;
    constexpr bool I24 = MODE == LS_RND24;
    constexpr uint32_t HOT_BIT = I24 ? LS24_HOT_BIT : LS_HOT_BIT, COL_MASK = I24 ? LS24_COL_MASK : LS_COL_MASK, PAD = COL_MASK;
    constexpr int START_SHIFT = I24 ? LS24_START_SHIFT : 30;
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
    AccVals<SV> v, vn;
    u32x4_t ix[2], ixn[2];
    // a lane's index words of tile t_: two 16-B loads, or (I24) the 16-B load of its low halves and the 8-B load of its high bytes
    auto idx_request = [&](u32x4_t (&w)[2], int64_t t_) {
        if constexpr (I24) {
            const unsigned char *tp = (const unsigned char *)sidx + t_ * LS24_TILE_BYTES;
            w[0] = __builtin_nontemporal_load((const u32x4_t *)tp + lane);
            const u32x2_t hi = __builtin_nontemporal_load((const u32x2_t *)(tp + ACC_TILE * 2) + lane);
            w[1].x = hi.x;
            w[1].y = hi.y;
        } else {
            const u32x4_t *ip = (const u32x4_t *)(sidx + t_ * ACC_TILE);
#pragma unroll
            for (int q = 0; q < 2; q++) w[q] = __builtin_nontemporal_load(ip + q * WAVE + lane);
        }
    };
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
        v.load(svals + t * ACC_TILE, lane);
        idx_request(ix, t);
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
        uint32_t e[ACC_K];
        if constexpr (I24) {
            // word j = {0, byte j of the high plane, half j of the low plane}: one v_perm_b32 each
            const uint32_t lo[4] = {ix[0].x, ix[0].y, ix[0].z, ix[0].w}, hi[2] = {ix[1].x, ix[1].y};
#pragma unroll
            for (int j = 0; j < ACC_K; j++)
                e[j] = __builtin_amdgcn_perm(hi[j >> 2], lo[j >> 1],
                                             0x0c000000u | ((4u + (j & 3)) << 16) | ((2u * (j & 1) + 1u) << 8) | (2u * (j & 1)));
        } else {
            const uint32_t w[ACC_K] = {ix[0].x, ix[0].y, ix[0].z, ix[0].w, ix[1].x, ix[1].y, ix[1].z, ix[1].w};
#pragma unroll
            for (int j = 0; j < ACC_K; j++) e[j] = w[j];
        }
        double a[ACC_K];
        v.get(a);
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
                const uint32_t c = e[j] & COL_MASK;
                const bool hot = (e[j] & HOT_BIT) != 0;
                inl[j] = c != PAD && (!hot || (int32_t)c < n_lds);      // (a padding slot multiplies 0 * 0)
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
            // (index words first: the next tile's gathers need them at its very top, the values only at its multiplies)
            idx_request(ixn, tn);
            vn.load(svals + tn * ACC_TILE, lane);
            tbn = tile_base[tn];
        }
        asm volatile("" ::: "memory");
        // row starts: bit j of st = entry j opens a row (index words only: this runs while the tile's x values arrive)
        uint32_t st = 0;
#pragma unroll
        for (int j = 0; j < ACC_K; j++) st |= ((e[j] >> START_SHIFT) & 1u) << j;
        const int cnt = __popc(st);
        const int S = wave_exscan_i32(cnt, lane);          // row starts in the lanes below
        const int total = wave_last_i32(S + cnt);            // row starts in the tile
        if (RND) {
#pragma unroll
            for (int j = 0; j < ACC_K; j++) {
                const uint32_t c = e[j] & COL_MASK;
                const bool cold = !(e[j] & HOT_BIT) && c != PAD;
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
            pr[j] = spmv_prod<R32>(a[j], inl[j] ? lv[j] : gv[j]);
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
        v = vn;
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
template <class P, int VT, bool R32 = false>
__global__ __launch_bounds__(256) void spmv_scalar_kernel(const P *__restrict__ rp, const int32_t *__restrict__ ci,
                                                         const void *__restrict__ vs, const double *__restrict__ x,
                                                         double *__restrict__ y, int32_t nrows)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    int64_t s = rp[r], e = rp[r + 1];
    double acc = 0.0;
    for (int64_t k = s; k < e; k++) acc += spmv_prod<R32>(ValLoad<VT>::at(vs, k), x[ci[k]]);
    y[r] = acc;
}

// xg[a_dst[k]] = x[a_col[k]] for the entries of one column block: the block's window of x is copied into LDS with
// coalesced loads and the gathers are LDS reads (a gather that misses L1 costs the CU ~4 clocks per lane to pull its
// 128-B line in, wherever the line comes from: as L2 gathers this pass took 78 us for 9 * 10^6 entries).
// Workgroup i takes block (i % 8) * (blocks / 8) + i / 8: the workgroups of one XCD (i % 8) walk consecutive blocks,
// so the short runs that neighbouring blocks write into one line of xg meet in one L2 before they are written back.
template <class XT>
__global__ __launch_bounds__(LS_STAGE_THREADS) void ls_stage_kernel(const XT *__restrict__ x, int32_t ncols, int32_t W,
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
    for (int i = threadIdx.x; i < W; i += LS_STAGE_THREADS) s_x[i] = c0 + i < ncols ? (double)x[c0 + i] : 0.0;
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

// the kernels that ask for more dynamic LDS than the default limit allows (per device: the builders call this)
int spmv_kernel_attributes()
{
    for (const void *f : {(const void *)spmv_acc_kernel<ACC_CB, ACC_THREADS>, (const void *)spmv_lstream_kernel<LS_PLAIN>,
                          (const void *)spmv_lstream_kernel<LS_RND>, (const void *)spmv_lstream_kernel<LS_PLAIN, true>,
                          (const void *)spmv_lstream_kernel<LS_RND, true>, (const void *)spmv_acc_kernel<ACC_CB, ACC_THREADS, true>,
                          (const void *)spmv_lstream_kernel<LS_PLAIN, false, true>, (const void *)spmv_lstream_kernel<LS_RND, false, true>,
                          (const void *)spmv_lstream_kernel<LS_PLAIN, true, true>, (const void *)spmv_lstream_kernel<LS_RND, true, true>,
                          (const void *)spmv_acc_kernel<ACC_CB, ACC_THREADS, false, float>,
                          (const void *)spmv_acc_kernel<ACC_CB, ACC_THREADS, true, float>,
                          (const void *)spmv_lstream_kernel<LS_PLAIN, false, false, float>, (const void *)spmv_lstream_kernel<LS_RND, false, false, float>,
                          (const void *)spmv_lstream_kernel<LS_PLAIN, true, false, float>, (const void *)spmv_lstream_kernel<LS_RND, true, false, float>,
                          (const void *)spmv_lstream_kernel<LS_PLAIN, false, true, float>, (const void *)spmv_lstream_kernel<LS_RND, false, true, float>,
                          (const void *)spmv_lstream_kernel<LS_PLAIN, true, true, float>, (const void *)spmv_lstream_kernel<LS_RND, true, true, float>,
                          (const void *)spmv_lstream_kernel<LS_RND24>, (const void *)spmv_lstream_kernel<LS_RND24, true>,
                          (const void *)spmv_lstream_kernel<LS_RND24, false, true>, (const void *)spmv_lstream_kernel<LS_RND24, true, true>,
                          (const void *)spmv_lstream_kernel<LS_RND24, false, false, float>, (const void *)spmv_lstream_kernel<LS_RND24, true, false, float>,
                          (const void *)spmv_lstream_kernel<LS_RND24, false, true, float>, (const void *)spmv_lstream_kernel<LS_RND24, true, true, float>,
                          (const void *)spmv_acc_kernel<ACC_CB, ACC_THREADS, false, double, float>,
                          (const void *)spmv_acc_kernel<ACC_CB, ACC_THREADS, true, double, float>,
                          (const void *)spmv_acc_kernel<ACC_CB, ACC_THREADS, false, float, float>,
                          (const void *)spmv_acc_kernel<ACC_CB, ACC_THREADS, true, float, float>})
        CSRK_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
    CSRK_HIP(hipFuncSetAttribute((const void *)ls_stage_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LS_STAGE_WMAX * 8)));
    CSRK_HIP(hipFuncSetAttribute((const void *)ls_stage_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LS_STAGE_WMAX * 8)));
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
        int rc = build_spmv_plan(m, p, want_split);
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

// XT = float: the caller's x is float32 and the plan serves every row without the raw-array kernels (spmv_dispatch checks)
template <class P, int VT, bool R32 = false, class XT = double>
static int launch_spmv(Matrix *m, SpmvPlan *p, const XT *d_x, double *d_y, hipStream_t s, int part)
{
    constexpr bool X64 = sizeof(XT) == 8;
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
                           const int32_t *cidx, const double *cval, int32_t extra = 0) -> int {
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
            J.extra = extra;
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
            ls_stage_kernel<XT><<<gs, LS_STAGE_THREADS, (size_t)p->ls.stage_w * 8, s>>>(
                d_x, m->ncols, p->ls.stage_w, p->ls.a_col.as<uint16_t>(), p->ls.a_dst.as<int32_t>(),
                p->ls.blk_start.as<int32_t>(), p->ls.n_stage_blk, p->ls.xg.as<double>());
            ks.stop();
            CSRK_LAUNCH_CHECK();
        }
        if (do_light && p->n_hot && !(p->ls.on && p->ls.n_cold)) {      // (with cold staging the pack is filled by ls_stage_kernel)
            hot_pack_kernel<XT><<<(unsigned)ceil_div(p->n_hot, 256), 256, 0, s>>>(d_x, p->hot_cols.as<int32_t>(), p->n_hot,
                                                                            p->xh.as<double>());
            CSRK_LAUNCH_CHECK();
        }
        bool t0_rides = false;      // tier 0's reduce ran inside the pair kernel's launch: the epilogue only scatters its sums
        if (do_heavy && p->n_heavy && !p->acc.empty()) {        // tier 0, accumulator form
            KernelTimer kh(p, s, 1);
            for (AccPanel *ap : p->acc) {
                if (ap->f32)
                    spmv_acc_kernel<ACC_CB, ACC_THREADS, R32, float, XT><<<(unsigned)ap->n_wg, ACC_THREADS, ap->lds, s>>>(
                        ap->vals.as<float>(), ap->idx.as<uint16_t>(), ap->tile_row0.as<int32_t>(), d_x, m->ncols,
                        ap->segs.as<AccSeg>(), ap->wg_seg.as<int32_t>(), ap->nrow, ap->partial.as<double>());
                else
                    spmv_acc_kernel<ACC_CB, ACC_THREADS, R32, double, XT><<<(unsigned)ap->n_wg, ACC_THREADS, ap->lds, s>>>(
                        ap->vals.as<double>(), ap->idx.as<uint16_t>(), ap->tile_row0.as<int32_t>(), d_x, m->ncols,
                        ap->segs.as<AccSeg>(), ap->wg_seg.as<int32_t>(), ap->nrow, ap->partial.as<double>());
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
        pn->carry_val.as<double>(), pn->nnz, pn->cb, rider
            PanelRider rider = {pn->groups, nullptr, nullptr, 0, 0};
            if (PANEL_T1 / WAVE == 4 && do_heavy && p->acc.size() == 1 && p->acc[0]->z.p) {
                AccPanel *ap = p->acc[0];
                rider.partial = ap->partial.as<double>();
                rider.z = ap->z.as<double>();
                rider.H = ap->nrow;
                rider.n_wg = ap->n_wg;
                t0_rides = true;
            }
            const unsigned grid = (unsigned)(pn->groups + (t0_rides ? ceil_div(rider.H, WAVE) : 0));
            if (pn->p64) spmv_panel_kernel<int64_t, PANEL_T1, R32, XT><<<grid, PANEL_T1, 0, s>>>(PANEL_ARGS(int64_t));
            else spmv_panel_kernel<int32_t, PANEL_T1, R32, XT><<<grid, PANEL_T1, 0, s>>>(PANEL_ARGS(int32_t));
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
                const double *x_cold = X64 ? (const double *)(const void *)d_x : nullptr, *x_pack = p->xh.as<double>();      // (float x: cold staging is on, spmv_dispatch checks)
                if (p->ls.n_cold) {      // cold staging: the unpacked columns' x values, in the stream's order
                    x_cold = p->ls.xg.as<double>();
                    x_pack = x_cold + p->ls.n_cold;
                }
#define LS_ARGS(SV)                                                                                                   \
    p->ls.vals.as<SV>(), p->ls.idx.as<uint32_t>(), p->ls.rowids.as<int32_t>(), p->ls.tile_base.as<int32_t>(),          \
        x_cold, x_pack, p->n_hot_lds, p->ls.n_tiles, p->ls.n_runs, p->ls.n_out, d_y,         \
        p->ls.carry_row.as<int32_t>(), p->ls.carry_val.as<double>()
                constexpr size_t rnd_lds = ((size_t)LS_RND_HOT + LS_RND_CAP + (size_t)(LS_THREADS / WAVE) * (ACC_TILE + 2)) * 8;
                const int32_t *nil = nullptr;
#define LS_GO(D, SV)                                                                                                   \
    do {                                                                                                               \
        if (p->ls.n_cold && p->ls.round_start.p && p->ls.idx24)                                                        \
            spmv_lstream_kernel<LS_RND24, D, R32, SV><<<p->ls.grid, LS_THREADS, rnd_lds, s>>>(                         \
                LS_ARGS(SV), p->ls.round_start.as<int32_t>(), p->ls.round_tile0.as<int32_t>(),                         \
                p->ls.wg_round0.as<int32_t>());                                                                        \
        else if (p->ls.n_cold && p->ls.round_start.p)                                                                  \
            spmv_lstream_kernel<LS_RND, D, R32, SV><<<p->ls.grid, LS_THREADS, rnd_lds, s>>>(                           \
                LS_ARGS(SV), p->ls.round_start.as<int32_t>(), p->ls.round_tile0.as<int32_t>(),                         \
                p->ls.wg_round0.as<int32_t>());                                                                        \
        else                                                                                                           \
            spmv_lstream_kernel<LS_PLAIN, D, R32, SV><<<p->ls.grid, LS_THREADS, ls_lds, s>>>(LS_ARGS(SV), nil, nil, nil); \
    } while (0)
                if (p->ls.f32) {
                    if (p->ls.dense) LS_GO(true, float);
                    else LS_GO(false, float);
                } else {
                    if (p->ls.dense) LS_GO(true, double);
                    else LS_GO(false, double);
                }
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
            } else if constexpr (X64) {
            KernelTimer kt(p, s);
            if (p->n_heavy)
                spmv_merge_kernel<P, VT, true, R32><<<grid, MERGE_THREADS, 0, s>>>(MERGE_ARGS_LIGHT(m->d_colinds));
            else
                spmv_merge_kernel<P, VT, false, R32><<<grid, MERGE_THREADS, 0, s>>>(MERGE_ARGS_FULL(m->d_colinds));
            kt.stop();
            CSRK_LAUNCH_CHECK();
            if (p->n_heavy)
                spmv_merge_fixup_short_kernel<<<(unsigned)ceil_div(p->n_tiles, 256), 256, 0, s>>>(
                    p->carry_row.as<int32_t>(), p->carry_val.as<double>(), p->n_tiles, d_y);
            else
                spmv_merge_fixup_kernel<<<(unsigned)ceil_div(p->n_tiles * WAVE, 256), 256, 0, s>>>(
                    p->carry_row.as<int32_t>(), p->carry_val.as<double>(), p->n_tiles, d_y);
            CSRK_LAUNCH_CHECK();
            } else {
                set_error("internal: a float32 vector reached the raw-array kernels");
                return CSRK_ERR_INVALID;
            }
#undef MERGE_ARGS_LIGHT
#undef MERGE_ARGS_FULL
        }
        if (do_heavy && p->n_heavy && !p->acc.empty() && t0_rides) {
            AccPanel *ap = p->acc[0];
            if (epi.n == 6) CSRK_TRY(flush_epi());
            EpiJob J = {};
            J.kind = 2;
            J.blocks = (int32_t)ceil_div(ap->nrow, EPI_THREADS);
            J.partial = ap->z.as<double>();
            J.row_list = ap->row_list.as<int32_t>();
            J.H = ap->nrow;
            epi.j[epi.n++] = J;
        } else if (do_heavy && p->n_heavy && !p->acc.empty())
            for (AccPanel *ap : p->acc)
                CSRK_TRY(add_red(ap->partial.as<double>(), ap->row_list.as<int32_t>(), ap->nrow, ap->n_wg, 4, nullptr, nullptr, nullptr));
        if (do_heavy && p->n_heavy && p->tier1.on) {
            // y[row] = sum over column blocks of the (block, row) partials, in block order, then the row's listed carries
            Panel *pn = &p->tier1;
            CSRK_TRY(add_red(pn->y.as<double>(), pn->row_list.as<int32_t>(), pn->nrow, pn->nb, pn->nb > 64 ? 4 : 2,
                             pn->crp.as<int32_t>(), pn->cidx.as<int32_t>(), pn->carry_val.as<double>(), pn->ncs));
        }
        CSRK_TRY(flush_epi());      // the light stream's carries and the ordered reduces of the tiers, one launch
        break;
    }
    case CSRK_SPMV_VECTOR: {
        if constexpr (!X64) {
            set_error("internal: a float32 vector reached the raw-array kernels");
            return CSRK_ERR_INVALID;
        } else
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
        if constexpr (!X64) {
            set_error("internal: a float32 vector reached the raw-array kernels");
            return CSRK_ERR_INVALID;
        } else {
        KernelTimer kt(p, s);
        // (R32: a float32 matrix of fewer than two entries under the merge algorithm lands here with its float32 products)
        spmv_scalar_kernel<P, VT, R32><<<(unsigned)ceil_div(m->nrows, 256), 256, 0, s>>>(rp, m->d_colinds, m->d_values, d_x,
                                                                                  d_y, m->nrows);
        kt.stop();
        CSRK_LAUNCH_CHECK();
        }
        break;
    }
    default:
        set_error("unknown spmv algo %d", p->algo);
        return CSRK_ERR_INVALID;
    }
    return CSRK_OK;
}

__global__ void widen_f32_kernel(const float *__restrict__ in, double *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}

// r32: float32 values times a (widened) float32 vector -- products rounded to float32 (spmv_prod); merge algorithm only
// d_x32: the caller's x is float32 (csrk_spmv_f32x[_device]).  When the plan serves every row from its own streams (tiers,
// light stream, cold staging: everything but a handle's first product) the kernels that read x widen it as they load it;
// otherwise it is widened into a buffer the plan keeps (no allocation and no synchronisation per call either way).
static int spmv_dispatch(Matrix *m, const double *d_x, double *d_y, hipStream_t s, int part = 3, bool r32 = false,
                         const float *d_x32 = nullptr)
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
    if (d_x32) {
        const bool streams_only = p->algo == CSRK_SPMV_MERGE && m->nnz >= 2 && p->ls.on && p->ls.n_cold > 0 && p->ls.round_start.p;
        if (streams_only) {
#define GOF(P, VT, R) return launch_spmv<P, VT, R, float>(m, p, d_x32, d_y, s, part)
            if (r32) {
                if (m->ptr64) GOF(int64_t, CSRK_VAL_F32, true);
                GOF(int32_t, CSRK_VAL_F32, true);
            }
            if (m->ptr64) {
                if (m->val_type == CSRK_VAL_F64) GOF(int64_t, CSRK_VAL_F64, false);
                if (m->val_type == CSRK_VAL_F32) GOF(int64_t, CSRK_VAL_F32, false);
                GOF(int64_t, CSRK_VAL_NONE, false);
            } else {
                if (m->val_type == CSRK_VAL_F64) GOF(int32_t, CSRK_VAL_F64, false);
                if (m->val_type == CSRK_VAL_F32) GOF(int32_t, CSRK_VAL_F32, false);
                GOF(int32_t, CSRK_VAL_NONE, false);
            }
#undef GOF
        }
        CSRK_TRY(p->xwide.ensure((size_t)m->ncols * 8 + 8));
        if (m->ncols) {
            widen_f32_kernel<<<(unsigned)ceil_div(m->ncols, 256), 256, 0, s>>>(d_x32, p->xwide.as<double>(), m->ncols);
            CSRK_LAUNCH_CHECK();
        }
        d_x = p->xwide.as<double>();
    }
#define GO(P, VT) return launch_spmv<P, VT>(m, p, d_x, d_y, s, part)
    if (r32) {
        CSRK_REQUIRE(m->val_type == CSRK_VAL_F32 && p->algo == CSRK_SPMV_MERGE, "float32 products need float32 values and the merge algorithm");
        if (m->ptr64) return launch_spmv<int64_t, CSRK_VAL_F32, true>(m, p, d_x, d_y, s, part);
        return launch_spmv<int32_t, CSRK_VAL_F32, true>(m, p, d_x, d_y, s, part);
    }
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
// headline workload; every other dtype combination multiplies in float64, which the planned kernels do.)  Serves the
// `vector` / `scalar` algorithms only: under `merge` (the default) the planned kernels round the products themselves (R32).
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
    DevBuf dx32, dy;
    CSRK_TRY(dx32.alloc((size_t)m->ncols * 4 + 4));
    CSRK_TRY(dy.alloc((size_t)m->nrows * 8));
    if (m->ncols) CSRK_HIP(hipMemcpy(dx32.p, x, (size_t)m->ncols * 4, hipMemcpyHostToDevice));
    const bool f32_values = m->val_type == CSRK_VAL_F32;
    const bool planned = m->spmv_algo == CSRK_SPMV_AUTO || m->spmv_algo == CSRK_SPMV_MERGE;
    if (f32_values && !planned) {
        // float32 x float32 under the `vector` / `scalar` baselines: a kernel of its own on the handle's arrays.
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
        // the usual kernels: float64 (or absent) values multiply in float64, as Numba types that loop; float32 values round
        // every product to float32 first (spmv_prod); x is widened as it is loaded, or once into the plan's buffer
        CSRK_TRY(spmv_dispatch(m, nullptr, dy.as<double>(), nullptr, 3, f32_values, dx32.as<float>()));
    }
    CSRK_HIP(hipMemcpy(y, dy.p, (size_t)m->nrows * 8, hipMemcpyDeviceToHost));
    return CSRK_OK;
}

int csrk_spmv_f32x_device(csrk_handle_t h, const float *d_x, double *d_y, void *stream)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE((d_x || m->ncols == 0) && (d_y || m->nrows == 0), "x or y is NULL");
    if (m->nrows == 0) return CSRK_OK;
    // (stream-ordered like csrk_spmv_device: nothing is allocated per call and nothing waited for)
    return spmv_dispatch(m, nullptr, d_y, (hipStream_t)stream, 3, m->val_type == CSRK_VAL_F32, d_x);
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
    // [26] tiles per staging round (0: nothing staged); [27] workgroups of the accumulator kernel; [28] 0 (was: tier 1 on a side
    // stream); [29..33] the plan's bytes by part (spmv_plan_bytes_by_part: tier 0, tier 1, light stream, staging + pack, rest)
    int64_t part[5];
    spmv_plan_bytes_by_part(p, part);
    const int64_t v[34] = {p->algo == CSRK_SPMV_MERGE ? p->n_tiles : p->n_segs,
                           p->algo == CSRK_SPMV_MERGE ? p->tile_items : VEC_SEG,
                           p->n_heavy, p->algo == CSRK_SPMV_MERGE ? p->nnz_light : m->nnz,
                           a_tiles, a_nb, p->heavy_min, af ? ACC_CB : 0, p->n_heavy ? 2 : 0, a_rows, a_nnz,
                           t1.nrow, t1.rows, t1.nnz, p->tier1_min, t1.cb,
                           p->n_hot, (int64_t)(p->hot_cover * 1e6), af ? 1 : 0, p->hot_slots,
                           p->ls.on ? 1 : 0, p->ls.n_tiles, p->ls.n_runs, p->ls.grid, p->ls.n_cold, plan_bytes,
                           p->ls.round_start.p ? p->ls.stage_tiles : 0,
                           af ? (int64_t)p->acc[0]->n_wg : 0, 0, part[0], part[1], part[2], part[3], part[4]};
    for (int i = 0; i < n && i < 34; i++) out[i] = v[i];
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
