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
// Sixteen lanes per row segment, four segments per wavefront (rows longer than 64 entries are split so the
// 10^5-entry rows of a power-law matrix do not serialise on one wave): a lane owns four consecutive panel
// columns (two 16-B loads per B row), so a wavefront has four DIFFERENT rows' gathers in flight -- the average
// row of the BASELINE matrix has 13 entries, and with one wavefront per row the kernel was a chain of dependent
// round trips (segment -> row pointer -> entries -> B rows -> store) that only wavefront count hid.  Segment
// partial panels are combined in segment order by a second kernel -- no float atomics, bitwise reproducible.
#include "common.h"

#include <algorithm>
#include <mutex>
#include <vector>

namespace csrk {

constexpr int MM_SEG = 64;       // entries per segment (light rows)
constexpr int MM_G = 16;         // lanes per unit (segment or row slice): 16 lanes x 4 columns = a 64-column chunk of the panel
constexpr int MM_CHUNK = 64;     // panel columns per pass over a unit's entries

// one segment of a light row: entries [start, start + n) of the CSR arrays; the result goes to row `row` of C
// (part < 0) or to partial panel `part` (split rows)
struct SegDesc {
    int64_t start;
    int32_t n;
    int32_t row;
    int64_t part;
};

struct SpmmPlan {
    Tier0View heavy;           // heavy rows come from the SpMV plan's column-block-major panel (or .on == false)
    // heavy rows: a row of L entries is cut into ceil(L / slice) parallel SLICES; slice p of P takes the p-th P-th of
    // every (column block, row) pair's entries, so all slices have about the same work in every block
    DevBuf slice;              // HeavySlice[n_slices]
    DevBuf slice_first;        // int32[n_rows + 1]: first slice of each heavy row
    int64_t n_slices = 0;
    int32_t n_wg = 0;          // persistent workgroups per XCD stream (CUs / 8)
    DevBuf hpart;              // double[n_slices * MM_STREAMS * k]: one partial panel per (slice, XCD stream)
    int32_t hpart_k = 0;
    int64_t n_segs = 0;
    int64_t n_multi = 0;       // segments belonging to split rows (need a partial panel)
    DevBuf part_off;           // int64[nrows + 1]: first partial slot of each split row
    DevBuf split_rows;         // int32[n_split]: rows with more than one segment
    int32_t n_split = 0;
    DevBuf seg_off;            // int64[nrows + 1]
    DevBuf seg;                // SegDesc[n_segs]
    DevBuf part;               // double[n_segs * k] (allocated on demand)
    int32_t part_k = 0;
    // opt-in (CSRK_SPMM_HOT=1): light rows with the hottest B rows resident in LDS (spmm_lseg_kernel) -- the columns the
    // light rows reference most, by popularity; the first n_hot of them (as many B rows as 128 KiB of LDS hold at the
    // panel width) are LDS slots, and ci_hot is a private copy of colinds in which an entry on slot s is stored as ~s
    std::vector<int32_t> hot_sorted;
    DevBuf hot_cols;           // int32[n_hot]
    DevBuf ci_hot;             // int32[nnz]
    int32_t n_hot = 0;         // slots ci_hot was encoded for (0: not built)
    bool hot_tried = false;
};

void free_spmm_plan(SpmmPlan *p) { delete p; }
int64_t spmm_plan_bytes(const SpmmPlan *p)
{
    int64_t b = 0;
    for (const DevBuf *d : {&p->slice, &p->slice_first, &p->hpart, &p->part_off, &p->split_rows, &p->seg_off, &p->seg, &p->part,
                            &p->hot_cols, &p->ci_hot})
        b += (int64_t)d->bytes;
    return b;
}

template <int VT>
__device__ __forceinline__ double mm_val(const void *v, int64_t k)
{
    if (VT == CSRK_VAL_F64) return ((const double *)v)[k];
    if (VT == CSRK_VAL_F32) return (double)((const float *)v)[k];
    return 1.0;
}


typedef double mm_f64x2 __attribute__((ext_vector_type(2)));
typedef mm_f64x2 MMF64x2 __attribute__((aligned(8)));

__device__ __forceinline__ int mm_wave_max4(int n)      // n is uniform inside each group of 16 lanes
{
    const int a = __builtin_amdgcn_readlane(n, 0), b = __builtin_amdgcn_readlane(n, 16);
    const int c = __builtin_amdgcn_readlane(n, 32), d = __builtin_amdgcn_readlane(n, 48);
    const int ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

// acc[i] += sum over the unit's entries [s, s + n) of val * B[col, c + i], i < 4, in storage order.  The 16 lanes of
// the unit load 16 entries' (col, val) with one coalesced load per array and walk them 4 at a time with the indices
// broadcast inside the group, so 4 B rows (x 4 units) are in flight per wavefront; `nmax` is the wavefront's longest
// unit (uniform trip counts keep the cross-lane broadcasts legal); entries past this unit's n re-read B row 0 / the
// unit's own rows and are selected away AFTER the multiply (0 * inf).  FULL4: k is a multiple of 4, so a live lane
// owns four whole columns and fetches them with two 16-B loads; otherwise column by column.
constexpr int MM_UNROLL = 4;
template <int VT, bool FULL4>
__device__ __forceinline__ void mm_unit(const int32_t *__restrict__ ci, const void *__restrict__ vs, int64_t s, int n,
                                        int nmax, const double *__restrict__ B, int64_t ldb, int32_t c, int32_t k,
                                        int sub, double acc[4])
{
    const int nv = c >= k ? 0 : (FULL4 ? 4 : (k - c < 4 ? k - c : 4));      // panel columns this lane owns
    for (int base = 0; base < nmax; base += MM_G) {
        const int idx = base + sub;
        const int32_t mycol = idx < n ? ci[s + idx] : 0;
        const double myval = idx < n ? mm_val<VT>(vs, s + idx) : 0.0;
        const int nb = nmax - base < MM_G ? nmax - base : MM_G;
        for (int j = 0; j < nb; j += MM_UNROLL) {
            double b[MM_UNROLL][4], av[MM_UNROLL];
#pragma unroll
            for (int u = 0; u < MM_UNROLL; u++) {
                const int32_t cu = __shfl(mycol, j + u, MM_G);
                av[u] = __shfl(myval, j + u, MM_G);
                const double *p = B + (int64_t)cu * ldb + c;
                if (FULL4) {
                    mm_f64x2 t0 = {0.0, 0.0}, t1 = {0.0, 0.0};
                    if (nv) {
                        t0 = *(const MMF64x2 *)p;
                        t1 = *(const MMF64x2 *)(p + 2);
                    }
                    b[u][0] = t0.x, b[u][1] = t0.y, b[u][2] = t1.x, b[u][3] = t1.y;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) b[u][i] = i < nv ? p[i] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < MM_UNROLL; u++) {
                const bool ok = base + j + u < n;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const double t = av[u] * b[u][i];
                    acc[i] += ok ? t : 0.0;                      // masked after the multiply (0 * inf)
                }
            }
        }
    }
}

template <bool FULL4>
__device__ __forceinline__ void mm_store4(double *__restrict__ dst, int32_t c, int32_t k, const double acc[4])
{
    if (c >= k) return;
    if (FULL4) {
        mm_f64x2 t0 = {acc[0], acc[1]}, t1 = {acc[2], acc[3]};
        // (non-temporal: panel rows are written once; kept out of L2 they leave more of it to the B rows: 2.61 -> 2.57 ms)
        __builtin_nontemporal_store(t0, (MMF64x2 *)(dst + c));
        __builtin_nontemporal_store(t1, (MMF64x2 *)(dst + c + 2));
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (c + i < k) dst[c + i] = acc[i];
    }
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
// blockIdx % 8 == block % 8; a speed assumption only).
//
// Block-synchronous slices.  The first version ran one wavefront per pair and wrote a partial panel per pair
// (1.5 * 10^6 pairs of ~16 entries on the BASELINE matrix: 0.75 GB written and read again, and the window shared L2
// with that stream: 73 % hit rate, the kernel bound by L1 line fills and partial traffic).  Now a heavy row of L
// entries is cut into ceil(L / slice) SLICES (slice p of P takes the p-th P-th of every pair of the row, so every slice
// has about the same number of entries in every block), and one persistent 1024-thread workgroup per CU -- 32 per XCD
// stream g -- holds MM_R slices per 16-lane unit with their four panel columns per lane in REGISTERS and walks the
// stream's blocks g, g + 8, ... with a workgroup barrier per block.  A workgroup's slices are a uniform sample of all
// slices (slice s goes to workgroup s mod 32), so the workgroups of a stream do the same amount of work per block to
// within a few per cent and move through the blocks together with no synchronisation between them: the B rows of the
// block they are at are what their XCD's L2 holds (a speed assumption only: any placement computes the same bits).
// One partial per (slice, stream) is written at the end and the reduce kernel adds a row's partials in order.
// (Units that walk the blocks on their own, without the barrier, drift apart by several blocks and lose the window:
// 22 % L2 hit rate, 12.5 GB of fabric reads -- measured; with it 91 % and 1.4 GB.)
constexpr int MM_STREAMS = 8;
constexpr int MM_R = 4;                      // slices per unit (their accumulators: 4 x 4 doubles per lane)
constexpr int MM_HEAVY_THREADS = 1024;
constexpr int MM_HEAVY_UNITS = MM_HEAVY_THREADS / MM_G;      // 64
#ifndef CSRK_MM_SLICE
#define CSRK_MM_SLICE 2048
#endif
constexpr int MM_SLICE = CSRK_MM_SLICE;     // smallest slice; doubled until the slices fit the persistent grid

struct HeavySlice {
    int32_t h;       // heavy-row index
    int32_t p, P;    // slice p of P
};

template <class PP, bool FULL4>
__global__ __launch_bounds__(MM_HEAVY_THREADS) void spmm_heavy_kernel(
    const PP *__restrict__ prp, const int32_t *__restrict__ pci, const double *__restrict__ pvs,
    const double *__restrict__ B, int32_t k, int64_t ldb, int32_t n_rows, int32_t n_blocks,
    const HeavySlice *__restrict__ slice, int64_t n_slices, int32_t n_wg, double *__restrict__ part)
{
    const int g = blockIdx.x % MM_STREAMS;
    const int wg = blockIdx.x / MM_STREAMS;
    const int unit = threadIdx.x / MM_G, sub = threadIdx.x & (MM_G - 1);
    // slice r of this unit: s = wg + n_wg * (unit + 64 r); (row, p, P) packed in two registers per slice
    const int64_t sid0 = wg + (int64_t)n_wg * unit, sstep = (int64_t)n_wg * MM_HEAVY_UNITS;
    int32_t sh[MM_R];
    uint32_t spP[MM_R];      // p in the low 16 bits, P in the high 16
#pragma unroll
    for (int r = 0; r < MM_R; r++) {
        sh[r] = 0, spP[r] = 1u << 16;
        if (sid0 + sstep * r < n_slices) {
            const HeavySlice t = slice[sid0 + sstep * r];
            sh[r] = t.h;
            spP[r] = (uint32_t)t.p | ((uint32_t)t.P << 16);
        }
    }
    for (int32_t c0 = 0; c0 < k; c0 += MM_CHUNK) {
        const int32_t c = c0 + 4 * sub;
        double acc[MM_R][4];
#pragma unroll
        for (int r = 0; r < MM_R; r++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[r][i] = 0.0;
        for (int64_t b = g; b < n_blocks; b += MM_STREAMS) {
            int32_t a0[MM_R], n0[MM_R];
#pragma unroll
            for (int r = 0; r < MM_R; r++) {                       // all pair ranges first: independent loads
                const int64_t q = b * n_rows + sh[r];
                const bool in = sid0 + sstep * r < n_slices;
                const int64_t t0 = in ? (int64_t)prp[q] : 0;
                n0[r] = in ? (int32_t)((int64_t)prp[q + 1] - t0) : 0;
                a0[r] = (int32_t)(t0 - (int64_t)prp[b * n_rows]);      // relative to the block's first entry (fits 32 bits)
            }
            const int64_t blk0 = (int64_t)prp[b * n_rows];
#pragma unroll
            for (int r = 0; r < MM_R; r++) {
                const int64_t p_ = spP[r] & 0xffffu, P_ = spP[r] >> 16;
                const int64_t lo = (int64_t)n0[r] * p_ / P_, hi = (int64_t)n0[r] * (p_ + 1) / P_;
                const int n = (int)(hi - lo);
                const int nmax = mm_wave_max4(n);
                if (nmax) mm_unit<CSRK_VAL_F64, FULL4>(pci, pvs, blk0 + a0[r] + lo, n, nmax, B, ldb, c, k, sub, acc[r]);
            }
            __syncthreads();      // the workgroup's 64 units enter the next block together
        }
#pragma unroll
        for (int r = 0; r < MM_R; r++)
            if (sid0 + sstep * r < n_slices)
                mm_store4<FULL4>(part + ((sid0 + sstep * r) * MM_STREAMS + g) * (int64_t)k, c, k, acc[r]);
    }
}

// C[row] = sum of the row's partials, slices in order, streams in order inside a slice: one wavefront per heavy row
__global__ __launch_bounds__(256) void spmm_heavy_reduce_kernel(const int32_t *__restrict__ slice_first,
                                                               const int32_t *__restrict__ row_list, int32_t n_rows,
                                                               int32_t k, const double *__restrict__ part,
                                                               double *__restrict__ C, int64_t ldc)
{
    const int64_t h = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (h >= n_rows) return;
    const int64_t r = row_list[h];
    const int64_t p0 = (int64_t)slice_first[h] * MM_STREAMS, p1 = (int64_t)slice_first[h + 1] * MM_STREAMS;
    for (int32_t c = lane; c < k; c += WAVE) {
        double acc = 0.0;
        int64_t q = p0;
        for (; q + 8 <= p1; q += 8) {                                 // 8 partial rows in flight, added in order
            double v[8];
#pragma unroll
            for (int t = 0; t < 8; t++) v[t] = part[(q + t) * (int64_t)k + c];
#pragma unroll
            for (int t = 0; t < 8; t++) acc += v[t];
        }
        for (; q < p1; q++) acc += part[q * (int64_t)k + c];
        C[r * ldc + c] = acc;
    }
}

// one thread per row: descriptors of its segments (plan time)
template <class P>
__global__ void mm_fill_kernel(const P *__restrict__ rp, const int64_t *__restrict__ seg_off, const int64_t *__restrict__ part_off,
                               int32_t nrows, SegDesc *__restrict__ seg)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const int64_t q0 = seg_off[r], nseg = seg_off[r + 1] - q0;
    const int64_t s = rp[r], e = rp[r + 1];
    for (int64_t i = 0; i < nseg; i++) {
        SegDesc d;
        d.start = s + i * MM_SEG;
        d.n = (int32_t)((e - d.start) < MM_SEG ? (e - d.start) : MM_SEG);
        d.row = (int32_t)r;
        d.part = nseg == 1 ? -1 : part_off[r] + i;
        seg[q0 + i] = d;
    }
}

// Sixteen lanes per row segment of the rows that are not served by the heavy-row kernels.
template <int VT, bool FULL4>
__global__ __launch_bounds__(256) void spmm_seg_kernel(const int32_t *__restrict__ ci, const void *__restrict__ vs,
                                                      const double *__restrict__ B, int32_t k, int64_t ldb,
                                                      double *__restrict__ C, int64_t ldc, const SegDesc *__restrict__ seg,
                                                      int64_t n_segs, double *__restrict__ part)
{
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / MM_G;
    const int sub = threadIdx.x & (MM_G - 1);
    SegDesc d;
    d.start = 0, d.n = 0, d.row = 0, d.part = -1;
    if (q < n_segs) d = seg[q];
    const int nmax = mm_wave_max4(d.n);
    double *dst = d.part < 0 ? C + (int64_t)d.row * ldc : part + d.part * (int64_t)k;
    for (int32_t c0 = 0; c0 < k; c0 += MM_CHUNK) {
        const int32_t c = c0 + 4 * sub;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        mm_unit<VT, FULL4>(ci, vs, d.start, d.n, nmax, B, ldb, c, k, sub, acc);
        if (q < n_segs) mm_store4<FULL4>(dst, c, k, acc);      // (an empty row's single segment stores its zeros)
    }
}

// ---- opt-in: light rows with the hottest B rows resident in LDS (north_star's "B tile staged in LDS") -----------------
// CSRK_SPMM_HOT=1.  Column popularity is as skewed as row length: on the BASELINE matrix the light rows put 29 % of their
// entries on the 256 most referenced columns.  One persistent 1024-thread workgroup per CU copies the n_hot most referenced
// B rows into LDS once (128 KiB: 256 rows at k = 64) and an entry on such a column reads its B row from there; the plan's
// private column array marks those entries (~slot), the padding lanes of a batch read slot 0 instead of B row 0.  Lane
// `sub` of a 16-lane unit owns panel columns {2 sub, 2 sub + 1} and {32 + 2 sub, 33 + 2 sub} of a 64-column chunk: each of
// its two 16-B loads then covers, with its 15 neighbours, 256 contiguous bytes -- all 64 LDS banks once -- so the lane
// groups ds_read_b128 is served in (which mix lanes of two units, i.e. two different rows) never clash when the row
// stride is a multiple of 256 B.  Sums in storage order, as in spmm_seg_kernel: same bits.
// MEASURED AND NOT THE DEFAULT: 1.61 ms against spmm_seg_kernel's 1.34 on the BASELINE matrix -- the
// rows it serves from LDS were L2 hits before, the cheapest line fills, and the persistent form runs 16 wavefronts per CU
// where the plain one runs 28.
constexpr int MM_L_THREADS = 1024;
constexpr int MM_L_UNITS = MM_L_THREADS / MM_G;
constexpr int MM_LDS_BYTES = 128 * 1024;
constexpr int MM_HOT_MAX = 8192;           // columns ranked at plan time (k = 2 fills the LDS budget with 8192 rows)

template <int VT>
__device__ __forceinline__ void mm_unit_h(const int32_t *__restrict__ ci, const void *__restrict__ vs, int64_t s, int n,
                                          int nmax, const double *__restrict__ B, int64_t ldb,
                                          const double *__restrict__ hotB, int32_t k, int32_t ca, int32_t cb, bool va,
                                          bool vb, int sub, double acc[4])
{
    for (int base = 0; base < nmax; base += MM_G) {
        const int idx = base + sub;
        const int32_t mycol = idx < n ? ci[s + idx] : -1;              // padding: LDS slot 0
        const double myval = idx < n ? mm_val<VT>(vs, s + idx) : 0.0;
        const int nb = nmax - base < MM_G ? nmax - base : MM_G;
        for (int j = 0; j < nb; j += MM_UNROLL) {
            mm_f64x2 ta[MM_UNROLL], tb[MM_UNROLL];
            double av[MM_UNROLL];
            int32_t w[MM_UNROLL];
            // the LDS reads first (short latency), then the global loads of the other entries into the SAME registers:
            // a global load with the hot lanes masked off leaves their LDS values in place, and the only wait between
            // the two groups is for the LDS.  (Either source chosen in one if / else per entry made the compiler wait for
            // every global load before the next entry's LDS read -- same destination registers, different return
            // queues: 2.08 ms instead of 1.61.)
#pragma unroll
            for (int u = 0; u < MM_UNROLL; u++) {
                w[u] = __shfl(mycol, j + u, MM_G);
                av[u] = __shfl(myval, j + u, MM_G);
                ta[u] = mm_f64x2{0.0, 0.0};
                tb[u] = mm_f64x2{0.0, 0.0};
                if (w[u] < 0) {
                    const double *p = hotB + (int64_t)(~w[u]) * k;
                    if (va) ta[u] = *(const mm_f64x2 *)(p + ca);
                    if (vb) tb[u] = *(const mm_f64x2 *)(p + cb);
                }
            }
#pragma unroll
            for (int u = 0; u < MM_UNROLL; u++) {
                if (w[u] >= 0) {
                    const double *p = B + (int64_t)w[u] * ldb;
                    if (va) ta[u] = *(const MMF64x2 *)(p + ca);
                    if (vb) tb[u] = *(const MMF64x2 *)(p + cb);
                }
            }
#pragma unroll
            for (int u = 0; u < MM_UNROLL; u++) {
                const bool ok = base + j + u < n;
                const double t0 = av[u] * ta[u].x, t1 = av[u] * ta[u].y, t2 = av[u] * tb[u].x, t3 = av[u] * tb[u].y;
                acc[0] += ok ? t0 : 0.0;                                // masked after the multiply (0 * inf)
                acc[1] += ok ? t1 : 0.0;
                acc[2] += ok ? t2 : 0.0;
                acc[3] += ok ? t3 : 0.0;
            }
        }
    }
}

// k even.  Dynamic LDS: n_hot * k doubles.
template <int VT>
__global__ __launch_bounds__(MM_L_THREADS) void spmm_lseg_kernel(const int32_t *__restrict__ ci_hot, const void *__restrict__ vs,
                                                                const double *__restrict__ B, int32_t k, int64_t ldb,
                                                                double *__restrict__ C, int64_t ldc,
                                                                const SegDesc *__restrict__ seg, int64_t n_segs,
                                                                double *__restrict__ part,
                                                                const int32_t *__restrict__ hot_cols, int32_t n_hot)
{
    extern __shared__ __align__(16) double mm_hotB[];
    const int32_t k2 = k / 2;
    for (int32_t i = threadIdx.x; i < n_hot * k2; i += MM_L_THREADS) {
        const int32_t r = i / k2, c = (i - r * k2) * 2;
        *(mm_f64x2 *)(mm_hotB + (int64_t)r * k + c) = *(const MMF64x2 *)(B + (int64_t)hot_cols[r] * ldb + c);
    }
    __syncthreads();
    const int unit = threadIdx.x / MM_G, sub = threadIdx.x & (MM_G - 1);
    const int64_t stride = (int64_t)gridDim.x * MM_L_UNITS;
    // a wavefront's four units take four consecutive segments: the trip count is uniform inside the wavefront
    for (int64_t q0 = (int64_t)blockIdx.x * MM_L_UNITS + (unit & ~3); q0 < n_segs; q0 += stride) {
        const int64_t q = q0 + (unit & 3);
        SegDesc d;
        d.start = 0, d.n = 0, d.row = 0, d.part = -1;
        if (q < n_segs) d = seg[q];
        const int nmax = mm_wave_max4(d.n);
        double *dst = d.part < 0 ? C + (int64_t)d.row * ldc : part + d.part * (int64_t)k;
        for (int32_t c0 = 0; c0 < k; c0 += MM_CHUNK) {
            const int32_t ca = c0 + 2 * sub, cb = c0 + 32 + 2 * sub;
            const bool va = ca < k, vb = cb < k;
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            mm_unit_h<VT>(ci_hot, vs, d.start, d.n, nmax, B, ldb, mm_hotB, k, ca, cb, va, vb, sub, acc);
            if (q < n_segs) {
                if (va) __builtin_nontemporal_store(mm_f64x2{acc[0], acc[1]}, (MMF64x2 *)(dst + ca));
                if (vb) __builtin_nontemporal_store(mm_f64x2{acc[2], acc[3]}, (MMF64x2 *)(dst + cb));
            }
        }
    }
}

// plan time: how often the light rows reference each column (sampled rows; rows the heavy kernels serve are skipped)
template <class P>
__global__ void mm_col_count_kernel(const P *__restrict__ rp, const int32_t *__restrict__ ci, int32_t nrows,
                                    int64_t row_stride, int32_t heavy_min, int32_t *__restrict__ cnt)
{
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / MM_G;
    const int sub = threadIdx.x & (MM_G - 1);
    const int64_t r = q * row_stride;
    if (r >= nrows) return;
    const int64_t s = rp[r], e = rp[r + 1];
    if (heavy_min > 0 && e - s >= heavy_min) return;
    for (int64_t i = s + sub; i < e; i += MM_G) atomicAdd(&cnt[ci[i]], 1);
}

__global__ void mm_slot_scatter_kernel(const int32_t *__restrict__ hot_cols, int32_t n_hot, int32_t *__restrict__ slot_of)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_hot) slot_of[hot_cols[i]] = i;
}

__global__ void mm_encode_hot_kernel(const int32_t *__restrict__ ci, int64_t nnz, const int32_t *__restrict__ slot_of,
                                     int32_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    const int32_t c = ci[i], sl = slot_of[c];
    out[i] = sl >= 0 ? ~sl : c;
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
    CSRK_TRY(p->seg.alloc((size_t)n * sizeof(SegDesc)));
    if (m->nrows > 0) {
        mm_fill_kernel<P><<<(unsigned)ceil_div(m->nrows, 256), 256, 0, s>>>((const P *)m->d_rowptrs, p->seg_off.as<int64_t>(),
                                                                           p->part_off.as<int64_t>(), m->nrows,
                                                                           p->seg.as<SegDesc>());
        CSRK_LAUNCH_CHECK();
    }
    return CSRK_OK;
}

// the slice table of the heavy rows (host: a few thousand rows)
template <class P>
__global__ void mm_heavy_len_kernel(const P *__restrict__ rp, const int32_t *__restrict__ row_list, int32_t n_rows,
                                    int64_t *__restrict__ len)
{
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h < n_rows) len[h] = (int64_t)rp[row_list[h] + 1] - (int64_t)rp[row_list[h]];
}

static int build_heavy_slices(Matrix *m, SpmmPlan *p, hipStream_t s)
{
    const Tier0View &hv = p->heavy;
    DevBuf dlen;
    CSRK_TRY(dlen.alloc((size_t)hv.n_rows * 8));
    const unsigned g = (unsigned)ceil_div(hv.n_rows, 256);
    if (m->ptr64)
        mm_heavy_len_kernel<int64_t><<<g, 256, 0, s>>>((const int64_t *)m->d_rowptrs, hv.row_list, hv.n_rows, dlen.as<int64_t>());
    else
        mm_heavy_len_kernel<int32_t><<<g, 256, 0, s>>>((const int32_t *)m->d_rowptrs, hv.row_list, hv.n_rows, dlen.as<int64_t>());
    CSRK_LAUNCH_CHECK();
    std::vector<int64_t> len((size_t)hv.n_rows);
    CSRK_HIP(hipMemcpyAsync(len.data(), dlen.p, (size_t)hv.n_rows * 8, hipMemcpyDeviceToHost, s));
    CSRK_HIP(hipStreamSynchronize(s));
    int cus = 0;
    CSRK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->device));
    p->n_wg = (cus > 0 ? cus : 256) / MM_STREAMS;
    if (p->n_wg < 1) p->n_wg = 1;
    const int64_t cap = (int64_t)p->n_wg * MM_HEAVY_UNITS * MM_R;      // slices the persistent grid holds per stream
    if (hv.n_rows > cap) {      // more heavy rows than accumulators: the pair panel is not used
        p->heavy.on = false;
        return CSRK_OK;
    }
    int64_t slice_len = MM_SLICE;
    for (;; slice_len *= 2) {
        int64_t cnt = 0, pmax = 0;
        for (int32_t h = 0; h < hv.n_rows; h++) {
            const int64_t P = len[(size_t)h] > slice_len ? ceil_div(len[(size_t)h], slice_len) : 1;
            cnt += P;
            pmax = std::max(pmax, P);
        }
        if ((cnt <= cap && pmax < 65536) || slice_len > (1ll << 40)) break;
    }
    std::vector<HeavySlice> sl;
    std::vector<int32_t> first((size_t)hv.n_rows + 1);
    for (int32_t h = 0; h < hv.n_rows; h++) {
        first[(size_t)h] = (int32_t)sl.size();
        const int32_t P = (int32_t)(len[(size_t)h] > slice_len ? ceil_div(len[(size_t)h], slice_len) : 1);
        for (int32_t q = 0; q < P; q++) sl.push_back(HeavySlice{h, q, P});
    }
    first[(size_t)hv.n_rows] = (int32_t)sl.size();
    p->n_slices = (int64_t)sl.size();
    CSRK_TRY(p->slice.alloc(sl.size() * sizeof(HeavySlice)));
    CSRK_TRY(p->slice_first.alloc(first.size() * 4));
    CSRK_HIP(hipMemcpyAsync(p->slice.p, sl.data(), sl.size() * sizeof(HeavySlice), hipMemcpyHostToDevice, s));
    CSRK_HIP(hipMemcpyAsync(p->slice_first.p, first.data(), first.size() * 4, hipMemcpyHostToDevice, s));
    CSRK_HIP(hipStreamSynchronize(s));      // `sl`, `first` are host temporaries
    return CSRK_OK;
}

// the columns the light rows reference most, by popularity (plan time: a sampled count on the device, the ranking of
// the few thousand candidates on the host)
static int build_hot_columns(Matrix *m, SpmmPlan *p)
{
    p->hot_tried = true;
    if (m->ncols <= 0 || m->nnz <= 0 || p->n_segs == 0) return CSRK_OK;
    DevBuf cnt;
    CSRK_TRY(cnt.alloc((size_t)m->ncols * 4));
    CSRK_HIP(hipMemsetAsync(cnt.p, 0, (size_t)m->ncols * 4, nullptr));
    const int64_t row_stride = m->nnz > (1ll << 25) ? m->nnz >> 25 : 1;
    const int64_t units = ceil_div((int64_t)m->nrows, row_stride);
    const unsigned g = (unsigned)ceil_div(units * MM_G, 256);
    const int32_t hmin = p->heavy.on ? p->heavy.min_entries : 0;
    if (m->ptr64)
        mm_col_count_kernel<int64_t><<<g, 256>>>((const int64_t *)m->d_rowptrs, m->d_colinds, m->nrows, row_stride, hmin, cnt.as<int32_t>());
    else
        mm_col_count_kernel<int32_t><<<g, 256>>>((const int32_t *)m->d_rowptrs, m->d_colinds, m->nrows, row_stride, hmin, cnt.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    std::vector<int32_t> c((size_t)m->ncols);
    CSRK_HIP(hipMemcpy(c.data(), cnt.p, (size_t)m->ncols * 4, hipMemcpyDeviceToHost));
    std::vector<int32_t> cand;
    for (int32_t j = 0; j < m->ncols; j++)
        if (c[(size_t)j] >= 2) cand.push_back(j);
    const size_t keep = std::min(cand.size(), (size_t)MM_HOT_MAX);
    auto hotter = [&](int32_t a, int32_t b) { return c[(size_t)a] != c[(size_t)b] ? c[(size_t)a] > c[(size_t)b] : a < b; };
    std::partial_sort(cand.begin(), cand.begin() + (std::ptrdiff_t)keep, cand.end(), hotter);
    cand.resize(keep);
    p->hot_sorted = std::move(cand);
    return CSRK_OK;
}

static int encode_hot_columns(Matrix *m, SpmmPlan *p, int32_t n_hot)
{
    DevBuf slot_of;
    CSRK_TRY(slot_of.alloc((size_t)m->ncols * 4));
    CSRK_HIP(hipMemsetAsync(slot_of.p, 0xff, (size_t)m->ncols * 4, nullptr));
    CSRK_HIP(hipDeviceSynchronize());      // earlier launches (any stream) may still read the old tables
    CSRK_TRY(p->hot_cols.alloc((size_t)n_hot * 4));
    CSRK_HIP(hipMemcpyAsync(p->hot_cols.p, p->hot_sorted.data(), (size_t)n_hot * 4, hipMemcpyHostToDevice, nullptr));
    if (!p->ci_hot.p) CSRK_TRY(p->ci_hot.alloc((size_t)m->nnz * 4));
    mm_slot_scatter_kernel<<<(unsigned)ceil_div(n_hot, 256), 256>>>(p->hot_cols.as<int32_t>(), n_hot, slot_of.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    mm_encode_hot_kernel<<<(unsigned)ceil_div(m->nnz, 256), 256>>>(m->d_colinds, m->nnz, slot_of.as<int32_t>(), p->ci_hot.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_HIP(hipDeviceSynchronize());      // slot_of is released here; the product may run on another stream
    p->n_hot = n_hot;
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
        if (hv.on) np->heavy = hv;
        // default stream + completion before use: see the caching allocator's contract (common.h)
        int rc = CSRK_OK;
        if (np->heavy.on) rc = build_heavy_slices(m, np, nullptr);      // (may turn the heavy path off: before the segment count)
        if (rc == CSRK_OK) rc = m->ptr64 ? build_mm_plan<int64_t>(m, np, nullptr) : build_mm_plan<int32_t>(m, np, nullptr);
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
        CSRK_TRY(p->hpart.alloc((size_t)p->n_slices * MM_STREAMS * k * 8));
        p->hpart_k = k;
    }
    if (grow_p) {           // some row is split: partial panels needed
        CSRK_TRY(p->part.alloc((size_t)p->n_multi * k * 8));
        p->part_k = k;
    }
    const bool full4 = (k % 4) == 0;
    if (p->heavy.on) {
        const Tier0View &hvw = p->heavy;
        const unsigned wgs = (unsigned)(MM_STREAMS * p->n_wg);
        const unsigned rgrid = (unsigned)ceil_div((int64_t)hvw.n_rows * WAVE, 256);
#define HEAVY(PP, F4)                                                                                                  \
    spmm_heavy_kernel<PP, F4><<<wgs, MM_HEAVY_THREADS, 0, s>>>((const PP *)hvw.rp, hvw.ci, hvw.vs, dB, k, ldb,         \
                                                              hvw.n_rows, hvw.n_blocks, p->slice.as<HeavySlice>(),     \
                                                              p->n_slices, p->n_wg, p->hpart.as<double>())
        if (hvw.p64) {
            if (full4) HEAVY(int64_t, true);
            else HEAVY(int64_t, false);
        } else {
            if (full4) HEAVY(int32_t, true);
            else HEAVY(int32_t, false);
        }
#undef HEAVY
        spmm_heavy_reduce_kernel<<<rgrid, 256, 0, s>>>(p->slice_first.as<int32_t>(), hvw.row_list, hvw.n_rows, k,
                                                      p->hpart.as<double>(), dC, ldc);
        CSRK_LAUNCH_CHECK();
    }
    if (p->n_segs == 0) return CSRK_OK;
    // opt-in (CSRK_SPMM_HOT=1; k even, >= 16 B rows fit 128 KiB): the light rows with the hottest B rows in LDS
    int32_t want_hot = 0;
    {
        const char *env = getenv("CSRK_SPMM_HOT");
        if (env && env[0] == '1' && k % 2 == 0 && (int64_t)k * 8 * 16 <= MM_LDS_BYTES) {
            if (!p->hot_tried) CSRK_TRY(build_hot_columns(m, p));
            want_hot = (int32_t)std::min<int64_t>((int64_t)p->hot_sorted.size(), MM_LDS_BYTES / ((int64_t)k * 8));
            if (want_hot < 16) want_hot = 0;
            if (want_hot && p->n_hot != want_hot) CSRK_TRY(encode_hot_columns(m, p, want_hot));
        }
    }
    if (want_hot) {
        static std::once_flag once;
        static hipError_t attr_err = hipSuccess;
        std::call_once(once, [] {
            for (const void *f : {(const void *)spmm_lseg_kernel<CSRK_VAL_F64>, (const void *)spmm_lseg_kernel<CSRK_VAL_F32>,
                                  (const void *)spmm_lseg_kernel<CSRK_VAL_NONE>}) {
                const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, MM_LDS_BYTES);
                if (e != hipSuccess) attr_err = e;
            }
        });
        CSRK_HIP(attr_err);
        int cus = 0;
        CSRK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->device));
        const unsigned lgrid = (unsigned)std::min<int64_t>(cus > 0 ? cus : 256, ceil_div(p->n_segs, MM_L_UNITS));
        const size_t lds = (size_t)want_hot * k * 8;
#define GOH(VT)                                                                                                        \
    spmm_lseg_kernel<VT><<<lgrid, MM_L_THREADS, lds, s>>>(p->ci_hot.as<int32_t>(), m->d_values, dB, k, ldb, dC, ldc,     \
                                                         p->seg.as<SegDesc>(), p->n_segs, p->part.as<double>(),        \
                                                         p->hot_cols.as<int32_t>(), want_hot)
        if (m->val_type == CSRK_VAL_F64) GOH(CSRK_VAL_F64);
        else if (m->val_type == CSRK_VAL_F32) GOH(CSRK_VAL_F32);
        else GOH(CSRK_VAL_NONE);
#undef GOH
    } else {
    const unsigned grid = (unsigned)ceil_div(p->n_segs * MM_G, 256);
#define GO(VT, F4)                                                                                                     \
    spmm_seg_kernel<VT, F4><<<grid, 256, 0, s>>>(m->d_colinds, m->d_values, dB, k, ldb, dC, ldc, p->seg.as<SegDesc>(),   \
                                                 p->n_segs, p->part.as<double>())
    if (m->val_type == CSRK_VAL_F64) {
        if (full4) GO(CSRK_VAL_F64, true);
        else GO(CSRK_VAL_F64, false);
    } else if (m->val_type == CSRK_VAL_F32) {
        if (full4) GO(CSRK_VAL_F32, true);
        else GO(CSRK_VAL_F32, false);
    } else {
        if (full4) GO(CSRK_VAL_NONE, true);
        else GO(CSRK_VAL_NONE, false);
    }
#undef GO
    }
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
