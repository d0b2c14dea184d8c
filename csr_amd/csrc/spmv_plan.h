// Shared between spmv.hip (the kernels a product runs, their launches, the C ABI) and spmv_plan.hip (what is built
// once per handle: the tiers, the hot-column pack, the light stream and its cold staging): the plan's structures, the
// tile / stream layout constants and the slot arithmetic both sides must agree on.
#pragma once
#include "common.h"
#include "wave.h"

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
    // The tiles' carries (static: a tile's carry belongs to the pair its last row end falls in) are added by the tier's
    // ordered reduce after the row's block partials, in tile order.  The j-th carry of long row h, j < ncs, is WRITTEN by its
    // tile into y[(nb + j) * nrow + h] -- `ncs` carry rows behind the nb block rows of the partials, -0.0 where a row has no
    // such carry (set once; x + -0.0 = x) -- so the reduce reads them like partials instead of chasing crp -> cidx -> carry_val
    // (three dependent round trips: 5.5 of the epilogue's 16 us).  Carries beyond ncs per row: carry_val[cidx[crp[h] ..
    // crp[h + 1])] as before (crp empty when there are none).
    int32_t ncs = 0;
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
    DevBuf z;              // double[nrow]: the rows' sums when the reduce rides in the pair kernel's launch (PanelRider)
    bool f32 = false;      // vals holds float32 (a float32 matrix: 6 B per entry instead of 10; widening them in the kernel is exact)
};

// A light stream (see "short rows: the light stream"): a private tiled copy of a set of rows ("runs") plus
// the tables the one-wavefront-per-tile kernel needs.
struct LightStream {
    bool on = false;
    int64_t n_tiles = 0;
    int32_t n_runs = 0, n_out = 0;      // non-empty runs; length of the output vector (rows, or pairs)
    unsigned grid = 0;
    DevBuf vals, idx, rowids, tile_base, carry_row, carry_val;
    bool idx24 = false;    // idx holds 3-byte words in two planes per tile (LS24_*: build_cold_stage), not uint32
    bool f32 = false;      // vals holds float32 (a float32 matrix: 8 B per entry instead of 12)
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
    int32_t tier1_min = 0;        // ... tier 1 the others down to this many
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
    DevBuf xwide;       // double[ncols]: a float32 x widened for the kernels that read the raw arrays (csrk_spmv_f32x*)
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

constexpr int MERGE_THREADS = 256;
constexpr int MERGE_IPT = 8;
constexpr int MERGE_ITEMS = MERGE_THREADS * MERGE_IPT;   // 2048 path items per tile
constexpr int MERGE_LONG = 64;                           // rows this long get a whole wave
constexpr int MERGE_MAXLONG = MERGE_ITEMS / MERGE_LONG + 2;

typedef int32_t i32x2_t __attribute__((ext_vector_type(2)));
typedef double f64x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef i32x2_t I32x2 __attribute__((aligned(4)));
typedef f64x2_t F64x2 __attribute__((aligned(4)));
typedef f32x2_t F32x2 __attribute__((aligned(4)));

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

// A tile's entries are stored in COLUMN order (panel_colsort_kernel): index word = {column - block start | position of
// the entry in the tile's row-major order: PANEL_POS_BITS bits}; inside a whole chunk of 128 ranks the even slots hold
// ranks 0..63, the odd slots ranks 64..127 (a lane's pair load then feeds two gathers of 64 consecutive ranks each).
constexpr int PANEL_POS_BITS = 11;
static_assert((1 << PANEL_POS_BITS) == MERGE_ITEMS, "a position inside a tile takes PANEL_POS_BITS bits");
static_assert(((int64_t)PANEL_CB1 << PANEL_POS_BITS) <= (1ll << 32), "the index word is 32 bits");
__host__ __device__ __forceinline__ int panel_slot_of_rank(int rank, int nn)
{
    const int chunk = rank >> 7, i = rank & 127;
    if (((chunk + 1) << 7) > nn) return rank;      // the tile's last, partial chunk: plain order
    return (chunk << 7) + (i < 64 ? 2 * i : 2 * (i - 64) + 1);
}

struct PanelTile {
    int64_t j0;      // first entry of the tile in M'
    int64_t cslot;   // where the tile's carry goes: an index into the partials' carry rows (Panel::ncs), or -1: carry_val[t]
    int32_t i0, i1;  // rows of M' completed before the tile start / end
    int32_t nn;      // entries in the tile
    int32_t blk;     // column block
};
constexpr int PANEL_CARRY_ROWS = 4;      // carry rows behind the block partials at most (a long row's first carries)

struct PanelGroup {
    int64_t t0;      // first tile
    int32_t nt;      // tiles handled by this workgroup (all in one column block)
    int32_t blk;
};
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
//   * At the end the workgroup stores its accumulators (H * 8 B) and the ordered reduce (spmv_epilogue_kernel, or the rider of tier 1's launch: PanelRider) sums the
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
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

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
// the same for a tile of values of type SV: float64 four 16-B loads per lane, float32 two
template <class SV> __host__ __device__ __forceinline__ int acc_slot_of(int e)
{
    return sizeof(SV) == 8 ? acc_val_slot(e) : acc_idx_slot(e);
}

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
// the same word in 3 bytes (LightStream::idx24, cold staging on: every column field is a pack slot or a round offset):
// hot bit 23, start bit 22, 22 bits of column; a tile = 512 low halves (a lane's eight: 16 B), then 512 high bytes (8 B)
#ifndef CSRK_LS_IDX24
#define CSRK_LS_IDX24 1
#endif
constexpr uint32_t LS24_HOT_BIT = 1u << 23, LS24_COL_MASK = (1u << 22) - 1;
constexpr int LS24_START_SHIFT = 22;
constexpr int64_t LS24_TILE_BYTES = 512 * 3;

constexpr int VEC_SEG = 4096;


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

// spmv_plan.hip: build the plan of `m` for p->algo (allow_split: with the long-row split, the pack and the streams --
// the lazy plan of the second product; else the tile coordinates only).  Default stream; the caller synchronises.
int build_spmv_plan(Matrix *m, SpmvPlan *p, bool allow_split);
// spmv.hip: the dynamic-LDS limits of the kernels a plan launches (per device: called by the builders)
int spmv_kernel_attributes();

}  // namespace csrk
