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
#include <cstdio>
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

// ---- heavy rows: accumulators in registers, B tiles staged in LDS (north_star: "dense B tile staged in LDS") ----------
// The longest rows hold most of a power-law matrix (BASELINE matrix: 46 % of the entries in its 512 longest rows, 52 % in
// 1024): read row by row, each entry pulls a 512-B row of B into a CU, and nothing is reused.  Turned round: a
// persistent 1024-thread workgroup owns HR_ROWS = 512 heavy rows -- wavefront w holds rows 32 w .. 32 w + 31, one
// accumulator per (row, panel column) in REGISTERS (lane = panel column: 32 doubles = 64 VGPRs per lane) -- and sweeps a
// range of column TILES of HR_TILE = 128 rows of B (64 KiB at 64 panel columns), copied into LDS with coalesced 16-B loads,
// double-buffered (tile t + 1 is in flight while tile t is used).  The tile's entries on the workgroup's rows come from a
// private stream bucketed by (tile, row group, wavefront): {slot of the row in its wavefront, row of B inside the tile,
// value}; an entry is one ds_read_b64 per lane (the 64 lanes read one contiguous 512-B row: no bank conflicts) and one
// FMA into the accumulator of its row.  The row is wavefront-uniform, so that accumulator is a dynamically indexed
// REGISTER: s_set_gpr_idx_on + v_fma_f64 with relative destination (the accumulators are pinned to v[64:127]; the
// compiler alone puts a dynamically indexed array of this size into scratch memory).  What bounds the kernel is the
// number of instructions an entry needs -- a SIMD starts one vector and the CU one scalar instruction per clock -- not
// bytes: see spmm_hrows_kernel.
// With G row groups and R column ranges (G R = number of CUs; the G workgroups of a column range sit on ONE XCD and walk
// the same tiles at the same pace, so the tile a workgroup stages is usually in that XCD's L2) every staged B row serves
// all the entries the workgroup's 512 rows have on it (11 on average for the BASELINE matrix' first row group, 1.6 for
// its second) instead of one: G x 1 GB of coalesced tile loads instead of 512 B of 128-B line fills per entry.  Each workgroup writes the partial panel of its
// rows over its column range; the reduce kernel adds a row's R partials in range order (fixed order: bitwise
// reproducible; inside a tile a row's entries keep their storage order).
constexpr int HR_THREADS = 1024;
constexpr int HR_WAVES = HR_THREADS / WAVE;      // 16
constexpr int HR_RPW = 32;                       // rows per wavefront
constexpr int HR_ROWS = HR_WAVES * HR_RPW;       // 512 rows per workgroup
constexpr int HR_TILE = 128;                     // rows of B per tile
constexpr int HR_KC = 64;                        // panel columns per launch (wider panels: one launch per 64 columns)
constexpr int HR_PAD = 2 * WAVE;                 // the entry arrays' padding: a wavefront loads 64 entries from a bucket's start whatever its length
constexpr int HR_MAXG = 8;                       // at most 8 row groups (4096 heavy rows)
constexpr size_t HR_LDS = (size_t)2 * HR_TILE * HR_KC * 8;

struct SpmmPlan {
    // heavy rows (spmm_hrows_kernel)
    bool hr_on = false;
    int32_t hr_min = 0;        // rows with at least this many entries are heavy
    int32_t hr_n = 0;          // heavy rows
    int32_t hr_G = 0, hr_R = 0, hr_tiles = 0;
    int64_t hr_nnz = 0;
    DevBuf hr_rows;            // int32[hr_n]: row id of heavy row i
    DevBuf hr_code;            // int32[hr_n]: its place (group << 9 | wavefront << 5 | slot)
    DevBuf hr_bp;              // int64[hr_tiles * G * 16 + 1]: bucket bounds, bucket = (tile * G + group) * 16 + wavefront
    DevBuf hr_idx;             // uint32[hr_nnz + padding]: 2 slot | row of B inside the tile << 9
    DevBuf hr_vals;            // double[hr_nnz + padding]
    DevBuf hr_range;           // int32[R + 1]: first tile of each column range
    DevBuf hr_part;            // double[R * G * 512 * 64]
    int64_t n_segs = 0;
    int64_t n_multi = 0;       // segments belonging to split rows (need a partial panel)
    DevBuf part_off;           // int64[nrows + 1]: first partial slot of each split row
    DevBuf split_rows;         // int32[n_split]: rows with more than one segment
    int32_t n_split = 0;
    DevBuf seg_off;            // int64[nrows + 1]
    DevBuf seg;                // SegDesc[n_segs]
    DevBuf part;               // double[n_segs * k] (allocated on demand)
    int32_t part_k = 0;
};

void free_spmm_plan(SpmmPlan *p) { delete p; }
int64_t spmm_plan_bytes(const SpmmPlan *p)
{
    int64_t b = 0;
    for (const DevBuf *d : {&p->hr_rows, &p->hr_code, &p->hr_bp, &p->hr_idx, &p->hr_vals, &p->hr_range, &p->hr_part, &p->part_off,
                            &p->split_rows, &p->seg_off, &p->seg, &p->part})
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

// Where a row of the product goes.  Plain (csrk_spmm_dense): row i at C + i * ldc, columns ascending.  As the reference's
// mult_ab returns a product with a fully populated B (csrk_spgemm_ab's dense route, below): the rows of A without
// entries have no entries in C (row_base < 0: nothing stored), the others k each at C + row_base[i], columns k - 1 .. 0
// (rev_k = k: panel column c is stored at k - 1 - c).
struct SpmmOut {
    const int64_t *row_base;      // nullptr: i * ldc
    int32_t rev_k;                // 0: ascending columns
};
__device__ __forceinline__ double *mm_row_dst(double *C, int64_t ldc, int64_t row, const SpmmOut &o)
{
    if (!o.row_base) return C + row * ldc;
    const int64_t b = o.row_base[row];
    return b < 0 ? nullptr : C + b;
}

// the same four sums stored for a reversed row: column c + i at k - 1 - (c + i)
template <bool FULL4>
__device__ __forceinline__ void mm_store4_rev(double *__restrict__ dst, int32_t c, int32_t k, const double acc[4])
{
    if (c >= k) return;
    if (FULL4) {
        mm_f64x2 t0 = {acc[3], acc[2]}, t1 = {acc[1], acc[0]};
        __builtin_nontemporal_store(t0, (MMF64x2 *)(dst + k - 4 - c));
        __builtin_nontemporal_store(t1, (MMF64x2 *)(dst + k - 2 - c));
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (c + i < k) dst[k - 1 - c - i] = acc[i];
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

typedef double hr_d8 __attribute__((ext_vector_type(8)));

// acc[I / 2] += V * Bv for eight entries, the accumulators being v[64:127] (A0 .. A3): VGPR index mode with relative
// destination and addend, switched on ONCE for the eight (s_set_gpr_idx_idx moves the index; entering and leaving the
// mode per entry cost more than everything else the entry needs).  Nothing but the FMAs may execute in between -- the
// mode applies to every vector instruction -- hence one asm statement.  s_set_gpr_idx_* write M0[7:0] and M0[15:12]; the
// compiler does not accept M0 in a clobber list (reserved), so the statement saves and restores it itself (`m0_keep`: a
// scratch SGPR the caller declares).
#define HR_FMA8(I, V, Bv) \
    asm volatile("s_mov_b32 %[m0s], m0\n\t" \
                 "s_set_gpr_idx_on %[i0], 0xc\n\t" \
                 "v_fma_f64 v[64:65], %[v0], %[b0], v[64:65]\n\t" \
                 "s_set_gpr_idx_idx %[i1]\n\t" \
                 "v_fma_f64 v[64:65], %[v1], %[b1], v[64:65]\n\t" \
                 "s_set_gpr_idx_idx %[i2]\n\t" \
                 "v_fma_f64 v[64:65], %[v2], %[b2], v[64:65]\n\t" \
                 "s_set_gpr_idx_idx %[i3]\n\t" \
                 "v_fma_f64 v[64:65], %[v3], %[b3], v[64:65]\n\t" \
                 "s_set_gpr_idx_idx %[i4]\n\t" \
                 "v_fma_f64 v[64:65], %[v4], %[b4], v[64:65]\n\t" \
                 "s_set_gpr_idx_idx %[i5]\n\t" \
                 "v_fma_f64 v[64:65], %[v5], %[b5], v[64:65]\n\t" \
                 "s_set_gpr_idx_idx %[i6]\n\t" \
                 "v_fma_f64 v[64:65], %[v6], %[b6], v[64:65]\n\t" \
                 "s_set_gpr_idx_idx %[i7]\n\t" \
                 "v_fma_f64 v[64:65], %[v7], %[b7], v[64:65]\n\t" \
                 "s_set_gpr_idx_off\n\t" \
                 "s_mov_b32 m0, %[m0s]" \
                 : "+{v[64:79]}"(A0), "+{v[80:95]}"(A1), "+{v[96:111]}"(A2), "+{v[112:127]}"(A3), [m0s] "=&s"(m0_keep) \
                 : [i0] "s"(I[0]), [i1] "s"(I[1]), [i2] "s"(I[2]), [i3] "s"(I[3]), [i4] "s"(I[4]), [i5] "s"(I[5]), [i6] "s"(I[6]), [i7] "s"(I[7]), [v0] "s"(V[0]), [v1] "s"(V[1]), [v2] "s"(V[2]), [v3] "s"(V[3]), [v4] "s"(V[4]), [v5] "s"(V[5]), [v6] "s"(V[6]), [v7] "s"(V[7]), [b0] "v"(Bv[0]), [b1] "v"(Bv[1]), [b2] "v"(Bv[2]), [b3] "v"(Bv[3]), [b4] "v"(Bv[4]), [b5] "v"(Bv[5]), [b6] "v"(Bv[6]), [b7] "v"(Bv[7]))

// The first M (<= 64) entries of a chunk held one per lane in CI (entry words) / CV (values): batches of eight at
// CONSTANT lanes (the loop is unrolled: v_readlane takes the lane as an immediate, no scalar arithmetic for it).  A batch
// that is not full multiplies what is not its own -- valid entries of the next bucket, or the arrays' padding -- by 0.
#define HR_CHUNK(CI, CV, M)                                                                                            \
    do {                                                                                                               \
        const int32_t m_ = (M);                                                                                        \
        _Pragma("unroll") for (int k_ = 0; k_ < WAVE / HR_BATCH; k_++)                                                 \
        {                                                                                                              \
            if (k_ * HR_BATCH >= m_) break;                                                                            \
            uint32_t ix_[HR_BATCH];                                                                                    \
            double v_[HR_BATCH], b_[HR_BATCH];                                                                         \
            _Pragma("unroll") for (int u_ = 0; u_ < HR_BATCH; u_++)                                                    \
            {                                                                                                          \
                ix_[u_] = (uint32_t)__builtin_amdgcn_readlane((int)(CI), k_ * HR_BATCH + u_);                          \
                v_[u_] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(CV), k_ * HR_BATCH + u_),           \
                                          __builtin_amdgcn_readlane(__double2loint(CV), k_ * HR_BATCH + u_));          \
                b_[u_] = *(const double *)(cur + (ix_[u_] & 0xfe00u));                                                 \
            }                                                                                                          \
            if ((k_ + 1) * HR_BATCH > m_) {                                                                            \
                _Pragma("unroll") for (int u_ = 0; u_ < HR_BATCH; u_++)                                                \
                {                                                                                                      \
                    const bool mine_ = k_ * HR_BATCH + u_ < m_;                                                        \
                    b_[u_] = mine_ ? b_[u_] : 0.0;                                                                     \
                    v_[u_] = mine_ ? v_[u_] : 0.0;                                                                     \
                }                                                                                                      \
            }                                                                                                          \
            uint32_t m0_keep;                                                                                          \
            HR_FMA8(ix_, v_, b_);                                                                                      \
        }                                                                                                              \
    } while (0)

// rows [row0, row0 + HR_TILE) x columns [0, kc) of B into registers: thread p owns the 16-B pieces p, p + 1024, ...
// (piece = (row, column pair)).  A tile inside the matrix under a full 64-column panel (`whole`, uniform) is four plain
// loads at a scalar base + the thread's fixed offset `voff`.  Otherwise: branch-free, so that the loads stay in flight
// until hr_tile_store needs them: a piece past the matrix or the panel reads a valid address instead and is zeroed when
// it is stored.  EVEN: kc is even (a piece is inside the panel or outside it: one 16-B load); else two 8-B loads.
template <bool EVEN>
__device__ __forceinline__ void hr_tile_load(const double *__restrict__ B, int64_t ldb, int32_t ncols, int32_t kc, int64_t row0,
                                             int tid, uint32_t voff, bool whole, mm_f64x2 v[4])
{
    if (whole) {
        const char *sb = (const char *)(B + row0 * ldb);
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = *(const MMF64x2 *)(sb + (size_t)i * 32 * (size_t)ldb * 8 + voff);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int p = tid + i * HR_THREADS;
        const int64_t row = row0 + (p >> 5);
        const int c = (p & 31) * 2;
        const double *src = B + (row < ncols ? row : (int64_t)ncols - 1) * ldb;
        if (EVEN) {
            v[i] = *(const MMF64x2 *)(src + (c < kc ? c : 0));
        } else {
            v[i].x = src[c < kc ? c : 0];
            v[i].y = src[c + 1 < kc ? c + 1 : 0];
        }
    }
}

__device__ __forceinline__ void hr_tile_store(double *__restrict__ buf, int32_t ncols, int32_t kc, int64_t row0, int tid, bool whole,
                                              const mm_f64x2 v[4])
{
    if (whole) {
#pragma unroll
        for (int i = 0; i < 4; i++) *(mm_f64x2 *)(buf + (size_t)(tid + i * HR_THREADS) * 2) = v[i];
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int p = tid + i * HR_THREADS;
        const bool in = row0 + (p >> 5) < ncols;
        const int c = (p & 31) * 2;
        mm_f64x2 t;
        t.x = in && c < kc ? v[i].x : 0.0;
        t.y = in && c + 1 < kc ? v[i].y : 0.0;
        *(mm_f64x2 *)(buf + (size_t)p * 2) = t;
    }
}

constexpr int HR_BATCH = 8;       // entries per HR_FMA8

// Entry word: bits [7:0] = 2 x slot of the row in its wavefront (what s_set_gpr_idx_* take as the register index: they
// read only those bits), bits [15:9] = row of B inside the tile (so word & 0xfe00 = the row's byte offset in the tile).
// Per entry that leaves: three v_readlane (word, value), one s_and + v_or for the LDS address, the LDS read, one
// s_set_gpr_idx_idx and the FMA -- the CU's ONE scalar unit and the four instructions a SIMD can start per entry time are
// what bound this kernel, not bytes.
template <bool EVEN>
__global__ __launch_bounds__(HR_THREADS) void spmm_hrows_kernel(const int64_t *__restrict__ bp, const uint32_t *__restrict__ idx,
                                                               const double *__restrict__ vals, const double *__restrict__ B,
                                                               int32_t kc, int64_t ldb, int32_t ncols,
                                                               const int32_t *__restrict__ range_tile, int32_t G,
                                                               double *__restrict__ part)
{
    extern __shared__ __align__(16) double hr_lds[];      // two tiles of HR_TILE x 64 doubles
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> (column range r, row group g): the G groups of a range share an XCD (blockIdx % 8)
    const int xcd = blockIdx.x & 7, place = blockIdx.x >> 3;
    const bool by_xcd = ((gridDim.x / G) & 7) == 0;       // (a small matrix may have fewer tiles than that allows)
    const int r = by_xcd ? xcd + 8 * (place / G) : blockIdx.x / G, g = by_xcd ? place % G : blockIdx.x % G;
    const int32_t t0 = range_tile[r], t1 = range_tile[r + 1];
    hr_d8 A0 = 0.0, A1 = 0.0, A2 = 0.0, A3 = 0.0;
    mm_f64x2 pre[4];
    // The wavefront's entries of a tile are loaded a tile ahead, one entry per lane (a bucket holds ~30 entries on the
    // BASELINE matrix; longer buckets fetch their later chunks of 64 when they get there), and handed out with
    // v_readlane at constant lanes: vector loads have a counter of their own, so they stay in flight across the LDS
    // waits -- scalar loads share theirs with the LDS reads, and every batch then pays a full memory latency.
    const int64_t bstep = (int64_t)G * HR_WAVES;
    int64_t bk = ((int64_t)t0 * G + g) * HR_WAVES + w;
    int64_t e0 = bp[bk];
    int32_t n = (int32_t)(bp[bk + 1] - e0);
    uint32_t ci = idx[e0 + lane];
    double cv = vals[e0 + lane];
    int64_t e0n = 0;
    int32_t nn = 0;
    if (t0 + 1 < t1) {
        e0n = bp[bk + bstep];
        nn = (int32_t)(bp[bk + bstep + 1] - e0n);
    }
    // (thread's offset inside a tile of B: row tid / 32, column pair tid % 32; fits 32 bits when a tile's rows do)
    const bool can_whole = kc == HR_KC && (int64_t)HR_TILE * ldb * 8 < (1ll << 31);
    const uint32_t voff = (uint32_t)(((int64_t)(tid >> 5) * ldb + (tid & 31) * 2) * 8);
    {
        const bool whole = can_whole && (int64_t)(t0 + 1) * HR_TILE <= ncols;
        hr_tile_load<EVEN>(B, ldb, ncols, kc, (int64_t)t0 * HR_TILE, tid, voff, whole, pre);
        hr_tile_store(hr_lds, ncols, kc, (int64_t)t0 * HR_TILE, tid, whole, pre);
    }
    __syncthreads();
    // (the entries must have ARRIVED where the loop starts -- and again at its end, below -- or the compiler, which cannot
    // tell which loads a register still waits for across the back edge, waits for everything in flight before the first
    // v_readlane: the prefetches just issued)
    asm volatile("" : "+v"(ci), "+v"(cv));
    const char *my = (const char *)hr_lds + lane * 8;
    for (int32_t t = t0; t < t1; t++) {
        const char *cur = my + (size_t)((t - t0) & 1) * (HR_TILE * HR_KC * 8);
        const bool more = t + 1 < t1;
        const bool whole_next = can_whole && (int64_t)(t + 2) * HR_TILE <= ncols;
        uint32_t ni = 0;
        double nv = 0.0;
        int64_t e0nn = 0;
        int32_t nnn = 0;
        if (more) {      // in flight while tile t is used: the next tile of B, the next bucket's first chunk, the bounds after it
            hr_tile_load<EVEN>(B, ldb, ncols, kc, (int64_t)(t + 1) * HR_TILE, tid, voff, whole_next, pre);
            ni = idx[e0n + lane];
            nv = vals[e0n + lane];
            if (t + 2 < t1) {
                e0nn = bp[bk + 2 * bstep];
                nnn = (int32_t)(bp[bk + 2 * bstep + 1] - e0nn);
            }
        }
        HR_CHUNK(ci, cv, n < WAVE ? n : WAVE);      // the bucket's first 64 entries: in registers since the previous tile
        for (int32_t c0 = WAVE; c0 < n; c0 += WAVE) {      // a bucket of more than 64 entries: the later chunks on demand
            const uint32_t di = idx[e0 + c0 + lane];
            const double dv = vals[e0 + c0 + lane];
            HR_CHUNK(di, dv, n - c0 < WAVE ? n - c0 : WAVE);
        }
        if (more) hr_tile_store(hr_lds + (size_t)((t + 1 - t0) & 1) * (HR_TILE * HR_KC), ncols, kc, (int64_t)(t + 1) * HR_TILE, tid, whole_next, pre);
        __syncthreads();      // tile t + 1 is complete; nobody reads tile t any more
        bk += bstep;
        e0 = e0n, n = nn, ci = ni, cv = nv;
        e0n = e0nn, nn = nnn;
        asm volatile("" : "+v"(ci), "+v"(cv));
    }
    // the workgroup's partial panel: part[((r * G + g) * 512 + 32 w + slot) * 64 + lane]
    double *o = part + (((int64_t)r * G + g) * HR_ROWS + (int64_t)w * HR_RPW) * HR_KC + lane;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        __builtin_nontemporal_store(A0[i], o + (size_t)i * HR_KC);
        __builtin_nontemporal_store(A1[i], o + (size_t)(8 + i) * HR_KC);
        __builtin_nontemporal_store(A2[i], o + (size_t)(16 + i) * HR_KC);
        __builtin_nontemporal_store(A3[i], o + (size_t)(24 + i) * HR_KC);
    }
}

// C[row of heavy row i, c0 + c] = sum over the column ranges, in order, of the workgroups' partials: one wavefront per row
__global__ __launch_bounds__(256) void spmm_hrows_reduce_kernel(const int32_t *__restrict__ rows, const int32_t *__restrict__ code,
                                                               int32_t n, int32_t G, int32_t R, int32_t kc,
                                                               const double *__restrict__ part, double *__restrict__ C, int64_t ldc,
                                                               int32_t c0, SpmmOut om)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (i >= n || lane >= kc) return;
    const int32_t cd = code[i];
    const int64_t g = cd >> 9, in_wg = cd & (HR_ROWS - 1);
    const double *p = part + (g * HR_ROWS + in_wg) * HR_KC + lane;
    const int64_t step = (int64_t)G * HR_ROWS * HR_KC;
    double acc = 0.0;
    int32_t q = 0;
    for (; q + 8 <= R; q += 8) {                                      // 8 partial rows in flight, added in order
        double v[8];
#pragma unroll
        for (int t = 0; t < 8; t++) v[t] = p[(q + t) * step];
#pragma unroll
        for (int t = 0; t < 8; t++) acc += v[t];
    }
    for (; q < R; q++) acc += p[q * step];
    const int32_t c = c0 + lane;
    mm_row_dst(C, ldc, rows[i], om)[om.rev_k ? om.rev_k - 1 - c : c] = acc;      // (a heavy row has entries: it has a place)
}

// plan time: the entries of heavy row i as sortable records, in the order (i, storage order)
template <class P, int VT>
__global__ __launch_bounds__(256) void hr_emit_kernel(const P *__restrict__ rp, const int32_t *__restrict__ ci, const void *__restrict__ vs,
                                                     const int32_t *__restrict__ rows, const int32_t *__restrict__ code,
                                                     const int64_t *__restrict__ off, int32_t G, int32_t *__restrict__ key,
                                                     int32_t *__restrict__ payload, double *__restrict__ val)
{
    const int i = blockIdx.x;
    const int64_t s = (int64_t)rp[rows[i]], e = (int64_t)rp[rows[i] + 1], o = off[i];
    const int32_t cd = code[i];
    const int32_t g = cd >> 9, w = (cd >> 5) & 15, slot = cd & 31;
    for (int64_t q = s + threadIdx.x; q < e; q += 256) {
        const int32_t c = ci[q];
        key[o + q - s] = ((c / HR_TILE) * G + g) * HR_WAVES + w;
        payload[o + q - s] = (slot << 1) | ((c % HR_TILE) << 9);      // (see spmm_hrows_kernel)
        val[o + q - s] = mm_val<VT>(vs, q);
    }
}

template <class P>
__global__ void hr_len_kernel(const P *__restrict__ rp, int32_t nrows, int64_t *__restrict__ len)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nrows) len[r] = (int64_t)rp[r + 1] - (int64_t)rp[r];
}

__global__ void hr_tile_totals_kernel(const int64_t *__restrict__ bp, int32_t n_tiles, int32_t per_tile, int64_t *__restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t <= n_tiles) out[t] = bp[(int64_t)t * per_tile];
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
                                                      int64_t n_segs, double *__restrict__ part, SpmmOut om)
{
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / MM_G;
    const int sub = threadIdx.x & (MM_G - 1);
    SegDesc d;
    d.start = 0, d.n = 0, d.row = 0, d.part = -1;
    if (q < n_segs) d = seg[q];
    const int nmax = mm_wave_max4(d.n);
    const bool to_c = d.part < 0;
    double *dst = to_c ? mm_row_dst(C, ldc, d.row, om) : part + d.part * (int64_t)k;
    const bool rev = to_c && om.rev_k != 0;
    for (int32_t c0 = 0; c0 < k; c0 += MM_CHUNK) {
        const int32_t c = c0 + 4 * sub;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        mm_unit<VT, FULL4>(ci, vs, d.start, d.n, nmax, B, ldb, c, k, sub, acc);
        if (q < n_segs && dst) {      // (an empty row's single segment stores its zeros -- or nothing, where the row has no place)
            if (rev) mm_store4_rev<FULL4>(dst, c, k, acc);
            else mm_store4<FULL4>(dst, c, k, acc);
        }
    }
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
                                                        int64_t ldc, SpmmOut om)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (i >= n_split) return;
    const int64_t r = split_rows[i];
    const int64_t n = seg_off[r + 1] - seg_off[r];
    const int64_t a = part_off[r];
    double *dst = mm_row_dst(C, ldc, r, om);      // (a split row has entries: it has a place)
    for (int32_t c = lane; c < k; c += WAVE) {
        double acc = 0.0;
        for (int64_t q = a; q < a + n; q++) acc += part[q * (int64_t)k + c];
        dst[om.rev_k ? k - 1 - c : c] = acc;
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
                                                                            p->hr_on ? p->hr_min : 0);
        CSRK_LAUNCH_CHECK();
    }
    CSRK_TRY(exclusive_scan_i64(p->seg_off.as<int64_t>(), p->seg_off.as<int64_t>(), m->nrows, s));
    CSRK_TRY(exclusive_scan_i64(p->part_off.as<int64_t>(), p->part_off.as<int64_t>(), m->nrows, s));
    int64_t n = 0, nm = 0;
    CSRK_TRY(stage_d2h(&n, p->seg_off.as<int64_t>() + m->nrows, 8, s));
    CSRK_TRY(stage_d2h(&nm, p->part_off.as<int64_t>() + m->nrows, 8, s));
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
        CSRK_TRY(stage_d2h(&p->n_split, cnt.p, 4, s));
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

int stable_sort_records_f64(const int32_t *keys, const int32_t *payload, const double *vals, int64_t n, int32_t key_range,
                            int64_t payload_range, int64_t *out_ptr, int32_t *out_payload, double *out_vals, hipStream_t s);   // transpose.hip

// Choose the heavy rows and build their bucketed stream (plan time).  `force`: CSRK_SPMM_HEAVY=1 (small test matrices).
template <class P>
static int build_hrows(Matrix *m, SpmmPlan *p, int32_t k, bool force, hipStream_t s)
{
    p->hr_on = false;
    if (m->nrows < 1 || m->ncols < 1 || m->nnz < 1) return CSRK_OK;
    int cus = 0;
    CSRK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->device));
    // row lengths -> the threshold that leaves at most HR_MAXG * 512 rows (and, unless forced, pays for its sweeps of B)
    std::vector<int64_t> len((size_t)m->nrows);
    {
        DevBuf dlen;
        CSRK_TRY(dlen.alloc((size_t)m->nrows * 8));
        hr_len_kernel<P><<<(unsigned)ceil_div(m->nrows, 256), 256, 0, s>>>((const P *)m->d_rowptrs, m->nrows, dlen.as<int64_t>());
        CSRK_LAUNCH_CHECK();
        CSRK_TRY(stage_d2h(len.data(), dlen.p, (size_t)m->nrows * 8, s));
        CSRK_HIP(hipStreamSynchronize(s));
    }
    std::vector<int32_t> cand;
    const int64_t floor_len = force ? 256 : 512;                            // shorter rows never pay for a register slot
    for (int32_t r = 0; r < m->nrows; r++)
        if (len[(size_t)r] >= floor_len) cand.push_back(r);
    if (cand.empty()) return CSRK_OK;
    std::sort(cand.begin(), cand.end(), [&](int32_t a, int32_t b) { return len[(size_t)a] != len[(size_t)b] ? len[(size_t)a] > len[(size_t)b] : a < b; });
    // Every row group sweeps all of B once (G n_cols rows of B staged in all), tile by tile, whatever it finds there: a
    // group costs its tiles (staging, a barrier, a partial batch per wavefront) plus ~21 ps per entry, and saves the ~63 ps
    // per entry of the light-row kernel.  Measured on the BASELINE matrix (2M columns; groups of 23.0 / 3.2 / 1.8 / 1.2 M
    // entries): one group 2.22 ms for the product, two 2.12, four 2.13, eight 2.55.  Take row groups, longest rows first,
    // while the group's entries are at least 1.25 x the rows of B it stages.
    int G = 0;
    int64_t nnz_h = 0;
    size_t taken = 0;
    const char *genv = getenv("CSRK_SPMM_HEAVY_GROUPS");
    const int g_forced = genv ? atoi(genv) : 0;
    while (G < HR_MAXG && taken < cand.size()) {
        const size_t end = std::min(cand.size(), taken + (size_t)HR_ROWS);
        int64_t gn = 0;
        for (size_t c = taken; c < end; c++) gn += len[(size_t)cand[c]];
        const bool pays = gn * 4 >= 5 * (int64_t)m->ncols;
        if (getenv("CSRK_PLAN_TRACE"))
            fprintf(stderr, "[csrk spmm plan] heavy row group %d: rows %zu..%zu (lengths %lld..%lld), %lld entries, %s\n", G, taken, end,
                    (long long)len[(size_t)cand[taken]], (long long)len[(size_t)cand[end - 1]], (long long)gn, pays ? "pays" : "does not pay");
        if (g_forced > 0 ? G >= g_forced : (!pays && !(force && G == 0))) break;
        nnz_h += gn;
        taken = end;
        G++;
    }
    if (G == 0) return CSRK_OK;
    // the threshold is a row LENGTH (the light path skips rows by length): rows as long as the last one taken come along
    // when they fit, else the threshold moves up to the next length
    int64_t hmin = len[(size_t)cand[taken - 1]];
    size_t n = taken;
    while (n < cand.size() && len[(size_t)cand[n]] >= hmin) n++;
    if (n > (size_t)G * HR_ROWS) {
        hmin++;
        n = taken;
        while (n > 0 && len[(size_t)cand[n - 1]] < hmin) n--;
    }
    if (n == 0 || hmin > INT32_MAX) return CSRK_OK;
    cand.resize(n);
    nnz_h = 0;
    for (int32_t r : cand) nnz_h += len[(size_t)r];
    while (G > 1 && (size_t)(G - 1) * HR_ROWS >= n) G--;
    // G R workgroups, one per CU; a range needs a tile; with R a multiple of 8 the G workgroups of a column range sit
    // side by side on one XCD (spmm_hrows_kernel)
    const int64_t n_tiles = ceil_div((int64_t)m->ncols, HR_TILE);
    int R = (int)std::min<int64_t>(std::max(cus, G) / G, n_tiles);
    if (R >= 8) R = R / 8 * 8;
    const int64_t n_buckets = n_tiles * G * HR_WAVES;
    if (n_buckets >= (1ll << 30)) return CSRK_OK;                           // (keys are 32-bit)
    if (!force && n_buckets * 8 > nnz_h * 12) return CSRK_OK;               // the bucket table would outweigh the stream
    size_t mfree = 0, mtotal = 0;
    CSRK_HIP(hipMemGetInfo(&mfree, &mtotal));
    if ((size_t)nnz_h * 40 + (size_t)n_buckets * 8 + (64u << 20) > mfree) return CSRK_OK;
    // places: rank i (longest first) -> group i % G, then wavefront and slot round robin, so every wavefront of every
    // group holds the same mix of lengths
    std::vector<int32_t> code(n);
    std::vector<int64_t> off(n + 1);
    for (size_t i = 0; i < n; i++) {
        const int32_t g = (int32_t)(i % (size_t)G), j = (int32_t)(i / (size_t)G);
        code[i] = (g << 9) | ((j % HR_WAVES) << 5) | (j / HR_WAVES);
        off[i] = i ? off[i - 1] + len[(size_t)cand[i - 1]] : 0;
    }
    off[n] = off[n - 1] + len[(size_t)cand[n - 1]];
    DevBuf doff, key, pay, val;
    CSRK_TRY(p->hr_rows.alloc(n * 4));
    CSRK_TRY(p->hr_code.alloc(n * 4));
    CSRK_TRY(doff.alloc((n + 1) * 8));
    CSRK_TRY(key.alloc((size_t)nnz_h * 4));
    CSRK_TRY(pay.alloc((size_t)nnz_h * 4));
    CSRK_TRY(val.alloc((size_t)nnz_h * 8));
    CSRK_TRY(p->hr_bp.alloc((size_t)(n_buckets + 1) * 8));
    CSRK_TRY(p->hr_idx.alloc((size_t)(nnz_h + HR_PAD) * 4));
    CSRK_TRY(p->hr_vals.alloc((size_t)(nnz_h + HR_PAD) * 8));
    CSRK_HIP(hipMemsetAsync(p->hr_idx.as<uint32_t>() + nnz_h, 0, HR_PAD * 4, s));      // (the batches of four read a little past a bucket)
    CSRK_HIP(hipMemsetAsync(p->hr_vals.as<double>() + nnz_h, 0, HR_PAD * 8, s));
    CSRK_TRY(stage_h2d(p->hr_rows.p, cand.data(), n * 4, s));
    CSRK_TRY(stage_h2d(p->hr_code.p, code.data(), n * 4, s));
    CSRK_TRY(stage_h2d(doff.p, off.data(), (n + 1) * 8, s));
#define EMIT(VT)                                                                                                       \
    hr_emit_kernel<P, VT><<<(unsigned)n, 256, 0, s>>>((const P *)m->d_rowptrs, m->d_colinds, m->d_values, p->hr_rows.as<int32_t>(), \
                                                      p->hr_code.as<int32_t>(), doff.as<int64_t>(), G, key.as<int32_t>(),  \
                                                      pay.as<int32_t>(), val.as<double>())
    if (m->val_type == CSRK_VAL_F64) EMIT(CSRK_VAL_F64);
    else if (m->val_type == CSRK_VAL_F32) EMIT(CSRK_VAL_F32);
    else EMIT(CSRK_VAL_NONE);
#undef EMIT
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(stable_sort_records_f64(key.as<int32_t>(), pay.as<int32_t>(), val.as<double>(), nnz_h, (int32_t)n_buckets, 1 << 16,
                                     p->hr_bp.as<int64_t>(), (int32_t *)p->hr_idx.p, p->hr_vals.as<double>(), s));
    // column ranges: whole tiles, balanced by entries plus a fixed cost per tile (staging + barrier ~ 200 entries' worth)
    std::vector<int64_t> tot((size_t)n_tiles + 1);
    {
        DevBuf dt;
        CSRK_TRY(dt.alloc((size_t)(n_tiles + 1) * 8));
        hr_tile_totals_kernel<<<(unsigned)ceil_div(n_tiles + 1, 256), 256, 0, s>>>(p->hr_bp.as<int64_t>(), (int32_t)n_tiles, G * HR_WAVES,
                                                                                 dt.as<int64_t>());
        CSRK_LAUNCH_CHECK();
        CSRK_TRY(stage_d2h(tot.data(), dt.p, (size_t)(n_tiles + 1) * 8, s));
        CSRK_HIP(hipStreamSynchronize(s));
    }
    const int64_t tile_cost = 200 * (int64_t)G;
    const double total_cost = (double)nnz_h + (double)tile_cost * (double)n_tiles;
    std::vector<int32_t> range((size_t)R + 1);
    range[0] = 0;
    {
        int64_t t = 0;
        for (int q = 1; q < R; q++) {
            const double goal = total_cost * q / R;
            while (t < n_tiles && (double)tot[(size_t)t + 1] + (double)tile_cost * (double)(t + 1) <= goal) t++;
            // every range keeps at least one tile, and leaves one for each range after it
            t = std::max<int64_t>(t, (int64_t)range[(size_t)q - 1] + 1);
            t = std::min<int64_t>(t, n_tiles - (R - q));
            range[(size_t)q] = (int32_t)t;
        }
    }
    range[(size_t)R] = (int32_t)n_tiles;
    CSRK_TRY(p->hr_range.alloc(((size_t)R + 1) * 4));
    CSRK_TRY(stage_h2d(p->hr_range.p, range.data(), ((size_t)R + 1) * 4, s));
    CSRK_TRY(p->hr_part.alloc((size_t)R * G * HR_ROWS * HR_KC * 8));
    CSRK_HIP(hipFuncSetAttribute((const void *)spmm_hrows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HR_LDS));
    CSRK_HIP(hipFuncSetAttribute((const void *)spmm_hrows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HR_LDS));
    CSRK_HIP(hipStreamSynchronize(s));      // host vectors of the async copies; the temporaries are freed on return
    p->hr_on = true;
    p->hr_min = (int32_t)hmin;
    p->hr_n = (int32_t)n;
    p->hr_G = G;
    p->hr_R = R;
    p->hr_tiles = (int32_t)n_tiles;
    p->hr_nnz = nnz_h;
    (void)k;
    return CSRK_OK;
}

static int spmm_device(Matrix *m, const double *dB, int32_t k, int64_t ldb, double *dC, int64_t ldc, hipStream_t s,
                       SpmmOut om = {nullptr, 0})
{
    CSRK_REQUIRE(k >= 0 && ldb >= k && ldc >= k, "bad panel geometry k=%d ldb=%lld ldc=%lld", k, (long long)ldb, (long long)ldc);
    if (m->nrows == 0 || k == 0) return CSRK_OK;
    SpmmPlan *p;
    // The launch group below shares the plan's partial-panel buffers (hr_part, part): the per-handle lock is held
    // across it, as spmv_dispatch does, so that concurrent callers are ordered by the stream instead of interleaving
    // (csrk.h: calls on the same handle serialise).
    std::lock_guard<std::mutex> lk(m->mu);
    if (s) m->used_user_stream = true;
    if (!m->spmm_plan) {
        SpmmPlan *np = new (std::nothrow) SpmmPlan();
        CSRK_REQUIRE(np, "out of host memory");
        // default stream + completion before use: see the caching allocator's contract (common.h)
        int rc = CSRK_OK;
        // the heavy-row form pays when B as a whole does not fit the caches (CSRK_SPMM_HEAVY=0 / 1: never / always)
        const char *env = getenv("CSRK_SPMM_HEAVY");
        const bool force = env && env[0] == '1';
        if (!(env && env[0] == '0') && ((int64_t)m->ncols * k * 8 > (64ll << 20) || force))
            rc = m->ptr64 ? build_hrows<int64_t>(m, np, k, force, nullptr) : build_hrows<int32_t>(m, np, k, force, nullptr);
        if (rc == CSRK_OK) rc = m->ptr64 ? build_mm_plan<int64_t>(m, np, nullptr) : build_mm_plan<int32_t>(m, np, nullptr);
        if (rc == CSRK_OK && hipDeviceSynchronize() != hipSuccess) rc = CSRK_ERR_HIP;
        if (rc != CSRK_OK) {
            delete np;
            return rc;
        }
        m->spmm_plan = np;
    }
    p = m->spmm_plan;
    // a wider panel than any before needs a larger partial buffer: earlier launches (any stream) may still be using the
    // old block, which DevBuf::alloc returns to the pool at once -- wait for the device first
    const bool grow_p = p->n_multi > 0 && p->part_k < k;
    if (grow_p && p->part.p) CSRK_HIP(hipDeviceSynchronize());
    if (grow_p) {           // some row is split: partial panels needed
        CSRK_TRY(p->part.alloc((size_t)p->n_multi * k * 8));
        p->part_k = k;
    }
    const bool full4 = (k % 4) == 0;
    if (p->hr_on) {
        const unsigned wgs = (unsigned)(p->hr_G * p->hr_R);
        const unsigned rgrid = (unsigned)ceil_div((int64_t)p->hr_n * WAVE, 256);
        for (int32_t c0 = 0; c0 < k; c0 += HR_KC) {        // 64 panel columns per launch (stream order: hr_part is reused)
            const int32_t kc = k - c0 < HR_KC ? k - c0 : HR_KC;
#define HROWS(EVEN)                                                                                                    \
    spmm_hrows_kernel<EVEN><<<wgs, HR_THREADS, HR_LDS, s>>>(p->hr_bp.as<int64_t>(), p->hr_idx.as<uint32_t>(),             \
                                                           p->hr_vals.as<double>(), dB + c0, kc, ldb, m->ncols,          \
                                                           p->hr_range.as<int32_t>(), p->hr_G, p->hr_part.as<double>())
            if (kc % 2 == 0) HROWS(true);
            else HROWS(false);
#undef HROWS
            CSRK_LAUNCH_CHECK();
            spmm_hrows_reduce_kernel<<<rgrid, 256, 0, s>>>(p->hr_rows.as<int32_t>(), p->hr_code.as<int32_t>(), p->hr_n, p->hr_G, p->hr_R,
                                                          kc, p->hr_part.as<double>(), dC, ldc, c0, om);
            CSRK_LAUNCH_CHECK();
        }
    }
    if (p->n_segs == 0) return CSRK_OK;
    const unsigned grid = (unsigned)ceil_div(p->n_segs * MM_G, 256);
#define GO(VT, F4)                                                                                                     \
    spmm_seg_kernel<VT, F4><<<grid, 256, 0, s>>>(m->d_colinds, m->d_values, dB, k, ldb, dC, ldc, p->seg.as<SegDesc>(),   \
                                                 p->n_segs, p->part.as<double>(), om)
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
    CSRK_LAUNCH_CHECK();
    if (p->n_multi > 0) {
        spmm_fixup_kernel<<<(unsigned)ceil_div((int64_t)p->n_split * WAVE, 256), 256, 0, s>>>(
            p->seg_off.as<int64_t>(), p->part_off.as<int64_t>(), p->split_rows.as<int32_t>(), p->n_split, k,
            p->part.as<double>(), dC, ldc, om);
        CSRK_LAUNCH_CHECK();
    }
    return CSRK_OK;
}

// ---- mult_ab(A, B) with B a fully populated CSR: the reference's own route to BASELINE configs[2] ---------------------
// The reference has no dense-panel entry: a caller computes A x dense B as A.multiply(CSR(B)) -> K.mult_ab
// (csr/csr.py:524-567, csr/kernels/numba/multiply.py:13-38).  When every row of B holds all k columns in ascending order
// its `values` array IS the row-major panel, and the product the reference returns is fixed by its loops alone: the first
// entry of a row of A discovers the k columns 0 .. k - 1 of its row of B, pushed one by one onto the FRONT of the row's
// list (multiply.py:79-82), copied out front to back (:94-97) -- every row of C whose row of A holds an entry is
// k - 1 .. 0, the others are empty -- and the values are the panel's (work[c] += a * b over the row's entries in storage
// order, :110-122; explicit zeros kept).  One kernel checks B, the dense-panel kernels above write C's values in place
// (SpmmOut), C's index arrays are filled beside them.  Anything else -- a row of B short of a column, or in another
// order; float32 values on BOTH operands, whose products the reference rounds to float32 -- takes the general product.
template <class P>
__global__ __launch_bounds__(256) void dense_b_check_kernel(const P *__restrict__ rp, const int32_t *__restrict__ ci,
                                                           int32_t nrows, int32_t k, int32_t *__restrict__ bad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)nrows * k;
    bool ok = true;
    if (i <= nrows) ok = (int64_t)rp[i] == i * k;
    if (i < n) ok = ok && ci[i] == (int32_t)(i % k);
    if (!ok) *bad = 1;      // (any one writer: the flag is all that is read)
}

template <class P>
__global__ void dense_c_live_kernel(const P *__restrict__ rp, int32_t nrows, int64_t *__restrict__ live)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrows) live[i] = rp[i + 1] > rp[i] ? 1 : 0;
}

// rank[i] = rows of A with entries before row i (rank[nrows] = all of them): C's row pointers and where its rows start
__global__ void dense_c_rows_kernel(const int64_t *__restrict__ rank, int32_t nrows, int32_t k, int32_t *__restrict__ crp,
                                    int64_t *__restrict__ row_base)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nrows) return;
    crp[i] = (int32_t)(rank[i] * k);
    if (i < nrows) row_base[i] = rank[i + 1] > rank[i] ? rank[i] * k : -1;
}

__global__ __launch_bounds__(256) void dense_c_cols_kernel(int32_t *__restrict__ ci, int64_t n, int32_t k)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(k - 1 - (int32_t)(i % k), ci + i);
}

__global__ void dense_widen_kernel(const float *__restrict__ in, double *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}

// *taken = false: B is not a row-major panel (or the route is switched off): the caller runs the general product
int spgemm_dense_b(Matrix *a, Matrix *b, Matrix **out, bool *taken)
{
    *taken = false;
    const char *env = getenv("CSRK_SPGEMM_DENSE");
    if (env && env[0] == '0') return CSRK_OK;
    const int32_t k = b->ncols;
    if (a->ncols != b->nrows || k < 1 || b->nrows < 1 || a->nnz == 0 || b->nnz != (int64_t)b->nrows * k) return CSRK_OK;
    if (a->val_type == CSRK_VAL_NONE || b->val_type == CSRK_VAL_NONE) return CSRK_OK;      // (the general product reports it)
    if (a->val_type == CSRK_VAL_F32 && b->val_type == CSRK_VAL_F32) return CSRK_OK;         // float32 products: multiply.py:120
    // (the look at B -- 4 bytes per entry: 0.08 ms for configs[2] -- is remembered by the handle: Matrix::dense_panel)
    int known;
    {
        std::lock_guard<std::mutex> lk(b->mu);
        known = b->dense_panel;
    }
    if (known == 0) return CSRK_OK;
    DevBuf bad;
    CSRK_TRY(bad.alloc(4));
    CSRK_HIP(hipMemsetAsync(bad.p, 0, 4, nullptr));
    if (known < 0) {
        const unsigned gb = (unsigned)ceil_div(b->nnz + 1, 256);
        if (b->ptr64) dense_b_check_kernel<int64_t><<<gb, 256>>>((const int64_t *)b->d_rowptrs, b->d_colinds, b->nrows, k, bad.as<int32_t>());
        else dense_b_check_kernel<int32_t><<<gb, 256>>>((const int32_t *)b->d_rowptrs, b->d_colinds, b->nrows, k, bad.as<int32_t>());
        CSRK_LAUNCH_CHECK();
    }
    // C's rows meanwhile: which rows of A hold an entry
    const int32_t nr = a->nrows;
    DevBuf rank, row_base;
    CSRK_TRY(rank.alloc((size_t)(nr + 1) * 8));
    CSRK_TRY(row_base.alloc((size_t)(nr + 1) * 8));
    const unsigned ga = (unsigned)ceil_div(nr + 1, 256);
    if (a->ptr64) dense_c_live_kernel<int64_t><<<ga, 256>>>((const int64_t *)a->d_rowptrs, nr, rank.as<int64_t>());
    else dense_c_live_kernel<int32_t><<<ga, 256>>>((const int32_t *)a->d_rowptrs, nr, rank.as<int64_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(exclusive_scan_i64(rank.as<int64_t>(), rank.as<int64_t>(), nr, nullptr));
    int32_t is_bad = 0;
    int64_t live = 0;
    CSRK_TRY(stage_d2h(&is_bad, bad.p, 4, nullptr));
    CSRK_TRY(stage_d2h(&live, rank.as<int64_t>() + nr, 8, nullptr));
    if (known < 0) {
        std::lock_guard<std::mutex> lk(b->mu);
        b->dense_panel = is_bad ? 0 : 1;
    }
    if (is_bad) return CSRK_OK;
    const int64_t c_nnz = live * k;
    if (c_nnz > INT32_MAX) {
        set_error("product has %lld entries; the reference's int32 row pointers (multiply.py:28) cannot hold it: "
                  "multiply row blocks of A instead", (long long)c_nnz);
        return CSRK_ERR_OVERFLOW;
    }
    Matrix *c = nullptr;
    CSRK_TRY(new_matrix(nr, k, c_nnz, 0, CSRK_VAL_F64, &c));
    dense_c_rows_kernel<<<ga, 256>>>(rank.as<int64_t>(), nr, k, (int32_t *)c->d_rowptrs, row_base.as<int64_t>());
    int rc = hipGetLastError() == hipSuccess ? CSRK_OK : CSRK_ERR_HIP;
    if (rc == CSRK_OK && c_nnz > 0) {
        dense_c_cols_kernel<<<(unsigned)ceil_div(c_nnz, 256), 256>>>(c->d_colinds, c_nnz, k);
        rc = hipGetLastError() == hipSuccess ? CSRK_OK : CSRK_ERR_HIP;
    }
    DevBuf wide;      // a float32 panel under float64 values of A: widened (exactly) once
    const double *panel = (const double *)b->d_values;
    if (rc == CSRK_OK && b->val_type == CSRK_VAL_F32) {
        rc = wide.alloc((size_t)b->nnz * 8);
        if (rc == CSRK_OK) {
            dense_widen_kernel<<<(unsigned)ceil_div(b->nnz, 256), 256>>>((const float *)b->d_values, wide.as<double>(), b->nnz);
            rc = hipGetLastError() == hipSuccess ? CSRK_OK : CSRK_ERR_HIP;
            panel = wide.as<double>();
        }
    }
    if (rc == CSRK_OK && c_nnz > 0) rc = spmm_device(a, panel, k, k, (double *)c->d_values, k, nullptr, SpmmOut{row_base.as<int64_t>(), k});
    // (row_base, the widened panel and C's arrays are recycled in default-stream order: the launches above are on it)
    if (rc == CSRK_OK && hipDeviceSynchronize() != hipSuccess) rc = CSRK_ERR_HIP;
    if (rc != CSRK_OK) {
        if (rc == CSRK_ERR_HIP) set_error("mult_ab (dense B): %s", hipGetErrorString(hipGetLastError()));
        delete c;
        return rc;
    }
    *out = c;
    *taken = true;
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

int csrk_spmm_plan_stats(csrk_handle_t h, int64_t *out, int n)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE(out && n >= 0, "out is NULL");
    std::lock_guard<std::mutex> lk(m->mu);
    const SpmmPlan *p = m->spmm_plan;
    const int64_t v[9] = {p && p->hr_on ? 1 : 0, p ? p->hr_min : 0, p ? p->hr_n : 0, p ? p->hr_G : 0, p ? p->hr_R : 0,
                          p ? p->hr_nnz : 0, p ? p->hr_tiles : 0, p ? p->n_segs : 0, p ? p->n_multi : 0};
    for (int i = 0; i < n && i < 9; i++) out[i] = p && !p->hr_on && i >= 1 && i <= 6 ? 0 : v[i];
    return CSRK_OK;
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
