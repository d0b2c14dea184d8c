// SpGEMM for libcsrk on gfx950: C = A B and C = A B^T, all three CSR.  Replaces the
// reference's SMMP implementation (csr/kernels/numba/multiply.py:13-38 mult_ab, :60-100
// _sym_mm, :103-129 _num_mm, :41-57 mult_abt) and lk_mkl_spmab / lk_mkl_spmabt
// (csr/kernels/mkl/mkl_ops.c).
//
// The reference is a two-pass row-by-row algorithm with a dense marker/work row.  The GPU
// version keeps the two passes (symbolic count -> exclusive scan -> numeric fill) but gives
// every output row its own accumulator so rows run in parallel, by product count (sum over A_i of |B_j|):
//   * <= 128 products: a quarter or half WAVEFRONT per row (16 lanes up to 32 products, 32 lanes up to 128), a private
//     64- or 256-slot hash table in LDS, rank sort;
//   * more than 1024 products, at least 4 per (A entry, 832-column strip) on average, B's rows ascending
//     (every A B^T block of a ratings matrix): COLUMN STRIPS -- one wavefront per (row, strip), the
//     strip's sums in 6.5 KiB of LDS, sub-range bounds of every (A entry, strip) found once -- see
//     "column strips";
//   * every other row above 128 products: EXPAND-SORT-COMPRESS -- the products written out in the
//     reference's order, two stable transposes, runs of equal columns added front to back -- see
//     "expand, sort, compress";
//   * fallbacks when those do not apply (B unsorted and the output dense, products beyond the sort's
//     budget): one 256-thread workgroup per row with a 2048-slot LDS hash table (<= 1024 products),
//     16384-column float64 tiles in LDS for nearly full rows, an 8192-slot LDS hash table for rows with
//     at most 4096 distinct outputs, dense float64 work rows in HBM (the reference's `work` / `index`
//     arrays, multiply.py:62,106) for the rest.
// sg_list_rows writes every row's route; the kernels of one path skip the others' rows.
// Like the reference, entries that cancel to exactly 0.0 are KEPT (csr/csr.py:555 filters
// them afterwards) and C's row pointers are int32 (multiply.py:28).
// Column order inside a row is ascending here; the reference's order (reverse discovery) is
// pinned by none of its tests (hard part 3).
//
// Determinism.  Every path gives each output entry its products in the reference's order -- A's entries jj in storage
// order, for each the row B_j (work[k] += a * b, multiply.py:117-121) -- with the products rounded before they are
// added, so the sums are bitwise reproducible run to run (test_spgemm_deterministic) and equal to the sequential
// loop's bit for bit whenever B's rows hold no column twice.  The workgroup paths do it by walking with ALL lanes on ONE
// jj at a time (sg_walk_products: inside one B_j the columns are distinct, a barrier separates the steps); the strips by
// giving every column to one wavefront, whose LDS operations complete in program order; expand-sort-compress by a stable
// sort.  (Round 1 let the wavefronts of a workgroup take different jj and race on the accumulators: last bits varied.)
#include "common.h"
#include "wave.h"

namespace csrk {

int transpose_matrix(Matrix *a, int with_values, Matrix **out, hipStream_t s);   // transpose.hip
bool spgemm_reference_order_wanted();                                            // spgemm_order.hip
int spgemm_apply_reference_order(Matrix *a, Matrix *b, Matrix *c, const DevBuf *products);

struct MatView {
    const void *rp;
    const int32_t *ci;
    const void *vs;
    int ptr64;
    int vt;
    int32_t nrows, ncols;
    int64_t nnz;
};

// FAST: both operands have int32 row pointers and float64 values (every A B^T -- the transpose is float64 -- and the
// usual A B): the accessors are plain typed loads.  With the run-time dtype tests of the general form every load sits
// behind its own branch, the compiler cannot keep several in flight, and a row of B costs one memory round trip per
// load (measured: the MovieLens-shaped A B^T block 8.7 ms in the general form).
template <bool FAST>
__device__ __forceinline__ int64_t rp_at(const MatView &m, int64_t i)
{
    if (FAST) return (int64_t)((const int32_t *)m.rp)[i];
    return m.ptr64 ? ((const int64_t *)m.rp)[i] : (int64_t)((const int32_t *)m.rp)[i];
}
template <bool FAST>
__device__ __forceinline__ double val_at(const MatView &m, int64_t k)
{
    if (FAST) return ((const double *)m.vs)[k];
    return m.vt == CSRK_VAL_F64 ? ((const double *)m.vs)[k] : (double)((const float *)m.vs)[k];
}

// one product a_ij * b_jk as the reference takes it (multiply.py:120, typed by Numba / NumPy by its operands): float32 times
// float32 is a float32 product -- one rounding -- before it is added to the float64 work array; everything else float64
template <bool FAST>
__device__ __forceinline__ double sg_mul(const MatView &a, const MatView &b, double av, double bv)
{
    if (!FAST && a.vt == CSRK_VAL_F32 && b.vt == CSRK_VAL_F32) return (double)__fmul_rn((float)av, (float)bv);
    return __dmul_rn(av, bv);
}

static MatView view_of(const Matrix *m)
{
    return MatView{m->d_rowptrs, m->d_colinds, m->d_values, m->ptr64, m->val_type, m->nrows, m->ncols, m->nnz};
}

// Walk the products of one output row -- A entries [as, ae), for each the whole row B_j -- in the reference's order with
// NT threads (a wavefront or a whole workgroup) in step: lane t takes B_j's entries t, t + NT, ...; `apply(k, a_ij, b_jk)`
// is called once per product; `step_done()` after the last product of every jj (a barrier for NT > 64, nothing for a
// single wavefront, whose LDS operations complete in order).  The (j, a_ij, row extent) of G consecutive jj and the first
// NT entries of each B_j are requested before the first product is applied, so a group costs two memory round trips
// instead of two per jj.  All trip counts are uniform over the NT threads.
constexpr int SG_U = 8;
template <bool FAST, int NT, int G, bool NEEDV, class Apply, class StepDone>
__device__ __forceinline__ void sg_walk_products(const MatView &a, const MatView &b, int64_t as, int64_t ae, int t,
                                                 Apply &&apply, StepDone &&step_done)
{
    for (int64_t jj0 = as; jj0 < ae; jj0 += G) {
        int64_t bs[G], be[G];
        double av[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int64_t jj = jj0 + g < ae ? jj0 + g : ae - 1;          // clamped: the loads stay unconditional
            const int32_t j = a.ci[jj];
            av[g] = NEEDV ? val_at<FAST>(a, jj) : 0.0;
            bs[g] = rp_at<FAST>(b, j);
            be[g] = jj0 + g < ae ? rp_at<FAST>(b, j + 1) : bs[g];              // past the row's end: an empty extent
        }
        // (loads past a row's end are clamped to B's last entry and selected away: a load behind a branch is waited for
        // before the next one is issued)
        const int64_t last = b.nnz - 1;
        int32_t k0[G];
        double v0[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int64_t kk = bs[g] + t;
            const bool in = kk < be[g];
            const int64_t kc = in ? kk : last;
            const int32_t kl = b.ci[kc];
            const double vl = NEEDV ? val_at<FAST>(b, kc) : 0.0;      // (the symbolic passes read the columns only)
            k0[g] = in ? kl : -1;
            v0[g] = vl;
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (jj0 + g < ae) {                                          // uniform
                if (k0[g] >= 0) apply(k0[g], av[g], v0[g]);
                // a long B_j: SG_U passes of NT entries requested at a time (one pass per round trip made a
                // 10^4-entry row of B a chain of ten memory latencies)
                for (int64_t kb = bs[g] + NT; kb < be[g]; kb += SG_U * (int64_t)NT) {      // uniform
                    int32_t kx[SG_U];
                    double vx[SG_U];
#pragma unroll
                    for (int u = 0; u < SG_U; u++) {
                        const int64_t kk = kb + (int64_t)u * NT + t;
                        const bool in = kk < be[g];
                        const int64_t kc = in ? kk : last;
                        const int32_t kl = b.ci[kc];
                        vx[u] = NEEDV ? val_at<FAST>(b, kc) : 0.0;
                        kx[u] = in ? kl : -1;
                    }
#pragma unroll
                    for (int u = 0; u < SG_U; u++)
                        if (kx[u] >= 0) apply(kx[u], av[g], vx[u]);
                }
                step_done();
            }
        }
    }
}

constexpr int SG_THREADS = 256;
constexpr int SG_SLOTS = 2048;
constexpr int SG_CAP = 1024;      // max products for the LDS hash path (load factor <= 0.5)
constexpr int SG_WAVE_CAP = 128;  // rows with at most this many products take the sub-wavefront kernels (sg_quad_kernel)

// products per output row: ub[i] = sum_{j in A_i} |B_j|.  (A thread per row spent 1.9 ms on 2000 rows of a MovieLens-shaped A,
// whose rows have up to 7000 entries.)  Eight lanes per row (most rows of a sparse product have a handful of entries: a wavefront per row took 159 us for 10^6
// rows); rows of more than SG_COUNT_LONG entries are listed for sg_count_products_long, a wavefront each.
constexpr int SG_COUNT_LONG = 64;
// The long rows are appended to SG_LONG_LISTS lists, workgroup b to list b % SG_LONG_LISTS (a counter per list, a cache line
// apart): appends to ONE counter are served one after the other, ~10 ns each -- 10^4 long rows of a 10^6-row power-law matrix
// cost 100 us of a 2.9-ms product.  List k holds at most long_cap rows (the rows of the workgroups that append to it).
constexpr int SG_LONG_LISTS = 64, SG_LONG_STRIDE = 32;      // (counters SG_LONG_STRIDE int32 apart)
// Rows of more than SG_COUNT_VLONG entries go to one more list (number SG_LONG_LISTS, at long_list + SG_LONG_LISTS * long_cap) and
// are counted by ALL wavefronts together, 256 entries at a time (a wavefront per row: the longest row of a power-law matrix
// alone was 0.10 ms).
constexpr int SG_COUNT_VLONG = 4096;
template <bool FAST>
__global__ __launch_bounds__(256) void sg_count_products(MatView a, MatView b, int64_t *__restrict__ ub,
                                                         int32_t *__restrict__ long_list, int32_t long_cap, int32_t *__restrict__ n_long)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / 8;
    const int l = threadIdx.x & 7;
    int64_t tot = 0;
    bool is_long = false;
    if (i < a.nrows) {
        const int64_t s = rp_at<FAST>(a, i), e = rp_at<FAST>(a, i + 1);
        is_long = e - s > SG_COUNT_LONG;
        if (!is_long)
            for (int64_t jj = s + l; jj < e; jj += 8) {
                const int32_t j = a.ci[jj];
                tot += rp_at<FAST>(b, j + 1) - rp_at<FAST>(b, j);
            }
    }
    for (int off = 4; off; off >>= 1) tot += __shfl_down(tot, off, 8);
    if (i < a.nrows && l == 0) {
        const bool vlong = rp_at<FAST>(a, i + 1) - rp_at<FAST>(a, i) > SG_COUNT_VLONG;
        const int k = vlong ? SG_LONG_LISTS : blockIdx.x % SG_LONG_LISTS;
        if (is_long) long_list[(int64_t)k * long_cap + atomicAdd(&n_long[k * SG_LONG_STRIDE], 1)] = (int32_t)i;
        ub[i] = is_long ? 0 : tot;
    }
}

// a wavefront per long row, four requests per lane in flight (one at a time: 0.10 ms for rows of a few thousand entries)
template <bool FAST>
__global__ __launch_bounds__(256) void sg_count_products_long(MatView a, MatView b, int64_t *__restrict__ ub,
                                                              const int32_t *__restrict__ long_list, int32_t long_cap,
                                                              const int32_t *__restrict__ n_long)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t n_waves = (int64_t)gridDim.x * blockDim.x / WAVE;
    const int64_t w0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    // the lists laid end to end: s_off[k] = rows in the lists before k (one load of the counters instead of one per list and wavefront)
    __shared__ int32_t s_off[SG_LONG_LISTS + 1];
    if (threadIdx.x < WAVE) {
        const int32_t n = n_long[threadIdx.x * SG_LONG_STRIDE];
        const int32_t ex = wave_exscan_i32(n, lane);
        s_off[threadIdx.x] = ex;
        if (threadIdx.x == WAVE - 1) s_off[SG_LONG_LISTS] = ex + n;
    }
    __syncthreads();
    const int32_t n_all = s_off[SG_LONG_LISTS];
    for (int64_t q = w0; q < n_all; q += n_waves) {
        int k = 0;
        for (int step = SG_LONG_LISTS / 2; step; step >>= 1)      // the list that holds position q
            if (s_off[k + step] <= q) k += step;
        const int32_t i = long_list[(int64_t)k * long_cap + (q - s_off[k])];
        const int64_t s = rp_at<FAST>(a, i), e = rp_at<FAST>(a, i + 1);
        int64_t tot = 0;
        for (int64_t j0 = s + lane; j0 < e; j0 += 4 * WAVE) {
            int64_t len[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int64_t jj = j0 + t * WAVE;
                const int32_t j = a.ci[jj < e ? jj : e - 1];
                len[t] = jj < e ? rp_at<FAST>(b, j + 1) - rp_at<FAST>(b, j) : 0;
            }
            tot += (len[0] + len[1]) + (len[2] + len[3]);
        }
        for (int off = WAVE / 2; off; off >>= 1) tot += __shfl_down(tot, off, WAVE);
        if (lane == 0) ub[i] = tot;
    }
    // the very long rows: every wavefront takes slices of 256 entries (integer adds: any order gives the same count)
    const int32_t n_vl = n_long[SG_LONG_LISTS * SG_LONG_STRIDE];
    for (int32_t q = 0; q < n_vl; q++) {
        const int32_t i = long_list[(int64_t)SG_LONG_LISTS * long_cap + q];
        const int64_t s = rp_at<FAST>(a, i), e = rp_at<FAST>(a, i + 1);
        int64_t tot = 0;
        for (int64_t j0 = s + w0 * (4 * WAVE) + lane; j0 < e; j0 += n_waves * (4 * WAVE)) {
            int64_t len[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int64_t jj = j0 + t * WAVE;
                const int32_t j = a.ci[jj < e ? jj : e - 1];
                len[t] = jj < e ? rp_at<FAST>(b, j + 1) - rp_at<FAST>(b, j) : 0;
            }
            tot += (len[0] + len[1]) + (len[2] + len[3]);
        }
        for (int off = WAVE / 2; off; off >>= 1) tot += __shfl_down(tot, off, WAVE);
        if (lane == 0 && tot) atomicAdd((unsigned long long *)&ub[i], (unsigned long long)tot);
    }
}

__device__ __forceinline__ uint32_t sg_hash(int32_t k) { return ((uint32_t)k * 2654435761u) >> 21; }   // 11 bits

// One workgroup per output row with 0 < ub <= SG_CAP.
template <bool NUMERIC, bool FAST>
__global__ __launch_bounds__(SG_THREADS) void sg_hash_kernel(MatView a, MatView b, const int64_t *__restrict__ ub,
                                                            const unsigned char *__restrict__ route, int32_t *__restrict__ cnt,
                                                            const int32_t *__restrict__ c_rp, int32_t *__restrict__ c_ci,
                                                            double *__restrict__ c_vs)
{
    __shared__ int32_t s_key[SG_SLOTS];
    __shared__ double s_val[NUMERIC ? SG_SLOTS : 1];
    __shared__ int32_t s_n;
    const int i = blockIdx.x, tid = threadIdx.x;
    const int64_t u = ub[i];
    if (u > SG_CAP || u <= SG_WAVE_CAP || route[i] != 0) return;   // heavy-row paths / sub-wavefront kernels / expand-sort-compress
    for (int s = tid; s < SG_SLOTS; s += SG_THREADS) {
        s_key[s] = -1;
        if (NUMERIC) s_val[s] = 0.0;
    }
    if (tid == 0) s_n = 0;
    __syncthreads();

    const int lane = tid & (WAVE - 1), w = tid / WAVE;
    const int64_t as = rp_at<FAST>(a, i), ae = rp_at<FAST>(a, i + 1);
    sg_walk_products<FAST, SG_THREADS, 4, NUMERIC>(
        a, b, as, ae, tid,
        [&](int32_t k, double av, double bv) {
            uint32_t slot = sg_hash(k);
            for (;;) {
                int32_t old = atomicCAS(&s_key[slot], -1, k);
                if (old == -1 || old == k) {
                    if (NUMERIC)
                        atomicAdd(&s_val[slot], sg_mul<FAST>(a, b, av, bv));
                    else if (old == -1)
                        atomicAdd(&s_n, 1);
                    break;
                }
                slot = (slot + 1) & (SG_SLOTS - 1);
            }
        },
        [&]() {
            if (NUMERIC) __syncthreads();      // (the distinct-column count of the symbolic pass does not depend on an order)
        });
    __syncthreads();
    if (!NUMERIC) {
        if (tid == 0) cnt[i] = s_n;
        return;
    }

    // compact the occupied slots to the front (in place is unsafe: use the packed order of a
    // block-wide scan over slot occupancy), then bitonic sort by column
    __shared__ int32_t s_ck[SG_CAP];
    __shared__ double s_cv[SG_CAP];
    __shared__ int32_t s_wsum[SG_THREADS / WAVE];
    int base = 0;
    for (int s0 = 0; s0 < SG_SLOTS; s0 += SG_THREADS) {
        const int s = s0 + tid;
        const bool occ = s_key[s] != -1;
        const unsigned long long bal = __ballot(occ);
        const int below = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wsum[w] = __popcll(bal);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < SG_THREADS / WAVE; k++) {
            if (k < w) woff += s_wsum[k];
            tot += s_wsum[k];
        }
        if (occ) {
            s_ck[base + woff + below] = s_key[s];
            s_cv[base + woff + below] = s_val[s];
        }
        base += tot;
        __syncthreads();
    }
    const int n = base;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int s = n + tid; s < np2; s += SG_THREADS) s_ck[s] = 0x7fffffff;
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < np2 / 2; t += SG_THREADS) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const int32_t kl = s_ck[lo], kh = s_ck[hi];
                if ((kl > kh) == up) {
                    s_ck[lo] = kh;
                    s_ck[hi] = kl;
                    const double vl = s_cv[lo];
                    s_cv[lo] = s_cv[hi];
                    s_cv[hi] = vl;
                }
            }
            __syncthreads();
        }
    }
    const int32_t o = c_rp[i];
    for (int t = tid; t < n; t += SG_THREADS) {
        c_ci[o + t] = s_ck[t];
        c_vs[o + t] = s_cv[t];
    }
}

// Rows with at most SG_WAVE_CAP products (most rows of a sparse product): a FRACTION of a wavefront per row -- 16 lanes for
// rows of at most 32 products (four rows per wavefront, sixteen per workgroup), 32 lanes up to 128 products.  A whole
// wavefront per row spent ~400 instructions on a row with a handful of its 64 lanes busy (10^6 rows: 0.95 ms in the two
// kernels); here the rows of a wavefront share one instruction stream.  Same table (SLOTS = 2 CAP per row, keys by
// atomicCAS, values by LDS float64 add) and the same order: a row's lanes take A's entries one at a time, ascending, and
// the (distinct) columns of one B_j LANES at a time; a wavefront's LDS operations complete in program order, and no
// two rows share a table, so there is no barrier at all.
template <int SGQ_LANES, int SGQ_CAP, bool NUMERIC, bool FAST>
__global__ __launch_bounds__(256) void sg_quad_kernel(MatView a, MatView b, const int64_t *__restrict__ ub, int64_t lo,
                                                     const unsigned char *__restrict__ route, int32_t *__restrict__ cnt,
                                                     const int32_t *__restrict__ c_rp, int32_t *__restrict__ c_ci,
                                                     double *__restrict__ c_vs, const int64_t *__restrict__ t_off)
{
    constexpr int SGQ_SLOTS = 2 * SGQ_CAP, SGQ_ROWS = 256 / SGQ_LANES;
    __shared__ int32_t s_key[SGQ_ROWS][SGQ_SLOTS];
    __shared__ double s_val[NUMERIC ? SGQ_ROWS : 1][NUMERIC ? SGQ_SLOTS : 1];
    __shared__ int32_t s_ck[SGQ_ROWS][SGQ_CAP];
    __shared__ double s_cv[NUMERIC ? SGQ_ROWS : 1][NUMERIC ? SGQ_CAP : 1];
    const int lane = threadIdx.x & (WAVE - 1), l = threadIdx.x & (SGQ_LANES - 1), r = threadIdx.x / SGQ_LANES;
    const int grp = lane / SGQ_LANES;                       // this row's place in the wavefront
    const int64_t i = (int64_t)blockIdx.x * SGQ_ROWS + r;
    const int64_t u = i < a.nrows ? ub[i] : 0;
    const bool mine = u > lo && u <= SGQ_CAP && route[i] == 0;
    for (int sl = l; sl < SGQ_SLOTS; sl += SGQ_LANES) {
        s_key[r][sl] = -1;
        if (NUMERIC) s_val[r][sl] = 0.0;
    }
    int64_t jj0 = 0, ae = 0;
    if (mine) {
        jj0 = rp_at<FAST>(a, i);
        ae = rp_at<FAST>(a, i + 1);
    }
    const int64_t last = b.nnz - 1;
    while (__any(jj0 < ae)) {                               // uniform: A's entries, LANES of every row at a time
        // lane l fetches (a_ij, extent of B_j) of entry jj0 + l: two round trips per LANES entries instead of per entry
        const bool have = jj0 + l < ae;
        const int64_t jm = have ? jj0 + l : 0;
        const int32_t j_m = a.ci[jm];
        const double av_m = NUMERIC ? val_at<FAST>(a, jm) : 0.0;
        const int64_t bs_m = have ? rp_at<FAST>(b, j_m) : 0;
        const int64_t be_m = have ? rp_at<FAST>(b, j_m + 1) : 0;
        const int nb = jj0 < ae ? (int)(ae - jj0 < SGQ_LANES ? ae - jj0 : SGQ_LANES) : 0;
        for (int st = 0; __any(st < nb); st++) {            // uniform: one A entry of every row per step
            const int64_t bs = __shfl(bs_m, st, SGQ_LANES);
            const int64_t be = st < nb ? __shfl(be_m, st, SGQ_LANES) : bs;
            const double av = NUMERIC ? __shfl(av_m, st, SGQ_LANES) : 0.0;
            int64_t kk = bs + l;
            while (__any(kk < be)) {                        // uniform: LANES entries of each row's B_j per pass
                const bool in = kk < be;
                const int64_t kc = in ? kk : last;
                const int32_t k = b.ci[kc];
                const double bv = NUMERIC ? val_at<FAST>(b, kc) : 0.0;
                if (in) {
                    uint32_t slot = ((uint32_t)k * 2654435761u) & (SGQ_SLOTS - 1);
                    for (;;) {
                        const int32_t old = atomicCAS(&s_key[r][slot], -1, k);
                        if (old == -1 || old == k) {
                            if (NUMERIC) atomicAdd(&s_val[r][slot], sg_mul<FAST>(a, b, av, bv));
                            break;
                        }
                        slot = (slot + 1) & (SGQ_SLOTS - 1);
                    }
                }
                kk += SGQ_LANES;
            }
        }
        jj0 += SGQ_LANES;
    }
    // compact the occupied slots (a row's 16 bits of the wavefront's ballot), then rank-sort by column (keys are unique)
    int n = 0;
    for (int s0 = 0; s0 < SGQ_SLOTS; s0 += SGQ_LANES) {
        const int32_t k = s_key[r][s0 + l];
        const bool occ = mine && k != -1;
        const unsigned bits = (unsigned)((__ballot(occ) >> (grp * SGQ_LANES)) & (SGQ_LANES == 32 ? 0xffffffffull : 0xffffull));
        if (occ) {
            const int o = n + __popc(bits & ((1u << l) - 1u));
            s_ck[r][o] = k;
            if (NUMERIC) s_cv[r][o] = s_val[r][s0 + l];
        }
        n += __popc(bits);
    }
    if (!mine) return;
    if (!NUMERIC || t_off) {
        if (l == 0) cnt[i] = n;
        if (!NUMERIC) return;
    }
    const int64_t o0 = t_off ? t_off[i] : (int64_t)c_rp[i];
    for (int t = l; t < n; t += SGQ_LANES) {
        const int32_t kt = s_ck[r][t];
        int rank = 0;
        for (int q = 0; q < n; q++) rank += s_ck[r][q] < kt;
        c_ci[o0 + rank] = kt;
        c_vs[o0 + rank] = s_cv[r][t];
    }
}

// room[i] = products of a row the sub-wavefront kernels serve, 0 for the others (their temporaries' sizes)
__global__ void sg_small_room(const int64_t *__restrict__ ub, const unsigned char *__restrict__ route, int32_t nrows,
                              int32_t *__restrict__ room)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    const int64_t u = ub[i];
    room[i] = (u > 0 && u <= SG_WAVE_CAP && route[i] == 0) ? (int32_t)u : 0;
}

// the sub-wavefront kernels' temporaries -> C: 16 lanes per row
__global__ __launch_bounds__(256) void sg_small_copy(const int32_t *__restrict__ room, int32_t nrows, const int32_t *__restrict__ cnt,
                                                     const int64_t *__restrict__ t_off, const int32_t *__restrict__ t_ci,
                                                     const double *__restrict__ t_vs, const int32_t *__restrict__ c_rp,
                                                     int32_t *__restrict__ c_ci, double *__restrict__ c_vs)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / 16;
    const int l = threadIdx.x & 15;
    if (i >= nrows || room[i] == 0) return;
    const int64_t src = t_off[i], dst = c_rp[i];
    const int32_t n = cnt[i];
    for (int32_t t = l; t < n; t += 16) {
        c_ci[dst + t] = t_ci[src + t];
        c_vs[dst + t] = t_vs[src + t];
    }
}

// Persistent workgroups for rows with ub > SG_CAP: dense work/marker rows in HBM (the reference's
// `work` / `index` arrays).  The columns a row touches are appended to a list as they are first
// marked; the list is sorted -- bitonic network, in LDS up to SG_LSORT entries, otherwise in a padded
// global scratch buffer -- and the sums are gathered through it.  (A first version swept the whole
// dense row instead: O(ncols) per row, 37 of the 45 ms of a 200k x 200k product.)
constexpr int SG_LSORT = 8192;

// ascending bitonic sort of p[0..np2) (np2 a power of two) by the whole workgroup
__device__ inline void wg_bitonic(int32_t *p, int np2, int tid)
{
    for (int size = 2; size <= np2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < np2 / 2; t += SG_THREADS) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const int32_t kl = p[lo], kh = p[hi];
                if ((kl > kh) == up && kl != kh) {
                    p[lo] = kh;
                    p[hi] = kl;
                }
            }
            __threadfence_block();
            __syncthreads();
        }
    }
}

template <bool NUMERIC, bool FAST>
__global__ __launch_bounds__(SG_THREADS) void sg_dense_kernel(MatView a, MatView b, const int32_t *__restrict__ list,
                                                             int32_t n_large, double *__restrict__ work_all,
                                                             int32_t *__restrict__ mark_all, int32_t *__restrict__ scratch_all,
                                                             int64_t scratch_len, int32_t *__restrict__ cnt,
                                                             const int32_t *__restrict__ c_rp, int32_t *__restrict__ c_ci,
                                                             double *__restrict__ c_vs)
{
    __shared__ int32_t s_n;
    __shared__ int32_t s_wsum[SG_THREADS / WAVE];
    __shared__ int32_t s_sort[NUMERIC ? SG_LSORT : 1];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int32_t nc = b.ncols;
    double *work = work_all + (int64_t)blockIdx.x * nc;
    int32_t *mark = mark_all + (int64_t)blockIdx.x * nc;
    int32_t *scratch = NUMERIC ? scratch_all + (int64_t)blockIdx.x * scratch_len : nullptr;
    for (int q = blockIdx.x; q < n_large; q += gridDim.x) {
        const int i = list[q];
        const int64_t as = rp_at<FAST>(a, i), ae = rp_at<FAST>(a, i + 1);
        const int32_t base = NUMERIC ? c_rp[i] : 0;
        const int32_t n_out = NUMERIC ? c_rp[i + 1] - base : 0;
        // where the touched-column list is collected and sorted
        int32_t *lst = NUMERIC ? (n_out <= SG_LSORT ? s_sort : scratch) : nullptr;
        // nearly full row: an ascending compaction sweep over the dense row costs O(ncols) ~ O(n) and
        // beats list + sort (MovieLens-shaped A B^T blocks); sparse rows list and sort
        const bool sweep = NUMERIC && (int64_t)n_out * 8 >= nc;
        if (tid == 0) s_n = 0;
        __syncthreads();
        sg_walk_products<FAST, SG_THREADS, 4, NUMERIC>(
            a, b, as, ae, tid,
            [&](int32_t k, double av, double bv) {
                if (NUMERIC) atomicAdd(&work[k], sg_mul<FAST>(a, b, av, bv));
                if (NUMERIC && sweep) {
                    mark[k] = 1;                       // the sweep only needs the marker
                } else if (atomicExch(&mark[k], 1) == 0) {
                    const int o = atomicAdd(&s_n, 1);
                    if (NUMERIC) lst[o] = k;
                }
            },
            [&]() {
                // the step's memory-side adds have been acknowledged (s_waitcnt vmcnt(0) in the barrier) before the
                // next jj's are issued
                if (NUMERIC) __syncthreads();
            });
        __threadfence_block();
        __syncthreads();
        if (!NUMERIC) {
            if (tid == 0) cnt[i] = s_n;
            // clear the markers by walking the products again
            for (int64_t jj = as + w; jj < ae; jj += SG_THREADS / WAVE) {
                const int32_t j = a.ci[jj];
                const int64_t bs = rp_at<FAST>(b, j), be = rp_at<FAST>(b, j + 1);
                for (int64_t kk = bs + lane; kk < be; kk += WAVE) mark[b.ci[kk]] = 0;
            }
            __syncthreads();
            continue;
        }
        if (sweep) {
            int pos = base;
            for (int32_t k0 = 0; k0 < nc; k0 += SG_THREADS) {
                const int32_t k = k0 + tid;
                const bool occ = k < nc && __hip_atomic_load(&mark[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                const unsigned long long bal = __ballot(occ);
                const int below = __popcll(bal & ((1ull << lane) - 1ull));
                if (lane == 0) s_wsum[w] = __popcll(bal);
                __syncthreads();
                int woff = 0, tot = 0;
#pragma unroll
                for (int t = 0; t < SG_THREADS / WAVE; t++) {
                    if (t < w) woff += s_wsum[t];
                    tot += s_wsum[t];
                }
                if (occ) {
                    c_ci[pos + woff + below] = k;
                    c_vs[pos + woff + below] = __hip_atomic_load(&work[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    work[k] = 0.0;
                    mark[k] = 0;
                }
                pos += tot;
                __syncthreads();
            }
            continue;
        }
        const int n = s_n;                    // == n_out (same products as the symbolic pass)
        int np2 = 1;
        while (np2 < n) np2 <<= 1;
        for (int t = n + tid; t < np2; t += SG_THREADS) lst[t] = 0x7fffffff;
        __threadfence_block();
        __syncthreads();
        wg_bitonic(lst, np2, tid);
        for (int t = tid; t < n; t += SG_THREADS) {
            const int32_t k = lst[t];
            c_ci[base + t] = k;
            // agent-scope load: the sum was formed by L2 atomics, which do not update this CU's L1
            c_vs[base + t] = __hip_atomic_load(&work[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            work[k] = 0.0;
            mark[k] = 0;
        }
        __threadfence_block();
        __syncthreads();
    }
}

// ---- rows with many products, accumulated in LDS -------------------------------------------------------
// The dense path above keeps a row's `work` / `index` arrays in HBM and pays one memory-side atomic per
// product (plus one on the marker): ~16 ps per product.  When the output row is nearly full -- every A B^T
// block of a MovieLens-shaped matrix -- it is cheaper to keep the accumulator ON CHIP:
//   symbolic: one LDS bit per output column (up to 2^20 columns), set with ds_or, counted with popcount;
//   numeric:  the output columns are taken in tiles of SGL_W = 20096 (157 KiB of float64 accumulators + a bit
//             per column); per tile the row's products are walked once (products outside the tile are
//             skipped), accumulated with ds_add_f64, and the tile is compacted in ascending column order.
// One persistent 1024-thread workgroup per CU.  Ascending columns, cancellation zeros kept, like the other
// paths; sums by LDS atomics (last bits may vary run to run, as before).
constexpr int SGL_THREADS = 1024;
constexpr int SGL_W = 20096;               // 157 KiB of accumulators + a bit per column: the LDS of a CU
constexpr int SGL_MAXBITS = 1 << 20;       // symbolic: columns per LDS bitmask (128 KiB)
constexpr int SGL_MAXTILES = 8;            // numeric: more column tiles than this -> the HBM path

template <bool FAST>
__global__ __launch_bounds__(SGL_THREADS) void sg_lds_symbolic_kernel(MatView a, MatView b, const int32_t *__restrict__ list,
                                                                     int32_t n_large, int32_t *__restrict__ cnt,
                                                                     int32_t *__restrict__ next)
{
    extern __shared__ uint32_t sgl_bits[];
    __shared__ int32_t s_tot, s_q;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int nwords = (b.ncols + 31) / 32;
    for (;;) {
        // rows are claimed one at a time (product counts differ by orders of magnitude between rows)
        if (tid == 0) s_q = atomicAdd(next, 1);
        __syncthreads();
        const int q = s_q;
        if (q >= n_large) break;
        const int i = list[q];
        for (int k = tid; k < nwords; k += SGL_THREADS) sgl_bits[k] = 0;
        if (tid == 0) s_tot = 0;
        __syncthreads();
        const int64_t as = rp_at<FAST>(a, i), ae = rp_at<FAST>(a, i + 1);
        // (the set of columns does not depend on the order: every wavefront walks its own slice of A_i)
        const int64_t per = (ae - as + SGL_THREADS / WAVE - 1) / (SGL_THREADS / WAVE);
        const int64_t ws = as + (tid / WAVE) * per < ae ? as + (tid / WAVE) * per : ae;
        const int64_t we = ws + per < ae ? ws + per : ae;
        sg_walk_products<FAST, WAVE, 8, false>(
            a, b, ws, we, lane, [&](int32_t k, double, double) { atomicOr(&sgl_bits[k >> 5], 1u << (k & 31)); }, [&]() {});
        __syncthreads();
        int c = 0;
        for (int k = tid; k < nwords; k += SGL_THREADS) c += __popc(sgl_bits[k]);
        for (int off = WAVE / 2; off; off >>= 1) c += __shfl_down(c, off, WAVE);
        if (lane == 0 && c) atomicAdd(&s_tot, c);
        __syncthreads();
        if (tid == 0) cnt[i] = s_tot;
        __syncthreads();
    }
}

template <bool FAST>
__global__ __launch_bounds__(SGL_THREADS) void sg_lds_numeric_kernel(MatView a, MatView b, const int32_t *__restrict__ list,
                                                                    int32_t n_rows, const int32_t *__restrict__ c_rp,
                                                                    int32_t *__restrict__ c_ci, double *__restrict__ c_vs,
                                                                    int32_t *__restrict__ next)
{
    extern __shared__ __align__(16) unsigned char sgl_smem[];
    double *s_work = (double *)sgl_smem;                         // SGL_W
    uint32_t *s_bits = (uint32_t *)(s_work + SGL_W);             // SGL_W / 32
    __shared__ int32_t s_wsum[SGL_THREADS / WAVE];
    __shared__ int32_t s_q;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int32_t nc = b.ncols;
    for (;;) {
        if (tid == 0) s_q = atomicAdd(next, 1);
        __syncthreads();
        const int q = s_q;
        if (q >= n_rows) break;
        const int i = list[q];
        const int64_t as = rp_at<FAST>(a, i), ae = rp_at<FAST>(a, i + 1);
        int pos = c_rp[i];
        for (int32_t t0 = 0; t0 < nc; t0 += SGL_W) {
            const int32_t t1 = t0 + SGL_W < nc ? t0 + SGL_W : nc;
            for (int k = tid; k < SGL_W; k += SGL_THREADS) s_work[k] = 0.0;
            for (int k = tid; k < SGL_W / 32; k += SGL_THREADS) s_bits[k] = 0;
            __syncthreads();
            sg_walk_products<FAST, SGL_THREADS, 8, true>(
                a, b, as, ae, tid,
                [&](int32_t k, double av, double bv) {
                    if (k >= t0 && k < t1) {
                        atomicAdd(&s_work[k - t0], sg_mul<FAST>(a, b, av, bv));
                        atomicOr(&s_bits[(k - t0) >> 5], 1u << ((k - t0) & 31));
                    }
                },
                [&]() { __syncthreads(); });
            __syncthreads();
            // ascending compaction of the tile, 1024 columns at a time
            for (int32_t k0 = 0; k0 < t1 - t0; k0 += SGL_THREADS) {
                const int32_t k = k0 + tid;
                const bool occ = k < t1 - t0 && ((s_bits[k >> 5] >> (k & 31)) & 1u);
                const unsigned long long bal = __ballot(occ);
                const int below = __popcll(bal & ((1ull << lane) - 1ull));
                if (lane == 0) s_wsum[w] = __popcll(bal);
                __syncthreads();
                int woff = 0, tot = 0;
#pragma unroll
                for (int u = 0; u < SGL_THREADS / WAVE; u++) {
                    if (u < w) woff += s_wsum[u];
                    tot += s_wsum[u];
                }
                if (occ) {
                    c_ci[pos + woff + below] = t0 + k;
                    c_vs[pos + woff + below] = s_work[k];
                }
                pos += tot;
                __syncthreads();
            }
        }
    }
}

// Rows with many products but few distinct output columns (n_out <= SGB_CAP, known from the symbolic pass):
// an 8192-slot hash table in LDS (keys by atomicCAS, values by ds_add_f64), then compaction and a bitonic
// sort by column -- the workgroup-per-row hash path above, sized by the OUTPUT count instead of the product
// count, with persistent 1024-thread workgroups that claim rows one at a time.  (These rows used to go
// through the HBM work rows: 9.8 of the 12 ms of a 200k x 200k power-law product.)
constexpr int SGB_SLOTS = 8192;
constexpr int SGB_CAP = 4096;
template <bool FAST>
__global__ __launch_bounds__(SGL_THREADS) void sg_hash_big_kernel(MatView a, MatView b, const int32_t *__restrict__ list,
                                                                 int32_t n_rows, const int32_t *__restrict__ c_rp,
                                                                 int32_t *__restrict__ c_ci, double *__restrict__ c_vs,
                                                                 int32_t *__restrict__ next)
{
    extern __shared__ __align__(16) unsigned char sgb_smem[];
    double *s_val = (double *)sgb_smem;                          // SGB_SLOTS
    double *s_cv = s_val + SGB_SLOTS;                            // SGB_CAP
    int32_t *s_key = (int32_t *)(s_cv + SGB_CAP);                // SGB_SLOTS
    int32_t *s_ck = s_key + SGB_SLOTS;                           // SGB_CAP
    __shared__ int32_t s_wsum[SGL_THREADS / WAVE];
    __shared__ int32_t s_q;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    for (;;) {
        if (tid == 0) s_q = atomicAdd(next, 1);
        __syncthreads();
        const int q = s_q;
        if (q >= n_rows) break;
        const int i = list[q];
        for (int sl = tid; sl < SGB_SLOTS; sl += SGL_THREADS) {
            s_key[sl] = -1;
            s_val[sl] = 0.0;
        }
        __syncthreads();
        const int64_t as = rp_at<FAST>(a, i), ae = rp_at<FAST>(a, i + 1);
        sg_walk_products<FAST, SGL_THREADS, 8, true>(
            a, b, as, ae, tid,
            [&](int32_t k, double av, double bv) {
                uint32_t slot = ((uint32_t)k * 2654435761u) >> 19;      // 13 bits
                for (;;) {
                    const int32_t old = atomicCAS(&s_key[slot], -1, k);
                    if (old == -1 || old == k) {
                        atomicAdd(&s_val[slot], sg_mul<FAST>(a, b, av, bv));
                        break;
                    }
                    slot = (slot + 1) & (SGB_SLOTS - 1);
                }
            },
            [&]() { __syncthreads(); });
        __syncthreads();
        // compact the occupied slots, then bitonic sort by column
        int base = 0;
        for (int s0 = 0; s0 < SGB_SLOTS; s0 += SGL_THREADS) {
            const int sl = s0 + tid;
            const bool occ = s_key[sl] != -1;
            const unsigned long long bal = __ballot(occ);
            const int below = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) s_wsum[w] = __popcll(bal);
            __syncthreads();
            int woff = 0, tot = 0;
#pragma unroll
            for (int u = 0; u < SGL_THREADS / WAVE; u++) {
                if (u < w) woff += s_wsum[u];
                tot += s_wsum[u];
            }
            if (occ) {
                s_ck[base + woff + below] = s_key[sl];
                s_cv[base + woff + below] = s_val[sl];
            }
            base += tot;
            __syncthreads();
        }
        const int n = base;                   // == c_rp[i + 1] - c_rp[i] <= SGB_CAP
        int np2 = 1;
        while (np2 < n) np2 <<= 1;
        for (int t = n + tid; t < np2; t += SGL_THREADS) s_ck[t] = 0x7fffffff;
        __syncthreads();
        for (int size = 2; size <= np2; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = tid; t < np2 / 2; t += SGL_THREADS) {
                    const int lo = 2 * t - (t & (stride - 1));
                    const int hi = lo + stride;
                    const bool up = (lo & size) == 0;
                    const int32_t kl = s_ck[lo], kh = s_ck[hi];
                    if ((kl > kh) == up) {
                        s_ck[lo] = kh;
                        s_ck[hi] = kl;
                        const double vl = s_cv[lo];
                        s_cv[lo] = s_cv[hi];
                        s_cv[hi] = vl;
                    }
                }
                __syncthreads();
            }
        }
        const int32_t o = c_rp[i];
        for (int t = tid; t < n; t += SGL_THREADS) {
            c_ci[o + t] = s_ck[t];
            c_vs[o + t] = s_cv[t];
        }
        __syncthreads();
    }
}

// ---- rows with many products per output column: column strips, one wavefront each ---------------------------------
// The heavy rows of an A B^T block of a ratings matrix have 10^5..10^6 products over 10^4 columns, and up to 7000 entries
// of A.  With a whole workgroup on one row, ordered accumulation is a chain of |A_i| steps with a barrier each, every step
// waiting for its own memory round trips: the longest row alone took 5 ms of the 8 ms call.  Here the row's output columns
// are cut into STRIPS of SGS_W columns and every (row, strip) is a unit of work for ONE wavefront, which keeps the
// strip's float64 accumulators in 9.5 KiB of LDS: sixteen units per CU, no barriers (one wavefront's LDS operations
// complete in program order, so each column receives its products in ascending jj -- the reference's order), and a heavy
// row spreads over as many wavefronts as it has strips.  A unit must find, for every A entry (i, j), the entries of B_j
// inside its strip: B's rows are strictly ascending (checked; A B^T's right operand comes out of the transpose that
// way), so these are a sub-range of B_j, and the sub-range bounds of every (A entry, strip boundary) are found once by
// binary search (sg_strip_table) -- the table is 4 (S+1) bytes per A entry, which is why only rows with at least
// SGS_MIN_PER_CELL products per (A entry, strip) on average take this path.
#ifndef CSRK_SGS_W
#define CSRK_SGS_W 832
#endif
// columns per strip (13 x 64): 6.5 KiB of sums + a tag byte per column, 20 wavefronts per CU.  Measured on the ratings
// blocks (2000 x 20000^T / 500 x 5000^T): 544 -> 1.85 / 1.08 ms, 704 -> 1.68 / 1.17, 832 -> 1.58 / 1.13, 960 -> 1.70 / 1.27,
// 1088 -> 1.65 / 1.23: narrower strips split the hot columns' ordered adds over more units, wider ones need fewer
// sub-ranges per product.
constexpr int SGS_W = CSRK_SGS_W;
constexpr int SGS_WAVES_PER_CU = 16 * 1088 / SGS_W;
constexpr int SGS_CHUNKS = SGS_W / WAVE;
constexpr int SGS_MAX_S = 256;              // strips per row (wider products fall back to the workgroup paths)
#ifndef CSRK_SGS_G
#define CSRK_SGS_G 8
#endif
constexpr int SGS_G = CSRK_SGS_G;         // sub-ranges requested per round trip
constexpr int SGS_MIN_PER_CELL = 4;
#ifndef CSRK_SGS_MIN_PER_UNIT
#define CSRK_SGS_MIN_PER_UNIT 512
#endif
constexpr int SGS_MIN_PER_UNIT = CSRK_SGS_MIN_PER_UNIT;     // products per (row, strip) on average
constexpr int SGS_HEAVY_J = 1024;           // rows with this many A entries are scheduled first (longest chains)
constexpr int64_t SGS_TABLE_BUDGET = 2ll << 30;
constexpr int64_t SGS_TEMP_BUDGET = 24ll << 30;      // bytes of compacted strips held between the one pass and the copy

// bad[0] = 1 unless every row of the matrix is strictly ascending in its columns
__global__ void sg_sorted_check(const int32_t *__restrict__ rp, const int32_t *__restrict__ ci, int32_t nrows, int64_t nnz,
                                int32_t *__restrict__ bad)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (k >= nnz) return;
    if (ci[k] > ci[k - 1]) return;
    // a descent is fine only across a row boundary: k must be the first entry of its row
    int32_t lo = 0, hi = nrows;            // largest r with rp[r] <= k
    while (hi - lo > 1) {
        const int32_t mid = lo + (hi - lo) / 2;
        if ((int64_t)rp[mid] <= k) lo = mid; else hi = mid;
    }
    if ((int64_t)rp[lo] != k) bad[0] = 1;
}

// Where every row with products goes (route[i]; 0 = the sub-wavefront kernels, <= SG_WAVE_CAP products, or the workgroup hash
// kernel, <= SG_CAP):
//   1 strips          more than SG_CAP products, at least SGS_MIN_PER_CELL per (A entry, strip) on average, strips usable
//                     (counters[1]; their A entries numbered from counters[2]);
//   2 expand-sort-compress   more than esc_min products and not (more than SG_CAP and that dense) (counters[3]), when it
//                     is on (esc_min >= 0);
//   3 the round-1 heavy-row paths (counters[0]): the other rows with more than SG_CAP products.
// Two launches: strip rows with long A rows first (pass 0), so the longest chains start first.
constexpr int SG_LIST_THREADS = 1024;
template <bool FAST>
__global__ __launch_bounds__(SG_LIST_THREADS) void sg_list_rows(MatView a, const int64_t *__restrict__ ub, int32_t nrows, int32_t s_dense, int strips_ok,
                             int64_t esc_min, int pass, unsigned char *__restrict__ route, int32_t *__restrict__ list_large,
                             int32_t *__restrict__ list_strip, int32_t *__restrict__ ebase, int32_t *__restrict__ list_esc,
                             int32_t *__restrict__ counters)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t u = i < nrows ? ub[i] : 0;
    const int64_t J = u > 0 ? rp_at<FAST>(a, i + 1) - rp_at<FAST>(a, i) : 0;
    // dense enough for the strips: enough products per table cell, and per (row, strip) unit -- with 241 strips a row of
    // 10^4 products would be 241 units of 40 products each
    const bool dense = s_dense > 0 && u >= (int64_t)SGS_MIN_PER_CELL * J * s_dense && u >= (int64_t)SGS_MIN_PER_UNIT * s_dense;
    int to = 0;
    if (u > SG_CAP && dense && strips_ok) to = ((J >= SGS_HEAVY_J) == (pass == 0)) ? 1 : 0;
    else if (pass != 0 || u <= 0) to = 0;
    else if (esc_min >= 0 && u > esc_min && !(dense && u > SG_CAP)) to = 2;
    else if (u > SG_CAP) to = 3;
    // one atomic per WORKGROUP and list: atomics on one address are served one after the other (~10 ns each) -- 10^5 rows
    // appending one by one took 59 us, 10^6 rows with one atomic per wavefront 176 us of a 2.9-ms product
    if (to == 1) {           // (few rows, and each needs its own A-entry base)
        const int32_t q = atomicAdd(&counters[1], 1);
        list_strip[q] = (int32_t)i;
        ebase[q] = atomicAdd(&counters[2], (int32_t)J);
        route[i] = 1;
    }
    __shared__ int32_t s_cnt[2][SG_LIST_THREADS / WAVE], s_base[2];
    const int w = threadIdx.x / WAVE;
    unsigned long long mk[2];
#pragma unroll
    for (int t = 2; t <= 3; t++) {
        mk[t - 2] = __ballot(to == t);
        if (lane == 0) s_cnt[t - 2][w] = (int32_t)__popcll(mk[t - 2]);
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        int32_t tot = 0;
        for (int k = 0; k < SG_LIST_THREADS / WAVE; k++) {
            const int32_t c = s_cnt[threadIdx.x][k];
            s_cnt[threadIdx.x][k] = tot;          // exclusive prefix over the workgroup's wavefronts
            tot += c;
        }
        s_base[threadIdx.x] = tot ? atomicAdd(&counters[threadIdx.x == 0 ? 3 : 0], tot) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int t = 2; t <= 3; t++) {
        if (to == t) {
            (t == 2 ? list_esc : list_large)[s_base[t - 2] + s_cnt[t - 2][w] + __popcll(mk[t - 2] & ((1ull << lane) - 1ull))] = (int32_t)i;
            route[i] = (unsigned char)t;
        }
    }
}

__global__ __launch_bounds__(256) void sg_strip_expand(const int32_t *__restrict__ a_rp, const int32_t *__restrict__ list_strip,
                                                       const int32_t *__restrict__ ebase, int32_t *__restrict__ emap)
{
    const int32_t i = list_strip[blockIdx.x], as = a_rp[i], J = a_rp[i + 1] - as, eb = ebase[blockIdx.x];
    for (int32_t t = threadIdx.x; t < J; t += 256) emap[eb + t] = as + t;
}

// T[s * E + e] = first entry of B_j (j = the column of A entry emap[e]) whose column is >= s * SGS_W, s = 0..S
__global__ __launch_bounds__(256) void sg_strip_table(const int32_t *__restrict__ a_ci, const int32_t *__restrict__ b_rp,
                                                      const int32_t *__restrict__ b_ci, const int32_t *__restrict__ emap,
                                                      int32_t E, int32_t S, int32_t *__restrict__ T)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)E * (S + 1)) return;
    const int32_t s = (int32_t)(idx / E), e = (int32_t)(idx - (int64_t)s * E);
    const int32_t j = a_ci[emap[e]];
    int32_t lo = b_rp[j], hi = b_rp[j + 1];
    if (s == S) lo = hi;
    else if (s > 0) {
        const int32_t target = s * SGS_W;
        while (lo < hi) {
            const int32_t mid = lo + (hi - lo) / 2;
            if (b_ci[mid] < target) lo = mid + 1; else hi = mid;
        }
    }
    T[idx] = lo;
}

// Diagnostic build only (-DCSRK_SG_STAMPS): per-unit (cycles, |A_i|, positions, chunks, start time) of the numeric strip kernel
#ifdef CSRK_SG_STAMPS
__device__ unsigned long long g_sg_stamps[65536 * 8];
#define SG_STAMP(I) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[I] += now_ - st_last; st_last = now_; }
#else
#define SG_STAMP(I)
#endif

// Strip pointers of every row of B: SP[j * (S + 1) + s] = first entry of B_j whose column is >= s * SGS_W (one wavefront
// per row, one pass over B), and the table gathered from them -- 0.03 ms where the binary searches of sg_strip_table take
// 0.20 on a ratings block; used when B has few enough rows for the array (else the searches).
__global__ __launch_bounds__(256) void sg_strip_rowstarts(const int32_t *__restrict__ b_rp, int32_t nrows, int32_t S,
                                                          uint32_t *__restrict__ start_bits, int32_t *__restrict__ SP)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nrows) return;
    const int32_t rs = b_rp[j], re = b_rp[j + 1];
    if (rs == re) {
        for (int32_t t = 0; t <= S; t++) SP[j * (S + 1) + t] = rs;       // an empty row: every boundary at its (empty) extent
    } else {
        atomicOr(&start_bits[rs >> 5], 1u << (rs & 31));
    }
}

// one thread per entry of B; only the entries that open a strip (and each row's last) look their row up
__global__ __launch_bounds__(256) void sg_strip_rowptrs(const int32_t *__restrict__ b_rp, const int32_t *__restrict__ b_ci,
                                                        int32_t nrows, int64_t nnz, int32_t S,
                                                        const uint32_t *__restrict__ start_bits, int32_t *__restrict__ SP)
{
    const int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= nnz) return;
    const bool first = (start_bits[pos >> 5] >> (pos & 31)) & 1u;
    const bool last = pos + 1 == nnz || ((start_bits[(pos + 1) >> 5] >> ((pos + 1) & 31)) & 1u);
    const int32_t sc = b_ci[pos] / SGS_W, sb = first ? -1 : b_ci[pos - 1] / SGS_W;
    if (sc == sb && !last) return;
    int32_t lo = 0, hi = nrows;            // the row of pos: the largest j with b_rp[j] <= pos (empty rows in between skipped)
    while (hi - lo > 1) {
        const int32_t mid = lo + (hi - lo) / 2;
        if ((int64_t)b_rp[mid] <= pos) lo = mid; else hi = mid;
    }
    int32_t *sp = SP + (int64_t)lo * (S + 1);
    for (int32_t t = sb + 1; t <= sc; t++) sp[t] = (int32_t)pos;          // the first entry at or beyond these boundaries
    if (last)
        for (int32_t t = sc + 1; t <= S; t++) sp[t] = (int32_t)pos + 1;
}

__global__ __launch_bounds__(256) void sg_strip_table_gather(const int32_t *__restrict__ a_ci, const int32_t *__restrict__ emap,
                                                             const int32_t *__restrict__ SP, int32_t E, int32_t S,
                                                             int32_t *__restrict__ T)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)E * (S + 1)) return;
    const int32_t s = (int32_t)(idx / E), e = (int32_t)(idx - (int64_t)s * E);
    T[idx] = SP[(int64_t)a_ci[emap[e]] * (S + 1) + s];
}

__device__ __forceinline__ double sg_readlane_f64(double x, int r)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), r), __builtin_amdgcn_readlane(__double2loint(x), r));
}

// SGS_G chunks (64 positions each) of a unit's products, requested from memory and not yet applied
template <bool NUMERIC>
struct SgsGroup {
    int32_t r[SGS_G], k[SGS_G];                    // per lane: A entry (of the 64 in the batch), column (-1: no product)
    double av[NUMERIC ? SGS_G : 1], v[NUMERIC ? SGS_G : 1];
    bool one[SGS_G];                               // uniform: the chunk lies inside ONE sub-range
};

// One wavefront per (strip row, strip).  MODE 2 (the default): ONE pass -- float64 accumulators and a tag byte per column
// (non-zero = the column received a product; entries that cancel to 0.0 stay, like the reference's), compacted into a
// temporary region sized by the unit's product count (t_off, from sg_strip_cap); the counts give the row pointers and
// sg_strip_copy moves the regions into C.  (A symbolic walk first cost 0.69 ms of a ratings block's 2.3; the copy 0.1.)
// MODE 0 / 1: the two-pass form for products whose temporary would not fit the budget -- symbolic: a flag byte per
// column -> the strip's occupancy words (occ, 64 columns each) and count; numeric: compacted through those words.
template <int MODE>
__global__ __launch_bounds__(WAVE) void sg_strip_kernel(const int32_t *__restrict__ a_rp, const double *__restrict__ a_vs,
                                                       const int32_t *__restrict__ b_ci, const double *__restrict__ b_vs,
                                                       int64_t b_last, const int32_t *__restrict__ list_strip,
                                                       const int32_t *__restrict__ ebase, int32_t n_strip, int32_t S, int32_t E,
                                                       const int32_t *__restrict__ T, int32_t *__restrict__ cnt_s,
                                                       unsigned long long *__restrict__ occ, int32_t *__restrict__ cnt,
                                                       const int32_t *__restrict__ c_rp, int32_t *__restrict__ c_ci,
                                                       double *__restrict__ c_vs, const int64_t *__restrict__ t_off,
                                                       int32_t *__restrict__ next)
{
    constexpr bool NUMERIC = MODE != 0, FUSED = MODE == 2;
    __shared__ double s_work[NUMERIC ? SGS_W : 1];
    __shared__ unsigned char s_flag[NUMERIC ? 4 : SGS_W];
    __shared__ unsigned char s_tag[NUMERIC ? SGS_W : 4];      // numeric: lane number + 1 of the last lane that added to the column
    __shared__ int32_t s_mark[WAVE];                          // the chunk search's markers
    const int lane = threadIdx.x;
    for (int c = 0; c < SGS_CHUNKS; c++) {
        if (NUMERIC) {
            s_work[c * WAVE + lane] = 0.0;
            s_tag[c * WAVE + lane] = 0;
        } else {
            s_flag[c * WAVE + lane] = 0;
        }
    }
    __syncthreads();
    const int64_t n_units = (int64_t)n_strip * S;
    int32_t u = 0;
    if (lane == 0) u = atomicAdd(next, 1);
    u = __builtin_amdgcn_readfirstlane(u);
    while (u < n_units) {
        int32_t u_next = 0;
        if (lane == 0) u_next = atomicAdd(next, 1);          // (needed only after this unit)
        const int32_t q = u / S, s = u - q * S;
        const int32_t i = list_strip[q], as = a_rp[i], J = a_rp[i + 1] - as;
        const int32_t *__restrict__ t_lo = T + (int64_t)s * E + ebase[q];
        const int32_t *__restrict__ t_hi = t_lo + E;
        const int32_t c0 = s * SGS_W;
#ifdef CSRK_SG_STAMPS
        const unsigned long long st0 = __builtin_amdgcn_s_memtime();
        unsigned long long st_pos = 0, st_chunks = 0, st_last = st0, st_acc[3] = {0, 0, 0};
#endif
        // the (sub-range, a_ij) of 64 A entries at a time, one per lane, the next 64 requested before these are used
        int32_t lo_n, hi_n;
        double av_n;
        auto fetch = [&](int32_t e0) {
            const int32_t e = e0 + lane;
            const bool in = e < J;
            const int32_t ec = in ? e : J - 1;
            const int32_t l = t_lo[ec], h = t_hi[ec];
            lo_n = l;
            hi_n = in ? h : l;
            av_n = NUMERIC ? a_vs[as + ec] : 0.0;
        };
        auto apply_group = [&](const SgsGroup<NUMERIC> &gr) {
#pragma unroll
            for (int g = 0; g < SGS_G; g++) {
                const bool in = gr.k[g] >= 0;
                if constexpr (!NUMERIC) {
                    if (in) s_flag[gr.k[g] - c0] = 1;
                    continue;
                }
                const double prod = __dmul_rn(gr.av[NUMERIC ? g : 0], gr.v[NUMERIC ? g : 0]);
                if (gr.one[g]) {
                    if (in) {
                        atomicAdd(&s_work[gr.k[g] - c0], prod);
                        if (FUSED) s_tag[gr.k[g] - c0] = 1;
                    }
                    continue;
                }
                // Lanes of one A entry hold distinct columns; two entries of the chunk may hold the same column, and the
                // order of same-address lanes inside one LDS instruction is the hardware's.  Only those lanes need an
                // order: every lane tags its column with its lane number + 1 and reads the tag back -- a lane that finds
                // another's number shares its column, and overwrites the tag with 0xff so that the lane that won
                // learns it too.  Columns seen once take ONE LDS add together; the others go entry by entry, ascending.
                // (One add per entry for all lanes cost the heaviest unit of a ratings block -- 7000 short sub-ranges,
                // ~30 entries per chunk, of which ~4 lanes clash -- 1.4 of its 2.5 M clocks.)
                unsigned char *tag = s_tag;
                const int32_t col = in ? gr.k[g] - c0 : 0;
                if (in) tag[col] = (unsigned char)(lane + 1);
                asm volatile("" ::: "memory");
                const bool lost = in && tag[col] != (unsigned char)(lane + 1);
                asm volatile("" ::: "memory");
                if (lost) tag[col] = 0xff;
                asm volatile("" ::: "memory");
                const bool dup = in && (lost || tag[col] == 0xff);
                asm volatile("" ::: "memory");
                if (in && !dup) atomicAdd(&s_work[col], prod);
                // (scalar unit only: an entry's lanes are a run of the chunk, `starts` marks where the runs begin)
                const unsigned long long starts = __builtin_amdgcn_ballot_w64(gr.r[g] != wave_up1_i32(gr.r[g], -1));
                unsigned long long rem = __builtin_amdgcn_ballot_w64(dup);
                while (rem) {                                           // uniform: the entries that hold clashing lanes, ascending
                    const int l = __builtin_ctzll(rem);
                    const unsigned long long above = starts & ~((2ull << l) - 1ull);      // run starts after lane l
                    const unsigned long long upto = above ? (1ull << __builtin_ctzll(above)) - 1ull : ~0ull;
                    if (__builtin_amdgcn_inverse_ballot_w64(rem & upto)) atomicAdd(&s_work[col], prod);
                    rem &= ~upto;
                }
            }
        };
        SgsGroup<NUMERIC> pend;
        bool have_pend = false;
        fetch(0);
        for (int32_t e0 = 0; e0 < J; e0 += WAVE) {
            const int32_t lo = lo_n, len = hi_n - lo_n;
            const double av = av_n;
            fetch(e0 + WAVE < J ? e0 + WAVE : e0);
            // The sub-ranges of these 64 A entries laid end to end: position p belongs to the last entry r whose start
            // is <= p and is B entry base_r + p.  64 positions (a chunk) are requested per load, SGS_G chunks per round
            // trip, whatever the sub-ranges' lengths (requesting sub-range by sub-range, the heaviest row's 7000 short
            // sub-ranges and its long ones' passes were 1100 dependent round trips: 1.5 ms for one unit).
            const int32_t start = wave_exscan_i32(len, lane);
            const int32_t total = __builtin_amdgcn_readlane(start + len, WAVE - 1);
            const int32_t base = lo - start;
            const int32_t n_chunks = (total + WAVE - 1) / WAVE;
#ifdef CSRK_SG_STAMPS
            st_pos += total;
            st_chunks += n_chunks;
#endif
            const int32_t end = start + len;
            SG_STAMP(0)
            for (int32_t cb = 0; cb < n_chunks; cb += SGS_G) {      // uniform
                // request this group's entries, THEN apply the group requested one step earlier: two groups are in
                // flight, across the 64-entry batches too (a group per round trip left the heaviest unit of a ratings
                // block -- 7000 A entries, 1400 chunks -- waiting 2 us per group: 1.1 ms for one wavefront)
                SgsGroup<NUMERIC> cur;
#pragma unroll
                for (int g = 0; g < SGS_G; g++) {
                    const int32_t p0 = (cb + g) * WAVE;        // (chunks past the end: no position is valid)
                    const bool in = p0 + lane < total;
                    // the entry that holds position p0: the last non-empty one that starts at or before it
                    const unsigned long long before = __ballot(len > 0 && start <= p0);
                    const int cur_e = before ? 63 - __builtin_clzll(before) : 0;
                    const int32_t cur_end = __builtin_amdgcn_readlane(end, cur_e);
                    const int32_t p1 = p0 + WAVE < total ? p0 + WAVE : total;
                    cur.one[g] = p1 <= cur_end;
                    int32_t bs;
                    if (cur.one[g]) {
                        // 92 % of a ratings block's products: scalars, no search
                        bs = __builtin_amdgcn_readlane(base, cur_e);
                        if constexpr (NUMERIC) cur.av[g] = sg_readlane_f64(av, cur_e);
                        cur.r[g] = cur_e;
                    } else {
                        // (the lanes talk through s_mark: without the compiler fence a lane's own stores are forwarded
                        // to its load; a volatile access would do, but costs an s_waitcnt vmcnt(0) -- every chunk a
                        // memory round trip of its own)
                        s_mark[lane] = 0;
                        const int32_t rel = start - p0;
                        if (len > 0 && rel >= 0 && rel < WAVE) s_mark[rel] = lane + 1;
                        asm volatile("" ::: "memory");
                        const int32_t mk = s_mark[lane];
                        asm volatile("" ::: "memory");
                        const int32_t m = max(wave_incl_max_i32(mk), cur_e + 1);
                        const int32_t r = in ? m - 1 : 0;
                        cur.r[g] = r;
                        bs = __shfl(base, r, WAVE);
                        if constexpr (NUMERIC) cur.av[g] = __shfl(av, r, WAVE);
                    }
                    const int64_t kc = in ? (int64_t)bs + p0 + lane : b_last;      // clamped: the loads stay unconditional
                    const int32_t kl = b_ci[kc];
                    if constexpr (NUMERIC) cur.v[g] = b_vs[kc];
                    cur.k[g] = in ? kl : -1;
                }
                SG_STAMP(1)
                if (have_pend) apply_group(pend);
                pend = cur;
                have_pend = true;
                SG_STAMP(2)
            }
        }
        if (have_pend) apply_group(pend);
        __syncthreads();                                       // (one wavefront: no s_barrier, the LDS queue drains)
#ifdef CSRK_SG_STAMPS
        if (NUMERIC && lane == 0 && u < 65536) {
            const unsigned long long st1 = __builtin_amdgcn_s_memtime();
            g_sg_stamps[u * 8 + 0] = st1 - st0;
            g_sg_stamps[u * 8 + 1] = J;
            g_sg_stamps[u * 8 + 2] = st_pos;
            g_sg_stamps[u * 8 + 3] = st_chunks;
            g_sg_stamps[u * 8 + 4] = st0;
            g_sg_stamps[u * 8 + 5] = st_acc[0];
            g_sg_stamps[u * 8 + 6] = st_acc[1];
            g_sg_stamps[u * 8 + 7] = st_acc[2];
        }
#endif
        if (FUSED) {
            const int64_t base = t_off[u];
            int32_t tot = 0;
            for (int c = 0; c < SGS_CHUNKS; c++) {
                const bool there = s_tag[c * WAVE + lane] != 0;
                const unsigned long long word = __builtin_amdgcn_ballot_w64(there);
                const double x = s_work[c * WAVE + lane];
                s_work[c * WAVE + lane] = 0.0;
                s_tag[c * WAVE + lane] = 0;
                if (there) {
                    const int64_t o = base + tot + __popcll(word & ((1ull << lane) - 1ull));
                    c_ci[o] = c0 + c * WAVE + lane;
                    c_vs[o] = x;
                }
                tot += __popcll(word);
            }
            if (lane == 0) {
                cnt_s[u] = tot;
                if (tot) atomicAdd(&cnt[i], tot);
            }
        } else if (NUMERIC) {
            int32_t pre = 0;
            for (int32_t s0 = 0; s0 < s; s0 += WAVE) pre += s0 + lane < s ? cnt_s[(int64_t)q * S + s0 + lane] : 0;
            for (int off = WAVE / 2; off; off >>= 1) pre += __shfl_xor(pre, off, WAVE);
            int32_t pos = c_rp[i] + pre;
            for (int c = 0; c < SGS_CHUNKS; c++) {
                const unsigned long long word = occ[(int64_t)u * SGS_CHUNKS + c];
                const double x = s_work[c * WAVE + lane];
                s_work[c * WAVE + lane] = 0.0;
                if ((word >> lane) & 1ull) {
                    const int32_t o = pos + __popcll(word & ((1ull << lane) - 1ull));
                    c_ci[o] = c0 + c * WAVE + lane;
                    c_vs[o] = x;
                }
                pos += __popcll(word);
            }
        } else {
            int32_t tot = 0;
            for (int c = 0; c < SGS_CHUNKS; c++) {
                const unsigned long long word = __ballot(s_flag[c * WAVE + lane] != 0);
                s_flag[c * WAVE + lane] = 0;
                if (lane == 0) occ[(int64_t)u * SGS_CHUNKS + c] = word;
                tot += __popcll(word);
            }
            if (lane == 0) {
                cnt_s[u] = tot;
                if (tot) atomicAdd(&cnt[i], tot);
            }
        }
        __syncthreads();
        u = __builtin_amdgcn_readfirstlane(u_next);
    }
}

// cap[u] = min(SGS_W, products of unit u): what its compacted strip can hold at most (one wavefront per unit)
__global__ __launch_bounds__(256) void sg_strip_cap(const int32_t *__restrict__ a_rp, const int32_t *__restrict__ list_strip,
                                                    const int32_t *__restrict__ ebase, int32_t n_strip, int32_t S, int32_t E,
                                                    const int32_t *__restrict__ T, int32_t *__restrict__ cap)
{
    const int64_t u = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (u >= (int64_t)n_strip * S) return;
    const int32_t q = (int32_t)(u / S), s = (int32_t)(u - (int64_t)q * S);
    const int32_t i = list_strip[q], J = a_rp[i + 1] - a_rp[i];
    const int32_t *t_lo = T + (int64_t)s * E + ebase[q], *t_hi = t_lo + E;
    int64_t tot = 0;
    for (int32_t e = lane; e < J; e += WAVE) tot += t_hi[e] - t_lo[e];
    for (int off = WAVE / 2; off; off >>= 1) tot += __shfl_xor(tot, off, WAVE);
    if (lane == 0) cap[u] = (int32_t)(tot < SGS_W ? tot : SGS_W);
}

// the units' compacted strips -> their places in C (one wavefront per unit)
__global__ __launch_bounds__(256) void sg_strip_copy(const int32_t *__restrict__ list_strip, int32_t n_strip, int32_t S,
                                                     const int32_t *__restrict__ cnt_s, const int64_t *__restrict__ t_off,
                                                     const int32_t *__restrict__ t_ci, const double *__restrict__ t_vs,
                                                     const int32_t *__restrict__ c_rp, int32_t *__restrict__ c_ci,
                                                     double *__restrict__ c_vs)
{
    const int64_t u = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (u >= (int64_t)n_strip * S) return;
    const int32_t q = (int32_t)(u / S), s = (int32_t)(u - (int64_t)q * S);
    int32_t pre = 0;
    for (int32_t s0 = 0; s0 < s; s0 += WAVE) pre += s0 + lane < s ? cnt_s[(int64_t)q * S + s0 + lane] : 0;
    for (int off = WAVE / 2; off; off >>= 1) pre += __shfl_xor(pre, off, WAVE);
    const int64_t src = t_off[u], dst = (int64_t)c_rp[list_strip[q]] + pre;
    const int32_t n = cnt_s[u];
    for (int32_t t = lane; t < n; t += WAVE) {
        c_ci[dst + t] = t_ci[src + t];
        c_vs[dst + t] = t_vs[src + t];
    }
}

// ---- heavy rows with a wide, sparse output: expand, sort, compress ---------------------------------------------------
// Rows with more than SGB_CAP distinct output columns that are not nearly full (power-law A times power-law B) have too
// many outputs for an LDS table and too few products per column for the strips.  Their products are written out in the
// reference's order (A entries ascending, each with its row of B), as the rows of a CSR matrix P with repeated columns;
// two stable transposes (transpose.hip: an LSD radix sort that keeps the input order inside a key) bring every row of P
// into ascending column order with the products of one column still in their original order, and one thread per run of
// equal columns adds the run front to back: the reference's sums, bit for bit, without atomics.  (Before: a dense float64
// work row in HBM per workgroup, a barrier per A entry: 7.5 ms of a 200k x 200k product's 13.8.)
constexpr int64_t SGE_BUDGET_PRODUCTS = 200ll << 20;     // above this the round-1 heavy-row paths are used instead
#ifndef CSRK_SGE_MIN
#define CSRK_SGE_MIN 32
#endif
constexpr int64_t SGE_MIN = CSRK_SGE_MIN;                // rows with more products than this (and not dense enough for strips)

__global__ void sg_esc_gather(const int32_t *__restrict__ list, int32_t n, const int64_t *__restrict__ ub, int64_t *__restrict__ pu)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) pu[q] = ub[list[q]];
}

__global__ void sg_esc_rowptr(const int64_t *__restrict__ poff, int32_t n, int32_t *__restrict__ rp)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q <= n) rp[q] = (int32_t)poff[q];
}

// exclusive scan of one value per thread over a 256-thread workgroup; *total = the sum.  s_w: 4 words of LDS.
__device__ __forceinline__ int64_t sg_block_exscan(int64_t v, int tid, int64_t *s_w, int64_t *total)
{
    const int lane = tid & (WAVE - 1), w = tid / WAVE;
    int64_t inc = v;
    for (int off = 1; off < WAVE; off <<= 1) {
        const int64_t o = __shfl_up(inc, off, WAVE);
        if (lane >= off) inc += o;
    }
    __syncthreads();
    if (lane == WAVE - 1) s_w[w] = inc;
    __syncthreads();
    int64_t before = 0, tot = 0;
#pragma unroll
    for (int u = 0; u < 256 / WAVE; u++) {
        if (u < w) before += s_w[u];
        tot += s_w[u];
    }
    *total = tot;
    return before + inc - v;
}

template <bool FAST>
__global__ __launch_bounds__(256) void sg_esc_expand(MatView a, MatView b, const int32_t *__restrict__ list,
                                                     const int64_t *__restrict__ poff, int32_t *__restrict__ p_ci,
                                                     double *__restrict__ p_vs)
{
    __shared__ int64_t s_off[256], s_bs[256], s_w[256 / WAVE];
    __shared__ double s_av[256];
    const int tid = threadIdx.x;
    const int32_t i = list[blockIdx.x];
    const int64_t as = rp_at<FAST>(a, i), ae = rp_at<FAST>(a, i + 1);
    int64_t out = poff[blockIdx.x];
    for (int64_t base = as; base < ae; base += 256) {
        const int64_t jj = base + tid;
        int64_t len = 0, bs = 0;
        double av = 0.0;
        if (jj < ae) {
            const int32_t j = a.ci[jj];
            bs = rp_at<FAST>(b, j);
            len = rp_at<FAST>(b, j + 1) - bs;
            av = val_at<FAST>(a, jj);
        }
        int64_t total;
        const int64_t off = sg_block_exscan(len, tid, s_w, &total);
        s_off[tid] = off;
        s_bs[tid] = bs;
        s_av[tid] = av;
        __syncthreads();
        for (int64_t p = tid; p < total; p += 256) {
            int lo = 0, hi = 256;                      // the last A entry of the batch whose products start at or before p
            while (hi - lo > 1) {
                const int mid = (lo + hi) / 2;
                if (s_off[mid] <= p) lo = mid; else hi = mid;
            }
            const int64_t kk = s_bs[lo] + (p - s_off[lo]);
            p_ci[out + p] = b.ci[kk];
            p_vs[out + p] = sg_mul<FAST>(a, b, s_av[lo], val_at<FAST>(b, kk));
        }
        out += total;
        __syncthreads();
    }
}

// Sorted product matrix -> rows of C: one output entry per run of equal columns inside a row, the run added front to back.
// Position-parallel (a workgroup per row left the longest row's 10^5 products to 256 threads: 1 ms): tiles of 1024
// positions count their run starts (WRITE = false), the counts are scanned, and the second launch numbers the runs --
// run g of the product matrix is entry g - first_run[q] of row q's output (first_run = scan of the symbolic counts).
constexpr int SGE_TILE = 1024;
template <int MODE>      // 0: count the run starts of every tile, 1: number the runs (first run of every row), 2: write C
__global__ __launch_bounds__(256) void sg_esc_runs(const int32_t *__restrict__ list, int32_t n_q, const int32_t *__restrict__ p_rp,
                                                   const int32_t *__restrict__ p_ci, const double *__restrict__ p_vs,
                                                   int32_t n_pos, int32_t *__restrict__ tile_cnt,
                                                   const int32_t *__restrict__ tile_off, int32_t *__restrict__ first_run,
                                                   const int32_t *__restrict__ c_rp, int32_t *__restrict__ c_ci,
                                                   double *__restrict__ c_vs)
{
    __shared__ int64_t s_w[256 / WAVE];
    const int tid = threadIdx.x;
    const int32_t t0 = blockIdx.x * SGE_TILE + tid * 4;
    // the row of position t0: the last q with p_rp[q] <= t0.  Two threads bracket the tile's rows with searches over all
    // rows; the others search inside the bracket (a search over 10^5 rows in every thread of three passes was most of the
    // write pass's 0.38 ms)
    __shared__ int32_t s_q[2];
    if (tid < 2) {
        const int32_t tp = tid == 0 ? blockIdx.x * SGE_TILE : (blockIdx.x * SGE_TILE + SGE_TILE - 1 < n_pos - 1 ? blockIdx.x * SGE_TILE + SGE_TILE - 1 : n_pos - 1);
        int32_t lo = 0, hi = n_q;
        while (hi - lo > 1) {
            const int32_t mid = lo + (hi - lo) / 2;
            if (p_rp[mid] <= tp) lo = mid; else hi = mid;
        }
        s_q[tid] = lo;
    }
    __syncthreads();
    int32_t lo = s_q[0], hi = s_q[1] + 1;
    while (hi - lo > 1) {
        const int32_t mid = lo + (hi - lo) / 2;
        if (p_rp[mid] <= t0) lo = mid; else hi = mid;
    }
    int32_t q = lo, q_end = t0 < n_pos ? p_rp[q + 1] : 0;
    int32_t prev = t0 > 0 && t0 <= n_pos ? p_ci[t0 - 1] : -1;
    bool start[4];
    int32_t qx[4], kx[4], kin[4] = {-1, -1, -1, -1};
    double vx[4] = {0.0, 0.0, 0.0, 0.0};
    // the thread's four positions with 16-byte loads (the arrays come from the pool: 256-byte aligned)
    if (t0 + 3 < n_pos) {
        const int4 k4 = *reinterpret_cast<const int4 *>(p_ci + t0);
        kin[0] = k4.x, kin[1] = k4.y, kin[2] = k4.z, kin[3] = k4.w;
        if (MODE == 2) {
            const double2 va = *reinterpret_cast<const double2 *>(p_vs + t0), vb = *reinterpret_cast<const double2 *>(p_vs + t0 + 2);
            vx[0] = va.x, vx[1] = va.y, vx[2] = vb.x, vx[3] = vb.y;
        }
    } else {
        for (int x = 0; x < 4; x++)
            if (t0 + x < n_pos) {
                kin[x] = p_ci[t0 + x];
                if (MODE == 2) vx[x] = p_vs[t0 + x];
            }
    }
    int n = 0;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        const int32_t t = t0 + x;
        start[x] = false;
        qx[x] = q;
        kx[x] = -1;
        if (t < n_pos) {
            while (t >= q_end) {                       // (rows without products are skipped)
                q++;
                q_end = p_rp[q + 1];
            }
            const int32_t k = kin[x];
            start[x] = t == p_rp[q] || k != prev;
            qx[x] = q;
            kx[x] = k;
            prev = k;
            n += start[x] ? 1 : 0;
        }
    }
    // (write pass: the tile's values and run-start flags go through LDS, so that a run that leaves its thread's four
    // positions is still summed without a look at memory -- with a dependent load per such run nearly every wavefront
    // waited two memory round trips: 0.33 ms against the counting pass's 0.07.  Only a run that leaves the TILE reads on.)
    __shared__ unsigned char s_f[MODE == 2 ? SGE_TILE : 1];
    __shared__ double s_v[MODE == 2 ? SGE_TILE : 1];
    __shared__ int32_t s_ok[MODE == 2 ? SGE_TILE : 1], s_oq[MODE == 2 ? SGE_TILE : 1];
    __shared__ double s_ov[MODE == 2 ? SGE_TILE : 1];
    if (MODE == 2) {
#pragma unroll
        for (int x = 0; x < 4; x++) {
            s_f[tid * 4 + x] = (start[x] || t0 + x >= n_pos) ? 1 : 0;      // (positions past the end close the last run)
            s_v[tid * 4 + x] = vx[x];
        }
    }
    int64_t total;
    const int64_t before = sg_block_exscan(n, tid, s_w, &total);
    if (MODE == 0) {
        if (tid == 0) tile_cnt[blockIdx.x] = (int32_t)total;
        return;
    }
    int32_t g = tile_off[blockIdx.x] + (int32_t)before;
#pragma unroll
    for (int x = 0; x < 4; x++) {
        if (!start[x]) continue;
        const int32_t t = t0 + x, e = p_rp[qx[x] + 1];
        if (MODE == 1) {
            if (t == p_rp[qx[x]]) first_run[qx[x]] = g;      // (every row here has products: its first position is a run start)
            g++;
            continue;
        }
        double sum = 0.0 + vx[x];                     // (the reference's work[k] starts from +0.0: -0.0 products)
        int pl = tid * 4 + x + 1;
        for (; pl < SGE_TILE && !s_f[pl]; pl++) sum += s_v[pl];
        if (pl == SGE_TILE)                           // the run may go on in the next tile
            for (int32_t u = blockIdx.x * SGE_TILE + SGE_TILE; u < e && p_ci[u] == kx[x]; u++) sum += p_vs[u];
        // staged through LDS by run number, so that consecutive lanes store consecutive entries of C below
        const int li = g - tile_off[blockIdx.x];
        s_ok[li] = kx[x];
        s_ov[li] = sum;
        s_oq[li] = qx[x];
        g++;
    }
    if (MODE == 2) {
        __syncthreads();
        for (int li = tid; li < (int)total; li += 256) {
            // (first_run holds c_rp[row] - the row's first run: sg_esc_rowbase)
            const int64_t o = (int64_t)first_run[s_oq[li]] + tile_off[blockIdx.x] + li;
            c_ci[o] = s_ok[li];
            c_vs[o] = s_ov[li];
        }
    }
}

// first_run[q] <- c_rp[row q] - first_run[q]: where run g of the sorted products goes in C is then first_run[q] + g (three
// dependent gathers per output entry in the write pass were 0.2 of its 0.42 ms)
__global__ void sg_esc_rowbase(const int32_t *__restrict__ list, int32_t n, const int32_t *__restrict__ c_rp,
                               int32_t *__restrict__ first_run)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) first_run[q] = c_rp[list[q]] - first_run[q];
}

// cnt[row] = runs of the row = first_run of the next row (the total after the last) - its own
__global__ void sg_esc_rowcnt(const int32_t *__restrict__ list, int32_t n, const int32_t *__restrict__ first_run,
                              const int32_t *__restrict__ total_runs, int32_t *__restrict__ cnt)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) cnt[list[q]] = (q + 1 < n ? first_run[q + 1] : total_runs[0]) - first_run[q];
}

// large rows (list[0..n_large)) -> (a) output nearly full and narrow enough for the LDS tiles, (h) few enough
// distinct output columns for the big LDS hash table, (b) the others (HBM work rows)
__global__ void sg_split_large(const int32_t *__restrict__ list, int32_t n_large, const int32_t *__restrict__ cnt, int32_t nc, int32_t hash_cap,
                               int32_t *__restrict__ list_a, int32_t *__restrict__ list_h, int32_t *__restrict__ list_b,
                               int32_t *__restrict__ n_ahb)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_large) return;
    const int32_t i = list[q];
    const bool lds = (int64_t)cnt[i] * 8 >= nc && (int64_t)nc <= (int64_t)SGL_W * SGL_MAXTILES;
    if (lds)
        list_a[atomicAdd(&n_ahb[0], 1)] = i;
    else if (cnt[i] <= hash_cap)
        list_h[atomicAdd(&n_ahb[1], 1)] = i;
    else
        list_b[atomicAdd(&n_ahb[2], 1)] = i;
}

template <bool FAST>
static int spgemm_run(Matrix *a, Matrix *b, Matrix **out, bool b_rows_ascend, DevBuf *products)
{
    CSRK_REQUIRE(a->ncols == b->nrows, "mult_ab: A is %d x %d but B is %d x %d", a->nrows, a->ncols, b->nrows, b->ncols);
    CSRK_REQUIRE(a->val_type != CSRK_VAL_NONE && b->val_type != CSRK_VAL_NONE,
                 "mult_ab needs values on both operands (csr/kernels/numba/multiply.py:115,120)");
    const int32_t nr = a->nrows;
    if (a->nnz == 0 || b->nnz == 0) {         // no products at all (also: the walker's clamped loads need an entry of B)
        Matrix *c = nullptr;
        CSRK_TRY(new_matrix(nr, b->ncols, 0, 0, CSRK_VAL_F64, &c));
        if (hipMemset(c->d_rowptrs, 0, (size_t)(nr + 1) * 4) != hipSuccess) {
            set_error("hipMemset failed");
            delete c;
            return CSRK_ERR_HIP;
        }
        *out = c;
        return CSRK_OK;
    }
    MatView av = view_of(a), bv = view_of(b);
    DevBuf ub, cnt, list, work, mark, scratch, list_a, list_h, list_b, n_ab, next;
    int grid_lds = 256;
    unsigned grid_strip = 0;
    int32_t n_lds = 0, n_hash = 0, n_hbm = 0;
    bool lds_symbolic = false;
    int64_t scratch_len = 1;
    while (scratch_len < (int64_t)b->ncols) scratch_len <<= 1;     // padded length for the bitonic network
    CSRK_TRY(ub.alloc((size_t)(nr + 1) * 8));
    CSRK_TRY(cnt.alloc((size_t)(nr + 1) * 4));
    CSRK_TRY(list.alloc((size_t)(nr + 1) * 4));
    CSRK_HIP(hipMemset(cnt.p, 0, (size_t)(nr + 1) * 4));
    int32_t n_large = 0;
    int grid_dense = 0;
    // strip rows (FAST operands only): their list, A-entry numbering, sub-range table, per-unit counts and occupancy words
    DevBuf list_s, ebase, emap, table, cnt_s, occ_s, counters, sorted_bad, route;
    bool strip_fused = false, hash_rows = true, mid_rows = true, small_fused = false, small_any = true;
    DevBuf small_room, small_off, small_tci, small_tvs;
    DevBuf strip_off, strip_tci, strip_tvs;
    // expand-sort-compress rows: their list, the sorted product matrix (kept from the counting step to the write), run numbering
    DevBuf list_e, esc_pu, esc_off, esc_tile_cnt, esc_tile_off, esc_first_run;
    Matrix *esc_ps = nullptr;
    struct EscGuard {
        Matrix **m;
        ~EscGuard() { delete *m; }
    } esc_guard{&esc_ps};
    int32_t n_esc = 0, esc_tiles = 0;
    int64_t esc_products = 0;
    int32_t n_strip = 0, n_strip_e = 0, strips = 0;
    if (nr > 0) {
        unsigned g = (unsigned)ceil_div(nr, 256);
        // (list_s / counters are scratch here: the list of long rows and its length)
        const int64_t count_blocks = ceil_div((int64_t)nr * 8, 256);
        const int32_t long_cap = (int32_t)(ceil_div(count_blocks, (int64_t)SG_LONG_LISTS) * 32);      // 32 rows per workgroup
        DevBuf long_cnt;
        CSRK_TRY(long_cnt.alloc((size_t)(SG_LONG_LISTS + 1) * SG_LONG_STRIDE * 4));
        CSRK_TRY(counters.alloc(16));
        // (the lists of long rows, then the list of very long rows: at most nnz / SG_COUNT_VLONG of them)
        CSRK_TRY(list_s.alloc((size_t)std::max<int64_t>((int64_t)nr + 1, (int64_t)SG_LONG_LISTS * long_cap + a->nnz / SG_COUNT_VLONG + 1) * 4));
        CSRK_HIP(hipMemsetAsync(long_cnt.p, 0, (size_t)(SG_LONG_LISTS + 1) * SG_LONG_STRIDE * 4, nullptr));
        sg_count_products<FAST><<<(unsigned)count_blocks, 256>>>(av, bv, ub.as<int64_t>(), list_s.as<int32_t>(), long_cap,
                                                                long_cnt.as<int32_t>());
        CSRK_LAUNCH_CHECK();
        sg_count_products_long<FAST><<<1024, 256>>>(av, bv, ub.as<int64_t>(), list_s.as<int32_t>(), long_cap, long_cnt.as<int32_t>());
        CSRK_LAUNCH_CHECK();
        CSRK_TRY(ebase.alloc((size_t)(nr + 1) * 4));
        const char *strips_env = getenv("CSRK_SPGEMM_STRIPS");      // 0: workgroup paths only (A/B measurements)
        if (FAST && !(strips_env && atoi(strips_env) == 0)) {
            strips = (int32_t)ceil_div(b->ncols, SGS_W);
            if (strips > SGS_MAX_S || (int64_t)nr * strips > (1ll << 30)) strips = 0;
        }
        if (strips > 0 && b->nnz > 1 && !b_rows_ascend) {          // the strips need B's rows strictly ascending (our transpose's are)
            CSRK_TRY(sorted_bad.alloc(4));
            CSRK_HIP(hipMemset(sorted_bad.p, 0, 4));
            sg_sorted_check<<<(unsigned)ceil_div(b->nnz - 1, 256), 256>>>((const int32_t *)b->d_rowptrs, b->d_colinds, b->nrows,
                                                                         b->nnz, sorted_bad.as<int32_t>());
            CSRK_LAUNCH_CHECK();
            int32_t bad = 0;
            CSRK_HIP(hipMemcpy(&bad, sorted_bad.p, 4, hipMemcpyDeviceToHost));
            if (bad) strips = 0;
        }
        // expand-sort-compress takes the rows above esc_min products that are not dense enough for the strips
        // (CSRK_SPGEMM_ESC=0: off -- the fallback paths run instead, as they do beyond the sort's budget)
        const char *esc_env = getenv("CSRK_SPGEMM_ESC");
        int64_t esc_min = (esc_env && atoi(esc_env) == 0) ? -1 : SGE_MIN;
        int32_t s_dense = (int32_t)ceil_div(b->ncols, SGS_W);       // the strips' density test, whether they can be used or not
        if (s_dense > SGS_MAX_S) s_dense = 0;
        CSRK_TRY(route.alloc((size_t)nr + 1));
        CSRK_TRY(list_e.alloc((size_t)(nr + 1) * 4));
        int32_t cnts[4] = {0, 0, 0, 0};
        for (;;) {
            CSRK_HIP(hipMemset(counters.p, 0, 16));
            CSRK_HIP(hipMemset(route.p, 0, (size_t)nr + 1));
            for (int pass = 0; pass < (strips > 0 ? 2 : 1); pass++) {
                sg_list_rows<FAST><<<(unsigned)ceil_div(nr, SG_LIST_THREADS), SG_LIST_THREADS>>>(av, ub.as<int64_t>(), nr, s_dense, strips > 0 ? 1 : 0, esc_min, pass,
                                               route.as<unsigned char>(), list.as<int32_t>(), list_s.as<int32_t>(),
                                               ebase.as<int32_t>(), list_e.as<int32_t>(), counters.as<int32_t>());
                CSRK_LAUNCH_CHECK();
            }
            CSRK_HIP(hipMemcpy(cnts, counters.p, 16, hipMemcpyDeviceToHost));
            if (strips > 0 && (int64_t)cnts[2] * (strips + 1) * 4 > SGS_TABLE_BUDGET) {
                strips = 0;                      // the sub-range table would not pay for itself: workgroup paths
                continue;
            }
            if (cnts[3] > 0) {                   // do the products of the expand-sort-compress rows fit its budget?
                CSRK_TRY(esc_pu.alloc((size_t)(cnts[3] + 1) * 8));
                CSRK_TRY(esc_off.alloc((size_t)(cnts[3] + 1) * 8));
                sg_esc_gather<<<(unsigned)ceil_div(cnts[3], 256), 256>>>(list_e.as<int32_t>(), cnts[3], ub.as<int64_t>(),
                                                                         esc_pu.as<int64_t>());
                CSRK_LAUNCH_CHECK();
                CSRK_TRY(exclusive_scan_i64(esc_pu.as<int64_t>(), esc_off.as<int64_t>(), cnts[3], nullptr));
                CSRK_HIP(hipMemcpy(&esc_products, esc_off.as<int64_t>() + cnts[3], 8, hipMemcpyDeviceToHost));
                if (esc_products > SGE_BUDGET_PRODUCTS) {
                    esc_min = -1;                // beyond the sort's budget: the round-1 heavy-row paths take these rows
                    continue;
                }
                // ... and into device memory?  The product matrix, its two transposes and the sort's scratch are ~52 B per
                // product at the peak; when that is not there even after the pool has given its cached blocks back, the same paths take over
                // instead of failing the whole product on an allocation half-way through.
                size_t mfree = 0, mtotal = 0;
                CSRK_HIP(hipMemGetInfo(&mfree, &mtotal));
                if ((size_t)esc_products * 52 + (64u << 20) > mfree) {
                    (void)csrk_trim_cache();
                    CSRK_HIP(hipMemGetInfo(&mfree, &mtotal));
                    if ((size_t)esc_products * 52 + (64u << 20) > mfree) {
                        esc_min = -1;
                        continue;
                    }
                }
            }
            break;
        }
        n_esc = cnts[3];
        // (with expand-sort-compress taking every row above 32 products, the 32-lane and workgroup-hash kernels have no rows)
        hash_rows = esc_min < 0 || esc_min > SG_WAVE_CAP;
        mid_rows = esc_min < 0 || esc_min > 32;
        n_large = cnts[0];
        n_strip = cnts[1];
        n_strip_e = cnts[2];
        int cus = 0;
        CSRK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, a->device));
        grid_lds = cus > 0 ? cus : 256;
        lds_symbolic = n_large > 0 && b->ncols <= SGL_MAXBITS;
        if (lds_symbolic) {
            CSRK_HIP(hipFuncSetAttribute((const void *)sg_lds_symbolic_kernel<FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
            CSRK_HIP(hipFuncSetAttribute((const void *)sg_lds_numeric_kernel<FAST>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)((size_t)SGL_W * 8 + SGL_W / 8)));
            CSRK_HIP(hipFuncSetAttribute((const void *)sg_hash_big_kernel<FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        }
        auto alloc_dense = [&]() -> int {       // HBM work / marker rows of the dense path, when some row needs it
            if (work.p) return CSRK_OK;
            const int64_t per = (int64_t)b->ncols * 12 + scratch_len * 4;
            int64_t g_max = (4ll << 30) / (per > 0 ? per : 1);
            if (g_max < 1) g_max = 1;
            grid_dense = (int)(n_large < 256 ? n_large : 256);
            if (grid_dense > g_max) grid_dense = (int)g_max;
            CSRK_TRY(work.alloc((size_t)grid_dense * b->ncols * 8));
            CSRK_TRY(mark.alloc((size_t)grid_dense * b->ncols * 4));
            CSRK_TRY(scratch.alloc((size_t)grid_dense * scratch_len * 4));
            CSRK_HIP(hipMemset(work.p, 0, work.bytes));
            CSRK_HIP(hipMemset(mark.p, 0, mark.bytes));
            return CSRK_OK;
        };
        if (n_large > 0 && !lds_symbolic) CSRK_TRY(alloc_dense());
        if (n_esc > 0) {
            // expand-sort-compress, first half: products out in the reference's order, two stable transposes, runs counted
            Matrix *pm = nullptr, *pt = nullptr;
            CSRK_TRY(new_matrix(n_esc, b->ncols, esc_products, 0, CSRK_VAL_F64, &pm));
            sg_esc_rowptr<<<(unsigned)ceil_div(n_esc + 1, 256), 256>>>(esc_off.as<int64_t>(), n_esc, (int32_t *)pm->d_rowptrs);
            int erc = hipGetLastError() == hipSuccess ? CSRK_OK : CSRK_ERR_HIP;
            sg_esc_expand<FAST><<<(unsigned)n_esc, 256>>>(av, bv, list_e.as<int32_t>(), esc_off.as<int64_t>(), pm->d_colinds,
                                                          (double *)pm->d_values);
            if (erc == CSRK_OK && hipGetLastError() != hipSuccess) erc = CSRK_ERR_HIP;
            if (erc != CSRK_OK) set_error("expand-sort-compress: kernel launch failed");
            if (erc == CSRK_OK) erc = transpose_matrix(pm, 1, &pt, nullptr);
            delete pm;
            if (erc == CSRK_OK) erc = transpose_matrix(pt, 1, &esc_ps, nullptr);
            delete pt;
            CSRK_TRY(erc);
            const int32_t n_pos = (int32_t)esc_products;
            esc_tiles = (int32_t)ceil_div(n_pos, SGE_TILE);
            CSRK_TRY(esc_tile_cnt.alloc((size_t)(esc_tiles + 1) * 4));
            CSRK_TRY(esc_tile_off.alloc((size_t)(esc_tiles + 1) * 4));
            CSRK_TRY(esc_first_run.alloc((size_t)(n_esc + 1) * 4));
#define ESC_RUNS(MODE, C_RP, C_CI, C_VS)                                                                                     \
    sg_esc_runs<MODE><<<(unsigned)esc_tiles, 256>>>(list_e.as<int32_t>(), n_esc, (const int32_t *)esc_ps->d_rowptrs,           \
                                                    esc_ps->d_colinds, (const double *)esc_ps->d_values, n_pos,                 \
                                                    esc_tile_cnt.as<int32_t>(), esc_tile_off.as<int32_t>(),                     \
                                                    esc_first_run.as<int32_t>(), C_RP, C_CI, C_VS)
            ESC_RUNS(0, nullptr, nullptr, nullptr);
            CSRK_LAUNCH_CHECK();
            CSRK_TRY(exclusive_scan_i32(esc_tile_cnt.as<int32_t>(), esc_tile_off.as<int32_t>(), esc_tiles, nullptr));
            ESC_RUNS(1, nullptr, nullptr, nullptr);
            CSRK_LAUNCH_CHECK();
            sg_esc_rowcnt<<<(unsigned)ceil_div(n_esc, 256), 256>>>(list_e.as<int32_t>(), n_esc, esc_first_run.as<int32_t>(),
                                                                   esc_tile_off.as<int32_t>() + esc_tiles, cnt.as<int32_t>());
            CSRK_LAUNCH_CHECK();
        }
        // symbolic (rows with no products keep the zero count of the memset)
        const unsigned gq = (unsigned)ceil_div(nr, 16), gq2 = (unsigned)ceil_div(nr, 8);
        // the rows of at most SG_WAVE_CAP products: one pass into temporaries when they fit (CSRK_SPGEMM_FUSED=0: two passes,
        // the form a product whose temporaries exceed the budget takes)
        {
            const char *sf_env = getenv("CSRK_SPGEMM_FUSED");
            if (!(sf_env && atoi(sf_env) == 0)) {
                CSRK_TRY(small_room.alloc((size_t)(nr + 1) * 4));
                CSRK_TRY(small_off.alloc((size_t)(nr + 1) * 8));
                sg_small_room<<<g, 256>>>(ub.as<int64_t>(), route.as<unsigned char>(), nr, small_room.as<int32_t>());
                CSRK_LAUNCH_CHECK();
                CSRK_TRY(exclusive_scan_i32_to_i64(small_room.as<int32_t>(), small_off.as<int64_t>(), nr, nullptr));
                int64_t total = 0;
                CSRK_HIP(hipMemcpy(&total, small_off.as<int64_t>() + nr, 8, hipMemcpyDeviceToHost));
                small_fused = total > 0 && total * 12 <= SGS_TEMP_BUDGET / 4;
                if (small_fused && (small_tci.alloc((size_t)total * 4) != CSRK_OK || small_tvs.alloc((size_t)total * 8) != CSRK_OK)) {
                    small_tci.release();             // no room for the temporary: two passes
                    small_tvs.release();
                    small_fused = false;
                }
                small_any = total > 0;
            }
        }
        if (small_fused) {
            sg_quad_kernel<16, 32, true, FAST><<<gq, 256>>>(av, bv, ub.as<int64_t>(), 0, route.as<unsigned char>(), cnt.as<int32_t>(),
                                                            nullptr, small_tci.as<int32_t>(), small_tvs.as<double>(),
                                                            small_off.as<int64_t>());
            CSRK_LAUNCH_CHECK();
            if (mid_rows)
                sg_quad_kernel<32, SG_WAVE_CAP, true, FAST><<<gq2, 256>>>(av, bv, ub.as<int64_t>(), 32, route.as<unsigned char>(),
                                                                      cnt.as<int32_t>(), nullptr, small_tci.as<int32_t>(),
                                                                      small_tvs.as<double>(), small_off.as<int64_t>());
            CSRK_LAUNCH_CHECK();
        } else if (small_any) {
            sg_quad_kernel<16, 32, false, FAST><<<gq, 256>>>(av, bv, ub.as<int64_t>(), 0, route.as<unsigned char>(), cnt.as<int32_t>(),
                                                             nullptr, nullptr, nullptr, nullptr);
            CSRK_LAUNCH_CHECK();
            if (mid_rows)
                sg_quad_kernel<32, SG_WAVE_CAP, false, FAST><<<gq2, 256>>>(av, bv, ub.as<int64_t>(), 32, route.as<unsigned char>(),
                                                                       cnt.as<int32_t>(), nullptr, nullptr, nullptr, nullptr);
            CSRK_LAUNCH_CHECK();
        }
        if (hash_rows) {
            sg_hash_kernel<false, FAST><<<(unsigned)nr, SG_THREADS>>>(av, bv, ub.as<int64_t>(), route.as<unsigned char>(), cnt.as<int32_t>(),
                                                                     nullptr, nullptr, nullptr);
            CSRK_LAUNCH_CHECK();
        }
        CSRK_TRY(next.alloc(20));
        CSRK_HIP(hipMemset(next.p, 0, 20));
        if constexpr (FAST) {
            if (n_strip > 0) {
                const int64_t n_units = (int64_t)n_strip * strips;
                CSRK_TRY(emap.alloc((size_t)n_strip_e * 4));
                CSRK_TRY(table.alloc((size_t)n_strip_e * (strips + 1) * 4));
                CSRK_TRY(cnt_s.alloc((size_t)n_units * 4));
                sg_strip_expand<<<(unsigned)n_strip, 256>>>((const int32_t *)a->d_rowptrs, list_s.as<int32_t>(), ebase.as<int32_t>(),
                                                            emap.as<int32_t>());
                CSRK_LAUNCH_CHECK();
                if ((int64_t)b->nrows <= 2 * (int64_t)n_strip_e) {
                    DevBuf sp, start_bits;
                    CSRK_TRY(sp.alloc((size_t)b->nrows * (strips + 1) * 4));
                    CSRK_TRY(start_bits.alloc((size_t)(b->nnz / 32 + 2) * 4));
                    CSRK_HIP(hipMemsetAsync(start_bits.p, 0, (size_t)(b->nnz / 32 + 2) * 4, nullptr));
                    sg_strip_rowstarts<<<(unsigned)ceil_div(b->nrows, 256), 256>>>((const int32_t *)b->d_rowptrs, b->nrows, strips,
                                                                                  start_bits.as<uint32_t>(), sp.as<int32_t>());
                    CSRK_LAUNCH_CHECK();
                    sg_strip_rowptrs<<<(unsigned)ceil_div(b->nnz, 256), 256>>>((const int32_t *)b->d_rowptrs, b->d_colinds, b->nrows,
                                                                              b->nnz, strips, start_bits.as<uint32_t>(),
                                                                              sp.as<int32_t>());
                    CSRK_LAUNCH_CHECK();
                    sg_strip_table_gather<<<(unsigned)ceil_div((int64_t)n_strip_e * (strips + 1), 256), 256>>>(
                        a->d_colinds, emap.as<int32_t>(), sp.as<int32_t>(), n_strip_e, strips, table.as<int32_t>());
                    CSRK_LAUNCH_CHECK();
                    // (`sp` goes back to the pool here: every launch of this call is on the default stream, in order)
                } else {
                    sg_strip_table<<<(unsigned)ceil_div((int64_t)n_strip_e * (strips + 1), 256), 256>>>(
                        a->d_colinds, (const int32_t *)b->d_rowptrs, b->d_colinds, emap.as<int32_t>(), n_strip_e, strips,
                        table.as<int32_t>());
                    CSRK_LAUNCH_CHECK();
                }
                grid_strip = (unsigned)(n_units < (int64_t)grid_lds * SGS_WAVES_PER_CU ? n_units : (int64_t)grid_lds * SGS_WAVES_PER_CU);
                // one pass into a temporary when it fits (CSRK_SPGEMM_FUSED=0: the two-pass form)
                const char *fu_env = getenv("CSRK_SPGEMM_FUSED");
                if (!(fu_env && atoi(fu_env) == 0)) {
                    DevBuf cap;
                    CSRK_TRY(cap.alloc((size_t)(n_units + 1) * 4));
                    CSRK_TRY(strip_off.alloc((size_t)(n_units + 1) * 8));
                    sg_strip_cap<<<(unsigned)ceil_div(n_units * WAVE, 256), 256>>>((const int32_t *)a->d_rowptrs, list_s.as<int32_t>(),
                                                                                ebase.as<int32_t>(), n_strip, strips, n_strip_e,
                                                                                table.as<int32_t>(), cap.as<int32_t>());
                    CSRK_LAUNCH_CHECK();
                    CSRK_TRY(exclusive_scan_i32_to_i64(cap.as<int32_t>(), strip_off.as<int64_t>(), n_units, nullptr));
                    int64_t total_cap = 0;
                    CSRK_HIP(hipMemcpy(&total_cap, strip_off.as<int64_t>() + n_units, 8, hipMemcpyDeviceToHost));
                    strip_fused = total_cap * 12 <= SGS_TEMP_BUDGET;
                    if (strip_fused &&
                        (strip_tci.alloc((size_t)(total_cap + 1) * 4) != CSRK_OK || strip_tvs.alloc((size_t)(total_cap + 1) * 8) != CSRK_OK)) {
                        strip_tci.release();         // no room for the temporary on this device now: two passes
                        strip_tvs.release();
                        strip_fused = false;
                    }
                }
                if (strip_fused) {
                    sg_strip_kernel<2><<<grid_strip, WAVE>>>(
                        (const int32_t *)a->d_rowptrs, (const double *)a->d_values, b->d_colinds, (const double *)b->d_values, b->nnz - 1,
                        list_s.as<int32_t>(), ebase.as<int32_t>(), n_strip, strips, n_strip_e, table.as<int32_t>(), cnt_s.as<int32_t>(),
                        nullptr, cnt.as<int32_t>(), nullptr, strip_tci.as<int32_t>(), strip_tvs.as<double>(),
                        strip_off.as<int64_t>(), next.as<int32_t>() + 3);
                } else {
                    CSRK_TRY(occ_s.alloc((size_t)n_units * SGS_CHUNKS * 8));
                    const unsigned grid_sym = (unsigned)(n_units < (int64_t)grid_lds * 32 ? n_units : (int64_t)grid_lds * 32);
                    sg_strip_kernel<0><<<grid_sym, WAVE>>>(
                        (const int32_t *)a->d_rowptrs, (const double *)a->d_values, b->d_colinds, (const double *)b->d_values, b->nnz - 1,
                        list_s.as<int32_t>(), ebase.as<int32_t>(), n_strip, strips, n_strip_e, table.as<int32_t>(), cnt_s.as<int32_t>(),
                        occ_s.as<unsigned long long>(), cnt.as<int32_t>(), nullptr, nullptr, nullptr, nullptr, next.as<int32_t>() + 3);
                }
                CSRK_LAUNCH_CHECK();
            }
        }
        if (n_large > 0 && lds_symbolic) {
            const size_t lds = (size_t)((b->ncols + 31) / 32) * 4;
            sg_lds_symbolic_kernel<FAST><<<(unsigned)(n_large < grid_lds ? n_large : grid_lds), SGL_THREADS, lds>>>(
                av, bv, list.as<int32_t>(), n_large, cnt.as<int32_t>(), next.as<int32_t>());
            CSRK_LAUNCH_CHECK();
            // numeric: nearly full rows -> LDS tiles, the others -> HBM work rows
            CSRK_TRY(list_a.alloc((size_t)n_large * 4));
            CSRK_TRY(list_h.alloc((size_t)n_large * 4));
            CSRK_TRY(list_b.alloc((size_t)n_large * 4));
            CSRK_TRY(n_ab.alloc(12));
            CSRK_HIP(hipMemset(n_ab.p, 0, 12));
            sg_split_large<<<(unsigned)ceil_div(n_large, 256), 256>>>(list.as<int32_t>(), n_large, cnt.as<int32_t>(), b->ncols,
                                                                     SGB_CAP, list_a.as<int32_t>(), list_h.as<int32_t>(),
                                                                     list_b.as<int32_t>(), n_ab.as<int32_t>());
            CSRK_LAUNCH_CHECK();
            int32_t nab[3] = {0, 0, 0};
            CSRK_HIP(hipMemcpy(nab, n_ab.p, 12, hipMemcpyDeviceToHost));
            n_lds = nab[0];
            n_hash = nab[1];
            n_hbm = nab[2];
            if (n_hbm > 0) CSRK_TRY(alloc_dense());
        } else if (n_large > 0) {
            sg_dense_kernel<false, FAST><<<grid_dense, SG_THREADS>>>(av, bv, list.as<int32_t>(), n_large, work.as<double>(),
                                                              mark.as<int32_t>(), nullptr, 0, cnt.as<int32_t>(), nullptr,
                                                              nullptr, nullptr);
            CSRK_LAUNCH_CHECK();
            n_hbm = n_large;
        }
    }
    // row pointers: int64 scan first so an overflowing product is detected, not wrapped
    DevBuf rp64;
    CSRK_TRY(rp64.alloc((size_t)(nr + 1) * 8));
    CSRK_TRY(exclusive_scan_i32_to_i64(cnt.as<int32_t>(), rp64.as<int64_t>(), nr, nullptr));
    int64_t c_nnz = 0;
    CSRK_HIP(hipMemcpy(&c_nnz, rp64.as<int64_t>() + nr, 8, hipMemcpyDeviceToHost));
    if (c_nnz > INT32_MAX) {
        set_error("product has %lld entries; the reference's int32 row pointers (multiply.py:28) cannot hold it: "
                  "multiply row blocks of A instead", (long long)c_nnz);
        return CSRK_ERR_OVERFLOW;
    }
    Matrix *c = nullptr;
    CSRK_TRY(new_matrix(nr, b->ncols, c_nnz, 0, CSRK_VAL_F64, &c));
    int rc = exclusive_scan_i32(cnt.as<int32_t>(), (int32_t *)c->d_rowptrs, nr, nullptr);
    if (rc == CSRK_OK && nr > 0 && c_nnz > 0) {
        if (small_fused) {
            sg_small_copy<<<(unsigned)ceil_div((int64_t)nr * 16, 256), 256>>>(small_room.as<int32_t>(), nr, cnt.as<int32_t>(),
                                                                             small_off.as<int64_t>(), small_tci.as<int32_t>(),
                                                                             small_tvs.as<double>(), (const int32_t *)c->d_rowptrs,
                                                                             c->d_colinds, (double *)c->d_values);
        } else if (small_any) {
            sg_quad_kernel<16, 32, true, FAST><<<(unsigned)ceil_div(nr, 16), 256>>>(av, bv, ub.as<int64_t>(), 0, route.as<unsigned char>(),
                                                                                    nullptr, (const int32_t *)c->d_rowptrs, c->d_colinds,
                                                                                    (double *)c->d_values, nullptr);
            if (mid_rows)
                sg_quad_kernel<32, SG_WAVE_CAP, true, FAST><<<(unsigned)ceil_div(nr, 8), 256>>>(
                av, bv, ub.as<int64_t>(), 32, route.as<unsigned char>(), nullptr, (const int32_t *)c->d_rowptrs, c->d_colinds,
                (double *)c->d_values, nullptr);
        }
        if (hash_rows)
            sg_hash_kernel<true, FAST><<<(unsigned)nr, SG_THREADS>>>(av, bv, ub.as<int64_t>(), route.as<unsigned char>(), nullptr,
                                                             (const int32_t *)c->d_rowptrs,
                                                             c->d_colinds, (double *)c->d_values);
        if constexpr (FAST) {
            if (n_strip > 0 && strip_fused)
                sg_strip_copy<<<(unsigned)ceil_div((int64_t)n_strip * strips * WAVE, 256), 256>>>(
                    list_s.as<int32_t>(), n_strip, strips, cnt_s.as<int32_t>(), strip_off.as<int64_t>(), strip_tci.as<int32_t>(),
                    strip_tvs.as<double>(), (const int32_t *)c->d_rowptrs, c->d_colinds, (double *)c->d_values);
            else if (n_strip > 0)
                sg_strip_kernel<1><<<grid_strip, WAVE>>>(
                    (const int32_t *)a->d_rowptrs, (const double *)a->d_values, b->d_colinds, (const double *)b->d_values, b->nnz - 1,
                    list_s.as<int32_t>(), ebase.as<int32_t>(), n_strip, strips, n_strip_e, table.as<int32_t>(), cnt_s.as<int32_t>(),
                    occ_s.as<unsigned long long>(), nullptr, (const int32_t *)c->d_rowptrs, c->d_colinds, (double *)c->d_values,
                    nullptr, next.as<int32_t>() + 4);
        }
        if (n_lds > 0)
            sg_lds_numeric_kernel<FAST><<<(unsigned)(n_lds < grid_lds ? n_lds : grid_lds), SGL_THREADS,
                                    (size_t)SGL_W * 8 + SGL_W / 8>>>(av, bv, list_a.as<int32_t>(), n_lds,
                                                                     (const int32_t *)c->d_rowptrs, c->d_colinds,
                                                                     (double *)c->d_values, next.as<int32_t>() + 1);
        if (n_hash > 0)
            sg_hash_big_kernel<FAST><<<(unsigned)(n_hash < grid_lds ? n_hash : grid_lds), SGL_THREADS,
                                 (size_t)SGB_SLOTS * 12 + (size_t)SGB_CAP * 12>>>(av, bv, list_h.as<int32_t>(), n_hash,
                                                                                  (const int32_t *)c->d_rowptrs, c->d_colinds,
                                                                                  (double *)c->d_values, next.as<int32_t>() + 2);
        if (n_esc > 0) {
            // expand-sort-compress, second half: every run of equal columns added front to back into its place in C
            const int32_t n_pos = (int32_t)esc_products;
            sg_esc_rowbase<<<(unsigned)ceil_div(n_esc, 256), 256>>>(list_e.as<int32_t>(), n_esc, (const int32_t *)c->d_rowptrs,
                                                                    esc_first_run.as<int32_t>());
            ESC_RUNS(2, (const int32_t *)c->d_rowptrs, c->d_colinds, (double *)c->d_values);
        }
#undef ESC_RUNS
        if (n_hbm > 0)
            sg_dense_kernel<true, FAST><<<grid_dense, SG_THREADS>>>(av, bv, lds_symbolic ? list_b.as<int32_t>() : list.as<int32_t>(),
                                                             n_hbm, work.as<double>(), mark.as<int32_t>(),
                                                             scratch.as<int32_t>(), scratch_len, nullptr,
                                                             (const int32_t *)c->d_rowptrs, c->d_colinds, (double *)c->d_values);
    }
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipGetLastError();
    if (rc == CSRK_OK && e != hipSuccess) {
        set_error("spgemm kernels failed: %s", hipGetErrorString(e));
        rc = CSRK_ERR_HIP;
    }
    if (rc != CSRK_OK) {
        delete c;
        return rc;
    }
    *out = c;
    if (products) {                                  // the rows' product counts: the ordering pass needs them too
        products->release();
        products->bytes = ub.bytes;
        products->p = ub.take();
    }
    return CSRK_OK;
}

// b_rows_ascend: the caller knows that B's rows hold strictly ascending columns (the transpose of a matrix without
// repeated entries); otherwise a kernel checks before the strips rely on it
int spgemm_dense_b(Matrix *a, Matrix *b, Matrix **out, bool *taken);      // spmm_dense.hip
static thread_local int t_last_route = 0;      // csrk_spgemm_last_route

static int spgemm_impl(Matrix *a, Matrix *b, Matrix **out, bool b_rows_ascend)
{
    {   // a fully populated B in row-major panel form: the dense-panel kernels, C as the reference returns it
        bool taken = false;
        t_last_route = 0;
        CSRK_TRY(spgemm_dense_b(a, b, out, &taken));
        t_last_route = taken ? 1 : 0;
        if (taken) return CSRK_OK;
    }
    const bool fast = !a->ptr64 && !b->ptr64 && a->val_type == CSRK_VAL_F64 && b->val_type == CSRK_VAL_F64;
    const bool ordered = spgemm_reference_order_wanted();
    DevBuf products;                            // products of every row of A B (int64), when the product kernels counted them
    CSRK_TRY(fast ? spgemm_run<true>(a, b, out, b_rows_ascend, ordered ? &products : nullptr)
                  : spgemm_run<false>(a, b, out, b_rows_ascend, ordered ? &products : nullptr));
    if (ordered) {                              // columns in the reference's reverse-discovery order (spgemm_order.hip)
        const int rc = spgemm_apply_reference_order(a, b, *out, &products);
        if (rc != CSRK_OK) {
            delete *out;
            *out = nullptr;
            return rc;
        }
    }
    return CSRK_OK;
}

}  // namespace csrk

using namespace csrk;

extern "C" {

#ifdef CSRK_SG_STAMPS
CSRK_API int csrk_debug_sg_stamps(unsigned long long *out, int n)
{
    CSRK_HIP(hipDeviceSynchronize());
    CSRK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sg_stamps), (size_t)n * 8));
    return CSRK_OK;
}
#endif

int csrk_spgemm_last_route(int *route)
{
    CSRK_REQUIRE(route, "route is NULL");
    *route = t_last_route;
    return CSRK_OK;
}

int csrk_spgemm_ab(csrk_handle_t ah, csrk_handle_t bh, csrk_handle_t *out)
{
    CSRK_REQUIRE(out, "out is NULL");
    *out = 0;
    Matrix *a = from_handle(ah), *b = from_handle(bh);
    if (!a || !b) return CSRK_ERR_INVALID;
    Matrix *c = nullptr;
    CSRK_TRY(spgemm_impl(a, b, &c, false));
    *out = to_handle(c);
    return CSRK_OK;
}

// A B^T = mult_ab(A, transpose(B)) exactly as the reference does it (multiply.py:54-57)
int csrk_spgemm_abt(csrk_handle_t ah, csrk_handle_t bh, csrk_handle_t *out)
{
    CSRK_REQUIRE(out, "out is NULL");
    *out = 0;
    Matrix *a = from_handle(ah), *b = from_handle(bh);
    if (!a || !b) return CSRK_ERR_INVALID;
    CSRK_REQUIRE(a->ncols == b->ncols, "mult_abt: A is %d x %d but B is %d x %d", a->nrows, a->ncols, b->nrows, b->ncols);
    CSRK_REQUIRE(b->val_type != CSRK_VAL_NONE, "mult_abt needs values on both operands");
    Matrix *bt = nullptr;
    CSRK_TRY(transpose_matrix(b, 1, &bt, nullptr));
    Matrix *c = nullptr;
    // (the transpose's rows ascend -- strictly, unless a row of B holds a column twice; the strips' sub-range bounds need
    // only the order)
    int rc = spgemm_impl(a, bt, &c, true);
    delete bt;
    if (rc != CSRK_OK) return rc;
    *out = to_handle(c);
    return CSRK_OK;
}

}  // extern "C"
