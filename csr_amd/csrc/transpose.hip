// Transpose for libcsrk on gfx950, bit-exact with the reference's stable counting sort
// (csr/structure.py:172-204 _transpose_values, :207-237 _transpose_structure, :240-247).
//
// The reference scatters entries in row-major order through one cursor per column, so
// inside every output row the entries appear in ascending SOURCE POSITION.  A GPU
// atomic-cursor scatter would lose that order; instead the entries are sorted by column
// with a STABLE least-significant-digit radix sort (8-bit digits, ceil(bits(ncols)/8)
// passes), which yields exactly the reference's arrangement, including for duplicate
// (row, col) entries and unsorted input rows.
//
// One pass = per-chunk digit histogram -> exclusive scan of the (digit, chunk) table ->
// stable scatter.  A chunk is 2048 consecutive records handled by one 256-thread
// workgroup in 8 in-order rounds; within a round the rank among equal digits comes from
// a ballot-based match (8 ballots), across wavefronts/rounds from LDS counters, so no
// ordering depends on atomics.  The first pass reads the CSR arrays directly (source row
// ids are recovered from rowptrs inside the kernel, values are widened to float64 as
// structure.py:177 requires); the last pass writes straight into the output matrix.
//
// HBM bytes per nnz, two passes with values: 4 (hist) + 12 (read) + 16 (write)
//   + 4 (hist) + 16 (read) + 12 (write) = 64; structure only: 32.
// Output rowptrs are read off the sorted keys (= the reference's histogram + running sum,
// structure.py:180-188), which avoids a contended global-atomic histogram.
//
// Two-pass sorts (257 .. 65536 keys -- the MovieLens-shaped transpose of BASELINE configs[4]) with payloads below 2^24
// take a leaner route, 56 instead of 72 bytes per record: pass 1 writes 12-byte records -- the float64 value and ONE
// word holding the key's high digit (bits 31..24) and the payload (bits 23..0) -- pass 2 reads that word for its
// histogram and its ranking and writes payload + value only, and its chunks are aligned to the runs pass 1 produced
// (a chunk never spans two low digits), so the exclusive scan of the pass-2 digit table IS the output row pointers:
// rowptr[hi * 256 + lo] = (records of smaller high digits) + (records of high digit hi in the chunks before low digit
// lo's run).  No sorted-key array is written or read.
#include "common.h"

namespace csrk {

#ifndef CSRK_RX_THREADS
#define CSRK_RX_THREADS 512
#endif
constexpr int RX_THREADS = CSRK_RX_THREADS;
constexpr int RX_ROUNDS = 8;
constexpr int RX_CHUNK = RX_THREADS * RX_ROUNDS;     // 4096 records per workgroup (8192: 0.67 vs 0.65 ms)
constexpr int RX_WAVES = RX_THREADS / WAVE;
constexpr int RX_WSPAN = WAVE * RX_ROUNDS;            // 512 consecutive records per wavefront

// ---- output row pointers from the sorted keys ----------------------------------------------
// A histogram with global atomics serialises on popular columns (1.8 ms for the MovieLens-shaped
// matrix, more than the sort itself); the sorted key sequence gives the same pointers for free:
// brp[c] = first position whose key is >= c (the reference's histogram + running sum,
// csr/structure.py:180-188).
template <class P>
__global__ __launch_bounds__(256) void rowptr_from_sorted_keys(const int32_t *__restrict__ keys, int64_t n, int32_t ncols,
                                                              P *__restrict__ brp)
{
    // four consecutive positions per thread: one 16-B load + the key before them
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 > n) return;
    int32_t k[5];
    k[0] = i0 == 0 ? -1 : keys[i0 - 1];
    if (i0 + 4 <= n) {
        const int4 v = *(const int4 *)(keys + i0);       // keys comes from the pool allocator: 256-B aligned
        k[1] = v.x;
        k[2] = v.y;
        k[3] = v.z;
        k[4] = v.w;
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) k[q + 1] = i0 + q < n ? keys[i0 + q] : ncols;      // position n: closes the last columns
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int64_t i = i0 + q;
        if (i > n) break;
        const int64_t lo = (int64_t)k[q] + 1;
        const int64_t hi = i == n ? (int64_t)ncols : (int64_t)k[q + 1];
        for (int64_t c = lo; c <= hi; c++) brp[c] = (P)i;
    }
}

// Workgroup -> chunk: consecutive chunks go to workgroups b, b + 8, b + 16, ... -- which share an XCD and
// hence an L2 (blocks are dealt round-robin over the 8 XCDs; a speed assumption only).  Neighbouring chunks
// write neighbouring pieces of every digit's output run, each piece a fraction of a 128-B line: in one L2 the
// pieces merge into whole lines before they are written back.
__device__ __forceinline__ int64_t rx_chunk_of(int64_t b, int64_t n_chunks)
{
    const int64_t per = (n_chunks + 7) / 8;             // chunks per XCD group
    const int64_t c = (b % 8) * per + b / 8;
    return c;                                           // may be >= n_chunks (grid is padded to 8 * per)
}

// ---- radix pass: histogram ----------------------------------------------------------------
// (cstart / ccnt: optional chunk descriptors -- first record and record count of every chunk -- for a pass whose chunks
// are aligned to the previous pass's runs; without them chunk c is records [c * RX_CHUNK, (c + 1) * RX_CHUNK))
// One workgroup counts RX_HC consecutive chunks: all their keys are requested before the first LDS atomic (a workgroup
// per chunk was a load -> wait -> 4096 LDS atomics -> store chain per 16 KB of keys: 2.5 TB/s).
constexpr int RX_HC = 4;
__global__ __launch_bounds__(RX_THREADS) void rx_hist_kernel(const int32_t *__restrict__ keys, int64_t n, int shift,
                                                            int64_t n_chunks, int64_t *__restrict__ table,
                                                            const int64_t *__restrict__ cstart, const int32_t *__restrict__ ccnt)
{
    __shared__ int32_t h[RX_HC][256];
    for (int k = threadIdx.x; k < RX_HC * 256; k += RX_THREADS) (&h[0][0])[k] = 0;
    int32_t key[RX_HC][RX_ROUNDS];
    int cnt[RX_HC];
#pragma unroll
    for (int q = 0; q < RX_HC; q++) {
        const int64_t chunk = (int64_t)blockIdx.x * RX_HC + q;
        const bool in = chunk < n_chunks;
        const int64_t base = !in ? 0 : (cstart ? cstart[chunk] : chunk * RX_CHUNK);
        cnt[q] = !in ? 0 : (cstart ? ccnt[chunk] : (int)(n - base < RX_CHUNK ? n - base : RX_CHUNK));
#pragma unroll
        for (int r = 0; r < RX_ROUNDS; r++) {        // unconditional (clamped) loads: all in flight together
            const int k = r * RX_THREADS + threadIdx.x;
            key[q][r] = keys[base + (k < cnt[q] ? k : (cnt[q] > 0 ? cnt[q] - 1 : 0))];
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RX_HC; q++)
#pragma unroll
        for (int r = 0; r < RX_ROUNDS; r++)
            if (r * RX_THREADS + (int)threadIdx.x < cnt[q]) atomicAdd(&h[q][(key[q][r] >> shift) & 255], 1);
    __syncthreads();
    for (int k = threadIdx.x; k < RX_HC * 256; k += RX_THREADS) {
        const int q = k >> 8, d = k & 255;
        const int64_t chunk = (int64_t)blockIdx.x * RX_HC + q;
        if (chunk < n_chunks) table[(int64_t)d * n_chunks + chunk] = h[q][d];      // (an unused chunk of an aligned pass: zeros)
    }
}

// ---- radix pass: table scan ------------------------------------------------------------------
// table[d][c] (digit-major) -> exclusive prefix inside digit d's row (in place) and total[d]; the scatter
// kernel adds the exclusive scan of the 256 totals itself.  One workgroup per digit, one launch -- instead
// of a generic three-launch device scan of the 256 * n_chunks counts.
constexpr int RXS_THREADS = 1024;
constexpr int RXS_IPT = 8;
__global__ __launch_bounds__(RXS_THREADS) void rx_scan_kernel(int64_t *__restrict__ table, int64_t n_chunks,
                                                             int64_t *__restrict__ total)
{
    __shared__ int64_t s_w[RXS_THREADS / WAVE];
    __shared__ int64_t s_carry;
    int64_t *row = table + (int64_t)blockIdx.x * n_chunks;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < n_chunks; c0 += RXS_THREADS * RXS_IPT) {
        const int64_t b = c0 + (int64_t)tid * RXS_IPT;
        int64_t v[RXS_IPT], tsum = 0;
#pragma unroll
        for (int k = 0; k < RXS_IPT; k++) {
            v[k] = b + k < n_chunks ? row[b + k] : 0;
            tsum += v[k];
        }
        int64_t inc = tsum;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const int64_t o = __shfl_up(inc, off, WAVE);
            if (lane >= off) inc += o;
        }
        if (lane == WAVE - 1) s_w[w] = inc;
        __syncthreads();
        int64_t run = s_carry + inc - tsum;
        for (int k = 0; k < w; k++) run += s_w[k];
#pragma unroll
        for (int k = 0; k < RXS_IPT; k++) {
            if (b + k < n_chunks) row[b + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == RXS_THREADS - 1) s_carry = run;
        __syncthreads();
    }
    if (tid == 0) total[blockIdx.x] = s_carry;
}

// first / last source row whose extent meets each chunk (first radix pass of a transpose)
template <class P>
__device__ __forceinline__ int32_t row_of(const P *__restrict__ rp, int64_t i, int32_t lo, int32_t hi);

template <class P>
__global__ void rx_rowbounds_kernel(const P *__restrict__ rp, int32_t nrows, int64_t n, int64_t n_chunks, int chunk,
                                    int32_t *__restrict__ rlo, int32_t *__restrict__ rhi)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const int64_t first = c * chunk;
    const int64_t last = first + chunk - 1 < n - 1 ? first + chunk - 1 : n - 1;
    const int32_t a = row_of(rp, first, 0, nrows - 1);
    rlo[c] = a;
    rhi[c] = row_of(rp, last, a, nrows - 1);
}

// source row of entry i, searched inside [lo, hi] (rows whose extents meet this chunk)
template <class P>
__device__ __forceinline__ int32_t row_of(const P *__restrict__ rp, int64_t i, int32_t lo, int32_t hi)
{
    // largest r with rp[r] <= i
    while (lo < hi) {
        int32_t mid = lo + (hi - lo + 1) / 2;
        if ((int64_t)rp[mid] <= i)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}

// ---- radix pass: stable scatter --------------------------------------------------------------
// MODE 1: first pass of a transpose -- records come from the CSR arrays (keys = colinds, row ids
//         recovered from rowptrs, values from the matrix in dtype VT).
// MODE 2: first pass of a COO ingest -- keys, payload and values (dtype VT) are given arrays.
// MODE 0: later pass -- keys, payload, float64 values from the previous pass.
// VT = CSRK_VAL_NONE: structure only.  keys_out may be NULL.
//
// A record's slot inside its chunk's output is  loff[digit] + (records of that digit placed by
// earlier rounds / earlier wavefronts / lower lanes)  -- stable by construction.  The chunk is first
// shuffled into that order in LDS and then written out with consecutive lanes on consecutive global
// addresses: a direct scatter issues one 4-8-byte write request per record and array and runs at the
// chip's request rate (~24 ps per record measured), whereas runs of equal digits (16 records on
// average) coalesce into full-line writes.
// PK (packed two-pass route, see the file header): 1 = its first pass -- rows_out receives the word {high digit of the
// key, payload} instead of the payload, no keys are written --, 2 = its second pass -- keys_in is that word (digit =
// bits 31..24 with shift 24, payload = bits 23..0), chunks come from the descriptors cstart / ccnt.
template <class P, int VT, int MODE, int PK = 0>
__global__ __launch_bounds__(RX_THREADS) void rx_scatter_kernel(
    const int32_t *__restrict__ keys_in, const int32_t *__restrict__ rows_in, const void *__restrict__ vals_in,
    const P *__restrict__ rp, int32_t nrows, int64_t n, int shift, int64_t n_chunks,
    const int64_t *__restrict__ table, const int64_t *__restrict__ total, const int32_t *__restrict__ chunk_rlo,
    const int32_t *__restrict__ chunk_rhi, int32_t *__restrict__ keys_out, int32_t *__restrict__ rows_out,
    double *__restrict__ vals_out, const int64_t *__restrict__ cstart = nullptr, const int32_t *__restrict__ ccnt = nullptr)
{
    constexpr bool HAS_V = VT != CSRK_VAL_NONE;
    constexpr bool FIRST = MODE == 1;
    __shared__ int64_t s_goff[256];               // global offset of this chunk's run per digit
    __shared__ int32_t s_loff[256];               // offset of the digit's run inside the chunk
    __shared__ int32_t s_wh[RX_WAVES][256];       // per-wavefront digit counts, then exclusive prefix over wavefronts
    __shared__ int32_t s_key[RX_CHUNK];           // the chunk in output order
    __shared__ int32_t s_row[RX_CHUNK];           // (FIRST: holds the source row of every entry first)
    __shared__ double s_val[HAS_V ? RX_CHUNK : 1];
    __shared__ int32_t s_tmax[RX_WAVES];
    __shared__ int64_t s_dtot[4];

    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int64_t chunk = rx_chunk_of(blockIdx.x, n_chunks);
    if (chunk >= n_chunks) return;
    const int64_t base = PK == 2 ? cstart[chunk] : chunk * RX_CHUNK;
    const int cnt = PK == 2 ? ccnt[chunk] : (int)(n - base < RX_CHUNK ? n - base : RX_CHUNK);
    if (cnt <= 0) return;                         // (an unused chunk of an aligned pass)
    if (tid < 256) {
        // global offset of this chunk's run of digit `tid` = exclusive scan of the digit totals + the
        // in-row prefix left by rx_scan_kernel
        const int64_t t = total[tid];
        int64_t inc = t;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const int64_t o = __shfl_up(inc, off, WAVE);
            if (lane >= off) inc += o;
        }
        if (lane == WAVE - 1) s_dtot[w] = inc;
        s_goff[tid] = inc - t + table[(int64_t)tid * n_chunks + chunk];
    }
#pragma unroll
    for (int k = tid; k < RX_WAVES * 256; k += RX_THREADS) (&s_wh[0][0])[k] = 0;

    // Record k of the chunk is owned by wavefront k / 512, round (k % 512) / 64, lane k % 64: a wavefront
    // holds 512 CONSECUTIVE records (coalesced 64-record loads per round), so stable order inside the chunk is
    // (wavefront, round, lane).
    int32_t key[RX_ROUNDS], row[RX_ROUNDS];
    double val[RX_ROUNDS];
    if (FIRST) {
        // Source rows for the whole chunk at once: mark each row's first entry with its id (the
        // largest id wins where empty rows share a position) and take a running maximum -- instead of
        // one binary search over rowptrs per entry.  s_row is the scratch array.
        const int32_t rlo = chunk_rlo[chunk], rhi = chunk_rhi[chunk];      // rx_rowbounds_kernel
        for (int k = tid; k < RX_CHUNK; k += RX_THREADS) s_row[k] = k == 0 ? rlo : 0;
        __syncthreads();
        for (int32_t r = rlo + 1 + tid; r <= rhi; r += RX_THREADS) {
            const int64_t pos = (int64_t)rp[r] - base;
            if (pos >= 0 && pos < RX_CHUNK) atomicMax(&s_row[pos], r);
        }
        __syncthreads();
        int32_t mx = 0;
#pragma unroll
        for (int k = 0; k < RX_ROUNDS; k++) {
            const int32_t v = s_row[tid * RX_ROUNDS + k];
            mx = v > mx ? v : mx;
        }
        int32_t inc = mx;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const int32_t o = __shfl_up(inc, off, WAVE);
            if (lane >= off) inc = o > inc ? o : inc;
        }
        if (lane == WAVE - 1) s_tmax[w] = inc;
        __syncthreads();
        int32_t pre = __shfl_up(inc, 1, WAVE);
        if (lane == 0) pre = 0;
        for (int k = 0; k < w; k++) pre = s_tmax[k] > pre ? s_tmax[k] : pre;
#pragma unroll
        for (int k = 0; k < RX_ROUNDS; k++) {
            const int32_t v = s_row[tid * RX_ROUNDS + k];
            pre = v > pre ? v : pre;
            s_row[tid * RX_ROUNDS + k] = pre;
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < RX_ROUNDS; r++) {
        const int k = w * RX_WSPAN + r * WAVE + lane;
        const int kc = k < cnt ? k : cnt - 1;                       // clamped: loads stay unconditional
        const int64_t i = base + kc;
        key[r] = keys_in[i];
        if (FIRST) {
            row[r] = s_row[kc];
            if (VT == CSRK_VAL_F64) val[r] = ((const double *)vals_in)[i];
            else if (VT == CSRK_VAL_F32) val[r] = (double)((const float *)vals_in)[i];
            else val[r] = 0.0;
        } else {
            row[r] = PK == 2 ? (key[r] & 0xffffff) : rows_in[i];
            if (MODE == 2 && VT == CSRK_VAL_F32) val[r] = (double)((const float *)vals_in)[i];
            else val[r] = HAS_V ? ((const double *)vals_in)[i] : 0.0;
        }
    }
    __syncthreads();                  // FIRST: everyone has read its rows out of s_row; s_wh is zeroed

    // Stable rank of every record among the records of its digit in this wavefront: the wavefront's own
    // digit counters (LDS, touched by this wavefront only, whose LDS operations complete in order) give the
    // records of earlier rounds, a ballot match the lower lanes of this round.  No workgroup barrier inside.
    int32_t rank[RX_ROUNDS];
#pragma unroll
    for (int r = 0; r < RX_ROUNDS; r++) {
        const bool valid = w * RX_WSPAN + r * WAVE + lane < cnt;
        const int d = (key[r] >> shift) & 255;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long bm = __ballot((d >> b) & 1);
            peers &= ((d >> b) & 1) ? bm : ~bm;
        }
        const int below = __popcll(peers & ((1ull << lane) - 1ull));
        const int old = s_wh[w][d];
        rank[r] = old + below;
        if (valid && below == 0) s_wh[w][d] = old + __popcll(peers);
    }
    __syncthreads();
    if (tid < 256) {
        // digit `tid`: counts per wavefront -> exclusive prefix over wavefronts; chunk total -> scan over digits
        int32_t run = 0;
#pragma unroll
        for (int k = 0; k < RX_WAVES; k++) {
            const int32_t c = s_wh[k][tid];
            s_wh[k][tid] = run;
            run += c;
        }
        int32_t inc = run;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const int32_t o = __shfl_up(inc, off, WAVE);
            if (lane >= off) inc += o;
        }
        if (lane == WAVE - 1) s_tmax[w] = inc;
        s_loff[tid] = inc - run;          // exclusive inside the wavefront; the other wavefronts' totals are added below
    }
    __syncthreads();
    if (tid < 256) {
        int32_t pre = s_loff[tid];
        int64_t gpre = 0;
        for (int k = 0; k < w; k++) {
            pre += s_tmax[k];
            gpre += s_dtot[k];
        }
        s_loff[tid] = pre;
        s_goff[tid] += gpre;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RX_ROUNDS; r++) {
        if (w * RX_WSPAN + r * WAVE + lane < cnt) {
            const int d = (key[r] >> shift) & 255;
            const int o = s_loff[d] + s_wh[w][d] + rank[r];
            s_key[o] = key[r];
            s_row[o] = PK == 1 ? (int32_t)(((uint32_t)(key[r] >> 8) << 24) | (uint32_t)row[r]) : row[r];
            if (HAS_V) s_val[o] = val[r];
        }
    }
    __syncthreads();

    // write out: consecutive lanes -> consecutive positions of a digit's run
    for (int j = tid; j < cnt; j += RX_THREADS) {
        const int32_t k = s_key[j];
        const int d = (k >> shift) & 255;
        const int64_t o = s_goff[d] + (j - s_loff[d]);
        if (keys_out) keys_out[o] = k;
        rows_out[o] = s_row[j];
        if (HAS_V) vals_out[o] = s_val[j];
    }
}

// ---- packed two-pass route: chunks of pass 2 aligned to the runs of pass 1 ---------------------------------
// run d (the records whose low digit is d, in pass-1 order) = [R[d], R[d + 1]) with R = exclusive scan of total1;
// it is cut into ceil(len / RX_CHUNK) chunks.  One workgroup of 256 threads: thread d scans, then fills its chunks.
__global__ __launch_bounds__(256) void rx_align_kernel(const int64_t *__restrict__ total1, int64_t n_chunks_cap,
                                                      int64_t *__restrict__ cstart, int32_t *__restrict__ ccnt,
                                                      int64_t *__restrict__ run_chunk0)
{
    __shared__ int64_t s_ws[4], s_wc[4];
    const int d = threadIdx.x, lane = d & (WAVE - 1), w = d / WAVE;
    const int64_t len = total1[d], nch = (len + RX_CHUNK - 1) / RX_CHUNK;
    int64_t is = len, ic = nch;                   // inclusive scans over the 256 digits: wavefront, then 4 wavefront totals
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const int64_t a = __shfl_up(is, off, WAVE), b = __shfl_up(ic, off, WAVE);
        if (lane >= off) is += a, ic += b;
    }
    if (lane == WAVE - 1) s_ws[w] = is, s_wc[w] = ic;
    __syncthreads();
    int64_t st = is - len, c0 = ic - nch, used = 0;
    for (int k = 0; k < 4; k++) {
        if (k < w) st += s_ws[k], c0 += s_wc[k];
        used += s_wc[k];
    }
    const int64_t c1 = c0 + nch;
    run_chunk0[d] = c0;
    if (d == 255) run_chunk0[256] = c1;
    for (int64_t c = c0; c < c1; c++) {
        cstart[c] = st + (c - c0) * RX_CHUNK;
        const int64_t left = len - (c - c0) * RX_CHUNK;
        ccnt[c] = (int32_t)(left < RX_CHUNK ? left : RX_CHUNK);
    }
    for (int64_t c = used + d; c < n_chunks_cap; c += 256) {      // the unused tail of the descriptor arrays
        cstart[c] = 0;
        ccnt[c] = 0;
    }
}

// rowptr[hi * 256 + lo] = records of smaller high digits + records of high digit hi in the chunks before run lo
// (table2 holds the exclusive prefix over chunks of every digit's row, total2 the rows' totals)
template <class P>
__global__ __launch_bounds__(256) void rx_rowptr_from_table_kernel(const int64_t *__restrict__ table2, const int64_t *__restrict__ total2,
                                                                  const int64_t *__restrict__ run_chunk0, int64_t n_chunks,
                                                                  int32_t key_range, int64_t n, P *__restrict__ out_ptr)
{
    __shared__ int64_t s_before[257];
    if (threadIdx.x == 0) {
        s_before[0] = 0;
        for (int q = 0; q < 256; q++) s_before[q + 1] = s_before[q] + total2[q];
    }
    __syncthreads();
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > key_range) return;
    if (c == key_range) {
        out_ptr[c] = (P)n;
        return;
    }
    const int hi = (int)(c >> 8), lo = (int)(c & 255);
    const int64_t ch = run_chunk0[lo];
    const int64_t in_row = ch < n_chunks ? table2[(int64_t)hi * n_chunks + ch] : total2[hi];
    out_ptr[c] = (P)(s_before[hi] + in_row);
}

// Stable sort of n records by a key in [0, key_range): payload (int32) -> out_payload, values (widened
// to float64) -> out_vals, and the run starts of every key -> out_ptr[0..key_range].  FROM_CSR: the records
// are the entries of a CSR matrix (keys = its colinds, payload = the entry's row, from `rp`); otherwise
// keys / payload / values are the given device arrays (COO ingest).
template <class P, int VT, bool FROM_CSR>
static int sort_records(const int32_t *keys, const int32_t *payload, const void *vals, const P *rp, int32_t rp_rows,
                        int64_t n, int32_t key_range, int64_t payload_range, P *out_ptr, int32_t *out_payload,
                        double *out_vals, hipStream_t s)
{
    constexpr bool HAS_V = VT != CSRK_VAL_NONE;
    if (n == 0) {
        if (out_ptr) CSRK_HIP(hipMemsetAsync(out_ptr, 0, (size_t)(key_range + 1) * sizeof(P), s));
        CSRK_HIP(hipStreamSynchronize(s));
        return CSRK_OK;
    }
    int bits = 0;
    while (bits < 31 && (1ll << bits) < (int64_t)key_range) bits++;
    int passes = bits <= 8 ? 1 : (bits + 7) / 8;
    const int64_t n_chunks = ceil_div(n, RX_CHUNK);

    DevBuf table, total, rlo, rhi, keyA, keyB, rowA, rowB, valA, valB, keyL;
    const bool packed = passes == 2 && payload_range <= (1ll << 24);
    const int64_t n_chunks2 = packed ? n_chunks + 256 : n_chunks;      // aligned chunks: at most one partial chunk per run more
    CSRK_TRY(table.alloc((size_t)(256 * n_chunks2 + 1) * 8));
    CSRK_TRY(total.alloc(256 * 8));
    if (FROM_CSR) {
        CSRK_TRY(rlo.alloc((size_t)n_chunks * 4));
        CSRK_TRY(rhi.alloc((size_t)n_chunks * 4));
        rx_rowbounds_kernel<P><<<(unsigned)ceil_div(n_chunks, 256), 256, 0, s>>>(rp, rp_rows, n, n_chunks, RX_CHUNK,
                                                                              rlo.as<int32_t>(), rhi.as<int32_t>());
        CSRK_LAUNCH_CHECK();
    }
    if (packed) {
        // pass 1 (low digit): {high digit, payload} words + values; pass 2 (high digit) on chunks aligned to pass 1's runs
        DevBuf cstart, ccnt, run0, total2;
        CSRK_TRY(rowA.alloc((size_t)n * 4));
        if (HAS_V) CSRK_TRY(valA.alloc((size_t)n * 8));
        CSRK_TRY(cstart.alloc((size_t)n_chunks2 * 8));
        CSRK_TRY(ccnt.alloc((size_t)n_chunks2 * 4));
        CSRK_TRY(run0.alloc(257 * 8));
        CSRK_TRY(total2.alloc(256 * 8));
        double *v_mid = HAS_V ? valA.as<double>() : nullptr;
        rx_hist_kernel<<<(unsigned)ceil_div(n_chunks, RX_HC), RX_THREADS, 0, s>>>(keys, n, 0, n_chunks, table.as<int64_t>(), nullptr, nullptr);
        CSRK_LAUNCH_CHECK();
        rx_scan_kernel<<<256, RXS_THREADS, 0, s>>>(table.as<int64_t>(), n_chunks, total.as<int64_t>());
        CSRK_LAUNCH_CHECK();
        rx_scatter_kernel<P, VT, FROM_CSR ? 1 : 2, 1><<<(unsigned)(8 * ceil_div(n_chunks, 8)), RX_THREADS, 0, s>>>(
            keys, payload, vals, rp, rp_rows, n, 0, n_chunks, table.as<int64_t>(), total.as<int64_t>(), rlo.as<int32_t>(),
            rhi.as<int32_t>(), nullptr, rowA.as<int32_t>(), v_mid);
        CSRK_LAUNCH_CHECK();
        rx_align_kernel<<<1, 256, 0, s>>>(total.as<int64_t>(), n_chunks2, cstart.as<int64_t>(), ccnt.as<int32_t>(), run0.as<int64_t>());
        CSRK_LAUNCH_CHECK();
        rx_hist_kernel<<<(unsigned)ceil_div(n_chunks2, RX_HC), RX_THREADS, 0, s>>>(rowA.as<int32_t>(), n, 24, n_chunks2, table.as<int64_t>(),
                                                                 cstart.as<int64_t>(), ccnt.as<int32_t>());
        CSRK_LAUNCH_CHECK();
        rx_scan_kernel<<<256, RXS_THREADS, 0, s>>>(table.as<int64_t>(), n_chunks2, total2.as<int64_t>());
        CSRK_LAUNCH_CHECK();
        constexpr int VMID2 = HAS_V ? CSRK_VAL_F64 : CSRK_VAL_NONE;
        rx_scatter_kernel<P, VMID2, 0, 2><<<(unsigned)(8 * ceil_div(n_chunks2, 8)), RX_THREADS, 0, s>>>(
            rowA.as<int32_t>(), nullptr, v_mid, rp, rp_rows, n, 24, n_chunks2, table.as<int64_t>(), total2.as<int64_t>(), nullptr,
            nullptr, nullptr, out_payload, out_vals, cstart.as<int64_t>(), ccnt.as<int32_t>());
        CSRK_LAUNCH_CHECK();
        rx_rowptr_from_table_kernel<P><<<(unsigned)ceil_div((int64_t)key_range + 1, 256), 256, 0, s>>>(
            table.as<int64_t>(), total2.as<int64_t>(), run0.as<int64_t>(), n_chunks2, key_range, n, out_ptr);
        CSRK_LAUNCH_CHECK();
        CSRK_HIP(hipStreamSynchronize(s));   // temporaries go back to the pool on return
        return CSRK_OK;
    }
    CSRK_TRY(keyL.alloc((size_t)n * 4));          // sorted keys of the last pass -> run starts
    if (passes > 1) {
        CSRK_TRY(keyA.alloc((size_t)n * 4));
        CSRK_TRY(rowA.alloc((size_t)n * 4));
        if (HAS_V) CSRK_TRY(valA.alloc((size_t)n * 8));
    }
    if (passes > 2) {
        CSRK_TRY(keyB.alloc((size_t)n * 4));
        CSRK_TRY(rowB.alloc((size_t)n * 4));
        if (HAS_V) CSRK_TRY(valB.alloc((size_t)n * 8));
    }

    const int32_t *k_in = keys;
    const int32_t *r_in = payload;
    const void *v_in = vals;
    for (int p = 0; p < passes; p++) {
        const bool first = p == 0, last = p == passes - 1;
        const int shift = 8 * p;
        int32_t *k_out = last ? keyL.as<int32_t>() : ((p & 1) ? keyB.as<int32_t>() : keyA.as<int32_t>());
        int32_t *r_out = last ? out_payload : ((p & 1) ? rowB.as<int32_t>() : rowA.as<int32_t>());
        double *v_out = !HAS_V ? nullptr : (last ? out_vals : ((p & 1) ? valB.as<double>() : valA.as<double>()));
        rx_hist_kernel<<<(unsigned)ceil_div(n_chunks, RX_HC), RX_THREADS, 0, s>>>(k_in, n, shift, n_chunks, table.as<int64_t>(), nullptr, nullptr);
        CSRK_LAUNCH_CHECK();
        rx_scan_kernel<<<256, RXS_THREADS, 0, s>>>(table.as<int64_t>(), n_chunks, total.as<int64_t>());
        CSRK_LAUNCH_CHECK();
        const unsigned grid = (unsigned)(8 * ceil_div(n_chunks, 8));
        constexpr int VMID = HAS_V ? CSRK_VAL_F64 : CSRK_VAL_NONE;   // intermediates are float64
#define RX_ARGS                                                                                                     \
    k_in, r_in, v_in, rp, rp_rows, n, shift, n_chunks, table.as<int64_t>(), total.as<int64_t>(), rlo.as<int32_t>(),    \
        rhi.as<int32_t>(), k_out, r_out, v_out
        if (first)
            rx_scatter_kernel<P, VT, FROM_CSR ? 1 : 2><<<grid, RX_THREADS, 0, s>>>(RX_ARGS);
        else
            rx_scatter_kernel<P, VMID, 0><<<grid, RX_THREADS, 0, s>>>(RX_ARGS);
#undef RX_ARGS
        CSRK_LAUNCH_CHECK();
        k_in = k_out;
        r_in = r_out;
        v_in = v_out;
    }
    if (out_ptr) {
        rowptr_from_sorted_keys<P><<<(unsigned)ceil_div(ceil_div(n + 1, 4), 256), 256, 0, s>>>(keyL.as<int32_t>(), n, key_range, out_ptr);
        CSRK_LAUNCH_CHECK();
    }
    CSRK_HIP(hipStreamSynchronize(s));   // temporaries go back to the pool on return
    return CSRK_OK;
}

template <class P, int VT>
static int transpose_impl(Matrix *a, Matrix *t, hipStream_t s)
{
    return sort_records<P, VT, true>(a->d_colinds, nullptr, a->d_values, (const P *)a->d_rowptrs, a->nrows, a->nnz,
                                     a->ncols, a->nrows, (P *)t->d_rowptrs, t->d_colinds, (double *)t->d_values, s);
}

// Stable sort of the int32 payloads by int32 keys in [0, key_range) -- the same radix passes, structure only.  Exposed to
// spgemm_order.hip (the reference's output order of mult_ab).
int stable_sort_payload_by_key(const int32_t *keys, const int32_t *payload, int64_t n, int32_t key_range,
                               int64_t payload_range, int32_t *out_payload, hipStream_t s)
{
    // (the run starts are not wanted: with a key range of 2^30 they would be 8 GB)
    DevBuf ptrs;
    const bool packed_route = key_range > 256 && key_range <= 65536 && payload_range <= (1ll << 24);      // (that route derives them anyway)
    if (packed_route) CSRK_TRY(ptrs.alloc((size_t)((int64_t)key_range + 1) * 8));
    return sort_records<int64_t, CSRK_VAL_NONE, false>(keys, payload, nullptr, (const int64_t *)nullptr, 0, n, key_range,
                                                       payload_range, packed_route ? ptrs.as<int64_t>() : (int64_t *)nullptr,
                                                       out_payload, nullptr, s);
}

// Stable sort of (key, int32 payload, float64 value) records by key in [0, key_range): the sorted payloads and values,
// and the run start of every key in out_ptr[0 .. key_range] (the COO-ingest form of the radix passes).  Exposed to
// spmm_dense.hip (the heavy rows' entries bucketed by (column tile, row group, wavefront)).
int stable_sort_records_f64(const int32_t *keys, const int32_t *payload, const double *vals, int64_t n, int32_t key_range,
                            int64_t payload_range, int64_t *out_ptr, int32_t *out_payload, double *out_vals, hipStream_t s)
{
    return sort_records<int64_t, CSRK_VAL_F64, false>(keys, payload, vals, (const int64_t *)nullptr, 0, n, key_range,
                                                      payload_range, out_ptr, out_payload, out_vals, s);
}

// Transpose `a` into a new matrix.  Exposed to the other translation units (spgemm_abt,
// order_columns).
int transpose_matrix(Matrix *a, int with_values, Matrix **out, hipStream_t s)
{
    int vt = (with_values && a->val_type != CSRK_VAL_NONE) ? CSRK_VAL_F64 : CSRK_VAL_NONE;
    Matrix *t = nullptr;
    CSRK_TRY(new_matrix(a->ncols, a->nrows, a->nnz, a->ptr64, vt, &t));
    int rc;
    int in_vt = vt == CSRK_VAL_NONE ? CSRK_VAL_NONE : a->val_type;
    if (a->ptr64) {
        if (in_vt == CSRK_VAL_F64) rc = transpose_impl<int64_t, CSRK_VAL_F64>(a, t, s);
        else if (in_vt == CSRK_VAL_F32) rc = transpose_impl<int64_t, CSRK_VAL_F32>(a, t, s);
        else rc = transpose_impl<int64_t, CSRK_VAL_NONE>(a, t, s);
    } else {
        if (in_vt == CSRK_VAL_F64) rc = transpose_impl<int32_t, CSRK_VAL_F64>(a, t, s);
        else if (in_vt == CSRK_VAL_F32) rc = transpose_impl<int32_t, CSRK_VAL_F32>(a, t, s);
        else rc = transpose_impl<int32_t, CSRK_VAL_NONE>(a, t, s);
    }
    if (rc != CSRK_OK) {
        delete t;
        return rc;
    }
    *out = t;
    return CSRK_OK;
}

}  // namespace csrk

using namespace csrk;

__global__ void coo_cast_f64_to_f32(const double *__restrict__ in, float *__restrict__ out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];       // exact: the doubles are widened floats
}

// COO -> CSR: a stable sort of the entries by row is exactly the reference's counting sort
// (csr/structure.py:11-58: entries of a row keep their input order).
extern "C" int csrk_from_coo(int32_t nrows, int32_t ncols, int64_t nnz, const int32_t *rows, const int32_t *cols,
                             const void *values, int val_type, csrk_handle_t *out)
{
    CSRK_REQUIRE(out, "out is NULL");
    *out = 0;
    CSRK_REQUIRE(nrows >= 0 && ncols >= 0 && nnz >= 0, "negative dimension");
    CSRK_REQUIRE(nnz == 0 || (rows && cols), "rows / cols is NULL");
    CSRK_REQUIRE(val_type == CSRK_VAL_NONE || val_type == CSRK_VAL_F32 || val_type == CSRK_VAL_F64, "unknown val_type %d", val_type);
    CSRK_REQUIRE(val_type == CSRK_VAL_NONE || nnz == 0 || values, "values is NULL but val_type=%d", val_type);
    const int ptr64 = nnz > INT32_MAX;
    Matrix *m = nullptr;
    CSRK_TRY(new_matrix(nrows, ncols, nnz, ptr64, val_type, &m));
    DevBuf d_rows, d_cols, d_vals, d_v64;
    int rc = CSRK_OK;
    const size_t vb = val_type == CSRK_VAL_F64 ? 8 : 4;
    do {
        if ((rc = d_rows.alloc((size_t)nnz * 4)) != CSRK_OK) break;
        if ((rc = d_cols.alloc((size_t)nnz * 4)) != CSRK_OK) break;
        hipError_t e = hipSuccess;
        if (nnz) e = hipMemcpy(d_rows.p, rows, (size_t)nnz * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess && nnz) e = hipMemcpy(d_cols.p, cols, (size_t)nnz * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess && val_type != CSRK_VAL_NONE) {
            if ((rc = d_vals.alloc((size_t)nnz * vb)) != CSRK_OK) break;
            if (nnz) e = hipMemcpy(d_vals.p, values, (size_t)nnz * vb, hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) {
            set_error("host-to-device copy failed: %s", hipGetErrorString(e));
            rc = CSRK_ERR_HIP;
            break;
        }
        double *v_out = (double *)m->d_values;
        if (val_type == CSRK_VAL_F32) {
            if ((rc = d_v64.alloc((size_t)nnz * 8)) != CSRK_OK) break;
            v_out = d_v64.as<double>();
        }
#define GO(P, VT)                                                                                                    \
    rc = sort_records<P, VT, false>(d_rows.as<int32_t>(), d_cols.as<int32_t>(), d_vals.p, (const P *)nullptr, 0, nnz,   \
                                    nrows, ncols, (P *)m->d_rowptrs, m->d_colinds, v_out, nullptr)
        if (ptr64) {
            if (val_type == CSRK_VAL_F64) GO(int64_t, CSRK_VAL_F64);
            else if (val_type == CSRK_VAL_F32) GO(int64_t, CSRK_VAL_F32);
            else GO(int64_t, CSRK_VAL_NONE);
        } else {
            if (val_type == CSRK_VAL_F64) GO(int32_t, CSRK_VAL_F64);
            else if (val_type == CSRK_VAL_F32) GO(int32_t, CSRK_VAL_F32);
            else GO(int32_t, CSRK_VAL_NONE);
        }
#undef GO
        if (rc == CSRK_OK && val_type == CSRK_VAL_F32 && nnz) {
            coo_cast_f64_to_f32<<<(unsigned)ceil_div(nnz, 256), 256>>>(d_v64.as<double>(), (float *)m->d_values, nnz);
            if (hipDeviceSynchronize() != hipSuccess) {
                set_error("from_coo value cast failed");
                rc = CSRK_ERR_HIP;
            }
        }
    } while (0);
    if (rc != CSRK_OK) {
        delete m;
        return rc;
    }
    *out = to_handle(m);
    return CSRK_OK;
}

extern "C" int csrk_transpose(csrk_handle_t h, int with_values, csrk_handle_t *out)
{
    CSRK_REQUIRE(out, "out is NULL");
    *out = 0;
    Matrix *a = from_handle(h);
    if (!a) return CSRK_ERR_INVALID;
    Matrix *t = nullptr;
    CSRK_TRY(transpose_matrix(a, with_values, &t, nullptr));
    *out = to_handle(t);
    return CSRK_OK;
}
