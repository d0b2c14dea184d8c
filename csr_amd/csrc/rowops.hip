// Row-wise operations of libcsrk on gfx950:
//   unit_rows / center_rows   csr/transform.py:29-66, :13-26   (in place on the values)
//   order_columns             csr/kernels/numba/__init__.py:47-52 -> csr/structure.py:156-169
//   filter_zeros              csr/_struct.py:61-76
//
// unit_rows / center_rows (round 3): every value is read ONCE where its row fits a register tile and twice where it does not,
// written once; reductions are accumulated in float64 whatever the storage dtype (float32 results are within one rounding
// of exact, inside the reference tests' rel 1e-6) and the element updates are done in the storage dtype with the same two
// roundings as the reference (`v *= prenorm`, then `v /= inorm`).  Three row classes, one pass over the row pointers:
//   A  rows of <= 8 entries      one LANE per row: the row's values sit in 8 registers between load, reduction and store
//                                (a power-law matrix: 9.4 of the headline matrix's 10 million rows);
//   B  rows of 9 .. 512 entries  8 lanes per row (up to 64 entries) or one WAVEFRONT per row, 8 values per lane in
//                                registers, coalesced segments;
//   C  longer rows               cut into CHUNKS of 4096 entries, one 512-thread workgroup per chunk: (C1) per-chunk
//                                partials with the chunk in registers -- its maximum, and the sum of squares under the
//                                chunk's OWN power-of-two prescale (a power-of-two factor moves through the float64 sum
//                                unchanged, so the row's sum is the chunks' sums rescaled) --, (C2) one thread per long
//                                row joins its chunks' partials in chunk order, (C3) the chunks are scaled.  A 10^6-entry
//                                row is then 245 workgroups instead of one (one workgroup per long row: 1.9 ms of the
//                                former 4.0 ms of kernels on the headline matrix, bound by its longest row on one CU).
// HBM traffic: values 1 read + 1 write (A, B), 2 reads + 1 write (C); row pointers once; norms once.
#include "common.h"

#include <atomic>

namespace csrk {

int transpose_matrix(Matrix *a, int with_values, Matrix **out, hipStream_t s);   // transpose.hip

__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return __shfl(v, 0, WAVE);
}

__device__ __forceinline__ double wmax_nan(double v, bool nan)
{
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) {
        double o = __shfl_down(v, off, WAVE);
        v = o > v ? o : v;
    }
    v = __shfl(v, 0, WAVE);
    return __any(nan) ? __builtin_nan("") : v;   // np.max propagates NaN (transform.py:52)
}

template <class T> struct FInfo;
template <> struct FInfo<double> { static constexpr int maxexp = 1024, minexp = -1022; };
template <> struct FInfo<float> { static constexpr int maxexp = 128, minexp = -126; };

constexpr int RS_A = 8;             // class A: one lane per row
constexpr int RS_K = 8;             // values per lane in classes B and C
constexpr int RS_B8 = 8 * RS_K;     // class B, 8 lanes per row (64)
constexpr int RS_B = WAVE * RS_K;   // class B, one wavefront per row (512)
#ifndef CSRK_RS_THREADS
#define CSRK_RS_THREADS 512      // (class C1 alone on the headline matrix: 1024 threads 0.357 ms, 512 0.270, 256 0.278)
#endif
constexpr int RS_THREADS = CSRK_RS_THREADS;
#ifndef CSRK_RS_C3_REVERSE
#define CSRK_RS_C3_REVERSE 1
#endif
constexpr bool RS_C3_REVERSE = CSRK_RS_C3_REVERSE != 0;
constexpr int RS_CHUNK = RS_THREADS * RS_K;      // class C: entries per chunk (4096)

// (m, e) = frexp(vmax); pnexp = clamp(-e, minexp, maxexp - 1); prenorm = 2^pnexp (transform.py:55-58); a NaN or infinite
// maximum leaves the exponent at 0
template <class T>
__device__ __forceinline__ int prenorm_exp(double vmax)
{
    int ve = 0;
    if (vmax == vmax && !isinf(vmax)) (void)frexp(vmax, &ve);
    int pnexp = -ve;
    pnexp = pnexp > FInfo<T>::maxexp - 1 ? FInfo<T>::maxexp - 1 : pnexp;
    pnexp = pnexp < FInfo<T>::minexp ? FInfo<T>::minexp : pnexp;
    return pnexp;
}

// The B (8 lanes / a wavefront per row) and C lists are built without atomics: a counting pass writes, per wavefront of
// 64 consecutive rows, how many of them fall in each class; one exclusive scan over [class][wavefront] turns the counts
// into list positions; the class A kernel -- one lane per row -- then drops its longer rows at their positions (rows
// ascend inside every list: deterministic).  (Appending with atomics serialised on the three counters: 2.0 ms of the
// kernel's 2.0 ms on the headline matrix with an atomic per row, 3.0 ms with one per wavefront and list.)
__device__ __forceinline__ int row_class(int64_t len)      // 0: class A (or no row), 1: B8, 2: B, 3: C
{
    return len <= RS_A ? 0 : (len <= RS_B8 ? 1 : (len <= RS_B ? 2 : 3));
}

template <class P>
__global__ __launch_bounds__(256) void row_class_count_kernel(const P *__restrict__ rp, int32_t nrows, int64_t n_waves,
                                                             int32_t *__restrict__ cnt)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int cls = r < nrows ? row_class((int64_t)rp[r + 1] - (int64_t)rp[r]) : 0;
    const int64_t w = r / WAVE;
    if (w >= n_waves) return;
#pragma unroll
    for (int c = 1; c <= 3; c++) {
        const unsigned long long mk = __ballot(cls == c);
        if ((threadIdx.x & (WAVE - 1)) == 0) cnt[(int64_t)(c - 1) * n_waves + w] = __popcll(mk);
    }
}

// Class A and the lists: one lane per row.  pos = the exclusive scan of the counts ([class][wavefront], then the total).
template <class P, class T, bool UNIT>
__global__ __launch_bounds__(256) void row_stat_a_kernel(const P *__restrict__ rp, T *__restrict__ vs, T *__restrict__ out,
                                                        int32_t nrows, int64_t n_waves, const int32_t *__restrict__ pos,
                                                        int32_t *__restrict__ list_b8, int32_t *__restrict__ list_b,
                                                        int32_t *__restrict__ list_c)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = r < nrows;
    const int64_t sp = ok ? (int64_t)rp[r] : 0, ep = ok ? (int64_t)rp[r + 1] : 0;
    const int64_t len = ep - sp;
    {
        const int cls = ok ? row_class(len) : 0;
        const int lane = threadIdx.x & (WAVE - 1);
        const unsigned long long below = lane ? (~0ull >> (WAVE - lane)) : 0ull;
        const int64_t w = r / WAVE;
#pragma unroll
        for (int c = 1; c <= 3; c++) {
            const unsigned long long mk = __ballot(cls == c);
            if (cls == c) {
                int32_t *list = c == 1 ? list_b8 : (c == 2 ? list_b : list_c);
                list[pos[(int64_t)(c - 1) * n_waves + w] - pos[(int64_t)(c - 1) * n_waves] + __popcll(mk & below)] = (int32_t)r;
            }
        }
    }
    if (!ok || len > RS_A) return;
    if (len == 0) {                       // empty row: norm / mean 0 (transform.py:36-38, :19-21)
        out[r] = (T)0;
        return;
    }
    T v[RS_A];
#pragma unroll
    for (int j = 0; j < RS_A; j++) v[j] = j < len ? vs[sp + j] : (T)0;
    if (UNIT) {
        double vmax = 0.0;
        bool nan = false;
#pragma unroll
        for (int j = 0; j < RS_A; j++) {
            const double a = fabs((double)v[j]);
            nan |= a != a;
            vmax = a > vmax ? a : vmax;
        }
        if (nan) vmax = __builtin_nan("");        // np.max propagates NaN (transform.py:52)
        const T prenorm = (T)ldexp(1.0, prenorm_exp<T>(vmax));
        double ss = 0.0;
#pragma unroll
        for (int j = 0; j < RS_A; j++) {
            v[j] = v[j] * prenorm;            // :59
            ss += (double)v[j] * (double)v[j];
        }
        const T inorm = (T)sqrt(ss);          // :62
        out[r] = inorm / prenorm;             // :63
#pragma unroll
        for (int j = 0; j < RS_A; j++)
            if (j < len) vs[sp + j] = v[j] / inorm;      // :64
    } else {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < RS_A; j++) s += (double)v[j];
        const T m = (T)(s / (double)len);
        out[r] = m;
#pragma unroll
        for (int j = 0; j < RS_A; j++)
            if (j < len) vs[sp + j] = v[j] - m;
    }
}

// reductions over aligned groups of LPR lanes (LPR a power of two <= 64); every lane gets the result
template <int LPR>
__device__ __forceinline__ double gsum(double v)
{
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}
template <int LPR>
__device__ __forceinline__ double gmax_nan(double v, bool nan)
{
    int f = nan ? 1 : 0;
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) {
        const double o = __shfl_xor(v, off, WAVE);
        v = o > v ? o : v;
        f |= __shfl_xor(f, off, WAVE);
    }
    return f ? __builtin_nan("") : v;   // np.max propagates NaN (transform.py:52)
}

// Class B: LPR lanes per listed row (8 for 9 .. 64 entries, 64 for 65 .. 512), the row in registers
template <class P, class T, bool UNIT, int LPR>
__global__ __launch_bounds__(256) void row_stat_b_kernel(const P *__restrict__ rp, T *__restrict__ vs, T *__restrict__ out,
                                                        const int32_t *__restrict__ list_b, int32_t n_b)
{
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPR;
    const int lane = threadIdx.x & (LPR - 1);
    if (q >= n_b) return;      // (LPR-lane groups leave together: the group shuffles below stay inside a group)
    const int32_t r = list_b[q];
    const int64_t sp = rp[r];
    const int len = (int)((int64_t)rp[r + 1] - sp);
    T v[RS_K];
#pragma unroll
    for (int j = 0; j < RS_K; j++) v[j] = j * LPR + lane < len ? vs[sp + j * LPR + lane] : (T)0;
    if (UNIT) {
        double vmax = 0.0;
        bool nan = false;
#pragma unroll
        for (int j = 0; j < RS_K; j++) {
            const double a = fabs((double)v[j]);
            nan |= a != a;
            vmax = a > vmax ? a : vmax;
        }
        vmax = gmax_nan<LPR>(vmax, nan);
        const T prenorm = (T)ldexp(1.0, prenorm_exp<T>(vmax));
        double ss = 0.0;
#pragma unroll
        for (int j = 0; j < RS_K; j++) {
            v[j] = v[j] * prenorm;
            ss += (double)v[j] * (double)v[j];
        }
        const T inorm = (T)sqrt(gsum<LPR>(ss));
        if (lane == 0) out[r] = inorm / prenorm;
#pragma unroll
        for (int j = 0; j < RS_K; j++)
            if (j * LPR + lane < len) vs[sp + j * LPR + lane] = v[j] / inorm;
    } else {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < RS_K; j++) s += (double)v[j];
        const T m = (T)(gsum<LPR>(s) / (double)len);
        if (lane == 0) out[r] = m;
#pragma unroll
        for (int j = 0; j < RS_K; j++)
            if (j * LPR + lane < len) vs[sp + j * LPR + lane] = v[j] - m;
    }
}

// Class C.  Chunk c of the listed long rows: entries [chunk_k0[c], chunk_k0[c] + chunk_len[c]) (one row's); long row i
// owns chunks row_chunk0[i] .. row_chunk0[i + 1].
template <class P>
__global__ void row_chunk_count_kernel(const P *__restrict__ rp, const int32_t *__restrict__ list_c, int32_t n_c,
                                       int32_t *__restrict__ cnt)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_c) cnt[i] = (int32_t)(((int64_t)rp[list_c[i] + 1] - (int64_t)rp[list_c[i]] + RS_CHUNK - 1) / RS_CHUNK);
}

template <class P>
__global__ void row_chunk_fill_kernel(const P *__restrict__ rp, const int32_t *__restrict__ list_c, int32_t n_c,
                                      const int32_t *__restrict__ row_chunk0, int64_t *__restrict__ chunk_k0,
                                      int32_t *__restrict__ chunk_len, int32_t *__restrict__ chunk_row)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_c) return;
    const int64_t sp = rp[list_c[i]], ep = rp[list_c[i] + 1];
    int32_t c = row_chunk0[i];
    for (int64_t k = sp; k < ep; k += RS_CHUNK, c++) {
        chunk_k0[c] = k;
        chunk_len[c] = (int32_t)(ep - k < RS_CHUNK ? ep - k : RS_CHUNK);
        chunk_row[c] = i;
    }
}

// workgroup-wide reductions: wavefront shuffles + one LDS stage, in a fixed order (deterministic)
__device__ __forceinline__ double block_sum(double v, double *s_red)
{
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    __syncthreads();
    if (lane == 0) s_red[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int k = 0; k < RS_THREADS / WAVE; k++) t += s_red[k];
    return t;
}

// C1: per-chunk partials.  UNIT: part_a = the chunk's maximum (NaN if it holds one), part_b = sum of squares of the
// chunk's values under the chunk's own prescale 2^part_e; CENTER: part_a = the chunk's sum.
// A row of ONE chunk (513 .. 4096 entries) is finished here, from the registers: read once, written once (a quarter of the
// headline matrix's class-C entries, nearly all of a MovieLens-shaped matrix's).
template <class T, bool UNIT>
__global__ __launch_bounds__(RS_THREADS) void row_stat_c1_kernel(T *__restrict__ vs, const int64_t *__restrict__ chunk_k0,
                                                                const int32_t *__restrict__ chunk_len,
                                                                const int32_t *__restrict__ chunk_row,
                                                                const int32_t *__restrict__ row_chunk0,
                                                                const int32_t *__restrict__ list_c, T *__restrict__ out,
                                                                double *__restrict__ part_a, double *__restrict__ part_b,
                                                                int32_t *__restrict__ part_e, const int32_t *__restrict__ n_chunks)
{
    __shared__ double s_red[RS_THREADS / WAVE];
    __shared__ int s_nan;
    if ((int32_t)blockIdx.x >= *n_chunks) return;      // (the grid is an upper bound of the chunk count)
    const int64_t k0 = chunk_k0[blockIdx.x];
    const int len = chunk_len[blockIdx.x];
    const int32_t row_i = chunk_row[blockIdx.x];
    const bool single = row_chunk0[row_i + 1] - row_chunk0[row_i] == 1;
    const int tid = threadIdx.x;
    double v[RS_K];
#pragma unroll
    for (int j = 0; j < RS_K; j++) v[j] = j * RS_THREADS + tid < len ? (double)vs[k0 + j * RS_THREADS + tid] : 0.0;
    if (!UNIT) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < RS_K; j++) s += v[j];
        s = block_sum(s, s_red);
        if (single) {
            const T m = (T)(s / (double)len);
#pragma unroll
            for (int j = 0; j < RS_K; j++)
                if (j * RS_THREADS + tid < len) vs[k0 + j * RS_THREADS + tid] = (T)v[j] - m;      // :24
            if (tid == 0) out[list_c[row_i]] = m;
            return;
        }
        if (tid == 0) part_a[blockIdx.x] = s;
        return;
    }
    if (tid == 0) s_nan = 0;
    __syncthreads();
    double vmax = 0.0;
    bool nan = false;
#pragma unroll
    for (int j = 0; j < RS_K; j++) {
        const double a = fabs(v[j]);
        nan |= a != a;
        vmax = a > vmax ? a : vmax;
    }
    if (nan) s_nan = 1;
    {      // the workgroup's maximum through the same LDS stage (max is order independent)
        const int lane = tid & (WAVE - 1), w = tid / WAVE;
#pragma unroll
        for (int off = WAVE / 2; off > 0; off >>= 1) {
            const double o = __shfl_down(vmax, off, WAVE);
            vmax = o > vmax ? o : vmax;
        }
        __syncthreads();
        if (lane == 0) s_red[w] = vmax;
        __syncthreads();
        vmax = 0.0;
        for (int k = 0; k < RS_THREADS / WAVE; k++) vmax = s_red[k] > vmax ? s_red[k] : vmax;
        if (s_nan) vmax = __builtin_nan("");
    }
    const int pe = prenorm_exp<T>(vmax);
    const double pre = ldexp(1.0, pe);
    double ss = 0.0;
#pragma unroll
    for (int j = 0; j < RS_K; j++) {
        const double u = (double)(T)((T)v[j] * (T)pre);      // the reference's `v *= prenorm` in the storage dtype (:59)
        ss += u * u;
    }
    ss = block_sum(ss, s_red);
    if (single) {
        const T prenorm = (T)pre, inorm = (T)sqrt(ss);      // :58, :62
#pragma unroll
        for (int j = 0; j < RS_K; j++)
            if (j * RS_THREADS + tid < len) vs[k0 + j * RS_THREADS + tid] = (T)((T)v[j] * prenorm) / inorm;      // :59, :64
        if (tid == 0) out[list_c[row_i]] = inorm / prenorm;      // :63
        return;
    }
    if (tid == 0) {
        part_a[blockIdx.x] = vmax;
        part_b[blockIdx.x] = ss;
        part_e[blockIdx.x] = pe;
    }
}

// C2: one wavefront per long row: join the chunks' partials (lane l takes chunks l, l + 64, ... in order, the lanes are
// joined by a fixed shuffle tree: deterministic).  UNIT: scale_a = prenorm, scale_b = inorm, out = inorm / prenorm;
// CENTER: scale_a = out = the mean.
template <class P, class T, bool UNIT>
__global__ __launch_bounds__(256) void row_stat_c2_kernel(const P *__restrict__ rp, const int32_t *__restrict__ list_c, int32_t n_c,
                                                         const int32_t *__restrict__ row_chunk0, const double *__restrict__ part_a,
                                                         const double *__restrict__ part_b, const int32_t *__restrict__ part_e,
                                                         T *__restrict__ scale_a, T *__restrict__ scale_b, T *__restrict__ out)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (i >= n_c) return;
    const int32_t r = list_c[i];
    const int32_t c0 = row_chunk0[i], c1 = row_chunk0[i + 1];
    if (c1 - c0 == 1) return;      // (finished by C1)
    if (!UNIT) {
        double s = 0.0;
        for (int32_t c = c0 + lane; c < c1; c += WAVE) s += part_a[c];
        const T m = (T)(wsum(s) / (double)((int64_t)rp[r + 1] - (int64_t)rp[r]));
        if (lane == 0) {
            scale_a[i] = m;
            out[r] = m;
        }
        return;
    }
    double vmax = 0.0;
    bool nan = false;
    for (int32_t c = c0 + lane; c < c1; c += WAVE) {
        const double a = part_a[c];
        nan |= a != a;
        vmax = a > vmax ? a : vmax;
    }
    vmax = wmax_nan(vmax, nan);
    const int pe = prenorm_exp<T>(vmax);
    double ss = 0.0;
    for (int32_t c = c0 + lane; c < c1; c += WAVE) ss += ldexp(part_b[c], 2 * (pe - part_e[c]));      // exact rescaling of each chunk's sum
    const T prenorm = (T)ldexp(1.0, pe);
    const T inorm = (T)sqrt(wsum(ss));
    if (lane == 0) {
        scale_a[i] = prenorm;
        scale_b[i] = inorm;
        out[r] = inorm / prenorm;
    }
}

// Exclusive scan of cnt[0 .. n) in place, cnt[n] = total: ONE workgroup, no temporaries (the long rows are a few ten
// thousand at most; the table-building chain runs on a side stream, where the pool-backed device scan must not be used)
__global__ __launch_bounds__(1024) void row_chunk_scan_kernel(int32_t *__restrict__ cnt, int32_t n)
{
    __shared__ int32_t s_w[16];
    __shared__ int32_t s_carry;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int32_t b0 = 0; b0 < n; b0 += 1024 * 8) {
        const int32_t b = b0 + tid * 8;
        int32_t v[8], t = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            v[k] = b + k < n ? cnt[b + k] : 0;
            t += v[k];
        }
        int32_t inc = t;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const int32_t o = __shfl_up(inc, off, WAVE);
            if (lane >= off) inc += o;
        }
        if (lane == WAVE - 1) s_w[w] = inc;
        __syncthreads();
        int32_t run = s_carry + inc - t;
        for (int k = 0; k < w; k++) run += s_w[k];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (b + k < n) cnt[b + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == 1023) s_carry = run;
        __syncthreads();
    }
    if (tid == 0) cnt[n] = s_carry;
}

// C3: apply
template <class T, bool UNIT>
__global__ __launch_bounds__(RS_THREADS) void row_stat_c3_kernel(T *__restrict__ vs, const int64_t *__restrict__ chunk_k0,
                                                                const int32_t *__restrict__ chunk_len,
                                                                const int32_t *__restrict__ chunk_row,
                                                                const int32_t *__restrict__ row_chunk0,
                                                                const T *__restrict__ scale_a, const T *__restrict__ scale_b,
                                                                const int32_t *__restrict__ n_chunks)
{
    if ((int32_t)blockIdx.x >= *n_chunks) return;
    // last chunk first: what C1 read last is what the 256-MiB memory-side cache still holds
    const int32_t cq = RS_C3_REVERSE ? *n_chunks - 1 - (int32_t)blockIdx.x : (int32_t)blockIdx.x;
    const int64_t k0 = chunk_k0[cq];
    const int len = chunk_len[cq];
    const int32_t i = chunk_row[cq];
    if (row_chunk0[i + 1] - row_chunk0[i] == 1) return;      // (finished by C1)
    const T a = scale_a[i], b = UNIT ? scale_b[i] : (T)0;
#pragma unroll
    for (int j = 0; j < RS_K; j++) {
        const int k = j * RS_THREADS + (int)threadIdx.x;
        if (k < len) {
            const T v = vs[k0 + k];
            vs[k0 + k] = UNIT ? (T)(v * a) / b : v - a;      // :59 then :64 / :24
        }
    }
}

// a non-blocking stream, two events and four words of pinned host memory per device, made on first use; `turn` is held by
// the call that is using them (calls on one device take turns, calls on different devices do not meet)
struct SideStream {
    hipStream_t st;
    hipEvent_t ev, ev2;
    int32_t *pinned;      // 4 words of pinned host memory a kernel on `st` can store to (hipHostMalloc)
    std::mutex *turn;
};

// the four list boundaries of the class scan -> pinned host memory, by a one-wavefront kernel on the side stream: a
// hipMemcpyAsync into pageable memory waits for the kernel running on the default stream (measured: the four copies
// started only when the class A kernel had finished, 73 us of idle card), a store from a kernel does not
__global__ void row_list_bounds_kernel(const int32_t *__restrict__ counts, int64_t n_waves, int32_t *__restrict__ out)
{
    if (threadIdx.x < 4) out[threadIdx.x] = counts[(int64_t)threadIdx.x * n_waves];
}
static int side_stream(SideStream *out)
{
    static std::mutex mu;
    static SideStream per_dev[64] = {};
    int dev = 0;
    CSRK_HIP(hipGetDevice(&dev));
    CSRK_REQUIRE(dev >= 0 && dev < 64, "device index out of range");
    std::lock_guard<std::mutex> lk(mu);
    if (!per_dev[dev].st) {
        CSRK_HIP(hipStreamCreateWithFlags(&per_dev[dev].st, hipStreamNonBlocking));
        CSRK_HIP(hipEventCreateWithFlags(&per_dev[dev].ev, hipEventDisableTiming));
        CSRK_HIP(hipEventCreateWithFlags(&per_dev[dev].ev2, hipEventDisableTiming));
        CSRK_HIP(hipHostMalloc((void **)&per_dev[dev].pinned, 4 * sizeof(int32_t), hipHostMallocDefault));
        per_dev[dev].turn = new std::mutex();
    }
    *out = per_dev[dev];
    return CSRK_OK;
}

// out_dev (optional): the norms / means stay on the device there (values dtype); out_host (optional): copied out
template <bool UNIT>
static int row_stat(Matrix *m, void *out_host, void *out_dev)
{
    CSRK_REQUIRE(m->val_type != CSRK_VAL_NONE, "matrix has no values");
    CSRK_REQUIRE(out_host || out_dev || m->nrows == 0, "output is NULL");
    if (m->nrows == 0) return CSRK_OK;
    std::lock_guard<std::mutex> lk(m->mu);
    invalidate_plans(m);          // the values change in place; the plans hold copies of them
    DevBuf d, list_b8, list_b, list_c, counts;
    const int64_t n_waves = ceil_div((int64_t)m->nrows, WAVE);
    void *out = out_dev;
    if (!out) {
        CSRK_TRY(d.alloc((size_t)m->nrows * m->val_bytes()));
        out = d.p;
    }
    // at most nnz / 9 rows reach class B, nnz / 513 class C
    CSRK_TRY(list_b8.alloc((size_t)(m->nnz / (RS_A + 1) + 1) * 4));
    CSRK_TRY(list_b.alloc((size_t)(m->nnz / (RS_B8 + 1) + 1) * 4));
    CSRK_TRY(list_c.alloc((size_t)(m->nnz / (RS_B + 1) + 1) * 4));
    CSRK_TRY(counts.alloc((size_t)(3 * n_waves + 1) * 4));
    // Every table of the long-row chain is allocated HERE, sized by its upper bound, before the first kernel that
    // rewrites values: a failed allocation later would return an error with the matrix half-normalised.  At most
    // nnz / 513 rows reach class C, and a long row has at most one partial chunk beyond its full ones.
    const int64_t n_c_max = m->nnz / (RS_B + 1) + 1;
    const int64_t n_chunks_max = n_c_max + m->nnz / RS_CHUNK;
    DevBuf rc0, ck0, clen, crow, pa, pb, pe, sa, sb;
    if (m->nnz > RS_B) {
        CSRK_TRY(rc0.alloc((size_t)(n_c_max + 2) * 4));
        CSRK_TRY(ck0.alloc((size_t)n_chunks_max * 8));
        CSRK_TRY(clen.alloc((size_t)n_chunks_max * 4));
        CSRK_TRY(crow.alloc((size_t)n_chunks_max * 4));
        CSRK_TRY(pa.alloc((size_t)n_chunks_max * 8));
        CSRK_TRY(pb.alloc((size_t)n_chunks_max * 8));
        CSRK_TRY(pe.alloc((size_t)n_chunks_max * 4));
        CSRK_TRY(sa.alloc((size_t)n_c_max * 8));
        CSRK_TRY(sb.alloc((size_t)n_c_max * 8));
    }
    const unsigned ga = (unsigned)ceil_div((int64_t)m->nrows, 256);
    SideStream side;
    CSRK_TRY(side_stream(&side));
    // The side stream, its events and its pinned words are the device's, not this call's: calls from several threads
    // (different handles) on ONE device take turns.  Nothing is lost: every call ends by draining the device.
    std::lock_guard<std::mutex> side_lk(*side.turn);
    // declared after every DevBuf above: on ANY early return the device drains (both streams) before the buffers go back to
    // the pool (the normal path ends in a synchronisation of its own and disarms this one)
    struct DrainOnExit {
        bool armed = true;
        ~DrainOnExit()
        {
            if (armed) (void)hipDeviceSynchronize();
        }
    } drain_on_exit;
#define GO(P, T)                                                                                                       \
    do {                                                                                                               \
        row_class_count_kernel<P><<<ga, 256>>>((const P *)m->d_rowptrs, m->nrows, n_waves, counts.as<int32_t>());       \
        CSRK_LAUNCH_CHECK();                                                                                           \
        CSRK_TRY(exclusive_scan_i32(counts.as<int32_t>(), counts.as<int32_t>(), 3 * n_waves, nullptr));                \
        /* the list lengths come back on a side stream WHILE the class A kernel runs: that kernel is launched first, the   \
           side stream waits only for the scan */                                                                      \
        CSRK_HIP(hipEventRecord(side.ev, nullptr));                                                                    \
        row_stat_a_kernel<P, T, UNIT><<<ga, 256>>>((const P *)m->d_rowptrs, (T *)m->d_values, (T *)out, m->nrows,       \
                                                   n_waves, counts.as<int32_t>(), list_b8.as<int32_t>(),               \
                                                   list_b.as<int32_t>(), list_c.as<int32_t>());                        \
        CSRK_LAUNCH_CHECK();                                                                                           \
        CSRK_HIP(hipEventRecord(side.ev2, nullptr));      /* the lists exist */                                        \
        CSRK_HIP(hipStreamWaitEvent(side.st, side.ev, 0));                                                             \
        int32_t *pin = side.pinned;                                                                                    \
        row_list_bounds_kernel<<<1, 64, 0, side.st>>>(counts.as<int32_t>(), n_waves, pin);                             \
        CSRK_LAUNCH_CHECK();                                                                                           \
        CSRK_HIP(hipStreamSynchronize(side.st));                                                                       \
        int32_t n_bc[4];      /* list starts in the scan: [0], [n_waves], [2 n_waves], total */                        \
        for (int c = 0; c < 4; c++) n_bc[c] = ((volatile int32_t *)pin)[c];                                            \
        const int32_t n_b8 = n_bc[1] - n_bc[0], n_b = n_bc[2] - n_bc[1], n_c = n_bc[3] - n_bc[2];                      \
        /* the long rows' chunk tables (three small kernels, the card nearly idle under them) are built on the side     \
           stream beside the class B kernels; every buffer they touch was allocated above */                           \
        const int64_t n_chunks = (int64_t)n_c + m->nnz / RS_CHUNK;      /* at most one partial chunk per long row */     \
        if (n_c > 0) {                                                                                                 \
            const unsigned gc = (unsigned)ceil_div(n_c, 256);                                                          \
            CSRK_HIP(hipStreamWaitEvent(side.st, side.ev2, 0));                                                        \
            row_chunk_count_kernel<P><<<gc, 256, 0, side.st>>>((const P *)m->d_rowptrs, list_c.as<int32_t>(), n_c,      \
                                                               rc0.as<int32_t>());                                     \
            CSRK_LAUNCH_CHECK();                                                                                       \
            row_chunk_scan_kernel<<<1, 1024, 0, side.st>>>(rc0.as<int32_t>(), n_c);                                     \
            CSRK_LAUNCH_CHECK();                                                                                       \
            row_chunk_fill_kernel<P><<<gc, 256, 0, side.st>>>((const P *)m->d_rowptrs, list_c.as<int32_t>(), n_c,       \
                                                              rc0.as<int32_t>(), ck0.as<int64_t>(), clen.as<int32_t>(), \
                                                              crow.as<int32_t>());                                     \
            CSRK_LAUNCH_CHECK();                                                                                       \
            CSRK_HIP(hipEventRecord(side.ev, side.st));                                                                \
        }                                                                                                              \
        if (n_b8 > 0) {                                                                                                \
            row_stat_b_kernel<P, T, UNIT, 8><<<(unsigned)ceil_div((int64_t)n_b8 * 8, 256), 256>>>(                     \
                (const P *)m->d_rowptrs, (T *)m->d_values, (T *)out, list_b8.as<int32_t>(), n_b8);                      \
            CSRK_LAUNCH_CHECK();                                                                                       \
        }                                                                                                              \
        if (n_b > 0) {                                                                                                 \
            row_stat_b_kernel<P, T, UNIT, WAVE><<<(unsigned)ceil_div((int64_t)n_b * WAVE, 256), 256>>>(                \
                (const P *)m->d_rowptrs, (T *)m->d_values, (T *)out, list_b.as<int32_t>(), n_b);                        \
            CSRK_LAUNCH_CHECK();                                                                                       \
        }                                                                                                              \
        if (n_c > 0) {                                                                                                 \
            CSRK_HIP(hipStreamWaitEvent(nullptr, side.ev, 0));      /* the chunk tables are ready */                   \
            row_stat_c1_kernel<T, UNIT><<<(unsigned)n_chunks, RS_THREADS>>>((T *)m->d_values, ck0.as<int64_t>(),        \
                                                                            clen.as<int32_t>(), crow.as<int32_t>(),     \
                                                                            rc0.as<int32_t>(), list_c.as<int32_t>(),    \
                                                                            (T *)out, pa.as<double>(),                  \
                                                                            pb.as<double>(), pe.as<int32_t>(),          \
                                                                            rc0.as<int32_t>() + n_c);                   \
            CSRK_LAUNCH_CHECK();                                                                                       \
            row_stat_c2_kernel<P, T, UNIT><<<(unsigned)ceil_div((int64_t)n_c * WAVE, 256), 256>>>(                     \
                                                        (const P *)m->d_rowptrs, list_c.as<int32_t>(), n_c,             \
                                                        rc0.as<int32_t>(), pa.as<double>(), pb.as<double>(),            \
                                                        pe.as<int32_t>(), sa.as<T>(), sb.as<T>(), (T *)out);            \
            CSRK_LAUNCH_CHECK();                                                                                       \
            row_stat_c3_kernel<T, UNIT><<<(unsigned)n_chunks, RS_THREADS>>>((T *)m->d_values, ck0.as<int64_t>(),        \
                                                                            clen.as<int32_t>(), crow.as<int32_t>(),     \
                                                                            rc0.as<int32_t>(), sa.as<T>(), sb.as<T>(),  \
                                                                            rc0.as<int32_t>() + n_c);                   \
            CSRK_LAUNCH_CHECK();                                                                                       \
            CSRK_HIP(hipDeviceSynchronize());      /* the chunk tables are released here */                            \
        }                                                                                                              \
    } while (0)
    if (m->ptr64) {
        if (m->val_type == CSRK_VAL_F64) GO(int64_t, double); else GO(int64_t, float);
    } else {
        if (m->val_type == CSRK_VAL_F64) GO(int32_t, double); else GO(int32_t, float);
    }
#undef GO
    CSRK_HIP(hipDeviceSynchronize());
    drain_on_exit.armed = false;      // both streams have drained
    if (out_host) CSRK_HIP(hipMemcpy(out_host, out, (size_t)m->nrows * m->val_bytes(), hipMemcpyDeviceToHost));
    return CSRK_OK;
}

// ---- filter_zeros ---------------------------------------------------------------------------
template <class C>
__global__ void nz_flag_kernel(const double *__restrict__ vs, int64_t nnz, C *__restrict__ flags)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nnz) flags[i] = vs[i] != 0.0 ? 1 : 0;   // NaN != 0 is true: NaN is kept (_struct.py:68)
}

template <class C>
__global__ void nz_compact_kernel(const int32_t *__restrict__ ci, const double *__restrict__ vs, int64_t nnz,
                                  const C *__restrict__ pos, int32_t *__restrict__ oci, double *__restrict__ ovs)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    double v = vs[i];
    if (v != 0.0) {
        C o = pos[i];
        oci[o] = ci[i];
        ovs[o] = v;
    }
}

template <class P, class C>
__global__ void nz_rowptr_kernel(const P *__restrict__ rp, int32_t nrows, const C *__restrict__ pos, P *__restrict__ orp)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= nrows) orp[r] = (P)pos[rp[r]];
}

template <class P, class C>
static int filter_impl(Matrix *m, Matrix **out)
{
    const int64_t nnz = m->nnz;
    DevBuf pos;
    CSRK_TRY(pos.alloc((size_t)(nnz + 1) * sizeof(C)));
    if (nnz > 0) {
        nz_flag_kernel<C><<<(unsigned)ceil_div(nnz, 256), 256>>>((const double *)m->d_values, nnz, pos.as<C>());
        CSRK_LAUNCH_CHECK();
    }
    if (sizeof(C) == 8)
        CSRK_TRY(exclusive_scan_i64((const int64_t *)pos.p, (int64_t *)pos.p, nnz, nullptr));
    else
        CSRK_TRY(exclusive_scan_i32((const int32_t *)pos.p, (int32_t *)pos.p, nnz, nullptr));
    C total = 0;
    CSRK_HIP(hipMemcpy(&total, pos.as<C>() + nnz, sizeof(C), hipMemcpyDeviceToHost));
    Matrix *f = nullptr;
    CSRK_TRY(new_matrix(m->nrows, m->ncols, (int64_t)total, m->ptr64, CSRK_VAL_F64, &f));
    if (nnz > 0) {
        nz_compact_kernel<C><<<(unsigned)ceil_div(nnz, 256), 256>>>(m->d_colinds, (const double *)m->d_values, nnz,
                                                                    pos.as<C>(), f->d_colinds, (double *)f->d_values);
    }
    nz_rowptr_kernel<P, C><<<(unsigned)ceil_div((int64_t)m->nrows + 1, 256), 256>>>((const P *)m->d_rowptrs, m->nrows,
                                                                                    pos.as<C>(), (P *)f->d_rowptrs);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("filter_zeros kernels failed: %s", hipGetErrorString(e));
        delete f;
        return CSRK_ERR_HIP;
    }
    *out = f;
    return CSRK_OK;
}

// ---- order_columns ----------------------------------------------------------------------------
__global__ void cast_f64_to_f32_kernel(const double *__restrict__ in, float *__restrict__ out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];   // exact: the doubles were widened floats
}

// ---- pick_rows (csr/csr.py:347-364 -> csr/structure.py:84-149) ---------------------------------------
// The reference walks the requested rows twice: sum their lengths, then copy each row's colinds (and
// values) to a running position that becomes the new row pointer.  Here: lengths -> exclusive scan ->
// one wavefront per picked row copies it.  A row may be picked more than once.  HBM-bound byte moving.
template <class P>
__global__ void pick_len_kernel(const P *__restrict__ rp, const int32_t *__restrict__ rows, int64_t nr, int32_t nrows,
                                int64_t *__restrict__ len, int32_t *__restrict__ bad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nr) return;
    int64_t l = 0;
    if (i < nr) {
        const int32_t r = rows[i];
        if (r < 0 || r >= nrows)
            atomicOr(bad, 1);
        else
            l = (int64_t)rp[r + 1] - (int64_t)rp[r];
    }
    len[i] = l;
}

// One thread per OUTPUT entry (consecutive lanes write consecutive addresses whatever the row lengths -- a
// wavefront per picked row needed 13 ms for 2M rows of the headline matrix, whose 10^6-entry rows each kept one
// wavefront busy): the workgroup's 2048 entries span the picked rows [i0, i1], found by two binary searches
// over the output offsets; each thread then searches only that span.
constexpr int PICK_EPT = 8;
template <class P, class PO, class T>
__global__ __launch_bounds__(256) void pick_copy_kernel(const P *__restrict__ rp, const int32_t *__restrict__ ci,
                                                       const T *__restrict__ vs, const int32_t *__restrict__ rows,
                                                       int64_t nr, const int64_t *__restrict__ off, int64_t nnz,
                                                       int32_t *__restrict__ oci, T *__restrict__ ovs)
{
    __shared__ int64_t s_span[2];
    const int64_t base = (int64_t)blockIdx.x * (256 * PICK_EPT);
    if (threadIdx.x < 2) {
        // last picked row whose offset is <= the workgroup's first (threadIdx 0) / last (1) entry
        int64_t o = threadIdx.x == 0 ? base : base + 256 * PICK_EPT - 1;
        o = o < nnz - 1 ? o : nnz - 1;
        int64_t lo = 0, hi = nr - 1;
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) >> 1;
            if (off[mid] <= o)
                lo = mid;
            else
                hi = mid - 1;
        }
        s_span[threadIdx.x] = lo;
    }
    __syncthreads();
    const int64_t i0 = s_span[0], i1 = s_span[1];
#pragma unroll
    for (int k = 0; k < PICK_EPT; k++) {
        const int64_t o = base + k * 256 + threadIdx.x;
        if (o >= nnz) break;
        int64_t lo = i0, hi = i1;
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) >> 1;
            if (off[mid] <= o)
                lo = mid;
            else
                hi = mid - 1;
        }
        const int64_t src = (int64_t)rp[rows[lo]] + (o - off[lo]);
        oci[o] = ci[src];
        if (vs) ovs[o] = vs[src];
    }
}

template <class PO>
__global__ void pick_ptr_kernel(const int64_t *__restrict__ off, int64_t nr, PO *__restrict__ orp)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= nr) orp[i] = (PO)off[i];
}

template <class P, class T>
static int pick_impl(Matrix *m, const int32_t *d_rows, int64_t nr, int vt, Matrix **out)
{
    DevBuf len, bad;
    CSRK_TRY(len.alloc((size_t)(nr + 2) * 8));
    CSRK_TRY(bad.alloc(4));
    CSRK_HIP(hipMemsetAsync(bad.p, 0, 4, nullptr));
    pick_len_kernel<P><<<(unsigned)ceil_div(nr + 1, 256), 256>>>((const P *)m->d_rowptrs, d_rows, nr, m->nrows,
                                                                len.as<int64_t>(), bad.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(exclusive_scan_i64(len.as<int64_t>(), len.as<int64_t>(), nr, nullptr));
    int64_t nnz = 0;
    int32_t is_bad = 0;
    CSRK_HIP(hipMemcpy(&nnz, len.as<int64_t>() + nr, 8, hipMemcpyDeviceToHost));
    CSRK_HIP(hipMemcpy(&is_bad, bad.p, 4, hipMemcpyDeviceToHost));
    CSRK_REQUIRE(!is_bad, "pick_rows: row index out of range [0, %d)", m->nrows);
    const int p64 = nnz > INT32_MAX;
    Matrix *t = nullptr;
    CSRK_TRY(new_matrix((int32_t)nr, m->ncols, nnz, p64, vt, &t));
    const unsigned gp = (unsigned)ceil_div(nr + 1, 256), grid = (unsigned)ceil_div(nnz, 256 * PICK_EPT);
    const T *vs = vt == CSRK_VAL_NONE ? (const T *)nullptr : (const T *)m->d_values;
    if (p64)
        pick_ptr_kernel<int64_t><<<gp, 256>>>(len.as<int64_t>(), nr, (int64_t *)t->d_rowptrs);
    else
        pick_ptr_kernel<int32_t><<<gp, 256>>>(len.as<int64_t>(), nr, (int32_t *)t->d_rowptrs);
    if (nnz > 0) {
        if (p64)
            pick_copy_kernel<P, int64_t, T><<<grid, 256>>>((const P *)m->d_rowptrs, m->d_colinds, vs, d_rows, nr,
                                                          len.as<int64_t>(), nnz, t->d_colinds, (T *)t->d_values);
        else
            pick_copy_kernel<P, int32_t, T><<<grid, 256>>>((const P *)m->d_rowptrs, m->d_colinds, vs, d_rows, nr,
                                                          len.as<int64_t>(), nnz, t->d_colinds, (T *)t->d_values);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();      // `len` goes back to the pool on return
    if (e != hipSuccess) {
        set_error("pick_rows failed: %s", hipGetErrorString(e));
        delete t;
        return CSRK_ERR_HIP;
    }
    *out = t;
    return CSRK_OK;
}

}  // namespace csrk

using namespace csrk;

extern "C" {

int csrk_unit_rows(csrk_handle_t h, void *norms)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    return row_stat<true>(m, norms, nullptr);
}

int csrk_center_rows(csrk_handle_t h, void *means)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    return row_stat<false>(m, means, nullptr);
}

int csrk_unit_rows_device(csrk_handle_t h, void *d_norms)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    return row_stat<true>(m, nullptr, d_norms);
}

int csrk_center_rows_device(csrk_handle_t h, void *d_means)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    return row_stat<false>(m, nullptr, d_means);
}

int csrk_pick_rows(csrk_handle_t h, const int32_t *rows, int64_t n_rows, int with_values, csrk_handle_t *out)
{
    CSRK_REQUIRE(out, "out is NULL");
    *out = 0;
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE(n_rows >= 0 && n_rows <= INT32_MAX - 1, "bad row count %lld", (long long)n_rows);
    CSRK_REQUIRE(rows || n_rows == 0, "rows is NULL");
    DevBuf d_rows;
    CSRK_TRY(d_rows.alloc((size_t)(n_rows ? n_rows : 1) * 4));
    if (n_rows) CSRK_HIP(hipMemcpy(d_rows.p, rows, (size_t)n_rows * 4, hipMemcpyHostToDevice));
    const int vt = (with_values && m->val_type != CSRK_VAL_NONE) ? m->val_type : CSRK_VAL_NONE;
    std::lock_guard<std::mutex> lk(m->mu);
    Matrix *t = nullptr;
    int rc;
    if (m->ptr64)
        rc = vt == CSRK_VAL_F32 ? pick_impl<int64_t, float>(m, d_rows.as<int32_t>(), n_rows, vt, &t)
                                : pick_impl<int64_t, double>(m, d_rows.as<int32_t>(), n_rows, vt, &t);
    else
        rc = vt == CSRK_VAL_F32 ? pick_impl<int32_t, float>(m, d_rows.as<int32_t>(), n_rows, vt, &t)
                                : pick_impl<int32_t, double>(m, d_rows.as<int32_t>(), n_rows, vt, &t);
    if (rc != CSRK_OK) return rc;
    *out = to_handle(t);
    return CSRK_OK;
}

int csrk_filter_zeros(csrk_handle_t h, csrk_handle_t *out)
{
    CSRK_REQUIRE(out, "out is NULL");
    *out = 0;
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE(m->val_type == CSRK_VAL_F64, "filter_zeros needs float64 values");
    std::lock_guard<std::mutex> lk(m->mu);
    Matrix *f = nullptr;
    int rc;
    if (m->ptr64)
        rc = filter_impl<int64_t, int64_t>(m, &f);
    else
        rc = filter_impl<int32_t, int32_t>(m, &f);
    if (rc != CSRK_OK) return rc;
    *out = to_handle(f);
    return CSRK_OK;
}

// Sorting every row by column, stably, is what two stable transposes do: the first orders
// entries by (col, source position), the second by (row, col, source position).
int csrk_order_columns(csrk_handle_t h)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    if (m->nnz == 0) return CSRK_OK;
    std::lock_guard<std::mutex> lk(m->mu);
    Matrix *t1 = nullptr, *t2 = nullptr;
    CSRK_TRY(transpose_matrix(m, 1, &t1, nullptr));
    int rc = transpose_matrix(t1, 1, &t2, nullptr);
    delete t1;
    if (rc != CSRK_OK) return rc;
    hipError_t e = hipMemcpy(m->d_colinds, t2->d_colinds, (size_t)m->nnz * 4, hipMemcpyDeviceToDevice);
    if (e == hipSuccess && m->val_type == CSRK_VAL_F64)
        e = hipMemcpy(m->d_values, t2->d_values, (size_t)m->nnz * 8, hipMemcpyDeviceToDevice);
    if (e == hipSuccess && m->val_type == CSRK_VAL_F32) {
        cast_f64_to_f32_kernel<<<(unsigned)ceil_div(m->nnz, 256), 256>>>((const double *)t2->d_values,
                                                                         (float *)m->d_values, m->nnz);
        e = hipDeviceSynchronize();
    }
    delete t2;
    if (e != hipSuccess) {
        set_error("order_columns copy-back failed: %s", hipGetErrorString(e));
        return CSRK_ERR_HIP;
    }
    invalidate_plans(m);      // the SpMV / SpMM plans hold re-ordered copies of colinds and values
    return CSRK_OK;
}

}  // extern "C"
