// Device-wide exclusive prefix sums for libcsrk (row-pointer construction in transpose,
// SpGEMM and filter_zeros -- the reference's sequential pointer loops,
// csr/structure.py:187-188 and csr/kernels/numba/multiply.py:92).
//
// Three-phase scan: per-block scan of 2048-element chunks -> recursive scan of the block
// totals -> uniform add.  HBM traffic: read n + write n (+ n/2048 totals), i.e. the
// minimum plus one extra write pass; fine for arrays that are <= 1/20 of the nnz streams.
#include "common.h"

namespace csrk {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_IPT = 8;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_IPT;

template <class T>
__device__ inline T wave_inclusive_scan(T v)
{
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        T o = __shfl_up(v, off, WAVE);
        if ((int)(threadIdx.x & (WAVE - 1)) >= off) v += o;
    }
    return v;
}

// out[i] = sum_{j<i} in[j] within each chunk (+ nothing else); sums[b] = chunk total.
// Logical input length is n (entries beyond n read as 0); outputs are written for i < n_out.
template <class Tin, class Tout>
__global__ __launch_bounds__(SCAN_THREADS) void scan_chunks(const Tin *__restrict__ in, Tout *__restrict__ out,
                                                           Tout *__restrict__ sums, int64_t n, int64_t n_out)
{
    __shared__ Tout wave_tot[SCAN_THREADS / WAVE];
    int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_IPT;
    Tout v[SCAN_IPT];
    Tout tsum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_IPT; k++) {
        int64_t i = base + k;
        v[k] = (i < n) ? (Tout)in[i] : (Tout)0;
        tsum += v[k];
    }
    Tout inc = wave_inclusive_scan(tsum);
    int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    if (lane == WAVE - 1) wave_tot[w] = inc;
    __syncthreads();
    Tout woff = 0;
#pragma unroll
    for (int k = 0; k < SCAN_THREADS / WAVE; k++)
        if (k < w) woff += wave_tot[k];
    Tout run = woff + inc - tsum;
#pragma unroll
    for (int k = 0; k < SCAN_IPT; k++) {
        int64_t i = base + k;
        if (i < n_out) out[i] = run;
        run += v[k];
    }
    if (threadIdx.x == SCAN_THREADS - 1 && sums) sums[blockIdx.x] = run;
}

template <class Tout>
__global__ __launch_bounds__(SCAN_THREADS) void add_chunk_offsets(Tout *__restrict__ out, const Tout *__restrict__ offs,
                                                                 int64_t n_out)
{
    Tout o = offs[blockIdx.x];
    int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_IPT;
#pragma unroll
    for (int k = 0; k < SCAN_IPT; k++) {
        int64_t i = base + k;
        if (i < n_out) out[i] += o;
    }
}

template <class Tin, class Tout>
static int scan_impl(const Tin *in, Tout *out, int64_t n, int64_t n_out, hipStream_t s)
{
    if (n_out <= 0) return CSRK_OK;
    int64_t nb = ceil_div(n_out, SCAN_CHUNK);
    if (nb == 1) {
        scan_chunks<Tin, Tout><<<1, SCAN_THREADS, 0, s>>>(in, out, (Tout *)nullptr, n, n_out);
        CSRK_LAUNCH_CHECK();
        return CSRK_OK;
    }
    DevBuf sums;
    CSRK_TRY(sums.alloc((size_t)(nb + 1) * sizeof(Tout)));
    scan_chunks<Tin, Tout><<<(unsigned)nb, SCAN_THREADS, 0, s>>>(in, out, sums.as<Tout>(), n, n_out);
    CSRK_LAUNCH_CHECK();
    CSRK_TRY((scan_impl<Tout, Tout>(sums.as<Tout>(), sums.as<Tout>(), nb, nb, s)));
    add_chunk_offsets<Tout><<<(unsigned)nb, SCAN_THREADS, 0, s>>>(out, sums.as<Tout>(), n_out);
    CSRK_LAUNCH_CHECK();
    // `sums` goes back to the caching allocator on return; reuse is ordered on the default stream.
    return CSRK_OK;
}

int exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t n, hipStream_t s)
{
    return scan_impl<int32_t, int32_t>(in, out, n, n + 1, s);
}
int exclusive_scan_i64(const int64_t *in, int64_t *out, int64_t n, hipStream_t s)
{
    return scan_impl<int64_t, int64_t>(in, out, n, n + 1, s);
}
int exclusive_scan_i32_to_i64(const int32_t *in, int64_t *out, int64_t n, hipStream_t s)
{
    return scan_impl<int32_t, int64_t>(in, out, n, n + 1, s);
}

}  // namespace csrk
