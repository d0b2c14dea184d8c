// The reference's column order inside the rows of a sparse product, on request.
//
// _sym_mm (csr/kernels/numba/multiply.py:60-100) walks row i of A entry by entry and, for each, row j of B entry by
// entry; a column seen for the first time is pushed onto the FRONT of the row's linked list (:79-82), and the list is
// copied out front to back (:94-97).  So a row of the reference's product holds its columns in REVERSE order of first
// discovery.  libcsrk's SpGEMM kernels emit ascending columns; no test of the reference pins the
// order (every one densifies or sorts), but its raw output is what a caller gets from from_handle before sort_rows().
// With CSRK_SPGEMM_ORDER=reference (or csrk_spgemm_set_order(1)) this pass re-orders a finished product:
//
//   1. key[e] for every entry e = (i, k) of C: the smallest (position of the A entry inside row i, position inside
//      the row of B) over the products that land on it -- the product that discovers k.  One wavefront per A entry
//      walks the B row; the entry of C is found by bisection (C's rows ascend), the key kept by a 64-bit atomic min
//      (an integer minimum: any order gives the same result).
//   2. entries sorted by (row, key descending): three stable radix sorts of an index permutation (transpose.hip's
//      passes) -- by B position, by A position, by row --, then one gather of columns and values.
//
// Values are not touched: each keeps the bits the product kernels gave it.  tests/test_gpu_ops.py compares the result
// with the reference's own raw arrays (tests/golden/spgemm.npz, c*_raw_*) bit for bit.
#include "common.h"

#include <atomic>

namespace csrk {

int stable_sort_payload_by_key(const int32_t *keys, const int32_t *payload, int64_t n, int32_t key_range,
                               int64_t payload_range, int32_t *out_payload, hipStream_t s);      // transpose.hip

static std::atomic<int> g_spgemm_order{-1};      // -1: follow CSRK_SPGEMM_ORDER; 0 ascending; 1 reference

bool spgemm_reference_order_wanted()
{
    const int o = g_spgemm_order.load();
    if (o >= 0) return o == 1;
    const char *e = getenv("CSRK_SPGEMM_ORDER");
    return e && (e[0] == 'r' || e[0] == 'R' || e[0] == '1');
}

template <class P>
__device__ __forceinline__ int64_t so_rp(const void *rp, int64_t i)
{
    return (int64_t)((const P *)rp)[i];
}

// row of every entry of a CSR (one thread per row writes its extent)
template <class P>
__global__ void so_row_of_kernel(const P *__restrict__ rp, int32_t nrows, int32_t *__restrict__ row_of)
{
    const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (w >= nrows) return;
    const int64_t s = rp[w], e = rp[w + 1];
    for (int64_t k = s + lane; k < e; k += WAVE) row_of[k] = (int32_t)w;
}

// one wavefront per entry of A: the products of that entry, in B's storage order
template <class PA, class PB>
__global__ __launch_bounds__(256) void so_first_kernel(const PA *__restrict__ a_rp, const int32_t *__restrict__ a_ci,
                                                      const int32_t *__restrict__ a_row, int64_t a_nnz,
                                                      const PB *__restrict__ b_rp, const int32_t *__restrict__ b_ci,
                                                      const int32_t *__restrict__ c_rp, const int32_t *__restrict__ c_ci,
                                                      unsigned long long *__restrict__ key)
{
    const int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x & (WAVE - 1);
    if (e >= a_nnz) return;
    const int32_t i = a_row[e], j = a_ci[e];
    const unsigned long long ea = (unsigned long long)(e - (int64_t)a_rp[i]);
    const int64_t bs = b_rp[j], be = b_rp[j + 1];
    const int32_t c0 = c_rp[i], c1 = c_rp[i + 1];
    for (int64_t t = bs + lane; t < be; t += WAVE) {
        const int32_t k = b_ci[t];
        int32_t lo = c0, hi = c1;                  // first position with c_ci >= k
        while (lo < hi) {
            const int32_t mid = lo + ((hi - lo) >> 1);
            if (c_ci[mid] < k)
                lo = mid + 1;
            else
                hi = mid;
        }
        if (lo < c1 && c_ci[lo] == k) atomicMin(&key[lo], (ea << 32) | (unsigned long long)(t - bs));
    }
}

// the two halves of the keys as descending sort keys, their maxima (for the digit counts), and the identity permutation
__global__ void so_split_kernel(const unsigned long long *__restrict__ key, int64_t n, int32_t *__restrict__ hi_part,
                                int32_t *__restrict__ lo_part, int32_t *__restrict__ ident, unsigned int *__restrict__ maxes)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int h = 0, l = 0, bad = 0;
    if (e < n) {
        const unsigned long long k = key[e];
        bad = k == ~0ull;
        h = (unsigned int)(k >> 32);
        l = (unsigned int)k;
        hi_part[e] = (int32_t)h;
        lo_part[e] = (int32_t)l;
        ident[e] = (int32_t)e;
    }
    // wavefront maxima, one atomic per wavefront and word
    for (int off = WAVE / 2; off > 0; off >>= 1) {
        const unsigned int oh = __shfl_xor(h, off, WAVE), ol = __shfl_xor(l, off, WAVE), ob = __shfl_xor(bad, off, WAVE);
        h = oh > h ? oh : h;
        l = ol > l ? ol : l;
        bad |= ob;
    }
    if ((threadIdx.x & (WAVE - 1)) == 0) {
        atomicMax(&maxes[0], bad ? 0u : h);
        atomicMax(&maxes[1], bad ? 0u : l);
        if (bad) atomicMax(&maxes[2], 1u);
    }
}

// out[q] = range - 1 - src[perm[q]]  (descending order through an ascending sort), or src[perm[q]] itself
__global__ void so_gather_key_kernel(const int32_t *__restrict__ src, const int32_t *__restrict__ perm, int64_t n, int32_t flip,
                                     int32_t *__restrict__ out)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int32_t v = src[perm ? perm[q] : q];
    out[q] = flip >= 0 ? flip - v : v;
}

__global__ void so_apply_kernel(const int32_t *__restrict__ perm, int64_t n, const int32_t *__restrict__ ci,
                                const double *__restrict__ vs, int32_t *__restrict__ oci, double *__restrict__ ovs)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int32_t e = perm[q];
    oci[q] = ci[e];
    ovs[q] = vs[e];
}

// c = a b as the product kernels left it (ascending columns, int32 row pointers, float64 values): re-ordered in place
int spgemm_apply_reference_order(Matrix *a, Matrix *b, Matrix *c)
{
    const int64_t n = c->nnz;
    if (n <= 1 || c->nrows == 0) return CSRK_OK;
    CSRK_REQUIRE(!c->ptr64 && c->val_type == CSRK_VAL_F64, "product has an unexpected layout");
    DevBuf key, a_row, hi_part, lo_part, permA, permB, keys, maxes, oci, ovs, c_row;
    CSRK_TRY(key.alloc((size_t)n * 8));
    CSRK_TRY(a_row.alloc((size_t)(a->nnz ? a->nnz : 1) * 4));
    CSRK_TRY(hi_part.alloc((size_t)n * 4));
    CSRK_TRY(lo_part.alloc((size_t)n * 4));
    CSRK_TRY(permA.alloc((size_t)n * 4));
    CSRK_TRY(permB.alloc((size_t)n * 4));
    CSRK_TRY(keys.alloc((size_t)n * 4));
    CSRK_TRY(c_row.alloc((size_t)n * 4));
    CSRK_TRY(maxes.alloc(16));
    CSRK_TRY(oci.alloc((size_t)n * 4));
    CSRK_TRY(ovs.alloc((size_t)n * 8));
    CSRK_HIP(hipMemsetAsync(key.p, 0xff, (size_t)n * 8, nullptr));
    CSRK_HIP(hipMemsetAsync(maxes.p, 0, 16, nullptr));
    const unsigned ga = (unsigned)ceil_div((int64_t)a->nrows * WAVE, 256);
    if (a->ptr64) so_row_of_kernel<int64_t><<<ga, 256>>>((const int64_t *)a->d_rowptrs, a->nrows, a_row.as<int32_t>());
    else so_row_of_kernel<int32_t><<<ga, 256>>>((const int32_t *)a->d_rowptrs, a->nrows, a_row.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    so_row_of_kernel<int32_t><<<(unsigned)ceil_div((int64_t)c->nrows * WAVE, 256), 256>>>((const int32_t *)c->d_rowptrs, c->nrows,
                                                                                         c_row.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    if (a->nnz > 0) {
        const unsigned gf = (unsigned)ceil_div(a->nnz * WAVE, 256);
#define FIRST(PA, PB)                                                                                                  \
    so_first_kernel<PA, PB><<<gf, 256>>>((const PA *)a->d_rowptrs, a->d_colinds, a_row.as<int32_t>(), a->nnz,           \
                                         (const PB *)b->d_rowptrs, b->d_colinds, (const int32_t *)c->d_rowptrs,         \
                                         c->d_colinds, key.as<unsigned long long>())
        if (a->ptr64) {
            if (b->ptr64) FIRST(int64_t, int64_t);
            else FIRST(int64_t, int32_t);
        } else {
            if (b->ptr64) FIRST(int32_t, int64_t);
            else FIRST(int32_t, int32_t);
        }
#undef FIRST
        CSRK_LAUNCH_CHECK();
    }
    const unsigned gn = (unsigned)ceil_div(n, 256);
    so_split_kernel<<<gn, 256>>>(key.as<unsigned long long>(), n, hi_part.as<int32_t>(), lo_part.as<int32_t>(),
                                 permA.as<int32_t>(), maxes.as<unsigned int>());
    CSRK_LAUNCH_CHECK();
    unsigned int mx[4] = {0, 0, 0, 0};
    CSRK_HIP(hipMemcpy(mx, maxes.p, 16, hipMemcpyDeviceToHost));
    CSRK_REQUIRE(mx[2] == 0, "an entry of the product has no product landing on it (internal error)");
    CSRK_REQUIRE(mx[0] < 0x7fffffffu && mx[1] < 0x7fffffffu, "row of A or B too long for the reference-order pass");
    // least significant key first: position inside the row of B (descending), position inside the row of A (descending), row
    so_gather_key_kernel<<<gn, 256>>>(lo_part.as<int32_t>(), nullptr, n, (int32_t)mx[1], keys.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(stable_sort_payload_by_key(keys.as<int32_t>(), permA.as<int32_t>(), n, (int32_t)mx[1] + 1, n, permB.as<int32_t>(), nullptr));
    so_gather_key_kernel<<<gn, 256>>>(hi_part.as<int32_t>(), permB.as<int32_t>(), n, (int32_t)mx[0], keys.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(stable_sort_payload_by_key(keys.as<int32_t>(), permB.as<int32_t>(), n, (int32_t)mx[0] + 1, n, permA.as<int32_t>(), nullptr));
    so_gather_key_kernel<<<gn, 256>>>(c_row.as<int32_t>(), permA.as<int32_t>(), n, -1, keys.as<int32_t>());
    CSRK_LAUNCH_CHECK();
    CSRK_TRY(stable_sort_payload_by_key(keys.as<int32_t>(), permA.as<int32_t>(), n, c->nrows, n, permB.as<int32_t>(), nullptr));
    so_apply_kernel<<<gn, 256>>>(permB.as<int32_t>(), n, c->d_colinds, (const double *)c->d_values, oci.as<int32_t>(),
                                 ovs.as<double>());
    CSRK_LAUNCH_CHECK();
    CSRK_HIP(hipMemcpyAsync(c->d_colinds, oci.p, (size_t)n * 4, hipMemcpyDeviceToDevice, nullptr));
    CSRK_HIP(hipMemcpyAsync(c->d_values, ovs.p, (size_t)n * 8, hipMemcpyDeviceToDevice, nullptr));
    CSRK_HIP(hipDeviceSynchronize());
    return CSRK_OK;
}

}  // namespace csrk

using namespace csrk;

extern "C" int csrk_spgemm_set_order(int order)
{
    CSRK_REQUIRE(order >= -1 && order <= 1, "order must be -1 (environment), 0 (ascending) or 1 (reference)");
    g_spgemm_order.store(order);
    return CSRK_OK;
}
