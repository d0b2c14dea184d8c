// scratch probe (GPU box): how many bytes does an 8-byte gather that misses L2 move, per load path?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_fill.bin tools/probe_fill.hip
//   tools/probe_fill.bin            (times)
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum ...
// Variants: vector global_load_dwordx2; scalar s_load_dwordx2 (constant address space, wave-uniform address);
// returning atomic (executes in L2); vector 4-byte load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef __attribute__((address_space(4))) const double cdouble;

__global__ void k_vector(const double *__restrict__ x, const uint32_t *__restrict__ idx, double *out, int64_t n)
{
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    double s = 0;
#pragma unroll 8
    for (int64_t i = t; i < n; i += stride) s += x[idx[i]];
    if (s == 1.2345) out[0] = s;
}

__global__ void k_vector4(const float *__restrict__ x, const uint32_t *__restrict__ idx, double *out, int64_t n)
{
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    float s = 0;
#pragma unroll 8
    for (int64_t i = t; i < n; i += stride) s += x[2 * (int64_t)idx[i]];
    if (s == 1.2345f) out[0] = s;
}

__global__ void k_atomic(double *x, const uint32_t *__restrict__ idx, double *out, int64_t n, unsigned long long zero)
{
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long s = 0;
#pragma unroll 8
    for (int64_t i = t; i < n; i += stride)
        s += __hip_atomic_fetch_add((unsigned long long *)&x[idx[i]], zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (s == 12345) out[0] = (double)s;
}

// one wavefront takes 64 indices with one coalesced load, then fetches x for each of them through the scalar
// cache: the address is made wave-uniform with readlane, the load goes to constant address space
__global__ void k_scalar(const double *x, const uint32_t *__restrict__ idx, double *out, int64_t n)
{
    cdouble *xc = (cdouble *)(uintptr_t)x;
    int lane = threadIdx.x & 63;
    int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    double s = 0;
    for (int64_t base = wave * 64; base < n; base += nwaves * 64) {
        uint32_t my = idx[base + lane];
#pragma unroll
        for (int j = 0; j < 64; j += 16) {
            double v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = xc[__builtin_amdgcn_readlane(my, j + q)];
#pragma unroll
            for (int q = 0; q < 16; ++q) s += v[q];
        }
    }
    if (s == 1.2345) out[0] = s;
}

// the same with G groups of 8 scalar loads in flight before one wait (hand-placed: the compiler waits after 4)
#define SL8(v, a, o) asm volatile( \
    "s_load_dwordx2 %0, %8, 0x0\n s_load_dwordx2 %1, %9, 0x0\n s_load_dwordx2 %2, %10, 0x0\n s_load_dwordx2 %3, %11, 0x0\n" \
    "s_load_dwordx2 %4, %12, 0x0\n s_load_dwordx2 %5, %13, 0x0\n s_load_dwordx2 %6, %14, 0x0\n s_load_dwordx2 %7, %15, 0x0\n" \
    : "=&s"(v[o]), "=&s"(v[o+1]), "=&s"(v[o+2]), "=&s"(v[o+3]), "=&s"(v[o+4]), "=&s"(v[o+5]), "=&s"(v[o+6]), "=&s"(v[o+7]) \
    : "s"(a[o]), "s"(a[o+1]), "s"(a[o+2]), "s"(a[o+3]), "s"(a[o+4]), "s"(a[o+5]), "s"(a[o+6]), "s"(a[o+7]))
#define WAIT8(v, o, txt) asm volatile(txt \
    : "+s"(v[o]), "+s"(v[o+1]), "+s"(v[o+2]), "+s"(v[o+3]), "+s"(v[o+4]), "+s"(v[o+5]), "+s"(v[o+6]), "+s"(v[o+7]))

template <int G>
__global__ void k_scalar_g(const double *x, const uint32_t *__restrict__ idx, double *out, int64_t n)
{
    int lane = threadIdx.x & 63;
    int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    double s = 0;
    for (int64_t base = wave * 64; base < n; base += nwaves * 64) {
        uint32_t my = idx[base + lane];
#pragma unroll
        for (int j = 0; j < 64; j += 8 * G) {
            uint64_t a[8 * G], v[8 * G];
#pragma unroll
            for (int q = 0; q < 8 * G; ++q) a[q] = (uint64_t)x + 8ull * __builtin_amdgcn_readlane(my, j + q);
#pragma unroll
            for (int g = 0; g < G; ++g) SL8(v, a, g * 8);
            WAIT8(v, 0, "s_waitcnt lgkmcnt(0)");
#pragma unroll
            for (int g = 1; g < G; ++g) WAIT8(v, g * 8, "");
#pragma unroll
            for (int q = 0; q < 8 * G; ++q) s += __longlong_as_double((long long)v[q]);
        }
    }
    if (s == 1.2345) out[0] = s;
}

// L2-resident gathers (a 3.4-MB table: the SpMV's packed columns) with the load's cache-policy bits: does a gather
// that skips L1 cost the CU less than pulling the whole 128-B line in (~4 clocks per lane)?
template <int AUX>
__global__ void k_l2(const double *xh, unsigned bytes, const uint32_t *__restrict__ idx, double *out, int64_t n, uint32_t mask)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)xh, (short)0, (int)bytes, 0x00020000);
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    double s = 0;
#pragma unroll 8
    for (int64_t i = t; i < n; i += stride) {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)((idx[i] & mask) * 8u), 0, AUX);
        s += __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
    }
    if (s == 1.2345) out[0] = s;
}

int main(int argc, char **argv)
{
    const int64_t ncols = 10000000, n = 1 << 25;
    int which = argc > 1 ? atoi(argv[1]) : -1;
    double *x, *out; uint32_t *idx;
    CK(hipMalloc(&x, ncols * 8)); CK(hipMalloc(&out, 8)); CK(hipMalloc(&idx, n * 4));
    CK(hipMemset(x, 0, ncols * 8));
    std::vector<uint32_t> h(n);
    uint64_t z = 88172645463325252ull;
    for (int64_t i = 0; i < n; ++i) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; h[i] = (uint32_t)(z % ncols); }
    CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"vector 8B", "scalar 8B", "atomic 8B", "vector 4B", "scalar x8 2048wg", "scalar x16 2048wg", "scalar x32 2048wg",
                           "scalar x16 1024wg", "scalar x16 4096wg", "scalar x32 1024wg"};
    {
        const uint32_t mask = (1u << 19) - 1;      // 524288 doubles = 4 MB... use 434k-ish: 2^19 slots, L2 + MALL resident
        const char *nm[5] = {"l2 gather default", "l2 gather sc0", "l2 gather nt", "l2 gather sc1", "l2 gather sc0 sc1"};
        for (int v = 0; v < 5 && which < 0; ++v) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                if (v == 0) k_l2<0><<<2048, 256>>>(x, (mask + 1) * 8, idx, out, n, mask);
                if (v == 1) k_l2<1><<<2048, 256>>>(x, (mask + 1) * 8, idx, out, n, mask);
                if (v == 2) k_l2<2><<<2048, 256>>>(x, (mask + 1) * 8, idx, out, n, mask);
                if (v == 3) k_l2<16><<<2048, 256>>>(x, (mask + 1) * 8, idx, out, n, mask);
                if (v == 4) k_l2<17><<<2048, 256>>>(x, (mask + 1) * 8, idx, out, n, mask);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("%-18s %8.3f ms  %6.2f ps/gather\n", nm[v], best, best * 1e9 / n);
        }
    }
    for (int v = 0; v < 10; ++v) {
        if (which >= 0 && which != v) continue;
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            if (v == 0) k_vector<<<2048, 256>>>(x, idx, out, n);
            if (v == 1) k_scalar<<<2048, 256>>>(x, idx, out, n);
            if (v == 2) k_atomic<<<2048, 256>>>(x, idx, out, n, (unsigned long long)(argc > 5));
            if (v == 3) k_vector4<<<2048, 256>>>((const float *)x, idx, out, n);
            if (v == 4) k_scalar_g<1><<<2048, 256>>>(x, idx, out, n);
            if (v == 5) k_scalar_g<2><<<2048, 256>>>(x, idx, out, n);
            if (v == 6) k_scalar_g<4><<<2048, 256>>>(x, idx, out, n);
            if (v == 7) k_scalar_g<2><<<1024, 256>>>(x, idx, out, n);
            if (v == 8) k_scalar_g<2><<<4096, 256>>>(x, idx, out, n);
            if (v == 9) k_scalar_g<4><<<1024, 256>>>(x, idx, out, n);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("%-18s %8.3f ms  %6.2f ps/gather  %7.1f GB/s at 128 B per gather\n", names[v], best, best * 1e9 / n,
               n * 128.0 / best / 1e6);
    }
    return 0;
}
