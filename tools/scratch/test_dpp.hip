// scratch: checks the DPP-based wavefront primitives of csrc/wave.h against their __shfl forms on the GPU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../csr_amd/csrc/wave.h"
using namespace csrk;

__global__ void k(const int *xi, const double *xd, const int *fl, int *o_scan, double *o_seg, int *o_up, double *o_upd, int *o_dn, int *o_last)
{
    const int lane = threadIdx.x & 63, g = blockIdx.x * blockDim.x + threadIdx.x;
    o_scan[g] = wave_exscan_i32(xi[g], lane);
    o_seg[g] = wave_segscan(xd[g], fl[g] != 0, lane);
    o_up[g] = wave_up1_i32(xi[g], -7);
    o_upd[g] = wave_up1_f64(xd[g], -7.0);
    o_dn[g] = wave_down1_i32(xi[g], -9);
    o_last[g] = wave_last_i32(xi[g]);
}

int main()
{
    const int N = 64 * 64;
    std::vector<int> xi(N), fl(N);
    std::vector<double> xd(N);
    srand(5);
    for (int i = 0; i < N; i++) {
        xi[i] = rand() % 9;
        xd[i] = (double)(rand() % 1000) / 7.0 - 50.0;
        int w = i / 64;
        fl[i] = (w == 0) ? 0 : (w == 1 ? 1 : (rand() % (1 + w % 13) == 0));
    }
    int *dxi, *dfl, *dscan, *dup, *ddn, *dlast;
    double *dxd, *dseg, *dupd;
    hipMalloc(&dxi, N * 4); hipMalloc(&dfl, N * 4); hipMalloc(&dscan, N * 4); hipMalloc(&dup, N * 4); hipMalloc(&ddn, N * 4); hipMalloc(&dlast, N * 4);
    hipMalloc(&dxd, N * 8); hipMalloc(&dseg, N * 8); hipMalloc(&dupd, N * 8);
    hipMemcpy(dxi, xi.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(dfl, fl.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(dxd, xd.data(), N * 8, hipMemcpyHostToDevice);
    k<<<N / 256, 256>>>(dxi, dxd, dfl, dscan, dseg, dup, dupd, ddn, dlast);
    std::vector<int> scan(N), up(N), dn(N), last(N);
    std::vector<double> seg(N), upd(N);
    hipMemcpy(scan.data(), dscan, N * 4, hipMemcpyDeviceToHost);
    hipMemcpy(up.data(), dup, N * 4, hipMemcpyDeviceToHost);
    hipMemcpy(dn.data(), ddn, N * 4, hipMemcpyDeviceToHost);
    hipMemcpy(last.data(), dlast, N * 4, hipMemcpyDeviceToHost);
    hipMemcpy(seg.data(), dseg, N * 8, hipMemcpyDeviceToHost);
    hipMemcpy(upd.data(), dupd, N * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < N / 64; w++) {
        int run = 0;
        double acc = 0.0;
        for (int l = 0; l < 64; l++) {
            const int i = w * 64 + l;
            if (scan[i] != run) bad++, printf("scan w%d l%d got %d want %d\n", w, l, scan[i], run);
            run += xi[i];
            if (fl[i]) acc = 0.0;
            acc += xd[i];
            if (fabs(seg[i] - acc) > 1e-9 * (1 + fabs(acc))) bad++, printf("seg w%d l%d got %g want %g\n", w, l, seg[i], acc);
            if (up[i] != (l ? xi[i - 1] : -7)) bad++, printf("up w%d l%d\n", w, l);
            if (upd[i] != (l ? xd[i - 1] : -7.0)) bad++, printf("upd w%d l%d\n", w, l);
            if (dn[i] != (l < 63 ? xi[i + 1] : -9)) bad++, printf("dn w%d l%d got %d\n", w, l, dn[i]);
            if (last[i] != xi[w * 64 + 63]) bad++, printf("last w%d l%d\n", w, l);
            if (bad > 20) return 1;
        }
    }
    printf(bad ? "FAILED %d\n" : "dpp primitives ok\n", bad);
    return bad != 0;
}
