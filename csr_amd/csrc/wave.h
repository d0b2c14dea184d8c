// Wavefront-wide primitives for the gfx950 kernels, built on DPP (data-parallel primitives: a VALU operand read from
// another lane of the wavefront) instead of ds_bpermute.  __shfl_up / __shfl_down compile to ds_bpermute_b32, an LDS
// crossbar round trip of ~100+ cycles behind an s_waitcnt lgkmcnt; the short-row SpMV kernel runs one wavefront per
// tile with ~45 of them chained per tile, and a wavefront's serial latency IS that kernel's time (4 wavefronts per
// SIMD: nothing else to switch to).  A DPP move costs a few cycles and no wait.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace csrk {

// dpp_ctrl encodings (CDNA ISA, "DPP_*"): row = 16 lanes
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138;      // whole-wavefront shift by one lane
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;   // lane 15 of a row -> the next row; lane 31 -> rows 2, 3

// src as seen through the DPP control; lanes that are masked off or have no source lane get `old`
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int dpp_i32(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, false);
}

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(double old, double src)
{
    const long long o = __double_as_longlong(old), s = __double_as_longlong(src);
    const int lo = dpp_i32<CTRL, ROW_MASK, BANK_MASK>((int)o, (int)s);
    const int hi = dpp_i32<CTRL, ROW_MASK, BANK_MASK>((int)(o >> 32), (int)(s >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// value of the lane below (lane 0: `first`) / above (lane 63: `last`)
__device__ __forceinline__ int wave_up1_i32(int v, int first) { return dpp_i32<DPP_WAVE_SHR1>(first, v); }
__device__ __forceinline__ double wave_up1_f64(double v, double first) { return dpp_f64<DPP_WAVE_SHR1>(first, v); }
__device__ __forceinline__ int wave_down1_i32(int v, int last) { return dpp_i32<DPP_WAVE_SHL1>(last, v); }
// lane 63's value, in every lane (a scalar read)
__device__ __forceinline__ int wave_last_i32(int v) { return __builtin_amdgcn_readlane(v, 63); }

// exclusive prefix sum over the 64 lanes
__device__ __forceinline__ int wave_exscan_i32(int v, int /*lane*/)
{
    int s = v;
    s += dpp_i32<DPP_ROW_SHR1>(0, s);
    s += dpp_i32<DPP_ROW_SHR2>(0, s);
    s += dpp_i32<DPP_ROW_SHR4>(0, s);
    s += dpp_i32<DPP_ROW_SHR8>(0, s);                       // inclusive inside each row of 16
    s += dpp_i32<DPP_ROW_BCAST15, 0xa>(0, s);               // rows 1 and 3 += the row before
    s += dpp_i32<DPP_ROW_BCAST31, 0xc>(0, s);               // rows 2 and 3 += rows 0..1
    return s - v;
}

// inclusive prefix maximum over the 64 lanes (values >= 0)
__device__ __forceinline__ int wave_incl_max_i32(int v)
{
    int s = v;
    s = max(s, dpp_i32<DPP_ROW_SHR1>(0, s));
    s = max(s, dpp_i32<DPP_ROW_SHR2>(0, s));
    s = max(s, dpp_i32<DPP_ROW_SHR4>(0, s));
    s = max(s, dpp_i32<DPP_ROW_SHR8>(0, s));
    s = max(s, dpp_i32<DPP_ROW_BCAST15, 0xa>(0, s));
    s = max(s, dpp_i32<DPP_ROW_BCAST31, 0xc>(0, s));
    return s;
}

// Inclusive segmented sum over the lanes: a lane with `reset` set does not take the running sum of the lanes below it.
// Fixed combination tree (inside rows of 16 by distances 1, 2, 4, 8, then whole rows): bitwise reproducible.
__device__ __forceinline__ double wave_segscan(double v, bool reset, int /*lane*/)
{
    int f = reset ? 1 : 0;
#define CSRK_SEG_STEP(CTRL, RM)                                                  \
    {                                                                            \
        const double vp = dpp_f64<CTRL, RM>(0.0, v);                             \
        const int fp = dpp_i32<CTRL, RM>(0, f);                                  \
        v = f ? v : v + vp;                                                      \
        f |= fp;                                                                 \
    }
    CSRK_SEG_STEP(DPP_ROW_SHR1, 0xf)
    CSRK_SEG_STEP(DPP_ROW_SHR2, 0xf)
    CSRK_SEG_STEP(DPP_ROW_SHR4, 0xf)
    CSRK_SEG_STEP(DPP_ROW_SHR8, 0xf)
    CSRK_SEG_STEP(DPP_ROW_BCAST15, 0xa)
    CSRK_SEG_STEP(DPP_ROW_BCAST31, 0xc)
#undef CSRK_SEG_STEP
    return v;
}

}  // namespace csrk
