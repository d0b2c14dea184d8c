// Internal definitions shared by the libcsrk translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "../../include/csrk.h"

namespace csrk {

constexpr int WAVE = 64;

// ---- error plumbing ------------------------------------------------------------------
void set_error(const char *fmt, ...);

#define CSRK_HIP(call)                                                                   \
    do {                                                                                 \
        hipError_t e__ = (call);                                                         \
        if (e__ != hipSuccess) {                                                         \
            ::csrk::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),    \
                              __FILE__, __LINE__);                                       \
            return CSRK_ERR_HIP;                                                         \
        }                                                                                \
    } while (0)

#define CSRK_REQUIRE(cond, ...)                                                          \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            ::csrk::set_error(__VA_ARGS__);                                              \
            return CSRK_ERR_INVALID;                                                     \
        }                                                                                \
    } while (0)

#define CSRK_TRY(expr)                                                                   \
    do {                                                                                 \
        int rc__ = (expr);                                                               \
        if (rc__ != CSRK_OK) return rc__;                                                \
    } while (0)

// Check the launch that was just issued.
#define CSRK_LAUNCH_CHECK() CSRK_HIP(hipGetLastError())

// ---- caching device allocator (handle.hip) ---------------------------------------------------
// hipMalloc / hipFree cost 50-100 us each and hipFree synchronises the device; a transpose or SpGEMM
// call makes a dozen of each (half of a 2.8 ms transpose was allocator time).  Temporaries and
// result arrays therefore come from a process-wide pool of previously freed blocks: a request is
// served by the smallest cached block of [n, 1.5 n] bytes on the same device, freed blocks go back to
// the pool, and the pool is trimmed above 16 GiB or by csrk_trim_cache().  All pooled memory is used
// in stream order on the default stream, so a recycled block is never touched by an earlier kernel
// that is still running; launches on a caller's stream mark their handle (Matrix::used_user_stream) and the
// handle's blocks are returned only after a device-wide synchronisation.
hipError_t pool_alloc(void **p, size_t n);
void pool_free(void *p);

// Tables between the host and the card go through one pinned staging buffer (grown on demand, shared by the process, one
// copy at a time) that a small kernel reads or writes over the bus: the first hipMemcpy a process makes between the card
// and the host costs 6 ms whatever its size, and the plan of the first matrix a process multiplies paid it (a fifth of the
// plan's cost).  Both wait for the copy (and for what the stream held before it); copies of a few words go straight through.
int stage_d2h(void *host_dst, const void *dev_src, size_t n, hipStream_t s);
int stage_h2d(void *dev_dst, const void *host_src, size_t n, hipStream_t s);

// ---- device buffer with RAII -----------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) pool_free(p);
        p = nullptr;
        bytes = 0;
    }
    // Allocates at least one byte so zero-length arrays still have a valid pointer.
    int alloc(size_t n)
    {
        release();
        CSRK_HIP(pool_alloc(&p, n ? n : 16));
        bytes = n;
        return CSRK_OK;
    }
    int ensure(size_t n)
    {
        if (p && bytes >= n) return CSRK_OK;
        return alloc(n);
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
    void *take()
    {
        void *q = p;
        p = nullptr;
        bytes = 0;
        return q;
    }
};

struct SpmvPlan;   // spmv.hip
struct SpmmPlan;   // spmm_dense.hip

// ---- the matrix behind a csrk_handle_t -----------------------------------------------
struct Matrix {
    uint32_t magic = 0x4353524b;   // 'CSRK'
    int32_t nrows = 0, ncols = 0;
    int64_t nnz = 0;
    int ptr64 = 0;
    int val_type = CSRK_VAL_NONE;
    int device = 0;
    bool owns = true;
    void *d_rowptrs = nullptr;
    int32_t *d_colinds = nullptr;
    void *d_values = nullptr;

    std::mutex mu;                 // serialises plan construction and host-API scratch use
    SpmvPlan *spmv_plan = nullptr;
    int spmv_algo = CSRK_SPMV_AUTO;
    int spmv_calls = 0;            // SpMV launches on this handle (the long-row split is built on the 2nd)
    SpmmPlan *spmm_plan = nullptr;
    int dense_panel = -1;          // as the right operand of mult_ab: 1 = fully populated, rows 0 .. ncols - 1 ascending (its values ARE a row-major
                                   // panel), 0 = not, -1 = not looked at yet (spgemm_dense_b; reset with the plans by the in-place operations)
    // A launch was issued on a caller's stream: the caching allocator recycles blocks in default-stream order only,
    // and a non-blocking stream is not ordered with the default one, so the handle's memory (arrays, plans, scratch)
    // goes back to the pool only after the device has drained (csrk_free, plan invalidation, scratch growth).
    bool used_user_stream = false;

    size_t ptr_bytes() const { return ptr64 ? 8 : 4; }
    size_t val_bytes() const { return val_type == CSRK_VAL_F64 ? 8 : (val_type == CSRK_VAL_F32 ? 4 : 0); }
    ~Matrix();
};

Matrix *from_handle(csrk_handle_t h);          // nullptr (and error set) if invalid
inline csrk_handle_t to_handle(Matrix *m) { return reinterpret_cast<csrk_handle_t>(m); }
void free_spmv_plan(SpmvPlan *p);
void free_spmm_plan(SpmmPlan *p);
int64_t spmv_plan_bytes(const SpmvPlan *p);     // device memory a plan holds
int64_t spmm_plan_bytes(const SpmmPlan *p);
// Drop the handle's SpMV / SpMM plans (they hold re-ordered COPIES of colinds and values, so every operation
// that changes the matrix in place -- unit_rows, center_rows, order_columns -- must call this).  Waits for the
// device first: a launch may still be reading the plan.  Caller holds m->mu.
void invalidate_plans(Matrix *m);
// Wait for every stream of the device if a caller's stream ever launched on this handle (before its memory is recycled).
void drain_user_streams(Matrix *m);

// Create an owning matrix with freshly allocated (uninitialised) device arrays.
int new_matrix(int32_t nrows, int32_t ncols, int64_t nnz, int ptr64, int val_type, Matrix **out);

// ---- shared device-side primitives (scan.hip) --------------------------------------------
// Exclusive prefix sum of n counts into out[0..n] (out[n] = total); in-place allowed when
// in == out (then out needs n+1 slots and in[n] is ignored).  T in {int32_t, int64_t}.
int exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t n, hipStream_t s);
int exclusive_scan_i64(const int64_t *in, int64_t *out, int64_t n, hipStream_t s);
// Widen int32 counts while scanning (used when the output pointer type is int64).
int exclusive_scan_i32_to_i64(const int32_t *in, int64_t *out, int64_t n, hipStream_t s);

// Stable LSD radix sort of (key, payload...) records, used by transpose / order_columns.
// See radix.hip.

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Row-pointer load that works for both widths.
template <class P> struct PtrTraits;
template <> struct PtrTraits<int32_t> { static constexpr int is64 = 0; };
template <> struct PtrTraits<int64_t> { static constexpr int is64 = 1; };

}  // namespace csrk
