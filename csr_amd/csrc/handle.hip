// Handle lifecycle of libcsrk: create / free / export / info, plus the trivial row-pointer
// queries.  Replaces to_handle / from_handle / release_handle of the reference's kernel
// protocol (csr/kernels/numba/__init__.py:16-44; csr/kernels/mkl/handle.py:61-148) and
// row_extent / row_nnzs (csr/_rows.py:9-13, csr/csr.py:432-441).
#include "common.h"

#include <cstring>
#include <ctime>

#include <map>
#include <unordered_map>
#include <unordered_set>

namespace csrk {

static thread_local std::string g_last_error;

// Live handles: a csrk_handle_t coming through the C ABI is validated against this set, so a
// stale or garbage integer is an error return, never a wild pointer dereference.
static std::mutex g_live_mu;
static std::unordered_set<const Matrix *> &live_set()
{
    static std::unordered_set<const Matrix *> s;
    return s;
}
static void register_matrix(const Matrix *m)
{
    std::lock_guard<std::mutex> lk(g_live_mu);
    live_set().insert(m);
}

// ---- caching device allocator -------------------------------------------------------------------
namespace {
struct PoolBlock {
    void *p;
    size_t bytes;
    int device;
};
std::mutex g_pool_mu;
std::multimap<size_t, PoolBlock> g_pool_free;              // cached blocks by size
std::unordered_map<void *, PoolBlock> g_pool_live;         // blocks handed out
size_t g_pool_cached = 0;
constexpr size_t POOL_CAP = 16ull << 30;

void pool_trim_locked(size_t keep)
{
    while (g_pool_cached > keep && !g_pool_free.empty()) {
        auto it = std::prev(g_pool_free.end());            // largest first
        g_pool_cached -= it->second.bytes;
        (void)hipFree(it->second.p);
        g_pool_free.erase(it);
    }
}
}  // namespace

hipError_t pool_alloc(void **out, size_t n)
{
    n = (n + 255) & ~(size_t)255;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (auto it = g_pool_free.lower_bound(n); it != g_pool_free.end() && it->first <= n + n / 2; ++it) {
            if (it->second.device != dev) continue;
            PoolBlock b = it->second;
            g_pool_free.erase(it);
            g_pool_cached -= b.bytes;
            g_pool_live[b.p] = b;
            *out = b.p;
            return hipSuccess;
        }
    }
    void *q = nullptr;
    static const bool trace = getenv("CSRK_PLAN_TRACE") != nullptr;      // (with the plan builders' laps: what the driver's allocations cost)
    timespec t0{}, t1{};
    if (trace) clock_gettime(CLOCK_MONOTONIC, &t0);
    e = hipMalloc(&q, n);
    if (trace) {
        clock_gettime(CLOCK_MONOTONIC, &t1);
        fprintf(stderr, "[csrk pool] hipMalloc %10.3f MB %8.3f ms\n", n / 1048576.0,
                (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6);
    }
    if (e != hipSuccess) {                                  // out of memory: give the cache back and retry
        (void)hipGetLastError();
        std::lock_guard<std::mutex> lk(g_pool_mu);
        pool_trim_locked(0);
        e = hipMalloc(&q, n);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_live[q] = PoolBlock{q, n, dev};
    *out = q;
    return hipSuccess;
}

void pool_free(void *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pool_live.find(p);
    if (it == g_pool_live.end()) {                          // not ours (should not happen)
        (void)hipFree(p);
        return;
    }
    PoolBlock b = it->second;
    g_pool_live.erase(it);
    g_pool_free.emplace(b.bytes, b);
    g_pool_cached += b.bytes;
    if (g_pool_cached > POOL_CAP) pool_trim_locked(POOL_CAP / 2);
}

namespace {
// One pinned staging buffer per DEVICE (plan builds on different GPUs -- dist.py's ranks in one process, nogil callers --
// do not wait for each other), grown on demand up to STAGE_CAP; larger tables go through the runtime's own copy.
constexpr int STAGE_MAX_DEV = 16;
constexpr size_t STAGE_DIRECT = 1024;          // copies of at most this many bytes: the runtime's own path
constexpr size_t STAGE_CAP = 64u << 20;        // ... and of more than this many
struct Stage {
    std::mutex mu;
    void *p = nullptr;
    size_t bytes = 0;
};
Stage g_stages[STAGE_MAX_DEV];

Stage *stage_of_current_device()
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    return &g_stages[(dev < 0 ? 0 : dev) % STAGE_MAX_DEV];
}

int stage_room_locked(Stage *st, size_t n)
{
    if (st->bytes >= n) return CSRK_OK;
    if (st->p) (void)hipHostFree(st->p);
    st->p = nullptr;
    st->bytes = 0;
    size_t want = 1u << 20;
    while (want < n) want <<= 1;
    CSRK_HIP(hipHostMalloc(&st->p, want, hipHostMallocPortable | hipHostMallocMapped));
    st->bytes = want;
    return CSRK_OK;
}

// the bytes move by a kernel that reads or writes the pinned buffer over the bus (the first hipMemcpy of a process between
// the card and the host costs 6 ms whatever its size -- the runtime sets its copy engine up --, and the plan of the first
// matrix a process multiplies paid it)
__global__ void stage_copy_kernel(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, size_t n, bool words)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (words) {
        if (i < n / 4) ((uint32_t *)dst)[i] = ((const uint32_t *)src)[i];
    } else if (i < n) {
        dst[i] = src[i];
    }
}

int stage_copy(const void *src, void *dst, size_t n, hipStream_t s)
{
    const bool words = n % 4 == 0 && (uintptr_t)src % 4 == 0 && (uintptr_t)dst % 4 == 0;
    const size_t items = words ? n / 4 : n;
    stage_copy_kernel<<<(unsigned)((items + 255) / 256), 256, 0, s>>>((const unsigned char *)src, (unsigned char *)dst, n, words);
    CSRK_LAUNCH_CHECK();
    CSRK_HIP(hipStreamSynchronize(s));
    return CSRK_OK;
}
}  // namespace

// the pinned staging buffers go back to the runtime (csrk_trim_cache)
void stage_release_all()
{
    for (Stage &st : g_stages) {
        std::lock_guard<std::mutex> lk(st.mu);
        if (st.p) (void)hipHostFree(st.p);
        st.p = nullptr;
        st.bytes = 0;
    }
}

int stage_d2h(void *host_dst, const void *dev_src, size_t n, hipStream_t s)
{
    if (n == 0) return CSRK_OK;
    if (n <= STAGE_DIRECT || n > STAGE_CAP) {
        CSRK_HIP(hipMemcpyAsync(host_dst, dev_src, n, hipMemcpyDeviceToHost, s));
        CSRK_HIP(hipStreamSynchronize(s));
        return CSRK_OK;
    }
    Stage *st = stage_of_current_device();
    std::lock_guard<std::mutex> lk(st->mu);
    CSRK_TRY(stage_room_locked(st, n));
    CSRK_TRY(stage_copy(dev_src, st->p, n, s));
    memcpy(host_dst, st->p, n);
    return CSRK_OK;
}

int stage_h2d(void *dev_dst, const void *host_src, size_t n, hipStream_t s)
{
    if (n == 0) return CSRK_OK;
    if (n <= STAGE_DIRECT || n > STAGE_CAP) {
        CSRK_HIP(hipMemcpyAsync(dev_dst, host_src, n, hipMemcpyHostToDevice, s));
        CSRK_HIP(hipStreamSynchronize(s));       // (the caller's buffer may go out of scope)
        return CSRK_OK;
    }
    Stage *st = stage_of_current_device();
    std::lock_guard<std::mutex> lk(st->mu);
    CSRK_TRY(stage_room_locked(st, n));
    memcpy(st->p, host_src, n);
    CSRK_TRY(stage_copy(st->p, dev_dst, n, s));      // (waits: the staging buffer is the next copy's)
    return CSRK_OK;
}

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

Matrix::~Matrix()
{
    if (spmv_plan) free_spmv_plan(spmv_plan);
    if (spmm_plan) free_spmm_plan(spmm_plan);
    if (owns) {
        pool_free(d_rowptrs);
        pool_free(d_colinds);
        pool_free(d_values);
    }
    magic = 0;
    std::lock_guard<std::mutex> lk(g_live_mu);
    live_set().erase(this);
}

Matrix *from_handle(csrk_handle_t h)
{
    Matrix *m = reinterpret_cast<Matrix *>(h);
    bool ok = false;
    if (m) {
        std::lock_guard<std::mutex> lk(g_live_mu);
        ok = live_set().count(m) != 0;
    }
    if (!ok || m->magic != 0x4353524b) {
        set_error("invalid csrk handle %p", (void *)h);
        return nullptr;
    }
    return m;
}

void drain_user_streams(Matrix *m)
{
    if (!m->used_user_stream) return;
    (void)hipDeviceSynchronize();
    m->used_user_stream = false;
}

void invalidate_plans(Matrix *m)
{
    m->dense_panel = -1;      // (order_columns may have made the rows ascending, or not)
    if (!m->spmv_plan && !m->spmm_plan) return;
    (void)hipDeviceSynchronize();
    if (m->spmm_plan) {                 // may hold a view into the SpMV plan: goes first
        free_spmm_plan(m->spmm_plan);
        m->spmm_plan = nullptr;
    }
    if (m->spmv_plan) {
        free_spmv_plan(m->spmv_plan);
        m->spmv_plan = nullptr;
    }
    m->spmv_calls = 0;
}

int new_matrix(int32_t nrows, int32_t ncols, int64_t nnz, int ptr64, int val_type, Matrix **out)
{
    Matrix *m = new (std::nothrow) Matrix();
    if (!m) {
        set_error("out of host memory");
        return CSRK_ERR_INVALID;
    }
    register_matrix(m);
    m->nrows = nrows;
    m->ncols = ncols;
    m->nnz = nnz;
    m->ptr64 = ptr64 ? 1 : 0;
    m->val_type = val_type;
    m->owns = true;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        set_error("hipGetDevice failed: %s", hipGetErrorString(e));
        delete m;
        return CSRK_ERR_HIP;
    }
    m->device = dev;
    size_t rp_bytes = (size_t)(nrows + 1) * m->ptr_bytes();
    size_t ci_bytes = (size_t)nnz * 4;
    size_t vs_bytes = (size_t)nnz * m->val_bytes();
    e = pool_alloc(&m->d_rowptrs, rp_bytes);
    if (e == hipSuccess) e = pool_alloc((void **)&m->d_colinds, ci_bytes ? ci_bytes : 16);
    if (e == hipSuccess && val_type != CSRK_VAL_NONE) e = pool_alloc(&m->d_values, vs_bytes ? vs_bytes : 16);
    if (e != hipSuccess) {
        set_error("hipMalloc failed for %lld-nnz matrix: %s", (long long)nnz, hipGetErrorString(e));
        delete m;
        return CSRK_ERR_HIP;
    }
    *out = m;
    return CSRK_OK;
}

template <class P>
__global__ void row_nnzs_kernel(const P *__restrict__ rp, P *__restrict__ out, int32_t nrows)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrows) out[i] = rp[i + 1] - rp[i];
}

}  // namespace csrk

using namespace csrk;

extern "C" {

int csrk_version(void) { return 1; }

const char *csrk_last_error(void) { return g_last_error.c_str(); }

int csrk_device_count(int *count)
{
    CSRK_REQUIRE(count, "count is NULL");
    *count = 0;
    CSRK_HIP(hipGetDeviceCount(count));
    return CSRK_OK;
}

int csrk_set_device(int device)
{
    CSRK_HIP(hipSetDevice(device));
    return CSRK_OK;
}

int csrk_trim_cache(void)
{
    CSRK_HIP(hipDeviceSynchronize());
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        pool_trim_locked(0);
    }
    stage_release_all();
    return CSRK_OK;
}

int csrk_partition_rows(int32_t nrows, const void *rowptrs, int ptr_is_64, int32_t parts, int32_t *bounds)
{
    CSRK_REQUIRE(nrows >= 0 && parts >= 1 && bounds && (rowptrs || nrows == 0), "bad arguments");
    auto rp = [&](int64_t i) -> int64_t { return ptr_is_64 ? ((const int64_t *)rowptrs)[i] : (int64_t)((const int32_t *)rowptrs)[i]; };
    const int64_t nnz = nrows ? rp(nrows) : 0;
    bounds[0] = 0;
    for (int32_t g = 1; g < parts; g++) {
        const int64_t target = nnz * g / parts;      // (nnz < 2^63 / parts for every matrix that fits a node)
        int64_t lo = 0, hi = nrows;                   // first i in [0, nrows] with rowptrs[i] >= target
        while (lo < hi) {
            const int64_t mid = lo + (hi - lo) / 2;
            if (rp(mid) < target) lo = mid + 1;
            else hi = mid;
        }
        bounds[g] = (int32_t)(lo > bounds[g - 1] ? lo : bounds[g - 1]);
    }
    bounds[parts] = nrows > bounds[parts - 1] ? nrows : bounds[parts - 1];
    return CSRK_OK;
}

int csrk_synchronize(void *stream)
{
    CSRK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return CSRK_OK;
}

static int check_dims(int32_t nrows, int32_t ncols, int64_t nnz, const void *rowptrs,
                      const void *colinds, const void *values, int val_type)
{
    CSRK_REQUIRE(nrows >= 0 && ncols >= 0 && nnz >= 0, "negative dimension (%d x %d, nnz %lld)",
                 nrows, ncols, (long long)nnz);
    CSRK_REQUIRE(rowptrs, "rowptrs is NULL");
    CSRK_REQUIRE(nnz == 0 || colinds, "colinds is NULL");
    CSRK_REQUIRE(val_type == CSRK_VAL_NONE || val_type == CSRK_VAL_F32 || val_type == CSRK_VAL_F64,
                 "unknown val_type %d", val_type);
    CSRK_REQUIRE(val_type == CSRK_VAL_NONE || nnz == 0 || values, "values is NULL but val_type=%d", val_type);
    return CSRK_OK;
}

int csrk_create(int32_t nrows, int32_t ncols, int64_t nnz, const void *rowptrs, int ptr_is_64,
                const int32_t *colinds, const void *values, int val_type, csrk_handle_t *out)
{
    CSRK_REQUIRE(out, "out is NULL");
    *out = 0;
    CSRK_TRY(check_dims(nrows, ncols, nnz, rowptrs, colinds, values, val_type));
    CSRK_REQUIRE(ptr_is_64 || nnz <= INT32_MAX, "nnz %lld needs 64-bit row pointers", (long long)nnz);
    // The host arrays are trusted like the reference trusts them, except for the one
    // invariant every kernel's bounds rely on.
    int64_t last = ptr_is_64 ? ((const int64_t *)rowptrs)[nrows] : (int64_t)((const int32_t *)rowptrs)[nrows];
    int64_t first = ptr_is_64 ? ((const int64_t *)rowptrs)[0] : (int64_t)((const int32_t *)rowptrs)[0];
    CSRK_REQUIRE(first == 0 && last == nnz, "rowptrs[0]=%lld, rowptrs[nrows]=%lld but nnz=%lld",
                 (long long)first, (long long)last, (long long)nnz);
    Matrix *m = nullptr;
    CSRK_TRY(new_matrix(nrows, ncols, nnz, ptr_is_64, val_type, &m));
    hipError_t e = hipMemcpy(m->d_rowptrs, rowptrs, (size_t)(nrows + 1) * m->ptr_bytes(), hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz) e = hipMemcpy(m->d_colinds, colinds, (size_t)nnz * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && nnz && val_type != CSRK_VAL_NONE)
        e = hipMemcpy(m->d_values, values, (size_t)nnz * m->val_bytes(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error("host-to-device copy failed: %s", hipGetErrorString(e));
        delete m;
        return CSRK_ERR_HIP;
    }
    *out = to_handle(m);
    return CSRK_OK;
}

int csrk_create_device(int32_t nrows, int32_t ncols, int64_t nnz, const void *d_rowptrs, int ptr_is_64,
                       const int32_t *d_colinds, const void *d_values, int val_type, csrk_handle_t *out)
{
    CSRK_REQUIRE(out, "out is NULL");
    *out = 0;
    CSRK_TRY(check_dims(nrows, ncols, nnz, d_rowptrs, d_colinds, d_values, val_type));
    CSRK_REQUIRE(ptr_is_64 || nnz <= INT32_MAX, "nnz %lld needs 64-bit row pointers", (long long)nnz);
    Matrix *m = new (std::nothrow) Matrix();
    CSRK_REQUIRE(m, "out of host memory");
    register_matrix(m);
    m->nrows = nrows;
    m->ncols = ncols;
    m->nnz = nnz;
    m->ptr64 = ptr_is_64 ? 1 : 0;
    m->val_type = val_type;
    m->owns = false;
    m->d_rowptrs = const_cast<void *>(d_rowptrs);
    m->d_colinds = const_cast<int32_t *>(d_colinds);
    m->d_values = const_cast<void *>(d_values);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        set_error("hipGetDevice failed: %s", hipGetErrorString(e));
        delete m;
        return CSRK_ERR_HIP;
    }
    m->device = dev;
    *out = to_handle(m);
    return CSRK_OK;
}

int csrk_free(csrk_handle_t h)
{
    if (h == 0) return CSRK_OK;
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    // launches on a caller's (possibly non-blocking) stream may still be reading the arrays and plans that
    // ~Matrix hands back to the caching allocator, whose next user runs in default-stream order
    {
        std::lock_guard<std::mutex> lk(m->mu);      // an operation in flight on another thread finishes its launch group first
        drain_user_streams(m);
    }
    delete m;
    return CSRK_OK;
}

int csrk_info(csrk_handle_t h, int32_t *nrows, int32_t *ncols, int64_t *nnz, int *ptr_is_64, int *val_type)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    if (nrows) *nrows = m->nrows;
    if (ncols) *ncols = m->ncols;
    if (nnz) *nnz = m->nnz;
    if (ptr_is_64) *ptr_is_64 = m->ptr64;
    if (val_type) *val_type = m->val_type;
    return CSRK_OK;
}

int csrk_device_bytes(csrk_handle_t h, int64_t *bytes)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE(bytes, "bytes is NULL");
    std::lock_guard<std::mutex> lk(m->mu);
    int64_t b = (int64_t)((size_t)(m->nrows + 1) * m->ptr_bytes() + (size_t)m->nnz * 4 + (size_t)m->nnz * m->val_bytes());
    if (m->spmv_plan) b += spmv_plan_bytes(m->spmv_plan);
    if (m->spmm_plan) b += spmm_plan_bytes(m->spmm_plan);
    *bytes = b;
    return CSRK_OK;
}

int csrk_export(csrk_handle_t h, void *rowptrs, int32_t *colinds, void *values)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    std::lock_guard<std::mutex> lk(m->mu);
    CSRK_HIP(hipDeviceSynchronize());
    if (rowptrs)
        CSRK_HIP(hipMemcpy(rowptrs, m->d_rowptrs, (size_t)(m->nrows + 1) * m->ptr_bytes(), hipMemcpyDeviceToHost));
    if (colinds && m->nnz)
        CSRK_HIP(hipMemcpy(colinds, m->d_colinds, (size_t)m->nnz * 4, hipMemcpyDeviceToHost));
    if (values && m->nnz && m->val_type != CSRK_VAL_NONE)
        CSRK_HIP(hipMemcpy(values, m->d_values, (size_t)m->nnz * m->val_bytes(), hipMemcpyDeviceToHost));
    return CSRK_OK;
}

int csrk_device_ptrs(csrk_handle_t h, void **d_rowptrs, void **d_colinds, void **d_values)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    if (d_rowptrs) *d_rowptrs = m->d_rowptrs;
    if (d_colinds) *d_colinds = m->d_colinds;
    if (d_values) *d_values = m->d_values;
    return CSRK_OK;
}

int csrk_row_nnzs(csrk_handle_t h, void *out)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE(out || m->nrows == 0, "out is NULL");
    if (m->nrows == 0) return CSRK_OK;
    std::lock_guard<std::mutex> lk(m->mu);
    DevBuf d;
    CSRK_TRY(d.alloc((size_t)m->nrows * m->ptr_bytes()));
    int grid = (int)ceil_div(m->nrows, 256);
    if (m->ptr64)
        row_nnzs_kernel<int64_t><<<grid, 256>>>((const int64_t *)m->d_rowptrs, d.as<int64_t>(), m->nrows);
    else
        row_nnzs_kernel<int32_t><<<grid, 256>>>((const int32_t *)m->d_rowptrs, d.as<int32_t>(), m->nrows);
    CSRK_LAUNCH_CHECK();
    CSRK_HIP(hipMemcpy(out, d.p, (size_t)m->nrows * m->ptr_bytes(), hipMemcpyDeviceToHost));
    return CSRK_OK;
}

int csrk_row_extent(csrk_handle_t h, int32_t row, int64_t *start, int64_t *end)
{
    Matrix *m = from_handle(h);
    if (!m) return CSRK_ERR_INVALID;
    CSRK_REQUIRE(row >= 0 && row < m->nrows, "row %d out of range [0, %d)", row, m->nrows);
    CSRK_REQUIRE(start && end, "start/end is NULL");
    if (m->ptr64) {
        int64_t se[2];
        CSRK_HIP(hipMemcpy(se, (const int64_t *)m->d_rowptrs + row, 16, hipMemcpyDeviceToHost));
        *start = se[0];
        *end = se[1];
    } else {
        int32_t se[2];
        CSRK_HIP(hipMemcpy(se, (const int32_t *)m->d_rowptrs + row, 8, hipMemcpyDeviceToHost));
        *start = se[0];
        *end = se[1];
    }
    return CSRK_OK;
}

}  // extern "C"
