// knn_host.cpp -- see knn_host.h.  Host-only translation unit (g++, no device code): a persistent pool of worker threads, the
// storage-type conversions they run, and the double-buffered pinned staging between the caller's pageable array and HBM.
#include "knn_host.h"
#include "../../include/trx_knn.h"

#include <atomic>
#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include <pthread.h>
#include <sched.h>

namespace trx {

// ---- worker pool ---------------------------------------------------------------------------------------------------------
// run(ntasks, f): f(task) for every task in [0, ntasks), spread over the workers AND the calling thread; returns when all are
// done.  The workers sleep on a condition variable between calls.  A fork()ed child (multiprocessing) inherits no threads:
// pthread_atfork drops the pool there and the next call makes a new one.
namespace {
class Pool {
public:
    explicit Pool(int nworkers) {
        for (int i = 0; i < nworkers; ++i) th_.emplace_back([this] { loop(); });
    }
    int workers() const { return (int)th_.size(); }
    void run(int ntasks, const std::function<void(int)>& f) {
        if (ntasks <= 0) return;
        std::lock_guard<std::mutex> one_caller(run_mu_);      // callers on different threads (two indexes, trx_host_convert) take turns
        {
            std::unique_lock<std::mutex> g(mu_);
            cv_done_.wait(g, [this] { return active_ == 0; });      // no straggler of the previous call is still looking at the counter
            fn_ = &f; ntasks_ = ntasks; next_.store(0, std::memory_order_relaxed); left_ = ntasks; ++gen_;
        }
        cv_go_.notify_all();
        drain();
        std::unique_lock<std::mutex> g(mu_);
        cv_done_.wait(g, [this] { return left_ == 0; });
    }
private:
    // (only ever entered with the call's state published under mu_ and unchanged until every worker has left again)
    void drain() {
        for (;;) {
            const int t = next_.fetch_add(1, std::memory_order_relaxed);
            if (t >= ntasks_) return;
            (*fn_)(t);
            std::lock_guard<std::mutex> g(mu_);
            if (--left_ == 0) cv_done_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_go_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                ++active_;
            }
            drain();
            std::lock_guard<std::mutex> g(mu_);
            if (--active_ == 0) cv_done_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_go_, cv_done_;
    const std::function<void(int)>* fn_ = nullptr;
    int ntasks_ = 0, left_ = 0, active_ = 0;
    std::atomic<int> next_{0};
    uint64_t gen_ = 0;
};

std::mutex g_pool_mu;
Pool* g_pool = nullptr;       // (never destroyed: its threads run until the process ends)
int g_threads = 0;

int affinity_cores() {
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) return c; }
    const unsigned h = std::thread::hardware_concurrency();
    return h ? (int)h : 1;
}

Pool& pool() {
    std::lock_guard<std::mutex> g(g_pool_mu);
    if (!g_pool) {
        static bool hooked = false;
        if (!hooked) { pthread_atfork(nullptr, nullptr, [] { g_pool = nullptr; new (&g_pool_mu) std::mutex(); }); hooked = true; }
        int n = std::min(affinity_cores(), 32);
        if (const char* e = getenv("TRX_HOST_THREADS")) n = std::max(1, std::min(1024, atoi(e)));
        g_threads = n;
        g_pool = new Pool(n - 1);      // the caller is the n-th
    }
    return *g_pool;
}
}  // namespace

int host_threads() { (void)pool(); return g_threads; }

// ---- storage-type conversions ----------------------------------------------------------------------------------------------
// Element ranges, contiguous on both sides.  The narrowing ones return nonzero when some value does not fit a signed byte.
#define TRX_CLONES __attribute__((target_clones("arch=skylake-avx512", "avx2", "default")))

template <class T> static inline uint64_t narrow_range(const T* s, int64_t n, int8_t* d) {
    uint64_t ov = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t v = (int64_t)s[i];
        ov |= ((uint64_t)v + 128u) >> 8;     // 0 iff -128 <= v <= 127 (unsigned: no overflow at the ends of int64)
        d[i] = (int8_t)v;
    }
    return ov;
}
TRX_CLONES static uint64_t narrow_i64(const int64_t* s, int64_t n, int8_t* d) { return narrow_range(s, n, d); }
TRX_CLONES static uint64_t narrow_i32(const int32_t* s, int64_t n, int8_t* d) { return narrow_range(s, n, d); }
TRX_CLONES static uint64_t narrow_i16(const int16_t* s, int64_t n, int8_t* d) { return narrow_range(s, n, d); }
TRX_CLONES static uint64_t narrow_u8(const uint8_t* s, int64_t n, int8_t* d) { return narrow_range(s, n, d); }

// the float32 FAISS' wrapper would have made of the array (numpy astype: round to nearest even)
template <class T> static inline void to_f32_range(const T* s, int64_t n, float* d) { for (int64_t i = 0; i < n; ++i) d[i] = (float)s[i]; }
TRX_CLONES static void f32_of_i64(const int64_t* s, int64_t n, float* d) { to_f32_range(s, n, d); }
TRX_CLONES static void f32_of_i32(const int32_t* s, int64_t n, float* d) { to_f32_range(s, n, d); }
TRX_CLONES static void f32_of_i16(const int16_t* s, int64_t n, float* d) { to_f32_range(s, n, d); }
TRX_CLONES static void f32_of_u8(const uint8_t* s, int64_t n, float* d) { to_f32_range(s, n, d); }
TRX_CLONES static void f32_of_f64(const double* s, int64_t n, float* d) { to_f32_range(s, n, d); }

int host_dtype_size(int dtype) {
    switch (dtype) {
    case TRX_DTYPE_F32: return 4; case TRX_DTYPE_BF16: return 2; case TRX_DTYPE_I8: return 1;
    case TRX_DTYPE_I64: return 8; case TRX_DTYPE_I32: return 4; case TRX_DTYPE_I16: return 2; case TRX_DTYPE_U8: return 1;
    case TRX_DTYPE_F64: return 8;
    default: return 0;
    }
}
bool host_dtype_is_wide_int(int dtype) {
    return dtype == TRX_DTYPE_I64 || dtype == TRX_DTYPE_I32 || dtype == TRX_DTYPE_I16 || dtype == TRX_DTYPE_U8;
}
static int staged_size(int staged) { return staged == STAGED_F32 ? 4 : staged == STAGED_BF16 ? 2 : 1; }
size_t staged_bytes(int64_t m, int d, int staged) { return (size_t)m * (size_t)d * staged_size(staged); }

// elements [e0, e0 + n) of the caller's array -> the staged form; nonzero = a value that the narrow form cannot hold
static uint64_t convert_range(const void* src, int dtype, int64_t e0, int64_t n, void* dst, int staged) {
    const char* s = (const char*)src + (size_t)e0 * host_dtype_size(dtype);
    if (staged == STAGED_I8) {
        switch (dtype) {
        case TRX_DTYPE_I64: return narrow_i64((const int64_t*)s, n, (int8_t*)dst);
        case TRX_DTYPE_I32: return narrow_i32((const int32_t*)s, n, (int8_t*)dst);
        case TRX_DTYPE_I16: return narrow_i16((const int16_t*)s, n, (int8_t*)dst);
        case TRX_DTYPE_U8: return narrow_u8((const uint8_t*)s, n, (int8_t*)dst);
        default: std::memcpy(dst, s, (size_t)n); return 0;                          // I8 as it is
        }
    }
    if (staged == STAGED_F32) {
        switch (dtype) {
        case TRX_DTYPE_I64: f32_of_i64((const int64_t*)s, n, (float*)dst); return 0;
        case TRX_DTYPE_I32: f32_of_i32((const int32_t*)s, n, (float*)dst); return 0;
        case TRX_DTYPE_I16: f32_of_i16((const int16_t*)s, n, (float*)dst); return 0;
        case TRX_DTYPE_U8: f32_of_u8((const uint8_t*)s, n, (float*)dst); return 0;
        case TRX_DTYPE_F64: f32_of_f64((const double*)s, n, (float*)dst); return 0;
        default: std::memcpy(dst, s, (size_t)n * 4); return 0;                      // F32 as it is
        }
    }
    std::memcpy(dst, s, (size_t)n * 2);                                               // BF16 as it is
    return 0;
}

// ---- pinned staging ------------------------------------------------------------------------------------------------------
namespace {
constexpr size_t CHUNK_BYTES = (size_t)16 << 20;      // staged bytes per chunk: 16 MiB crosses PCIe in ~0.3 ms
struct Stager {
    std::mutex mu;
    void* pin[2] = {nullptr, nullptr};
    hipError_t ensure() {
        for (int i = 0; i < 2; ++i)
            if (!pin[i]) { hipError_t e = hipHostMalloc(&pin[i], CHUNK_BYTES, hipHostMallocPortable); if (e != hipSuccess) { pin[i] = nullptr; return e; } }
        return hipSuccess;
    }
};
Stager g_stager;

struct Events {
    hipEvent_t ev[2] = {nullptr, nullptr}; bool pending[2] = {false, false};
    hipError_t init() { for (auto& e : ev) { hipError_t r = hipEventCreateWithFlags(&e, hipEventDisableTiming); if (r != hipSuccess) return r; } return hipSuccess; }
    ~Events() { for (auto& e : ev) if (e) (void)hipEventDestroy(e); }
};

// run `conv(e0, n, dst)` over elements [e0, e0 + n) with the pool; the tasks are contiguous slices (a multiple of 4096 elements)
template <class F> uint64_t parallel_convert(int64_t count, size_t out_esz, char* out, F conv) {
    Pool& p = pool();
    const int want = std::max(1, (p.workers() + 1) * 2);
    int64_t per = (count + want - 1) / want;
    per = std::max<int64_t>(4096, (per + 4095) / 4096 * 4096);
    const int ntasks = (int)((count + per - 1) / per);
    std::atomic<uint64_t> ov{0};
    std::function<void(int)> f = [&](int t) {
        const int64_t e0 = (int64_t)t * per, n = std::min(per, count - e0);
        const uint64_t o = conv(e0, n, out + (size_t)e0 * out_esz);
        if (o) ov.fetch_or(o, std::memory_order_relaxed);
    };
    p.run(ntasks, f);
    return ov.load();
}
}  // namespace

hipError_t stage_host_rows(const void* src, int dtype, int64_t m, int d, void* dev, int* staged, hipStream_t copy_stream) {
    if (m <= 0) return hipSuccess;
    int form;
    if (dtype == TRX_DTYPE_F32 || dtype == TRX_DTYPE_F64) form = STAGED_F32;
    else if (dtype == TRX_DTYPE_BF16) form = STAGED_BF16;
    else if (dtype == TRX_DTYPE_I8) form = STAGED_I8;
    else form = *staged == STAGED_I8 ? STAGED_I8 : STAGED_F32;
    std::lock_guard<std::mutex> g(g_stager.mu);
    hipError_t e = g_stager.ensure(); if (e != hipSuccess) return e;
    Events evs; if ((e = evs.init()) != hipSuccess) return e;
    const int64_t total = m * (int64_t)d;
    for (;;) {
        const size_t oesz = (size_t)staged_size(form);
        const int64_t per_chunk = (int64_t)(CHUNK_BYTES / oesz);
        bool redo = false;
        int c = 0;
        for (int64_t e0 = 0; e0 < total; e0 += per_chunk, ++c) {
            const int b = c & 1;
            const int64_t n = std::min(per_chunk, total - e0);
            if (evs.pending[b]) { if ((e = hipEventSynchronize(evs.ev[b])) != hipSuccess) return e; evs.pending[b] = false; }
            const uint64_t ov = parallel_convert(n, oesz, (char*)g_stager.pin[b],
                                                 [&](int64_t o, int64_t cnt, void* dst) { return convert_range(src, dtype, e0 + o, cnt, dst, form); });
            if (ov) { redo = true; break; }      // an integer beyond a signed byte: the block goes as the float32 FAISS would have seen
            if ((e = hipMemcpyAsync((char*)dev + (size_t)e0 * oesz, g_stager.pin[b], (size_t)n * oesz, hipMemcpyHostToDevice, copy_stream)) != hipSuccess) return e;
            if ((e = hipEventRecord(evs.ev[b], copy_stream)) != hipSuccess) return e;
            evs.pending[b] = true;
        }
        if ((e = hipStreamSynchronize(copy_stream)) != hipSuccess) return e;      // the pinned buffers are free again; the rows are in HBM
        evs.pending[0] = evs.pending[1] = false;
        if (!redo) break;
        form = STAGED_F32;
    }
    *staged = form;
    return hipSuccess;
}

int convert_host_rows(const void* src, int dtype, int64_t count, void* dst, int staged) {
    const uint64_t ov = parallel_convert(count, (size_t)staged_size(staged), (char*)dst,
                                         [&](int64_t o, int64_t cnt, void* d2) { return convert_range(src, dtype, o, cnt, d2, staged); });
    return ov ? 1 : 0;
}

hipError_t unstage_to_host(void* dst, const void* dev, size_t bytes, hipStream_t copy_stream) {
    if (!bytes) return hipSuccess;
    std::lock_guard<std::mutex> g(g_stager.mu);
    hipError_t e = g_stager.ensure(); if (e != hipSuccess) return e;
    // device -> pinned chunk c + 1 travels while the workers copy chunk c out of its pinned buffer
    const size_t nchunks = (bytes + CHUNK_BYTES - 1) / CHUNK_BYTES;
    auto len = [&](size_t c) { return std::min(CHUNK_BYTES, bytes - c * CHUNK_BYTES); };
    if ((e = hipMemcpyAsync(g_stager.pin[0], dev, len(0), hipMemcpyDeviceToHost, copy_stream)) != hipSuccess) return e;
    for (size_t c = 0; c < nchunks; ++c) {
        if ((e = hipStreamSynchronize(copy_stream)) != hipSuccess) return e;
        if (c + 1 < nchunks &&
            (e = hipMemcpyAsync(g_stager.pin[(c + 1) & 1], (const char*)dev + (c + 1) * CHUNK_BYTES, len(c + 1), hipMemcpyDeviceToHost, copy_stream)) != hipSuccess)
            return e;
        const char* s = (const char*)g_stager.pin[c & 1];
        char* o = (char*)dst + c * CHUNK_BYTES;
        const size_t n = len(c);
        if (n < ((size_t)1 << 20)) std::memcpy(o, s, n);
        else parallel_convert((int64_t)n, 1, o, [&](int64_t e0, int64_t cnt, void* d2) { std::memcpy(d2, s + e0, (size_t)cnt); return (uint64_t)0; });
    }
    return hipSuccess;
}

}  // namespace trx
