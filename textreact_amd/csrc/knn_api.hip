// knn_api.hip -- the C ABI of include/trx_knn.h over the gfx950 kernels.  Host-side orchestration
// only: storage of the flat index in HBM, operand layout decisions, launch geometry, the
// certificate-driven fall-back.  No CPU compute path exists: without a HIP device every entry
// point that would compute returns TRX_ENODEV / TRX_EHIP.
#include "../../include/trx_knn.h"
#include "knn_common.h"
#include "knn_host.h"

#include <algorithm>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

namespace trx {
hipError_t launch_row_stats(const void*, int, int64_t, int, int64_t, void*, float*, float*, hipStream_t);
hipError_t launch_build_operand(const void*, int, int, int64_t, int, int64_t, bf16_t*, int, hipStream_t);
hipError_t launch_fill_bias(const float*, int64_t, int64_t, float*, hipStream_t);
hipError_t launch_widen_rows(const bf16_t*, int64_t, int, int64_t, float*, hipStream_t);
hipError_t launch_widen_i8(const signed char*, int64_t, int, bf16_t*, hipStream_t);
hipError_t launch_exact_scan(int, int, int, const int*, int, const int*, int64_t, const void*, int64_t, const void*,
                             int64_t, int, int, double*, float*, int64_t*, double*, hipStream_t);
hipError_t launch_classify(const void*, int, int, float, int, int, int, int, int*, hipStream_t);
hipError_t launch_build_operand_fp4(const void*, int, int64_t, int, int64_t, unsigned char*, int, hipStream_t);
hipError_t launch_build_operand_i8(const void*, int, int64_t, int, int64_t, signed char*, int, float, hipStream_t);
hipError_t launch_fill_bias_i32(const float*, int64_t, int64_t, int*, hipStream_t);
hipError_t launch_slack(const float*, int64_t, int64_t, float, float, float, const float*, float, int, float*, hipStream_t);
hipError_t launch_bigk_seeds(int, const float*, const int64_t*, int, int, int, const float*, float, float, float, const float*, float, int*, int*, float*, hipStream_t);
hipError_t launch_merge(int, int, int64_t, int, const double*, const int64_t*, float*, int64_t*, double*, hipStream_t);
hipError_t launch_faiss_ties(int64_t, int, int, const double*, const int64_t*, float*, int64_t*, double*, hipStream_t);
hipError_t launch_sanitize_hostile(float*, int64_t, int64_t, bf16_t*, int, int*, int*, hipStream_t);
}  // namespace trx

using namespace trx;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess)                                                             \
            return fail(e__ == hipErrorOutOfMemory ? TRX_ENOMEM : TRX_EHIP,                \
                        std::string(#expr) + ": " + hipGetErrorString(e__));               \
    } while (0)

struct HostStats { uint32_t inexact_any, nonint_any, maxabs_bits, maxnorm2_bits, nonfp4_any, maxerr2_bits, hostile_any; };
static float bits2f(uint32_t b) { float f; std::memcpy(&f, &b, 4); return f; }

// grow-only device buffer
struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    int reserve(size_t need) {
        if (need <= bytes) return TRX_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        size_t want = need + need / 8;
        HIPCHK(hipMalloc(&p, want));
        bytes = want;
        return TRX_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

// The large search workspaces -- candidate lists (2 GiB for a 65,536-query batch), their counts and bounds, the shared
// thresholds, the padded query operand -- are dead as soon as the select kernel of a batch has run, and that kernel is
// enqueued inside trx_index_search_device_begin.  So every index of a process on one device shares ONE set of them (a
// process with 8 indexes used to hold 8 x 2.3 GiB): a search enqueues under the pool's mutex, first makes its stream wait
// for the event the previous user recorded behind its last kernel, and records its own at the end.  Searches of different
// indexes therefore serialise on the GPU where they touch the pool -- each of them fills the device anyway.  What a
// search still reads after begin() has returned (certificate counts, flagged queries, the exact fall-back's rows) stays
// per index.
struct DevPool {
    std::mutex mu;
    DevBuf cand, cnt, thr, gthr, qg, qg2, gthr2, qg8, qg4, slk, d1, i1;
    hipEvent_t last = nullptr;
    int users = 0;
};
static DevPool g_pool[64];
static DevPool& pool_of(int device) { return g_pool[device & 63]; }

// PLAIN: every value is exact in bf16, the operand IS the data.  APPROX (round 4): the operand is the bf16 ROUNDING of fp32 data
// (K = d), the fp32 rows are kept for the exact pass, and the key error this admits (2^-7 |x||y|) is paid for by listing every row
// within twice that error of a query's bound and certifying on the exact scores (search_batch: slack).  SPLIT (rounds 1-3; since
// round 6 reachable in the LAB build only, -DTRX_KNN_LAB + TRX_FP32_SPLIT=1): a three-term bf16 split of the fp32 data, K = 3d, keys good to 2^-16.
enum { MODE_EMPTY = 0, MODE_PLAIN = 1, MODE_SPLIT = 2, MODE_APPROX = 3 };
#ifdef TRX_KNN_LAB      // lab build (make knnlab; tools/experiments/lab_checks_knn.py): TRX_FP32_SPLIT=1 selects the three-term split
static int inexact_mode() { return getenv("TRX_FP32_SPLIT") ? MODE_SPLIT : MODE_APPROX; }      // (read when an index first meets such data)
#else
static int inexact_mode() { return MODE_APPROX; }
#endif
static bool keeps_f32(int mode) { return mode == MODE_SPLIT || mode == MODE_APPROX; }
static int round_up(int64_t v, int m) { return (int)((v + m - 1) / m * m); }
static int64_t round_up64(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

struct trx_index {
    int d = 0, metric = 0, device = 0;
    int64_t n = 0, cap = 0;  // rows stored, row capacity (multiple of TILE_M)
    int mode = MODE_EMPTY;
    int Kp = 0;              // row stride of Cg
    bf16_t* Cg = nullptr;    // GEMM operand [cap][Kp]
    float* Co = nullptr;     // exact values as f32 [cap][d], SPLIT mode only
    float* cnorm2 = nullptr; // [cap]
    float* cbias = nullptr;  // [cap]
    float maxabs = 0.f, maxnorm2 = 0.f;
    float maxerr2 = 0.f;      // max over the rows of |y - bf16(y)|^2: the corpus side of the measured rounding bound (knn_common.h: round_term)
    bool nonint = false;
    bool nonfp4 = false;      // some value is not one of 0, +-1, +-2, +-3, +-4, +-6 (what the fp4 form of the scan can hold)
    // the int8 form of the scan for the integer class (knn_scan.hip, I8): an int8 copy of the operand rows and the int32 start
    // values of the L2 accumulators, made by the first search that can use them after the index changed
    signed char* C8 = nullptr; int* cbias8 = nullptr;
    int64_t c8_cap = 0, c8_rows = -1; int Kp8 = 0;
    unsigned char* C4 = nullptr;      // the fp4 copy of a bit-vector corpus (two components per byte)
    int64_t c4_cap = 0, c4_rows = -1; int Kp4 = 0;
    // workspaces
    DevBuf w_stamp, w_stats, w_qnorm2, w_flag, w_exact, w_io, w_tmp, w_cls;      // (the big ones are shared: DevPool)
    DevBuf w_stage[2], w_wide; // host entry points: a block of the caller's rows as staged (knn_host.h), and int8 rows widened to bf16
    int cus = 0;               // compute units of the device (choose_splits), read once
    int64_t reserve_rows = 0;  // trx_index_add: the rows the index will hold when the call is through (one allocation, not one per block)
    // hostile rows (knn_common.h): their ids, sorted, [MAX_SPECIAL] ints + the device-side counter behind them; more than MAX_SPECIAL
    // of them and every search of this index is the exact fp64 scan
    DevBuf w_special; int n_special = 0; bool special_overflow = false;
    DevBuf w_tie;             // TRX_TIES_FAISS: the canonical top 2k (D, I, S) the FAISS order is derived from
    int tie_rule = TRX_TIES_BY_ID;
    trx_search_stats stats{};
    bool timing = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t copy_stream = nullptr;      // host entry points: the next block's host-to-device copy under this block's search
    // a search that has been enqueued (trx_index_search_device_begin) and not yet finished: what
    // trx_index_search_finish needs to read the certificate counts back and, rarely, complete the fall-back
    struct PendingBatch { int* nflag; int* flagged; int final_slot; const void* q; float* D; int64_t* I; double* S64; };
    struct Pending {
        bool active = false;
        hipStream_t st = nullptr;
        int is_bf = 0, k = 0, corpus_is_bf16 = 0, no_fallback = 0, tried_i8 = 0, approx = 0; float eps_round = 0.f;
        const void* corpus_orig = nullptr; int64_t ld_c = 0;
        std::vector<PendingBatch> batches;
        // TRX_TIES_FAISS: the search ran for k2 = 2k into w_tie; these are the caller's arrays the tie kernel fills
        int tie_k = 0, tie_k2 = 0; int64_t tie_nq = 0; float* tie_D = nullptr; int64_t* tie_I = nullptr; double* tie_S = nullptr;
    } pend;

    // an even number of K-steps, at least 4: the scan kernel's loop handles two per iteration (LDS stage = K-step
    // parity) and treats the first and the last pair of a tile differently
    int Kp_for(int mode_) const { return std::max(4 * BK, round_up(mode_ == MODE_SPLIT ? 3 * (int64_t)d : d, 2 * BK)); }
};

static int set_device(const trx_index* idx) { HIPCHK(hipSetDevice(idx->device)); return TRX_OK; }

static int read_stats(trx_index* idx, hipStream_t st, HostStats* out) {
    HIPCHK(hipMemcpyAsync(out, idx->w_stats.p, sizeof(HostStats), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return TRX_OK;
}

// (re)allocate corpus arrays for `newcap` rows in `newmode`, carrying existing rows over.
static int restructure(trx_index* idx, int64_t newcap, int newmode, hipStream_t st) {
    const int newKp = idx->Kp_for(newmode);
    bf16_t* nCg = nullptr; float* nCo = nullptr; float* nn2 = nullptr; float* nb = nullptr;
    // an error return below (HIPCHK) must not leak the arrays allocated so far; `commit` disarms the guard
    struct Guard {
        bf16_t*& a; float*& b; float*& c; float*& d; bool commit = false;
        ~Guard() { if (!commit) { if (a) (void)hipFree(a); if (b) (void)hipFree(b); if (c) (void)hipFree(c); if (d) (void)hipFree(d); } }
    } guard{nCg, nCo, nn2, nb};
    // one spare tile behind the last row: the scan's DMA cursors run up to two K-steps past the end
    const size_t rows_alloc = (size_t)newcap + TILE_M;
    HIPCHK(hipMalloc((void**)&nCg, rows_alloc * newKp * sizeof(bf16_t)));
    HIPCHK(hipMemsetAsync(nCg, 0, rows_alloc * newKp * sizeof(bf16_t), st));
    HIPCHK(hipMalloc((void**)&nn2, (size_t)newcap * sizeof(float)));
    HIPCHK(hipMemsetAsync(nn2, 0, (size_t)newcap * sizeof(float), st));
    HIPCHK(hipMalloc((void**)&nb, rows_alloc * sizeof(float)));
    if (keeps_f32(newmode)) {
        HIPCHK(hipMalloc((void**)&nCo, (size_t)newcap * idx->d * sizeof(float)));
    }
    if (idx->n > 0) {
        if (idx->mode == newmode) {
            HIPCHK(hipMemcpyAsync(nCg, idx->Cg, (size_t)idx->n * newKp * sizeof(bf16_t),
                                  hipMemcpyDeviceToDevice, st));
            if (keeps_f32(newmode))
                HIPCHK(hipMemcpyAsync(nCo, idx->Co, (size_t)idx->n * idx->d * sizeof(float),
                                      hipMemcpyDeviceToDevice, st));
        } else {
            // PLAIN -> SPLIT / APPROX: exact values are the bf16 ones; widen, then lay out [hi|lo=0|hi] / the rows as they are
            HIPCHK(launch_widen_rows(idx->Cg, idx->n, idx->d, idx->Kp, nCo, st));
            HIPCHK(launch_build_operand(nCo, 0, newmode == MODE_SPLIT ? 1 : 0, idx->n, idx->d, idx->d, nCg, newKp, st));
        }
        HIPCHK(hipMemcpyAsync(nn2, idx->cnorm2, (size_t)idx->n * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    if (idx->Cg) (void)hipFree(idx->Cg);
    if (idx->Co) (void)hipFree(idx->Co);
    if (idx->cnorm2) (void)hipFree(idx->cnorm2);
    if (idx->cbias) (void)hipFree(idx->cbias);
    idx->Cg = nCg; idx->Co = nCo; idx->cnorm2 = nn2; idx->cbias = nb;
    idx->cap = newcap; idx->mode = newmode; idx->Kp = newKp;
    guard.commit = true;
    return TRX_OK;
}

extern "C" {

const char* trx_last_error(void) { return g_err.c_str(); }
const char* trx_version(void) { return "trxknn 0.2 (gfx950)"; }
int trx_search_stats_size(void) { return (int)sizeof(trx_search_stats); }

int trx_index_create(int d, int metric, int device, trx_index** out) {
    if (!out) return fail(TRX_EINVAL, "out is null");
    *out = nullptr;
    if (d <= 0 || d > (1 << 20)) return fail(TRX_EINVAL, "d must be in [1, 2^20]");
    if (metric != TRX_METRIC_IP && metric != TRX_METRIC_L2) return fail(TRX_EINVAL, "unknown metric");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(TRX_ENODEV, "no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(TRX_EINVAL, "device ordinal out of range");
    trx_index* idx = new (std::nothrow) trx_index();
    if (!idx) return fail(TRX_ENOMEM, "host allocation failed");
    idx->d = d; idx->metric = metric; idx->device = device;
    { DevPool& pl = pool_of(device); std::lock_guard<std::mutex> g(pl.mu); pl.users++; }
    *out = idx;
    return TRX_OK;
}

void trx_index_destroy(trx_index* idx) {
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    if (idx->Cg) (void)hipFree(idx->Cg);
    if (idx->Co) (void)hipFree(idx->Co);
    if (idx->cnorm2) (void)hipFree(idx->cnorm2);
    if (idx->cbias) (void)hipFree(idx->cbias);
    if (idx->C8) (void)hipFree(idx->C8);
    if (idx->cbias8) (void)hipFree(idx->cbias8);
    if (idx->C4) (void)hipFree(idx->C4);
    DevBuf* bufs[] = {&idx->w_stamp, &idx->w_stats, &idx->w_qnorm2, &idx->w_flag, &idx->w_exact, &idx->w_io, &idx->w_tmp, &idx->w_cls, &idx->w_tie,
                      &idx->w_stage[0], &idx->w_stage[1], &idx->w_wide, &idx->w_special};
    for (DevBuf* b : bufs) b->release();
    {   // the last index of the process on this device takes the shared workspaces with it
        DevPool& pl = pool_of(idx->device);
        std::lock_guard<std::mutex> g(pl.mu);
        if (--pl.users <= 0) {
            pl.users = 0;
            (void)hipDeviceSynchronize();
            DevBuf* shared[] = {&pl.cand, &pl.cnt, &pl.thr, &pl.gthr, &pl.qg, &pl.qg2, &pl.gthr2, &pl.qg8, &pl.qg4, &pl.slk, &pl.d1, &pl.i1};
            for (DevBuf* b : shared) b->release();
            // (the ordering event stays for the life of the process: a few bytes, and nothing can hold a stale handle to it)
        }
    }
    for (auto& e : idx->ev) if (e) (void)hipEventDestroy(e);
    if (idx->copy_stream) (void)hipStreamDestroy(idx->copy_stream);
    delete idx;
}

int64_t trx_index_ntotal(const trx_index* idx) { return idx ? idx->n : -1; }
int trx_index_dim(const trx_index* idx) { return idx ? idx->d : -1; }

int trx_index_reset(trx_index* idx) {
    if (!idx) return fail(TRX_EINVAL, "index is null");
    int rc = set_device(idx); if (rc) return rc;
    if (idx->Cg) (void)hipFree(idx->Cg);
    if (idx->Co) (void)hipFree(idx->Co);
    if (idx->cnorm2) (void)hipFree(idx->cnorm2);
    if (idx->cbias) (void)hipFree(idx->cbias);
    idx->Cg = nullptr; idx->Co = nullptr; idx->cnorm2 = nullptr; idx->cbias = nullptr;
    if (idx->C8) (void)hipFree(idx->C8);
    if (idx->cbias8) (void)hipFree(idx->cbias8);
    idx->C8 = nullptr; idx->cbias8 = nullptr; idx->c8_cap = 0; idx->c8_rows = -1;
    if (idx->C4) (void)hipFree(idx->C4);
    idx->C4 = nullptr; idx->c4_cap = 0; idx->c4_rows = -1; idx->nonfp4 = false;
    idx->n = 0; idx->cap = 0; idx->mode = MODE_EMPTY; idx->Kp = 0;
    idx->maxabs = 0.f; idx->maxnorm2 = 0.f; idx->maxerr2 = 0.f; idx->nonint = false;
    idx->n_special = 0; idx->special_overflow = false;
    return TRX_OK;
}

int trx_index_set_timing(trx_index* idx, int enabled) {
    if (!idx) return fail(TRX_EINVAL, "index is null");
    int rc = set_device(idx); if (rc) return rc;
    idx->timing = enabled != 0;
    if (idx->timing)
        for (auto& e : idx->ev) if (!e) HIPCHK(hipEventCreate(&e));
    return TRX_OK;
}

int trx_index_last_stats(const trx_index* idx, trx_search_stats* out) {
    if (!idx || !out) return fail(TRX_EINVAL, "null argument");
    *out = idx->stats;
    return TRX_OK;
}

int trx_index_add_device(trx_index* idx, const void* x, int64_t n, int dtype, void* stream) {
    if (!idx) return fail(TRX_EINVAL, "index is null");
    if (n < 0 || (n > 0 && !x)) return fail(TRX_EINVAL, "bad vector block");
    if (dtype != TRX_DTYPE_F32 && dtype != TRX_DTYPE_BF16) return fail(TRX_EINVAL, "unknown dtype");
    if (n == 0) return TRX_OK;
    if (idx->n + n >= (int64_t)0x7fffff00) return fail(TRX_EINVAL, "more than 2^31 rows per index");
    int rc = set_device(idx); if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int is_bf = dtype == TRX_DTYPE_BF16;

    // classify the block
    rc = idx->w_stats.reserve(sizeof(HostStats)); if (rc) return rc;
    rc = idx->w_tmp.reserve((size_t)n * sizeof(float)); if (rc) return rc;
    HIPCHK(hipMemsetAsync(idx->w_stats.p, 0, sizeof(HostStats), st));
    HIPCHK(launch_row_stats(x, is_bf, n, idx->d, idx->d, idx->w_stats.p, (float*)idx->w_tmp.p, nullptr, st));
    HostStats hs; rc = read_stats(idx, st, &hs); if (rc) return rc;

    int newmode = idx->mode;
    // a hostile row (knn_common.h) needs the f32 copy: its operand row is zeroed below, its values must survive for the exact pass
    const bool inexact = hs.inexact_any || hs.hostile_any;
    if (idx->mode == MODE_EMPTY) newmode = inexact ? inexact_mode() : MODE_PLAIN;
    else if (idx->mode == MODE_PLAIN && inexact) newmode = inexact_mode();
    int64_t need = idx->n + n, newcap = idx->cap;
    if (need > newcap) newcap = round_up64(std::max<int64_t>(std::max(need, idx->reserve_rows), idx->cap + idx->cap / 2), TILE_M);
    if (newcap != idx->cap || newmode != idx->mode) { rc = restructure(idx, newcap, newmode, st); if (rc) return rc; }

    // append
    bf16_t* dstg = idx->Cg + idx->n * idx->Kp;
    if (keeps_f32(idx->mode)) {
        HIPCHK(launch_build_operand(x, is_bf, idx->mode == MODE_SPLIT ? 1 : 0, n, idx->d, idx->d, dstg, idx->Kp, st));
        float* dsto = idx->Co + idx->n * idx->d;
        if (is_bf) HIPCHK(launch_widen_rows((const bf16_t*)x, n, idx->d, idx->d, dsto, st));
        else HIPCHK(hipMemcpyAsync(dsto, x, (size_t)n * idx->d * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else {
        HIPCHK(launch_build_operand(x, is_bf, 0, n, idx->d, idx->d, dstg, idx->Kp, st));
    }
    if (hs.hostile_any) {
        rc = idx->w_special.reserve((MAX_SPECIAL + 1) * sizeof(int)); if (rc) return rc;
        int* spec = (int*)idx->w_special.p; int* cnt = spec + MAX_SPECIAL;
        HIPCHK(hipMemcpyAsync(cnt, &idx->n_special, sizeof(int), hipMemcpyHostToDevice, st));
        HIPCHK(launch_sanitize_hostile((float*)idx->w_tmp.p, n, idx->n, dstg, idx->Kp, spec, cnt, st));
        int total = 0;
        HIPCHK(hipMemcpyAsync(&total, cnt, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (total > MAX_SPECIAL) { idx->special_overflow = true; total = MAX_SPECIAL; }
        std::vector<int> ids((size_t)total);      // appended in any order by the kernel: merge_special_kernel searches a sorted list
        HIPCHK(hipMemcpy(ids.data(), spec, (size_t)total * sizeof(int), hipMemcpyDeviceToHost));
        std::sort(ids.begin(), ids.end());
        HIPCHK(hipMemcpy(spec, ids.data(), (size_t)total * sizeof(int), hipMemcpyHostToDevice));
        idx->n_special = total;
    }
    HIPCHK(hipMemcpyAsync(idx->cnorm2 + idx->n, idx->w_tmp.p, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st));
    idx->n += n;
    idx->c8_rows = -1;       // the int8 copy (if any) is rebuilt by the next search that can use it
    idx->c4_rows = -1;       // (the fp4 copy likewise)
    HIPCHK(launch_fill_bias(idx->cnorm2, idx->n, idx->cap + TILE_M, idx->cbias, st));
    idx->maxabs = std::max(idx->maxabs, bits2f(hs.maxabs_bits));
    idx->maxnorm2 = std::max(idx->maxnorm2, bits2f(hs.maxnorm2_bits));
    idx->maxerr2 = std::max(idx->maxerr2, bits2f(hs.maxerr2_bits));
    idx->nonint = idx->nonint || hs.nonint_any;
    idx->nonfp4 = idx->nonfp4 || hs.nonfp4_any;
    HIPCHK(hipStreamSynchronize(st));  // x may be freed by the caller on return
    return TRX_OK;
}

// what a block of host rows of `dtype` can need on the device: integers may fall back to float32 (knn_host.h)
static size_t stage_capacity(int64_t m, int d, int dtype) {
    const int worst = dtype == TRX_DTYPE_BF16 ? STAGED_BF16 : dtype == TRX_DTYPE_I8 ? STAGED_I8 : STAGED_F32;
    return staged_bytes(m, d, worst);
}

int trx_index_add(trx_index* idx, const void* x, int64_t n, int dtype) {
    if (!idx) return fail(TRX_EINVAL, "index is null");
    if (n < 0 || (n > 0 && !x)) return fail(TRX_EINVAL, "bad vector block");
    const size_t esz = (size_t)host_dtype_size(dtype);
    if (!esz) return fail(TRX_EINVAL, "unknown dtype");
    if (n == 0) return TRX_OK;
    int rc = set_device(idx); if (rc) return rc;
    if (!idx->copy_stream) HIPCHK(hipStreamCreateWithFlags(&idx->copy_stream, hipStreamNonBlocking));
    // The rows go through the staging pipeline of knn_host.cpp in blocks whose staged + widened copies stay under 1.5 GiB;
    // int8 rows (given, or narrowed from wider integers) cross PCIe as bytes and become bf16 -- every int8 value is one -- on
    // the device.
    const int64_t rows_per = std::max<int64_t>(1, ((int64_t)1 << 30) / ((int64_t)idx->d * 4));
    struct Release { trx_index* i; ~Release() { i->w_stage[0].release(); i->w_wide.release(); i->reserve_rows = 0; } } guard{idx};
    idx->reserve_rows = idx->n + n;
    int prefer = STAGED_I8;
    for (int64_t r0 = 0; r0 < n; r0 += rows_per) {
        const int64_t m = std::min(rows_per, n - r0);
        if ((rc = idx->w_stage[0].reserve(stage_capacity(m, idx->d, dtype)))) return rc;
        int form = prefer;
        HIPCHK(stage_host_rows((const char*)x + (size_t)r0 * idx->d * esz, dtype, m, idx->d, idx->w_stage[0].p, &form, idx->copy_stream));
        if (form == STAGED_F32) prefer = STAGED_F32;      // (a wide integer turned up: later blocks do not try the narrow form first)
        if (form == STAGED_I8) {
            if ((rc = idx->w_wide.reserve(staged_bytes(m, idx->d, STAGED_BF16)))) return rc;
            HIPCHK(launch_widen_i8((const signed char*)idx->w_stage[0].p, m, idx->d, (bf16_t*)idx->w_wide.p, nullptr));
            rc = trx_index_add_device(idx, idx->w_wide.p, m, TRX_DTYPE_BF16, nullptr);
        } else {
            rc = trx_index_add_device(idx, idx->w_stage[0].p, m, form == STAGED_F32 ? TRX_DTYPE_F32 : TRX_DTYPE_BF16, nullptr);
        }
        if (rc) return rc;
    }
    return TRX_OK;
}

// ---- search --------------------------------------------------------------------------------

// certificate failures of a batch that are re-done by the exact scan inside the enqueued work (a device-side count
// decides how many of these slots do anything); more than that -- near-duplicate clusters, adversarial data -- are
// completed by trx_index_search_finish once the count has been read back
constexpr int INLINE_FALLBACK = 4;
// per batch: four counters -- [0] flagged by the select kernel, [1] still uncertified after the wide re-score, [2] after the
// re-scan (-> exact scan), [3] how many the re-scan took -- three lists of query numbers and two of thresholds (floats)
constexpr size_t FLAG_WORDS = 8 + 5 * 65536;      // (8 header words: the four counters, [4] the re-scan tier's form)
constexpr int RESCAN_MAX = 16384;                 // queries per batch the re-scan is sized for (64 query tiles); more take the exact scan

// The number of corpus splits of a scan launch: the one that a simple price list makes cheapest.  A launch is nqt x S
// workgroups of ceil(ntiles / S) corpus tiles each, 256 at a time (one per CU); a last round that is not full still costs most
// of a round (its workgroups run alone on their CUs, at a higher clock: 0.7 + 0.3 x its fill); fewer than 4 splits share
// thresholds and L2 lines worse (measured at 1M rows: S = 1 + 5.8 %, 2 + 3.8 %), more than 4 slightly worse too (8: + 2 %);
// and every split adds its lists to the select kernel's work (0.11 ms per split and 65,536 queries).
// cus: the device's compute units = the workgroups resident at once (one per CU: the kernel takes most of a CU's LDS), read from
// the device (256 on MI355X; the advisor's note on round 5: not a literal).  form_scale: the time of a tile relative to the bf16
// form at the same Kp -- 0.5 when the int8 form may run, 0.25 for fp4 (which form runs is decided on the device; the host knows
// which ones the index allows): it weighs the scan against the select pass's cost per split.  The per-split penalties are one
// MI355X sweep's (profiles/r05_split_sweep.json).
static int choose_splits(int nqt, int ntiles, int Kp, int cus, double form_scale = 1.0, int cap = 0) {      // cap: the most the caller's list storage takes (0: none)
    cus = std::max(1, cus);
    int smax = std::max(1, std::min(std::min(ntiles, cus), std::max(4, 8 * cus / std::max(1, nqt))));
    if (cap > 0) smax = std::max(1, std::min(smax, cap));
    const double t_tile = 0.0172 * Kp / 768.0 * form_scale;        // ms per 256 x 256 tile of the scan on one CU
    double best = 1e300;
    int bs = 1;
    for (int s = 1; s <= smax; ++s) {
        const int tps = (ntiles + s - 1) / s;
        if ((ntiles + tps - 1) / tps != s) continue;               // (trailing splits would be empty: the launch of a smaller s)
        const double r = (double)nqt * s / (double)cus;
        const double full = std::floor(r), frac = r - full;
        const double rounds = full + (frac > 1e-9 ? 0.7 + 0.3 * frac : 0.0);
        const double pen = s == 1 ? 0.058 : s == 2 ? 0.038 : s == 3 ? 0.02 : s <= 8 ? 0.005 * (s - 4) : 0.02;
        const double cost = rounds * tps * t_tile * (1.0 + pen) + 0.11 * s * nqt / 256.0;
        if (cost < best * (1.0 - 1e-9)) { best = cost; bs = s; }
    }
    return bs;
}

static int device_cus(trx_index* idx) {
    if (idx->cus <= 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, idx->device) != hipSuccess || v <= 0) v = 256;
        idx->cus = v;
    }
    return idx->cus;
}

static int search_batch(trx_index* idx, const void* q, const float* qnorm2, const float* qerr2, int64_t nq, int is_bf, int q_split, int batch_no,
                        float eps_rel, int k, float* D, int64_t* I, double* S64, hipStream_t st) {
    const int d = idx->d, Kp = idx->Kp;
    const int64_t q_pad = round_up64(nq, TILE_N);
    const int nqt = (int)(q_pad / TILE_N);
    const int64_t n_pad = round_up64(idx->n, TILE_M);
    const int ntiles = (int)(n_pad / TILE_M);
    // Corpus splits per query tile.  At least enough workgroups to fill the 256 CUs, and at least 4: with the
    // XCD-contiguous block order of the scan kernel the 32 workgroups that share an XCD are then 8 query tiles x 4
    // splits, so the 8 query tiles (3 MB at K = 768) stay resident in the XCD's 4 MiB L2 while 4 corpus streams
    // pass through it, each read by 8 workgroups (tools/scan_lab.hip: fill path 45 -> 67 GB/s per CU).
    // Round 5: that is the rule for query counts that fill whole rounds of 256 workgroups (C1: 256 query tiles x 4).  For the
    // others choose_splits prices every count: 18 query tiles x 15 splits = 270 workgroups ran two rounds where 14 splits run one
    // (7.96 -> 4.40 ms over 800,000 rows), 50 x 6 = 300 likewise (11.75 -> 7.52 ms over 500,000; profiles/r05_split_sweep.json).
    // (the forms this index allows: the int8 copy needs small integers in a plain index, the fp4 one values E2M1 holds)
    const bool may8 = idx->mode == MODE_PLAIN && !idx->nonint && idx->maxabs <= 127.f && d >= 256 && !q_split && !getenv("TRX_NO_I8");
    const bool may4 = may8 && !idx->nonfp4 && !getenv("TRX_NO_FP4");
    int nsplits = choose_splits(nqt, ntiles, Kp, device_cus(idx), may4 ? 0.25 : may8 ? 0.5 : 1.0);
    { const char* e = getenv("TRX_NSPLITS"); if (e) nsplits = std::max(1, std::min(atoi(e), ntiles)); }
    int tps = (ntiles + nsplits - 1) / nsplits;
    nsplits = (ntiles + tps - 1) / tps;  // drop empty trailing splits
    // TRX_FAST_MAX_K < k <= TRX_WIDE_MAX_K: the first scan serves kf = 24 (its exact scores seed the second scan's threshold)
    const bool bigk = k > TRX_FAST_MAX_K;
    const int kf = bigk ? TRX_FAST_MAX_K : k;
    const int kprime = kf <= 12 ? 16 : 32;
    const int nlists = nsplits * LISTS_PER_SPLIT;
    // a lane lists ~20 (kprime 16) to ~40 (kprime 32) rows per split on random data; a list that could not take another
    // tile (32 rows) is compacted by the scan kernel, so leave room: 127 = the most a 7-bit counter counts
    const int cap = 127, cap_alloc = cap + 1;

    int rc;
    DevPool& pl = pool_of(idx->device);      // (the caller holds its mutex)
    // bf16 queries of the operand's own width, a whole number of query tiles, 16-byte aligned: the scan kernel reads them
    // where they lie (no padded copy: 0.27 ms per 65,536 x 768 batch)
    const bool direct = is_bf && !q_split && Kp == d && q_pad == nq && (((uintptr_t)q) & 15) == 0;
    if (!direct && (rc = pl.qg.reserve((size_t)q_pad * Kp * sizeof(bf16_t)))) return rc;
    if ((rc = pl.cand.reserve((size_t)q_pad * nlists * cap_alloc * sizeof(u64)))) return rc;
    if ((rc = pl.cnt.reserve((size_t)q_pad * nlists * sizeof(u32)))) return rc;
    if ((rc = pl.thr.reserve((size_t)q_pad * nlists * sizeof(u64)))) return rc;
    if ((rc = pl.gthr.reserve((size_t)q_pad * 4 * sizeof(u32)))) return rc;

    // query operand + norms
    if (!direct) {
        HIPCHK(hipMemsetAsync(pl.qg.p, 0, (size_t)q_pad * Kp * sizeof(bf16_t), st));
        HIPCHK(launch_build_operand(q, is_bf, q_split ? 2 : 0, nq, d, d, (bf16_t*)pl.qg.p, Kp, st));
    }
    // this batch's slice of the flag workspace (reserved for all batches by the caller: a DevBuf may not grow while
    // earlier batches of the same call still point into it)
    // [0] queries the select kernel could not certify, [1] those the wide re-score could not either (-> exact scan)
    int* nflag = (int*)idx->w_flag.p + (size_t)batch_no * FLAG_WORDS;
    int* flagged = nflag + 8;
    int* flagged2 = flagged + 65536;
    int* flagged3 = flagged2 + 65536;
    float* seed1 = (float*)(flagged3 + 65536);
    float* seed2 = seed1 + 65536;
    HIPCHK(hipMemsetAsync(nflag, 0, 8 * sizeof(int), st));

    ScanParams sp{};
    sp.corpus = idx->Cg; sp.queries = direct ? (const bf16_t*)q : (const bf16_t*)pl.qg.p; sp.cbias = idx->cbias;
    sp.Kp = Kp; sp.n_valid = (int)idx->n; sp.ntiles = ntiles; sp.tiles_per_split = tps; sp.nsplits = nsplits;
    sp.nqtiles = nqt; sp.nq_valid = (int)nq; sp.kprime = kprime; sp.cap = cap; sp.cap_alloc = cap_alloc;
    sp.cand = (u64*)pl.cand.p; sp.cand_cnt = (u32*)pl.cnt.p; sp.cand_thr = (u64*)pl.thr.p;
    { const char* dbg = getenv("TRX_SCAN_DEBUG"); sp.debug = dbg ? atoi(dbg) : 0; }
    sp.stamp_out = nullptr;
#ifdef TRX_STAMP_BUILD
    if ((rc = idx->w_stamp.reserve((size_t)nqt * nsplits * 8 * 12 * sizeof(unsigned long long)))) return rc;
    HIPCHK(hipMemsetAsync(idx->w_stamp.p, 0, (size_t)nqt * nsplits * 8 * 12 * sizeof(unsigned long long), st));
    sp.stamp_out = (unsigned long long*)idx->w_stamp.p;
#endif
    sp.g_thr = (u32*)pl.gthr.p;
    HIPCHK(hipMemsetAsync(sp.g_thr, 0, (size_t)q_pad * 4 * sizeof(u32), st));
    if (idx->pend.approx) {
        // approximate operands: a query's listing threshold is its bound minus 2 eps_q (the select kernel's eps_q), so that the
        // lists hold every row that can reach the top k and the candidates certify on their exact scores without a second scan
        if ((rc = pl.slk.reserve((size_t)q_pad * sizeof(float)))) return rc;
        HIPCHK(launch_slack(qnorm2, nq, q_pad, eps_rel, idx->pend.eps_round, idx->maxnorm2, qerr2, idx->maxerr2, idx->metric == TRX_METRIC_L2 ? 1 : 0, (float*)pl.slk.p, st));
        sp.slack = (const float*)pl.slk.p;
    }
    // The integer class has an int8 form of the scan (knn_scan.hip, I8: twice the MACs per instruction, half the bytes per
    // component).  What the HOST knows is the corpus side -- integers, |value| <= 127, stored as they are (plain mode);
    // whether the queries qualify is known on the device only (w_cls[1], classify_kernel), so both launches are enqueued
    // and gated there: exactly one of them runs, the other returns at once.  Thresholds (bootstrap, g_thr), lists and
    // everything downstream are the same keys in both forms.
    const bool try8 = idx->mode == MODE_PLAIN && !idx->nonint && idx->maxabs <= 127.f && d >= 256 && !q_split && !getenv("TRX_NO_I8");
    if (try8) {
        const int Kp8 = (int)round_up64(std::max(d, 512), 256);
        const int64_t rows8 = idx->cap + TILE_M;
        if (idx->c8_cap != rows8 || idx->Kp8 != Kp8) {
            if (idx->C8) (void)hipFree(idx->C8);
            if (idx->cbias8) (void)hipFree(idx->cbias8);
            idx->C8 = nullptr; idx->cbias8 = nullptr; idx->c8_cap = 0; idx->c8_rows = -1;
            HIPCHK(hipMalloc((void**)&idx->C8, (size_t)rows8 * Kp8));
            HIPCHK(hipMalloc((void**)&idx->cbias8, (size_t)rows8 * sizeof(int)));
            idx->c8_cap = rows8; idx->Kp8 = Kp8;
        }
        if (idx->c8_rows != idx->n) {
            HIPCHK(hipMemsetAsync(idx->C8, 0, (size_t)rows8 * Kp8, st));
            HIPCHK(launch_build_operand_i8(idx->Cg, 1, idx->n, d, Kp, idx->C8, Kp8, 1.f, st));
            HIPCHK(launch_fill_bias_i32(idx->cnorm2, idx->n, rows8, idx->cbias8, st));
            idx->c8_rows = idx->n;
        }
        if ((rc = pl.qg8.reserve((size_t)q_pad * Kp8))) return rc;
        HIPCHK(hipMemsetAsync(pl.qg8.p, 0, (size_t)q_pad * Kp8, st));
        HIPCHK(launch_build_operand_i8(q, is_bf, nq, d, d, (signed char*)pl.qg8.p, Kp8, idx->metric == TRX_METRIC_L2 ? 2.f : 1.f, st));
        sp.gate = (const int*)idx->w_cls.p + 1; sp.gate_want = 0;
        idx->pend.tried_i8 = 1;
    }
    // ... and a corpus of bit vectors (every value one of 0, +-1, +-2, +-3, +-4, +-6) an fp4 form: a third launch of each pair,
    // run when the queries are of that kind too (w_cls[1] == 2)
    const bool try4 = try8 && !idx->nonfp4 && !getenv("TRX_NO_FP4");
    if (try4) {
        const int Kp4 = (int)round_up64(std::max(d, 1024), 512) / 2;      // bytes per row: a K-step is 128 of them = 256 components, at least 4, an even number
        const int64_t rows4 = idx->cap + TILE_M;
        if (idx->c4_cap != rows4 || idx->Kp4 != Kp4) {
            if (idx->C4) (void)hipFree(idx->C4);
            idx->C4 = nullptr; idx->c4_cap = 0; idx->c4_rows = -1;
            HIPCHK(hipMalloc((void**)&idx->C4, (size_t)rows4 * Kp4));
            idx->c4_cap = rows4; idx->Kp4 = Kp4;
        }
        if (idx->c4_rows != idx->n) {
            HIPCHK(hipMemsetAsync(idx->C4, 0, (size_t)rows4 * Kp4, st));
            HIPCHK(launch_build_operand_fp4(idx->Cg, 1, idx->n, d, Kp, idx->C4, Kp4, st));
            idx->c4_rows = idx->n;
        }
        if ((rc = pl.qg4.reserve((size_t)q_pad * Kp4))) return rc;
        HIPCHK(hipMemsetAsync(pl.qg4.p, 0, (size_t)q_pad * Kp4, st));
        HIPCHK(launch_build_operand_fp4(q, is_bf, nq, d, d, (unsigned char*)pl.qg4.p, Kp4, st));
    }
    // seed the shared thresholds: every query tile scans a few tiles (selection bookkeeping only, no lists)
    const bool boot = !getenv("TRX_NO_BOOT") && ntiles > 2;
    if (boot) {
        ScanParams bp = sp;
        bp.bootstrap = 1; bp.nsplits = 1; bp.tiles_per_split = ntiles;
        bp.boot_tiles = getenv("TRX_BOOT_TILES") ? std::max(2, atoi(getenv("TRX_BOOT_TILES"))) : 16;
        HIPCHK(launch_scan(bp, idx->metric, st));
        if (try8) {      // the int8 twin of the bootstrap, gated like the main pair
            ScanParams b8 = bp;
            b8.i8 = 1; b8.gate_want = 1;
            b8.corpus = (const bf16_t*)idx->C8; b8.queries = (const bf16_t*)pl.qg8.p; b8.cbias = (const float*)idx->cbias8; b8.Kp = idx->Kp8 / 2;
            HIPCHK(launch_scan(b8, idx->metric, st));
        }
        if (try4) {      // (the fp4 form computes in the bf16 form's arithmetic: its float bias, its accumulators)
            ScanParams b4 = bp;
            b4.i8 = 2; b4.gate_want = 2;
            b4.corpus = (const bf16_t*)idx->C4; b4.queries = (const bf16_t*)pl.qg4.p; b4.Kp = idx->Kp4 / 2;
            HIPCHK(launch_scan(b4, idx->metric, st));
        }
    }
    sp.bootstrap = 0; sp.boot_tiles = 0;
    if (idx->timing) HIPCHK(hipEventRecord(idx->ev[0], st));
    HIPCHK(launch_scan(sp, idx->metric, st));
    if (try8) {
        ScanParams s8 = sp;
        s8.i8 = 1; s8.gate_want = 1;
        s8.corpus = (const bf16_t*)idx->C8; s8.queries = (const bf16_t*)pl.qg8.p; s8.cbias = (const float*)idx->cbias8;
        s8.Kp = idx->Kp8 / 2;          // the kernel counts 2-byte units
        HIPCHK(launch_scan(s8, idx->metric, st));
    }
    if (try4) {
        ScanParams s4 = sp;
        s4.i8 = 2; s4.gate_want = 2;
        s4.corpus = (const bf16_t*)idx->C4; s4.queries = (const bf16_t*)pl.qg4.p; s4.Kp = idx->Kp4 / 2;
        HIPCHK(launch_scan(s4, idx->metric, st));
    }
    sp.gate = nullptr;      // (everything below -- the re-scan of uncertified queries -- is the bf16 form, ungated)
    if (idx->timing) HIPCHK(hipEventRecord(idx->ev[1], st));
#ifdef TRX_STAMP_BUILD
    {
        const int nwg = nqt * nsplits;
        std::vector<unsigned long long> h((size_t)nwg * 8 * 12);
        HIPCHK(hipMemcpyAsync(h.data(), sp.stamp_out, h.size() * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        double cyc = 0, comp = 0, tiles = 0, wt = 0, bl = 0, mf = 0, bm = 0, ld = 0, rounds = 0;
        for (size_t i = 0; i < h.size(); i += 12) {
            cyc += (double)h[i]; comp += (double)h[i + 1]; tiles += (double)h[i + 3];
            wt += (double)h[i + 4]; bl += (double)h[i + 5]; mf += (double)h[i + 6]; bm += (double)h[i + 7]; ld += (double)h[i + 8]; rounds += (double)h[i + 9];
        }
        // how far apart the workgroups that share a corpus stream run: the stream of split s is read by the 8 query tiles an XCD
        // holds together (v = qtile * nsplits + split, 32 consecutive v per XCD and round); spread = latest - earliest wall clock
        // (s_memrealtime, 100 MHz) at the head of the split's middle tile, per sharing group
        {
            // the kernel's block remap: XCD x gets the contiguous range of v whose size is q or q + 1 (xcd_remap in knn_scan.hip)
            std::vector<double> mid((size_t)nwg, 0.0);
            const int qd = nwg >> 3, rr = nwg & 7;
            for (int bid = 0; bid < nwg; ++bid) {
                const int x = bid & 7;
                const int base = x < rr ? x * (qd + 1) : rr * (qd + 1) + (x - rr) * qd;
                const int v = base + (bid >> 3);
                mid[v] = (double)h[((size_t)bid * 8 + 0) * 12 + 10];       // wave 0 of the block
            }
            double sum = 0, mx = 0; int groups = 0;
            for (int x = 0; x < 8; ++x) {
                const int base = x < rr ? x * (qd + 1) : rr * (qd + 1) + (x - rr) * qd, cnt = x < rr ? qd + 1 : qd;
                for (int c0 = 0; c0 < cnt; c0 += 32) {
                    for (int sp = 0; sp < nsplits; ++sp) {
                        double lo = 1e300, hi = 0; int members = 0;
                        for (int j = c0; j < std::min(cnt, c0 + 32); ++j) {
                            const int v = base + j;
                            if (v % nsplits != sp || mid[v] == 0.0) continue;
                            lo = std::min(lo, mid[v]); hi = std::max(hi, mid[v]); ++members;
                        }
                        if (members >= 2) { sum += (hi - lo) * 0.01; mx = std::max(mx, (hi - lo) * 0.01); ++groups; }
                    }
                }
            }
            if (groups) fprintf(stderr, "[stamp] workgroups sharing a corpus stream (%d groups): mean spread %.1f us, largest %.1f us at the head of the middle tile "
                                        "(a tile takes ~17.5 us, an L2 line lives ~12 us)\n", groups, sum / groups, mx);
        }
        // (s_memtime ticks are shader cycles)
        if (rounds > 0)
            fprintf(stderr, "[stamp] per wave and L/M round (shader cycles): load issue %.2f, counter waits %.2f, barrier after L %.2f, MFMA phase %.2f, barrier after M %.2f "
                            "(sum %.2f; %.0f rounds per wave-tile)\n", ld / rounds, wt / rounds, bl / rounds, mf / rounds, bm / rounds,
                    (ld + wt + bl + mf + bm) / rounds, rounds / tiles);
        // listed rows: the final list counts (exact when nothing was compacted)
        std::vector<u32> hc((size_t)q_pad * nlists);
        HIPCHK(hipMemcpyAsync(hc.data(), sp.cand_cnt, hc.size() * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        double app = 0; u32 mx = 0;
        for (u32 c : hc) { app += c; mx = std::max(mx, c); }
        fprintf(stderr, "[stamp] per wave and tile: bookkeeping %.0f cycles, %.5f lists compacted, %.2f rows listed (%.0f per query; fullest list %u of %d)\n",
                cyc / tiles, comp / tiles, app / tiles, app / (double)q_pad, mx, cap);
    }
#endif

    SelectParams se{};
    se.special = (const int*)idx->w_special.p; se.nspecial = idx->n_special;
    se.cand = sp.cand; se.cand_cnt = sp.cand_cnt; se.cand_thr = sp.cand_thr; se.nlists = nlists; se.cap_alloc = cap_alloc;
    if (keeps_f32(idx->mode)) { se.corpus_orig = idx->Co; se.ld_c = d; se.corpus_is_bf16 = 0; }
    else { se.corpus_orig = idx->Cg; se.ld_c = Kp; se.corpus_is_bf16 = 1; }
    se.query_orig = q; se.ld_q = d; se.query_is_bf16 = is_bf;
    se.d = d; se.metric = idx->metric; se.k = kf; se.nq = (int)nq; se.n = idx->n; se.exact_class = (const int*)idx->w_cls.p;
    se.eps_rel = eps_rel; se.eps_round = idx->pend.eps_round; se.qnorm2 = qnorm2; se.ymax_norm2 = idx->maxnorm2;
    se.qerr2 = qerr2; se.yerr2_max = idx->maxerr2;
    se.D = D; se.I = I; se.S64 = S64; se.flagged = flagged; se.nflagged = nflag; se.flag_seed = seed1; se.compact = 0;
    if (bigk) {      // the first scan's 24 exact scores per query go to a scratch of their own
        if ((rc = pl.d1.reserve((size_t)q_pad * kf * sizeof(float)))) return rc;
        if ((rc = pl.i1.reserve((size_t)q_pad * kf * sizeof(int64_t)))) return rc;
        se.D = (float*)pl.d1.p; se.I = (int64_t*)pl.i1.p; se.S64 = nullptr;
    }
    HIPCHK(launch_select(se, st));
    // The re-scan tier (third tier of a k <= 24 search, third scan of the two-scan path): the still-unproven queries of `list`,
    // packed, scanned again with every threshold FIXED at the query's seed, then the wide re-score over those lists; what it
    // cannot prove goes to out_list, and so does what exceeds its capacity cap_q.  Usually a handful of queries -- one query
    // tile -- and with the search's own corpus splits that is 4 workgroups streaming the whole corpus: 14.8 ms at 1,000,000
    // rows, as long as a scan of 16,384 queries.  So the tier is enqueued in TWO forms and a device-side flag (written by the
    // gather kernel from the count) runs one of them: up to an eighth of the query tiles with eight times the splits (same
    // list storage: an eighth of the queries x eight times the lists), or all of cap_q with the search's splits.
    auto rescan_tier = [&](const int* list, const int* list_cnt, const float* seeds, int cap_q, SelectParams we, int* out_list, int* out_cnt) -> int {
        int rc2;
        if ((rc2 = pl.qg2.reserve((size_t)cap_q * Kp * sizeof(bf16_t)))) return rc2;
        if ((rc2 = pl.gthr2.reserve((size_t)cap_q * 4 * sizeof(u32)))) return rc2;
        const int tiles = cap_q / TILE_N, small_tiles = std::max(1, tiles / 8), factor = tiles / small_tiles;
        int ns_small = choose_splits(small_tiles, ntiles, Kp, device_cus(idx), 1.0, std::min(std::min(nsplits * factor, ntiles), 256));      // (whole rounds here too)
        const int tps_small = (ntiles + ns_small - 1) / ns_small;
        ns_small = (ntiles + tps_small - 1) / tps_small;
        const bool two = factor > 1 && ns_small > nsplits && !getenv("TRX_RESCAN_ONE_FORM");
        int* gate = nflag + 4;
        HIPCHK(launch_gather_rescan(list, list_cnt, seeds, cap_q, sp.queries, Kp, (bf16_t*)pl.qg2.p, (u32*)pl.gthr2.p, nflag + 3,
                                    small_tiles * TILE_N, two ? gate : nullptr, st));
        ScanParams rp = sp;
        rp.queries = (const bf16_t*)pl.qg2.p; rp.nqtiles = cap_q / TILE_N; rp.nq_valid = cap_q; rp.nq_valid_dev = nflag + 3;
        rp.g_thr = (u32*)pl.gthr2.p; rp.fixed_thr = 1; rp.bootstrap = 0; rp.boot_tiles = 0; rp.slack = nullptr;      // (the seeds are lowered already)
        rp.kprime = we.k <= 12 ? 16 : 32;
        we.compact = 1; we.extrap = 0;
        if (two) {
            ScanParams rs = rp;
            rs.nqtiles = small_tiles; rs.nq_valid = small_tiles * TILE_N; rs.nsplits = ns_small; rs.tiles_per_split = tps_small;
            rs.gate = gate; rs.gate_want = 1;
            HIPCHK(launch_scan(rs, idx->metric, st));
            SelectParams ws = we; ws.nlists = ns_small * LISTS_PER_SPLIT; ws.gate = gate; ws.gate_want = 1;
            HIPCHK(launch_wide_rescore(ws, list, nflag + 3, nullptr, out_list, out_cnt, nullptr, st));
            rp.gate = gate; rp.gate_want = 0; we.gate = gate; we.gate_want = 0;
        }
        HIPCHK(launch_scan(rp, idx->metric, st));
        HIPCHK(launch_wide_rescore(we, list, nflag + 3, nullptr, out_list, out_cnt, nullptr, st));
        HIPCHK(launch_append_tail(list, list_cnt, cap_q, out_list, out_cnt, st));
        return TRX_OK;
    };
    int* final_list; int* final_cnt;
    if (bigk) {
        // Second scan, for EVERY query: thresholds fixed at a guess of the k-th best key (bigk_seed_kernel), every row above it
        // listed by construction; the wide re-score ranks those rows exactly and proves the answer -- k exact scores above
        // (threshold + eps).  A query it cannot prove (the guess left fewer than k rows: clustered data; or a crowd of ties
        // at the threshold) comes back with a better threshold, from the rows the second scan did find, for a third scan of
        // those queries alone; what fails that too, or exceeds its capacity, takes the exact scan.
        //   counters: [1] all queries (the identity list), [0] unproven after the second scan, [3] of those, the ones the
        //   third scan has room for, [2] unproven after that (-> exact scan)
        HIPCHK(hipMemsetAsync(nflag, 0, 4 * sizeof(int), st));      // (the select kernel's flags concerned the 24)
        HIPCHK(launch_bigk_seeds(idx->metric == TRX_METRIC_L2 ? 1 : 0, (const float*)pl.d1.p, (const int64_t*)pl.i1.p, (int)nq, kf, k, qnorm2, eps_rel,
                                 idx->pend.eps_round, idx->maxnorm2, qerr2, idx->maxerr2, flagged2, nflag + 1, seed2, st));
        const int rq2 = (int)q_pad;
        if ((rc = pl.qg2.reserve((size_t)rq2 * Kp * sizeof(bf16_t)))) return rc;
        if ((rc = pl.gthr2.reserve((size_t)rq2 * 4 * sizeof(u32)))) return rc;
        HIPCHK(launch_gather_rescan(flagged2, nflag + 1, seed2, rq2, sp.queries, Kp, (bf16_t*)pl.qg2.p, (u32*)pl.gthr2.p, nflag + 1, 0, nullptr, st));
        ScanParams rp = sp;
        rp.queries = (const bf16_t*)pl.qg2.p; rp.nqtiles = rq2 / TILE_N; rp.nq_valid = rq2; rp.nq_valid_dev = nflag + 1;
        rp.g_thr = (u32*)pl.gthr2.p; rp.fixed_thr = 1; rp.bootstrap = 0; rp.boot_tiles = 0; rp.slack = nullptr; rp.kprime = 32;
        HIPCHK(launch_scan(rp, idx->metric, st));
        SelectParams we = se; we.compact = 1; we.k = k; we.D = D; we.I = I; we.S64 = S64; we.extrap = 1;
        HIPCHK(launch_wide_rescore(we, flagged2, nflag + 1, seed2, flagged, nflag, seed1, st));
        if ((rc = rescan_tier(flagged, nflag, seed1, (int)std::min<int64_t>(RESCAN_MAX, q_pad), we, flagged3, nflag + 2))) return rc;
        final_list = flagged3; final_cnt = nflag + 2;
    } else {
    // second tier: flagged queries are re-scored over ALL their listed rows (the lists are still in the shared workspace
    // here); what that cannot certify either goes on to the exact scan
    HIPCHK(launch_wide_rescore(se, flagged, nflag, seed1, flagged2, nflag + 1, seed2, st));
    // third tier: the MFMA scan again, for the still-uncertified queries only, with every threshold FIXED at the query's seed --
    // (exact k-th score of the candidates re-scored so far) - 2 eps, a key no row of the true top k falls below -- so that
    // every row that can matter is listed by construction; then the wide re-score over THOSE lists.  A crowd of near-ties
    // around the k-th place costs a few MFMA tile-passes instead of an fp64 scan of the index; what overflows the lists (more
    // than ~4,000 rows above the seed) or the re-scan's capacity goes on to the exact scan.
    final_list = flagged2; final_cnt = nflag + 1;
    if (sp.debug == 0 && !getenv("TRX_NO_RESCAN")) {
        if ((rc = rescan_tier(flagged2, nflag + 1, seed2, (int)std::min<int64_t>(RESCAN_MAX, q_pad), se, flagged3, nflag + 2))) return rc;
        final_list = flagged3; final_cnt = nflag + 2;
    }
    }

    // certificate failures -> exact scan of those queries.  The count stays on the device: the first INLINE_FALLBACK of
    // them are re-done right here, stream-ordered (slots beyond the count leave at once: ~10 us when nothing failed, the
    // common case); trx_index_search_finish reads the count and completes what is left.  No host synchronisation.
    if (sp.debug == 0)      // (timing-only debug modes of the scan kernel produce wrong lists: no fall-back then)
        HIPCHK(launch_exact_scan(idx->metric, se.corpus_is_bf16, is_bf, final_list, INLINE_FALLBACK, final_cnt, idx->n,
                                 se.corpus_orig, se.ld_c, q, d, d, k, (double*)idx->w_exact.p, D, I, S64, st));
    if (idx->timing) {      // timing mode is synchronous by contract (trx_index_set_timing)
        HIPCHK(hipStreamSynchronize(st));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, idx->ev[0], idx->ev[1]));
        idx->stats.scan_ms += ms;
    }
    idx->stats.scan_launches += 1;
    idx->stats.n_splits = nsplits;
    idx->pend.no_fallback = sp.debug != 0;
    idx->pend.corpus_orig = se.corpus_orig; idx->pend.ld_c = se.ld_c; idx->pend.corpus_is_bf16 = se.corpus_is_bf16;
    idx->pend.batches.push_back({nflag, final_list, (int)(final_cnt - nflag), q, D, I, S64});
    return TRX_OK;
}

// read back what the enqueued search left on the device; complete the fall-back where more queries failed their
// certificate than the inline slots cover
static int finish_impl(trx_index* idx) {
    auto& pd = idx->pend;
    if (!pd.active) return TRX_OK;
    pd.active = false;
    hipStream_t st = pd.st;
    int cls[2] = {0, 0};
    std::vector<int> nf(pd.batches.size(), 0);
    std::vector<int> cnt4(pd.batches.size() * 4, 0);      // the four counters of every batch (FLAG_WORDS)
    for (size_t b = 0; b < pd.batches.size(); ++b)
        HIPCHK(hipMemcpyAsync(&cnt4[4 * b], pd.batches[b].nflag, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
    if (idx->w_cls.p && !pd.batches.empty()) HIPCHK(hipMemcpyAsync(cls, idx->w_cls.p, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (!pd.batches.empty()) { idx->stats.exact_class = cls[0]; idx->stats.int8_scan = pd.tried_i8 ? cls[1] : 0; }
    const int64_t per = std::max<int64_t>(INLINE_FALLBACK, ((int64_t)1 << 29) / std::max<int64_t>(1, idx->n * 8));
    bool late = false;
    for (size_t b = 0; b < pd.batches.size(); ++b) {
        nf[b] = cnt4[4 * b + pd.batches[b].final_slot];
        idx->stats.n_uncertified += nf[b];
        idx->stats.n_rescored += cnt4[4 * b];
        idx->stats.n_rescanned += cnt4[4 * b + 3];
        if (pd.no_fallback || nf[b] <= INLINE_FALLBACK) continue;
        late = true;
        int rc = idx->w_exact.reserve((size_t)std::min<int64_t>(per, nf[b]) * idx->n * sizeof(double)); if (rc) return rc;
        const auto& pb = pd.batches[b];
        for (int f0 = INLINE_FALLBACK; f0 < nf[b]; f0 += (int)per) {
            const int m = (int)std::min<int64_t>(per, nf[b] - f0);
            HIPCHK(launch_exact_scan(idx->metric, pd.corpus_is_bf16, pd.is_bf, pb.flagged + f0, m, nullptr, idx->n,
                                     pd.corpus_orig, pd.ld_c, pb.q, idx->d, idx->d, pd.k, (double*)idx->w_exact.p, pb.D, pb.I, pb.S64, st));
        }
    }
    if (late && pd.tie_k) {      // the FAISS order was derived from lists the late fall-back has just changed: derive it again
        const float* D2 = (const float*)idx->w_tie.p; (void)D2;
        const int64_t* I2 = (const int64_t*)((char*)idx->w_tie.p + (size_t)pd.tie_nq * pd.tie_k2 * sizeof(float));
        const double* S2 = (const double*)((const char*)I2 + (size_t)pd.tie_nq * pd.tie_k2 * sizeof(int64_t));
        HIPCHK(launch_faiss_ties(pd.tie_nq, pd.tie_k2, pd.tie_k, S2, I2, pd.tie_D, pd.tie_I, pd.tie_S, st));
    }
    if (late) HIPCHK(hipStreamSynchronize(st));
    idx->stats.late_fallback = late ? 1 : 0;
    pd.batches.clear();
    return TRX_OK;
}

static int search_device_core(trx_index* idx, const void* q, int64_t nq, int dtype, int k, float* D,
                              int64_t* I, double* S64, void* stream);

// TRX_TIES_FAISS on an inner-product index: the search runs for k2 = 2k in the canonical order into a scratch of the index,
// and faiss_tie_kernel (knn_select.hip) derives what FAISS' heap returns from it; everything else goes straight through.
static int search_device_impl(trx_index* idx, const void* q, int64_t nq, int dtype, int k, float* D,
                              int64_t* I, double* S64, void* stream) {
    if (!idx || idx->tie_rule != TRX_TIES_FAISS || idx->metric != TRX_METRIC_IP || nq <= 0 || k <= 0)
        return search_device_core(idx, q, nq, dtype, k, D, I, S64, stream);
    if (2 * k > TRX_MAX_K) return fail(TRX_EINVAL, "TRX_TIES_FAISS needs the canonical top 2k: k must be in [1, 1024]");
    if (!q || !D || !I) return fail(TRX_EINVAL, "null query/result pointer");
    int rc = set_device(idx); if (rc) return rc;
    if (idx->pend.active) { rc = finish_impl(idx); if (rc) return rc; }      // (its tie kernel reads w_tie)
    const int k2 = 2 * k;
    if ((rc = idx->w_tie.reserve((size_t)nq * k2 * (sizeof(float) + sizeof(int64_t) + sizeof(double))))) return rc;
    float* D2 = (float*)idx->w_tie.p;
    int64_t* I2 = (int64_t*)((char*)D2 + (size_t)nq * k2 * sizeof(float));
    double* S2 = (double*)((char*)I2 + (size_t)nq * k2 * sizeof(int64_t));
    rc = search_device_core(idx, q, nq, dtype, k2, D2, I2, S2, stream); if (rc) return rc;
    idx->pend.tie_k = k; idx->pend.tie_k2 = k2; idx->pend.tie_nq = nq; idx->pend.tie_D = D; idx->pend.tie_I = I; idx->pend.tie_S = S64;
    HIPCHK(launch_faiss_ties(nq, k2, k, S2, I2, D, I, S64, (hipStream_t)stream));
    if (idx->timing) HIPCHK(hipEventRecord(idx->ev[3], (hipStream_t)stream));
    return TRX_OK;
}

static int search_device_core(trx_index* idx, const void* q, int64_t nq, int dtype, int k, float* D,
                              int64_t* I, double* S64, void* stream) {
    if (!idx) return fail(TRX_EINVAL, "index is null");
    if (nq < 0 || (nq > 0 && (!q || !D || !I))) return fail(TRX_EINVAL, "null query/result pointer");
    if (k <= 0 || k > TRX_MAX_K) return fail(TRX_EINVAL, "k must be in [1, 2048]");
    if (dtype != TRX_DTYPE_F32 && dtype != TRX_DTYPE_BF16) return fail(TRX_EINVAL, "unknown dtype");
    int rc = set_device(idx); if (rc) return rc;
    if (idx->pend.active) { rc = finish_impl(idx); if (rc) return rc; }     // a search begun and never finished
    hipStream_t st = (hipStream_t)stream;
    idx->stats = trx_search_stats{};
    idx->stats.nq = nq;
    if (nq == 0) return TRX_OK;
    const int is_bf = dtype == TRX_DTYPE_BF16;
    const int d = idx->d;
    if (idx->timing) HIPCHK(hipEventRecord(idx->ev[2], st));
    idx->pend = trx_index::Pending{};
    idx->pend.active = true; idx->pend.st = st; idx->pend.is_bf = is_bf; idx->pend.k = k;

    if (idx->n == 0 || k > TRX_WIDE_MAX_K || idx->special_overflow) {
        // (more hostile rows than the index folds in per query: knn_common.h)
        // empty index: all pads.  k > TRX_WIDE_MAX_K: exact scan for every query (documented slow path)
        const int64_t per = std::max<int64_t>(1, ((int64_t)1 << 29) / std::max<int64_t>(1, idx->n * 8));
        if ((rc = idx->w_exact.reserve((size_t)std::min<int64_t>(per, nq) * std::max<int64_t>(idx->n, 1) * sizeof(double)))) return rc;
        const void* corig = keeps_f32(idx->mode) ? (const void*)idx->Co : (const void*)idx->Cg;
        const int64_t ldc = keeps_f32(idx->mode) ? d : idx->Kp;
        const int cbf = keeps_f32(idx->mode) ? 0 : 1;
        const size_t esz = is_bf ? 2 : 4;
        for (int64_t f0 = 0; f0 < nq; f0 += per) {
            const int m = (int)std::min<int64_t>(per, nq - f0);
            HIPCHK(launch_exact_scan(idx->metric, cbf, is_bf, nullptr, m, nullptr, idx->n, corig, ldc,
                                     (const char*)q + (size_t)f0 * d * esz, d, d, k, (double*)idx->w_exact.p,
                                     D + f0 * k, I + f0 * k, S64 ? S64 + f0 * k : nullptr, st));
        }
        idx->stats.n_uncertified = nq;
        return TRX_OK;              // enqueued; trx_index_search_finish waits for it
    }

    // classify the queries once
    rc = idx->w_stats.reserve(sizeof(HostStats)); if (rc) return rc;
    HIPCHK(hipMemsetAsync(idx->w_stats.p, 0, sizeof(HostStats), st));
    // one pass over the queries: the class flags and every query's fp32 norm (the select kernel's error bound)
    // ... and, for fp32 queries, the norm of its rounding error to bf16 (second half of the buffer): with the corpus side's
    // maximum it replaces the a-priori rounding bound of the approximate mode by a measured one (knn_common.h: round_term;
    // TRX_ROUND_BOUND_APRIORI=1 keeps round 4's bound, the A/B switch)
    const int64_t nq_pad = round_up64(nq, TILE_N);
    rc = idx->w_qnorm2.reserve((size_t)2 * nq_pad * sizeof(float)); if (rc) return rc;
    float* qerr2_all = getenv("TRX_ROUND_BOUND_APRIORI") ? nullptr : (float*)idx->w_qnorm2.p + nq_pad;
    HIPCHK(launch_row_stats(q, is_bf, nq, d, d, idx->w_stats.p, (float*)idx->w_qnorm2.p, qerr2_all, st));
    // The ONE host decision a search can need: fp32 queries that are not exact in bf16 against an index that holds only
    // bf16 data turn the index into its split form.  bf16 queries are exact by construction, and a split index stays
    // split, so only fp32 queries on a plain index read their statistics back; everything else stays on the device.
    // (Approx mode, the default since round 4: nothing about the index changes -- the queries' rounding to bf16 is one more
    // term of the key error, paid for by the listing slack; the read-back only chooses the bound.)
    bool q_inexact = false;
    if (!is_bf && idx->mode == MODE_PLAIN) {
        HostStats hs; rc = read_stats(idx, st, &hs); if (rc) return rc;
        if (hs.inexact_any && inexact_mode() == MODE_SPLIT) {
            rc = restructure(idx, idx->cap, MODE_SPLIT, st); if (rc) return rc;
            HIPCHK(launch_fill_bias(idx->cnorm2, idx->n, idx->cap + TILE_M, idx->cbias, st));
        } else if (hs.inexact_any) q_inexact = true;
    }
    const int q_split = idx->mode == MODE_SPLIT;
    // approximate operands: the corpus (approx mode) and / or the queries (fp32 that bf16 does not hold) reach the scan rounded
    // to bf16: |x^.y^ - x.y| <= u (2 + u) |x||y| with u = 2^-8 (round to nearest, 8 significant bits).  fp32 queries against an
    // approx index are rounded whatever their values (no read-back to find out): the bound covers it.
    const bool approx = idx->mode == MODE_APPROX || q_inexact;
    // exact class (integer inputs small enough that every fp32 partial sum and the L2 key are exact): decided by a
    // one-thread kernel from the statistics just gathered; the select kernel reads the flag from device memory
    rc = idx->w_cls.reserve(4 * sizeof(int)); if (rc) return rc;
    HIPCHK(launch_classify(idx->w_stats.p, q_split, idx->nonint ? 1 : 0, idx->maxabs, idx->Kp, idx->d,
                           idx->metric == TRX_METRIC_L2 ? 1 : 0, (idx->nonfp4 || getenv("TRX_NO_FP4")) ? 1 : 0, (int*)idx->w_cls.p, st));
    const float eps_rel = (float)((idx->Kp + 64) * std::ldexp(1.0, -23)) + (q_split ? (float)std::ldexp(1.0, -15) : 0.f);
    // the rounding's share: relative to the product term alone (the L2 key's |y|^2 comes from the exact rows)
    const float eps_round = approx ? (float)(std::ldexp(1.0, -7) * (1.0 + std::ldexp(1.0, -8))) : 0.f;
    idx->pend.approx = approx ? 1 : 0; idx->pend.eps_round = eps_round;
    idx->stats.k_split = idx->Kp;

    // queries per scan launch.  One launch of 65,536 queries is 1,024 workgroups = four rounds of the chip's 256 resident
    // slots; TRX_QUERY_BATCH (a multiple of 256, at most 65,536) cuts a call into shorter launches
    int64_t QB = 65536;
    { const char* e = getenv("TRX_QUERY_BATCH"); if (e) QB = std::max<int64_t>(256, std::min<int64_t>(65536, (atoll(e) / 256) * 256)); }
    const size_t esz = is_bf ? 2 : 4;
    const int nbatches = (int)((nq + QB - 1) / QB);
    DevPool& pl = pool_of(idx->device);
    std::lock_guard<std::mutex> pool_guard(pl.mu);
    // behind whatever search -- of this or another index, on this or another stream -- used the shared workspaces last
    if (pl.last) HIPCHK(hipStreamWaitEvent(st, pl.last, 0));
    else HIPCHK(hipEventCreateWithFlags(&pl.last, hipEventDisableTiming));
    if ((rc = idx->w_flag.reserve((size_t)nbatches * FLAG_WORDS * sizeof(int)))) return rc;
    // the inline fall-back's score rows (INLINE_FALLBACK x n doubles)
    if ((rc = idx->w_exact.reserve((size_t)INLINE_FALLBACK * idx->n * sizeof(double)))) return rc;
    for (int64_t q0 = 0; q0 < nq; q0 += QB) {
        const int64_t m = std::min(QB, nq - q0);
        rc = search_batch(idx, (const char*)q + (size_t)q0 * d * esz, (const float*)idx->w_qnorm2.p + q0, qerr2_all ? qerr2_all + q0 : nullptr, m, is_bf, q_split,
                          (int)(q0 / QB), eps_rel, k,
                          D + q0 * k, I + q0 * k, S64 ? S64 + q0 * k : nullptr, st);
        if (rc) return rc;
    }
    if (idx->n_special > 0)      // the index's hostile rows, by their canonical scores, into every result (knn_select.hip: merge_special_kernel)
        HIPCHK(launch_merge_special(idx->metric, keeps_f32(idx->mode) ? 0 : 1, is_bf, keeps_f32(idx->mode) ? (const void*)idx->Co : (const void*)idx->Cg,
                                    keeps_f32(idx->mode) ? d : idx->Kp, q, d, d, nq, k, (const int*)idx->w_special.p, idx->n_special, idx->n, D, I, S64, st));
    HIPCHK(hipEventRecord(pl.last, st));
    if (idx->timing) HIPCHK(hipEventRecord(idx->ev[3], st));
    return TRX_OK;
}

int trx_index_search_finish(trx_index* idx) {
    if (!idx) return fail(TRX_EINVAL, "index is null");
    int rc = set_device(idx); if (rc) return rc;
    const bool was = idx->pend.active;
    hipStream_t st = idx->pend.st;
    rc = finish_impl(idx); if (rc) return rc;       // (synchronises the stream)
    (void)st;
    if (was && idx->timing && idx->stats.scan_launches > 0) {
        HIPCHK(hipEventSynchronize(idx->ev[3]));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, idx->ev[2], idx->ev[3]));
        idx->stats.total_ms = ms;
    }
    return TRX_OK;
}

int trx_index_search_device_begin(trx_index* idx, const void* q, int64_t nq, int dtype, int k, float* D,
                                  int64_t* I, double* S64, void* stream) {
    return search_device_impl(idx, q, nq, dtype, k, D, I, S64, stream);
}

int trx_index_search_device(trx_index* idx, const void* q, int64_t nq, int dtype, int k, float* D,
                            int64_t* I, void* stream) {
    int rc = search_device_impl(idx, q, nq, dtype, k, D, I, nullptr, stream);
    return rc ? rc : trx_index_search_finish(idx);
}

// internal-but-exported: same as above plus the fp64 scores the cross-shard merge orders by
int trx_index_search_device_s64(trx_index* idx, const void* q, int64_t nq, int dtype, int k, float* D,
                                int64_t* I, double* S64, void* stream) {
    int rc = search_device_impl(idx, q, nq, dtype, k, D, I, S64, stream);
    return rc ? rc : trx_index_search_finish(idx);
}

int trx_index_search(trx_index* idx, const void* q, int64_t nq, int dtype, int k, float* D, int64_t* I) {
    if (!idx) return fail(TRX_EINVAL, "index is null");
    if (nq < 0 || (nq > 0 && (!q || !D || !I))) return fail(TRX_EINVAL, "null query/result pointer");
    if (k <= 0 || k > TRX_MAX_K) return fail(TRX_EINVAL, "k must be in [1, 2048]");
    const size_t esz = (size_t)host_dtype_size(dtype);
    if (!esz) return fail(TRX_EINVAL, "unknown dtype");
    if (nq == 0) return TRX_OK;
    int rc = set_device(idx); if (rc) return rc;
    // Host arrays in, host arrays out (the FAISS protocol as retrieve_faiss.py:71 calls it), in blocks of 65,536 queries: block
    // c is searched (enqueued, stream-ordered) while the worker threads of knn_host.cpp bring block c + 1 into its staged form
    // (int64 counts narrowed to int8, ...) and across PCIe on a stream of its own, and the results of block c go back while
    // block c + 1 is searched (two staging buffers, two result buffers).  Rows staged as int8 are widened to bf16 on the device.
    constexpr int64_t BLOCK = 65536;
    const int64_t rows = std::min<int64_t>(BLOCK, nq);
    const size_t db = (size_t)round_up64((int64_t)((size_t)rows * k * sizeof(float)), 256), ib = (size_t)rows * k * sizeof(int64_t);
    if ((rc = idx->w_io.reserve(2 * (db + ib)))) return rc;
    char* base = (char*)idx->w_io.p;
    float* Dd[2]; int64_t* Id[2];
    for (int i = 0; i < 2; ++i) { Dd[i] = (float*)(base + i * (db + ib)); Id[i] = (int64_t*)((char*)Dd[i] + db); }
    if (!idx->copy_stream) HIPCHK(hipStreamCreateWithFlags(&idx->copy_stream, hipStreamNonBlocking));
    const char* qh = (const char*)q;
    trx_search_stats acc{};
    const int64_t nblk = (nq + BLOCK - 1) / BLOCK;
    auto rows_of = [&](int64_t c) { return std::min<int64_t>(BLOCK, nq - c * BLOCK); };
    int form[2] = {STAGED_I8, STAGED_I8}, prefer = STAGED_I8;
    auto stage = [&](int64_t c) -> int {      // block c of the caller's queries -> w_stage[c & 1] (returns when they are in HBM)
        int r = idx->w_stage[c & 1].reserve(stage_capacity(rows_of(c), idx->d, dtype)); if (r) return r;
        form[c & 1] = prefer;
        HIPCHK(stage_host_rows(qh + (size_t)c * BLOCK * idx->d * esz, dtype, rows_of(c), idx->d, idx->w_stage[c & 1].p, &form[c & 1], idx->copy_stream));
        if (form[c & 1] == STAGED_F32) prefer = STAGED_F32;
        return TRX_OK;
    };
    auto begin = [&](int64_t c) -> int {      // enqueue the search of block c into result set c & 1
        const void* qd = idx->w_stage[c & 1].p;
        int qdt = form[c & 1] == STAGED_F32 ? TRX_DTYPE_F32 : TRX_DTYPE_BF16;
        if (form[c & 1] == STAGED_I8) {
            int r = idx->w_wide.reserve(staged_bytes(rows_of(c), idx->d, STAGED_BF16)); if (r) return r;
            HIPCHK(launch_widen_i8((const signed char*)qd, rows_of(c), idx->d, (bf16_t*)idx->w_wide.p, nullptr));
            qd = idx->w_wide.p;
        }
        return search_device_impl(idx, qd, rows_of(c), qdt, k, Dd[c & 1], Id[c & 1], nullptr, nullptr);
    };
    if ((rc = stage(0))) return rc;
    if ((rc = begin(0))) return rc;
    for (int64_t c = 0; c < nblk; ++c) {
        const int64_t m = rows_of(c);
        if (c + 1 < nblk && (rc = stage(c + 1))) return rc;      // the next block is prepared and crosses PCIe while this one is searched
        rc = trx_index_search_finish(idx); if (rc) return rc;
        const trx_search_stats s1 = idx->stats;
        acc.nq += s1.nq; acc.n_uncertified += s1.n_uncertified; acc.n_rescored += s1.n_rescored; acc.n_rescanned += s1.n_rescanned;
        acc.scan_launches += s1.scan_launches; acc.scan_ms += s1.scan_ms; acc.total_ms += s1.total_ms;
        acc.late_fallback |= s1.late_fallback;
        acc.n_splits = s1.n_splits; acc.k_split = s1.k_split; acc.exact_class = s1.exact_class; acc.int8_scan = s1.int8_scan;
        if (c + 1 < nblk && (rc = begin(c + 1))) return rc;      // ... and this block's results go back while the next one is searched
        HIPCHK(unstage_to_host(D + c * BLOCK * k, Dd[c & 1], (size_t)m * k * sizeof(float), idx->copy_stream));
        HIPCHK(unstage_to_host(I + c * BLOCK * k, Id[c & 1], (size_t)m * k * sizeof(int64_t), idx->copy_stream));
    }
    idx->stats = acc;
    return TRX_OK;
}

int trx_host_convert(const void* src, int dtype, int64_t count, void* dst, int to_dtype) {
    if (count < 0 || (count > 0 && (!src || !dst))) return fail(TRX_EINVAL, "null pointer");
    if (!host_dtype_size(dtype)) return fail(TRX_EINVAL, "unknown dtype");
    if (to_dtype != TRX_DTYPE_I8 && to_dtype != TRX_DTYPE_F32) return fail(TRX_EINVAL, "to_dtype is TRX_DTYPE_I8 or TRX_DTYPE_F32");
    if (to_dtype == TRX_DTYPE_I8 && !host_dtype_is_wide_int(dtype) && dtype != TRX_DTYPE_I8) return fail(TRX_EINVAL, "only integer arrays narrow to int8");
    if (to_dtype == TRX_DTYPE_F32 && (dtype == TRX_DTYPE_BF16 || dtype == TRX_DTYPE_I8)) return fail(TRX_EINVAL, "bf16 / int8 rows are staged as they are");
    return convert_host_rows(src, dtype, count, dst, to_dtype == TRX_DTYPE_I8 ? STAGED_I8 : STAGED_F32);
}

int trx_host_threads(void) { return host_threads(); }

int trx_merge_topk_device_s64(int metric, int nlists, int64_t nq, int k, const double* S_lists,
                              const int64_t* I_lists, float* D, int64_t* I, double* S, void* stream) {
    if (metric != TRX_METRIC_IP && metric != TRX_METRIC_L2) return fail(TRX_EINVAL, "unknown metric");
    if (nlists <= 0 || nlists > 16) return fail(TRX_EINVAL, "nlists must be in [1, 16]");
    if (nq < 0 || k <= 0 || k > TRX_MAX_K) return fail(TRX_EINVAL, "bad nq / k");
    if (nq > 0 && (!S_lists || !I_lists || !D || !I)) return fail(TRX_EINVAL, "null pointer");
    HIPCHK(launch_merge(metric, nlists, nq, k, S_lists, I_lists, D, I, S, (hipStream_t)stream));
    return TRX_OK;
}

int trx_merge_topk_device(int metric, int nlists, int64_t nq, int k, const double* S_lists,
                          const int64_t* I_lists, float* D, int64_t* I, void* stream) {
    return trx_merge_topk_device_s64(metric, nlists, nq, k, S_lists, I_lists, D, I, nullptr, stream);
}

int trx_index_set_tie_rule(trx_index* idx, int rule) {
    if (!idx) return fail(TRX_EINVAL, "index is null");
    if (rule != TRX_TIES_BY_ID && rule != TRX_TIES_FAISS) return fail(TRX_EINVAL, "unknown tie rule");
    if (idx->pend.active) { int rc = set_device(idx); if (rc) return rc; rc = finish_impl(idx); if (rc) return rc; }
    idx->tie_rule = rule;
    return TRX_OK;
}

int trx_faiss_tie_order_device(int64_t nq, int k2, int k, const double* S2, const int64_t* I2, float* D, int64_t* I, void* stream) {
    if (nq < 0 || k <= 0 || k2 < 2 * k || k2 > TRX_MAX_K) return fail(TRX_EINVAL, "need 1 <= k, 2k <= k2 <= 2048");
    if (nq > 0 && (!S2 || !I2 || !D || !I)) return fail(TRX_EINVAL, "null pointer");
    HIPCHK(launch_faiss_ties(nq, k2, k, S2, I2, D, I, nullptr, (hipStream_t)stream));
    return TRX_OK;
}

}  // extern "C"
