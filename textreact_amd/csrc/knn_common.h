// knn_common.h -- shared definitions of the gfx950 flat k-NN kernels (internal; the public
// boundary is include/trx_knn.h).  Wave = 64 lanes everywhere; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace trx {

typedef unsigned short bf16_t;  // raw bf16 bits
typedef unsigned long long u64;
typedef unsigned int u32;

// ---- geometry of the scan kernel ---------------------------------------------------------
constexpr int TILE_M = 256;     // corpus rows per tile (MFMA M side)
constexpr int TILE_N = 256;     // queries per workgroup (MFMA N side)
constexpr int BK = 64;          // K-step staged through LDS
constexpr int SCAN_THREADS = 512;
constexpr int LISTS_PER_SPLIT = 8;  // a list belongs to ONE LANE of the scan kernel: (query, corpus split, wave row 0-1, row quad 0-3);
                                    // its slot counter lives in a register of that lane, nobody else writes the list
constexpr int KEEP = 32;        // entries the select kernel re-scores exactly (>= TRX_FAST_MAX_K)

// ---- bf16 helpers ------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((u32)h) << 16); }
// round-to-nearest-even; NaN stays NaN (quiet), inputs are finite in every caller that matters
__device__ __forceinline__ bf16_t f32_to_bf16_rn(float f) {
    u32 u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// ---- order-preserving packing of (key, id) -------------------------------------------------
// comp = ordkey(key) << 32 | ~id : larger comp == ranked earlier (key descending, id ascending).
// comp == 0 is the "empty" sentinel (ordkey never produces 0 for a non-NaN key with ~id == 0
// only when id == 0xFFFFFFFF, which is never a valid row).
__device__ __forceinline__ u32 ordkey(float f) {
    u32 u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordkey_inv(u32 o) {
    u32 u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}
__device__ __forceinline__ u64 make_comp(float key, u32 id) {
    return ((u64)ordkey(key) << 32) | (u64)(~id);
}
__device__ __forceinline__ float comp_key(u64 c) { return ordkey_inv((u32)(c >> 32)); }
__device__ __forceinline__ u32 comp_id(u64 c) { return ~(u32)c; }

// order-preserving map of a finite-or-inf double to u64 (larger == larger value)
__device__ __forceinline__ u64 orddbl(double d) {
    u64 u = (u64)__double_as_longlong(d);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// ---- wave-wide bitonic sorts (64 lanes, one element per lane) -------------------------------
__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m) {
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return ((u64)hi << 32) | lo;
}

// After the call lane 0 holds the largest value, lane 63 the smallest.
__device__ __forceinline__ u64 wave_sort_desc(u64 v, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            u64 o = shfl_xor_u64(v, j);
            bool lower = (lane & j) == 0;
            bool desc = (lane & k) == 0;          // k == 64: always true -> final order descending
            bool keep_max = (lower == desc);
            u64 mx = v > o ? v : o, mn = v > o ? o : v;
            v = keep_max ? mx : mn;
        }
    }
    return v;
}

// Sort (score key, id) pairs: lane 0 = largest skey, ties -> smallest id.  skey is an orddbl()
// style key where LARGER means ranked earlier (callers negate for L2).
template <int W = 64>   // W = 32: the two 32-lane halves of the wave are sorted separately, each best first
__device__ __forceinline__ void wave_sort_pairs(u64& skey, u32& id, int lane) {
#pragma unroll
    for (int k = 2; k <= W; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            u64 ok = shfl_xor_u64(skey, j);
            u32 oi = __shfl_xor(id, j, 64);
            bool lower = (lane & j) == 0;
            bool desc = k == W || (lane & k) == 0;
            bool keep_first = (lower == desc);
            bool mine_first = (skey > ok) || (skey == ok && id < oi);
            bool take_mine = (keep_first == mine_first);
            skey = take_mine ? skey : ok;
            id = take_mine ? id : oi;
        }
    }
}

// ---- the operand-rounding share of a key's error (approximate operands: the scan sees the bf16 roundings x^ = x + dx, y^ = y + dy
// of fp32 data) ----
//   x^.y^ - x.y = dx.y^ + x.dy   =>   |x^.y^ - x.y| <= |dx| |y^| + |x| |dy| <= |dx| (1 + 2^-8) max|y| + |x| max|dy|
// A priori |dx| <= 2^-8 |x| elementwise, which gives eps_round |x| max|y| with eps_round = 2^-7 (1 + 2^-8) -- rounds 4's bound.
// Round 5 measures the norms instead: |dx|^2 per query comes out of the statistics pass that rounds the queries anyway, max|dy|^2
// out of the passes over the corpus blocks at add time.  On real-valued data a rounding error is uniform inside its half-ulp, so
// the norms are ~ 0.29 of their a-priori bound and the listing slack and the certificate's epsilon shrink 3.4 x; the bound stays
// rigorous (Cauchy-Schwarz on the measured vectors; 1.001 covers the fp32 accumulation of the two sums of squares).  L2: the key
// is 2 x.y - |y|^2 with |y|^2 from the exact rows: twice the product's error.  qerr2 == nullptr: the a-priori bound.
__device__ __forceinline__ double round_term(float eps_round, float bx, float xn2, const float* qerr2, int64_t q, float ymax_norm2,
                                             float yerr2_max, bool l2) {
    if (eps_round == 0.f) return 0.0;
    const double apriori = (double)eps_round * (double)bx;
    if (!qerr2) return apriori;
    const double e = sqrt((double)qerr2[q]) * sqrt((double)ymax_norm2) * (1.0 + 1.0 / 256.0) + sqrt((double)xn2) * sqrt((double)yerr2_max);
    const double measured = (l2 ? 2.0 : 1.0) * e * 1.001;
    return measured < apriori ? measured : apriori;
}

// ---- hostile rows ----------------------------------------------------------------------------
// A row (of the corpus or of the queries) whose fp32 |row|^2 is not a number <= 1e36 -- a NaN or +-inf component, or magnitudes
// whose products leave float32 -- has no usable approximate key: the MFMA sums overflow to +-inf and inf - inf to NaN.  Such rows
// are kept out of the statistics every bound is made of (knn_prep.hip: row_stats_kernel) and out of the scan: a corpus row's
// operand is zeroed and its L2 bias -inf (sanitize_hostile_kernel), its exact values stay in the f32 copy, and its canonical
// fp64 score -- which may well be finite, or +-inf, and then ranks -- reaches the answer through merge_special_kernel
// (knn_select.hip), which folds the index's hostile rows into every query's result at the end of a search.  A hostile QUERY
// is never certified and so takes the exact fp64 scan.  Below the limit |x||y| <= 1e36 < FLT_MAX: no key overflows.
constexpr float HOSTILE_NORM2 = 1e36f;
__host__ __device__ __forceinline__ bool hostile_norm2(float n2) { return !(n2 <= HOSTILE_NORM2); }
__device__ __forceinline__ bool in_sorted_ids(const int* ids, int n, u32 id) {      // binary search; n == 0: no
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((u32)ids[mid] < id) lo = mid + 1; else hi = mid; }
    return lo < n && (u32)ids[lo] == id;
}
constexpr int MAX_SPECIAL = 1024;      // hostile corpus rows an index folds in per query; beyond that every search is the exact scan

// ---- parameters ----------------------------------------------------------------------------
struct ScanParams {
    const bf16_t* corpus;    // [n_pad + TILE_M][Kp] bf16, n_pad multiple of TILE_M, pad rows zero
    const bf16_t* queries;   // [q_pad][Kp] bf16, q_pad multiple of TILE_N, pad rows zero
    const float* cbias;      // L2: -|y|^2 per corpus row (pad rows -inf), [n_pad + TILE_M]; IP: unused
    int Kp;                  // padded contraction length, multiple of 2 * BK
    int n_valid;             // corpus rows that exist
    int ntiles;              // n_pad / TILE_M
    int tiles_per_split;
    int nsplits;
    int nqtiles;             // q_pad / TILE_N
    int nq_valid;            // queries that exist: the zero pad rows of the last query tile are never listed (see knn_scan.hip)
    const int* nq_valid_dev; // optional: the same count in device memory (the re-scan of uncertified queries: knn_api.hip); workgroups
                             // whose query tile lies beyond it leave at once
    int fixed_thr;           // 1: thresholds stay at their seeds from g_thr (re-scan: every row that reaches the seed is listed)
    int kprime;              // 16 or 32: rows behind a query's threshold (8 per tracked maximum of a lane)
    int cap;                 // usable slots of a list, 63 or 127 (counter = 7 bits of a packed register)
    int cap_alloc;           // slots allocated per list (cap + 1)
    u64* cand;               // [q_pad][nsplits][8][cap_alloc] packed (key,id), append order
    u32* cand_cnt;           // [q_pad][nsplits][8]
    u64* cand_thr;           // [q_pad][nsplits][8] every unlisted row of the list's rows has comp <= this
    u32* g_thr;              // [q_pad / 256][4 slots][256] ordkeys; the smallest of a query's 4 slots is a key that at least kprime corpus rows reach; shared by all
                             // workgroups of a query (atomicMax, monotone; a stale read is only looser)
    const float* slack;      // optional [q_pad]: a query's listing threshold = the bound its tracked maxima give minus this (accumulator
                             // units; approximate operands: every row within twice the key error of the bound is listed)
    const int* gate;         // optional DEVICE int: the launch runs only when *gate == gate_want (the bf16 / int8 pair of the exact class)
    int gate_want;
    int i8;                  // 1: the int8 form (corpus / queries hold int8 rows of 2 Kp bytes, cbias int32 -|y|^2, L2 queries doubled);
                             // 2: the fp4 form (rows of 2 Kp bytes = 4 Kp components; bias and arithmetic of the bf16 form)
    int bootstrap;           // 1: threshold bootstrap launch (boot_tiles tiles per query tile, publish g_thr only)
    int boot_tiles;
    int debug;               // timing-only diagnostics (TRX_SCAN_DEBUG), 0 in production
    unsigned long long* stamp_out;  // diagnostic build (-DTRX_STAMP_BUILD) only
};

struct SelectParams {
    const u64* cand;
    const u32* cand_cnt;
    const u64* cand_thr;
    int nlists;               // lists per query = corpus splits x LISTS_PER_SPLIT
    int cap_alloc;            // slots allocated per list
    const void* corpus_orig;  // exact values: bf16 [.. ][ld_c] or f32 [..][ld_c]
    int64_t ld_c;             // row stride in elements
    const void* query_orig;   // exact query values, bf16 or f32, row stride ld_q
    int64_t ld_q;
    int corpus_is_bf16;
    int query_is_bf16;
    int d;
    int metric;
    int k;
    int nq;
    int64_t n;                // corpus rows: ids >= n are pad rows of the last tile (an inner-product scan may list them)
    const int* exact_class;   // DEVICE int: 1 = no certificate needed (integer inputs, every partial sum exact)
    float eps_rel;            // certificate slack, relative to bound_q
    float eps_round;          // approximate operands (knn_api.hip, approx mode): their rounding's share of the key error, relative to |x||y| (L2: 2|x||y|)
    const float* qerr2;       // optional [nq]: |x_q - bf16(x_q)|^2, the rounding error norm of each query (0 for bf16 queries); with
    float yerr2_max;          // max over the corpus of |y - bf16(y)|^2 it turns the a-priori bound into round_term()'s measured one
    const float* qnorm2;      // fp32 |x_q|^2 (upper-bound use only)
    float ymax_norm2;         // max |y|^2 over the corpus
    float* D;
    int64_t* I;
    double* S64;              // optional fp64 scores [nq][k] (sharded merge), may be null
    int* flagged;             // queries that could not be certified
    int* nflagged;
    float* flag_seed;         // [nq] per query: a key every row that can still reach the query's top k exceeds (see knn_api.hip, tier 3)
    int compact;              // 1: the lists are indexed by the position in the flagged list (re-scan), not by the query number
    const int* gate;          // optional DEVICE int: the wide re-score runs only when *gate == gate_want (the two list layouts of the
    int gate_want;            // re-scan tier: few queries x many corpus splits, or many queries x the search's own splits)
    const int* special;       // the index's hostile rows (sorted ids; knn_common.h): never candidates here -- an inner-product scan may list their
    int nspecial;             // zeroed operand rows (key 0) -- they are folded in by merge_special_kernel
    int extrap;               // wide re-score of the two-scan path (k > TRX_FAST_MAX_K): a query with fewer than k rows above its guessed
                              // threshold gets a lower one, extrapolated from the rows it did find (seed_out), for one more scan
};

// launchers implemented in the .hip files
hipError_t launch_scan(const ScanParams& p, int metric, hipStream_t st);
hipError_t launch_select(const SelectParams& p, hipStream_t st);
hipError_t launch_wide_rescore(const SelectParams& p, const int* flagged, const int* nflagged, const float* seed_in, int* flagged2, int* nflagged2,
                               float* seed_out, hipStream_t st);
// fold the index's hostile rows (ids special[0 .. nspecial)) into the results: D / I / S64 [nq][k], sorted, pads last
hipError_t launch_merge_special(int metric, int corpus_is_bf16, int query_is_bf16, const void* corpus_orig, int64_t ld_c, const void* queries, int64_t ld_q,
                                int d, int64_t nq, int k, const int* special, int nspecial, int64_t n, float* D, int64_t* I, double* S64, hipStream_t st);
hipError_t launch_append_tail(const int* flagged, const int* nflagged, int from, int* out, int* nout, hipStream_t st);
hipError_t launch_gather_rescan(const int* flagged, const int* nflagged, const float* seed, int max_q, const bf16_t* queries, int Kp,
                                bf16_t* qg2, u32* gthr2, int* count_out, int small_q, int* gate_out, hipStream_t st);

}  // namespace trx
