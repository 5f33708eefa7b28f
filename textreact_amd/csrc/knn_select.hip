// knn_select.hip -- everything after the scan kernel: merge the per-split candidate lists of a
// query, re-score the survivors with the canonical fp64 fma chain, order them by the total order
// (score best first, id ascending), certify the answer, and the exact fall-back scan for queries
// that cannot be certified.  Also the cross-shard merge of the row-sharded multi-GPU search.
//
// Canonical score = oracle/flat_knn_ref.c:trxo_score_canonical (same operations, same order):
//   IP: s = fma((double)x[k], (double)y[k], s)      L2: t = (double)x[k]-(double)y[k]; s = fma(t,t,s)
// for k = 0..d-1.  fp64 fma is IEEE on gfx950 (v_fma_f64), so the bits agree with the host.
//
// Replaces: faiss HeapBlockResultHandler::end_multiple / heap_reorder behind
// retrieve/retrieve_faiss.py:71, and gives the (D, I) it returns.
#include "../../include/trx_knn.h"
#include "knn_common.h"
#include <float.h>
#include <algorithm>
#include <cstdlib>
#include <cstdint>

namespace trx {

template <bool BF>
__device__ __forceinline__ double load_as_double(const void* base, int64_t off) {
    if (BF) return (double)bf16_to_f32(reinterpret_cast<const bf16_t*>(base)[off]);
    return (double)reinterpret_cast<const float*>(base)[off];
}

// canonical score of (query row q, corpus row c); one thread walks the whole row in k order.
template <bool L2, bool CBF, bool QBF>
__device__ __forceinline__ double canonical_score(const void* qrow, const void* crow, int d) {
    double s = 0.0;
    // 8 components per step: 16-byte loads for bf16 rows, 2 x 16 bytes for f32 rows -- when both rows start on a dword: a
    // multi-dword load needs that much, and bf16 rows of an odd d start on odd halfwords every other row (device bf16 queries; since
    // round 6 also every integer host array, which reaches the device as int8 and is widened to bf16 rows of d components: the fuzzer
    // found d = 65).  Otherwise component by component: the same chain in the same order, the same bits.
    int k = 0;
    const bool dword_rows = ((reinterpret_cast<uintptr_t>(qrow) | reinterpret_cast<uintptr_t>(crow)) & 3u) == 0;
    for (; dword_rows && k + 8 <= d; k += 8) {
        double x[8], y[8];
        if (QBF) {
            uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(qrow) + k);
            const u32 w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                x[2 * i] = (double)__uint_as_float(w[i] << 16);
                x[2 * i + 1] = (double)__uint_as_float(w[i] & 0xffff0000u);
            }
        } else {
            const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(qrow) + k);
            const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(qrow) + k + 4);
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        }
        if (CBF) {
            uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(crow) + k);
            const u32 w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                y[2 * i] = (double)__uint_as_float(w[i] << 16);
                y[2 * i + 1] = (double)__uint_as_float(w[i] & 0xffff0000u);
            }
        } else {
            const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(crow) + k);
            const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(crow) + k + 4);
            y[0] = a.x; y[1] = a.y; y[2] = a.z; y[3] = a.w; y[4] = b.x; y[5] = b.y; y[6] = b.z; y[7] = b.w;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (L2) { const double t = x[i] - y[i]; s = __builtin_fma(t, t, s); }
            else s = __builtin_fma(x[i], y[i], s);
        }
    }
    for (; k < d; ++k) {
        const double x = load_as_double<QBF>(qrow, k), y = load_as_double<CBF>(crow, k);
        if (L2) { const double t = x - y; s = __builtin_fma(t, t, s); }
        else s = __builtin_fma(x, y, s);
    }
    return s;
}

#ifndef TRX_SEL_ABL
#define TRX_SEL_ABL 0   // diagnostic builds: 1 = candidates are not re-scored, 2 = lists are not read (results are wrong)
#endif
// ------------------------------------------------------------------------------------------
// select: one wave per PAIR of queries.  The lists of the two queries are merged one after the other, wave-wide; the
// survivors of the first move to lanes 32..63, those of the second stay in lanes 0..31, and the re-scoring -- the bulk of
// the kernel's instructions, one lane per candidate -- runs once for both, as do the final sort (two 32-lane sorts side by
// side) and the certificate.  (One wave per query left 50+ lanes of every fp64 instruction idle: ~11 candidates survive
// the epsilon window.  The kernel is VALU-issue bound: 5.4 k vector instructions per query before, 2.7 k now:
// profiles/r02_select_pmc.json.)
template <bool L2, bool CBF, bool QBF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void knn_select_kernel(SelectParams p) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, hl = lane & 31, half = lane >> 5;
    const int exact_class = *p.exact_class;         // decided on the device from the query statistics (knn_prep.hip: classify_kernel)
    const int q0 = (blockIdx.x * 4 + wv) * 2;       // q0 -> lanes 32..63, q0 + 1 -> lanes 0..31
    if (q0 >= p.nq) return;                         // wave-uniform; the kernel has no workgroup barrier
    // LDS per wave: the row slices of the re-scoring (64 rows x 144 bytes); the merge phase, over before the re-scoring
    // starts, keeps its queue (128 entries) and the lists' prefix sums in the same bytes.  38.9 KB per workgroup: 4 per CU
    __shared__ __attribute__((aligned(16))) char sel_lds[4][64 * (128 + 16)];
    u64* qu = reinterpret_cast<u64*>(sel_lds[wv]);
    int* lpre = reinterpret_cast<int*>(sel_lds[wv] + 128 * 8);
    u64 vA = 0ull, tauA = 0ull, vB = 0ull, tauB = 0ull;
    int nkeepA = 0, nkeepB = 0;
    double epsA = 0.0, epsB = 0.0;

#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        const int q = q0 + h;
        if (q >= p.nq) break;
        // ---- 1. merge the lists of the query by packed approximate (key,id); keep the best KEEP ----
        // A query has nlists lists (corpus splits x wave rows of the scan workgroup), each with its own bound: every row of
        // the list's rows that is NOT listed has a packed value <= bound.  T = the largest bound: every row with a packed
        // value > T is listed somewhere, and at least kprime >= k listed entries reach T (the bound is either a key that
        // kprime rows reach, all of them listed, or the kprime-th entry of a compacted list).  So entries below T are
        // dropped unread by the sort (the lists are append-only logs: most of their entries date from before the
        // thresholds tightened), and T is the starting value of tau, the best packed value of anything not kept.
        u64 tau = 0ull;
        for (int l = lane; l < p.nlists; l += 64) {
            const u64 b = p.cand_thr[(int64_t)q * p.nlists + l];
            tau = b > tau ? b : tau;
        }
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) { const u64 o = shfl_xor_u64(tau, s); tau = o > tau ? o : tau; }
        const u64 T = tau;
        int qn = 0;                        // entries waiting in the queue (wave-uniform), at most 128
        bool fresh = true;                 // nothing sorted yet: the first sort takes 64 entries, one per lane
        u64 v = 0ull;                      // lanes 0..31: running best-32, sorted; lanes 32..63: incoming
        auto drain = [&](bool all) {
            while (qn > 64 || (all && qn > 0)) {
                const int room = fresh ? 64 : 32, l0_ = fresh ? 0 : 32;
                const int take = qn < room ? qn : room;
                __builtin_amdgcn_wave_barrier();
                if (lane >= l0_) v = (lane - l0_) < take ? qu[qn - take + (lane - l0_)] : 0ull;
                fresh = false;
                qn -= take;
                v = wave_sort_desc(v, lane);
                // a row may have been listed twice (the scan repeats a column after compacting a full list): equal packed
                // values are adjacent now; keep the first of each run
                const u64 prev = shfl_u64(v, lane > 0 ? lane - 1 : 0);
                const bool dup = lane > 0 && v != 0ull && v == prev;
                if (__any(dup)) {
                    if (dup) v = 0ull;
                    v = wave_sort_desc(v, lane);
                }
                const u64 dropped = shfl_u64(v, KEEP);   // best of the lanes about to be replaced
                tau = dropped > tau ? dropped : tau;
            }
        };
        auto take = [&](u64 e) {
            const bool keep = e != 0ull && e >= T && (int64_t)comp_id(e) < p.n && !(p.nspecial && in_sorted_ids(p.special, p.nspecial, comp_id(e)));
            const u64 km = __ballot(keep);
            if (keep) qu[qn + __popcll(km & ((1ull << lane) - 1ull))] = e;
            qn += __popcll(km);
            __builtin_amdgcn_wave_barrier();
            drain(false);
        };
        // The lists are short (a lane of the scan kernel lists ~20 rows per split on random data) and there are many of
        // them (8 per split).  64 lists at a time: their counts in one load (a lane per list), an exclusive prefix sum in
        // LDS, and then the entries of all of them as ONE flat sequence, 64 per load (a lane finds its list by bisection of
        // the prefix sums), four loads in flight.  A load per list cost as many vector instructions for 18 entries as this
        // does for 64.
#if TRX_SEL_ABL == 2
        if (0)
#endif
        for (int l0 = 0; l0 < p.nlists; l0 += 64) {
            const int64_t o0 = (int64_t)q * p.nlists + l0;
            const int mycnt = (l0 + lane) < p.nlists ? (int)p.cand_cnt[o0 + lane] : 0;
            int pre = mycnt;
#pragma unroll
            for (int s_ = 1; s_ < 64; s_ <<= 1) { const int t_ = __shfl_up(pre, s_, 64); if (lane >= s_) pre += t_; }
            const int total = __shfl(pre, 63, 64);
            __builtin_amdgcn_wave_barrier();
            lpre[lane] = pre - mycnt;
            __builtin_amdgcn_wave_barrier();
            for (int base = 0; base < total; base += 256) {
                u64 e[4];
#pragma unroll
                for (int r_ = 0; r_ < 4; ++r_) {
                    const int i = base + r_ * 64 + lane;
                    int l = 0;
#pragma unroll
                    for (int step = 32; step; step >>= 1) { if (lpre[l + step] <= i) l += step; }   // last list starting at or before i
                    e[r_] = i < total ? p.cand[(o0 + l) * p.cap_alloc + (i - lpre[l])] : 0ull;
                }
#pragma unroll
                for (int r_ = 0; r_ < 4; ++r_) { if (base + r_ * 64 < total) take(e[r_]); }
            }
        }
        drain(true);
        if (lane >= KEEP) v = 0ull;
        // error bound of an approximate key (the same one the certificate uses below)
        const float xn2 = p.qnorm2[q];
        const float bq = L2 ? (2.0f * sqrtf(xn2 * p.ymax_norm2) + p.ymax_norm2) : sqrtf(xn2 * p.ymax_norm2);
        const float bx = L2 ? 2.0f * sqrtf(xn2 * p.ymax_norm2) : sqrtf(xn2 * p.ymax_norm2);      // what operand rounding scales with: the product term alone
        const double eps = ((double)p.eps_rel * (double)bq + round_term(p.eps_round, bx, xn2, p.qerr2, q, p.ymax_norm2, p.yerr2_max, L2)) * 1.0001 + 1e-30;
        // ---- 1b. epsilon window: |approximate - exact| <= eps for every row, so the k candidates with the best
        // approximate keys all have exact keys >= a_k - eps (a_k = the k-th best approximate key); a candidate whose
        // approximate key is below a_k - 2 eps has an exact key < a_k - eps and cannot reach the top k: it is not
        // re-scored (its row is never fetched) and counts as dropped.  The lanes are sorted, so the pruned ones
        // form a suffix; a non-finite eps prunes nothing.
        int nkeep = KEEP;
        if (p.k <= KEEP) {
            const u64 vk = shfl_u64(v, p.k - 1);
            const bool prune = v != 0ull && lane >= p.k && vk != 0ull && ((double)comp_key(v) + eps < (double)comp_key(vk) - eps);
            const u64 pm = __ballot(prune);
            if (pm) {
                nkeep = __ffsll((long long)pm) - 1;
                const u64 dropped = shfl_u64(v, nkeep);
                tau = dropped > tau ? dropped : tau;
                if (lane >= nkeep) v = 0ull;
            }
        }
        if (h == 0) { vA = shfl_u64(v, hl); tauA = tau; nkeepA = nkeep; epsA = eps; }
        else { vB = v; tauB = tau; nkeepB = nkeep; epsB = eps; }
    }
    // from here on a lane works for the query of its half
    const int q = half ? q0 : q0 + 1;
    const bool live = q < p.nq;
    const u64 v = half ? vA : (lane < KEEP ? vB : 0ull);
    const u64 tau = half ? tauA : tauB;
    const double eps = half ? epsA : epsB;
    const bool have = v != 0ull;
    const u32 id = have ? comp_id(v) : 0xffffffffu;

    // ---- 2. canonical fp64 score of each kept candidate (one lane per candidate, k ascending) ----
    // The candidate rows are brought in slices of 128 bytes by ALL 64 lanes with coalesced 16-byte loads (8 lanes per row
    // slice), staged in this wave's LDS region, and each scoring lane then walks its own row slice out of LDS.  (One lane
    // streaming its own row from global memory re-fetched every 128-byte line 8 times.)  The loads of slice s + 1 are
    // issued as soon as slice s has left the registers for LDS, before it is scored.  The query slices take the same route (a lane per component, read back as LDS
    // broadcasts of four fp32 values) instead of same-address global loads.  The summation order is unchanged:
    // k = 0 .. d-1 per candidate.
    constexpr int QE = QBF ? 2 : 4, CE = CBF ? 2 : 4;   // element bytes
    const char* qrowA = reinterpret_cast<const char*>(p.query_orig) + (int64_t)q0 * p.ld_q * QE;
    const char* qrowB = q0 + 1 < p.nq ? qrowA + (int64_t)p.ld_q * QE : qrowA;
    const char* qrow = half ? qrowA : qrowB;
    constexpr int SLICE = 128 / CE;                 // components per slice (64 bf16 or 32 f32)
    constexpr int ROWB = 128 + 16;                  // LDS bytes per row slice (+16: bank spread)
    constexpr int LPR = 8;                          // lanes per row slice
    constexpr int PASSES = 8;                       // 64 rows, 8 per pass
    __shared__ __attribute__((aligned(16))) float sel_xq[4][2][64];
    char* wl = sel_lds[wv];
    float* xq = sel_xq[wv][half];
    double sc = 0.0, xx = 0.0;
    const bool vec_ok = (p.d % SLICE == 0) && ((p.ld_c * CE) % 16 == 0);
    // integer inputs whose keys are exact (classify_kernel, flag [2]): the key of a candidate IS its canonical score -- x.y, or
    // |x|^2 - dist with every term an integer below 2^24, the value the fp64 fma chain arrives at as well -- so no row is
    // fetched (32 rows of 2 - 4 KB per query: 0.45 of the 0.66 ms this kernel took per 65,536 fingerprint queries)
    const bool keys_are_scores = p.exact_class[2] != 0 && TRX_SEL_ABL == 0;
    if (keys_are_scores) {
        if (have) { const double key = (double)comp_key(v); sc = L2 ? (double)p.qnorm2[q] - key : key; }
    } else
#if TRX_SEL_ABL == 1
    if (0) {
#else
    if (vec_ok) {
#endif
        struct Stage { uint4 r[PASSES]; float x[2]; };
        Stage st0;
        const int nsl = p.d / SLICE;
        u32 rid[PASSES];
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) rid[ps] = __shfl(id, ps * 8 + lane / LPR, 64);
        const int part = lane % LPR;
        auto xload = [&](const char* row, int k) {
            return QBF ? __uint_as_float((u32)reinterpret_cast<const unsigned short*>(row)[k] << 16) : reinterpret_cast<const float*>(row)[k];
        };
        auto fetch = [&](Stage& st, int sl) {
            if (sl >= nsl) return;
            const int k0 = sl * SLICE;
            if (SLICE == 64) { st.x[0] = xload(qrowB, k0 + lane); st.x[1] = xload(qrowA, k0 + lane); }
            else st.x[0] = xload(qrow, k0 + hl);
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                st.r[ps] = make_uint4(0, 0, 0, 0);
                if (rid[ps] != 0xffffffffu)
                    st.r[ps] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.corpus_orig) +
                                                               ((int64_t)rid[ps] * p.ld_c + k0) * CE + part * 16);
            }
        };
        auto score = [&](Stage& st, int sl) {
            if (sl >= nsl) return;
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                if ((ps & 3) * 8 >= ((ps >> 2) ? nkeepA : nkeepB)) continue;   // wave-uniform: rows of this pass all pruned
                *reinterpret_cast<uint4*>(wl + (ps * 8 + lane / LPR) * ROWB + part * 16) = st.r[ps];
            }
            if (SLICE == 64) { sel_xq[wv][0][lane] = st.x[0]; sel_xq[wv][1][lane] = st.x[1]; }
            else sel_xq[wv][half][hl] = st.x[0];
            __builtin_amdgcn_wave_barrier();
            fetch(st, sl + 1);
            if (have) {
                const char* rp = wl + lane * ROWB;
#pragma unroll 2
                for (int c = 0; c < LPR; ++c) {     // 16 bytes of the row slice at a time, k ascending
                    const uint4 u = *reinterpret_cast<const uint4*>(rp + c * 16);
                    const u32 w[4] = {u.x, u.y, u.z, u.w};
                    constexpr int NE = 16 / CE;     // 8 bf16 or 4 f32
                    double y[NE];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (CBF) { y[2 * i] = (double)__uint_as_float(w[i] << 16); y[2 * i + 1] = (double)__uint_as_float(w[i] & 0xffff0000u); }
                        else y[i] = (double)__uint_as_float(w[i]);
                    }
#pragma unroll
                    for (int i = 0; i < NE; ++i) {
                        const double x = (double)xq[c * NE + i];
                        if (L2) { const double t = x - y[i]; sc = __builtin_fma(t, t, sc); xx = __builtin_fma(x, x, xx); }
                        else sc = __builtin_fma(x, y[i], sc);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        };
        fetch(st0, 0);
        __builtin_amdgcn_wave_barrier();     // the merge phase's LDS reads are done
        for (int sl = 0; sl < nsl; ++sl) score(st0, sl);
    } else if (have && TRX_SEL_ABL != 1) {
        const char* crow = reinterpret_cast<const char*>(p.corpus_orig) + (int64_t)id * p.ld_c * CE;
        sc = canonical_score<L2, CBF, QBF>(qrow, crow, p.d);
        if (L2 && !exact_class)
            for (int i = 0; i < p.d; ++i) {     // |x|^2 in fp64, k-ordered (the certificate's key = |x|^2 - dist)
                const double t = load_as_double<QBF>(qrow, i);
                xx = __builtin_fma(t, t, xx);
            }
    }
    // NaN scores never rank (oracle: skipped)
    const bool ranked = have && (sc == sc);
    // sort key: larger == earlier.  IP: score itself; L2: negated.
    u64 skey = ranked ? orddbl(L2 ? -sc : sc) : 0ull;
    u32 sid = ranked ? id : 0xffffffffu;
    // carry the score along by re-deriving it from skey after the sort (orddbl is a bijection)
    wave_sort_pairs<32>(skey, sid, lane);
    const u64 rm = __ballot(skey != 0ull);
    const int nranked = __popc(half ? (u32)(rm >> 32) : (u32)rm);

    // decode score
    double ssc;
    {
        const u64 u = (skey >> 63) ? (skey & 0x7fffffffffffffffull) : ~skey;
        const double dv = __longlong_as_double((long long)u);
        ssc = L2 ? -dv : dv;
    }

    // ---- 3. certificate: can any row outside the kept set reach the k-th place? ----
    bool certified = true;
    {
        const int ksrc = (lane & 32) + (p.k <= 32 ? p.k - 1 : 31);
        const double s_k = __shfl(ssc, ksrc, 64);          // exact score in k-th place
        const double xx0 = __shfl(xx, lane & 32, 64);      // the first candidate's lane has walked the whole query row
        // a key that every row of the true top k exceeds by more than eps: the exact k-th score of these candidates is a lower
        // bound of the true k-th, so a row that can still matter has an approximate key above (that score - eps) > seed
        // (written for every query, by query number: only a flagged query's is ever read -- tier 3 of the fall-back re-scans
        // with it, knn_api.hip -- and nothing stays alive for the rare branch below)
        if (p.flag_seed && hl == 0 && live)
            p.flag_seed[q] = (nranked >= p.k && !exact_class) ? __double2float_rd((L2 ? xx0 - s_k : s_k) - 2.0 * eps) : -FLT_MAX;
        if (tau != 0ull && nranked < p.k) {
            certified = false;      // fewer ranked candidates than k although rows were dropped: exact inputs or not, redo it
        } else if (!exact_class && tau != 0ull) {
            const float tau_key = comp_key(tau);
            const double bound = (double)tau_key + eps;  // upper bound of an outsider's exact key
            certified = L2 ? (xx0 - s_k) > bound : s_k > bound;   // L2: key = |x|^2 - dist
        }
    }

    // ---- 4. write D, I (and fp64 scores for the sharded merge) ----
    if (hl < p.k && live) {
        const bool ok = hl < nranked;
        const int64_t o = (int64_t)q * p.k + hl;
        p.D[o] = ok ? (float)ssc : (L2 ? FLT_MAX : -FLT_MAX);
        p.I[o] = ok ? (int64_t)sid : (int64_t)-1;
        if (p.S64) p.S64[o] = ok ? ssc : (L2 ? (double)FLT_MAX : -(double)FLT_MAX);
    }
    if (live && hostile_norm2(p.qnorm2[q])) certified = false;      // a hostile query (knn_common.h): its keys mean nothing -> the exact scan
    if (!certified && hl == 0 && live && TRX_SEL_ABL == 0) {
        const int pos = atomicAdd(p.nflagged, 1);
        p.flagged[pos] = q;
    }
}

hipError_t launch_select(const SelectParams& p, hipStream_t st) {
    dim3 grid((p.nq + 7) / 8), block(256);
    if (p.nq <= 0) return hipSuccess;
    const int sel = (p.metric ? 4 : 0) | (p.corpus_is_bf16 ? 2 : 0) | (p.query_is_bf16 ? 1 : 0);
    switch (sel) {
        case 0: hipLaunchKernelGGL((knn_select_kernel<false, false, false>), grid, block, 0, st, p); break;
        case 1: hipLaunchKernelGGL((knn_select_kernel<false, false, true>), grid, block, 0, st, p); break;
        case 2: hipLaunchKernelGGL((knn_select_kernel<false, true, false>), grid, block, 0, st, p); break;
        case 3: hipLaunchKernelGGL((knn_select_kernel<false, true, true>), grid, block, 0, st, p); break;
        case 4: hipLaunchKernelGGL((knn_select_kernel<true, false, false>), grid, block, 0, st, p); break;
        case 5: hipLaunchKernelGGL((knn_select_kernel<true, false, true>), grid, block, 0, st, p); break;
        case 6: hipLaunchKernelGGL((knn_select_kernel<true, true, false>), grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL((knn_select_kernel<true, true, true>), grid, block, 0, st, p); break;
    }
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------
// merge_special: the hostile rows of the index (knn_common.h) into every query's result.  The scan never lists them (zero operand,
// -inf bias) and their approximate keys would mean nothing; their canonical fp64 scores do -- finite (|values| ~ 1e30: the fp32
// sums overflow, the fp64 chain does not), +-inf (an inf component), or NaN (never ranks) -- and the oracle ranks them like any
// other row.  One workgroup per query: A = the result so far without special ids (scores recomputed by the same chain: the same
// bits), B = the special rows with a score, ordered by rank counting; the two sorted lists are merged by binary search (ids are
// distinct, so the total order is strict) and the first k are written back.  Idempotent: results that already hold special rows
// (the exact fp64 scan sees the original rows) come out as they went in.  Runs only when the index has such rows.
template <bool L2, bool CBF, bool QBF>
__global__ __launch_bounds__(256) void merge_special_kernel(const void* corpus, int64_t ld_c, const void* queries, int64_t ld_q, int d, int64_t nq, int k,
                                                            const int* special, int ns, int64_t n, float* D, int64_t* I, double* S64) {
    __shared__ u64 a_key[TRX_MAX_K]; __shared__ u32 a_id[TRX_MAX_K];
    __shared__ u64 b_key[MAX_SPECIAL]; __shared__ u32 b_id[MAX_SPECIAL];
    __shared__ u64 c_key[MAX_SPECIAL]; __shared__ u32 c_id[MAX_SPECIAL];
    __shared__ int cnt[257];
    __shared__ int nb_s;
    const int tid = threadIdx.x;
    constexpr int CE = CBF ? 2 : 4, QE = QBF ? 2 : 4;
    auto is_special = [&](u32 id) { return in_sorted_ids(special, ns, id); };      // (`special` is sorted ascending)
    auto before = [](u64 ka, u32 ia, u64 kb, u32 ib) { return ka > kb || (ka == kb && ia < ib); };      // larger key = ranked earlier
    for (int64_t q = blockIdx.x; q < nq; q += gridDim.x) {
        const char* qrow = reinterpret_cast<const char*>(queries) + q * ld_q * QE;
        __syncthreads();
        if (tid == 0) nb_s = 0;
        // ---- A: the entries that stay, in order (a thread owns a contiguous run so that the compaction keeps the order) ----
        const int per = (k + 255) / 256, t0 = tid * per, t1 = min(k, t0 + per);
        int keep = 0;
        for (int t = t0; t < t1; ++t) { const int64_t id = I[q * k + t]; keep += (id >= 0 && id < n && !is_special((u32)id)) ? 1 : 0; }
        cnt[tid + 1] = keep;
        __syncthreads();
        if (tid == 0) { cnt[0] = 0; for (int i = 1; i <= 256; ++i) cnt[i] += cnt[i - 1]; }
        __syncthreads();
        const int na = cnt[256];
        {
            int o = cnt[tid];
            for (int t = t0; t < t1; ++t) {
                const int64_t id = I[q * k + t];
                if (id < 0 || id >= n || is_special((u32)id)) continue;      // (id >= n: a query the late fall-back has yet to write)
                const double sc = canonical_score<L2, CBF, QBF>(qrow, reinterpret_cast<const char*>(corpus) + id * ld_c * CE, d);
                a_key[o] = orddbl(L2 ? -sc : sc); a_id[o] = (u32)id; ++o;
            }
        }
        // ---- B: the special rows that have a score ----
        for (int s_ = tid; s_ < ns; s_ += 256) {
            const u32 id = (u32)special[s_];
            const double sc = canonical_score<L2, CBF, QBF>(qrow, reinterpret_cast<const char*>(corpus) + (int64_t)id * ld_c * CE, d);
            if (sc == sc) { const int o = atomicAdd(&nb_s, 1); b_key[o] = orddbl(L2 ? -sc : sc); b_id[o] = id; }
        }
        __syncthreads();
        const int nb = nb_s;
        for (int i = tid; i < nb; i += 256) {        // rank counting: nb is small
            int r = 0;
            for (int j = 0; j < nb; ++j) r += before(b_key[j], b_id[j], b_key[i], b_id[i]) ? 1 : 0;
            c_key[r] = b_key[i]; c_id[r] = b_id[i];
        }
        __syncthreads();
        // ---- merge: the place of an element = its place in its own list + the elements of the other list before it ----
        auto put = [&](int pos, u64 key, u32 id) {
            if (pos >= k) return;
            const u64 u = (key >> 63) ? (key & 0x7fffffffffffffffull) : ~key;      // orddbl's inverse
            const double dv = __longlong_as_double((long long)u), sc = L2 ? -dv : dv;
            const int64_t o = q * k + pos;
            D[o] = (float)sc; I[o] = (int64_t)id;
            if (S64) S64[o] = sc;
        };
        for (int i = tid; i < na; i += 256) {
            int lo = 0, hi = nb;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (before(c_key[mid], c_id[mid], a_key[i], a_id[i])) lo = mid + 1; else hi = mid; }
            put(i + lo, a_key[i], a_id[i]);
        }
        for (int j = tid; j < nb; j += 256) {
            int lo = 0, hi = na;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (before(a_key[mid], a_id[mid], c_key[j], c_id[j])) lo = mid + 1; else hi = mid; }
            put(j + lo, c_key[j], c_id[j]);
        }
        for (int pos = na + nb + tid; pos < k; pos += 256) {
            const int64_t o = q * k + pos;
            D[o] = L2 ? FLT_MAX : -FLT_MAX; I[o] = -1;
            if (S64) S64[o] = L2 ? (double)FLT_MAX : -(double)FLT_MAX;
        }
    }
}
hipError_t launch_merge_special(int metric, int corpus_is_bf16, int query_is_bf16, const void* corpus_orig, int64_t ld_c, const void* queries, int64_t ld_q,
                                int d, int64_t nq, int k, const int* special, int nspecial, int64_t n, float* D, int64_t* I, double* S64, hipStream_t st) {
    if (nq <= 0 || nspecial <= 0) return hipSuccess;
    const dim3 grid((unsigned)std::min<int64_t>(nq, 65536)), block(256);
#define TRX_MS(a, b, c) hipLaunchKernelGGL((merge_special_kernel<a, b, c>), grid, block, 0, st, corpus_orig, ld_c, queries, ld_q, d, nq, k, special, nspecial, n, D, I, S64)
    const int sel = (metric ? 4 : 0) | (corpus_is_bf16 ? 2 : 0) | (query_is_bf16 ? 1 : 0);
    switch (sel) {
        case 0: TRX_MS(false, false, false); break; case 1: TRX_MS(false, false, true); break;
        case 2: TRX_MS(false, true, false); break;  case 3: TRX_MS(false, true, true); break;
        case 4: TRX_MS(true, false, false); break;  case 5: TRX_MS(true, false, true); break;
        case 6: TRX_MS(true, true, false); break;   default: TRX_MS(true, true, true); break;
    }
#undef TRX_MS
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// wide re-score: the second tier, between the select kernel and the exact scan.  The select kernel re-scores the best KEEP =
// 32 candidates of a query and flags the query when it cannot prove the answer from those: more than KEEP - k rows lie within
// the rounding bound of the k-th score (near-duplicate passages; the nearly collinear [CLS] embeddings of an untrained
// encoder), or fewer than k ranked.  The candidate lists still hold EVERY row whose approximate key exceeds T, the largest
// bound of the query's lists; so one workgroup per flagged query re-scores all listed rows that reach T -- tens to a few
// thousand rows instead of the whole corpus -- ranks them by the canonical fp64 score, and the answer is proven exact when
// the k-th exact score beats T + eps (no unlisted row can reach it).  Only queries that fail THAT go on to the exact scan
// (flagged2 / nflagged2).  One workgroup of 256 threads per query, a thread per candidate row.
// (key descending, id ascending) over the first N (a power of two) entries; 256 threads
__device__ __forceinline__ void bitonic_sort_pairs(u64* key, u32* id, int N, int tid) {
    for (int size = 2; size <= N; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (N >> 1); i += 256) {
                const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
                const u64 ka = key[lo], kb = key[hi];
                const u32 ia = id[lo], ib = id[hi];
                const bool a_first = ka > kb || (ka == kb && ia < ib);
                const bool want_a_first = (lo & size) == 0;
                if (a_first != want_a_first && !(ka == kb && ia == ib)) { key[lo] = kb; key[hi] = ka; id[lo] = ib; id[hi] = ia; }
            }
            __syncthreads();
        }
    }
}

constexpr int WIDE_MAX = 4096;     // listed rows of a query that reach T: at most nlists x 127 = 32 x 127 = 4064 at 4 splits
constexpr int WIDE_K_MAX = TRX_WIDE_MAX_K;      // the largest k it ranks (the two-scan path of TRX_FAST_MAX_K < k <= TRX_WIDE_MAX_K: knn_api.hip)

template <bool L2, bool CBF, bool QBF>
__global__ __launch_bounds__(256) void wide_rescore_kernel(SelectParams p, const int* flagged, const int* nflagged, const float* seed_in,
                                                           int* flagged2, int* nflagged2, float* seed_out) {
    __shared__ u64 w_key[WIDE_MAX];      // orddbl of the canonical score (L2: of its negative): larger = ranked earlier; 0 = not a number
    __shared__ u32 w_id[WIDE_MAX];
    __shared__ u64 red_key[256];
    __shared__ u32 red_id[256];
    __shared__ u64 w_T;
    __shared__ int w_cnt, w_got;
    __shared__ u32 w_okmax;
    __shared__ int w_hist[256];
    __shared__ double pick_s[WIDE_K_MAX];
    __shared__ u32 pick_i[WIDE_K_MAX];
    const int tid = threadIdx.x;
    if (p.gate && *p.gate != p.gate_want) return;
    const int nf = *nflagged;
    const int exact_class = *p.exact_class;
    constexpr int QE = QBF ? 2 : 4, CE = CBF ? 2 : 4;
    for (int f = blockIdx.x; f < nf; f += gridDim.x) {
        const int q = flagged[f];
        const int64_t lq = p.compact ? f : q;        // where this query's lists are: by query number, or by place in the flagged list (re-scan)
        __syncthreads();
        // ---- T = the largest bound of the query's lists ----
        u64 t = 0ull;
        for (int l = tid; l < p.nlists; l += 256) { const u64 b = p.cand_thr[lq * p.nlists + l]; t = b > t ? b : t; }
        red_key[tid] = t;
        if (tid == 0) { w_cnt = 0; w_okmax = 0u; }
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) { if (tid < w && red_key[tid + w] > red_key[tid]) red_key[tid] = red_key[tid + w]; __syncthreads(); }
        if (tid == 0) w_T = red_key[0];
        __syncthreads();
        // ---- every listed row that reaches T (a row listed twice after a compaction appears twice: the ranking below skips it).
        // A wave per list, a lane per entry.  More than WIDE_MAX of them (the two-scan path's guess was low and the query has
        // many lists): T is raised to the lowest of 256 key steps that leaves at most WIDE_MAX -- still a bound of everything
        // left out, so the certificate below stands as it is
        u64 T = w_T;
        bool overflow = false;
        for (int pass = 0; pass < 2; ++pass) {
            for (int l = tid >> 6; l < p.nlists; l += 4) {
                const int64_t o = lq * p.nlists + l;
                const int cnt = (int)p.cand_cnt[o];
                for (int i = tid & 63; i < cnt; i += 64) {
                    const u64 e = p.cand[o * p.cap_alloc + i];
                    if (e != 0ull && e >= T && (int64_t)comp_id(e) < p.n && !(p.nspecial && in_sorted_ids(p.special, p.nspecial, comp_id(e)))) {
                        const int pos = atomicAdd(&w_cnt, 1);
                        if (pos < WIDE_MAX) w_id[pos] = comp_id(e);
                        atomicMax(&w_okmax, (u32)(e >> 32));
                    }
                }
            }
            __syncthreads();
            if (w_cnt <= WIDE_MAX) break;
            if (pass == 1) { overflow = true; break; }
            const u32 ok0 = (u32)(T >> 32);
            const u64 range = (u64)(w_okmax - ok0) + 1ull;
            w_hist[tid] = 0;
            __syncthreads();
            for (int l = tid >> 6; l < p.nlists; l += 4) {
                const int64_t o = lq * p.nlists + l;
                const int cnt = (int)p.cand_cnt[o];
                for (int i = tid & 63; i < cnt; i += 64) {
                    const u64 e = p.cand[o * p.cap_alloc + i];
                    if (e != 0ull && e >= T && (int64_t)comp_id(e) < p.n && !(p.nspecial && in_sorted_ids(p.special, p.nspecial, comp_id(e)))) atomicAdd(&w_hist[(int)((((u64)((u32)(e >> 32) - ok0)) << 8) / range)], 1);
                }
            }
            __syncthreads();
            if (tid == 0) {
                int b = 256, sum = 0;
                while (b > 0 && sum + w_hist[b - 1] <= WIDE_MAX) { sum += w_hist[b - 1]; --b; }
                w_T = (u64)(ok0 + (u32)(((u64)b * range + 255ull) >> 8)) << 32;      // keys of step b and above
                w_cnt = 0;
            }
            __syncthreads();
            T = w_T;
        }
        int C = w_cnt;
        if (C > WIDE_MAX) C = WIDE_MAX;
        // ---- canonical scores, a thread per row ----
        const char* qrow = reinterpret_cast<const char*>(p.query_orig) + (int64_t)q * p.ld_q * QE;
        for (int i = tid; i < C; i += 256) {
            const char* crow = reinterpret_cast<const char*>(p.corpus_orig) + (int64_t)w_id[i] * p.ld_c * CE;
            const double sv = canonical_score<L2, CBF, QBF>(qrow, crow, p.d);
            w_key[i] = sv == sv ? orddbl(L2 ? -sv : sv) : 0ull;
        }
        __syncthreads();
        int got = 0;
        if (p.k > 32) {
            // ---- the two-scan path's k (up to WIDE_K_MAX): sort all C rows once, then the first k distinct ones ----
            int N = 2; while (N < C) N <<= 1;
            for (int i = C + tid; i < N; i += 256) { w_key[i] = 0ull; w_id[i] = 0xffffffffu; }
            __syncthreads();
            bitonic_sort_pairs(w_key, w_id, N, tid);
            if (tid == 0) {
                int g = 0;
                for (int i = 0; i < C && g < p.k; ++i) {
                    const u64 wk = w_key[i];
                    if (wk == 0ull) break;
                    if (i > 0 && wk == w_key[i - 1] && w_id[i] == w_id[i - 1]) continue;      // listed twice
                    const u64 u = (wk >> 63) ? (wk & 0x7fffffffffffffffull) : ~wk;
                    const double dv = __longlong_as_double((long long)u);
                    pick_s[g] = L2 ? -dv : dv; pick_i[g] = w_id[i];
                    ++g;
                }
                w_got = g;
            }
            __syncthreads();
            got = w_got;
        } else {
        // ---- the k best by (score, id): k rounds of "next after the previous pick" ----
        u64 prev_key = ~0ull; u32 prev_id = 0u; bool first = true;
        for (int r = 0; r < p.k; ++r) {
            u64 bk = 0ull; u32 bi = 0xffffffffu;
            for (int i = tid; i < C; i += 256) {
                const u64 key = w_key[i];
                if (key == 0ull) continue;
                const u32 id = w_id[i];
                const bool after = first || key < prev_key || (key == prev_key && id > prev_id);
                if (!after) continue;
                if (bi == 0xffffffffu || key > bk || (key == bk && id < bi)) { bk = key; bi = id; }
            }
            red_key[tid] = bk; red_id[tid] = bi;
            __syncthreads();
            for (int w = 128; w > 0; w >>= 1) {
                if (tid < w) {
                    const u64 ok = red_key[tid + w]; const u32 oi = red_id[tid + w];
                    const u64 mk = red_key[tid]; const u32 mi = red_id[tid];
                    if (oi != 0xffffffffu && (mi == 0xffffffffu || ok > mk || (ok == mk && oi < mi))) { red_key[tid] = ok; red_id[tid] = oi; }
                }
                __syncthreads();
            }
            const u64 wk = red_key[0]; const u32 wi = red_id[0];
            __syncthreads();
            if (wi == 0xffffffffu) break;
            if (tid == 0) {
                const u64 u = (wk >> 63) ? (wk & 0x7fffffffffffffffull) : ~wk;
                const double dv = __longlong_as_double((long long)u);
                pick_s[r] = L2 ? -dv : dv; pick_i[r] = wi;
            }
            prev_key = wk; prev_id = wi; first = false;
            ++got;
        }
        }
        __syncthreads();
        // ---- certificate (same bound as the select kernel's) ----
        bool certified;
        if (overflow || hostile_norm2(p.qnorm2[q])) certified = false;  // (a hostile query: knn_common.h)
        else if (T == 0ull) certified = true;                         // nothing was ever dropped: every row is listed
        else if (got < p.k) certified = false;
        else if (exact_class) certified = true;                       // approximate order is the exact order
        else {
            const float xn2 = p.qnorm2[q];
            const float bq = L2 ? (2.0f * sqrtf(xn2 * p.ymax_norm2) + p.ymax_norm2) : sqrtf(xn2 * p.ymax_norm2);
            const float bx = L2 ? 2.0f * sqrtf(xn2 * p.ymax_norm2) : sqrtf(xn2 * p.ymax_norm2);      // what operand rounding scales with: the product term alone
            const double eps = ((double)p.eps_rel * (double)bq + round_term(p.eps_round, bx, xn2, p.qerr2, q, p.ymax_norm2, p.yerr2_max, L2)) * 1.0001 + 1e-30;
            const double bound = (double)comp_key(T) + eps;
            double xx = 0.0;
            if (L2) for (int i = 0; i < p.d; ++i) { const double v = load_as_double<QBF>(qrow, i); xx = __builtin_fma(v, v, xx); }
            const double s_k = pick_s[p.k - 1];
            certified = L2 ? (xx - s_k) > bound : s_k > bound;
        }
        if (certified) {
            for (int r = tid; r < p.k; r += 256) {
                const bool ok = r < got;
                const int64_t o = (int64_t)q * p.k + r;
                p.D[o] = ok ? (float)pick_s[r] : (L2 ? FLT_MAX : -FLT_MAX);
                p.I[o] = ok ? (int64_t)pick_i[r] : (int64_t)-1;
                if (p.S64) p.S64[o] = ok ? pick_s[r] : (L2 ? (double)FLT_MAX : -(double)FLT_MAX);
            }
        } else if (tid == 0) {
            const int pos2 = atomicAdd(nflagged2, 1);
            flagged2[pos2] = q;
            if (seed_out) {     // the threshold tier 3 re-scans with: the select kernel's, raised by what this pass has found
                float sd = seed_in ? seed_in[q] : -FLT_MAX;        // (the select kernel's, by query number)
                const float xn2 = p.qnorm2[q];
                const float bq = L2 ? (2.0f * sqrtf(xn2 * p.ymax_norm2) + p.ymax_norm2) : sqrtf(xn2 * p.ymax_norm2);
                const float bx = L2 ? 2.0f * sqrtf(xn2 * p.ymax_norm2) : sqrtf(xn2 * p.ymax_norm2);      // what operand rounding scales with: the product term alone
                const double eps = ((double)p.eps_rel * (double)bq + round_term(p.eps_round, bx, xn2, p.qerr2, q, p.ymax_norm2, p.yerr2_max, L2)) * 1.0001 + 1e-30;
                double xx = 0.0;
                if (L2) for (int i = 0; i < p.d; ++i) { const double v = load_as_double<QBF>(qrow, i); xx = __builtin_fma(v, v, xx); }
                if (got >= p.k && !exact_class && !overflow) {
                    const float mine = __double2float_rd((L2 ? xx - pick_s[p.k - 1] : pick_s[p.k - 1]) - 2.0 * eps);
                    sd = (mine > sd || p.extrap) ? mine : sd;      // (two-scan path: seed_in was a guess, within eps of the k-th score here)
                } else if (p.extrap && got < p.k && !overflow) {
                    // two-scan path: the guess (seed_in) left fewer than k rows above it.  The `got` rows that are there say how
                    // the scores thin out (bigk_seed_kernel's estimate, from got spacings instead of 23, twice its margin), and
                    // the new threshold lies at least a quarter of the span covered so far below the old one
                    if (got >= 2) {
                        double csum = 0.0;
                        for (int r = 1; r < got; ++r) csum += (double)r * fmax(L2 ? pick_s[r] - pick_s[r - 1] : pick_s[r - 1] - pick_s[r], 0.0);
                        const double c = csum / (double)(got - 1);
                        const double k1 = L2 ? xx - pick_s[0] : pick_s[0], kl = L2 ? xx - pick_s[got - 1] : pick_s[got - 1];
                        const double ext = 2.0 * (1.8 * log((double)p.k / (double)got) + 2.5 * sqrt(1.0 / (double)got - 1.0 / (double)p.k));
                        const double lower = fmin(kl - ext * c, (double)sd - 0.25 * (k1 - (double)sd));
                        sd = lower > -1e38 ? __double2float_rd(lower - 3.0 * eps - 1e-6 * (fabs(kl) + (double)xn2)) : -FLT_MAX;
                    } else sd = -FLT_MAX;
                }
                seed_out[pos2] = sd;
            }
        }
    }
}

hipError_t launch_wide_rescore(const SelectParams& p, const int* flagged, const int* nflagged, const float* seed_in, int* flagged2, int* nflagged2,
                               float* seed_out, hipStream_t st) {
    if (p.nq <= 0) return hipSuccess;
    dim3 grid((unsigned)(p.nq < 2048 ? p.nq : 2048)), block(256);      // blocks beyond the (device-side) count leave at once
    const int sel = (p.metric ? 4 : 0) | (p.corpus_is_bf16 ? 2 : 0) | (p.query_is_bf16 ? 1 : 0);
#define TRX_WR(a, b, c) hipLaunchKernelGGL((wide_rescore_kernel<a, b, c>), grid, block, 0, st, p, flagged, nflagged, seed_in, flagged2, nflagged2, seed_out)
    switch (sel) {
        case 0: TRX_WR(false, false, false); break;
        case 1: TRX_WR(false, false, true); break;
        case 2: TRX_WR(false, true, false); break;
        case 3: TRX_WR(false, true, true); break;
        case 4: TRX_WR(true, false, false); break;
        case 5: TRX_WR(true, false, true); break;
        case 6: TRX_WR(true, true, false); break;
        default: TRX_WR(true, true, true); break;
    }
#undef TRX_WR
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// TRX_FAST_MAX_K < k <= TRX_WIDE_MAX_K (knn_api.hip: the two-scan path): the first scan ranked kf = 24 candidates of every query
// exactly; the threshold of the second scan is a guess at the k-th best key -- the tail of the scores extrapolated from those 24
// with a margin, minus 3 eps -- and only a guess: the wide re-score over the second scan's lists PROVES what it returns (k exact
// scores above the threshold + eps, every row above the threshold listed).  A guess that was too low costs time only (more rows
// to rank; lists that fill up raise their own bounds, and the wide re-score raises T when more than WIDE_MAX rows reach it); a
// guess that was too high (clustered data: the scores fall off a cliff the first 24 know nothing about) leaves fewer than k
// rows, and the wide re-score extrapolates again from all of those for a third scan of that query (SelectParams::extrap).
// Every query is "flagged": the list is the identity.
template <bool L2>
__global__ void bigk_seed_kernel(const float* D1, const int64_t* I1, int nq, int kf, int k, const float* qnorm2, float eps_rel, float eps_round,
                                 float ymax_norm2, const float* qerr2, float yerr2_max, int* list, int* count, float* seed) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q == 0) *count = nq;
    if (q >= nq) return;
    list[q] = q;
    float sd = -FLT_MAX;
    if (I1[(int64_t)q * kf + kf - 1] >= 0) {      // kf ranked rows: their exact scores bound the tail from above
        const float xn2 = qnorm2[q];
        const double k1 = L2 ? (double)xn2 - (double)D1[(int64_t)q * kf] : (double)D1[(int64_t)q * kf];
        const double kl = L2 ? (double)xn2 - (double)D1[(int64_t)q * kf + kf - 1] : (double)D1[(int64_t)q * kf + kf - 1];
        const float bq = L2 ? (2.0f * sqrtf(xn2 * ymax_norm2) + ymax_norm2) : sqrtf(xn2 * ymax_norm2);
        const float bx = L2 ? 2.0f * sqrtf(xn2 * ymax_norm2) : sqrtf(xn2 * ymax_norm2);
        const double eps = ((double)eps_rel * (double)bq + round_term(eps_round, bx, xn2, qerr2, q, ymax_norm2, yerr2_max, L2)) * 1.0001 + 1e-30;
        // the spacings of the order statistics of an exponential tail are exponentials of mean c / r: s_kf - s_k has mean
        // c ln(k / kf) and deviation c sqrt(1 / kf - 1 / k).  c from ALL kf scores (the mean of r (s_r - s_r+1), each an
        // exponential of mean c: a fifth of the noise of the one difference s_1 - s_kf).  1.8 means + 2.5 deviations: on
        // Gaussian, uniform, Laplace and offset data (100,000 rows, k 32 .. 256) about 1 query in 200 is left with fewer than
        // k rows and the median query lists 4 k; on clustered data up to 1 in 5 at k = 256 -- those take the third scan
        double csum = 0.0, prev = k1;
        for (int r = 1; r < kf; ++r) {
            const double cur = L2 ? (double)xn2 - (double)D1[(int64_t)q * kf + r] : (double)D1[(int64_t)q * kf + r];
            csum += (double)r * fmax(prev - cur, 0.0);
            prev = cur;
        }
        const double gap = csum / (double)(kf - 1);      // c
        const double ext = 1.8 * log((double)k / (double)kf) + 2.5 * sqrt(1.0 / (double)kf - 1.0 / (double)k);
        // (L2: |x|^2 - dist is the key up to the rounding of the fp32 distance: one more eps-sized allowance)
        sd = __double2float_rd(kl - ext * gap - 3.0 * eps - 1e-6 * (fabs(kl) + (double)xn2));
    }
    seed[q] = sd;
}
hipError_t launch_bigk_seeds(int metric, const float* D1, const int64_t* I1, int nq, int kf, int k, const float* qnorm2, float eps_rel, float eps_round,
                             float ymax_norm2, const float* qerr2, float yerr2_max, int* list, int* count, float* seed, hipStream_t st) {
    if (nq <= 0) return hipSuccess;
    dim3 grid((unsigned)((nq + 255) / 256)), block(256);
    if (metric) hipLaunchKernelGGL(bigk_seed_kernel<true>, grid, block, 0, st, D1, I1, nq, kf, k, qnorm2, eps_rel, eps_round, ymax_norm2, qerr2, yerr2_max, list, count, seed);
    else hipLaunchKernelGGL(bigk_seed_kernel<false>, grid, block, 0, st, D1, I1, nq, kf, k, qnorm2, eps_rel, eps_round, ymax_norm2, qerr2, yerr2_max, list, count, seed);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// tier 3, step 1: the operand rows of the still-uncertified queries, packed into a compact query matrix for the re-scan, and
// their thresholds in the scan kernel's shared-threshold array (all four slots of a query: the bound stands alone).  Queries
// beyond max_q (more than the re-scan is sized for) stay out: count_out = min(count, max_q), the others keep their place in the
// flagged list and take the exact scan.  One workgroup per query row (Kp / 8 16-byte chunks).
__global__ __launch_bounds__(128) void gather_rescan_kernel(const int* flagged, const int* nflagged, const float* seed, int max_q,
                                                            const bf16_t* queries, int Kp, bf16_t* qg2, u32* gthr2, int* count_out,
                                                            int small_q, int* gate_out) {
    const int n = min(*nflagged, max_q);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *count_out = n;
        if (gate_out) *gate_out = n <= small_q ? 1 : 0;      // few enough for the many-splits form of the re-scan (knn_api.hip)
    }
    for (int f = blockIdx.x; f < n; f += gridDim.x) {
        const int q = flagged[f];
        const uint4* src = reinterpret_cast<const uint4*>(queries + (int64_t)q * Kp);
        uint4* dst = reinterpret_cast<uint4*>(qg2 + (int64_t)f * Kp);
        for (int c = threadIdx.x; c < Kp / 8; c += 128) dst[c] = src[c];
        if (threadIdx.x < 4) gthr2[((int64_t)(f >> 8) * 4 + threadIdx.x) * 256 + (f & 255)] = ordkey(seed[f]);
    }
}
// queries beyond the re-scan's capacity keep their place in line for the exact scan
__global__ void append_tail_kernel(const int* flagged, const int* nflagged, int from, int* out, int* nout) {
    const int n = *nflagged;
    for (int f = from + blockIdx.x * blockDim.x + threadIdx.x; f < n; f += gridDim.x * blockDim.x) out[atomicAdd(nout, 1)] = flagged[f];
}
hipError_t launch_append_tail(const int* flagged, const int* nflagged, int from, int* out, int* nout, hipStream_t st) {
    hipLaunchKernelGGL(append_tail_kernel, dim3(64), dim3(256), 0, st, flagged, nflagged, from, out, nout);
    return hipGetLastError();
}
hipError_t launch_gather_rescan(const int* flagged, const int* nflagged, const float* seed, int max_q, const bf16_t* queries, int Kp,
                                bf16_t* qg2, u32* gthr2, int* count_out, int small_q, int* gate_out, hipStream_t st) {
    hipLaunchKernelGGL(gather_rescan_kernel, dim3(1024), dim3(128), 0, st, flagged, nflagged, seed, max_q, queries, Kp, qg2, gthr2, count_out,
                       small_q, gate_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// exact scan (fall-back and k > TRX_FAST_MAX_K): canonical scores of `nf` listed queries against
// every corpus row, then k rounds of "next best after the previous pick" per query.
// nf_dev (optional): the number of listed queries lives on the device (the certificate failures of a search that has
// not been read back): the launch covers `nf` slots and the blocks beyond *nf_dev leave at once.
template <bool L2, bool CBF, bool QBF>
__global__ __launch_bounds__(256) void exact_scores_kernel(const int* qlist, int nf, const int* nf_dev, int64_t n,
                                                           const void* corpus, int64_t ld_c,
                                                           const void* queries, int64_t ld_q, int d,
                                                           double* out /* [nf][n] */) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.y;
    if (nf_dev && f >= *nf_dev) return;
    if (f >= nf || j >= n) return;
    const int q = qlist ? qlist[f] : f;
    const char* qrow = reinterpret_cast<const char*>(queries) + (int64_t)q * ld_q * (QBF ? 2 : 4);
    const char* crow = reinterpret_cast<const char*>(corpus) + j * ld_c * (CBF ? 2 : 4);
    out[(int64_t)f * n + j] = canonical_score<L2, CBF, QBF>(qrow, crow, d);
}

// The same scores, tiled: a workgroup takes 256 corpus rows (a thread per row) and QT listed queries.  Row slices of 128
// bytes are brought into LDS by all threads with coalesced 16-byte loads (8 threads per row slice; one thread streaming its
// own row from global memory touches 64 cache lines per wave and load), each thread converts ITS slice to fp64 once and
// runs the QT fma chains over it, the queries' slices coming from LDS as fp64 broadcasts.  A corpus row is read once per QT
// queries instead of once per query, and the conversions are shared: 0.2 ms per query and 204,800 rows before, ~10 us now.
// The summation order of every (query, row) pair is unchanged: k = 0 .. d-1, one fma per component.
template <bool L2, bool CBF, bool QBF, int QT>
__global__ __launch_bounds__(256) void exact_scores_tiled_kernel(const int* qlist, int nf, const int* nf_dev, int64_t n,
                                                                 const void* corpus, int64_t ld_c,
                                                                 const void* queries, int64_t ld_q, int d,
                                                                 double* out /* [nf][n] */) {
    constexpr int CE = CBF ? 2 : 4, SLICE = 128 / CE, ROWB = 128 + 16;
    __shared__ __attribute__((aligned(16))) char rows_lds[256 * ROWB];
    __shared__ __attribute__((aligned(16))) double xq[QT][SLICE];
    const int tid = threadIdx.x;
    const int f0 = blockIdx.y * QT;
    const int nfl = nf_dev ? min(nf, *nf_dev) : nf;
    if (f0 >= nfl) return;
    const int64_t row0 = (int64_t)blockIdx.x * 256;
    int qn[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) { const int f = min(f0 + t, nfl - 1); qn[t] = qlist ? qlist[f] : f; }
    double acc[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) acc[t] = 0.0;
    const int lrow = tid >> 3, part = tid & 7;          // loader role: 32 rows per pass, 8 passes
    for (int k0 = 0; k0 < d; k0 += SLICE) {
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int r = ps * 32 + lrow;
            const int64_t gr = row0 + r < n ? row0 + r : n - 1;
            const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(corpus) + (gr * ld_c + k0) * CE + part * 16);
            *reinterpret_cast<uint4*>(rows_lds + r * ROWB + part * 16) = u;
        }
        for (int i = tid; i < QT * SLICE; i += 256) {
            const int t = i / SLICE, c = i - t * SLICE;
            xq[t][c] = load_as_double<QBF>(reinterpret_cast<const char*>(queries) + (int64_t)qn[t] * ld_q * (QBF ? 2 : 4), k0 + c);
        }
        __syncthreads();
        const char* rp = rows_lds + tid * ROWB;
#pragma unroll 2
        for (int c16 = 0; c16 < 8; ++c16) {
            const uint4 u = *reinterpret_cast<const uint4*>(rp + c16 * 16);
            const u32 w[4] = {u.x, u.y, u.z, u.w};
            constexpr int NE = 16 / CE;
            double y[NE];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (CBF) { y[2 * i] = (double)__uint_as_float(w[i] << 16); y[2 * i + 1] = (double)__uint_as_float(w[i] & 0xffff0000u); }
                else y[i] = (double)__uint_as_float(w[i]);
            }
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const double x = xq[t][c16 * NE + i];
                    if (L2) { const double dd = x - y[i]; acc[t] = __builtin_fma(dd, dd, acc[t]); }
                    else acc[t] = __builtin_fma(x, y[i], acc[t]);
                }
        }
    }
    if (row0 + tid < n) {
#pragma unroll
        for (int t = 0; t < QT; ++t)
            if (f0 + t < nfl) out[(int64_t)(f0 + t) * n + row0 + tid] = acc[t];
    }
}

// The k best of one query's n canonical scores, by (score, id).  One workgroup per listed query, a few passes over the row:
//   1. radix descent on the order-preserving 64-bit key, 11 bits a level: a histogram of the rows inside the current prefix,
//      then the digit in which the k-th best lies.  It stops as soon as the rows at or above that digit's lower edge fit the
//      sort (PICK_MAX): two or three levels on real scores;
//   2. those rows (unordered) into LDS, a bitonic sort by (key descending, id ascending), the first k written out.
//   All 64 bits consumed and still too many rows = more than PICK_MAX - k rows TIE at the k-th score (integer data: Morgan
//   counts): the rows above it go to the sort as they are, and of the tied ones the first k - above in row order -- an ordered
//   compaction that stops as soon as it has them.
// (Round 3 ran k rounds of "best after the previous pick" over all n scores: 150 ms per query at n = 1,000,000 and k = 256.)
constexpr int PICK_MAX = 4096;      // >= 2 TRX_MAX_K

template <bool L2>
__global__ __launch_bounds__(256) void exact_pick_kernel(const int* qlist, int nf, const int* nf_dev, int64_t n, int k,
                                                         const double* sc, float* D, int64_t* I,
                                                         double* S64) {
    __shared__ u64 s_key[PICK_MAX];
    __shared__ u32 s_id[PICK_MAX];
    __shared__ u32 hist[2048];
    __shared__ u64 sh_prefix;
    __shared__ u32 sh_above, sh_cnt, sh_pos, sh_wave[4];
    __shared__ int sh_bits, sh_done;
    const int tid = threadIdx.x;
    const int f = blockIdx.x;
    if (f >= nf || (nf_dev && f >= *nf_dev)) return;
    const int q = qlist ? qlist[f] : f;
    const double* row = sc + (int64_t)f * n;
    if (tid == 0) { sh_prefix = 0ull; sh_above = 0u; sh_bits = 0; sh_done = 0; sh_pos = 0u; }
    __syncthreads();
    // ---- 1. radix descent ----
    while (true) {
        const int bits = sh_bits;
        const int nb = 64 - bits < 11 ? 64 - bits : 11;
        const int shift = 64 - bits - nb;
        const u64 prefix = sh_prefix;
        for (int i = tid; i < 2048; i += 256) hist[i] = 0u;
        __syncthreads();
        int cur = -1; u32 run = 0u;      // a thread adds a run of equal digits at once (the top level: nearly every row in one digit)
#pragma unroll 4
        for (int64_t j = tid; j < n; j += 256) {
            const double sv = row[j];
            if (!(sv == sv)) continue;
            const u64 key = orddbl(L2 ? -sv : sv);
            if (bits != 0 && (key >> (64 - bits)) != prefix) continue;
            const int bin = (int)((key >> shift) & (u64)((1 << nb) - 1));
            if (bin == cur) ++run;
            else { if (run) atomicAdd(&hist[cur], run); cur = bin; run = 1u; }
        }
        if (run) atomicAdd(&hist[cur], run);
        __syncthreads();
        if (tid == 0) {
            u32 cum = sh_above; int b = (1 << nb) - 1;
            while (b > 0 && cum + hist[b] < (u32)k) { cum += hist[b]; --b; }      // (b = 0: fewer than k valid rows in all -- everything)
            if (cum + hist[b] < (u32)k) {      // digit 0 included and still short: take every row of the prefix's range and below
                sh_prefix = 0ull; sh_bits = 0; sh_above = 0u; sh_cnt = cum + hist[b]; sh_done = 1;      // lower edge 0: all valid rows
            } else {
                sh_above = cum; sh_prefix = (prefix << nb) | (u64)b; sh_bits = bits + nb; sh_cnt = cum + hist[b];
                if (cum + hist[b] <= (u32)PICK_MAX) sh_done = 1;
                else if (bits + nb == 64) sh_done = 2;      // ties at the k-th score
            }
        }
        __syncthreads();
        if (sh_done) break;
    }
    // ---- 2. gather ----
    const int mode = sh_done;
    const int bits = sh_bits;
    const u64 edge = bits == 0 ? 0ull : (bits == 64 ? sh_prefix : sh_prefix << (64 - bits));      // lower edge of the digit (mode 2: the tied key)
    u32 total;
    if (mode == 1) {
        total = sh_cnt;
#pragma unroll 4
        for (int64_t j = tid; j < n; j += 256) {
            const double sv = row[j];
            if (!(sv == sv)) continue;
            const u64 key = orddbl(L2 ? -sv : sv);
            if (key >= edge) { const u32 pos = atomicAdd(&sh_pos, 1u); s_key[pos] = key; s_id[pos] = (u32)j; }
        }
        __syncthreads();
    } else {
        const u32 above = sh_above, need = (u32)k - above;
        total = (u32)k;
#pragma unroll 4
        for (int64_t j = tid; j < n; j += 256) {
            const double sv = row[j];
            if (!(sv == sv)) continue;
            const u64 key = orddbl(L2 ? -sv : sv);
            if (key > edge) { const u32 pos = atomicAdd(&sh_pos, 1u); s_key[pos] = key; s_id[pos] = (u32)j; }
        }
        __syncthreads();
        // the first `need` tied rows in row order: 2,048 rows a step (8 consecutive rows per thread), stop when they are found
        u32 found = 0u;
        for (int64_t base = 0; base < n && found < need; base += 2048) {
            u32 m = 0u;
            const int64_t j0 = base + (int64_t)tid * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int64_t j = j0 + e;
                if (j < n) { const double sv = row[j]; if (sv == sv && orddbl(L2 ? -sv : sv) == edge) m |= 1u << e; }
            }
            const u32 mine = (u32)__popc(m);
            // exclusive prefix over the 256 threads: within the wave by shuffles, across the 4 waves through LDS
            u32 incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += t; }
            if ((tid & 63) == 63) sh_wave[tid >> 6] = incl;
            __syncthreads();
            u32 before = found + incl - mine;
            for (int w = 0; w < (tid >> 6); ++w) before += sh_wave[w];
            const u32 step_total = sh_wave[0] + sh_wave[1] + sh_wave[2] + sh_wave[3];
            for (int e = 0; e < 8; ++e)
                if ((m >> e) & 1u) { if (before < need) { s_key[above + before] = edge; s_id[above + before] = (u32)(j0 + e); } ++before; }
            found += step_total;
            __syncthreads();
        }
    }
    // ---- 3. sort, write ----
    int N = 2; while (N < (int)total) N <<= 1;
    for (int i = (int)total + tid; i < N; i += 256) { s_key[i] = 0ull; s_id[i] = 0xffffffffu; }      // (orddbl of a number is never 0)
    __syncthreads();
    if (total > 1u) bitonic_sort_pairs(s_key, s_id, N, tid);
    for (int t = tid; t < k; t += 256) {
        const int64_t o = (int64_t)q * k + t;
        if (t < (int)total) {
            const double sv = row[s_id[t]];
            D[o] = (float)sv; I[o] = (int64_t)s_id[t]; if (S64) S64[o] = sv;
        } else {
            D[o] = L2 ? FLT_MAX : -FLT_MAX; I[o] = -1;
            if (S64) S64[o] = L2 ? (double)FLT_MAX : -(double)FLT_MAX;
        }
    }
}

hipError_t launch_exact_scan(int metric, int cbf, int qbf, const int* qlist, int nf, const int* nf_dev, int64_t n,
                             const void* corpus, int64_t ld_c, const void* queries, int64_t ld_q,
                             int d, int k, double* sc, float* D, int64_t* I, double* S64,
                             hipStream_t st) {
    if (nf <= 0) return hipSuccess;
    if (n > 0) {
        const int ce = cbf ? 2 : 4;
        // the tiled kernel needs whole 128-byte slices and 16-byte aligned rows (every index built by knn_api.hip has them
        // unless d is not a multiple of 64 / 32); the generic one takes anything
        const bool tiled = (d % (128 / ce) == 0) && ((ld_c * ce) % 16 == 0) && ((uintptr_t)corpus % 16 == 0) && !getenv("TRX_EXACT_GENERIC");
        constexpr int QT = 8;
        dim3 grid((unsigned)((n + 255) / 256), tiled ? (nf + QT - 1) / QT : nf), block(256);
        const int sel = (metric ? 4 : 0) | (cbf ? 2 : 0) | (qbf ? 1 : 0);
#define TRX_ES(a, b, c)                                                                                                           \
        if (tiled) hipLaunchKernelGGL((exact_scores_tiled_kernel<a, b, c, QT>), grid, block, 0, st, qlist, nf, nf_dev, n, corpus, ld_c, queries, ld_q, d, sc); \
        else hipLaunchKernelGGL((exact_scores_kernel<a, b, c>), grid, block, 0, st, qlist, nf, nf_dev, n, corpus, ld_c, queries, ld_q, d, sc)
        switch (sel) {
            case 0: TRX_ES(false, false, false); break;
            case 1: TRX_ES(false, false, true); break;
            case 2: TRX_ES(false, true, false); break;
            case 3: TRX_ES(false, true, true); break;
            case 4: TRX_ES(true, false, false); break;
            case 5: TRX_ES(true, false, true); break;
            case 6: TRX_ES(true, true, false); break;
            default: TRX_ES(true, true, true); break;
        }
#undef TRX_ES
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (metric) hipLaunchKernelGGL(exact_pick_kernel<true>, dim3(nf), dim3(256), 0, st, qlist, nf, nf_dev, n, k, sc, D, I, S64);
    else hipLaunchKernelGGL(exact_pick_kernel<false>, dim3(nf), dim3(256), 0, st, qlist, nf, nf_dev, n, k, sc, D, I, S64);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// cross-shard merge: nlists sorted lists per query (fp64 scores, global ids) -> one list.
// One thread per query; nlists * k is at most a few hundred.
template <bool L2>
__global__ void merge_lists_kernel(int nlists, int64_t nq, int k, const double* S, const int64_t* Il,
                                   float* D, int64_t* I, double* So) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    int pos[16];
    for (int l = 0; l < nlists; ++l) pos[l] = 0;
    for (int t = 0; t < k; ++t) {
        int best = -1; double bs = 0.0; int64_t bi = 0;
        for (int l = 0; l < nlists; ++l) {
            if (pos[l] >= k) continue;
            const int64_t o = ((int64_t)l * nq + q) * k + pos[l];
            const int64_t id = Il[o];
            if (id < 0) { pos[l] = k; continue; }
            const double s = S[o];
            const bool better = best < 0 || (L2 ? s < bs : s > bs) || (s == bs && id < bi);
            if (better) { best = l; bs = s; bi = id; }
        }
        const int64_t o = q * k + t;
        if (best < 0) { D[o] = L2 ? FLT_MAX : -FLT_MAX; I[o] = -1; if (So) So[o] = L2 ? (double)FLT_MAX : -(double)FLT_MAX; }
        else { D[o] = (float)bs; I[o] = bi; if (So) So[o] = bs; pos[best]++; }
    }
}

hipError_t launch_merge(int metric, int nlists, int64_t nq, int k, const double* S, const int64_t* Il,
                        float* D, int64_t* I, double* So, hipStream_t st) {
    if (nq <= 0) return hipSuccess;
    dim3 grid((unsigned)((nq + 127) / 128)), block(128);
    if (metric) hipLaunchKernelGGL(merge_lists_kernel<true>, grid, block, 0, st, nlists, nq, k, S, Il, D, I, So);
    else hipLaunchKernelGGL(merge_lists_kernel<false>, grid, block, 0, st, nlists, nq, k, S, Il, D, I, So);
    return hipGetLastError();
}

// ---- FAISS' order among EXACT inner-product ties (TRX_TIES_FAISS, include/trx_knn.h) -------------------------------
// faiss.IndexFlatIP keeps its k best in a min-heap ordered by (score, id) [faiss/utils/Heap.h, ordered_key_value.h: CMin],
// fed in id order with a STRICT admission test [faiss/impl/ResultHandler.h], and empties it with heap_reorder
// (oracle/flat_knn_ref.c restates all three).  What that does to rows with EQUAL scores has a closed form:
//   * a row whose score ties with the heap's minimum is admitted only while the heap is not full -- so of the rows tied
//     at the k-th score (G, ids ascending) only those among the first k rows, by id, of {rows above the k-th score} u G
//     ever enter (t0 of them);
//   * every row above the k-th score that arrives once the heap is full evicts the heap's top = the tied row with the
//     SMALLEST id -- so of those t0 the m = k - a with the LARGEST ids remain (a = rows above the k-th score);
//   * heap_reorder pops smallest (score, id) first into the last place: the output is (score desc, id DESC).
// Input: the canonical top k2 = 2k of each query (score desc on the fp64 value, id asc; pads id = -1 last), which holds
// the a < k better rows and the first k tied ones -- all the rule can ask for.  One wave per query.
__global__ __launch_bounds__(64) void faiss_tie_kernel(int64_t nq, int k2, int k, const double* S2, const int64_t* I2, float* D, int64_t* I, double* So) {
    const int64_t q = blockIdx.x;
    const int lane = threadIdx.x;
    const double* s = S2 + q * k2;
    const int64_t* id = I2 + q * k2;
    auto wsum = [](int v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; };
    int c = 0;
    for (int t = lane; t < k2; t += 64) c += id[t] >= 0 ? 1 : 0;
    const int nvalid = wsum(c);
    int a, t0 = 0;
    if (nvalid <= k) a = nvalid;       // fewer rows than k: all of them, tie runs reversed, pads behind
    else {
        const double sk = s[k - 1];
        c = 0;
        for (int t = lane; t < k; t += 64) c += s[t] > sk ? 1 : 0;
        a = wsum(c);
        const int gend = min(nvalid, a + k);
        c = 0;
        for (int t = a + lane; t < gend; t += 64) {
            if (s[t] != sk) continue;      // (the run of the k-th score is contiguous from a on)
            int below = 0;                 // rows above the k-th score with a smaller id: they come first
            for (int b = 0; b < a; ++b) below += id[b] < id[t] ? 1 : 0;
            c += (t - a) + below < k ? 1 : 0;
        }
        t0 = wsum(c);
    }
    for (int p = lane; p < k; p += 64) {
        int src;
        if (p < a) {        // inside the run of equal scores it belongs to, mirrored
            int rs = p, re = p + 1;
            while (rs > 0 && s[rs - 1] == s[p]) --rs;
            while (re < a && s[re] == s[p]) ++re;
            src = rs + (re - 1 - p);
        } else if (nvalid > k) src = a + (t0 - 1 - (p - a));
        else src = -1;
        const int64_t o = q * k + p;
        if (src < 0) { D[o] = -FLT_MAX; I[o] = -1; if (So) So[o] = -(double)FLT_MAX; }
        else { D[o] = (float)s[src]; I[o] = id[src]; if (So) So[o] = s[src]; }
    }
}

hipError_t launch_faiss_ties(int64_t nq, int k2, int k, const double* S2, const int64_t* I2, float* D, int64_t* I, double* So, hipStream_t st) {
    if (nq <= 0) return hipSuccess;
    hipLaunchKernelGGL(faiss_tie_kernel, dim3((unsigned)nq), dim3(64), 0, st, nq, k2, k, S2, I2, D, I, So);
    return hipGetLastError();
}

}  // namespace trx
