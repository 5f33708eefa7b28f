// knn_scan.hip -- the hot kernel: Q x C^T as a bf16 MFMA GEMM with the top-k selection fused into
// the epilogue, so the Q x N score matrix never exists.  gfx950 only.
//
// Replaces the inside of faiss IndexFlat::search called at retrieve/retrieve_faiss.py:71 (the
// sgemm blocks + per-query heap of SURVEY.md section 2b rows N2/N3).
//
// Shape of the work
//   workgroup = 512 threads = 8 waves, owns ONE tile of 256 queries and walks a contiguous range
//   ("split") of 256-row corpus tiles.  Operands are swapped with respect to the usual GEMM
//   naming: A = corpus rows (MFMA M side), B = queries (MFMA N side), so in the accumulator the
//   QUERY is on the lane (col = lane & 15) and a lane's registers are consecutive corpus rows:
//   the per-query running maximum and threshold test are lane-local, no cross-lane traffic.
//   waves are laid out 2 (M) x 4 (N): each wave owns 128 corpus rows x 64 queries
//   = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16 = 128 accumulator registers per lane.
//
// LDS (one dynamic array, 16-byte aligned carve, 136,208 B -> one workgroup per CU)
//   A[2][256 rows][128 B] , B[2][256 rows][128 B] : K-step of 64 bf16 per row, double buffered,
//   filled by LDS-DMA (global_load_lds_dwordx4, no staging registers); 16-byte chunk c of row r
//   sits at slot (c ^ ((r >> 1) & 7)) -- the permutation is applied to the per-lane SOURCE address
//   -- which makes the ds_read_b128 fragment reads conflict-free (bank analysis in DESIGN.md).
//   thr_comp[256] u64, thr_key[256] f32, cnt[256] u32, ovf[256] u32, flags.
//
// Selection (per query, per split), exact with respect to the approximate key:
//   cand[...][64] is an append buffer.  A row is appended when comp(key, id) > thr_comp, where
//   thr is the kprime-th best packed (key,id) at the last compaction (0 = none yet).  When a
//   buffer passes `csoft` entries one wave sorts its 64 slots (bitonic, one slot per lane), keeps
//   the best kprime and raises thr.  Thresholds only tighten at compactions, so a query compacts
//   O(log tiles) times.  If a single tile overflows a buffer (always true for the first tile of a
//   split, rare afterwards) the workgroup dumps the 256 x 256 key tile to an L2-resident scratch
//   and the affected queries are rebuilt from {older entries} U {all 256 keys of the tile}: no
//   row is ever lost, whatever the data order.  Result: the list holds the top-kprime of the
//   split by (key desc, id asc) plus stale extras, and every unlisted row has comp <= thr.
//   Thresholds are shared across the splits of a query through g_thr[q] (atomicMax of the key of
//   a split's kprime-th best): a row below ANY split's kprime-th best has kprime better rows in
//   that split's list, so no split needs to keep it.  Reads of g_thr may be stale (cross-XCD L2s
//   are not coherent): a stale value is only a looser threshold, never a wrong one.  A tiny
//   bootstrap launch (one tile per query tile, dense rebuild) seeds g_thr so the real scan starts
//   warm and appends O(kprime log N) rows per query in total instead of per split.
#include "knn_common.h"
#include <atomic>
#include <cstdlib>

namespace trx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int LDS_A0 = 0;
constexpr int LDS_B0 = 2 * TILE_M * 128;                  // 65536
constexpr int LDS_THRC = LDS_B0 + 2 * TILE_N * 128;       // 131072
constexpr int LDS_THRK = LDS_THRC + TILE_N * 8;
constexpr int LDS_CNT = LDS_THRK + TILE_N * 4;
constexpr int LDS_OVF = LDS_CNT + TILE_N * 4;
constexpr int LDS_FLAGS = LDS_OVF + TILE_N * 4;
constexpr int LDS_WL = LDS_FLAGS + 16;     // flags[2]: one word per tile parity; then the compaction work list
constexpr int LDS_TOTAL = LDS_WL + 16 + TILE_N * 4;   // count + up to 256 query numbers

constexpr u32 FLAG_COMPACT = 1u;
constexpr u32 FLAG_DENSE = 2u;

__device__ __forceinline__ u64 ld_u64_l2(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_f32_l2(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// XCD-contiguous bijective remap of the block id (blocks b and b+8 share an XCD under the
// observed round-robin placement; speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// BOOT only gives the threshold-bootstrap launch its own symbol, so that profiles list the two
// launches separately (the main scan's average duration is the roofline number).
template <bool L2, int AUXA, int AUXB, bool BOOT>
__global__ __launch_bounds__(SCAN_THREADS, 2) void knn_scan_kernel(ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* lds_thrc = reinterpret_cast<u64*>(smem + LDS_THRC);
    float* lds_thrk = reinterpret_cast<float*>(smem + LDS_THRK);
    u32* lds_cnt = reinterpret_cast<u32*>(smem + LDS_CNT);
    u32* lds_ovf = reinterpret_cast<u32*>(smem + LDS_OVF);
    u32* lds_flags = reinterpret_cast<u32*>(smem + LDS_FLAGS);
    u32* lds_wlcnt = reinterpret_cast<u32*>(smem + LDS_WL);
    u32* lds_wl = lds_wlcnt + 4;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 2;  // 0..1 : corpus half of the tile
    const int wave_n = wave & 3;   // 0..3 : 64-query slice

    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int split = v % p.nsplits;
    const int qtile = v / p.nsplits;
    int tile0 = split * p.tiles_per_split;
    int tile1 = tile0 + p.tiles_per_split;
    if (p.bootstrap) {  // one full tile, spread over the corpus so query tiles do not collide
        const int span = p.ntiles > 1 ? p.ntiles - 1 : 1;
        tile0 = (int)(((long long)qtile * 97) % span);
        tile1 = tile0 + 1;
    }
    if (tile1 > p.ntiles) tile1 = p.ntiles;
    const int ntl = tile1 > tile0 ? tile1 - tile0 : 0;
    const int64_t qbase = (int64_t)qtile * TILE_N;

    // ---- init selection state ----
    if (tid < TILE_N) {
        const u32 g = (p.have_boot && !p.bootstrap)
                          ? __hip_atomic_load(p.g_thr + qbase + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                          : 0u;
        lds_thrc[tid] = (u64)g << 32;  // id part 0 == worst id: ties with the shared key still pass
        lds_thrk[tid] = g ? ordkey_inv(g) : -__builtin_inff();
        lds_cnt[tid] = 0u;
        lds_ovf[tid] = 0u;
    }
    if (tid == 0) { lds_flags[0] = 0u; lds_flags[1] = 0u; }

    const int ksteps = p.Kp / BK;
    const int total_steps = ntl * ksteps;

    // ---- staging geometry (LDS-DMA): wave w moves pieces p = 4w..4w+3 of each operand; a piece is
    // one global_load_lds_dwordx4 = 8 rows x 128 B, written lane-linear (row p*8 + lane/8, slot
    // lane%8).  The swizzle lives on the SOURCE: slot s of row r receives global chunk s ^ f(r),
    // f(r) = (r >> 1) & 7, which is what the fragment reads below undo.
    const int prow = lane >> 3, pslot = lane & 7;
    const int c_even = pslot ^ (prow >> 1);        // pieces with (p & 1) == 0
    const int c_odd = pslot ^ (4 + (prow >> 1));   // pieces with (p & 1) == 1
    const int64_t rowKp = (int64_t)p.Kp;
    int poff[4];                                   // element offset of this lane inside a tile, piece i
#pragma unroll
    for (int i = 0; i < 4; ++i)
        poff[i] = (int)(((wave * 4 + i) * 8 + prow) * rowKp) + ((i & 1) ? c_odd : c_even) * 8;
    const bf16_t* gA = p.corpus + (int64_t)tile0 * TILE_M * rowKp;
    const bf16_t* gB = p.queries + qbase * rowKp;
    const int lds_piece0 = wave * 4 * 1024;        // byte offset of this wave's first piece in a stage

    // ---- fragment read geometry ----
    const int frow = lane & 15, fq = lane >> 4;
    const int swz = frow >> 1;
    const int r_off0 = frow * 128 + ((fq ^ swz) << 4);
    const int r_off1 = frow * 128 + (((4 + fq) ^ swz) << 4);
    const int a_base = wave_m * 128 * 128;  // byte offset of this wave's first A row
    const int b_base = wave_n * 64 * 128;

    f32x4 acc[8][4];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

#ifdef TRX_STAMP_BUILD
    // diagnostic build: cycles of this wave in the per-tile filter, in the compaction / dense section (barriers
    // included), and how many lists it compacted
    unsigned long long st_filter = 0, st_compact = 0, st_ncomp = 0, st_t0 = 0;
#define TRX_T0() st_t0 = __builtin_readcyclecounter()
#define TRX_T1(ACC) ACC += __builtin_readcyclecounter() - st_t0
#else
#define TRX_T0()
#define TRX_T1(ACC)
#endif
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void gbl_void;
#define TRX_STAGE(S, BUF)                                                                           \
    {                                                                                               \
        const int tl_ = (S) / ksteps, ks_ = (S) - tl_ * ksteps;                                     \
        const bf16_t* a_ = gA + (int64_t)((p.debug & 1) ? 0 : tl_) * TILE_M * rowKp + ks_ * BK;                           \
        const bf16_t* b_ = gB + ks_ * BK;                                                           \
        char* la_ = smem + LDS_A0 + (BUF) * (TILE_M * 128) + lds_piece0;                            \
        char* lb_ = smem + LDS_B0 + (BUF) * (TILE_N * 128) + lds_piece0;                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                          \
            __builtin_amdgcn_global_load_lds((gbl_void*)(a_ + poff[i_]), (lds_void*)(la_ + i_ * 1024), 16, 0, AUXA); \
            __builtin_amdgcn_global_load_lds((gbl_void*)(b_ + poff[i_]), (lds_void*)(lb_ + i_ * 1024), 16, 0, AUXB); \
        }                                                                                           \
    }

    if (total_steps > 0) {
        TRX_STAGE(0, 0);
    }
    __syncthreads();

    int cur = 0;
    u32 gnext = 0u;
    int ks_in_tile = 0;
    int tl = 0;
    for (int s = 0; s < total_steps; ++s) {
        const bool has_next = (s + 1 < total_steps);
        if (has_next && !((p.debug & 4) && s > 2)) TRX_STAGE(s + 1, cur ^ 1);

        // ---- MFMA over this K-step: two 32-deep sub-steps ----
        const char* Ab = smem + LDS_A0 + cur * (TILE_M * 128) + a_base;
        const char* Bb = smem + LDS_B0 + cur * (TILE_N * 128) + b_base;
        if (!(p.debug & 8)) {
            // Hand-rotated fragment pipeline: 8 groups of 8 MFMAs (group g = (kk, pair of M tiles));
            // the LDS reads of group g+1 are issued before the MFMAs of group g, and the
            // sched_barriers keep hipcc from re-serialising them (it otherwise reads two fragments,
            // waits lgkmcnt(0), issues 8 MFMAs, and so on: 38 % MFMA utilisation).
            bf16x8 bq[2][4], ap[2][2];
#define TRX_LOAD_B(KK)                                                                          \
    _Pragma("unroll") for (int nt_ = 0; nt_ < 4; ++nt_)                                         \
        bq[KK][nt_] = *reinterpret_cast<const bf16x8*>(Bb + nt_ * 2048 + ((KK) ? r_off1 : r_off0));
#define TRX_LOAD_A(SLOT, KK, MP)                                                                \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                            \
        ap[SLOT][j_] = *reinterpret_cast<const bf16x8*>(Ab + ((MP) * 2 + j_) * 2048 + ((KK) ? r_off1 : r_off0));
            TRX_LOAD_B(0);
            TRX_LOAD_A(0, 0, 0);
#ifdef TRX_PRIO_EXPERIMENT
            __builtin_amdgcn_s_setprio(TRX_PRIO_EXPERIMENT);
#endif
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int kk = g >> 2, mp = g & 3;
                if (g + 1 < 8) {
                    const int kk2 = (g + 1) >> 2, mp2 = (g + 1) & 3;
                    TRX_LOAD_A((g + 1) & 1, kk2, mp2);
                    if (mp2 == 0) TRX_LOAD_B(1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mp * 2 + j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            ap[g & 1][j], bq[kk][nt], acc[mp * 2 + j][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef TRX_PRIO_EXPERIMENT
            __builtin_amdgcn_s_setprio(0);
#endif
#undef TRX_LOAD_A
#undef TRX_LOAD_B
        }

        if (ks_in_tile == 0 && tid < TILE_N)  // latency hidden under the tile's K loop
            gnext = __hip_atomic_load(p.g_thr + qbase + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool tile_done = (++ks_in_tile == ksteps);
        const int tile_row0 = (tile0 + tl) * TILE_M;

        if (tile_done && !(p.debug & 2)) {
            // ---- epilogue part 1: threshold filter + append (per wave, no barrier) ----
            // flags word alternates with the tile parity so that a fast wave's appends for the
            // next tile can never be seen by a slow wave still deciding about this one.
            TRX_T0();
            u32* flagw = lds_flags + (tl & 1);
            f32x4 bias[8];
            if (L2) {
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
                    bias[mt] = *reinterpret_cast<const f32x4*>(
                        p.cbias + tile_row0 + wave_m * 128 + mt * 16 + fq * 4);
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[mt][nt][r] = __builtin_fmaf(2.0f, acc[mt][nt][r], bias[mt][r]);
            }
            if (tl == 0 && (p.bootstrap || !p.have_boot)) {
                // cold start: no threshold yet, every row would pass -- go straight
                // to the dense rebuild instead of 65,536 contended appends.
                if (tid < TILE_N) lds_ovf[tid] = 1u;
                if (tid == 0) atomicOr(flagw, FLAG_DENSE);
            } else
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int ql = wave_n * 64 + nt * 16 + frow;
                const float tk = lds_thrk[ql];
                // two-level filter: per-lane maximum of each 4-row group, then of all 32 rows.
                // Most (wave, nt) pairs have some passing lane on most tiles, so the work after
                // the first ballot must stay cheap: 8 wave-uniform group tests, and only the
                // groups that really hold a passing row look at their 4 elements.
                float gm[8];
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
                    gm[mt] = fmaxf(fmaxf(acc[mt][nt][0], acc[mt][nt][1]), fmaxf(acc[mt][nt][2], acc[mt][nt][3]));
                const float m = fmaxf(fmaxf(fmaxf(gm[0], gm[1]), fmaxf(gm[2], gm[3])),
                                      fmaxf(fmaxf(gm[4], gm[5]), fmaxf(gm[6], gm[7])));
                if (__any(m >= tk)) {
                    const u64 tc = lds_thrc[ql];
                    u64* cq = p.cand + ((qbase + ql) * p.nsplits + split) * CAP;
#pragma unroll
                    for (int mt = 0; mt < 8; ++mt) {
                        if (!__any(gm[mt] >= tk)) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float key = acc[mt][nt][r] + 0.0f;  // -0 -> +0
                            if (key >= tk) {
                                const u32 id = (u32)(tile_row0 + wave_m * 128 + mt * 16 + fq * 4 + r);
                                const u64 c = make_comp(key, id);
                                if (id < (u32)p.n_valid && c > tc) {
                                    const u32 pos = atomicAdd(&lds_cnt[ql], 1u);
                                    if (pos < (u32)CAP) {
                                        cq[pos] = c;
                                        if (pos >= (u32)p.csoft) atomicOr(flagw, FLAG_COMPACT);
                                    } else {
                                        lds_ovf[ql] = 1u;
                                        atomicOr(flagw, FLAG_DENSE);
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }

        if (tile_done && !(p.debug & 2)) { TRX_T1(st_filter); }
        __syncthreads();
        cur ^= 1;

        if (tile_done) {
            // ---- epilogue part 2: rare compaction / dense rebuild (workgroup-uniform) ----
            const u32 fl = lds_flags[tl & 1];
            if (fl) {
                TRX_T0();
                float* scr = reinterpret_cast<float*>(p.scratch) + (int64_t)blockIdx.x * (TILE_N * TILE_M);
                if (fl & FLAG_DENSE) {
#pragma unroll
                    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const int ql = wave_n * 64 + nt * 16 + frow;
                            *reinterpret_cast<f32x4*>(scr + ql * TILE_M + wave_m * 128 + mt * 16 + fq * 4) =
                                acc[mt][nt];
                        }
                }
                if (tid == 0) *lds_wlcnt = 0u;
                __syncthreads();
                if (tid == 0) lds_flags[tl & 1] = 0u;
                // work list of the queries whose list must be compacted (or rebuilt): built by 256 threads at once and
                // dealt round-robin to the 8 waves -- a wave used to walk its own 32 queries one after the other, and
                // the whole workgroup waited at the barrier below for the wave that happened to own most of them
                if (tid < TILE_N) {
                    const bool need = lds_ovf[tid] != 0u || lds_cnt[tid] > (u32)p.csoft;
                    const u64 mk = __ballot(need);
                    u32 base = 0u;
                    if (lane == 0 && mk) base = atomicAdd(lds_wlcnt, (u32)__popcll(mk));
                    base = __shfl(base, 0, 64);
                    if (need) lds_wl[base + (u32)__popcll(mk & ((1ull << lane) - 1ull))] = (u32)tid;
                }
                __syncthreads();
                const int nwork = (int)*lds_wlcnt;
                for (int i = wave; i < nwork; i += 8) {
                    const int ql = (int)lds_wl[i];
                    const u32 c = lds_cnt[ql];
                    const bool dense = lds_ovf[ql] != 0u;
#ifdef TRX_STAMP_BUILD
                    ++st_ncomp;
#endif
                    u64* cq = p.cand + ((qbase + ql) * p.nsplits + split) * CAP;
                    const u32 cc = c < (u32)CAP ? c : (u32)CAP;
                    u64 val = (u32)lane < cc ? ld_u64_l2(cq + lane) : 0ull;
                    if (dense) {
                        // entries of the current tile are re-derived from the dump
                        if (val != 0ull && comp_id(val) >= (u32)tile_row0) val = 0ull;
                        val = wave_sort_desc(val, lane);
                        for (int ch = 0; ch < TILE_M / 32; ++ch) {
                            if (lane >= 32) {
                                const int rl = ch * 32 + (lane - 32);
                                const float key = ld_f32_l2(scr + ql * TILE_M + rl) + 0.0f;
                                const u32 id = (u32)(tile_row0 + rl);
                                val = (id < (u32)p.n_valid && key == key) ? make_comp(key, id) : 0ull;
                            }
                            val = wave_sort_desc(val, lane);
                        }
                    } else {
                        val = wave_sort_desc(val, lane);
                    }
                    if (lane < p.kprime) cq[lane] = val;
                    const u64 kept = __ballot(lane < p.kprime && val != 0ull);
                    const u32 ncnt = (u32)__popcll(kept);
                    const u64 tval = shfl_u64(val, p.kprime - 1);
                    if (lane == 0) {
                        lds_cnt[ql] = ncnt;
                        lds_ovf[ql] = 0u;
                        if (ncnt == (u32)p.kprime && tval > lds_thrc[ql]) {
                            lds_thrc[ql] = tval;
                            lds_thrk[ql] = comp_key(tval);
                            atomicMax(p.g_thr + qbase + ql, (u32)(tval >> 32));
                        }
                    }
                }
                __syncthreads();
                TRX_T1(st_compact);
            }
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // pick up thresholds other splits have published meanwhile (read by the next filter a
            // whole tile of barriers later)
            if (tid < TILE_N && gnext > (u32)(lds_thrc[tid] >> 32)) {
                lds_thrc[tid] = (u64)gnext << 32;
                lds_thrk[tid] = ordkey_inv(gnext);
            }
            ks_in_tile = 0;
            ++tl;
        }
    }

#ifdef TRX_STAMP_BUILD
    if (p.stamp_out && lane == 0) {
        unsigned long long* o = p.stamp_out + ((size_t)blockIdx.x * 8 + wave) * 4;
        o[0] = st_filter; o[1] = st_compact; o[2] = st_ncomp; o[3] = (unsigned long long)ntl;
    }
#endif
    // ---- publish per-(query, split) count and bound ----
    __syncthreads();
    if (tid < TILE_N && !p.bootstrap) {
        const int64_t o = (qbase + tid) * p.nsplits + split;
        const u32 c = lds_cnt[tid];
        p.cand_cnt[o] = c < (u32)CAP ? c : (u32)CAP;
        p.cand_thr[o] = lds_thrc[tid];
    }
}

template <bool L2, int AUXA, int AUXB, bool BOOT>
static hipError_t launch_one_b(const ScanParams& p, hipStream_t st) {
    // the attribute is per device (the ABI takes a device ordinal): one bit per ordinal, per instantiation
    static std::atomic<unsigned long long> attr_devs{0ull};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_scan_kernel<L2, AUXA, AUXB, BOOT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
        if (e != hipSuccess) return e;
        attr_devs.fetch_or(bit, std::memory_order_release);
    }
    dim3 grid(p.bootstrap ? p.nqtiles : p.nqtiles * p.nsplits), block(SCAN_THREADS);
    hipLaunchKernelGGL((knn_scan_kernel<L2, AUXA, AUXB, BOOT>), grid, block, LDS_TOTAL, st, p);
    return hipGetLastError();
}

template <bool L2, int AUXA, int AUXB>
static hipError_t launch_one(const ScanParams& p, hipStream_t st) {
    return p.bootstrap ? launch_one_b<L2, AUXA, AUXB, true>(p, st) : launch_one_b<L2, AUXA, AUXB, false>(p, st);
}

hipError_t launch_scan(const ScanParams& p, int metric, hipStream_t st) {
    // cache policy of the two LDS-DMA streams is the default one: `nt` on the corpus stream, the
    // query stream or both measured 107-122 ms against 92 ms, so only <0, 0>
    // is instantiated.
#ifdef TRX_POLICY_EXPERIMENT
    static const int pol = getenv("TRX_POLICY") ? atoi(getenv("TRX_POLICY")) : 0;   // 1: nt on queries, 2: nt on corpus
    if (metric != 1 && pol == 2) return launch_one<false, 2, 0>(p, st);
    if (metric != 1 && pol == 1) return launch_one<false, 0, 2>(p, st);
#endif
    return metric == 1 ? launch_one<true, 0, 0>(p, st) : launch_one<false, 0, 0>(p, st);
}

}  // namespace trx
