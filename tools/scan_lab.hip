// tools/scan_lab.hip -- loop-structure laboratory for knn_scan_kernel (gfx950).
//
// The production kernel = a 256 x 256 x K bf16 MFMA contraction per (query tile, corpus tile) + a
// selection epilogue.  This harness keeps only the contraction (the epilogue is a running maximum
// per query, which also serves as the cross-variant check) so that loop structures, grid mappings
// and cache behaviour can be compared in one process on one device:
//
//   scan_lab <variant> <nsplits> <flags> [ntiles] [reps]
//     variant 0 : the round-1 loop (DMA burst at the top of a K-step, C++ fragment reads,
//                 __syncthreads per K-step)
//     variant 1 : two-group ping-pong (waves 0-3 / 4-7 one interval apart), 32 MFMAs per phase,
//                 LDS-DMA issued inside the load phases in half-tile units with >= 2 intervals of
//                 lead, counted vmcnt, raw s_barrier, asm fragment reads
//     flags bit0: no DMA after the prologue (MFMA + LDS reads only)
//           bit1: no MFMA (fill only)
//           bit2: no per-tile maximum
//           bit3: (variant 4) fragments read once and reused: no LDS read traffic
//           bits 8+: workgroup (query tile q of its XCD group of 8) starts (q & 7) * (flags >> 8) * ~1024 cycles late: the 8 workgroups
//                 that share a corpus stream stop asking for the same tile at the same moment
//           bit7: (variants 1-4) odd corpus tiles walk their K-steps downwards (boustrophedon): see lab_v2
//           bit6: (variants 1-4, 9) drift gate: the 8 workgroups sharing a corpus stream wait for the slowest (LAB_DRIFT)
//           bit5: (variants 4, 9) a prefetching load per wave and K-step for the corpus slice LAB_PF_DIST K-steps ahead
//           bit4: (variants 4, 9) every tile re-reads the split's first corpus tile: the fill never misses L2
//     variant 9 : one wave per SIMD: 4 waves x (128 x 128), accumulators = the whole AGPR file (see lab_v3)
// build: hipcc --offload-arch=gfx950 -O3 tools/scan_lab.hip -o tools/scan_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

constexpr int TILE_M = 256, TILE_N = 256, BK = 64, THREADS = 512;
constexpr int LDS_A0 = 0, LDS_B0 = 2 * TILE_M * 128, LDS_TOTAL = LDS_B0 + 2 * TILE_N * 128;
constexpr int LDS_PF = LDS_TOTAL, LDS_ALLOC = LDS_TOTAL + 2048;   // 256 B per wave: where the prefetching loads of flags bit5 land
#ifndef LAB_PF_SHARE
#define LAB_PF_SHARE 1
#endif
#ifndef LAB_PF_DIST
#define LAB_PF_DIST 3          // K-steps between a prefetch and the LDS-DMA that asks for the same lines
#endif

struct LabParams {
    const bf16_t* corpus;
    const bf16_t* queries;
    float* out;          // [grid][8 waves][64 lanes][4]
    int Kp, ksteps, ntiles, tiles_per_split, nsplits, flags;
    unsigned long long* clk;   // [grid][4]: s_memtime / s_memrealtime at the start and at the end (wave 0)
    unsigned* gate;            // [nqtiles / 8][nsplits]: flags bit6, see drift_gate
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// flags bit6: the 8 workgroups that share a corpus stream (one split, query tiles 8g .. 8g+7: one XCD, one round of its 32
// workgroups) stay within LAB_DRIFT checkpoints (a checkpoint = two K-steps = 64 KiB of the stream) of the slowest of them,
// so that a corpus line is still in the L2 when the last of them asks for it.  One lane per workgroup; the wait is bounded.
#ifndef LAB_SNAKE
#define LAB_SNAKE 0     // 1: the query block index runs back and forth over a row of MFMAs (variants 4-7, 9)
#endif
#ifndef LAB_DRIFT
#define LAB_DRIFT 2
#endif
__device__ __forceinline__ void drift_gate(unsigned* ctr, unsigned done) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done > LAB_DRIFT) {
        const unsigned need = 8u * (done - LAB_DRIFT);
        for (int it = 0; it < 50000; ++it) {
            if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) break;
            __builtin_amdgcn_s_sleep(4);
        }
    }
}
#define LAB_GATE() if ((p.flags & 64) && tid == 0) { drift_gate(p.gate + (qtile >> 3) * p.nsplits + split, ++gate_done); }

#define LAB_MAX_ACC()                                                                                     \
    if (!(p.flags & 4)) {                                                                                 \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                \
            float m_ = mx[nt];                                                                            \
            _Pragma("unroll") for (int mt = 0; mt < 8; ++mt)                                              \
                m_ = fmaxf(m_, fmaxf(fmaxf(acc[mt][nt][0], acc[mt][nt][1]), fmaxf(acc[mt][nt][2], acc[mt][nt][3]))); \
            mx[nt] = m_;                                                                                  \
        }                                                                                                 \
    } else {                                                                                              \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) asm volatile("" :: "v"(acc[0][nt]), "v"(acc[7][nt])); \
    }                                                                                                     \
    _Pragma("unroll") for (int mt = 0; mt < 8; ++mt)                                                      \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

// ------------------------------------------------------------------------------------------------
// variant 0: the round-1 loop
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS, 2) void lab_v0(LabParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 2, wave_n = wave & 3;
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int split = v % p.nsplits, qtile = v / p.nsplits;
    if (p.flags >> 8) { const int n_ = (qtile & 7) * (p.flags >> 8); for (int i_ = 0; i_ < n_; ++i_) __builtin_amdgcn_s_sleep(16); }   // flags >> 8: start skew, ~1024 cycles per unit and per query tile of the XCD group
    int tile0 = split * p.tiles_per_split, tile1 = tile0 + p.tiles_per_split;
    if (tile1 > p.ntiles) tile1 = p.ntiles;
    const int ntl = tile1 > tile0 ? tile1 - tile0 : 0;
    const int ksteps = p.ksteps, total_steps = ntl * ksteps;
    const int prow = lane >> 3, pslot = lane & 7;
    const int c_even = pslot ^ (prow >> 1), c_odd = pslot ^ (4 + (prow >> 1));
    const int64_t rowKp = p.Kp;
    int poff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) poff[i] = (int)(((wave * 4 + i) * 8 + prow) * rowKp) + ((i & 1) ? c_odd : c_even) * 8;
    const bf16_t* gA = p.corpus + (int64_t)tile0 * TILE_M * rowKp;
    const bf16_t* gB = p.queries + (int64_t)qtile * TILE_N * rowKp;
    const int lds_piece0 = wave * 4 * 1024;
    const int frow = lane & 15, fq = lane >> 4, swz = frow >> 1;
    const int r_off0 = frow * 128 + ((fq ^ swz) << 4), r_off1 = frow * 128 + (((4 + fq) ^ swz) << 4);
    const int a_base = wave_m * 128 * 128, b_base = wave_n * 64 * 128;
    f32x4 acc[8][4];
    float mx[4] = {-3e38f, -3e38f, -3e38f, -3e38f};
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define V0_STAGE(S, BUF)                                                                                  \
    {                                                                                                     \
        const int tl_ = (S) / ksteps, ks_ = (S) - tl_ * ksteps;                                           \
        const bf16_t* a_ = gA + (int64_t)tl_ * TILE_M * rowKp + ks_ * BK;                                 \
        const bf16_t* b_ = gB + ks_ * BK;                                                                 \
        char* la_ = smem + LDS_A0 + (BUF) * (TILE_M * 128) + lds_piece0;                                  \
        char* lb_ = smem + LDS_B0 + (BUF) * (TILE_N * 128) + lds_piece0;                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                \
            __builtin_amdgcn_global_load_lds((gbl_void*)(a_ + poff[i_]), (lds_void*)(la_ + i_ * 1024), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((gbl_void*)(b_ + poff[i_]), (lds_void*)(lb_ + i_ * 1024), 16, 0, 0); \
        }                                                                                                 \
    }
    if (total_steps > 0) V0_STAGE(0, 0);
    __syncthreads();
    int cur = 0, ks_in_tile = 0;
    for (int s = 0; s < total_steps; ++s) {
        if (s + 1 < total_steps && !((p.flags & 1) && s > 2)) V0_STAGE(s + 1, cur ^ 1);
        const char* Ab = smem + LDS_A0 + cur * (TILE_M * 128) + a_base;
        const char* Bb = smem + LDS_B0 + cur * (TILE_N * 128) + b_base;
        if (!(p.flags & 2)) {
            bf16x8 bq[2][4], ap[2][2];
#define V0_LOAD_B(KK) _Pragma("unroll") for (int nt_ = 0; nt_ < 4; ++nt_) \
        bq[KK][nt_] = *reinterpret_cast<const bf16x8*>(Bb + nt_ * 2048 + ((KK) ? r_off1 : r_off0));
#define V0_LOAD_A(SLOT, KK, MP) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) \
        ap[SLOT][j_] = *reinterpret_cast<const bf16x8*>(Ab + ((MP) * 2 + j_) * 2048 + ((KK) ? r_off1 : r_off0));
            V0_LOAD_B(0);
            V0_LOAD_A(0, 0, 0);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int kk = g >> 2, mp = g & 3;
                if (g + 1 < 8) {
                    const int kk2 = (g + 1) >> 2, mp2 = (g + 1) & 3;
                    V0_LOAD_A((g + 1) & 1, kk2, mp2);
                    if (mp2 == 0) V0_LOAD_B(1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mp * 2 + j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[g & 1][j], bq[kk][nt], acc[mp * 2 + j][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (++ks_in_tile == ksteps) {
            LAB_MAX_ACC();
            ks_in_tile = 0;
        }
        __syncthreads();
        cur ^= 1;
    }
    float* o = p.out + ((size_t)blockIdx.x * THREADS + tid) * 4;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) o[nt] = mx[nt];
}

// ------------------------------------------------------------------------------------------------
// variant 1: two-group ping-pong
//
// Intervals (between two workgroup barriers) alternate roles: while waves 0-3 (group 0, corpus rows
// 0-127 of the tile) issue MFMAs, waves 4-7 (group 1, rows 128-255), which share their SIMDs, read
// fragments and issue LDS-DMA, and the other way round.  Phase p = (K-step u, half kk): L(p) reads
// 8 A + 4 B fragments of the 32-deep half, M(p) issues 32 MFMAs.  Group 1 runs one interval late.
//   interval 4u   : G0 L(u,0) [DMA B rows 0-127 of K-step u+1]    G1 M(u-1,1)
//   interval 4u+1 : G0 M(u,0)                                     G1 L(u,0) [DMA B rows 128-255 of u+1]
//   interval 4u+2 : G0 L(u,1) [DMA A rows 128-255 of u+1]         G1 M(u,0)
//   interval 4u+3 : G0 M(u,1)                                     G1 L(u,1) [DMA A rows 0-127 of u+2]
// Buffers: K-step u lives in stage u & 1.  Write-after-read: every fragment read is retired
// (lgkmcnt(0)) before the barrier that ends its interval, and each DMA above is issued at least one
// barrier after the last read of the half-tile it overwrites (A rows 0-127 are read by group 0 only,
// last in interval 4u+2; everything else last in interval 4u+3).  Read-after-write: a wave waits
// vmcnt(4) at the end of every load phase, i.e. for the pieces of its PREVIOUS load phase (two
// intervals old), and every half-tile has a barrier between that wait and its first read.
// ------------------------------------------------------------------------------------------------
template <int OPT>   // bit 0: fragment reads before the DMA issue of a load phase; bit 1: priority on the load phases instead of the MFMA phases
__global__ __launch_bounds__(THREADS, 2) void lab_v1(LabParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 2, wave_n = wave & 3;
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int split = v % p.nsplits, qtile = v / p.nsplits;
    if (p.flags >> 8) { const int n_ = (qtile & 7) * (p.flags >> 8); for (int i_ = 0; i_ < n_; ++i_) __builtin_amdgcn_s_sleep(16); }   // flags >> 8: start skew, ~1024 cycles per unit and per query tile of the XCD group
    int tile0 = split * p.tiles_per_split, tile1 = tile0 + p.tiles_per_split;
    if (tile1 > p.ntiles) tile1 = p.ntiles;
    const int ntl = tile1 > tile0 ? tile1 - tile0 : 0;
    const int ksteps = p.ksteps;          // even
    const int prow = lane >> 3, pslot = lane & 7;
    const int c_even = pslot ^ (prow >> 1), c_odd = pslot ^ (4 + (prow >> 1));
    const int Kp = p.Kp;
    // this wave's four pieces of a 128-row half: rows 32 * wave_n + 8 i + prow
    int poff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) poff[i] = ((wave_n * 4 + i) * 8 + prow) * Kp + ((i & 1) ? c_odd : c_even) * 8;
    const int half_elems = 128 * Kp;
    const bf16_t* gA = p.corpus + (int64_t)tile0 * TILE_M * Kp;
    const bf16_t* gB = p.queries + (int64_t)qtile * TILE_N * Kp;
    const int lds_piece0 = wave_n * 4 * 1024;    // inside a 128-row half (16 KiB)
    const int frow = lane & 15, fq = lane >> 4, swz = frow >> 1;
    const int r_off0 = frow * 128 + ((fq ^ swz) << 4), r_off1 = frow * 128 + (((4 + fq) ^ swz) << 4);
    // LDS byte addresses of the fragment reads (stage 0); stage 1 = +32768 through the offset field
    const unsigned aA0 = LDS_A0 + wave_m * 128 * 128 + r_off0, aA1 = LDS_A0 + wave_m * 128 * 128 + r_off1;
    const unsigned aB0 = LDS_B0 + wave_n * 64 * 128 + r_off0, aB1 = LDS_B0 + wave_n * 64 * 128 + r_off1;

    f32x4 acc[8][4];
    float mx[4] = {-3e38f, -3e38f, -3e38f, -3e38f};
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (ntl == 0) return;

    // ---- prologue: K-step 0 in full and A rows 0-127 of K-step 1, by all waves ----
    {
        const int w4 = wave * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pc = w4 + i;          // piece 0..31 of a 256-row operand stage
            const int off = (pc * 8 + prow) * Kp + ((pc & 1) ? c_odd : c_even) * 8;
            __builtin_amdgcn_global_load_lds((gbl_void*)(gA + off), (lds_void*)(smem + LDS_A0 + pc * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(gB + off), (lds_void*)(smem + LDS_B0 + pc * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pc = wave * 2 + i;    // piece 0..15: rows 0-127
            const int off = (pc * 8 + prow) * Kp + ((pc & 1) ? c_odd : c_even) * 8;
            __builtin_amdgcn_global_load_lds((gbl_void*)(gA + off + BK), (lds_void*)(smem + LDS_A0 + 32768 + pc * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // DMA cursors (wave-uniform).  B stream: K-steps 1, 2, ... (columns wrap per tile).  A stream:
    // group 0 stages rows 128-255 of K-steps 1, 2, ...; group 1 rows 0-127 of K-steps 2, 3, ...
    const bf16_t* srcB = gB + (wave_m ? half_elems : 0);
    int ksB = 1;
    const bf16_t* baseA1 = gA + (wave_m ? 0 : half_elems);
    int ksA = wave_m ? 2 : 1;
    // flags bit7 (as in variant 4): odd tiles walk their K-steps downwards
    const bool bous1 = (p.flags & 128) != 0;
    int tlB1 = 0, tlA1 = 0;
#define V1_KSE(KS, T) ((bous1 && ((T) & 1)) ? ksteps - 1 - (KS) : (KS))
    int kbe1 = V1_KSE(ksB, tlB1);
    const bf16_t* srcA = baseA1 + V1_KSE(ksA, tlA1) * BK;
    const bool dma_on = !(p.flags & 1);
    const bool mfma_on = !(p.flags & 2);

    if (wave_m) __builtin_amdgcn_s_barrier();       // group 1 runs one interval late

    bf16x8 fa[8], fb[4];
#define V1_READ(KK, STG)                                                                                   \
    {                                                                                                      \
        _Pragma("unroll") for (int nt_ = 0; nt_ < 4; ++nt_)                                                \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[nt_]) : "v"((KK) ? aB1 : aB0), "n"(nt_ * 2048 + (STG) * 32768) : "memory"); \
        _Pragma("unroll") for (int mt_ = 0; mt_ < 8; ++mt_)                                                \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[mt_]) : "v"((KK) ? aA1 : aA0), "n"(mt_ * 2048 + (STG) * 32768) : "memory"); \
    }
#define V1_WAIT_L()                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(4)" ::: "memory");                               \
    if (OPT & 2) __builtin_amdgcn_s_setprio(0);                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    __builtin_amdgcn_s_barrier();                                                                          \
    __builtin_amdgcn_sched_barrier(0);
#define V1_MFMA()                                                                                          \
    if (mfma_on) {                                                                                         \
        if (!(OPT & 2)) __builtin_amdgcn_s_setprio(1);                                                     \
        _Pragma("unroll") for (int mt_ = 0; mt_ < 8; ++mt_) {                                              \
            if ((OPT & 4) && mt_ == 6) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } \
            _Pragma("unroll") for (int nt_ = 0; nt_ < 4; ++nt_)                                            \
                acc[mt_][nt_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt_], fb[nt_], acc[mt_][nt_], 0, 0, 0); \
        }                                                                                                  \
        if (!(OPT & 2)) __builtin_amdgcn_s_setprio(0);                                                     \
    } else {                                                                                               \
        if (OPT & 4) __builtin_amdgcn_s_barrier();                                                         \
        _Pragma("unroll") for (int mt_ = 0; mt_ < 8; ++mt_) asm volatile("" :: "v"(fa[mt_]));              \
        _Pragma("unroll") for (int nt_ = 0; nt_ < 4; ++nt_) asm volatile("" :: "v"(fb[nt_]));              \
    }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);
#define V1_END_M()                                                                                         \
    if (!(OPT & 4)) __builtin_amdgcn_s_barrier();                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    if (OPT & 2) __builtin_amdgcn_s_setprio(1);
    // B pieces of this wave's half for the K-step after the current one -> stage STG
#define V1_DMA_B(STG)                                                                                      \
    if (dma_on) {                                                                                          \
        const bf16_t* s_ = srcB + kbe1 * BK;                                                                  \
        char* l_ = smem + LDS_B0 + (STG) * 32768 + wave_m * 16384 + lds_piece0;                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                   \
            __builtin_amdgcn_global_load_lds((gbl_void*)(s_ + poff[i_]), (lds_void*)(l_ + i_ * 1024), 16, 0, 0); \
    }                                                                                                      \
    if (++ksB == ksteps) { ksB = 0; ++tlB1; }                                                              \
    kbe1 = V1_KSE(ksB, tlB1);
    // A pieces: group 0 -> rows 128-255, group 1 -> rows 0-127, of the cursor's K-step -> stage STG
#define V1_DMA_A(STG)                                                                                      \
    if (dma_on) {                                                                                          \
        char* l_ = smem + LDS_A0 + (STG) * 32768 + (wave_m ? 0 : 16384) + lds_piece0;                      \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                   \
            __builtin_amdgcn_global_load_lds((gbl_void*)(srcA + poff[i_]), (lds_void*)(l_ + i_ * 1024), 16, 0, 0); \
    }                                                                                                      \
    if (++ksA == ksteps) { ksA = 0; ++tlA1; baseA1 += 256 * Kp; }                                          \
    srcA = baseA1 + V1_KSE(ksA, tlA1) * BK;

#define V1_L(KK, STG, DMA)                                                                                 \
    if (OPT & 1) { V1_READ(KK, STG); DMA; } else { DMA; V1_READ(KK, STG); }
    unsigned long long t0 = 0, r0 = 0;
    if (p.clk) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    if (OPT & 2) __builtin_amdgcn_s_setprio(1);
    unsigned gate_done = 0;
    for (int tl = 0; tl < ntl; ++tl) {
        for (int ks = 0; ks < ksteps; ks += 2) {
            LAB_GATE()
            // ---------------- K-step u = even (stage 0) ----------------
            V1_L(0, 0, V1_DMA_B(1));     // B of u+1 -> stage 1
            V1_WAIT_L();
            V1_MFMA();
            V1_END_M();
            // G1: A lo of u+2 -> stage 0; G0: A hi of u+1 -> stage 1
            V1_L(1, 0, if (wave_m) { V1_DMA_A(0); } else { V1_DMA_A(1); });
            V1_WAIT_L();
            V1_MFMA();
            V1_END_M();
            // ---------------- K-step u+1 = odd (stage 1) ----------------
            V1_L(0, 1, V1_DMA_B(0));     // B of u+2 -> stage 0
            V1_WAIT_L();
            V1_MFMA();
            V1_END_M();
            // G1: A lo of u+3 -> stage 1; G0: A hi of u+2 -> stage 0
            V1_L(1, 1, if (wave_m) { V1_DMA_A(1); } else { V1_DMA_A(0); });
            V1_WAIT_L();
            V1_MFMA();
            if (ks + 2 == ksteps) { LAB_MAX_ACC(); }
            V1_END_M();
        }
    }
    if (p.clk && tid == 0) {
        unsigned long long* c = p.clk + (size_t)blockIdx.x * 4;
        c[0] = t0; c[1] = r0; c[2] = __builtin_amdgcn_s_memtime(); c[3] = __builtin_amdgcn_s_memrealtime();
    }
    if (!wave_m) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* o = p.out + ((size_t)blockIdx.x * THREADS + tid) * 4;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) o[nt] = mx[nt];
}

// ------------------------------------------------------------------------------------------------
// variant 4: ping-pong with whole-K-step phases (64 MFMAs, 24 fragment reads, 8 DMA pieces per wave
// and phase; two barriers per K-step instead of four)
//   interval 2u   : G0 L(u) [DMA B, all 256 rows, of K-step u+1]              G1 M(u-1)
//   interval 2u+1 : G0 M(u)                                                   G1 L(u) [DMA A rows 128-255 of u+1, A rows 0-127 of u+2]
// ------------------------------------------------------------------------------------------------
template <int EARLY>   // MFMAs of a phase issued AFTER its closing barrier (0: barrier after the last one)
__global__ __launch_bounds__(THREADS, 2) void lab_v2(LabParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 2, wave_n = wave & 3;
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int split = v % p.nsplits, qtile = v / p.nsplits;
    if (p.flags >> 8) { const int n_ = (qtile & 7) * (p.flags >> 8); for (int i_ = 0; i_ < n_; ++i_) __builtin_amdgcn_s_sleep(16); }   // flags >> 8: start skew, ~1024 cycles per unit and per query tile of the XCD group
    int tile0 = split * p.tiles_per_split, tile1 = tile0 + p.tiles_per_split;
    if (tile1 > p.ntiles) tile1 = p.ntiles;
    const int ntl = tile1 > tile0 ? tile1 - tile0 : 0;
    const int ksteps = p.ksteps;          // even
    const int prow = lane >> 3, pslot = lane & 7;
    const int c_even = pslot ^ (prow >> 1), c_odd = pslot ^ (4 + (prow >> 1));
    const int Kp = p.Kp;
    int poff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) poff[i] = ((wave_n * 4 + i) * 8 + prow) * Kp + ((i & 1) ? c_odd : c_even) * 8;
    const int half_elems = 128 * Kp;
    const bf16_t* gA = p.corpus + (int64_t)tile0 * TILE_M * Kp;
    const bf16_t* gB = p.queries + (int64_t)qtile * TILE_N * Kp;
    const int lds_piece0 = wave_n * 4 * 1024;
    const int frow = lane & 15, fq = lane >> 4, swz = frow >> 1;
    const int r_off0 = frow * 128 + ((fq ^ swz) << 4), r_off1 = frow * 128 + (((4 + fq) ^ swz) << 4);
    const unsigned aA0 = LDS_A0 + wave_m * 128 * 128 + r_off0, aA1 = LDS_A0 + wave_m * 128 * 128 + r_off1;
    const unsigned aB0 = LDS_B0 + wave_n * 64 * 128 + r_off0, aB1 = LDS_B0 + wave_n * 64 * 128 + r_off1;
    f32x4 acc[8][4];
    float mx[4] = {-3e38f, -3e38f, -3e38f, -3e38f};
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (ntl == 0) return;
    {
        const int w4 = wave * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pc = w4 + i;
            const int off = (pc * 8 + prow) * Kp + ((pc & 1) ? c_odd : c_even) * 8;
            __builtin_amdgcn_global_load_lds((gbl_void*)(gA + off), (lds_void*)(smem + LDS_A0 + pc * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(gB + off), (lds_void*)(smem + LDS_B0 + pc * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pc = wave * 2 + i;
            const int off = (pc * 8 + prow) * Kp + ((pc & 1) ? c_odd : c_even) * 8;
            __builtin_amdgcn_global_load_lds((gbl_void*)(gA + off + BK), (lds_void*)(smem + LDS_A0 + 32768 + pc * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    int ksB = 1;                                       // G0: B of K-step u+1
    int ksH = 1;                                       // G1: A rows 128-255 of K-step u+1
    int ksL = 2;                                       // G1: A rows 0-127 of K-step u+2
    // flags bit7: odd corpus tiles walk the K-steps DOWN (11 .. 0), even ones up: the query tile's slices are then re-used most
    // recent first, which is what an LRU cache slightly smaller than the working set (Traffic) can serve
    const bool bous = (p.flags & 128) != 0;
    int tB = 0, tH = 0, tL = 0;                        // the corpus tile each cursor is in
    const int tile_step = (p.flags & 16) ? 0 : 256 * Kp;
    const bf16_t* baseH = gA + half_elems;
    const bf16_t* baseL = gA;
#define V2_KSE(KS, T) ((bous && ((T) & 1)) ? ksteps - 1 - (KS) : (KS))
    const bf16_t* srcH = baseH + BK;
    const bf16_t* srcL = baseL + 2 * BK;
    // flags bit4: every tile re-reads the split's FIRST corpus tile (the cursors step back instead of on): the whole fill is
    // served from L2 -- what the fill costs when nothing misses (results differ from variant 0 by construction)
    const int wrapA = (p.flags & 16) ? -Kp : 255 * Kp;
    const bool dma_on = !(p.flags & 1);
    const bool mfma_on = !(p.flags & 2);
    const bool rd_on = !(p.flags & 8);     // flags bit3: fragments are read once and reused (no LDS read traffic)
    const bool pf_on = (p.flags & 32) && dma_on && !(p.flags & 128);   // flags bit5: group 1 touches the corpus lines LAB_PF_DIST K-steps ahead (into L2)
    const int pf_row = (wave_n * 64 + lane) * Kp;
    if (wave_m) __builtin_amdgcn_s_barrier();
    bf16x8 fa[2][8], fb[2][4];
#define V2_READ(STG)                                                                                       \
    _Pragma("unroll") for (int kk_ = 0; kk_ < 2; ++kk_) {                                                  \
        _Pragma("unroll") for (int nt_ = 0; nt_ < 4; ++nt_)                                                \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[kk_][nt_]) : "v"(kk_ ? aB1 : aB0), "n"(nt_ * 2048 + (STG) * 32768) : "memory"); \
        _Pragma("unroll") for (int mt_ = 0; mt_ < 8; ++mt_)                                                \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[kk_][mt_]) : "v"(kk_ ? aA1 : aA0), "n"(mt_ * 2048 + (STG) * 32768) : "memory"); \
    }
    // STG = stage of the K-step being READ (u & 1)
#define V2_DMA(STG)                                                                                        \
    if (!wave_m) {                                                                                         \
        if (dma_on) {                                                                                      \
            const bf16_t* s_ = gB + V2_KSE(ksB, tB) * BK;                                                  \
            char* l_ = smem + LDS_B0 + ((STG) ^ 1) * 32768 + lds_piece0;                                   \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                             \
                __builtin_amdgcn_global_load_lds((gbl_void*)(s_ + poff[i_]), (lds_void*)(l_ + i_ * 1024), 16, 0, 0); \
                __builtin_amdgcn_global_load_lds((gbl_void*)(s_ + half_elems + poff[i_]), (lds_void*)(l_ + 16384 + i_ * 1024), 16, 0, 0); \
            }                                                                                              \
        }                                                                                                  \
        if (++ksB == ksteps) { ksB = 0; ++tB; }                                                            \
    } else {                                                                                               \
        if (dma_on) {                                                                                      \
            char* lh_ = smem + LDS_A0 + ((STG) ^ 1) * 32768 + 16384 + lds_piece0;                          \
            char* ll_ = smem + LDS_A0 + (STG) * 32768 + lds_piece0;                                        \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                               \
                __builtin_amdgcn_global_load_lds((gbl_void*)(srcH + poff[i_]), (lds_void*)(lh_ + i_ * 1024), 16, 0, 0); \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                               \
                __builtin_amdgcn_global_load_lds((gbl_void*)(srcL + poff[i_]), (lds_void*)(ll_ + i_ * 1024), 16, 0, 0); \
        }                                                                                                  \
        if (pf_on) __builtin_amdgcn_global_load_lds((gbl_void*)(srcL + LAB_PF_DIST * BK + (ksL + LAB_PF_DIST >= ksteps ? wrapA : 0) + pf_row), \
                                                    (lds_void*)(smem + LDS_PF + wave * 256), 4, 0, 0);     \
        if (++ksH == ksteps) { ksH = 0; ++tH; baseH += tile_step; }                                        \
        if (++ksL == ksteps) { ksL = 0; ++tL; baseL += tile_step; }                                        \
        srcH = baseH + V2_KSE(ksH, tH) * BK; srcL = baseL + V2_KSE(ksL, tL) * BK;                          \
    }
    // end of a load phase: fragments in registers; G1 additionally needs its A rows 0-127 of the PREVIOUS
    // load phase landed (read by G0 in the next interval): all but the 8 pieces just issued
#define V2_WAIT_L()                                                                                        \
    if (pf_on && wave_m) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(10)" ::: "memory");   /* + this phase's prefetch and the previous one */ \
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(8)" ::: "memory");                          \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    __builtin_amdgcn_s_barrier();                                                                          \
    __builtin_amdgcn_sched_barrier(0);
    // end of an MFMA phase: G0 needs all its B pieces landed (read in the next interval), G1 its A rows
    // 128-255 (the first four of the eight it issued)
#define V2_BAR_M()                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    if (wave_m) { if (pf_on) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); } \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                  \
    __builtin_amdgcn_s_barrier();                                                                          \
    __builtin_amdgcn_sched_barrier(0);
#define V2_END_M() if (EARLY == 0) { V2_BAR_M(); }
#define V2_MFMA()                                                                                          \
    if (mfma_on) {                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 64; ++i_) {                                                \
            const int kk_ = i_ >> 5, mt_ = (i_ >> 2) & 7, nt_ = (LAB_SNAKE && (mt_ & 1)) ? 3 - (i_ & 3) : (i_ & 3); \
            if (EARLY > 0 && i_ == 64 - EARLY) { V2_BAR_M(); }                                             \
            acc[mt_][nt_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk_][mt_], fb[kk_][nt_], acc[mt_][nt_], 0, 0, 0); \
        }                                                                                                  \
        __builtin_amdgcn_s_setprio(0);                                                                     \
    } else {                                                                                               \
        if (EARLY > 0) { V2_BAR_M(); }                                                                     \
        _Pragma("unroll") for (int kk_ = 0; kk_ < 2; ++kk_) {                                              \
            _Pragma("unroll") for (int mt_ = 0; mt_ < 8; ++mt_) asm volatile("" :: "v"(fa[kk_][mt_]));     \
            _Pragma("unroll") for (int nt_ = 0; nt_ < 4; ++nt_) asm volatile("" :: "v"(fb[kk_][nt_]));     \
        }                                                                                                  \
    }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0 = 0, r0 = 0;
    if (p.clk) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    unsigned gate_done = 0;
    for (int tl = 0; tl < ntl; ++tl) {
        for (int ks = 0; ks < ksteps; ks += 2) {
            LAB_GATE()
            if (rd_on || (tl == 0 && ks == 0)) { V2_READ(0); } V2_DMA(0); V2_WAIT_L(); V2_MFMA(); V2_END_M();
            if (rd_on) { V2_READ(1); } V2_DMA(1); V2_WAIT_L(); V2_MFMA();
            if (ks + 2 == ksteps) { LAB_MAX_ACC(); }
            V2_END_M();
        }
    }
    if (p.clk && tid == 0) {
        unsigned long long* c = p.clk + (size_t)blockIdx.x * 4;
        c[0] = t0; c[1] = r0; c[2] = __builtin_amdgcn_s_memtime(); c[3] = __builtin_amdgcn_s_memrealtime();
    }
    if (!wave_m) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* o = p.out + ((size_t)blockIdx.x * THREADS + tid) * 4;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) o[nt] = mx[nt];
}

// ------------------------------------------------------------------------------------------------
// variant 9: ONE wave per SIMD -- 4 waves x (128 x 128), 256 accumulator registers per wave (the whole AGPR file), the
// fragments of the next 32-deep slice read into a second register set while the 64 MFMAs of the current one issue, the
// LDS-DMA pieces of K-step u+2 issued inside the second slice of K-step u, ONE barrier per K-step (between the slices:
// by then every wave holds all fragments of stage u&1 and its own pieces of K-step u+1 have landed).
// Same LDS image as the other variants (two 64 KiB stages), 16 fragment reads per 128x128x32 instead of 12 per 128x64x32:
// a third fewer LDS fragment bytes per MAC.  The first slice of a tile starts from C = 0 (no clearing pass), the per-query
// maximum of a tile sits in the gaps of its last slice.
// ------------------------------------------------------------------------------------------------
constexpr int V3_THREADS = 256;
// -DV3_EARLY_DMA=1: the 16 pieces of a K-step behind every 2nd MFMA of the slice instead of every 4th (longer lead).
// Measured: SLOWER (74.9 against 67.4 ms: a wave holds about 12 LDS-DMA instructions in flight, tools/experiments/
// ldsdma_counters.hip; the 13th stalls the wave's whole instruction stream, MFMAs included, and one wave per SIMD has nobody
// to cover it) -- and the check against variant 0 FAILS with it (the last two query pieces of waves 1 and 3 read stale;
// same with all 16 back to back, fine with up to 8 early): not understood, the option stays for whoever looks next.
#ifndef V3_EARLY_DMA
#define V3_EARLY_DMA 0
#endif
template <int FL>      // the run's flags bits 0..3, compiled in (a runtime test per MFMA would sit inside the stream)
__global__ __launch_bounds__(V3_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void lab_v3(LabParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int split = v % p.nsplits, qtile = v / p.nsplits;
    if (p.flags >> 8) { const int n_ = (qtile & 7) * (p.flags >> 8); for (int i_ = 0; i_ < n_; ++i_) __builtin_amdgcn_s_sleep(16); }   // flags >> 8: start skew, ~1024 cycles per unit and per query tile of the XCD group
    int tile0 = split * p.tiles_per_split, tile1 = tile0 + p.tiles_per_split;
    if (tile1 > p.ntiles) tile1 = p.ntiles;
    const int ntl = tile1 > tile0 ? tile1 - tile0 : 0;
    const int ksteps = p.ksteps;          // even, >= 6
    const int prow = lane >> 3, pslot = lane & 7;
    const int c_even = pslot ^ (prow >> 1), c_odd = pslot ^ (4 + (prow >> 1));
    const int Kp = p.Kp;
    int poff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) poff[i] = ((wave * 8 + i) * 8 + prow) * Kp + ((i & 1) ? c_odd : c_even) * 8;
    const bf16_t* gA = p.corpus + (int64_t)tile0 * TILE_M * Kp;
    const bf16_t* gB = p.queries + (int64_t)qtile * TILE_N * Kp;
    const int lds_piece0 = wave * 8 * 1024;
    const int frow = lane & 15, fq = lane >> 4, swz = frow >> 1;
    const int r_off0 = frow * 128 + ((fq ^ swz) << 4), r_off1 = frow * 128 + (((4 + fq) ^ swz) << 4);
    const unsigned aA0 = LDS_A0 + wave_m * 128 * 128 + r_off0, aA1 = LDS_A0 + wave_m * 128 * 128 + r_off1;
    const unsigned aB0 = LDS_B0 + wave_n * 128 * 128 + r_off0, aB1 = LDS_B0 + wave_n * 128 * 128 + r_off1;
    f32x4 acc[8][8];
    float mx[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) mx[nt] = -3e38f;
    if (ntl == 0) return;
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_global_load_lds((gbl_void*)(gA + st * BK + poff[i]), (lds_void*)(smem + LDS_A0 + st * 32768 + lds_piece0 + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(gB + st * BK + poff[i]), (lds_void*)(smem + LDS_B0 + st * 32768 + lds_piece0 + i * 1024), 16, 0, 0);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bf16_t* srcA = gA + 2 * BK; int ksA = 2;     // K-step u+2 of the corpus stream
    int ksB = 2;
    const int wrapA = (p.flags & 16) ? -Kp : 255 * Kp;
    constexpr bool dma_on = !(FL & 1), mfma_on = !(FL & 2), max_on = !(FL & 4), rd_on = !(FL & 8), pf_on = (FL & 32) != 0;
    int kcount = 0;
    const int pf_row = (wave * 64 + lane) * Kp;     // flags bit5: this lane's row of the slice LAB_PF_DIST K-steps ahead (one 128-byte line)
    bf16x8 fa[2][8], fb[2][8];
    // one fragment read: number R 0..7 = B column blocks, 8..15 = A row blocks, of slice KK in stage STG, into set SET
#define V3_RD1(SET, KK, STG, R)                                                                            \
    if ((R) < 8) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[SET][(R) & 7]) : "v"((KK) ? aB1 : aB0), "n"(((R) & 7) * 2048 + (STG) * 32768) : "memory"); \
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[SET][(R) & 7]) : "v"((KK) ? aA1 : aA0), "n"(((R) & 7) * 2048 + (STG) * 32768) : "memory");
    // one LDS-DMA piece of K-step u+2 into stage STG: piece D 0..7 corpus, 8..15 queries
#define V3_DMA1(STG, D)                                                                                    \
    if (dma_on) {                                                                                          \
        if ((D) < 8) __builtin_amdgcn_global_load_lds((gbl_void*)(srcA + poff[(D) & 7]), (lds_void*)(smem + LDS_A0 + (STG) * 32768 + lds_piece0 + ((D) & 7) * 1024), 16, 0, 0); \
        else __builtin_amdgcn_global_load_lds((gbl_void*)(gB + ksB * BK + poff[(D) & 7]), (lds_void*)(smem + LDS_B0 + (STG) * 32768 + lds_piece0 + ((D) & 7) * 1024), 16, 0, 0); \
    }
#define V3_MAXROW(MT)                                                                                      \
    if (max_on) {                                                                                  \
        _Pragma("unroll") for (int nt_ = 0; nt_ < 8; ++nt_)                                                \
            mx[nt_] = fmaxf(mx[nt_], fmaxf(fmaxf(acc[MT][nt_][0], acc[MT][nt_][1]), fmaxf(acc[MT][nt_][2], acc[MT][nt_][3]))); \
    } else { _Pragma("unroll") for (int nt_ = 0; nt_ < 8; ++nt_) asm volatile("" :: "v"(acc[MT][nt_])); }
    // a phase = the 64 MFMAs of fragment set SET; beside them: READ -> the 16 fragment reads of (slice RKK, stage RSTG) into
    // the other set; DMA -> the 16 pieces of K-step u+2 into stage DSTG; FIRST -> C = 0; LAST -> the tile's maxima
#define V3_PHASE(SET, READ, RKK, RSTG, DMA, DSTG, FIRST, LAST)                                             \
    {                                                                                                      \
        __builtin_amdgcn_s_setprio(1);                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 64; ++i_) {                                                \
            const int mt_ = i_ >> 3, nt_ = (LAB_SNAKE && (mt_ & 1)) ? 7 - (i_ & 7) : (i_ & 7);   /* LAB_SNAKE: one operand changes per MFMA */ \
            if (mfma_on) {   /* asm, accumulators pinned to the AGPR file in place: with the builtin hipcc re-homes part of */ \
                             /* them in VGPRs (v_accvgpr_read + s_nop 7 behind every such MFMA) */         \
                if (FIRST) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[mt_][nt_]) : "v"(fa[SET][mt_]), "v"(fb[SET][nt_])); \
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mt_][nt_]) : "v"(fa[SET][mt_]), "v"(fb[SET][nt_])); \
            }                                                                                              \
            if ((READ) && (i_ & 3) == 1) { V3_RD1((SET) ^ 1, RKK, RSTG, i_ >> 2) __builtin_amdgcn_sched_barrier(0); } \
            if ((DMA) && (V3_EARLY_DMA ? (i_ < 32 && (i_ & 1) == 0) : ((i_ & 3) == 3))) {                  \
                V3_DMA1(DSTG, V3_EARLY_DMA ? (i_ >> 1) : (i_ >> 2)) __builtin_amdgcn_sched_barrier(0);     \
            }                                                                                              \
            if ((LAST) && (i_ & 7) == 7 && mt_ >= 2) { V3_MAXROW(mt_ - 2) }                                     \
        }                                                                                                  \
        if (LAST) { V3_MAXROW(6) V3_MAXROW(7) }                                                            \
        if (!mfma_on) {                                                                                    \
            _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) { asm volatile("" :: "v"(fa[SET][j_])); asm volatile("" :: "v"(fb[SET][j_])); } \
            if (FIRST) { _Pragma("unroll") for (int a_ = 0; a_ < 8; ++a_) _Pragma("unroll") for (int b_ = 0; b_ < 8; ++b_) acc[a_][b_] = (f32x4){0.f, 0.f, 0.f, 0.f}; } \
        }                                                                                                  \
        __builtin_amdgcn_s_setprio(0);                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }
    // between the two slices of a K-step: all my fragments of this stage are in registers, my pieces of the next K-step landed
#define V3_SYNC()                                                                                          \
    if (pf_on) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(1)" ::: "memory");   /* the newest operation is the prefetch */ \
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)" ::: "memory");                          \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    __builtin_amdgcn_s_barrier();                                                                          \
    __builtin_amdgcn_sched_barrier(0);
#define V3_ADVANCE()                                                                                       \
    if (pf_on && dma_on) { /* one dword per lane, 64 lines per wave: the whole 256-row slice between the four waves; the 8 workgroups */ \
        /* that share the stream take turns (LAB_PF_SHARE), the others re-touch a query line so that the wave's VMEM count stays the same */ \
        const bool mine_ = !LAB_PF_SHARE || ((kcount & 7) == (qtile & 7));                                 \
        const bf16_t* a_ = mine_ ? srcA + LAB_PF_DIST * BK + (ksA + LAB_PF_DIST >= ksteps ? wrapA : 0) + pf_row : gB; \
        __builtin_amdgcn_global_load_lds((gbl_void*)a_, (lds_void*)(smem + LDS_PF + wave * 256), 4, 0, 0); \
        ++kcount;                                                                                          \
    }                                                                                                      \
    srcA += BK; if (++ksA == ksteps) { ksA = 0; srcA += wrapA; }                                           \
    ksB = (ksB + 1 == ksteps) ? 0 : ksB + 1;
#define V3_FRAGS_LANDED() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    // K-step in stage STG: slice 0 from set 0 (reading slice 1 of the same stage into set 1), sync, slice 1 from set 1
    // (issuing K-step u+2 into this stage and reading slice 0 of the other stage into set 0)
#define V3_KSTEP(STG, FIRST, LAST)                                                                         \
    V3_PHASE(0, rd_on, 1, STG, false, 0, FIRST, false)                                                     \
    V3_SYNC()                                                                                              \
    V3_PHASE(1, rd_on, 0, (STG) ^ 1, true, STG, false, LAST)                                               \
    V3_ADVANCE()                                                                                           \
    V3_FRAGS_LANDED()
#pragma unroll
    for (int r = 0; r < 16; ++r) { V3_RD1(0, 0, 0, r) }
    if (!rd_on) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { V3_RD1(1, 1, 0, r) }
    }
    V3_FRAGS_LANDED()
    unsigned long long t0 = 0, r0 = 0;
    if (p.clk) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    unsigned gate_done = 0;
    for (int tl = 0; tl < ntl; ++tl) {
        LAB_GATE()
        V3_KSTEP(0, true, false)
        V3_KSTEP(1, false, false)
        for (int ks = 2; ks + 2 < ksteps; ks += 2) {
            LAB_GATE()
            V3_KSTEP(0, false, false)
            V3_KSTEP(1, false, false)
        }
        LAB_GATE()
        V3_KSTEP(0, false, false)
        V3_KSTEP(1, false, true)
    }
    if (p.clk && tid == 0) {
        unsigned long long* c = p.clk + (size_t)blockIdx.x * 4;
        c[0] = t0; c[1] = r0; c[2] = __builtin_amdgcn_s_memtime(); c[3] = __builtin_amdgcn_s_memrealtime();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* o = p.out + ((size_t)blockIdx.x * V3_THREADS + tid) * 8;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) o[nt] = mx[nt];
}

static unsigned short f2bf(float f) {
    unsigned u; memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const int nsplits = argc > 2 ? atoi(argv[2]) : 1;
    const int flags = argc > 3 ? atoi(argv[3]) : 0;
    const int ntiles = argc > 4 ? atoi(argv[4]) : 3907;
    const int reps = argc > 5 ? atoi(argv[5]) : 3;
    const int Kp = 768, ksteps = Kp / BK, nqtiles = 256;
    const size_t crow = (size_t)(ntiles + 2) * TILE_M, qrow = (size_t)nqtiles * TILE_N;
    // random bf16 operands (full-range values: zero-filled operands clock higher and read fast)
    std::vector<unsigned short> h(crow * Kp);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& x : h) x = f2bf(rnd());
    bf16_t *corpus, *queries; float* out;
    CK(hipMalloc(&corpus, crow * Kp * 2));
    CK(hipMemcpy(corpus, h.data(), crow * Kp * 2, hipMemcpyHostToDevice));
    h.resize(qrow * Kp);
    for (auto& x : h) x = f2bf(rnd());
    CK(hipMalloc(&queries, qrow * Kp * 2));
    CK(hipMemcpy(queries, h.data(), qrow * Kp * 2, hipMemcpyHostToDevice));
    const int grid = nqtiles * nsplits;
    const size_t out_n = (size_t)grid * THREADS * 4;
    CK(hipMalloc(&out, out_n * 4));
    unsigned long long* clk; CK(hipMalloc(&clk, (size_t)grid * 32)); CK(hipMemset(clk, 0, (size_t)grid * 32));
    unsigned* gate; const size_t gate_bytes = (size_t)(nqtiles / 8) * nsplits * 4; CK(hipMalloc(&gate, gate_bytes));
    LabParams p{corpus, queries, out, Kp, ksteps, ntiles, (ntiles + nsplits - 1) / nsplits, nsplits, flags, clk, gate};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define LAB_GO(K) { CK(hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ALLOC)); \
                    hipLaunchKernelGGL(K, dim3(grid), dim3(THREADS), LDS_ALLOC, 0, p); }
    auto launch = [&](int var) {
        if (var == 0) LAB_GO(lab_v0)
        else if (var == 1) LAB_GO(lab_v1<0>)
        else if (var == 2) LAB_GO(lab_v1<1>)
        else if (var == 3) LAB_GO(lab_v1<3>)
        else if (var == 4) LAB_GO(lab_v2<0>)
        else if (var == 5) LAB_GO(lab_v1<5>)
        else if (var == 6) LAB_GO(lab_v2<12>)
        else if (var == 7) LAB_GO(lab_v2<20>)
        else if (var == 8) LAB_GO(lab_v1<7>)
        else if (var == 9) {
#define LAB_GO3(F) { CK(hipFuncSetAttribute((const void*)lab_v3<F>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ALLOC)); \
                     hipLaunchKernelGGL(lab_v3<F>, dim3(grid), dim3(V3_THREADS), LDS_ALLOC, 0, p); }
            switch (flags & 47) {      // bit6 (the drift gate) is read at run time
                case 32: LAB_GO3(32) break;
                case 0: LAB_GO3(0) break; case 1: LAB_GO3(1) break; case 2: LAB_GO3(2) break; case 4: LAB_GO3(4) break;
                case 8: LAB_GO3(8) break; case 9: LAB_GO3(9) break; case 5: LAB_GO3(5) break;
                default: printf("variant 9: flags %d not instantiated\n", flags); exit(1);
            }
        }
        else { printf("unknown variant\n"); exit(1); }
    };
    // per-query maximum over the whole corpus, reduced on the host: out[wg][wave][lane][nt]
    auto reduce = [&](std::vector<float>& q, int var) {
        const int T = var == 9 ? V3_THREADS : THREADS, NT = var == 9 ? 8 : 4, WN = var == 9 ? 1 : 3;
        std::vector<float> ho(out_n);
        CK(hipMemcpy(ho.data(), out, out_n * 4, hipMemcpyDeviceToHost));
        q.assign(qrow, -3e38f);
        for (int b = 0; b < grid; ++b) {
            // invert xcd_remap on the host
            const int nwg = grid, qq = nwg >> 3, r = nwg & 7, x = b & 7;
            const int base = x < r ? x * (qq + 1) : r * (qq + 1) + (x - r) * qq;
            const int v = base + (b >> 3), qt = v / nsplits;
            for (int t = 0; t < T; ++t) {
                const int wave = t >> 6, lane = t & 63, wn = wave & WN, fr = lane & 15;
                for (int nt = 0; nt < NT; ++nt) {
                    float& d = q[(size_t)qt * TILE_N + wn * (NT * 16) + nt * 16 + fr];
                    const float val = ho[((size_t)b * T + t) * NT + nt];
                    if (val > d) d = val;
                }
            }
        }
    };
    std::vector<float> ref, got;
    const double flop = 2.0 * qrow * (double)ntiles * TILE_M * Kp;
    const double fill = (double)grid * p.tiles_per_split * ksteps * 65536.0;
    for (int rep = 0; rep < reps; ++rep) {
        CK(hipMemset(gate, 0, gate_bytes));
        CK(hipEventRecord(e0));
        launch(variant);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("variant %d nsplits %d flags %d ntiles %d: %.2f ms  %.1f TFLOP/s  fill %.1f GB/s per CU\n", variant, nsplits, flags,
               ntiles, ms, flop / ms / 1e9, fill / ms / 1e6 / 256);
    }
    if (variant != 0) {      // in-kernel clock: s_memtime ticks per 100 MHz s_memrealtime tick, median over workgroups
        std::vector<unsigned long long> hc((size_t)grid * 4);
        CK(hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> f;
        for (int b = 0; b < grid; ++b) if (hc[b * 4 + 3] > hc[b * 4 + 1]) f.push_back((double)(hc[b * 4 + 2] - hc[b * 4]) / (double)(hc[b * 4 + 3] - hc[b * 4 + 1]) * 0.1);
        if (!f.empty()) { std::sort(f.begin(), f.end()); printf("  in-kernel clock: median %.3f GHz (min %.3f, max %.3f); cycles per K-step %.0f\n", f[f.size() / 2], f[0], f.back(),
                                 (double)(hc[2] - hc[0]) / ((double)p.tiles_per_split * ksteps)); }
    }
    if ((flags == 0 || flags == 128) && variant != 0 && ntiles <= 600) {      // check against variant 0 (same operands; flags 0: same accumulation order, bit for bit; 128: the K order differs, to rounding)
        reduce(got, variant);
        CK(hipMemset(out, 0, out_n * 4));
        CK(hipMemset(gate, 0, gate_bytes));
        launch(0); CK(hipDeviceSynchronize());
        reduce(ref, 0);
        size_t bad = 0;
        for (size_t i = 0; i < qrow; ++i) if (flags == 128 ? (fabsf(ref[i] - got[i]) > 1e-3f * fabsf(ref[i])) : (ref[i] != got[i])) { if (bad < 5) printf("  mismatch q %zu: %g vs %g\n", i, got[i], ref[i]); ++bad; }
        printf("check vs variant 0: %zu of %zu queries differ\n", bad, qrow);
        if (bad) {      // which 16-query blocks of a tile, and how many tiles
            int blk[16] = {0}; size_t tiles = 0;
            for (size_t t = 0; t < qrow / TILE_N; ++t) { bool any = false; for (int c = 0; c < TILE_N; ++c) if (ref[t * TILE_N + c] != got[t * TILE_N + c]) { ++blk[c >> 4]; any = true; } tiles += any; }
            printf("  per 16-query block:"); for (int i = 0; i < 16; ++i) printf(" %d", blk[i]); printf("  (%zu query tiles affected)\n", tiles);
        }
    }
    return 0;
}
