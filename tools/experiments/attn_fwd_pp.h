// attn_fwd_pp.h -- attention forward on the matrix cores, two wave groups in ping-pong (included by nn_ops.hip).
//
// The same math, fragment layouts and LDS images as attention_fwd_mfma_kernel (S^T = K Q^T with
// v_mfma_f32_32x32x16_bf16, the query on the lane, P straight from the accumulators into O^T += V^T P^T), but a
// different schedule.  That kernel's wave runs QK^T -> softmax -> PV as one dependent chain per key tile, and the three
// waves a SIMD holds (from three unrelated workgroups) overlap each other's matrix and vector work only by accident:
// the timeline shows a tile-wave costing the SUM of its matrix (512 cycles) and vector (~700) time, not the maximum
// (DESIGN.md 6.3).  Here the overlap is the structure:
//
//   workgroup = 8 waves = 256 queries of one (batch, head); waves 0-3 (group A) own queries 0-127, waves 4-7 (group B)
//   queries 128-255; the two waves of a SIMD belong to different groups and run ONE barrier interval apart:
//
//     interval      group A (waves 0-3)                                   group B (waves 4-7)
//     2j+1          M(j): S_j = K_j Q^T  and  O += V_{j-1}^T P_{j-1}      V(j-1): softmax(S_{j-1}) -> P_{j-1}, fragment reads
//     2j+2          V(j): softmax(S_j) -> P_j, fragment reads, DMA        M(j)
//
//   An M interval is 16 MFMAs and nothing else; a V interval is the tile's whole vector work plus the LDS reads of the
//   fragments the wave's next M interval consumes (K_{j+1}, V_j^T) and -- group A only -- the LDS-DMA of tile j + 3.  So
//   on every SIMD one wave feeds the matrix pipe while its partner feeds the vector pipe, by construction.
//   (cdna guide: "Two waves per SIMD", the 8-wave attention structure.)
//
// Key tiles of 64 in a ring of FOUR 16 KiB buffers [K 8 KiB | V 8 KiB]: tile t is DMA'd in interval 2t - 4 (A's V(t-3)),
// retired by A's vmcnt(4) at the end of interval 2t - 2, first read in interval 2t (A's V(t-1): K_t fragments), last read
// in interval 2t + 3 (B's V(t): V_t^T fragments), and its buffer is written again in interval 2t + 4 (tile t + 4).  Every
// fragment read is retired (lgkmcnt(0)) before the barrier that ends its interval.
//
// For Lq >= 256 and Lk <= 1024 (the encoder's self-attention; one workgroup per CU, 68 KiB of dynamic LDS); mask modes
// NONE / KEY, with and without dropout.
//
// MEASURED (round 3, B 32 x H 12 x 512 x 512, same box, interleaved): 69.4 us against 52.0 us for attention_fwd_mfma_kernel --
// SLOWER, so the launcher uses it only under TRX_NN_ATTN_PP=1.  Why: at head size 64 the vector side of a key tile (the
// softmax's ~105 instructions, 24 + 8 LDS reads with their waits, the DMA issue) takes about twice the 512 matrix cycles,
// so the V interval, not the M interval, paces every barrier and the matrix pipe idles half of each period; with the
// fragments of the next M interval and the softmax state both live a wave needs 231 registers, so there is no third
// wave to fill that.  The single-chain kernel at three waves per SIMD already sits within ~15 % of the same vector-issue
// bound (1,270 cycles per tile-wave measured against ~1,100 of vector issue), which is what bounds attention at this head
// size on this chip: fewer vector instructions per score, not a different overlap, is what would move it.
// Two compiler traps cost most of the bring-up and are kept from recurring by the form of the code: (i) an asm load and
// its wait must be ONE statement (hipcc copied fragment registers ahead of a wait placed in a later statement);
// (ii) asm outputs of the struct type uint2 did not survive the loop's back edge (the in-loop MFMAs read zeros) -- native
// vector types do.
#pragma once
constexpr int PP_LDS_BYTES = 4 * 16384 + 4096;

template <int MM, bool DROP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention_fwd_pp_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, const float* __restrict__ mask,
    int causal, int B, int H, int Lq, int Lk, float scale, bf16_t* __restrict__ out, float* __restrict__ lse, DropArgs da) {
    // dynamic LDS (68 KiB is past the 64 KiB a kernel may declare statically; the launcher raises the limit): the ring of
    // four key tiles, then the key mask of <= 1024 keys, pre-divided by the scale
    extern __shared__ __attribute__((aligned(128))) char pp_smem[];
    char* const lds = pp_smem;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void gbl_void;
    typedef __attribute__((ext_vector_type(4))) float f32x4_pp;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int r = lane & 31, hh = lane >> 5;
    const int nqb = (Lq + 255) / 256;
    int bid = blockIdx.x;
    {   // the query blocks of one (batch, head) on ONE XCD (they re-read the same K / V through its L2)
        const int nwg = gridDim.x, per = nwg >> 3, main_ = per << 3;
        if (bid < main_) bid = (bid & 7) * per + (bid >> 3);
    }
    const int qb = bid % nqb, h = (bid / nqb) % H, b = bid / (nqb * H);
    const int qwave0 = qb * 256 + grp * 128 + w4 * 32;       // first query of this wave
    const int qidx = qwave0 + r;
    const int qc = qidx < Lq ? qidx : Lq - 1;
    bf16x8 qf[4];
    {
        const bf16_t* qp = q + ((int64_t)b * Lq + qc) * (da.ldq ? da.ldq : H * 64) + h * 64 + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    }
    f32x16 o0, o1, s0, s1;
#pragma unroll
    for (int t = 0; t < 16; ++t) { o0[t] = 0.f; o1[t] = 0.f; }
    const float sl2 = scale * 1.44269504088896340736f;
    const float inv_scale = 1.0f / scale;
    float m = -__builtin_inff(), lsum = 0.f;
    const int off = Lk - Lq;
    int nkb = (Lk + 63) / 64;
    if (causal) {
        const int lastq = min(Lq - 1, qb * 256 + 255);
        nkb = min(nkb, (lastq + off) / 64 + 1);
    }
    const int klim = causal ? min(Lk - 1, qidx + off) : Lk - 1;
    const int klim_wave_min = causal ? min(Lk - 1, qwave0 + off) : Lk - 1;

    // ---- LDS-DMA geometry: as attention_fwd_mfma_kernel, issued by group A's waves (w4 moves pieces 2 w4, 2 w4 + 1 of K and of V)
    const int prow = lane >> 3, pslot = lane & 7;
    const unsigned rowbytes = (unsigned)(da.ldk ? da.ldk : H * 64) * 2u;
    const int64_t kvbs = da.kv_bs ? da.kv_bs : (int64_t)Lk * (da.ldk ? da.ldk : H * 64);
    const char* kbase = reinterpret_cast<const char*>(k + (int64_t)b * kvbs + h * 64);
    const char* vbase = reinterpret_cast<const char*>(v + (int64_t)b * kvbs + h * 64);
    unsigned kofs[2], vofs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const unsigned rw = (unsigned)(8 * (2 * w4 + i) + prow);
        kofs[i] = rw * rowbytes + (unsigned)((pslot ^ (4 * i + (prow >> 1))) * 16);
        vofs[i] = rw * rowbytes + (unsigned)((pslot ^ ((prow & 3) << 1)) * 16);
    }
#define PP_STAGE(KB)                                                                                        \
    {                                                                                                       \
        const int buf_ = (KB) & 3;                                                                          \
        const char* kt_ = kbase + (int64_t)(KB) * 64 * rowbytes;                                            \
        const char* vt_ = vbase + (int64_t)(KB) * 64 * rowbytes;                                            \
        if ((KB) * 64 + 64 <= Lk) {                                                                         \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                              \
                __builtin_amdgcn_global_load_lds((gbl_void*)(kt_ + kofs[i_]),                               \
                                                 (lds_void*)(lds + buf_ * 16384 + (2 * w4 + i_) * 1024), 16, 0, 0);        \
                __builtin_amdgcn_global_load_lds((gbl_void*)(vt_ + vofs[i_]),                               \
                                                 (lds_void*)(lds + buf_ * 16384 + 8192 + (2 * w4 + i_) * 1024), 16, 0, 0); \
            }                                                                                               \
        } else { /* tail tile: rows past the last key re-read the last key (they are hidden anyway) */      \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                              \
                const int rw_ = min(8 * (2 * w4 + i_) + prow, Lk - 1 - (KB) * 64);                          \
                const unsigned ro_ = (unsigned)rw_ * rowbytes;                                              \
                __builtin_amdgcn_global_load_lds((gbl_void*)(kt_ + (ro_ + (unsigned)((pslot ^ (4 * i_ + (prow >> 1))) * 16))), \
                                                 (lds_void*)(lds + buf_ * 16384 + (2 * w4 + i_) * 1024), 16, 0, 0);        \
                __builtin_amdgcn_global_load_lds((gbl_void*)(vt_ + (ro_ + (unsigned)((pslot ^ ((prow & 3) << 1)) * 16))), \
                                                 (lds_void*)(lds + buf_ * 16384 + 8192 + (2 * w4 + i_) * 1024), 16, 0, 0); \
            }                                                                                               \
        }                                                                                                   \
    }
    const unsigned ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned ldsMbase = ldsbase + 65536u;
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const unsigned vtrA0 = ldsbase + (unsigned)(8192 + (4 * (g >> 1) + qq) * 128 + (((2 * (g & 1) + (pp >> 1)) ^ (qq << 1)) << 4) + 8 * (pp & 1));
    const int kswz = (r >> 1) & 7;
    unsigned kfa[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) kfa[s] = ldsbase + (unsigned)(r * 128 + (((2 * s + hh) ^ kswz) << 4));
    constexpr bool keymask = MM == TRX_NN_MASK_KEY;
    const unsigned xdrop = DROP ? drop_base_da(da, (unsigned)(b * H + h)) + (unsigned)qidx * DROP_C1 + (unsigned)(2 * hh) * DROP_C2 : 0u;

    // ---- prologue: the key mask of all <= 1024 keys (its loads first: they are consumed first), tiles 0, 1, 2 in flight (group A) ----
    float mv_[4] = {0.f, 0.f, 0.f, 0.f};
    if (keymask && tid < 256) {
        const float* mkey = mask + (int64_t)b * Lk;
#pragma unroll
        for (int i_ = 0; i_ < 4; ++i_) mv_[i_] = mkey[min(4 * tid + i_, Lk - 1)];
    }
    if (!grp) {
        PP_STAGE(0);
        if (nkb > 1) PP_STAGE(1);
        if (nkb > 2) PP_STAGE(2);
    }
    if (keymask && tid < 256) {     // (asm: a C++ store to the DMA's array would make hipcc drain vmcnt first)
#pragma unroll
        for (int i_ = 0; i_ < 4; ++i_) mv_[i_] = fmaxf(mv_[i_] * inv_scale, -268435456.0f / (scale * 1.44269504088896340736f));
        asm volatile("ds_write_b128 %0, %1" :: "v"(ldsMbase + (unsigned)(16 * tid)), "v"(*reinterpret_cast<const f32x4_pp*>(mv_)) : "memory");
    }
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));
    // tile 0 landed (group A: all but the pieces of tiles 1 and 2), mask written: visible to everyone after the barrier
    if (!grp) {
        if (nkb > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (nkb > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (grp) __builtin_amdgcn_s_barrier();       // group B runs one interval late

    bf16x8 ka[4][2];
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_pp;     // (a native vector type: asm outputs of the struct type uint2 did not survive the loop back edge)
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_pp;
    u32x2_pp vt[4][2][2];
    unsigned pk[2][8];
    // Fragment reads through inline asm (a C++ read of `lds` would make hipcc drain vmcnt: the DMA writes that array), and
    // every group of reads TOGETHER WITH ITS WAIT in one statement: hipcc counts an asm load's destination as written at
    // the end of the statement and is free to copy it before a wait placed in a later statement -- it did (register copies
    // of the K fragments ahead of the s_waitcnt lgkmcnt(0), on the branch where two copies of the wait met: garbage scores
    // whenever the reads had not landed yet).
#define PP_READ_K(T)                                                                                          \
    {                                                                                                         \
        const unsigned kb0_ = (unsigned)(((T) & 3) * 16384);                                                  \
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:4096\n\t"                            \
                     "ds_read_b128 %2, %9\n\tds_read_b128 %3, %9 offset:4096\n\t"                            \
                     "ds_read_b128 %4, %10\n\tds_read_b128 %5, %10 offset:4096\n\t"                          \
                     "ds_read_b128 %6, %11\n\tds_read_b128 %7, %11 offset:4096\n\t"                          \
                     "s_waitcnt lgkmcnt(0)"                                                                   \
                     : "=&v"(ka[0][0]), "=&v"(ka[0][1]), "=&v"(ka[1][0]), "=&v"(ka[1][1]),                    \
                       "=&v"(ka[2][0]), "=&v"(ka[2][1]), "=&v"(ka[3][0]), "=&v"(ka[3][1])                     \
                     : "v"(kfa[0] + kb0_), "v"(kfa[1] + kb0_), "v"(kfa[2] + kb0_), "v"(kfa[3] + kb0_) : "memory"); \
    }
#define PP_READ_V(T)                                                                                          \
    {                                                                                                         \
        const unsigned vtrA_ = vtrA0 + (unsigned)(((T) & 3) * 16384), vtrB_ = vtrA_ ^ 64u;                    \
        asm volatile("ds_read_b64_tr_b16 %0, %16 offset:0\n\tds_read_b64_tr_b16 %1, %16 offset:1024\n\t"     \
                     "ds_read_b64_tr_b16 %2, %17 offset:0\n\tds_read_b64_tr_b16 %3, %17 offset:1024\n\t"     \
                     "ds_read_b64_tr_b16 %4, %16 offset:2048\n\tds_read_b64_tr_b16 %5, %16 offset:3072\n\t"  \
                     "ds_read_b64_tr_b16 %6, %17 offset:2048\n\tds_read_b64_tr_b16 %7, %17 offset:3072\n\t"  \
                     "ds_read_b64_tr_b16 %8, %16 offset:4096\n\tds_read_b64_tr_b16 %9, %16 offset:5120\n\t"  \
                     "ds_read_b64_tr_b16 %10, %17 offset:4096\n\tds_read_b64_tr_b16 %11, %17 offset:5120\n\t" \
                     "ds_read_b64_tr_b16 %12, %16 offset:6144\n\tds_read_b64_tr_b16 %13, %16 offset:7168\n\t" \
                     "ds_read_b64_tr_b16 %14, %17 offset:6144\n\tds_read_b64_tr_b16 %15, %17 offset:7168\n\t" \
                     "s_waitcnt lgkmcnt(0)"                                                                   \
                     : "=&v"(vt[0][0][0]), "=&v"(vt[0][0][1]), "=&v"(vt[0][1][0]), "=&v"(vt[0][1][1]),        \
                       "=&v"(vt[1][0][0]), "=&v"(vt[1][0][1]), "=&v"(vt[1][1][0]), "=&v"(vt[1][1][1]),        \
                       "=&v"(vt[2][0][0]), "=&v"(vt[2][0][1]), "=&v"(vt[2][1][0]), "=&v"(vt[2][1][1]),        \
                       "=&v"(vt[3][0][0]), "=&v"(vt[3][0][1]), "=&v"(vt[3][1][0]), "=&v"(vt[3][1][1])         \
                     : "v"(vtrA_), "v"(vtrB_) : "memory");                                                    \
    }
    // the accumulators of the next S tile start from the additive mask in raw-score units (or zero): plain C++ reads of the
    // mask's own LDS array at the end of the V interval, straight into the registers of the S tile the softmax has just consumed
    auto init_s = [&](int T) __attribute__((always_inline)) {
        if (keymask) {
            f32x4_pp a[4], c[4];
            const unsigned ma = ldsMbase + (unsigned)((T * 64 + 4 * hh) * 4);
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:32\n\tds_read_b128 %2, %8 offset:64\n\tds_read_b128 %3, %8 offset:96\n\t"
                         "ds_read_b128 %4, %8 offset:128\n\tds_read_b128 %5, %8 offset:160\n\tds_read_b128 %6, %8 offset:192\n\tds_read_b128 %7, %8 offset:224\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])
                         : "v"(ma) : "memory");
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                s0[4 * t4] = a[t4][0]; s0[4 * t4 + 1] = a[t4][1]; s0[4 * t4 + 2] = a[t4][2]; s0[4 * t4 + 3] = a[t4][3];
                s1[4 * t4] = c[t4][0]; s1[4 * t4 + 1] = c[t4][1]; s1[4 * t4 + 2] = c[t4][2]; s1[4 * t4 + 3] = c[t4][3];
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) { s0[t] = 0.f; s1[t] = 0.f; }
        }
    };
    // end of a V-type interval: this wave's LDS reads are retired (inside their own statements; the mask reads by hipcc's own
    // count); group A: all DMA pieces but the four just issued (MORE) or all of them
#define PP_END_V(MORE)                                                                                        \
    if (!grp) { if (MORE) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    __builtin_amdgcn_s_barrier();                                                                             \
    __builtin_amdgcn_sched_barrier(0);
#define PP_END_M()                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    __builtin_amdgcn_s_barrier();                                                                             \
    __builtin_amdgcn_sched_barrier(0);

    // ---- L(0): K_0 fragments and the mask of tile 0 ----
    // (defined before the loop: the first M interval does not use them, but a loop-carried value that starts undefined lets
    // hipcc merge all of them into one register)
#pragma unroll
    for (int S = 0; S < 4; ++S) { vt[S][0][0] = (u32x2_pp){0u, 0u}; vt[S][0][1] = (u32x2_pp){0u, 0u}; vt[S][1][0] = (u32x2_pp){0u, 0u}; vt[S][1][1] = (u32x2_pp){0u, 0u}; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { pk[0][i] = 0u; pk[1][i] = 0u; }
    PP_READ_K(0);
    init_s(0);
    PP_END_V(nkb > 2);       // (group A) tile 1 landed: all but tile 2's pieces

    for (int j = 0; j < nkb; ++j) {
        // ================= M(j): S_j = K_j Q^T; O += V_{j-1}^T P_{j-1} =================
        __builtin_amdgcn_s_setprio(1);
        if (j > 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int hb = s >> 1, ss = s & 1;
                const bf16x8 pf = __builtin_bit_cast(bf16x8, (u32x4_pp){pk[hb][4 * ss], pk[hb][4 * ss + 1], pk[hb][4 * ss + 2], pk[hb][4 * ss + 3]});
                const u32x4_pp v0 = __builtin_shufflevector(vt[s][0][0], vt[s][0][1], 0, 1, 2, 3);
                const u32x4_pp v1 = __builtin_shufflevector(vt[s][1][0], vt[s][1][1], 0, 1, 2, 3);
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s][0], qf[s], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s][1], qf[s], s1, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0), pf, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1), pf, o1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s][0], qf[s], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s][1], qf[s], s1, 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        PP_END_M();
        // ================= V(j): softmax(S_j) -> P_j; reads for M(j+1); DMA of tile j + 3 =================
        if (!grp && j + 3 < nkb) PP_STAGE(j + 3);
        const int key0 = j * 64;
        const bool vis = key0 + 63 > klim_wave_min;
        const unsigned xd = xdrop + (unsigned)(j * 32) * DROP_C2;
        if (vis) {      // hidden keys: minus infinity in place, then the one tile form (as in attention_fwd_mfma_kernel, round 5)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int kr_ = (t & 3) + 8 * (t >> 2) + 4 * hh;
                s0[t] = (key0 + kr_ > klim) ? -__builtin_inff() : s0[t];
                s1[t] = (key0 + 32 + kr_ > klim) ? -__builtin_inff() : s1[t];
            }
        }
        attn_softmax_tile<false, DROP>(s0, s1, o0, o1, m, lsum, sl2, key0, hh, klim, xd, da.thr, pk);
        __builtin_amdgcn_sched_barrier(0);             // the fragment reads below go AFTER the softmax (they would cost it 64 registers)
        PP_READ_V(j);                                  // V_j^T: consumed by PV in M(j+1)
        if (j + 1 < nkb) {
            PP_READ_K(j + 1);
            init_s(j + 1);
        }
        PP_END_V(j + 3 < nkb);
    }
    // ================= final M: O += V_{nkb-1}^T P_{nkb-1} =================
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int hb = s >> 1, ss = s & 1;
        const bf16x8 pf = __builtin_bit_cast(bf16x8, (u32x4_pp){pk[hb][4 * ss], pk[hb][4 * ss + 1], pk[hb][4 * ss + 2], pk[hb][4 * ss + 3]});
        const u32x4_pp v0 = __builtin_shufflevector(vt[s][0][0], vt[s][0][1], 0, 1, 2, 3);
        const u32x4_pp v1 = __builtin_shufflevector(vt[s][1][0], vt[s][1][1], 0, 1, 2, 3);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0), pf, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1), pf, o1, 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (!grp) __builtin_amdgcn_s_barrier();          // group A waits for the interval group B is behind
#undef PP_STAGE
#undef PP_READ_K
#undef PP_READ_V
#undef PP_END_V
#undef PP_END_M
    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    if (qidx < Lq) {
        const float inv = (DROP ? da.inv_keep : 1.0f) / ltot;
        if (lse && hh == 0) lse[((int64_t)b * H + h) * Lq + qidx] = (m + __builtin_amdgcn_logf(ltot)) * 0.69314718055994530942f;
        bf16_t* op = out + ((int64_t)b * Lq + qidx) * H * 64 + (int64_t)h * 64;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            uint2 w0, w1;
            w0.x = pack2bf(o0[4 * gq] * inv, o0[4 * gq + 1] * inv); w0.y = pack2bf(o0[4 * gq + 2] * inv, o0[4 * gq + 3] * inv);
            w1.x = pack2bf(o1[4 * gq] * inv, o1[4 * gq + 1] * inv); w1.y = pack2bf(o1[4 * gq + 2] * inv, o1[4 * gq + 3] * inv);
            *reinterpret_cast<uint2*>(op + 8 * gq + 4 * hh) = w0;
            *reinterpret_cast<uint2*>(op + 32 + 8 * gq + 4 * hh) = w1;
        }
    }
}
