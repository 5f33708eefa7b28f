// attn_fwd_pipe.h -- attention forward on the matrix cores, software-pipelined.  AN EXPERIMENT THAT LOST, kept out of the
// product (not included by nn_ops.hip): correct (tests/test_predictor_gpu.py and tools/attn_fuzz.py passed with it as the
// default path) but slower than the plain schedule -- 69 vs 59 us at B32 H12 512x512, 39 vs 35 us at 160x512 (stamped
// builds, tools/attn_lab.hip).  Hiding the score MFMAs under the exponentials saves a wave ~256 of ~1200 cycles per
// tile; the second score accumulator costs the third wave per SIMD that was covering the stalls.  DESIGN.md section 6.3.
//
// Same math, layouts and LDS geometry as attention_fwd_mfma_kernel (nn_ops.hip: S^T = K Q^T with 32x32x16 MFMAs, a lane
// owns one query, probabilities feed O^T += V^T P^T from registers), different schedule.  Measured on the plain kernel
// (tools/attn_lab.hip, B32 H12 512x512): a key tile costs a wave ~700 cycles of vector instructions (32 exp at 8
// cycles, ~110 others at 4) and 512 cycles of MFMA, and the SIMD spent their SUM plus ~600 cycles of stalls per tile:
// QK MFMAs -> softmax -> PV MFMAs is one dependent chain per wave, and with 3 waves per SIMD nothing covers it.
// Here the score MFMAs of tile j+1 are issued INSIDE the exponential phase of tile j (one basic block, interleaved
// with sched_group_barrier), so a wave's matrix work runs under its own vector work:
//
//   iteration j:  wait tile j+1 | barrier | DMA tile j+3 | K fragments (j+1) | V^T reads (j)
//                 max / rescale (j)                              [vector]
//                 exp, pack, row sum (j)  ||  S(j+1) = K Q^T     [vector || matrix]
//                 O += V^T P^T (j)                               [matrix; the SIMD's other wave has the vector pipe]
//
// Cost: a second score accumulator (32 VGPRs -> ~200, 2 waves per SIMD) and deeper rings: K tiles in a ring of 3 (tile j+1
// read, j+2 in flight, j+3 being written), V tiles in a ring of 4 (V(j) is read one iteration after K(j)): 56 KiB + the
// key mask, 2 workgroups per CU.  Key masks up to 1024 keys (one mask chunk); longer ones and full [Lq, Lk] masks take the
// plain kernel.  Reference: transformers BertSelfAttention.forward as used by textreact/model.py:10-37.
#pragma once

template <int MM, bool DROP>   // MM: TRX_NN_MASK_NONE or TRX_NN_MASK_KEY
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void attention_fwd_pipe_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                               const float* __restrict__ mask, int causal, int B, int H, int Lq, int Lk, float scale,
                               bf16_t* __restrict__ out, float* __restrict__ lse, DropArgs da) {
    __shared__ __attribute__((aligned(128))) char lds[7 * 8192];    // K tiles [3][8 KiB], then V tiles [4][8 KiB]
    __shared__ __attribute__((aligned(16))) float ldsM[1024];       // key mask / scale, clamped
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void gbl_void;
    const int tid = threadIdx.x, lane = tid & 63;
    TRX_STAMP(0, __builtin_amdgcn_s_memrealtime()); TRX_STAMP(2, __builtin_amdgcn_s_memtime());
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int nqb = (Lq + 127) / 128;
    int bid = blockIdx.x;
    {   // query blocks of one (batch, head) neighbours on one XCD (shared K/V in its L2)
        const int nwg = gridDim.x, per = nwg >> 3, main_ = per << 3;
        if (bid < main_) bid = (bid & 7) * per + (bid >> 3);
    }
    const int qb = bid % nqb, h = (bid / nqb) % H, b = bid / (nqb * H);
    const int qidx = qb * 128 + wave * 32 + r;
    const int qc = qidx < Lq ? qidx : Lq - 1;
    bf16x8 qf[4];
    {
        const bf16_t* qp = q + ((int64_t)b * Lq + qc) * (da.ldq ? da.ldq : H * 64) + h * 64 + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    }
    f32x16 o0, o1;
#pragma unroll
    for (int t = 0; t < 16; ++t) { o0[t] = 0.f; o1[t] = 0.f; }
    const float sl2 = scale * 1.44269504088896340736f;
    const float inv_scale = 1.0f / scale;
    float m = -__builtin_inff(), lsum = 0.f;
    const int off = Lk - Lq;
    int nkb = (Lk + 63) / 64;
    if (causal) {
        const int lastq = min(Lq - 1, qb * 128 + 127);
        nkb = min(nkb, (lastq + off) / 64 + 1);
    }
    const int klim = causal ? min(Lk - 1, qidx + off) : Lk - 1;
    const int klim_wave_min = causal ? min(Lk - 1, qb * 128 + wave * 32 + off) : Lk - 1;

    // LDS-DMA geometry as in the plain kernel: a piece = 8 rows x 128 B, wave w moves pieces 2w, 2w+1 of K and of V
    const int prow = lane >> 3, pslot = lane & 7;
    const unsigned rowbytes = (unsigned)(da.ldk ? da.ldk : H * 64) * 2u;
    const int64_t kvbs = da.kv_bs ? da.kv_bs : (int64_t)Lk * (da.ldk ? da.ldk : H * 64);
    const char* kbase = reinterpret_cast<const char*>(k + (int64_t)b * kvbs + h * 64);
    const char* vbase = reinterpret_cast<const char*>(v + (int64_t)b * kvbs + h * 64);
    unsigned kofs[2], vofs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const unsigned rw = (unsigned)(8 * (2 * wave + i) + prow);
        kofs[i] = rw * rowbytes + (unsigned)((pslot ^ (4 * i + (prow >> 1))) * 16);
        vofs[i] = rw * rowbytes + (unsigned)((pslot ^ ((prow & 3) << 1)) * 16);
    }
    constexpr int VRING = 3 * 8192;
    auto stage = [&](int kb, int kbuf) {     // K ring slot kbuf = kb % 3, V ring slot kb & 3
        const char* kt_ = kbase + (int64_t)kb * 64 * rowbytes;
        const char* vt_ = vbase + (int64_t)kb * 64 * rowbytes;
        char* dst = lds + kbuf * 8192 + 2 * wave * 1024;
        char* dstv = lds + VRING + (kb & 3) * 8192 + 2 * wave * 1024;
        if (kb * 64 + 64 <= Lk) {
#pragma unroll
            for (int i_ = 0; i_ < 2; ++i_) {
                __builtin_amdgcn_global_load_lds((gbl_void*)(kt_ + kofs[i_]), (lds_void*)(dst + i_ * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_void*)(vt_ + vofs[i_]), (lds_void*)(dstv + i_ * 1024), 16, 0, 0);
            }
        } else {   // tail tile: rows past the last key re-read the last key (hidden anyway)
#pragma unroll
            for (int i_ = 0; i_ < 2; ++i_) {
                const int rw_ = min(8 * (2 * wave + i_) + prow, Lk - 1 - kb * 64);
                const unsigned ro_ = (unsigned)rw_ * rowbytes;
                __builtin_amdgcn_global_load_lds((gbl_void*)(kt_ + (ro_ + (unsigned)((pslot ^ (4 * i_ + (prow >> 1))) * 16))),
                                                 (lds_void*)(dst + i_ * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_void*)(vt_ + (ro_ + (unsigned)((pslot ^ ((prow & 3) << 1)) * 16))),
                                                 (lds_void*)(dstv + i_ * 1024), 16, 0, 0);
            }
        }
    };
    const unsigned ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const unsigned vtrA0 = ldsbase + (unsigned)(VRING + (4 * (g >> 1) + qq) * 128 + (((2 * (g & 1) + (pp >> 1)) ^ (qq << 1)) << 4) + 8 * (pp & 1));
    const int kswz = (r >> 1) & 7;
    unsigned kfa[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) kfa[s] = ldsbase + (unsigned)(r * 128 + (((2 * s + hh) ^ kswz) << 4));
    constexpr bool keymask = MM == TRX_NN_MASK_KEY;
    const unsigned xdrop = DROP ? drop_base(da.seed_lo, da.seed_hi, (unsigned)(b * H + h)) + (unsigned)qidx * DROP_C1 + (unsigned)(2 * hh) * DROP_C2 : 0u;

    float mv_[4] = {0.f, 0.f, 0.f, 0.f};
    if (keymask) {   // all of the key mask (Lk <= 1024), 4 keys per thread
        const float* mkey = mask + (int64_t)b * Lk;
#pragma unroll
        for (int i_ = 0; i_ < 4; ++i_) mv_[i_] = mkey[min(4 * tid + i_, Lk - 1)];
    }
    stage(0, 0);
    if (nkb > 1) stage(1, 1);
    if (nkb > 2) stage(2, 2);
    if (keymask) {
#pragma unroll
        for (int i_ = 0; i_ < 4; ++i_) ldsM[4 * tid + i_] = fmaxf(mv_[i_] * inv_scale, -1e30f);
    }
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));   // retire the Q loads here

    // scores of a tile start from the additive key mask (raw-score units) or 0
    auto init_scores = [&](f32x16& a0, f32x16& a1, int kb) {
        if (keymask) {
            const float* mt = ldsM + kb * 64 + 4 * hh;
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const float4 a = *reinterpret_cast<const float4*>(mt + 8 * t4);
                const float4 c = *reinterpret_cast<const float4*>(mt + 32 + 8 * t4);
                a0[4 * t4] = a.x; a0[4 * t4 + 1] = a.y; a0[4 * t4 + 2] = a.z; a0[4 * t4 + 3] = a.w;
                a1[4 * t4] = c.x; a1[4 * t4 + 1] = c.y; a1[4 * t4 + 2] = c.z; a1[4 * t4 + 3] = c.w;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) { a0[t] = 0.f; a1[t] = 0.f; }
        }
    };
    bf16x8 ka[4][2];
    // K fragments through inline asm (a C++ read of `lds` would make the compiler drain vmcnt first)
#define TRXP_K_READ(BUF)                                                                                      \
    {                                                                                                         \
        const unsigned kb0_ = (unsigned)((BUF) * 8192);                                                      \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                                      \
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096"                             \
                         : "=&v"(ka[s_][0]), "=&v"(ka[s_][1]) : "v"(kfa[s_] + kb0_) : "memory");              \
    }
#define TRXP_K_WAIT(CNT)                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")"                                                                \
                 : "+v"(ka[0][0]), "+v"(ka[0][1]), "+v"(ka[1][0]), "+v"(ka[1][1]),                            \
                   "+v"(ka[2][0]), "+v"(ka[2][1]), "+v"(ka[3][0]), "+v"(ka[3][1]) :: "memory");
    uint2 vt[4][2][2];
#define TRXP_VT_READ(S)                                                                                       \
    asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%6\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\t"            \
                 "ds_read_b64_tr_b16 %2, %5 offset:%6\n\tds_read_b64_tr_b16 %3, %5 offset:%7"                 \
                 : "=&v"(vt[S][0][0]), "=&v"(vt[S][0][1]), "=&v"(vt[S][1][0]), "=&v"(vt[S][1][1])             \
                 : "v"(vtrA), "v"(vtrB), "n"((S) * 2048), "n"((S) * 2048 + 1024) : "memory");
#define TRXP_VT_WAIT(S0, S1, CNT)                                                                             \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")"                                                                \
                 : "+v"(vt[S0][0][0]), "+v"(vt[S0][0][1]), "+v"(vt[S0][1][0]), "+v"(vt[S0][1][1]),            \
                   "+v"(vt[S1][0][0]), "+v"(vt[S1][0][1]), "+v"(vt[S1][1][0]), "+v"(vt[S1][1][1]) :: "memory");
#define TRXP_PV_STEP(S, HB)                                                                                   \
    {                                                                                                         \
        constexpr int ss = (S) & 1;                                                                           \
        const bf16x8 pf = __builtin_bit_cast(bf16x8, uint4{pk[HB][4 * ss], pk[HB][4 * ss + 1], pk[HB][4 * ss + 2], pk[HB][4 * ss + 3]}); \
        uint4 v0; v0.x = vt[S][0][0].x; v0.y = vt[S][0][0].y; v0.z = vt[S][0][1].x; v0.w = vt[S][0][1].y;     \
        uint4 v1; v1.x = vt[S][1][0].x; v1.y = vt[S][1][0].y; v1.z = vt[S][1][1].x; v1.w = vt[S][1][1].y;     \
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0), pf, o0, 0, 0, 0);        \
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1), pf, o1, 0, 0, 0);        \
    }

    // ---- prologue: S(0) ----
    f32x16 sA0, sA1, sB0, sB1;
    if (nkb > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nkb > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();            // tile 0 and the key mask are in LDS for everyone (only the first wait of the kernel drains)
    init_scores(sA0, sA1, 0);
    TRXP_K_READ(0)
    TRXP_K_WAIT(0)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        sA0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s][0], qf[s], sA0, 0, 0, 0);
        sA1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s][1], qf[s], sA1, 0, 0, 0);
    }
    TRX_STAMP(3, __builtin_amdgcn_s_memtime());

    // one key tile: CUR holds S(kb); NXT receives S(kb + 1)
    int k3 = 0;                                 // kb % 3
    auto tile = [&](f32x16& c0, f32x16& c1, f32x16& n0, f32x16& n1, int kb) {
        const bool has_next = kb + 1 < nkb;     // wave-uniform
        const int k3n = k3 == 2 ? 0 : k3 + 1;   // (kb + 1) % 3
        if (has_next) {
            // tile kb + 1 has landed for this wave when only the group of tile kb + 2 may still fly
            if (kb + 2 < nkb) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();       // ... for everyone; and K slot kb % 3 (read last iteration) and V slot
            asm volatile("" ::: "memory");        // (kb - 1) & 3 (read at the end of the last iteration) are free
            if (kb + 3 < nkb) stage(kb + 3, k3);
            init_scores(n0, n1, kb + 1);
            TRXP_K_READ(k3n)
        }
        const unsigned vtrA = vtrA0 + (unsigned)((kb & 3) * 8192), vtrB = vtrA ^ 64u;
        TRXP_VT_READ(0) TRXP_VT_READ(1)
        // ---- max, running maximum, rescale ----
        const int key0 = kb * 64;
        const bool vis = key0 + 63 > klim_wave_min;
        float mb = -__builtin_inff();
        if (vis) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int kr_ = hb * 32 + (t & 3) + 8 * (t >> 2) + 4 * hh;
                    float val = hb ? c1[t] : c0[t];
                    val = (key0 + kr_ > klim) ? -__builtin_inff() : val;
                    if (hb) c1[t] = val; else c0[t] = val;
                    mb = fmaxf(mb, val);
                }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) mb = fmaxf(mb, fmaxf(c0[t], c1[t]));
        }
        mb = fmaxf(mb, __shfl_xor(mb, 32, 64)) * sl2;
        const float mn = fmaxf(m, mb);
        const float mref = (mn == -__builtin_inff()) ? 0.f : mn;
        const float alpha = __builtin_amdgcn_exp2f(m - mref);
        const float nref = -mref;
        m = mn;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int t = 0; t < 16; ++t) { o0[t] *= alpha; o1[t] *= alpha; }
        }
        // ---- exponentials, pack, row sum  ||  S(kb + 1) ----
        unsigned pk[2][8];
        float ps = 0.f;
        auto exp_phase = [&]() {
#pragma unroll
            for (int t = 0; t < 16; ++t) c0[t] = __builtin_amdgcn_exp2f(__builtin_fmaf(c0[t], sl2, nref));
#pragma unroll
            for (int t = 0; t < 16; ++t) c1[t] = __builtin_amdgcn_exp2f(__builtin_fmaf(c1[t], sl2, nref));
            if (DROP) {
                const unsigned xd = xdrop + (unsigned)(kb * 32) * DROP_C2;
#pragma unroll
                for (int t = 0; t < 16; ++t) ps += c0[t] + c1[t];
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int t = 0; t < 16; t += 2) {
                        const unsigned bits = lowbias32(xd + (unsigned)(hb * 16 + ((t & 3) >> 1) + 4 * (t >> 2)) * DROP_C2);
                        const float e0 = hb ? c1[t] : c0[t], e1 = hb ? c1[t + 1] : c0[t + 1];
                        pk[hb][t >> 1] = pack2bf(drop_keep(bits, 0, da.thr) ? e0 : 0.f, drop_keep(bits, 1, da.thr) ? e1 : 0.f);
                    }
            } else {
                const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int t = 0; t < 16; t += 2) {
                        const unsigned w = pack2bf(hb ? c1[t] : c0[t], hb ? c1[t + 1] : c0[t + 1]);
                        pk[hb][t >> 1] = w;
                        ps = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), ones, ps, false);
                    }
            }
        };
        if (has_next) {
            TRXP_K_WAIT(8)      // the K fragments are back (the 8 V^T reads issued after them may still fly)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s][0], qf[s], n0, 0, 0, 0);
                n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s][1], qf[s], n1, 0, 0, 0);
            }
            exp_phase();
            // one MFMA, then a share of the vector work, eight times; what is left follows
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, DROP ? 30 : 12, 0);
            }
        } else {
            exp_phase();
        }
        lsum = lsum * alpha + ps;
        // ---- O^T += V^T P^T ----
        TRXP_VT_READ(2) TRXP_VT_READ(3)
        TRXP_VT_WAIT(0, 1, 8)
        TRXP_PV_STEP(0, 0) TRXP_PV_STEP(1, 0)
        TRXP_VT_WAIT(2, 3, 0)
        TRXP_PV_STEP(2, 1) TRXP_PV_STEP(3, 1)
        TRX_STAMP(4 + (kb < 26 ? kb : 26), __builtin_amdgcn_s_memtime());
        k3 = k3n;
    };
    for (int kb = 0; kb < nkb; kb += 2) {
        tile(sA0, sA1, sB0, sB1, kb);
        if (kb + 1 < nkb) tile(sB0, sB1, sA0, sA1, kb + 1);
    }
#undef TRXP_K_READ
#undef TRXP_K_WAIT
#undef TRXP_VT_READ
#undef TRXP_VT_WAIT
#undef TRXP_PV_STEP
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
    TRX_STAMP(31, __builtin_amdgcn_s_memtime()); TRX_STAMP(1, __builtin_amdgcn_s_memrealtime());
}
