// attn_fwd_persist.h -- the encoder's shape class of attention_fwd_mfma_kernel (nn_ops.hip) as PERSISTENT workgroups: bf16,
// Lq a multiple of 128 (every 128-query block full: four 32-query waves, no key split), no causal mask, key mask or none, 128 to
// 512 keys in whole tiles of 64.  Included by nn_ops.hip behind that kernel (its helpers and macros are in scope).
//
// Why (round 5, profiles/r05_attention_timeline.txt, r05_attention_wait_pmc.json): a 128-query workgroup of the 512 x 512 launch
// lives 36 k cycles of which 5.5 k are its prologue (Q, the mask and the first two K / V tiles arrive) and 2.4 k its epilogue;
// the 1,536 workgroups run as two rounds of 768 resident ones that start together, so the prologues of a round coincide.
// Here 768 workgroups stay and take items w, w + 768, ...: the K / V ring runs on across items -- the last two tiles' staging
// slots of an item fetch the first two tiles of the next -- and the next item's Q and mask are requested at the start of the
// epilogue, behind the last MFMA, so they fly under the output stores.
// MEASURED (profiles/r05_attention_dropout_ab.json, same box, interleaved, outputs bit-identical): 39.8 against 39.9 us at 512 x 512,
// 52.3 against 52.0 with dropout -- no gain: while one of a CU's three workgroups is in its prologue the other two have the SIMDs.
// Off by default (TRX_NN_ATTN_PERSIST=1); kept with its test (tests/test_predictor_gpu.py::test_persistent_forward_kernel).
//
// Vector-memory order per wave (what the counted waits rely on): ... D(p) D(p+1) | iteration p: D(p+2) ... where p runs over
// the tiles of ALL of the workgroup's items; between an item's last iteration and the next item's first sit the C++ loads of
// the next Q / mask and the output stores (hipcc counts those itself: its wait in front of the next item's first use leaves only
// the stores outstanding, and everything older -- both staged tiles -- has landed by then).
template <int MM, bool DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TRX_ATT_WAVES, TRX_ATT_WAVES)))      // (dropout too: 150 registers without the hidden-key form)
void attention_fwd_persist_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                  const float* __restrict__ mask, int B, int H, int Lq, int Lk, float scale,
                                  bf16_t* __restrict__ out, float* __restrict__ lse, DropArgs da, int nitems) {
    __shared__ __attribute__((aligned(128))) char lds[3 * 16384];   // ring of 3: [K 8 KiB | V 8 KiB]
    __shared__ __attribute__((aligned(16))) float ldsM[1024];       // the key mask / scale of two items (512 keys each), alternating
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int nqb = Lq >> 7;
    constexpr bool keymask = MM == TRX_NN_MASK_KEY;
    const float sl2 = scale * 1.44269504088896340736f;
    const float inv_scale = 1.0f / scale;
    const float mask_floor = -268435456.0f / sl2;                   // (attention_fwd_mfma_kernel: TRX_MASK_INIT)
    const int nkb = (Lk + 63) / 64;                                 // 2 .. 8
    const int klim = Lk - 1;
    const int ldq = da.ldq ? da.ldq : H * 64;
    const int prow = lane >> 3, pslot = lane & 7;
    const unsigned rowbytes = (unsigned)(da.ldk ? da.ldk : H * 64) * 2u;
    const int64_t kvbs = da.kv_bs ? da.kv_bs : (int64_t)Lk * (da.ldk ? da.ldk : H * 64);
    unsigned kofs[2], vofs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const unsigned rw = (unsigned)(8 * (2 * wave + i) + prow);
        kofs[i] = rw * rowbytes + (unsigned)((pslot ^ (4 * i + (prow >> 1))) * 16);
        vofs[i] = rw * rowbytes + (unsigned)((pslot ^ ((prow & 3) << 1)) * 16);
    }
    const unsigned ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned lds_w = (unsigned)__builtin_amdgcn_readfirstlane((int)(ldsbase + (unsigned)(2 * wave * 1024)));
    // one key tile of the (batch, head) whose K / V rows start at KB_ / VB_ into ring slot BUF (see attention_fwd_mfma_kernel)
#define TRX_PST_STAGE(KB_, VB_, T, BUF)                                                                     \
    {                                                                                                       \
        const unsigned long long kt_ = (unsigned long long)((KB_) + (int64_t)(T) * 64 * rowbytes);          \
        const unsigned long long vt_ = (unsigned long long)((VB_) + (int64_t)(T) * 64 * rowbytes);          \
        const unsigned l_ = lds_w + (unsigned)((BUF) * 16384);                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                  \
            TRX_GLDS16(kt_, kofs[i_], l_ + i_ * 1024);                                                      \
            TRX_GLDS16(vt_, vofs[i_], l_ + 8192 + i_ * 1024);                                               \
        }                                                                                                   \
    }
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const int other_half = (lane ^ 32) << 2;
    const unsigned vtrA0 = ldsbase + (unsigned)(8192 + (4 * (g >> 1) + qq) * 128 + (((2 * (g & 1) + (pp >> 1)) ^ (qq << 1)) << 4) + 8 * (pp & 1));
    const int kswz = (r >> 1) & 7;
    unsigned kfa[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) kfa[s] = ldsbase + (unsigned)(r * 128 + (((2 * s + hh) ^ kswz) << 4));

    // item -> (query block, head, batch): the numbering of attention_fwd_mfma_kernel's workgroups (XCD x of a launch of nitems
    // workgroups would get the items x, x + 8, ...: the query blocks of one (batch, head) side by side on one XCD); item it is
    // taken by workgroup it % gridDim.x, and gridDim.x is a multiple of 8, so it still runs on XCD it % 8
    const int per_ = nitems >> 3, main_ = per_ << 3;
#define TRX_PST_COORDS(IT, QB, HH_, BB)                                                                     \
    {                                                                                                       \
        int bid_ = (IT);                                                                                    \
        if (bid_ < main_) bid_ = (bid_ & 7) * per_ + (bid_ >> 3);                                           \
        /* (the divisions run on the vector pipe: back to scalar registers, or every item's coordinates and row pointers \
           live in vector registers across the tile loop) */                                                \
        QB = __builtin_amdgcn_readfirstlane(bid_ % nqb); HH_ = __builtin_amdgcn_readfirstlane((bid_ / nqb) % H);  \
        BB = __builtin_amdgcn_readfirstlane(bid_ / (nqb * H));                                              \
    }
    int it = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
    if (it >= nitems) return;
    int qb, h, b;
    TRX_PST_COORDS(it, qb, h, b)
    const char* kbase = reinterpret_cast<const char*>(k + (int64_t)b * kvbs + h * 64);
    const char* vbase = reinterpret_cast<const char*>(v + (int64_t)b * kvbs + h * 64);
    // the first item's Q and mask (plain loads: hipcc waits for them where they are first named below), its first two tiles
    bf16x8 qf[4];
    float mv[2] = {0.f, 0.f};
    {
        const bf16_t* qp = q + ((int64_t)b * Lq + qb * 128 + wave * 32 + r) * ldq + h * 64 + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
        if (keymask) {
            const float* mk = mask + (int64_t)b * Lk;
            mv[0] = mk[min(2 * tid, Lk - 1)]; mv[1] = mk[min(2 * tid + 1, Lk - 1)];
        }
    }
    TRX_PST_STAGE(kbase, vbase, 0, 0)
    TRX_PST_STAGE(kbase, vbase, 1, 1)
    int buf = 0, par = 0;
    for (;;) {
        const int it_n = __builtin_amdgcn_readfirstlane(it + (int)gridDim.x);
        const bool has_next = it_n < nitems;
        int qb_n = 0, h_n = 0, b_n = 0;
        if (has_next) TRX_PST_COORDS(it_n, qb_n, h_n, b_n)
        const char* kbase_n = reinterpret_cast<const char*>(k + (int64_t)b_n * kvbs + h_n * 64);
        const char* vbase_n = reinterpret_cast<const char*>(v + (int64_t)b_n * kvbs + h_n * 64);
        const int qidx = qb * 128 + wave * 32 + r;
        // Q and the mask of this item have landed: named here so that hipcc puts its wait HERE and not in front of every tile's
        // first MFMA (for every item but the first, everything issued before them -- both staged tiles -- has landed as well)
        asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]), "+v"(mv[0]), "+v"(mv[1]));
        float* const mhalf = ldsM + par * 512;
        if (keymask) { mhalf[2 * tid] = fmaxf(mv[0] * inv_scale, mask_floor); mhalf[2 * tid + 1] = fmaxf(mv[1] * inv_scale, mask_floor); }
        f32x16 o0, o1;
#pragma unroll
        for (int t = 0; t < 16; ++t) { o0[t] = 0.f; o1[t] = 0.f; }
        float m = -__builtin_inff(), lsum = 0.f;
        const unsigned xdrop = DROP ? drop_base_da(da, (unsigned)(b * H + h)) + (unsigned)qidx * DROP_C1 + (unsigned)(2 * hh) * DROP_C2 : 0u;
        for (int kb = 0; kb < nkb; ++kb) {
            // tile kb of this item is in LDS for everyone once each wave's own pieces have landed: they are older than the four
            // pieces staged one iteration ago -- unless nothing was staged then (the last tile of the workgroup's last item)
            if (kb + 1 < nkb || has_next) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int buf1 = buf == 2 ? 0 : buf + 1, buf2 = buf == 0 ? 2 : buf - 1;
            if (kb + 2 < nkb) TRX_PST_STAGE(kbase, vbase, kb + 2, buf2)
            else if (has_next) TRX_PST_STAGE(kbase_n, vbase_n, kb + 2 - nkb, buf2)
            // ---- S^T = K Q^T for both 32-key halves ----
            f32x16 s0, s1;
            const int key0 = kb * 64;
            if (keymask) {
                const float* mt = mhalf + kb * 64 + 4 * hh;
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const float4 a = *reinterpret_cast<const float4*>(mt + 8 * t4);
                    const float4 c = *reinterpret_cast<const float4*>(mt + 32 + 8 * t4);
                    s0[4 * t4] = a.x; s0[4 * t4 + 1] = a.y; s0[4 * t4 + 2] = a.z; s0[4 * t4 + 3] = a.w;
                    s1[4 * t4] = c.x; s1[4 * t4 + 1] = c.y; s1[4 * t4 + 2] = c.z; s1[4 * t4 + 3] = c.w;
                }
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) { s0[t] = 0.f; s1[t] = 0.f; }
            }
            bf16x8 ka[4][2];
            {
                const unsigned kb0 = (unsigned)(buf * 16384);
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096"
                                 : "=&v"(ka[s][0]), "=&v"(ka[s][1]) : "v"(kfa[s] + kb0) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(ka[0][0]), "+v"(ka[0][1]), "+v"(ka[1][0]), "+v"(ka[1][1]),
                               "+v"(ka[2][0]), "+v"(ka[2][1]), "+v"(ka[3][0]), "+v"(ka[3][1]) :: "memory");
            }
            if (TRX_ATT_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s][0], qf[s], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s][1], qf[s], s1, 0, 0, 0);
            }
            if (TRX_ATT_PRIO) __builtin_amdgcn_s_setprio(0);
            const unsigned vtrA = vtrA0 + (unsigned)(buf * 16384), vtrB = vtrA ^ 64u;
            uint2 vt[4][2][2];
#define TRX_VT_READ(S)                                                                                        \
    asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%6\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\t"            \
                 "ds_read_b64_tr_b16 %2, %5 offset:%6\n\tds_read_b64_tr_b16 %3, %5 offset:%7"                 \
                 : "=&v"(vt[S][0][0]), "=&v"(vt[S][0][1]), "=&v"(vt[S][1][0]), "=&v"(vt[S][1][1])             \
                 : "v"(vtrA), "v"(vtrB), "n"((S) * 2048), "n"((S) * 2048 + 1024) : "memory");
#define TRX_VT_WAIT(S0, S1, CNT)                                                                              \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")"                                                                \
                 : "+v"(vt[S0][0][0]), "+v"(vt[S0][0][1]), "+v"(vt[S0][1][0]), "+v"(vt[S0][1][1]),            \
                   "+v"(vt[S1][0][0]), "+v"(vt[S1][0][1]), "+v"(vt[S1][1][0]), "+v"(vt[S1][1][1]) :: "memory");
#define TRX_PV_STEP(S, HB)                                                                                    \
    {                                                                                                         \
        constexpr int ss = (S) & 1;                                                                           \
        const bf16x8 pf = __builtin_bit_cast(bf16x8, uint4{pk[HB][4 * ss], pk[HB][4 * ss + 1], pk[HB][4 * ss + 2], pk[HB][4 * ss + 3]}); \
        uint4 v0; v0.x = vt[S][0][0].x; v0.y = vt[S][0][0].y; v0.z = vt[S][0][1].x; v0.w = vt[S][0][1].y;     \
        uint4 v1; v1.x = vt[S][1][0].x; v1.y = vt[S][1][0].y; v1.z = vt[S][1][1].x; v1.w = vt[S][1][1].y;     \
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0), pf, o0, 0, 0, 0);        \
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1), pf, o1, 0, 0, 0);        \
    }
            TRX_VT_READ(0) TRX_VT_READ(1)
            const unsigned xd = xdrop + (unsigned)(kb * 32) * DROP_C2;
            unsigned pk[2][8];
            // (Lk is a multiple of 64: no tile hides a key.  With the hidden-key form of the tile beside this one the kernel needs
            // 195 registers instead of 138 -- two waves per SIMD -- which is why that case stays with attention_fwd_mfma_kernel.)
            attn_softmax_tile<false, DROP>(s0, s1, o0, o1, m, lsum, sl2, key0, hh, klim, xd, da.thr, pk, other_half);
            TRX_VT_READ(2) TRX_VT_READ(3)
            TRX_VT_WAIT(0, 1, 8)
            if (TRX_ATT_PRIO) __builtin_amdgcn_s_setprio(1);
            TRX_PV_STEP(0, 0) TRX_PV_STEP(1, 0)
            TRX_VT_WAIT(2, 3, 0)
            TRX_PV_STEP(2, 1) TRX_PV_STEP(3, 1)
            if (TRX_ATT_PRIO) __builtin_amdgcn_s_setprio(0);
#undef TRX_VT_READ
#undef TRX_VT_WAIT
#undef TRX_PV_STEP
            buf = buf1;
        }
        // ---- the item's epilogue; the next item's Q and mask are requested first and arrive under the stores ----
        if (has_next) {
            const bf16_t* qp = q + ((int64_t)b_n * Lq + qb_n * 128 + wave * 32 + r) * ldq + h_n * 64 + 8 * hh;
#pragma unroll
            for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
            if (keymask) {
                const float* mk = mask + (int64_t)b_n * Lk;
                mv[0] = mk[min(2 * tid, Lk - 1)]; mv[1] = mk[min(2 * tid + 1, Lk - 1)];
            }
        }
        const float ltot = lsum + __shfl_xor(lsum, 32, 64);
        if (lse && hh == 0) lse[((int64_t)b * H + h) * Lq + qidx] = (m + __builtin_amdgcn_logf(ltot)) * 0.69314718055994530942f;
        store_row_bf16(out + ((int64_t)b * Lq + qidx) * H * 64 + (int64_t)h * 64, true, TRX_ATT_WIDE_STORE != 0, hh, o0, o1,
                       (DROP ? da.inv_keep : 1.0f) / ltot);
        if (!has_next) break;
        it = it_n; qb = qb_n; h = h_n; b = b_n; kbase = kbase_n; vbase = vbase_n; par ^= 1;
    }
#undef TRX_PST_STAGE
#undef TRX_PST_COORDS
}
