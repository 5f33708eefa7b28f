// Attention backward on the matrix cores (bf16 in, fp32 accumulate).  Included by nn_ops.hip inside
// its anonymous namespace, after the forward kernel whose fragment conventions it shares:
//
//   v_mfma_f32_32x32x16_bf16, lane = (r = lane & 31, hh = lane >> 5)
//     A operand: A[row r][k = 8 hh .. 8 hh + 7]        B operand: B[k = 8 hh .. 8 hh + 7][col r]
//     D: register t holds D[row (t & 3) + 8 (t >> 2) + 4 hh][col r]
//   An accumulator tile can be fed back as a B operand (k = its rows): registers 8 ss .. 8 ss + 7 of
//   lane (r, hh) are k-slots 8 hh .. 8 hh + 7 of the 16-row k-step ss, in the row order
//   (j & 3) + 8 (j >> 2) + 4 hh; the matching A operand comes from two ds_read_b64_tr_b16 (rows
//   4 hh + qq and 8 rows further), which deliver exactly that row order.
//
// Everything is deterministic (no atomics): probabilities are recomputed from the forward's
// log-sum-exp, once per pass.
//   pass 1  workgroup = 128 queries (a lane per query), streams 64-key tiles of K and V; first makes the per-query scalars
//           negl[b,h,i] = -lse_i / scale, negd[b,h,i] = -(dO_i . O_i) and leaves them for pass 2 (a launch of their own until round 4):
//             S^T = K Q^T (+ mask/scale),  dP^T = V dO^T - delta,  P = exp(scale S - lse),
//             dS = P (dP - delta),  dQ^T += K^T dS^T                       -> dQ = scale * acc
//   pass 2  workgroup = 128 keys (a lane per key), streams 64-query tiles of Q and dO:
//             S = Q K^T - lse/scale,  dP = dO V^T - delta,  P, dS as above,
//             dV^T += dO^T P,  dK^T += Q^T dS                              -> dK = scale * acc
// LDS tiles are rows of 128 B (64 bf16); 16-byte chunk c of row `row` sits at slot c ^ bwd_sw(row).
// bwd_sw is a bijection of (row >> 1) & 7, which keeps the ds_read_b128 row reads conflict-free
// (16-lane groups see 16 distinct (row parity, slot) pairs), and it moves rows 4a and 4a + 2 four
// slots apart, which is what the transposed reads need (a 32-lane half reads 4 rows x 64 B).

typedef __attribute__((ext_vector_type(4))) float f32x4v;

// lane id, recomputed where an epilogue needs it (volatile: hipcc otherwise keeps the lane-derived output addresses alive in
// registers the loop does not have, i.e. spills them before the loop and reloads them after it)
__device__ __forceinline__ int lane_again() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

__device__ __forceinline__ int bwd_sw(int row) {
    const int x = (row >> 1) & 7;
    return ((x & 1) << 2) | (x >> 1);
}

template <int N> struct ic_ { static constexpr int value = N; };     // a compile-time switch handed to a generic lambda
// 8 rows x 128 B of a [rows][H][64] bf16 tensor into LDS, one global_load_lds_dwordx4 per wave:
// lane -> (row base + lane / 8, slot lane % 8), source chunk slot ^ bwd_sw(row)
#define TRX_BWD_SW_OFS(ROWINTILE, PSLOT) ((unsigned)(((PSLOT) ^ bwd_sw(ROWINTILE)) * 16))

// row-fragment reads (A operand rows r and r + 32 of a tile), k-step s = d 16 s .. 16 s + 15
#define TRX_BWD_ROWS8(DST, BASE)                                                                            \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                                        \
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096"                               \
                     : "=&v"(DST[s_][0]), "=&v"(DST[s_][1]) : "v"(rfa[s_] + (BASE)) : "memory");
#define TRX_BWD_WAIT8(DST)                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                     \
                 : "+v"(DST[0][0]), "+v"(DST[0][1]), "+v"(DST[1][0]), "+v"(DST[1][1]),                      \
                   "+v"(DST[2][0]), "+v"(DST[2][1]), "+v"(DST[3][0]), "+v"(DST[3][1]) :: "memory");
// transposed reads of k-step S (16 tile rows) for both d blocks: DST[db][0 / 1] = low / high 4 rows
#define TRX_BWD_TR(DST, S, A0, A1)                                                                          \
    asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%8\n\tds_read_b64_tr_b16 %1, %5 offset:%9\n\t"           \
                 "ds_read_b64_tr_b16 %2, %6 offset:%8\n\tds_read_b64_tr_b16 %3, %7 offset:%9"               \
                 : "=&v"(DST[0][0]), "=&v"(DST[0][1]), "=&v"(DST[1][0]), "=&v"(DST[1][1])                   \
                 : "v"(A0), "v"(A1), "v"((A0) ^ 64u), "v"((A1) ^ 64u), "n"((S) * 2048), "n"((S) * 2048 + 1024) : "memory");
#define TRX_BWD_TRWAIT(DST, CNT)                                                                            \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")"                                                              \
                 : "+v"(DST[0][0]), "+v"(DST[0][1]), "+v"(DST[1][0]), "+v"(DST[1][1]) :: "memory");
#define TRX_BWD_PACK8(SV, SS)                                                                               \
    __builtin_bit_cast(bf16x8, uint4{pack2bf(SV[8 * (SS) + 0], SV[8 * (SS) + 1]), pack2bf(SV[8 * (SS) + 2], SV[8 * (SS) + 3]), \
                                     pack2bf(SV[8 * (SS) + 4], SV[8 * (SS) + 5]), pack2bf(SV[8 * (SS) + 6], SV[8 * (SS) + 7])})
#define TRX_BWD_TRFRAG(DST, DB)                                                                             \
    __builtin_bit_cast(bf16x8, uint4{DST[DB][0].x, DST[DB][0].y, DST[DB][1].x, DST[DB][1].y})

// ---- pass 1: dQ ------------------------------------------------------------------------------------
template <int MM, bool DROP>
#ifndef TRX_DQ_HALF      // 1: the dq pass works on half tiles (32 keys) and fits three waves per SIMD
#define TRX_DQ_HALF 1
#endif
#ifndef TRX_DQ_WAVES
#define TRX_DQ_WAVES (TRX_DQ_HALF ? 3 : 2)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MM == TRX_NN_MASK_FULL ? 2 : TRX_DQ_WAVES, MM == TRX_NN_MASK_FULL ? 2 : TRX_DQ_WAVES))) void attention_bwd_dq_mfma_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, const float* __restrict__ mask,
    int causal, int B, int H, int Lq, int Lk, float scale, const bf16_t* __restrict__ dout,
    const bf16_t* __restrict__ o, const float* __restrict__ lse, float* __restrict__ negl, float* __restrict__ negd,
    bf16_t* __restrict__ dq, DropArgs da) {
    __shared__ __attribute__((aligned(128))) char lds[3 * 16384];   // ring of 3: [K 8 KiB | V 8 KiB]
    __shared__ __attribute__((aligned(16))) float ldsM[1024];       // key mask / scale of 16 tiles
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void gbl_void;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int nqb = (Lq + 127) / 128;
    int bid = blockIdx.x;
    int qsw = 0;
    {   // query blocks of one (batch, head) next to each other on one XCD (they share K and V)
        const int nwg = gridDim.x, per = nwg >> 3, main_ = per << 3;
        if (nqb == 2 && !(per & 1) && bid < main_) qsw = (bid >> 3) >> 5;      // full and short blocks mixed per CU (see the forward kernel)
        if (bid < main_) bid = (bid & 7) * per + (bid >> 3);
    }
    const int qb = (bid + qsw) % nqb, h = (bid / nqb) % H, b = bid / (nqb * H);
    // a block with one or two 32-query units splits its KEYS over the waves that would otherwise compute on padding, as the
    // forward kernel does (nn_ops.hip: attention_fwd_mfma_kernel); dQ is linear in the keys, so the merge is a sum through LDS
    const int nuq = min(4, (Lq - qb * 128 + 31) / 32);
    int KS = nuq == 1 ? 4 : (nuq == 2 ? 2 : 1);               // block-uniform
    {   // (only with at least two key tiles per wave to deal out, as in the forward kernel)
        int nkb_ = (Lk + 63) / 64;
        if (causal) nkb_ = min(nkb_, (min(Lq - 1, qb * 128 + 127) + Lk - Lq) / 64 + 1);
        if (nkb_ < 2 * KS) KS = 1;
    }
    const int uq = KS == 4 ? 0 : (KS == 2 ? (wave & 1) : wave);
    const int kp = KS == 4 ? wave : (KS == 2 ? (wave >> 1) : 0);
    // the same two as scalars, for the epilogue only (they sit in scalar registers across the loop; the vector copies above
    // die with it instead of being spilled around it)
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int b_s = __builtin_amdgcn_readfirstlane(b), h_s = __builtin_amdgcn_readfirstlane(h), qb_s = __builtin_amdgcn_readfirstlane(qb);   // (the division that made them ran on the vector pipe)
    const int uq_s = KS == 4 ? 0 : (KS == 2 ? (wave_s & 1) : wave_s), kp_s = KS == 4 ? wave_s : (KS == 2 ? (wave_s >> 1) : 0);
    const int qidx = qb * 128 + uq * 32 + r;
    const int qc = qidx < Lq ? qidx : Lq - 1;
    const int ldq = da.ldq ? da.ldq : H * 64, ldk = da.ldk ? da.ldk : H * 64;
    bf16x8 qf[4], dof[4];
    {
        const int64_t roq = ((int64_t)b * Lq + qc) * ldq + h * 64 + 8 * hh;      // q may be a slice of a packed projection
        const int64_t rod = (((int64_t)b * Lq + qc) * H + h) * 64 + 8 * hh;      // dout is dense
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = *reinterpret_cast<const bf16x8*>(q + roq + 16 * s);
            dof[s] = *reinterpret_cast<const bf16x8*>(dout + rod + 16 * s);
        }
    }
    f32x16 a0, a1;   // dQ^T: d blocks 0..31 / 32..63 x this wave's 32 queries
#pragma unroll
    for (int t = 0; t < 16; ++t) { a0[t] = 0.f; a1[t] = 0.f; }
    const int off = Lk - Lq;
    int nkb = (Lk + 63) / 64;
    if (causal) nkb = min(nkb, (min(Lq - 1, qb * 128 + 127) + off) / 64 + 1);
    const int klim = causal ? min(Lk - 1, qidx + off) : Lk - 1;
    const int klim_wave_min = causal ? min(Lk - 1, qb * 128 + uq * 32 + off) : Lk - 1;
    const float* mrow = (MM == TRX_NN_MASK_FULL) ? mask + ((int64_t)b * Lq + qc) * Lk : nullptr;
    constexpr bool keymask = MM == TRX_NN_MASK_KEY;
    const float* mkey = keymask ? mask + (int64_t)b_s * Lk : nullptr;      // a scalar base (b itself sits in a vector register)
    // dropout hash input of (this lane's query, key pair 0) -- the same function the forward evaluated
    const unsigned xdrop = DROP ? drop_base_da(da, (unsigned)(b * H + h)) + (unsigned)qidx * DROP_C1 + (unsigned)(2 * hh) * DROP_C2 : 0u;

    const int prow = lane >> 3, pslot = lane & 7;
    const unsigned rowbytes = (unsigned)ldk * 2u;
    const char* kbase = reinterpret_cast<const char*>(k + (int64_t)b * Lk * ldk + h * 64);
    const char* vbase = reinterpret_cast<const char*>(v + (int64_t)b * Lk * ldk + h * 64);
    unsigned sofs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rw = 8 * (2 * wave + i) + prow;
        sofs[i] = (unsigned)rw * rowbytes + TRX_BWD_SW_OFS(rw, pslot);
    }
    const unsigned ldsbase0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned lds_w = (unsigned)__builtin_amdgcn_readfirstlane((int)(ldsbase0 + (unsigned)(2 * wave * 1024)));   // this wave's pieces, scalar
#define TRX_BWD1_STAGE(KB, BUF)                                                                             \
    {                                                                                                       \
        const char* kt_ = kbase + (int64_t)(KB) * 64 * rowbytes;                                            \
        const char* vt_ = vbase + (int64_t)(KB) * 64 * rowbytes;                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                  \
            unsigned so_ = sofs[i_];                                                                        \
            if ((KB) * 64 + 64 > Lk) { /* tail tile: rows past the last key re-read it (hidden anyway) */   \
                const int rw_ = 8 * (2 * wave + i_) + prow;                                                 \
                so_ = (unsigned)min(rw_, Lk - 1 - (KB) * 64) * rowbytes + TRX_BWD_SW_OFS(rw_, pslot);       \
            }                                                                                               \
            TRX_GLDS16((unsigned long long)kt_, so_, lds_w + (unsigned)((BUF) * 16384 + i_ * 1024));                      \
            TRX_GLDS16((unsigned long long)vt_, so_, lds_w + (unsigned)((BUF) * 16384 + 8192 + i_ * 1024));               \
        }                                                                                                   \
    }
    const unsigned ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    unsigned rfa[4];   // row-fragment addresses: row r (+32 by offset), chunk 2 s + hh
#pragma unroll
    for (int s = 0; s < 4; ++s) rfa[s] = ldsbase + (unsigned)(r * 128 + (((2 * s + hh) ^ bwd_sw(r)) << 4));
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    // transposed reads: rows 4 hh + qq (+ 8), chunk 2 (g & 1) + (pp >> 1) (+ 4 for d block 1 = ^ 64)
    const int tsw = ((qq >> 1) << 2) | hh;   // bwd_sw(4 hh + qq); 8 rows further it is tsw ^ 2
    const int tc = 2 * (g & 1) + (pp >> 1);
    const unsigned tra0 = ldsbase + (unsigned)((4 * hh + qq) * 128 + ((tc ^ tsw) << 4) + 8 * (pp & 1));
    const unsigned tra1 = ldsbase + (unsigned)((4 * hh + qq) * 128 + ((tc ^ tsw ^ 2) << 4) + 8 * (pp & 1));

    // Everything the prologue reads is requested before anything is waited for (round 5): Q and dO above, here the first 1,024
    // keys' mask and the LDS-DMA of tiles 0 and 1, then O and the log-sum-exp below.  The per-query scalars used to be finished
    // -- a wait for Q, dO, O -- before the first tile was staged, and the mask was loaded after that: three memory latencies
    // in a row in front of the first MFMA, now one.
    float mv0[4] = {0.f, 0.f, 0.f, 0.f};
    if (keymask) {
#pragma unroll
        for (int i_ = 0; i_ < 4; ++i_) mv0[i_] = mkey[min(4 * tid + i_, Lk - 1)];
    }
    TRX_BWD1_STAGE(0, 0);
    if (nkb > 1) TRX_BWD1_STAGE(1, 1);
    const float sl2 = scale * 1.44269504088896340736f;
    const float inv_scale = 1.0f / scale;
    const float mask_floor = -268435456.0f / sl2;     // the forward kernel's floor of a masked score (nn_ops.hip: TRX_MASK_INIT)
    // The per-query scalars are made HERE (rounds 1-3: a launch of their own, 12 us in front of every backward): a lane holds
    // half of its query's dO row as fragments already, the same half of the O row is four more loads, and the two halves
    // meet through one cross-half shuffle.  Written out for the dk/dv pass, which runs after this launch.
    float nd;
    {
        const int64_t rod = (((int64_t)b * Lq + qc) * H + h) * 64 + 8 * hh;
        float part = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint4 ow = *reinterpret_cast<const uint4*>(o + rod + 16 * s);
            const uint4 dw = __builtin_bit_cast(uint4, dof[s]);
            const unsigned aw[4] = {ow.x, ow.y, ow.z, ow.w}, bw[4] = {dw.x, dw.y, dw.z, dw.w};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                part = __builtin_fmaf(__uint_as_float(aw[w] << 16), __uint_as_float(bw[w] << 16), part);
                part = __builtin_fmaf(__uint_as_float(aw[w] & 0xffff0000u), __uint_as_float(bw[w] & 0xffff0000u), part);
            }
        }
        nd = -(part + __shfl_xor(part, 32, 64));                      // -delta = -(dO . O)
    }
    const int64_t wq_ = ((int64_t)b * H + h) * Lq + qc;
    const float nl = -lse[wq_] / scale;
    const float nlsl2 = nl * sl2;                                     // -lse * log2 e
    if (kp == 0 && hh == 0 && qidx < Lq) { negl[wq_] = nl; negd[wq_] = nd; }
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));
    asm volatile("" : "+v"(dof[0]), "+v"(dof[1]), "+v"(dof[2]), "+v"(dof[3]));
    if (keymask) {   // (before the loop: written inside it, the four registers stayed live -- spilled -- across all tiles)
#pragma unroll
        for (int i_ = 0; i_ < 4; ++i_) ldsM[4 * tid + i_] = fmaxf(mv0[i_] * inv_scale, mask_floor);
    }
    int buf = 0;
    for (int kc = 0; kc < nkb; kc += 16) {
    if (keymask && kc > 0) {                              // (rare: Lk > 1024) the next 1,024 keys' mask
        __syncthreads();
        float mv_[4];
        const int tid_ = lane_again() + 64 * wave_s;      // recomputed: the thread id would otherwise be kept (spilled) across the tile loop
#pragma unroll
        for (int i_ = 0; i_ < 4; ++i_) mv_[i_] = mkey[min(kc * 64 + 4 * tid_ + i_, Lk - 1)];
        float sc_ = scale;
        asm volatile("" : "+s"(sc_));        // (opaque: made again here, or hipcc keeps -- spills -- the prologue's 1 / scale across the tiles)
        const float isc_ = 1.0f / sc_, floor_ = -268435456.0f / (sc_ * 1.44269504088896340736f);
#pragma unroll
        for (int i_ = 0; i_ < 4; ++i_) ldsM[4 * tid_ + i_] = fmaxf(mv_[i_] * isc_, floor_);
    }
    const int kend = min(nkb, kc + 16);
    for (int kb = kc; kb < kend; ++kb) {
        if (kb + 1 < nkb) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int buf1 = buf == 2 ? 0 : buf + 1, buf2 = buf == 0 ? 2 : buf - 1;
        if (kb + 2 < nkb) TRX_BWD1_STAGE(kb + 2, buf2);
        if ((kb & (KS - 1)) == kp) {          // wave-uniform: this wave's tile
        const unsigned bofs = (unsigned)(buf * 16384);
        const int key0 = kb * 64;
#if TRX_DQ_HALF
        // Round 5: the tile as two halves of 32 keys, one after the other -- S, dP, the elementwise part and the dQ update of a
        // half use 16 + 16 accumulator registers instead of 32 + 32 and four row fragments instead of eight: 145 registers
        // instead of 186-198, THREE waves per SIMD instead of two.  Every accumulator sees the same products in the same order
        // as in the whole-tile form (bit-identical results).
        const unsigned ta0 = tra0 + bofs, ta1 = tra1 + bofs;
        const bool vis = key0 + 63 > klim_wave_min;
        auto half = [&](auto HB) __attribute__((always_inline)) {
            constexpr int hb = decltype(HB)::value;
            // ---- S^T = K Q^T (+ mask / scale) ----
            f32x16 sh;
            if (keymask) {
                const float* mt = ldsM + (kb - kc) * 64 + 32 * hb + 4 * hh;
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const float4 a = *reinterpret_cast<const float4*>(mt + 8 * t4);
                    sh[4 * t4] = a.x; sh[4 * t4 + 1] = a.y; sh[4 * t4 + 2] = a.z; sh[4 * t4 + 3] = a.w;
                }
            } else if (MM == TRX_NN_MASK_FULL) {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int kr_ = (t & 3) + 8 * (t >> 2) + 4 * hh;
                    sh[t] = fmaxf(mrow[min(key0 + 32 * hb + kr_, Lk - 1)] * inv_scale, mask_floor);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) sh[t] = 0.f;
            }
            bf16x8 fr[4];
#define TRX_BWD_ROWS4(DST, BASE)                                                                            \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                                        \
        asm volatile("ds_read_b128 %0, %1" : "=&v"(DST[s_]) : "v"(rfa[s_] + (BASE)) : "memory");
#define TRX_BWD_WAIT4(DST) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(DST[0]), "+v"(DST[1]), "+v"(DST[2]), "+v"(DST[3]) :: "memory");
            TRX_BWD_ROWS4(fr, bofs + (unsigned)(hb * 4096))
            TRX_BWD_WAIT4(fr)
#pragma unroll
            for (int s = 0; s < 4; ++s) sh = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s], qf[s], sh, 0, 0, 0);
            // ---- dP^T - delta = V dO^T - delta ----
            f32x16 ph;
#pragma unroll
            for (int t = 0; t < 16; ++t) ph[t] = DROP ? 0.f : nd;   // dropout rescales dP before - delta
            TRX_BWD_ROWS4(fr, bofs + 8192u + (unsigned)(hb * 4096))
            TRX_BWD_WAIT4(fr)
#undef TRX_BWD_ROWS4
#undef TRX_BWD_WAIT4
#pragma unroll
            for (int s = 0; s < 4; ++s) ph = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s], dof[s], ph, 0, 0, 0);
            // K^T fragments of this half's 32 keys fly under the elementwise part
            uint2 kt0[2][2], kt1[2][2];
            TRX_BWD_TR(kt0, 2 * hb, ta0, ta1)
            TRX_BWD_TR(kt1, 2 * hb + 1, ta0, ta1)
            // ---- dS = P (dP - delta), P = exp2(scale log2e S - lse log2e) ----
            if (DROP) {
                const unsigned xd = xdrop + (unsigned)(kb * 32) * DROP_C2;
#pragma unroll
                for (int t = 0; t < 16; t += 2) {
                    const unsigned bits = lowbias32(xd + (unsigned)(hb * 16 + ((t & 3) >> 1) + 4 * (t >> 2)) * DROP_C2);
                    const float k0 = drop_keep(bits, 0, da.thr) ? da.inv_keep : 0.f, k1 = drop_keep(bits, 1, da.thr) ? da.inv_keep : 0.f;
                    ph[t] = __builtin_fmaf(ph[t], k0, nd); ph[t + 1] = __builtin_fmaf(ph[t + 1], k1, nd);
                }
            }
            if (vis) {      // hidden keys: a score of minus infinity in place (its probability is exp2(-inf) = 0)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int kr_ = 32 * hb + (t & 3) + 8 * (t >> 2) + 4 * hh;
                    sh[t] = (key0 + kr_ > klim) ? -__builtin_inff() : sh[t];
                }
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float pr = __builtin_amdgcn_exp2f(fminf(__builtin_fmaf(sh[t], sl2, nlsl2), 0.f));   // p <= 1
                sh[t] = pr * ph[t];
            }
            // ---- dQ^T += K^T dS^T ----
            TRX_BWD_TRWAIT(kt0, 4)
            {
                const bf16x8 ds = TRX_BWD_PACK8(sh, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(kt0, 0), ds, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(kt0, 1), ds, a1, 0, 0, 0);
            }
            TRX_BWD_TRWAIT(kt1, 0)
            {
                const bf16x8 ds = TRX_BWD_PACK8(sh, 1);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(kt1, 0), ds, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(kt1, 1), ds, a1, 0, 0, 0);
            }
        };
        half(ic_<0>{});
        half(ic_<1>{});
#else
        // ---- S^T = K Q^T (+ mask / scale) ----
        f32x16 s0, s1;
        if (keymask) {
            const float* mt = ldsM + (kb - kc) * 64 + 4 * hh;
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const float4 a = *reinterpret_cast<const float4*>(mt + 8 * t4);
                const float4 c = *reinterpret_cast<const float4*>(mt + 32 + 8 * t4);
                s0[4 * t4] = a.x; s0[4 * t4 + 1] = a.y; s0[4 * t4 + 2] = a.z; s0[4 * t4 + 3] = a.w;
                s1[4 * t4] = c.x; s1[4 * t4 + 1] = c.y; s1[4 * t4 + 2] = c.z; s1[4 * t4 + 3] = c.w;
            }
        } else if (MM == TRX_NN_MASK_FULL) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int kr_ = (t & 3) + 8 * (t >> 2) + 4 * hh;
                s0[t] = fmaxf(mrow[min(key0 + kr_, Lk - 1)] * inv_scale, mask_floor);
                s1[t] = fmaxf(mrow[min(key0 + 32 + kr_, Lk - 1)] * inv_scale, mask_floor);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) { s0[t] = 0.f; s1[t] = 0.f; }
        }
        bf16x8 fr[4][2];
        TRX_BWD_ROWS8(fr, bofs)
        TRX_BWD_WAIT8(fr)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s][0], qf[s], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s][1], qf[s], s1, 0, 0, 0);
        }
        // ---- dP^T - delta = V dO^T - delta ----
        f32x16 p0, p1;
#pragma unroll
        for (int t = 0; t < 16; ++t) { p0[t] = DROP ? 0.f : nd; p1[t] = DROP ? 0.f : nd; }   // dropout rescales dP before - delta
        TRX_BWD_ROWS8(fr, bofs + 8192u)
        TRX_BWD_WAIT8(fr)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            p0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s][0], dof[s], p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s][1], dof[s], p1, 0, 0, 0);
        }
        // K^T fragments of the first 32 keys fly under the elementwise part
        uint2 kt0[2][2], kt1[2][2];
        const unsigned ta0 = tra0 + bofs, ta1 = tra1 + bofs;
        TRX_BWD_TR(kt0, 0, ta0, ta1)
        TRX_BWD_TR(kt1, 1, ta0, ta1)
        // ---- dS = P (dP - delta), P = exp2(scale log2e S - lse log2e) ----
        const bool vis = key0 + 63 > klim_wave_min;
        if (DROP) {
            const unsigned xd = xdrop + (unsigned)(kb * 32) * DROP_C2;
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int t = 0; t < 16; t += 2) {
                    const unsigned bits = lowbias32(xd + (unsigned)(hb * 16 + ((t & 3) >> 1) + 4 * (t >> 2)) * DROP_C2);
                    const float k0 = drop_keep(bits, 0, da.thr) ? da.inv_keep : 0.f, k1 = drop_keep(bits, 1, da.thr) ? da.inv_keep : 0.f;
                    if (hb) { p1[t] = __builtin_fmaf(p1[t], k0, nd); p1[t + 1] = __builtin_fmaf(p1[t + 1], k1, nd); }
                    else { p0[t] = __builtin_fmaf(p0[t], k0, nd); p0[t + 1] = __builtin_fmaf(p0[t + 1], k1, nd); }
                }
        }
        // (rounds 3-4: two copies of the elementwise block behind one wave-uniform branch -- written as `if (vis) pr = ...` inside the
        // loop hipcc if-converts the test into a v_cmp + v_cndmask per element on EVERY tile; round 5: the hidden scores are set to
        // minus infinity in place behind that branch and ONE copy follows: 40-50 registers fewer, 1-2 % faster,
        // profiles/r05_attention_bwd_inplace_ab.txt)
        if (vis) {      // hidden keys: a score of minus infinity, in place (its probability is exp2(-inf) = 0), then ONE form for all tiles
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int kr_ = (t & 3) + 8 * (t >> 2) + 4 * hh;
                s0[t] = (key0 + kr_ > klim) ? -__builtin_inff() : s0[t];
                s1[t] = (key0 + 32 + kr_ > klim) ? -__builtin_inff() : s1[t];
            }
        }
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float pr = __builtin_amdgcn_exp2f(fminf(__builtin_fmaf(hb ? s1[t] : s0[t], sl2, nlsl2), 0.f));   // p <= 1
                if (hb) s1[t] = pr * p1[t]; else s0[t] = pr * p0[t];
            }
        // ---- dQ^T += K^T dS^T ----
        TRX_BWD_TRWAIT(kt0, 4)
        {
            const bf16x8 ds = TRX_BWD_PACK8(s0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(kt0, 0), ds, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(kt0, 1), ds, a1, 0, 0, 0);
        }
        TRX_BWD_TR(kt0, 2, ta0, ta1)
        TRX_BWD_TRWAIT(kt1, 4)
        {
            const bf16x8 ds = TRX_BWD_PACK8(s0, 1);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(kt1, 0), ds, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(kt1, 1), ds, a1, 0, 0, 0);
        }
        TRX_BWD_TR(kt1, 3, ta0, ta1)
        TRX_BWD_TRWAIT(kt0, 4)
        {
            const bf16x8 ds = TRX_BWD_PACK8(s1, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(kt0, 0), ds, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(kt0, 1), ds, a1, 0, 0, 0);
        }
        TRX_BWD_TRWAIT(kt1, 0)
        {
            const bf16x8 ds = TRX_BWD_PACK8(s1, 1);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(kt1, 0), ds, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(kt1, 1), ds, a1, 0, 0, 0);
        }
#endif
        }
        buf = buf1;
    }
    }
#undef TRX_BWD1_STAGE
    // the epilogue's index arithmetic starts again from scalars the compiler cannot connect with the prologue's (an empty asm
    // makes them opaque): every vector value of the prologue then dies with the loop instead of being spilled around it
    int b_e = b_s, h_e = h_s, qb_e = qb_s;
    asm volatile("" : "+s"(b_e), "+s"(h_e), "+s"(qb_e));
    const int lane_e = lane_again(), hh_e = lane_e >> 5, qidx_e = qb_e * 128 + uq_s * 32 + (lane_e & 31);
    if (KS > 1) {     // block-uniform: sum the key-split waves of every query unit, in wave order (deterministic)
        const int lane = lane_e, uq = uq_s, kp = kp_s;
        __syncthreads();
        float4* part = reinterpret_cast<float4*>(lds);       // [slot][8][64 lanes] float4 in the K / V ring
        const int per = 4 / KS;
        if (kp > 0) {
            float4* w = part + ((kp - 1) * per + uq) * 8 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w[i * 64] = make_float4(a0[4 * i], a0[4 * i + 1], a0[4 * i + 2], a0[4 * i + 3]);
                w[(4 + i) * 64] = make_float4(a1[4 * i], a1[4 * i + 1], a1[4 * i + 2], a1[4 * i + 3]);
            }
        }
        __syncthreads();
        if (kp == 0) {
            for (int p_ = 1; p_ < KS; ++p_) {
                const float4* rd = part + ((p_ - 1) * per + uq) * 8 * 64 + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 x = rd[i * 64], y = rd[(4 + i) * 64];
                    a0[4 * i] += x.x; a0[4 * i + 1] += x.y; a0[4 * i + 2] += x.z; a0[4 * i + 3] += x.w;
                    a1[4 * i] += y.x; a1[4 * i + 1] += y.y; a1[4 * i + 2] += y.z; a1[4 * i + 3] += y.w;
                }
            }
        }
    }
    {
        const bool live = qidx_e < Lq && kp_s == 0;
        store_row_bf16(dq + ((int64_t)b_e * Lq + (live ? qidx_e : 0)) * (da.ldq ? da.ldq : H * 64) + h_e * 64, live, TRX_ATT_WIDE_STORE && KS == 1,
                       hh_e, a0, a1, scale);
    }
}

// ---- pass 2: dK, dV --------------------------------------------------------------------------------
constexpr int BWD2_STAGE = 16384 + 512;   // Q 8 KiB | dO 8 KiB | -lse/scale of 64 queries | -delta of 64 queries

template <int MM, bool DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention_bwd_dkv_mfma_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, const float* __restrict__ mask,
    int causal, int B, int H, int Lq, int Lk, float scale, const bf16_t* __restrict__ dout,
    const float* __restrict__ negl, const float* __restrict__ negd, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, DropArgs da) {
    __shared__ __attribute__((aligned(128))) char lds[3 * BWD2_STAGE];
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void gbl_void;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int nkblk = (Lk + 127) / 128;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, per = nwg >> 3, main_ = per << 3;
        if (bid < main_) bid = (bid & 7) * per + (bid >> 3);
    }
    const int kblk = bid % nkblk, h = (bid / nkblk) % H, b = bid / (nkblk * H);
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);     // for the epilogue (see the dq pass)
    const int b_s = __builtin_amdgcn_readfirstlane(b), h_s = __builtin_amdgcn_readfirstlane(h), kblk_s = __builtin_amdgcn_readfirstlane(kblk);
    const int kidx = kblk * 128 + wave * 32 + r;   // this lane's key
    const int kc = kidx < Lk ? kidx : Lk - 1;
    const int ldq = da.ldq ? da.ldq : H * 64, ldk = da.ldk ? da.ldk : H * 64;
    bf16x8 kf[4], vf[4];
    {
        const int64_t ro = ((int64_t)b * Lk + kc) * ldk + h * 64 + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = *reinterpret_cast<const bf16x8*>(k + ro + 16 * s);
            vf[s] = *reinterpret_cast<const bf16x8*>(v + ro + 16 * s);
        }
    }
    constexpr float L2E = 1.44269504088896340736f;
    const float sl2 = scale * L2E;
    const float mk2 = (MM == TRX_NN_MASK_KEY) ? fmaxf(mask[(int64_t)b * Lk + kc] * L2E, -268435456.0f) : 0.f;     // the forward's floor, in the exponent's units
    f32x16 ak0, ak1, av0, av1;   // dK^T, dV^T: d blocks x this wave's 32 keys
#pragma unroll
    for (int t = 0; t < 16; ++t) { ak0[t] = 0.f; ak1[t] = 0.f; av0[t] = 0.f; av1[t] = 0.f; }
    const int off = Lk - Lq;
    const int nqt = (Lq + 63) / 64;
    // causal: query i sees key j iff j <= i + off; the first query that sees any key of this workgroup
    const int qt0 = causal ? max(0, kblk * 128 - off) / 64 : 0;
    const int qmin = causal ? kidx - off : 0;                       // this lane's key is visible to queries >= qmin
    const int qmin_wave_max = causal ? kblk * 128 + wave * 32 + 31 - off : 0;
    // dropout hash input of (query 4 hh, this lane's key pair); the key's half of the hash is bit 4 of dshift
    const unsigned xdrop = DROP ? drop_base_da(da, (unsigned)(b * H + h)) + (unsigned)(4 * hh) * DROP_C1 + ((unsigned)kc >> 1) * DROP_C2 : 0u;
    const unsigned dshift = ((unsigned)kc & 1u) << 4;
    const bool lane_odd = (lane & 1) != 0;
    const unsigned xdrop_p = xdrop + (lane_odd ? DROP_C1 : 0u);      // the hash input of this lane's own parity of queries (see the elementwise part)

    const int prow = lane >> 3, pslot = lane & 7;
    const unsigned rowbytes = (unsigned)ldq * 2u, rowbytes_d = (unsigned)H * 128u;   // q rows (maybe packed) / dout rows (dense)
    const char* qbase = reinterpret_cast<const char*>(q + (int64_t)b * Lq * ldq + h * 64);
    const char* dobase = reinterpret_cast<const char*>(dout + ((int64_t)b * Lq * H + h) * 64);
    const float* nlbase = negl + ((int64_t)b * H + h) * Lq;
    const float* ndbase = negd + ((int64_t)b * H + h) * Lq;
    const float* mfull = (MM == TRX_NN_MASK_FULL) ? mask + (int64_t)b * Lq * Lk : nullptr;     // this batch's [Lq][Lk] mask (Lq * Lk < 2^30)
    unsigned sofs[2], sofd[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rw = 8 * (2 * wave + i) + prow;
        sofs[i] = (unsigned)rw * rowbytes + TRX_BWD_SW_OFS(rw, pslot);
        sofd[i] = (unsigned)rw * rowbytes_d + TRX_BWD_SW_OFS(rw, pslot);
    }
    // per stage and wave: 4 tile loads (+ 2 loads of the per-query scalars on wave 0)
    const unsigned ldsbase0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned lds_w = (unsigned)__builtin_amdgcn_readfirstlane((int)(ldsbase0 + (unsigned)(2 * wave * 1024)));
#define TRX_BWD2_STAGE(QT, BUF)                                                                             \
    {                                                                                                       \
        const char* qt_ = qbase + (int64_t)(QT) * 64 * rowbytes;                                            \
        const char* dt_ = dobase + (int64_t)(QT) * 64 * rowbytes_d;                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                  \
            unsigned so_ = sofs[i_], sd_ = sofd[i_];                                                        \
            if ((QT) * 64 + 64 > Lq) { /* tail tile: rows past the last query re-read it (masked below) */  \
                const int rw_ = 8 * (2 * wave + i_) + prow;                                                 \
                so_ = (unsigned)min(rw_, Lq - 1 - (QT) * 64) * rowbytes + TRX_BWD_SW_OFS(rw_, pslot);       \
                sd_ = (unsigned)min(rw_, Lq - 1 - (QT) * 64) * rowbytes_d + TRX_BWD_SW_OFS(rw_, pslot);     \
            }                                                                                               \
            TRX_GLDS16((unsigned long long)qt_, so_, lds_w + (unsigned)((BUF) * BWD2_STAGE + i_ * 1024));                 \
            TRX_GLDS16((unsigned long long)dt_, sd_, lds_w + (unsigned)((BUF) * BWD2_STAGE + 8192 + i_ * 1024));          \
        }                                                                                                   \
        if (wave == 0) {                                                                                    \
            const int qi_ = min((QT) * 64 + lane, Lq - 1);                                                  \
            TRX_GLDS4((unsigned long long)nlbase, (unsigned)qi_ * 4u, ldsbase0 + (unsigned)((BUF) * BWD2_STAGE + 16384));       \
            TRX_GLDS4((unsigned long long)ndbase, (unsigned)qi_ * 4u, ldsbase0 + (unsigned)((BUF) * BWD2_STAGE + 16384 + 256)); \
        }                                                                                                   \
    }
    const unsigned ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    unsigned rfa[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) rfa[s] = ldsbase + (unsigned)(r * 128 + (((2 * s + hh) ^ bwd_sw(r)) << 4));
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const int tsw = ((qq >> 1) << 2) | hh;
    const int tc = 2 * (g & 1) + (pp >> 1);
    const unsigned tra0 = ldsbase + (unsigned)((4 * hh + qq) * 128 + ((tc ^ tsw) << 4) + 8 * (pp & 1));
    const unsigned tra1 = ldsbase + (unsigned)((4 * hh + qq) * 128 + ((tc ^ tsw ^ 2) << 4) + 8 * (pp & 1));
    const unsigned sca = ldsbase + 16384u + (unsigned)(16 * hh);   // per-query scalars: rows 4 hh .. (+8 per float4)

    if (qt0 < nqt) TRX_BWD2_STAGE(qt0, 0);
    if (qt0 + 1 < nqt) TRX_BWD2_STAGE(qt0 + 1, 1);
    asm volatile("" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]));
    asm volatile("" : "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]));
    int buf = 0;
    for (int qt = qt0; qt < nqt; ++qt) {
        if (qt + 1 < nqt) {
            if (wave == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int buf1 = buf == 2 ? 0 : buf + 1, buf2 = buf == 0 ? 2 : buf - 1;
        if (qt + 2 < nqt) TRX_BWD2_STAGE(qt + 2, buf2);
        const unsigned bofs = (unsigned)(buf * BWD2_STAGE);
        const int q0 = qt * 64;
        // some query row of the tile is past Lq, or hidden from some key of this wave by causality
        const bool vis = (q0 + 64 > Lq) || (causal && q0 < qmin_wave_max);
        const unsigned ta0 = tra0 + bofs, ta1 = tra1 + bofs;
        auto half_tile = [&](auto HB) __attribute__((always_inline)) {   // 32 queries at a time; hb is a compile-time constant (asm immediates)
            constexpr int hb = decltype(HB)::value;
            // accumulators start from the per-query scalars: rows (t & 3) + 8 (t >> 2) + 4 hh
            f32x16 s, p;
            {
                f32x4v nl4[4], nd4[4];
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4)
                    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                                 : "=&v"(nl4[t4]), "=&v"(nd4[t4]) : "v"(sca + bofs), "n"(hb * 128 + 32 * t4), "n"(256 + hb * 128 + 32 * t4) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(nl4[0]), "+v"(nl4[1]), "+v"(nl4[2]), "+v"(nl4[3]), "+v"(nd4[0]), "+v"(nd4[1]), "+v"(nd4[2]), "+v"(nd4[3]) :: "memory");
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    s[4 * t4] = nl4[t4].x; s[4 * t4 + 1] = nl4[t4].y; s[4 * t4 + 2] = nl4[t4].z; s[4 * t4 + 3] = nl4[t4].w;
                    p[4 * t4] = nd4[t4].x; p[4 * t4 + 1] = nd4[t4].y; p[4 * t4 + 2] = nd4[t4].z; p[4 * t4 + 3] = nd4[t4].w;
                }
            }
            f32x16 ndv;   // dropout rescales dP before - delta: keep - delta aside, start dP from 0
            if (DROP) {
#pragma unroll
                for (int t = 0; t < 16; ++t) { ndv[t] = p[t]; p[t] = 0.f; }
            }
            // ---- S - lse/scale = Q K^T ... ;  dP - delta = dO V^T ... : rows 32 hb + r of the tiles ----
            bf16x8 fq[4], fd[4];
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
                asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                             : "=&v"(fq[s_]), "=&v"(fd[s_]) : "v"(rfa[s_] + bofs), "n"(hb * 4096), "n"(8192 + hb * 4096) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(fq[0]), "+v"(fq[1]), "+v"(fq[2]), "+v"(fq[3]), "+v"(fd[0]), "+v"(fd[1]), "+v"(fd[2]), "+v"(fd[3]) :: "memory");
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fq[s_], kf[s_], s, 0, 0, 0);
                p = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd[s_], vf[s_], p, 0, 0, 0);
            }
            // transposed fragments of this half: k-steps 2 hb, 2 hb + 1 of dO (for dV) and of Q (for dK).  They fly under the
            // elementwise part -- except with a per-element mask, whose 16 values per half take the registers the fragments
            // would hold meanwhile (that variant spilled into scratch, and a spill reload waits behind the LDS-DMA queue)
            uint2 td0[2][2], td1[2][2], tq0[2][2], tq1[2][2];
#define TRX_BWD2_TRANSPOSED()                                                                               \
            if (hb == 0) {                                                                                  \
                TRX_BWD_TR(td0, 0, ta0 + 8192u, ta1 + 8192u) TRX_BWD_TR(td1, 1, ta0 + 8192u, ta1 + 8192u)   \
                TRX_BWD_TR(tq0, 0, ta0, ta1) TRX_BWD_TR(tq1, 1, ta0, ta1)                                   \
            } else {                                                                                        \
                TRX_BWD_TR(td0, 2, ta0 + 8192u, ta1 + 8192u) TRX_BWD_TR(td1, 3, ta0 + 8192u, ta1 + 8192u)   \
                TRX_BWD_TR(tq0, 2, ta0, ta1) TRX_BWD_TR(tq1, 3, ta0, ta1)                                   \
            }
            if (MM != TRX_NN_MASK_FULL) { TRX_BWD2_TRANSPOSED() }
            // ---- P = exp2(scale log2e (S - lse/scale) + mask log2e) ; dS = P (dP - delta) ----  (two copies behind one
            // wave-uniform branch, as in the dq pass)
            if (vis) {      // hidden queries: a score of minus infinity in place (probability exp2(-inf) = 0), then ONE form for all tiles
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int qr_ = q0 + hb * 32 + (t & 3) + 8 * (t >> 2) + 4 * hh;
                    s[t] = (qr_ >= Lq || qr_ < qmin) ? -__builtin_inff() : s[t];
                }
            }
            {
                // Dropout decisions.  A hash serves a PAIR of keys (its two 16-bit halves), and here a lane is a key: lanes 2 j and
                // 2 j + 1 would compute the same 16 hashes each.  Round 5: each computes the eight of its own parity (queries
                // t = 2 i + (lane & 1)) and takes the other eight from its neighbour with one quad-permute move each -- 8 hashes + 8
                // moves + 16 selects instead of 16 hashes (a hash is two quarter-rate multiplies and six more instructions).
                unsigned hbits[16];
                if (DROP) {
                    unsigned own[8], nbr[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int ci = ((2 * i) & 3) + 8 * (i >> 1);       // (t & 3) + 8 (t >> 2) of t = 2 i, without the lane's parity
                        own[i] = lowbias32(xdrop_p + (unsigned)(q0 + hb * 32 + ci) * DROP_C1);
                        nbr[i] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own[i], 0xB1, 0xf, 0xf, false);     // quad_perm [1, 0, 3, 2]
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        hbits[2 * i] = lane_odd ? nbr[i] : own[i];
                        hbits[2 * i + 1] = lane_odd ? own[i] : nbr[i];
                    }
                }
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int qr_ = q0 + hb * 32 + (t & 3) + 8 * (t >> 2) + 4 * hh;
                    float val = __builtin_fmaf(s[t], sl2, mk2);
                    if (MM == TRX_NN_MASK_FULL)   // wave-uniform base + a 32-bit offset: row (uniform) * Lk + this lane's key
                        val += fmaxf(mfull[(unsigned)(min(qr_, Lq - 1) * Lk) + (unsigned)kc] * L2E, -268435456.0f);
                    const float pr = __builtin_amdgcn_exp2f(fminf(val, 0.f));   // p <= 1
                    if (DROP) {
                        const unsigned bits = hbits[t];
                        const float km = (((bits >> dshift) & 0xffffu) >= da.thr) ? da.inv_keep : 0.f;
                        s[t] = pr * km;                                    // what the forward multiplied V with
                        p[t] = pr * __builtin_fmaf(p[t], km, ndv[t]);
                    } else {
                        s[t] = pr;
                        p[t] = pr * p[t];
                    }
                }
            }
            if (MM == TRX_NN_MASK_FULL) { TRX_BWD2_TRANSPOSED() }
#undef TRX_BWD2_TRANSPOSED
            // ---- dV^T += dO^T P ;  dK^T += Q^T dS ----
            TRX_BWD_TRWAIT(td0, 12)
            {
                const bf16x8 pb = TRX_BWD_PACK8(s, 0);
                av0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(td0, 0), pb, av0, 0, 0, 0);
                av1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(td0, 1), pb, av1, 0, 0, 0);
            }
            TRX_BWD_TRWAIT(td1, 8)
            {
                const bf16x8 pb = TRX_BWD_PACK8(s, 1);
                av0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(td1, 0), pb, av0, 0, 0, 0);
                av1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(td1, 1), pb, av1, 0, 0, 0);
            }
            TRX_BWD_TRWAIT(tq0, 4)
            {
                const bf16x8 db = TRX_BWD_PACK8(p, 0);
                ak0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(tq0, 0), db, ak0, 0, 0, 0);
                ak1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(tq0, 1), db, ak1, 0, 0, 0);
            }
            TRX_BWD_TRWAIT(tq1, 0)
            {
                const bf16x8 db = TRX_BWD_PACK8(p, 1);
                ak0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(tq1, 0), db, ak0, 0, 0, 0);
                ak1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_BWD_TRFRAG(tq1, 1), db, ak1, 0, 0, 0);
            }
        };
        half_tile(ic_<0>{});
        // wave-uniform: the second half lies wholly past the last query at Lq = 160's third tile and in every tile of a
        // decoder with 7 positions -- arithmetic on padding otherwise
        if (q0 + 32 < Lq) half_tile(ic_<1>{});
        buf = buf1;
    }
#undef TRX_BWD2_STAGE
    int b_e = b_s, h_e = h_s, kblk_e = kblk_s;      // opaque copies, as in the dq pass
    asm volatile("" : "+s"(b_e), "+s"(h_e), "+s"(kblk_e));
    const int lane_e = lane_again(), hh_e = lane_e >> 5, kidx_e = kblk_e * 128 + wave_s * 32 + (lane_e & 31);
    {
        const bool live = kidx_e < Lk;
        const bool wide = TRX_ATT_WIDE_STORE && kblk_e * 128 + 128 <= Lk;      // (a full block of 128 keys: every lane of the workgroup stores)
        const int64_t ro_e = ((int64_t)b_e * Lk + (live ? kidx_e : 0)) * (da.ldk ? da.ldk : H * 64) + h_e * 64;
        store_row_bf16(dk + ro_e, live, wide, hh_e, ak0, ak1, scale);
        store_row_bf16(dv + ro_e, live, wide, hh_e, av0, av1, 1.0f);
    }
}
