// attn_aug.h -- LAB ONLY (make -C textreact_amd/csrc nnvar NAME=aug D=-DTRX_ATT_AUG=1): the forward tile with the scale and the
// softmax reference folded into the first product (round 6).  It removes the 16 v_pk_fma_f32 of a wave-tile -- 849 -> 738 vector
// issue cycles on the common path (profiles/r06_attention_valu_table.json) -- and is NOT faster: 43.7 against 43.9 us at
// 512 x 512, 11.1 against 10.7 us at causal 160 x 160 (profiles/r06_attention_aug_ab.json).  The ablations beside it
// (profiles/r06_attention_ablation.json) say why: the tile's vector arithmetic is not what the launch waits for.
// Included by nn_ops.hip inside its anonymous namespace when TRX_ATT_AUG is 1.
// Round 6: the same tile with the scale and the reference taken out of the vector ALUs.  Q arrives pre-multiplied by
// scale * log2 e (once per workgroup), the mask in the same units, and ONE more k-step of the first product carries the
// reference: A = ones in three slots, B = -mref as three bf16 terms (exact: 3 x 8 significant bits), so that the accumulators
// leave the matrix core as s'' = score * scale * log2 e + mask - mref -- the exponent's argument itself.  What the vector ALUs
// did per score element (v_pk_fma_f32: 16 per tile, 8 issue cycles each -- 128 of a wave-tile's ~860, tools/isa_valu_table.py)
// becomes two MFMAs (16 issue cycles).  mref is the reference the NEXT tile's MFMAs will subtract; it follows the lazy reference
// m, and when m moves (a wave-uniform, rare branch) the tile at hand is corrected in registers by delta = new - old.  A reference
// of mask-floor size (every key so far masked) is never handed to the matrix core -- s' + 2^28 would round the scores away --:
// mref stays 0 then and such rows take the branch every tile.
__device__ __forceinline__ bf16x8 attn_aug_operand(float x, int hh) {      // x as hi + mid + lo in slots 0..2 (lanes hh == 0), else zeros
    const unsigned xb = __builtin_bit_cast(unsigned, x);
    const unsigned hb = xb & 0xffff0000u;
    const float r1 = x - __builtin_bit_cast(float, hb);
    const unsigned mb = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, mb);
    const unsigned w0 = hh ? 0u : ((hb >> 16) | mb);
    const unsigned w1 = hh ? 0u : (__builtin_bit_cast(unsigned, r2) >> 16);      // r2 has at most 8 significant bits: exact
    return __builtin_bit_cast(bf16x8, uint4{w0, w1, 0u, 0u});
}
template <bool DROP>
__device__ __forceinline__ void attn_softmax_tile_aug(f32x16& s0, f32x16& s1, f32x16& o0, f32x16& o1, float& m, float& lsum, float& mref,
                                                      bf16x8& qaug, int hh, unsigned xd, unsigned thr, unsigned (&pk)[2][8]) {
    float mb = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < 16; ++t) mb = fmaxf(mb, s0[t]);       // (one chain: hipcc folds it into v_max3; a tree gets a canonicalising
#pragma unroll
    for (int t = 0; t < 16; ++t) mb = fmaxf(mb, s1[t]);       //  v_max x, x in front of every leaf)
    {
        float ma = mb, mc = mb;      // (v_permlane32_swap: see attn_softmax_tile)
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mc));
        mb = fmaxf(ma, mc) + mref;   // the row's maximum in absolute units (-inf stays -inf)
    }
    const float mt_ = fmaxf(m, mb);
    const float mn = (TRX_ATT_LAZY > 0 && !(mt_ > m + (float)TRX_ATT_LAZY)) ? m : mt_;
    const float nref = (mn == -__builtin_inff()) ? 0.f : mn;
    const float alpha = __builtin_amdgcn_exp2f(m - nref);
    const float delta = nref - mref;
    if (__builtin_amdgcn_ballot_w64(delta != 0.f || alpha != 1.0f) != 0) {      // wave-uniform: the reference moved for some lane
#pragma unroll
        for (int t = 0; t < 16; ++t) { s0[t] -= delta; s1[t] -= delta; o0[t] *= alpha; o1[t] *= alpha; }
        mref = fabsf(nref) < 16777216.0f ? nref : 0.f;
        qaug = attn_aug_operand(-mref, hh);
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) { s0[t] = __builtin_amdgcn_exp2f(s0[t]); s1[t] = __builtin_amdgcn_exp2f(s1[t]); }
    float ps = 0.f;
    if (DROP) {
#pragma unroll
        for (int t = 0; t < 16; ++t) ps += s0[t] + s1[t];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                const unsigned bits = lowbias32(xd + (unsigned)(hb * 16 + ((t & 3) >> 1) + 4 * (t >> 2)) * DROP_C2);
                const float e0 = hb ? s1[t] : s0[t], e1 = hb ? s1[t + 1] : s0[t + 1];
                pk[hb][t >> 1] = pack2bf(drop_keep(bits, 0, thr) ? e0 : 0.f, drop_keep(bits, 1, thr) ? e1 : 0.f);
            }
    } else {
        const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                const unsigned w = pack2bf(hb ? s1[t] : s0[t], hb ? s1[t + 1] : s0[t + 1]);
                pk[hb][t >> 1] = w;
                ps = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), ones, ps, false);
            }
    }
    lsum = lsum * alpha + ps;
    m = mn;
}

