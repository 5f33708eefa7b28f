// attn_valu.h -- LAB ONLY (make -C textreact_amd/csrc lab -> libtrxnn_lab.so, -DTRX_NN_LAB; TRX_NN_ATTN_VALU=1 selects them).
// The round-1 attention kernels on the vector ALUs, one lane per query (forward) / per key (backward pass 2).  Replaced in the
// product by the matrix-core kernels of nn_ops.hip (bf16: attention_fwd_mfma_kernel, attn_bwd_mfma.h; fp32: attn_fwd_f32.h,
// attn_bwd_f32.h), which are 5-20x faster on every shape measured (profiles/r05_attention_f32.jsonl).  Included by nn_ops.hip
// inside its anonymous namespace, twice: TRX_VALU_PART 1 = forward, 2 = backward.
#if TRX_VALU_PART == 1
// ---- attention forward, fp32 math, one lane per query row --------------------------------------
// A wave owns 64 consecutive queries of one (batch, head); K and V rows are the same for every lane,
// so hipcc fetches them through the scalar cache (s_load) and the products are v_fmac with an SGPR
// operand: no LDS, no barriers.  q and the running output stay in registers (2 x 64 VGPRs), keys are
// consumed in chunks of 8 with an online softmax.  FLOPs = 4 * B * H * Lq * Lk * 64.
constexpr int DH = 64, KC = 8;
template <bool BF, bool DROP>   // DROP is a template flag: the hash in the key loop costs the plain variant its scalar registers
__global__ __launch_bounds__(64) void attention_fwd_kernel(const void* __restrict__ q, const void* __restrict__ k,
                                                           const void* __restrict__ v, const float* __restrict__ mask,
                                                           int mask_mode, int causal, int B, int H, int Lq, int Lk,
                                                           float scale, void* __restrict__ out, float* __restrict__ lse, DropArgs da) {
    const int qblocks = (Lq + 63) / 64;
    const int bid = blockIdx.x;
    const int qb = bid % qblocks, h = (bid / qblocks) % H, b = bid / (qblocks * H);
    const int i = qb * 64 + threadIdx.x;
    const bool live = i < Lq;
    const int ii = live ? i : Lq - 1;
    const unsigned dbase = DROP ? drop_base_da(da, (unsigned)(b * H + h)) : 0u;
    const int64_t kvbs = da.kv_bs ? da.kv_bs : (int64_t)Lk * H * DH;
    float qr[DH], o[DH];
    const int64_t qoff = (((int64_t)b * Lq + ii) * H + h) * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) { qr[d] = ld<BF>(q, qoff + d) * scale; o[d] = 0.f; }
    float m = -__builtin_inff(), l = 0.f;
    const int jmax_row = causal ? ii + (Lk - Lq) : Lk - 1;            // last visible key of this row
    const int jend = causal ? min(Lk, qb * 64 + 63 + (Lk - Lq) + 1) : Lk;  // wave-uniform bound
    for (int j0 = 0; j0 < jend; j0 += KC) {
        float s[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int j = j0 + c;
            const int jj = j < Lk ? j : Lk - 1;
            const int64_t koff = (int64_t)b * kvbs + ((int64_t)jj * H + h) * DH;   // wave-uniform
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) a = __builtin_fmaf(qr[d], ld<BF>(k, koff + d), a);
            if (mask_mode == TRX_NN_MASK_KEY) a += mask[(int64_t)b * Lk + jj];
            else if (mask_mode == TRX_NN_MASK_FULL) a += mask[((int64_t)b * Lq + ii) * Lk + jj];
            s[c] = (j < Lk && j <= jmax_row) ? a : -__builtin_inff();   // hidden keys never set the maximum
        }
        float cm = s[0];
#pragma unroll
        for (int c = 1; c < KC; ++c) cm = fmaxf(cm, s[c]);
        const float mn = fmaxf(m, cm);
        const float alpha = (m == mn) ? 1.0f : __expf(m - mn);   // also covers m == mn == -inf
        l *= alpha;
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] *= alpha;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int j = j0 + c;
            const int jj = j < Lk ? j : Lk - 1;
            // rows hidden by the causal / length bound contribute exactly 0 (additive masks with a
            // finite large negative value behave like the reference: exp underflows to 0)
            float pj = (j < Lk && j <= jmax_row) ? __expf(s[c] - mn) : 0.f;
            l += pj;
            if (DROP) pj = drop_keep(drop_bits(dbase, (unsigned)ii, (unsigned)j >> 1), (unsigned)j, da.thr) ? pj : 0.f;
            const int64_t voff = (int64_t)b * kvbs + ((int64_t)jj * H + h) * DH;
#pragma unroll
            for (int d = 0; d < DH; ++d) o[d] = __builtin_fmaf(pj, ld<BF>(v, voff + d), o[d]);
        }
        m = mn;
    }
    if (live && lse) lse[((int64_t)b * H + h) * Lq + i] = m + __logf(l);
    if (live) {
        const float inv = (DROP ? da.inv_keep : 1.0f) / l;
        const int64_t ooff = ((int64_t)b * Lq + i) * H * DH + (int64_t)h * DH;
#pragma unroll
        for (int d = 0; d < DH; ++d) st<BF>(out, ooff + d, o[d] * inv);
    }
}

#else
// ---- attention backward, fp32 math, probabilities recomputed from lse --------------------------
// pass 1 (a lane per query row i):  delta_i = dO_i . O_i ;  dS_ij = p_ij (dO_i . V_j - delta_i) ;
//                                   dQ_i = scale * sum_j dS_ij K_j
// pass 2 (a lane per key row j):    dV_j = sum_i p_ij dO_i ;  dK_j = scale * sum_i dS_ij Q_i
// In both passes the "other" operand row (K_j, V_j / Q_i, dO_i) is wave-uniform -> scalar loads.
template <bool BF, bool DROP>
__global__ __launch_bounds__(64) void attention_bwd_dq_kernel(const void* __restrict__ q, const void* __restrict__ k,
                                                              const void* __restrict__ v, const float* __restrict__ mask,
                                                              int mask_mode, int causal, int B, int H, int Lq, int Lk,
                                                              float scale, const void* __restrict__ o, const void* __restrict__ dout,
                                                              const float* __restrict__ lse, void* __restrict__ dq, DropArgs da) {
    const int qblocks = (Lq + 63) / 64;
    const int bid = blockIdx.x;
    const int qb = bid % qblocks, h = (bid / qblocks) % H, b = bid / (qblocks * H);
    const int i = qb * 64 + threadIdx.x;
    const bool live = i < Lq;
    const int ii = live ? i : Lq - 1;
    const unsigned dbase = DROP ? drop_base_da(da, (unsigned)(b * H + h)) : 0u;
    float qr[DH], dor[DH], acc[DH];
    const int64_t qoff = (((int64_t)b * Lq + ii) * H + h) * DH;
    const int64_t ooff = ((int64_t)b * Lq + ii) * H * DH + (int64_t)h * DH;
    float delta = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) {
        qr[d] = ld<BF>(q, qoff + d) * scale; dor[d] = ld<BF>(dout, ooff + d); acc[d] = 0.f;
        delta = __builtin_fmaf(dor[d], ld<BF>(o, ooff + d), delta);
    }
    const float L = lse[((int64_t)b * H + h) * Lq + ii];
    const int jmax_row = causal ? ii + (Lk - Lq) : Lk - 1;
    const int jend = causal ? min(Lk, qb * 64 + 63 + (Lk - Lq) + 1) : Lk;
    for (int j = 0; j < jend; ++j) {
        const int64_t koff = (((int64_t)b * Lk + j) * H + h) * DH;   // wave-uniform
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) { s = __builtin_fmaf(qr[d], ld<BF>(k, koff + d), s); dp = __builtin_fmaf(dor[d], ld<BF>(v, koff + d), dp); }
        if (mask_mode == TRX_NN_MASK_KEY) s += mask[(int64_t)b * Lk + j];
        else if (mask_mode == TRX_NN_MASK_FULL) s += mask[((int64_t)b * Lq + ii) * Lk + j];
        const float p = j <= jmax_row ? __expf(s - L) : 0.f;
        if (DROP) dp = drop_keep(drop_bits(dbase, (unsigned)ii, (unsigned)j >> 1), (unsigned)j, da.thr) ? dp * da.inv_keep : 0.f;
        const float ds = p * (dp - delta);
#pragma unroll
        for (int d = 0; d < DH; ++d) acc[d] = __builtin_fmaf(ds, ld<BF>(k, koff + d), acc[d]);
    }
    if (live) {
#pragma unroll
        for (int d = 0; d < DH; ++d) st<BF>(dq, qoff + d, acc[d] * scale);
    }
}

template <bool BF, bool DROP>
__global__ __launch_bounds__(64) void attention_bwd_dkv_kernel(const void* __restrict__ q, const void* __restrict__ k,
                                                               const void* __restrict__ v, const float* __restrict__ mask,
                                                               int mask_mode, int causal, int B, int H, int Lq, int Lk,
                                                               float scale, const void* __restrict__ o, const void* __restrict__ dout,
                                                               const float* __restrict__ lse, void* __restrict__ dk, void* __restrict__ dv, DropArgs da) {
    const int kblocks = (Lk + 63) / 64;
    const int bid = blockIdx.x;
    const int kb = bid % kblocks, h = (bid / kblocks) % H, b = bid / (kblocks * H);
    const unsigned dbase = DROP ? drop_base_da(da, (unsigned)(b * H + h)) : 0u;
    const int j = kb * 64 + threadIdx.x;
    const bool live = j < Lk;
    const int jj = live ? j : Lk - 1;
    float kr[DH], vr[DH], ak[DH], av[DH];
    const int64_t koff = (((int64_t)b * Lk + jj) * H + h) * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) { kr[d] = ld<BF>(k, koff + d); vr[d] = ld<BF>(v, koff + d); ak[d] = 0.f; av[d] = 0.f; }
    const float mkey = mask_mode == TRX_NN_MASK_KEY ? mask[(int64_t)b * Lk + jj] : 0.f;
    // first query row that can see any key of this block (causal): i >= j - (Lk - Lq)
    const int i0 = causal ? max(0, kb * 64 - (Lk - Lq)) : 0;
    for (int i = i0; i < Lq; ++i) {
        const int64_t qoff = (((int64_t)b * Lq + i) * H + h) * DH;          // wave-uniform
        const int64_t ooff = ((int64_t)b * Lq + i) * H * DH + (int64_t)h * DH;
        float s = 0.f, dp = 0.f, delta = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            const float qd = ld<BF>(q, qoff + d), dod = ld<BF>(dout, ooff + d);
            s = __builtin_fmaf(qd, kr[d], s); dp = __builtin_fmaf(dod, vr[d], dp);
            delta = __builtin_fmaf(dod, ld<BF>(o, ooff + d), delta);        // wave-uniform value
        }
        s = s * scale + mkey;
        if (mask_mode == TRX_NN_MASK_FULL) s += mask[((int64_t)b * Lq + i) * Lk + jj];
        const float L = lse[((int64_t)b * H + h) * Lq + i];
        const bool vis = !causal || jj <= i + (Lk - Lq);
        const float p = vis ? __expf(s - L) : 0.f;
        float pd = p;   // the probability as the forward used it for the output: dropped and rescaled
        if (DROP) {
            const float km = drop_keep(drop_bits(dbase, (unsigned)i, (unsigned)jj >> 1), (unsigned)jj, da.thr) ? da.inv_keep : 0.f;
            pd = p * km; dp *= km;
        }
        const float ds = p * (dp - delta);
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            av[d] = __builtin_fmaf(pd, ld<BF>(dout, ooff + d), av[d]);
            ak[d] = __builtin_fmaf(ds, ld<BF>(q, qoff + d), ak[d]);
        }
    }
    if (live) {
#pragma unroll
        for (int d = 0; d < DH; ++d) { st<BF>(dk, koff + d, ak[d] * scale); st<BF>(dv, koff + d, av[d]); }
    }
}

#endif
