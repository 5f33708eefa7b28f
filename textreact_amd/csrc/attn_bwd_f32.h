// attn_bwd_f32.h -- fp32 attention backward on the matrix cores (included by nn_ops.hip after attn_fwd_f32.h).
//
// The fp32 twin of attn_bwd_mfma.h on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate): the gradients of a
// --precision 32 run and of every fp32 parity test used to come from one lane per row on the VALU
// (attention_bwd_dq_kernel / attention_bwd_dkv_kernel).  Same two deterministic passes, no atomics; probabilities are
// recomputed from the forward's log-sum-exp:
//   pass 1, a lane per QUERY (64 queries per workgroup, 16 per wave, K / V tiles of 32 keys through LDS):
//       S^T = K Q^T, dP^T = V dO^T (both with the query on the lane, as in attn_fwd_f32.h), p = exp(s - lse),
//       dS = p (dP keep/(1-p_drop) - delta), dQ^T += K^T dS^T -- dS never leaves registers: register i of a 16-key block is
//       the B operand of the k-step whose keys are {4 g + i}, and K is read at those keys; delta = dO . O is made here
//       and handed to pass 2 through a scratch of B H Lq floats
//   pass 2, a lane per KEY (64 keys per workgroup, Q / dO tiles of 32 queries through LDS, lse and delta beside them):
//       S = Q K^T, dP = dO V^T (key on the lane, K and V fragments in registers), then dV^T += dO^T Pd and dK^T += Q^T dS
//       with Pd / dS straight from the accumulators.
// 7 GEMM units instead of the textbook 5 (S and dP are recomputed per pass) buy determinism.  Layouts, masks and dropout
// decisions are the VALU kernels'.  LDS rows are padded to 68 floats (attn_fwd_f32.h: conflict-free b128 and b32 reads).
#pragma once

template <int MM, bool DROP>
__global__ __launch_bounds__(256) void attention_bwd_dq_f32_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                        const float* __restrict__ v, const float* __restrict__ mask,
                                                                        int causal, int B, int H, int Lq, int Lk, float scale,
                                                                        const float* __restrict__ o, const float* __restrict__ dout,
                                                                        const float* __restrict__ lse, float* __restrict__ dq,
                                                                        float* __restrict__ delta_out, DropArgs da) {
    __shared__ __attribute__((aligned(16))) float sK[F32_KT * F32_PITCH];
    __shared__ __attribute__((aligned(16))) float sV[F32_KT * F32_PITCH];
    const int qblocks = (Lq + 63) / 64;
    const int bid = blockIdx.x;
    const int qb = bid % qblocks, h = (bid / qblocks) % H, b = bid / (qblocks * H);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qn = lane & 15, lg = lane >> 4;
    const int qi = qb * 64 + wave * 16 + qn;
    const int qc = qi < Lq ? qi : Lq - 1;
    const int64_t rs = (int64_t)H * 64;
    const float* kb_ = k + (int64_t)b * Lk * rs + (int64_t)h * 64;
    const float* vb_ = v + (int64_t)b * Lk * rs + (int64_t)h * 64;
    constexpr float LOG2E = 1.4426950408889634f;
    const float sl2 = scale * LOG2E;

    f32x4_t qf[4], dof[4];
    float delta = 0.f;
    {
        const int64_t ro = (((int64_t)b * Lq + qc) * H + h) * 64 + 4 * lg;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = *reinterpret_cast<const f32x4_t*>(q + ro + 16 * s);
            dof[s] = *reinterpret_cast<const f32x4_t*>(dout + ro + 16 * s);
            const f32x4_t ov = *reinterpret_cast<const f32x4_t*>(o + ro + 16 * s);
#pragma unroll
            for (int i = 0; i < 4; ++i) delta = __builtin_fmaf(dof[s][i], ov[i], delta);
        }
        delta += __shfl_xor(delta, 16, 64);
        delta += __shfl_xor(delta, 32, 64);
    }
    if (delta_out && lg == 0 && qi < Lq) delta_out[((int64_t)b * H + h) * Lq + qi] = delta;
    const float L2 = lse[((int64_t)b * H + h) * Lq + qc] * LOG2E;
    const int off = Lk - Lq;
    const int jend = causal ? min(Lk, qb * 64 + 63 + off + 1) : Lk;
    const int ntiles = jend > 0 ? (jend + F32_KT - 1) / F32_KT : 0;
    const int jmax_row = causal ? qc + off : Lk - 1;
    const unsigned dbase = DROP ? drop_base_da(da, (unsigned)(b * H + h)) : 0u;

    f32x4_t pk_[2], pv_[2];
    auto gload = [&](int t) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int idx = tid + 256 * r, row = idx >> 4, c4 = idx & 15;
            int key = t * F32_KT + row;
            key = key < Lk ? key : Lk - 1;
            pk_[r] = *reinterpret_cast<const f32x4_t*>(kb_ + (int64_t)key * rs + 4 * c4);
            pv_[r] = *reinterpret_cast<const f32x4_t*>(vb_ + (int64_t)key * rs + 4 * c4);
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int idx = tid + 256 * r, row = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<f32x4_t*>(sK + row * F32_PITCH + 4 * c4) = pk_[r];
            *reinterpret_cast<f32x4_t*>(sV + row * F32_PITCH + 4 * c4) = pv_[r];
        }
    };

    f32x4_t acc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) acc[db] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (ntiles > 0) { gload(0); sstore(); }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) gload(t + 1);
        f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, d0 = s0, d1 = s0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4_t k0 = *reinterpret_cast<const f32x4_t*>(sK + qn * F32_PITCH + 16 * s + 4 * lg);
            const f32x4_t k1 = *reinterpret_cast<const f32x4_t*>(sK + (16 + qn) * F32_PITCH + 16 * s + 4 * lg);
            const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(sV + qn * F32_PITCH + 16 * s + 4 * lg);
            const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(sV + (16 + qn) * F32_PITCH + 16 * s + 4 * lg);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(k0[i], qf[s][i], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k1[i], qf[s][i], s1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v0[i], dof[s][i], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v1[i], dof[s][i], d1, 0, 0, 0);
            }
        }
        const int key0 = t * F32_KT + 4 * lg;
        float ds[2][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = key0 + 16 * kb + i;
                const int kc = key < Lk ? key : Lk - 1;
                float a = (kb ? s1[i] : s0[i]) * sl2;
                if (MM == TRX_NN_MASK_KEY) a = __builtin_fmaf(mask[(int64_t)b * Lk + kc], LOG2E, a);
                else if (MM == TRX_NN_MASK_FULL) a = __builtin_fmaf(mask[((int64_t)b * Lq + qc) * Lk + kc], LOG2E, a);
                const float p = (key < Lk && key <= jmax_row) ? __builtin_amdgcn_exp2f(a - L2) : 0.f;
                float dp = kb ? d1[i] : d0[i];
                if (DROP) dp = drop_keep(drop_bits(dbase, (unsigned)qc, (unsigned)key >> 1), (unsigned)key, da.thr) ? dp * da.inv_keep : 0.f;
                ds[kb][i] = p * (dp - delta);
            }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* kr = sK + (16 * kb + 4 * lg + i) * F32_PITCH + qn;
#pragma unroll
                for (int db = 0; db < 4; ++db) acc[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[16 * db], ds[kb][i], acc[db], 0, 0, 0);
            }
        __syncthreads();
        if (t + 1 < ntiles) sstore();
        __syncthreads();
    }
    if (qi < Lq) {
        float* op = dq + (((int64_t)b * Lq + qi) * H + h) * 64 + 4 * lg;
#pragma unroll
        for (int db = 0; db < 4; ++db) *reinterpret_cast<f32x4_t*>(op + 16 * db) = acc[db] * scale;
    }
}

template <int MM, bool DROP>
__global__ __launch_bounds__(256) void attention_bwd_dkv_f32_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                         const float* __restrict__ v, const float* __restrict__ mask,
                                                                         int causal, int B, int H, int Lq, int Lk, float scale,
                                                                         const float* __restrict__ dout, const float* __restrict__ lse,
                                                                         const float* __restrict__ delta, float* __restrict__ dk,
                                                                         float* __restrict__ dv, DropArgs da) {
    __shared__ __attribute__((aligned(16))) float sQ[F32_KT * F32_PITCH];
    __shared__ __attribute__((aligned(16))) float sD[F32_KT * F32_PITCH];
    __shared__ float sL[F32_KT], sDl[F32_KT];
    const int kblocks = (Lk + 63) / 64;
    const int bid = blockIdx.x;
    const int kb_ = bid % kblocks, h = (bid / kblocks) % H, b = bid / (kblocks * H);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kn = lane & 15, lg = lane >> 4;
    const int kj = kb_ * 64 + wave * 16 + kn;          // this lane's key
    const int kc = kj < Lk ? kj : Lk - 1;
    const int64_t rs = (int64_t)H * 64;
    const float* qb_ = q + (int64_t)b * Lq * rs + (int64_t)h * 64;
    const float* db_ = dout + (int64_t)b * Lq * rs + (int64_t)h * 64;
    const float* lse_ = lse + ((int64_t)b * H + h) * Lq;
    const float* dl_ = delta + ((int64_t)b * H + h) * Lq;
    constexpr float LOG2E = 1.4426950408889634f;
    const float sl2 = scale * LOG2E;

    f32x4_t kf[4], vf[4];      // head components 16 s + 4 lg .. + 3 of this lane's key
    {
        const int64_t ro = (((int64_t)b * Lk + kc) * H + h) * 64 + 4 * lg;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = *reinterpret_cast<const f32x4_t*>(k + ro + 16 * s);
            vf[s] = *reinterpret_cast<const f32x4_t*>(v + ro + 16 * s);
        }
    }
    const float mkey = MM == TRX_NN_MASK_KEY ? mask[(int64_t)b * Lk + kc] * LOG2E : 0.f;
    const int off = Lk - Lq;
    // first query that can see any key of this workgroup (causal): i >= j - (Lk - Lq)
    const int i0 = causal ? max(0, kb_ * 64 - off) : 0;
    const int t0 = i0 / F32_KT, ntq = (Lq + F32_KT - 1) / F32_KT;
    const unsigned dbase = DROP ? drop_base_da(da, (unsigned)(b * H + h)) : 0u;

    f32x4_t pq_[2], pd_[2];
    float pl_ = 0.f, pdl_ = 0.f;
    auto gload = [&](int t) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int idx = tid + 256 * r, row = idx >> 4, c4 = idx & 15;
            int qr = t * F32_KT + row;
            qr = qr < Lq ? qr : Lq - 1;
            pq_[r] = *reinterpret_cast<const f32x4_t*>(qb_ + (int64_t)qr * rs + 4 * c4);
            pd_[r] = *reinterpret_cast<const f32x4_t*>(db_ + (int64_t)qr * rs + 4 * c4);
        }
        if (tid < F32_KT) {
            int qr = t * F32_KT + tid;
            qr = qr < Lq ? qr : Lq - 1;
            pl_ = lse_[qr] * LOG2E; pdl_ = dl_[qr];
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int idx = tid + 256 * r, row = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<f32x4_t*>(sQ + row * F32_PITCH + 4 * c4) = pq_[r];
            *reinterpret_cast<f32x4_t*>(sD + row * F32_PITCH + 4 * c4) = pd_[r];
        }
        if (tid < F32_KT) { sL[tid] = pl_; sDl[tid] = pdl_; }
    };

    f32x4_t ak[4], av[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) { ak[db] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; av[db] = ak[db]; }
    if (t0 < ntq) { gload(t0); sstore(); }
    __syncthreads();
    for (int t = t0; t < ntq; ++t) {
        if (t + 1 < ntq) gload(t + 1);
        // S = Q K^T and dP = dO V^T for the tile's two 16-query blocks (key on the lane)
        f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, d0 = s0, d1 = s0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4_t q0 = *reinterpret_cast<const f32x4_t*>(sQ + kn * F32_PITCH + 16 * s + 4 * lg);
            const f32x4_t q1 = *reinterpret_cast<const f32x4_t*>(sQ + (16 + kn) * F32_PITCH + 16 * s + 4 * lg);
            const f32x4_t g0 = *reinterpret_cast<const f32x4_t*>(sD + kn * F32_PITCH + 16 * s + 4 * lg);
            const f32x4_t g1 = *reinterpret_cast<const f32x4_t*>(sD + (16 + kn) * F32_PITCH + 16 * s + 4 * lg);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(q0[i], kf[s][i], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q1[i], kf[s][i], s1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(g0[i], vf[s][i], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(g1[i], vf[s][i], d1, 0, 0, 0);
            }
        }
        float pd[2][4], ds[2][4];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rl = 16 * blk + 4 * lg + i;          // row of the tile = query t * 32 + rl
                const int qr = t * F32_KT + rl;
                float a = __builtin_fmaf(blk ? s1[i] : s0[i], sl2, mkey);
                if (MM == TRX_NN_MASK_FULL) a = __builtin_fmaf(mask[((int64_t)b * Lq + (qr < Lq ? qr : Lq - 1)) * Lk + kc], LOG2E, a);
                const bool vis = qr < Lq && (!causal || kc <= qr + off);
                const float p = vis ? __builtin_amdgcn_exp2f(a - sL[rl]) : 0.f;
                float dp = blk ? d1[i] : d0[i];
                float pdv = p;                                   // the probability as the forward used it: dropped and rescaled
                if (DROP) {
                    const float km = drop_keep(drop_bits(dbase, (unsigned)qr, (unsigned)kc >> 1), (unsigned)kc, da.thr) ? da.inv_keep : 0.f;
                    pdv = p * km; dp *= km;
                }
                pd[blk][i] = pdv;
                ds[blk][i] = p * (dp - sDl[rl]);
            }
        // dV^T += dO^T Pd and dK^T += Q^T dS: the A operands are read at the queries {16 blk + 4 g + i}
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* gr = sD + (16 * blk + 4 * lg + i) * F32_PITCH + kn;
                const float* qr_ = sQ + (16 * blk + 4 * lg + i) * F32_PITCH + kn;
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    av[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(gr[16 * db], pd[blk][i], av[db], 0, 0, 0);
                    ak[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(qr_[16 * db], ds[blk][i], ak[db], 0, 0, 0);
                }
            }
        __syncthreads();
        if (t + 1 < ntq) sstore();
        __syncthreads();
    }
    if (kj < Lk) {
        const int64_t ro = (((int64_t)b * Lk + kj) * H + h) * 64 + 4 * lg;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            *reinterpret_cast<f32x4_t*>(dk + ro + 16 * db) = ak[db] * scale;
            *reinterpret_cast<f32x4_t*>(dv + ro + 16 * db) = av[db];
        }
    }
}
