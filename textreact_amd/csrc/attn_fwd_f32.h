// attn_fwd_f32.h -- fp32 attention forward on the matrix cores (included by nn_ops.hip).
//
// The --precision 32 path and every fp32 parity test used to run softmax(q k^T / 8 + mask) v one lane per query on the VALU
// (attention_fwd_kernel: 0.05-0.25 of the vector peak, slower than PyTorch eager at cross-attention).  gfx950 has an
// f32-in / f32-accumulate MFMA, v_mfma_f32_16x16x4_f32: exact fp32 (bitwise a k-ordered fmaf chain, one rounding per
// product) at the fp32 vector peak, 157 TFLOP/s -- so the arithmetic of this kernel is the VALU kernel's, not a reduced one.
//
// Workgroup = 4 waves = 64 queries of one (batch, head); a wave owns 16 queries and walks the keys in tiles of 32 that all
// 256 threads stage through LDS (next tile's global loads in flight under the current tile's math).  Orientation as in the
// bf16 kernel: the score tile is computed TRANSPOSED, S^T = K Q^T (A = K from LDS, B = Q held in registers), so the QUERY is
// on the lane (col = lane & 15) and a lane's 4 accumulator registers of a 16-key block are keys 4 (lane >> 4) + i: the
// softmax statistics are lane-local plus two exchanges (lane ^ 16, lane ^ 32), and the probabilities feed the second product
// O^T += V^T P^T without leaving registers -- register i of a block is the B operand of a k-step whose four k indices are the
// keys {4 g + i : g = lane >> 4}, and the A operand reads V at those same keys.  The contraction index of the first product
// is permuted the same way on both sides (lane group g, step i <-> head component 16 s + 4 g + i), which turns the K reads
// into ds_read_b128 and the Q loads into 16-byte global loads.  LDS rows are padded to 68 floats: the b128 reads of K (16
// rows x 16-byte columns 4 apart) and the b32 reads of V (4 key rows x 16 consecutive columns) are conflict-free.
// FLOPs = 4 B H Lq Lk 64.  Layouts, masks, dropout decisions and the log-sum-exp are attention_fwd_kernel's.
#pragma once

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int F32_KT = 32;         // keys per tile
constexpr int F32_PITCH = 68;      // floats per LDS row

template <int MM, bool DROP>
__global__ __launch_bounds__(256) void attention_fwd_f32_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                     const float* __restrict__ v, const float* __restrict__ mask,
                                                                     int causal, int B, int H, int Lq, int Lk, float scale,
                                                                     float* __restrict__ out, float* __restrict__ lse, DropArgs da) {
    __shared__ __attribute__((aligned(16))) float sK[F32_KT * F32_PITCH];
    __shared__ __attribute__((aligned(16))) float sV[F32_KT * F32_PITCH];
    const int qblocks = (Lq + 63) / 64;
    const int bid = blockIdx.x;
    const int qb = bid % qblocks, h = (bid / qblocks) % H, b = bid / (qblocks * H);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qn = lane & 15, lg = lane >> 4;
    const int qi = qb * 64 + wave * 16 + qn;        // this lane's query
    const int qc = qi < Lq ? qi : Lq - 1;
    const int64_t kvbs = da.kv_bs ? da.kv_bs : (int64_t)Lk * H * 64;
    const int64_t rs = (int64_t)H * 64;
    const float* kb_ = k + (int64_t)b * kvbs + (int64_t)h * 64;
    const float* vb_ = v + (int64_t)b * kvbs + (int64_t)h * 64;
    constexpr float LOG2E = 1.4426950408889634f;
    const float sl2 = scale * LOG2E;

    f32x4_t qf[4];      // head components 16 s + 4 lg .. + 3 of this lane's query
    {
        const float* qp = q + (((int64_t)b * Lq + qc) * H + h) * 64 + 4 * lg;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const f32x4_t*>(qp + 16 * s);
    }
    const int off = Lk - Lq;
    const int jend = causal ? min(Lk, qb * 64 + 63 + off + 1) : Lk;      // workgroup-uniform
    const int ntiles = jend > 0 ? (jend + F32_KT - 1) / F32_KT : 0;
    const int jmax_row = causal ? qc + off : Lk - 1;                      // last visible key of this lane's query
    const unsigned dbase = DROP ? drop_base_da(da, (unsigned)(b * H + h)) : 0u;

    // staging: a tile is 32 rows x 16 float4 per matrix; thread -> chunks tid and tid + 256
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

    float m = -__builtin_inff(), lsum = 0.f;
    f32x4_t o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (ntiles > 0) { gload(0); sstore(); }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) gload(t + 1);
        // ---- S^T = K Q^T: two 16-key blocks, two independent accumulator chains ----
        f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4_t k0 = *reinterpret_cast<const f32x4_t*>(sK + qn * F32_PITCH + 16 * s + 4 * lg);
            const f32x4_t k1 = *reinterpret_cast<const f32x4_t*>(sK + (16 + qn) * F32_PITCH + 16 * s + 4 * lg);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(k0[i], qf[s][i], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k1[i], qf[s][i], s1, 0, 0, 0);
            }
        }
        // ---- masks, online softmax (base 2) ----
        const int key0 = t * F32_KT + 4 * lg;       // register i of block kb: key key0 + 16 kb + i
        float x[2][4];
        float mb = -__builtin_inff();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = key0 + 16 * kb + i;
                const int kc = key < Lk ? key : Lk - 1;
                float a = (kb ? s1[i] : s0[i]) * sl2;
                if (MM == TRX_NN_MASK_KEY) a = __builtin_fmaf(mask[(int64_t)b * Lk + kc], LOG2E, a);
                else if (MM == TRX_NN_MASK_FULL) a = __builtin_fmaf(mask[((int64_t)b * Lq + qc) * Lk + kc], LOG2E, a);
                a = (key < Lk && key <= jmax_row) ? a : -__builtin_inff();      // hidden keys never set the maximum
                x[kb][i] = a;
                mb = fmaxf(mb, a);
            }
        mb = fmaxf(mb, __shfl_xor(mb, 16, 64));
        mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
        const float mn = fmaxf(m, mb);
        const float mref = (mn == -__builtin_inff()) ? 0.f : mn;      // everything hidden so far: exp2(-inf - 0) = 0, no NaN
        const float alpha = __builtin_amdgcn_exp2f(m - mref);
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float pj = __builtin_amdgcn_exp2f(x[kb][i] - mref);
                ps += pj;
                if (DROP) {
                    const unsigned key = (unsigned)(key0 + 16 * kb + i);
                    pj = drop_keep(drop_bits(dbase, (unsigned)qc, key >> 1), key, da.thr) ? pj : 0.f;
                }
                x[kb][i] = pj;
            }
        lsum = lsum * alpha + ps;       // this lane's share of the row sum (its 8 keys of every tile); joined at the end
        m = mn;
#pragma unroll
        for (int db = 0; db < 4; ++db) o[db] *= alpha;
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* vr = sV + (16 * kb + 4 * lg + i) * F32_PITCH + qn;
#pragma unroll
                for (int db = 0; db < 4; ++db) o[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[16 * db], x[kb][i], o[db], 0, 0, 0);
            }
        __syncthreads();
        if (t + 1 < ntiles) sstore();
        __syncthreads();
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    if (qi < Lq) {
        if (lse && lg == 0) lse[((int64_t)b * H + h) * Lq + qi] = (m + __builtin_amdgcn_logf(lsum)) * 0.6931471805599453f;   // v_log_f32 is log2
        const float inv = (DROP ? da.inv_keep : 1.0f) / lsum;
        float* op = out + ((int64_t)b * Lq + qi) * H * 64 + (int64_t)h * 64 + 4 * lg;
#pragma unroll
        for (int db = 0; db < 4; ++db) *reinterpret_cast<f32x4_t*>(op + 16 * db) = o[db] * inv;
    }
}
