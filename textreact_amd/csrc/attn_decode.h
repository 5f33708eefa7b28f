// attn_decode.h -- one-token self-attention of beam-search decoding against a key/value cache that is NEVER re-ordered
// (included by nn_ops.hip; entry point trx_attention_decode_gather).
//
// Beam search keeps, per surviving beam, the keys / values of its ancestors.  Re-ordering the cache by the parents at
// every step (what Hugging Face's _reorder_cache does, main.py:218-226 via generate) moves the whole cache: 315 MB per
// layer at 32 inputs x 20 beams x 160 positions.  Here row i of the cache always holds what beam SLOT i wrote, and a
// small table anc[i][s] = the slot whose entry at position s belongs to beam i's history is re-ordered instead
// (n x T ints).  The kernel follows the table: one wave per (beam, head),
//   pass 1: lane = position: 128-byte key row of (anc[i][s], s), dot with q (v_dot2_f32_bf16), softmax over the wave;
//   pass 2: lane = dimension: out[d] = sum_s p_s * V[anc[i][s]][s][d] (one coalesced 128-byte row per position).
// The current length arrives through a device pointer, so that a captured graph can replay the launch for every step.
#pragma once

namespace {

constexpr int DEC_MAX_T = 256;     // positions a wave keeps probabilities for (max_dec_length is 160 in the scripts)

__global__ __launch_bounds__(256) void attention_decode_gather_kernel(const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kv,
                                                                      const int* __restrict__ anc, const long long* __restrict__ t_ptr,
                                                                      bf16_t* __restrict__ out, int n, int H, int T, float scale_log2e) {
    __shared__ float pl[4][DEC_MAX_T];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int w = blockIdx.x * 4 + wv;
    if (w >= n * H) return;                                   // wave-uniform
    const int i = w / H, h = w % H;
    int len = (int)(*t_ptr) + 1;                              // positions 0 .. t are filled
    len = __builtin_amdgcn_readfirstlane(len < 1 ? 1 : (len > T ? T : len));
    const int64_t pos_stride = (int64_t)2 * H * 64, row_stride = (int64_t)T * pos_stride;
    // q (64 bf16 = 8 x 16 bytes), the same for every lane
    uint4 qv[8];
    const uint4* qp = reinterpret_cast<const uint4*>(q + (int64_t)i * ldq + h * 64);
#pragma unroll
    for (int c = 0; c < 8; ++c) qv[c] = qp[c];
    const int np = (len + 63) >> 6;
    float sc[DEC_MAX_T / 64];
    int an[DEC_MAX_T / 64];
    float m = -INFINITY;
#pragma unroll
    for (int p = 0; p < DEC_MAX_T / 64; ++p) {
        sc[p] = -INFINITY; an[p] = 0;
        const int s = p * 64 + lane;
        if (p < np && s < len) {
            an[p] = anc[(int64_t)i * T + s];
            const uint4* kp = reinterpret_cast<const uint4*>(kv + an[p] * row_stride + s * pos_stride + h * 64);
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint4 kk = kp[c];
                const unsigned kw[4] = {kk.x, kk.y, kk.z, kk.w}, qw[4] = {qv[c].x, qv[c].y, qv[c].z, qv[c].w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, kw[e]), __builtin_bit_cast(bf16x2_t, qw[e]), acc, false);
            }
            sc[p] = acc * scale_log2e;
            m = fmaxf(m, sc[p]);
        }
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float l = 0.f;
#pragma unroll
    for (int p = 0; p < DEC_MAX_T / 64; ++p) {
        sc[p] = (p < np && p * 64 + lane < len) ? exp2f(sc[p] - m) : 0.f;
        l += sc[p];
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) l += __shfl_xor(l, o, 64);
    const float inv = 1.0f / l;
#pragma unroll
    for (int p = 0; p < DEC_MAX_T / 64; ++p)
        if (p < np) pl[wv][p * 64 + lane] = sc[p] * inv;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // pass 2: lane = dimension
    const bf16_t* vbase = kv + (int64_t)H * 64 + h * 64 + lane;
    float acc = 0.f;
#pragma unroll
    for (int p = 0; p < DEC_MAX_T / 64; ++p) {
        if (p >= np) break;
        const int cnt = min(64, len - p * 64);
        int s0 = 0;
        for (; s0 + 8 <= cnt; s0 += 8) {
            bf16_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int a = __builtin_amdgcn_readlane(an[p], s0 + u);
                v[u] = vbase[a * row_stride + (int64_t)(p * 64 + s0 + u) * pos_stride];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(pl[wv][p * 64 + s0 + u], __uint_as_float((unsigned)v[u] << 16), acc);
        }
        for (; s0 < cnt; ++s0) {
            const int a = __builtin_amdgcn_readlane(an[p], s0);
            acc = fmaf(pl[wv][p * 64 + s0], __uint_as_float((unsigned)vbase[a * row_stride + (int64_t)(p * 64 + s0) * pos_stride] << 16), acc);
        }
    }
    out[(int64_t)i * H * 64 + h * 64 + lane] = f2bf(acc);
}

}  // namespace
