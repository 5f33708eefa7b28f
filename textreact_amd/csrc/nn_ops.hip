// nn_ops.hip -- predictor hot spots on gfx950.  C ABI in include/trx_nn.h.  Wave = 64.
//
// Replaces (inside the Hugging Face modules the reference instantiates at textreact/model.py:21-31):
//   LayerNorm(dropout(dense(h)) + residual)  -- BertSelfOutput / BertOutput / embeddings / lm_head
//   softmax(q k^T / 8 + mask) v, with dropout on the probabilities in training
//                                            -- Bert/Roberta self-, cross- and causal attention, heads of 64
// In this file: the dropout decision function; fused (dropout +) add + LayerNorm forward / backward
// (vector kernels incl. the bf16-x / fp32-stream case of autocast, scalar fallbacks); attention forward
// in fp32 (VALU, the <= 1e-3 parity path) and in bf16 on the matrix cores; the fp32 attention backward;
// the C entry points.  The matrix-core attention backward is attn_bwd_mfma.h (included below).
#include <atomic>
#include "../../include/trx_nn.h"
#include <hip/hip_runtime.h>
#include <string>
#include <cstdlib>
#include <cmath>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& m) { g_err = m; return code; }

// ---- dropout: a counter-based decision per element, recomputed wherever it is needed --------------
// keep(stream, row, col) depends only on (seed, stream, row, col): the forward kernels, both backward
// passes and trx_dropout_keep_mask evaluate the same function, nothing is stored.  One 32-bit hash
// (lowbias32, two multiply-xorshift rounds) serves the two columns 2c, 2c+1 with 16 bits each;
// an element is dropped when its 16 bits are below thr = round(p * 65536).
// Seed source.  By default a launch carries its 64-bit seed by value.  With trx_nn_set_seed_device(ptr) every dropout launch
// carries ptr as well and its kernels take  mix(*ptr, seed)  as the seed: `seed` then only numbers the dropout site, and
// the value behind ptr -- bumped once per optimisation step by the caller, on the stream -- makes the decisions change from
// step to step although the launch arguments do not: what a HIP graph of the whole training step needs.
__host__ __device__ __forceinline__ unsigned long long mix_seed(unsigned long long dev, unsigned long long site) {
    unsigned long long z = dev + site * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct Drop { unsigned base, thr; float inv_keep; const unsigned long long* seed_dev; unsigned seed_lo, seed_hi; };   // thr == 0: dropout off
// attention: the dropout base is per (batch, head).  kv_bs: elements between consecutive batch items of k and
// of v (0 = dense, Lk * H * 64); set by the key/value-cache entry point, forward kernels only
// ldq / ldk: elements between consecutive rows (tokens) of q and of k / v (0 = dense, H * 64): operands that are
// slices of one packed projection output [B, L, 3 * H * 64]; matrix-core kernels only
struct DropArgs { unsigned seed_lo, seed_hi, thr; float inv_keep; long long kv_bs; int ldq, ldk; const unsigned long long* seed_dev; };
__host__ __device__ __forceinline__ unsigned lowbias32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ unsigned drop_base(unsigned seed_lo, unsigned seed_hi, unsigned stream) {
    return lowbias32(seed_lo ^ lowbias32(seed_hi + stream * 0x9E3779B9u));
}
// the seed a kernel uses: the launch's own, or mixed with the device-side value (see mix_seed)
__device__ __forceinline__ unsigned drop_base_da(const DropArgs& da, unsigned stream) {
    unsigned lo = da.seed_lo, hi = da.seed_hi;
    if (da.seed_dev) { const unsigned long long s = mix_seed(*da.seed_dev, ((unsigned long long)hi << 32) | lo); lo = (unsigned)s; hi = (unsigned)(s >> 32); }
    return drop_base(lo, hi, stream);
}
constexpr unsigned DROP_C1 = 0x9E3779B1u, DROP_C2 = 0x85EBCA77u;
__device__ __forceinline__ unsigned drop_bits(unsigned base, unsigned row, unsigned colpair) {
    return lowbias32(base + row * DROP_C1 + colpair * DROP_C2);
}
__device__ __forceinline__ bool drop_keep(unsigned bits, unsigned col, unsigned thr) {
    return ((bits >> ((col & 1u) << 4)) & 0xffffu) >= thr;
}
__device__ __forceinline__ float drop_mult(const Drop& d, unsigned row, unsigned col) {   // 0 or 1/(1-p)
    if (d.thr == 0) return 1.f;
    return drop_keep(drop_bits(d.base, row, col >> 1), col, d.thr) ? d.inv_keep : 0.f;
}
__device__ __forceinline__ Drop resolve_drop(Drop d) {     // add+LayerNorm kernels: the base of stream 0
    if (d.seed_dev) {
        const unsigned long long s = mix_seed(*d.seed_dev, ((unsigned long long)d.seed_hi << 32) | d.seed_lo);
        d.base = drop_base((unsigned)s, (unsigned)(s >> 32), 0u);
    }
    return d;
}
static const unsigned long long* g_seed_dev = nullptr;       // trx_nn_set_seed_device
unsigned drop_thr(float p) {
    const long t = lrintf(p * 65536.0f);
    return (unsigned)(t < 0 ? 0 : (t > 65535 ? 65535 : t));
}
Drop make_drop(float p, uint64_t seed, unsigned stream) {
    Drop d;
    d.thr = p > 0.f ? drop_thr(p) : 0u;
    d.base = drop_base((unsigned)seed, (unsigned)(seed >> 32), stream);
    d.seed_dev = g_seed_dev; d.seed_lo = (unsigned)seed; d.seed_hi = (unsigned)(seed >> 32);
    d.inv_keep = 1.0f / (1.0f - p);
    return d;
}
__global__ void dropout_keep_mask_kernel(unsigned seed_lo, unsigned seed_hi, unsigned thr, int64_t streams, int64_t rows,
                                         int64_t cols, unsigned char* __restrict__ keep, const unsigned long long* seed_dev) {
    if (seed_dev) { const unsigned long long s = mix_seed(*seed_dev, ((unsigned long long)seed_hi << 32) | seed_lo); seed_lo = (unsigned)s; seed_hi = (unsigned)(s >> 32); }
    const int64_t n = streams * rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t col = i % cols, sr = i / cols;
        const unsigned base = drop_base(seed_lo, seed_hi, (unsigned)(sr / rows));
        keep[i] = drop_keep(drop_bits(base, (unsigned)(sr % rows), (unsigned)(col >> 1)), (unsigned)col, thr) ? 1 : 0;
    }
}

typedef unsigned short bf16_t;
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// round-to-nearest-even, quiet NaN: gfx950's v_cvt_pk_bf16_f32
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ unsigned pack2bf(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
template <bool BF> __device__ __forceinline__ float ld(const void* p, int64_t i) {
    return BF ? bf2f(reinterpret_cast<const bf16_t*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}
template <bool BF> __device__ __forceinline__ void st(void* p, int64_t i, float v) {
    if (BF) reinterpret_cast<bf16_t*>(p)[i] = f2bf(v); else reinterpret_cast<float*>(p)[i] = v;
}
// sum over the 64 lanes, the same value in every lane: four DPP steps inside each row of 16 lanes (quad swaps, half-mirror,
// mirror), then the four row sums through v_readlane.  (__shfl_xor compiles to six ds_bpermute_b32 round trips through the
// LDS crossbar, ~100 cycles each and serial: the reductions were most of a row's latency in the LayerNorm kernels.)
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);    // quad_perm [1, 0, 3, 2]
    v += dpp_f32<0x4E>(v);    // quad_perm [2, 3, 0, 1]
    v += dpp_f32<0x141>(v);   // row_half_mirror
    v += dpp_f32<0x140>(v);   // row_mirror
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

// ---- add + LayerNorm forward: one wave per row, the row cached in registers -----------------------
// HBM-bound: algorithmic bytes = rows * cols * (x + res + y) * sizeof(dtype).
// Fast path: 16 bytes per lane per access (8 bf16 / 4 f32), up to NCH chunks per lane in registers
// (cols <= 2048 bf16 / 1024 f32; 768 = 1.5 / 3 chunks per lane).  Other shapes: scalar fallback.
constexpr int CPL = 32;  // scalar fallback: cached values per lane
constexpr int NCH = 4;
inline int chunks_per_lane(int cols, int V) { return (cols / V + 63) / 64; }      // 1 .. NCH on the vector path
#define TRX_NC_SWITCH(NC, M) switch (NC) { case 1: { M(1) } break; case 2: { M(2) } break; case 3: { M(3) } break; default: { M(4) } break; }
template <bool BF> struct Vec16 { static constexpr int N = BF ? 8 : 4; };
template <bool BF> __device__ __forceinline__ void unpack16(const uint4& u, float* f) {
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
    if (BF) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(w[i]);
    }
}
template <bool BF> __device__ __forceinline__ uint4 pack16(const float* f) {
    unsigned w[4];
    if (BF) {
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = pack2bf(f[2 * i], f[2 * i + 1]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = __float_as_uint(f[i]);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// 4 bf16 (8 bytes) -> 4 floats: the x operand of the mixed case (bf16 dense output, fp32 residual stream)
__device__ __forceinline__ void unpack8bf(const uint2& u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
}
__device__ __forceinline__ uint2 pack8bf(const float* f) { return make_uint2(pack2bf(f[0], f[1]), pack2bf(f[2], f[3])); }

// MIX: x is bf16 while res / y are fp32 (then BF is false and a lane step is 4 elements: 8 bytes of x)
template <bool BF, bool MIX, int NC>   // NC: 16-byte chunks a lane holds (cols <= 64 * NC * V): 768 columns are 2 (bf16) or 3 (fp32), and registers cut to that keep more rows in flight
__global__ __launch_bounds__(256) void add_ln_fwd_vec_kernel(const void* x, const void* res, const float* gamma,
                                                             const float* beta, float eps, int64_t rows, int cols,
                                                             void* y, float* mean, float* rstd, Drop drop_in, void* y16 = nullptr,
                                                             const float* xbias = nullptr) {
    const Drop drop = resolve_drop(drop_in);
    constexpr int V = Vec16<BF>::N;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = cols / V;
    // Every load of the row first -- x, res, the bias of x, gamma, beta, as whole 16- / 8-byte pieces with no branch between
    // them (an absent operand is read from gamma and ANDed away, a chunk past the row's end re-reads the last one) -- then the
    // arithmetic: the row costs one memory latency, not one per operand and chunk (round 5; see the backward).  (One row per
    // wave and 4,096 small workgroups stay: a row loop over a grid of whole rounds -- 1,024 / 1,280 / 1,366 resident workgroups,
    // gamma and beta loaded once -- measured 24.1-24.4 us against 23.3, tools/r05/twentysecond.sh.)
    const char* xr = reinterpret_cast<const char*>(x) + row * cols * ((BF || MIX) ? 2 : 4);
    const bool has_res = res != nullptr, has_xb = MIX && xbias != nullptr;
    const char* rr = has_res ? reinterpret_cast<const char*>(res) + row * cols * (BF ? 2 : 4) : reinterpret_cast<const char*>(gamma);
    const char* xbp = reinterpret_cast<const char*>(has_xb ? xbias : gamma);
    const unsigned m_res = has_res ? ~0u : 0u, m_xb = has_xb ? ~0u : 0u;
    char* yr = reinterpret_cast<char*>(y) + row * cols * (BF ? 2 : 4);
    uint4 rx4[MIX ? 1 : NC], rres[NC], rxb[MIX ? NC : 1], rg[NC][V / 4], rb[NC][V / 4];
    uint2 rx2[MIX ? NC : 1];
    unsigned lm[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        lm[i] = c < nchunk ? ~0u : 0u;
        const unsigned o16 = (unsigned)min(c, nchunk - 1) * 16u;
        if (MIX) {
            rx2[i] = *reinterpret_cast<const uint2*>(xr + (o16 >> 1));
            rxb[i] = *reinterpret_cast<const uint4*>(xbp + o16);
        } else rx4[i] = *reinterpret_cast<const uint4*>(xr + o16);
        rres[i] = *reinterpret_cast<const uint4*>(rr + o16);
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const size_t oc = (size_t)min(lane + 64 * i, nchunk - 1) * V;
#pragma unroll
        for (int j = 0; j < V / 4; ++j) {
            rg[i][j] = *reinterpret_cast<const uint4*>(gamma + oc + 4 * j);
            rb[i][j] = *reinterpret_cast<const uint4*>(beta + oc + 4 * j);
        }
    }
    float v[NC][V];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (MIX) {
            unpack8bf(make_uint2(rx2[i].x & lm[i], rx2[i].y & lm[i]), v[i]);
            // the bias of the Linear that produced x, kept out of its GEMM (see the backward)
            const unsigned k = m_xb & lm[i];
            v[i][0] += __uint_as_float(rxb[i].x & k); v[i][1] += __uint_as_float(rxb[i].y & k);
            v[i][2] += __uint_as_float(rxb[i].z & k); v[i][3] += __uint_as_float(rxb[i].w & k);
        } else unpack16<BF>(make_uint4(rx4[i].x & lm[i], rx4[i].y & lm[i], rx4[i].z & lm[i], rx4[i].w & lm[i]), v[i]);
        if (drop.thr) {   // dropout acts on x only, before the residual is added
#pragma unroll
            for (int j = 0; j < V; j += 2) {
                const unsigned bits = drop_bits(drop.base, (unsigned)row, (unsigned)(c * V + j) >> 1);
                v[i][j] *= drop_keep(bits, 0, drop.thr) ? drop.inv_keep : 0.f;
                v[i][j + 1] *= drop_keep(bits, 1, drop.thr) ? drop.inv_keep : 0.f;
            }
        }
        float r[V];
        const unsigned k = m_res & lm[i];
        unpack16<BF>(make_uint4(rres[i].x & k, rres[i].y & k, rres[i].z & k, rres[i].w & k), r);
#pragma unroll
        for (int j = 0; j < V; ++j) { v[i][j] += r[j]; s += v[i][j]; }      // (a chunk past the row's end holds zeros)
    }
    const float mu = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
        if (lane + 64 * i < nchunk) {
#pragma unroll
            for (int j = 0; j < V; ++j) { const float d = v[i][j] - mu; q += d * d; }
        }
    const float rs = rsqrtf(wave_sum(q) / (float)cols + eps);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            float o[V], gv[V], bv[V];
#pragma unroll
            for (int j = 0; j < V / 4; ++j) {
                gv[4 * j] = __uint_as_float(rg[i][j].x); gv[4 * j + 1] = __uint_as_float(rg[i][j].y);
                gv[4 * j + 2] = __uint_as_float(rg[i][j].z); gv[4 * j + 3] = __uint_as_float(rg[i][j].w);
                bv[4 * j] = __uint_as_float(rb[i][j].x); bv[4 * j + 1] = __uint_as_float(rb[i][j].y);
                bv[4 * j + 2] = __uint_as_float(rb[i][j].z); bv[4 * j + 3] = __uint_as_float(rb[i][j].w);
            }
#pragma unroll
            for (int j = 0; j < V; ++j) o[j] = (v[i][j] - mu) * rs * gv[j] + bv[j];
            *reinterpret_cast<uint4*>(yr + (size_t)c * 16) = pack16<BF>(o);
            // MIX: a bf16 copy of the fp32 stream for the next Linear (what autocast would cast per use)
            if (MIX && y16) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(y16) + (size_t)row * cols * 2 + (size_t)c * 8) = pack8bf(o);
        }
    }
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
}

// scalar fallback (any cols)
template <bool BF>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const void* x, const void* res, const float* gamma,
                                                         const float* beta, float eps, int64_t rows, int cols,
                                                         void* y, float* mean, float* rstd, Drop drop_in) {
    const Drop drop = resolve_drop(drop_in);
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t base = row * cols;
    float v[CPL];
    const bool cached = cols <= 64 * CPL;
    float s = 0.f;
    if (cached) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < cols ? ld<BF>(x, base + c) * drop_mult(drop, (unsigned)row, (unsigned)c) + (res ? ld<BF>(res, base + c) : 0.f) : 0.f;
            s += v[i];
        }
    } else {
        for (int c = lane; c < cols; c += 64) s += ld<BF>(x, base + c) * drop_mult(drop, (unsigned)row, (unsigned)c) + (res ? ld<BF>(res, base + c) : 0.f);
    }
    const float mu = wave_sum(s) / (float)cols;
    float q = 0.f;
    if (cached) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) { const int c = lane + 64 * i; const float d = c < cols ? v[i] - mu : 0.f; q += d * d; }
    } else {
        for (int c = lane; c < cols; c += 64) { const float d = ld<BF>(x, base + c) * drop_mult(drop, (unsigned)row, (unsigned)c) + (res ? ld<BF>(res, base + c) : 0.f) - mu; q += d * d; }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)cols + eps);
    if (cached) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int c = lane + 64 * i;
            if (c < cols) st<BF>(y, base + c, (v[i] - mu) * rs * gamma[c] + beta[c]);
        }
    } else {
        for (int c = lane; c < cols; c += 64) {
            const float z = ld<BF>(x, base + c) * drop_mult(drop, (unsigned)row, (unsigned)c) + (res ? ld<BF>(res, base + c) : 0.f);
            st<BF>(y, base + c, (z - mu) * rs * gamma[c] + beta[c]);
        }
    }
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
}

// ---- backward: dz, and per-block partial column sums of dgamma / dbeta ----
#ifndef TRX_LN_BWD_WAVES   // register budget of the vector kernel, in waves per SIMD (3: 168 registers; the widest rows -- bf16 beyond 1024
                          // columns, fp32 beyond 768 -- get 2: they would spill at 3)
#define TRX_LN_BWD_WAVES 3
#endif
constexpr int BWD_MAX_BLOCKS = 768;   // workgroups of 4 waves, rows dealt round-robin to the waves (cutting the registers to 128 for a fourth
                                      // workgroup per CU, 1024 blocks: measured, no gain -- 43.8 against 43.4 us mixed, tools/ln_bench.py)
template <bool BF>
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(const void* dy, const void* x, const void* res,
                                                         const float* gamma, const float* mean, const float* rstd,
                                                         int64_t rows, int cols, void* dz, void* dx, float* ws, int nblk, Drop drop_in) {
    const Drop drop = resolve_drop(drop_in);
    extern __shared__ float sm[];  // [4 waves][2][cols]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* pg = sm + (wave * 2) * cols;
    float* pb = pg + cols;
    for (int c = lane; c < cols; c += 64) { pg[c] = 0.f; pb[c] = 0.f; }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const int64_t base = row * cols;
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < cols; c += 64) {
            const float km = drop_mult(drop, (unsigned)row, (unsigned)c);
            const float z = ld<BF>(x, base + c) * km + (res ? ld<BF>(res, base + c) : 0.f);
            const float xh = (z - mu) * rs, g = ld<BF>(dy, base + c) * gamma[c];
            s1 += g; s2 += g * xh;
        }
        s1 = wave_sum(s1) / (float)cols; s2 = wave_sum(s2) / (float)cols;
        for (int c = lane; c < cols; c += 64) {
            const float km = drop_mult(drop, (unsigned)row, (unsigned)c);
            const float z = ld<BF>(x, base + c) * km + (res ? ld<BF>(res, base + c) : 0.f);
            const float xh = (z - mu) * rs, d = ld<BF>(dy, base + c);
            const float gz = rs * (d * gamma[c] - s1 - xh * s2);
            st<BF>(dz, base + c, gz);
            if (dx) st<BF>(dx, base + c, gz * km);   // gradient of x through its dropout
            pg[c] += d * xh; pb[c] += d;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += 256) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < 4; ++w) { a += sm[(w * 2) * cols + c]; b += sm[(w * 2 + 1) * cols + c]; }
        ws[(int64_t)blockIdx.x * cols + c] = a;
        ws[((int64_t)nblk + blockIdx.x) * cols + c] = b;
    }
}
// Fast path (same conditions as the forward's): a wave per row, the row lives in registers (16-byte
// accesses, one pass over dy / x / res), and because a lane owns the same columns in every row it also
// carries the dgamma / dbeta partial sums of its columns in registers over all its rows; the four
// waves of a workgroup are combined through LDS at the end.
template <bool BF, bool MIX, int NC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(((BF && NC >= 3) || NC >= 4) ? 2 : TRX_LN_BWD_WAVES))) void add_ln_bwd_vec_kernel(const void* dy, const void* x, const void* res,
                                                             const float* gamma, const float* mean, const float* rstd,
                                                             int64_t rows, int cols, void* dz, void* dx, float* ws, int nblk, Drop drop_in,
                                                             const void* dy16 = nullptr, const float* xbias = nullptr) {
    const Drop drop = resolve_drop(drop_in);
    extern __shared__ float sm[];  // [4 waves][2][cols]
    constexpr int V = Vec16<BF>::N;
    constexpr int ES = BF ? 2 : 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = cols / V;
    float ag[NC][V], ab[NC][V], gm[NC][V];
    float ax[MIX ? NC : 1][MIX ? V : 1];   // MIX + xbias: column sums of dx = the gradient of that bias
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            ag[i][j] = 0.f; ab[i][j] = 0.f;
            if (MIX) ax[i][j] = 0.f;
        }
        // gamma of this lane's columns: whole 16-byte loads of the clamped chunk, zeroed past the row's end (no branch per element)
        const unsigned keep = c < nchunk ? ~0u : 0u;
        const float* gp = gamma + (size_t)min(c, nchunk - 1) * V;
#pragma unroll
        for (int j = 0; j < V; j += 4) {
            const uint4 u = *reinterpret_cast<const uint4*>(gp + j);
            gm[i][j] = __uint_as_float(u.x & keep); gm[i][j + 1] = __uint_as_float(u.y & keep);
            gm[i][j + 2] = __uint_as_float(u.z & keep); gm[i][j + 3] = __uint_as_float(u.w & keep);
        }
    }
    // One row per wave and trip, in two phases: every load of the row is issued first, into raw registers, without a branch
    // between them -- an operand the call does not have (dy or its bf16 twin, res) is read from gamma with row stride 0 and
    // ANDed away, a chunk past the row's end re-reads the last one and is ANDed away too -- then the arithmetic.  (Round 5: the
    // kernel used to test each pointer where it was used; every test was a basic block, every block ended in s_waitcnt
    // vmcnt(0), and a row cost nine memory latencies one after the other instead of one: mixed 16384 x 768 38.5 -> 35.9 us,
    // bf16 33.6 -> 28.6, profiles/r05_ln_ab.json.  Requesting the NEXT row as well before this row's arithmetic costs a
    // wave per SIMD in registers and was slower: 38.4 us.)
    const bool has_dy = !MIX || dy != nullptr, has_dy16 = MIX && dy16 != nullptr, has_res = res != nullptr;
    const char* const p_dy = has_dy ? reinterpret_cast<const char*>(dy) : reinterpret_cast<const char*>(gamma);
    const char* const p_dy16 = has_dy16 ? reinterpret_cast<const char*>(dy16) : reinterpret_cast<const char*>(gamma);
    const char* const p_res = has_res ? reinterpret_cast<const char*>(res) : reinterpret_cast<const char*>(gamma);
    const size_t rs_full = (size_t)cols * ES, rs_half = rs_full >> 1;
    const size_t st_dy = has_dy ? rs_full : 0, st_dy16 = has_dy16 ? rs_half : 0, st_res = has_res ? rs_full : 0;
    const unsigned m_dy = has_dy ? ~0u : 0u, m_dy16 = has_dy16 ? ~0u : 0u, m_res = has_res ? ~0u : 0u;   // wave-uniform
    unsigned lm[NC]; unsigned o16[NC];          // lane mask of chunk i (all ones: inside the row), its clamped byte offset
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        lm[i] = c < nchunk ? ~0u : 0u;
        o16[i] = (unsigned)min(c, nchunk - 1) * 16u;
    }
    float xb[MIX ? NC : 1][MIX ? V : 1];        // the bias of x, once per kernel (zeros without one)
    if (MIX) {
        const unsigned m_xb = xbias ? ~0u : 0u;
        const char* const p_xb = reinterpret_cast<const char*>(xbias ? xbias : gamma);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const uint4 u = *reinterpret_cast<const uint4*>(p_xb + o16[i]);
            const unsigned keep = m_xb & lm[i];
            xb[i][0] = __uint_as_float(u.x & keep); xb[i][1] = __uint_as_float(u.y & keep);
            xb[i][2] = __uint_as_float(u.z & keep); xb[i][3] = __uint_as_float(u.w & keep);
        }
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const size_t rb = (size_t)row * cols * ES;
        uint4 rdy[NC], rres[NC], rx4[MIX ? 1 : NC];
        uint2 rdy16[MIX ? NC : 1], rx2[MIX ? NC : 1];
        const float mu = mean[row], rs = rstd[row];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            rdy[i] = *reinterpret_cast<const uint4*>(p_dy + (size_t)row * st_dy + o16[i]);
            if (MIX) {
                rdy16[i] = *reinterpret_cast<const uint2*>(p_dy16 + (size_t)row * st_dy16 + (o16[i] >> 1));
                rx2[i] = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(x) + (size_t)row * rs_half + (o16[i] >> 1));
            } else rx4[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(x) + (size_t)row * rs_full + o16[i]);
            rres[i] = *reinterpret_cast<const uint4*>(p_res + (size_t)row * st_res + o16[i]);
        }
        float g[NC][V], xh[NC][V];
        float dd[NC][V], zz[NC][V], rr[NC][V];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const unsigned k_dy = m_dy & lm[i], k_dy16 = m_dy16 & lm[i], k_res = m_res & lm[i];
            unpack16<BF>(make_uint4(rdy[i].x & k_dy, rdy[i].y & k_dy, rdy[i].z & k_dy, rdy[i].w & k_dy), dd[i]);
            if (MIX) {   // + the gradient that arrived through the bf16 copy of y
                float e[4];
                unpack8bf(make_uint2(rdy16[i].x & k_dy16, rdy16[i].y & k_dy16), e);
#pragma unroll
                for (int j = 0; j < 4; ++j) dd[i][j] += e[j];
                unpack8bf(make_uint2(rx2[i].x & lm[i], rx2[i].y & lm[i]), zz[i]);
            } else unpack16<BF>(make_uint4(rx4[i].x & lm[i], rx4[i].y & lm[i], rx4[i].z & lm[i], rx4[i].w & lm[i]), zz[i]);
            unpack16<BF>(make_uint4(rres[i].x & k_res, rres[i].y & k_res, rres[i].z & k_res, rres[i].w & k_res), rr[i]);
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            float* d = dd[i]; float* z = zz[i];
            if (MIX) {
#pragma unroll
                for (int j = 0; j < V; ++j) z[j] += xb[i][j];
            }
            if (drop.thr) {
#pragma unroll
                for (int j = 0; j < V; j += 2) {
                    const unsigned bits = drop_bits(drop.base, (unsigned)row, (unsigned)(c * V + j) >> 1);
                    z[j] *= drop_keep(bits, 0, drop.thr) ? drop.inv_keep : 0.f;
                    z[j + 1] *= drop_keep(bits, 1, drop.thr) ? drop.inv_keep : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < V; ++j) z[j] += rr[i][j];
            // (a chunk past the row's end: d = 0 and gm = 0, so it adds nothing to s1, s2, ag, ab; its xh is never stored)
#pragma unroll
            for (int j = 0; j < V; ++j) {
                xh[i][j] = (z[j] - mu) * rs;
                g[i][j] = d[j] * gm[i][j];
                s1 += g[i][j]; s2 += g[i][j] * xh[i][j];
                ag[i][j] += d[j] * xh[i][j]; ab[i][j] += d[j];
            }
        }
        s1 = wave_sum(s1) / (float)cols; s2 = wave_sum(s2) / (float)cols;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                float o[V];
#pragma unroll
                for (int j = 0; j < V; ++j) o[j] = rs * (g[i][j] - s1 - xh[i][j] * s2);
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(dz) + rb + (size_t)c * 16) = pack16<BF>(o);
                if (dx) {   // gradient of x: through its dropout, and in x's own storage type
                    if (drop.thr) {
#pragma unroll
                        for (int j = 0; j < V; j += 2) {
                            const unsigned bits = drop_bits(drop.base, (unsigned)row, (unsigned)(c * V + j) >> 1);
                            o[j] *= drop_keep(bits, 0, drop.thr) ? drop.inv_keep : 0.f;
                            o[j + 1] *= drop_keep(bits, 1, drop.thr) ? drop.inv_keep : 0.f;
                        }
                    }
                    if (MIX) {
                        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(dx) + (rb >> 1) + (size_t)c * 8) = pack8bf(o);
#pragma unroll
                        for (int j = 0; j < V; ++j) ax[i][j] += o[j];
                    } else *reinterpret_cast<uint4*>(reinterpret_cast<char*>(dx) + rb + (size_t)c * 16) = pack16<BF>(o);
                }
            }
        }
    }
    float* pg = sm + (wave * 2) * cols;
    float* pb = pg + cols;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
#pragma unroll
            for (int j = 0; j < V; ++j) { pg[c * V + j] = ag[i][j]; pb[c * V + j] = ab[i][j]; }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += 256) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < 4; ++w) { a += sm[(w * 2) * cols + c]; b += sm[(w * 2 + 1) * cols + c]; }
        ws[(int64_t)blockIdx.x * cols + c] = a;
        ws[((int64_t)nblk + blockIdx.x) * cols + c] = b;
    }
    if (MIX && xbias) {   // third partial row: the bias gradient, through the same LDS area
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
#pragma unroll
                for (int j = 0; j < V; ++j) pg[c * V + j] = ax[i][j];
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < cols; c += 256) {
            float a = 0.f;
            for (int w = 0; w < 4; ++w) a += sm[(w * 2) * cols + c];
            ws[((int64_t)2 * nblk + blockIdx.x) * cols + c] = a;
        }
    }
}

// second stage: sum the nblk partial rows of ws.  A workgroup owns 16 columns of one of the two
// arrays (blockIdx.y: 0 dgamma, 1 dbeta); its 16 x 16 threads walk the partial rows 16 at a time.
__global__ __launch_bounds__(256) void add_ln_bwd_reduce_kernel(const float* ws, int nblk, int cols, float* dgamma, float* dbeta,
                                                                float* dxbias = nullptr) {
    __shared__ float part[16][17];
    const int tc = threadIdx.x & 15, tr = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tc;
    const float* src = ws + (int64_t)blockIdx.y * nblk * cols;
    float a = 0.f;
    if (c < cols) {
        // 8 independent loads in flight per thread: a plain loop waited for each partial row before asking for the next
        float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int i = tr;
        for (; i + 7 * 16 < nblk; i += 8 * 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a8[u] += src[(int64_t)(i + 16 * u) * cols + c];
        }
        for (; i < nblk; i += 16) a8[0] += src[(int64_t)i * cols + c];
        a = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    }
    part[tr][tc] = a;
    __syncthreads();
    if (tr == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += part[i][tc];
        (blockIdx.y == 0 ? dgamma : (blockIdx.y == 1 ? dbeta : dxbias))[c] = t;
    }
}

// the second stage for MANY backward calls at once (round 5: trx_add_layernorm_bwd_reduce_many): the table travels by value
// (a kernel argument of 48 x 40 bytes -- no device copy, nothing a stream capture has to keep alive); blockIdx.z picks the
// call, the rest is add_ln_bwd_reduce_kernel's walk, so the sums are bit for bit those of the per-call second stage
struct LnReduceTable { trx_ln_reduce_item it[TRX_LN_REDUCE_MAX]; };
__global__ __launch_bounds__(256) void add_ln_bwd_reduce_many_kernel(LnReduceTable t, int cols) {
    __shared__ float part[16][17];
    const trx_ln_reduce_item& e = t.it[blockIdx.z];
    float* const dst = blockIdx.y == 0 ? e.dgamma : (blockIdx.y == 1 ? e.dbeta : e.dxbias);
    if (!dst) return;                                  // (block-uniform: a call without a bias gradient)
    const int nblk = e.nblk;
    const int tc = threadIdx.x & 15, tr = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tc;
    const float* src = e.ws + (int64_t)blockIdx.y * nblk * cols;
    float a = 0.f;
    if (c < cols) {
        float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int i = tr;
        for (; i + 7 * 16 < nblk; i += 8 * 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a8[u] += src[(int64_t)(i + 16 * u) * cols + c];
        }
        for (; i < nblk; i += 16) a8[0] += src[(int64_t)i * cols + c];
        a = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    }
    part[tr][tc] = a;
    __syncthreads();
    if (tr == 0 && c < cols) {
        float s_ = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s_ += part[i][tc];
        dst[c] = s_;
    }
}

#ifdef TRX_NN_LAB      // the vector-ALU kernels of round 1: lab builds only (tools/experiments/attn_valu.h)
#define TRX_VALU_PART 1
#include "../../tools/experiments/attn_valu.h"
#undef TRX_VALU_PART
#endif

// ---- attention forward on the matrix cores (bf16 storage, fp32 accumulate) -----------------------
// Workgroup = 4 waves = 128 queries of one (batch, head); a wave owns 32 queries and walks the keys in
// tiles of 32.  The score tile is computed TRANSPOSED, S^T = K Q^T with v_mfma_f32_32x32x16_bf16
// (A = K rows from LDS, B = Q fragments held in registers), so a lane owns ONE query (col = lane & 31)
// and its 16 accumulator registers are 16 of the tile's 32 keys: the softmax row statistics are
// lane-local plus one exchange with lane ^ 32.  Key tiles of 64, next tile's global loads in flight
// under the current tile's math, exponentials as exp2 with the scale folded in.  The probabilities then feed the second product
// without leaving registers: O^T += V^T P^T, where registers 8s..8s+7 of the S^T accumulator,
// converted to bf16, ARE the B fragment of k-step s (cdna guide section 3, "an accumulator tile as
// the next MFMA's operand"), and the matching V^T fragments -- keys 16s + 8(j>>2) + 4h + (j&3) for
// element j of lane half h -- are read column-major from the row-major V tile with
// ds_read_b64_tr_b16.  FLOPs = 4 B H Lq Lk 64; K/V tiles are staged through LDS by all 256 threads.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// A lane's row of 64 values in the transposed accumulator layout of the attention kernels (r0: d 0..31, r1: d 32..63; register
// 4 gq + j of a half is d 8 gq + 4 hh + j, hh = lane >> 5) -> bf16 in memory, times `mul`.
// A 16-byte piece of the row (d 8 gq .. 8 gq + 7) is half in this lane and half in lane ^ 32.  wide (wave-uniform; every lane of
// the wave must get here): the two lanes trade -- the lower one gives its odd pieces and takes the partner's even ones;
// v_permlane32_swap X, Y leaves [own even | partner's even] in the lower lanes and [partner's odd | own odd] in the upper ones --
// and the row leaves in 4 dwordx4 stores per lane, 32 contiguous bytes of a row per instruction, instead of 8 dwordx2 with 16.
// Round 5, same box, interleaved (tools/attn_shapes_ab.py): forward 512 x 512 41.0 -> 38.4 us, 160 x 512 20.4 -> 19.6; blocks
// with one or two 32-row units (7 queries, the tail of 160) are 1-3 % faster narrow -- the swap sits on their short path --
// so the callers pass wide = "this workgroup's block is full".
__device__ __forceinline__ void store_row_bf16(bf16_t* rowp, bool live, bool wide, int hh, const f32x16& r0, const f32x16& r1, float mul) {
    if (wide) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int ge = 8 * pr, go = 8 * pr + 4;
                unsigned xx, xy, yx, yy;
                if (blk == 0) {
                    xx = pack2bf(r0[ge] * mul, r0[ge + 1] * mul); xy = pack2bf(r0[ge + 2] * mul, r0[ge + 3] * mul);
                    yx = pack2bf(r0[go] * mul, r0[go + 1] * mul); yy = pack2bf(r0[go + 2] * mul, r0[go + 3] * mul);
                } else {
                    xx = pack2bf(r1[ge] * mul, r1[ge + 1] * mul); xy = pack2bf(r1[ge + 2] * mul, r1[ge + 3] * mul);
                    yx = pack2bf(r1[go] * mul, r1[go + 1] * mul); yy = pack2bf(r1[go + 2] * mul, r1[go + 3] * mul);
                }
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(xx), "+v"(yx));
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(xy), "+v"(yy));
                if (live) *reinterpret_cast<uint4*>(rowp + 32 * blk + 8 * (2 * pr + hh)) = uint4{xx, xy, yx, yy};
            }
    } else if (live) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            uint2 w0, w1;
            w0.x = pack2bf(r0[4 * gq] * mul, r0[4 * gq + 1] * mul); w0.y = pack2bf(r0[4 * gq + 2] * mul, r0[4 * gq + 3] * mul);
            w1.x = pack2bf(r1[4 * gq] * mul, r1[4 * gq + 1] * mul); w1.y = pack2bf(r1[4 * gq + 2] * mul, r1[4 * gq + 3] * mul);
            *reinterpret_cast<uint2*>(rowp + 8 * gq + 4 * hh) = w0;
            *reinterpret_cast<uint2*>(rowp + 32 + 8 * gq + 4 * hh) = w1;
        }
    }
}


// one 64-key tile of the online softmax, in the transposed accumulator layout: register t of half
// hb holds key row hb*32 + (t&3) + 8*(t>>2) + 4*hh of the tile for this lane's query.  The additive
// mask is already inside the scores (the accumulators start from mask / scale); scores move to
// base 2 here (scale * log2 e in one multiply).  VIS: the tile has keys that are out of range or
// hidden by causality for some lane of the wave (wave-uniform).
// DROP: after the row sum took the probabilities, the dropped ones are zeroed for the P V product
// (xd = hash input of this lane's row at the tile's first key pair, see drop_bits).
#ifndef TRX_ATT_PK_FMA
#define TRX_ATT_PK_FMA 1
#endif
#ifndef TRX_ATT_ROWSUM_DOT2
#define TRX_ATT_ROWSUM_DOT2 1
#endif
#ifndef TRX_ATT_ABL      // timing-only ablations of the forward tile (tools/attn_ablate.sh; results are WRONG with any bit set): 1 no exponentials,
#define TRX_ATT_ABL 0    // 2 no row maximum, 4 no barrier, 8 no K fragment reads, 16 no second product, 32 no first product, 64 no V reads, 128 no K / V staging (LDS-DMA)
#endif
template <bool VIS, bool DROP>
__device__ __forceinline__ void attn_softmax_tile(f32x16& s0, f32x16& s1, f32x16& o0, f32x16& o1, float& m, float& lsum,
                                                  float sl2, int key0, int hh, int klim, unsigned xd, unsigned thr,
                                                  unsigned (&pk)[2][8], int other_half = -1) {
    // other_half: (lane ^ 32) * 4, the ds_bpermute address of the lane that holds this query's other 32 keys, computed ONCE by the
    // caller (__shfl_xor recomputes it with seven vector instructions at every call: one per key tile, on the critical path)
    // the maximum is taken on the raw scores (scale > 0 commutes with it); the scaling then rides in the
    // exponent's fma: p = exp2(s * scale log2e - m)
    float mb = -__builtin_inff();
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int kr_ = hb * 32 + (t & 3) + 8 * (t >> 2) + 4 * hh;
            float val = hb ? s1[t] : s0[t];
            if (VIS) {
                val = (key0 + kr_ > klim) ? -__builtin_inff() : val;
                if (hb) s1[t] = val; else s0[t] = val;
            }
            if (!(TRX_ATT_ABL & 2)) mb = fmaxf(mb, val);
        }
    if (TRX_ATT_ABL & 2) mb = 0.f;
    {   // this query's other 32 keys sit in lane ^ 32.  v_permlane32_swap vdst, src swaps lanes 32-63 of vdst with lanes 0-31 of src:
        // with the maximum in both, vdst = [lo | lo] and src = [hi | hi], and their maximum is the row's in every lane -- one
        // instruction instead of the ds_bpermute round trip of __shfl_xor (~100 cycles on the tile's critical path: -3 %).  asm
        // with its own two wait states: this hipcc's builtin returns vdst for both results and does not pad the VALU -> permlane
        // hazard (tools/experiments/permlane_swap_check.hip).
        float ma = mb, mc = mb;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mc));
        mb = fmaxf(ma, mc) * sl2;
    }
    (void)other_half;
    // The reference the exponents are taken against only moves when the row's maximum has outgrown it by more than 2^TRX_ATT_LAZY
    // (8: probabilities up to 256 before normalisation -- bf16 and the fp32 sums do not care, they would with fp16): on most
    // tiles no lane moves it, alpha is 1 everywhere and the 16 v_pk_mul of the output rescale are skipped.  Any reference that
    // is the same for all keys of a row is exact; the log-sum-exp written at the end is m + log2(sum) whatever m is.
#ifndef TRX_ATT_LAZY
#define TRX_ATT_LAZY 8
#endif
    const float mt_ = fmaxf(m, mb);
    const float mn = (TRX_ATT_LAZY > 0 && !(mt_ > m + (float)TRX_ATT_LAZY)) ? m : mt_;      // m = -inf: any finite maximum moves it
    const float mref = (mn == -__builtin_inff()) ? 0.f : mn;   // all hidden so far: exp2(-inf - 0) = 0, no NaN
    const float alpha = __builtin_amdgcn_exp2f(m - mref);
    const float nref = -mref;
    // the exponent's argument: TRX_ATT_PK_FMA=1 two registers at a time (v_pk_fma_f32: one issue slot for two fmas), =0 one
    // v_fma_f32 each through asm (plain C++ is SLP-packed by hipcc into the same v_pk_fma).  The microarchitecture guide
    // prices a packed f32 instruction beside MFMAs at +22 cycles over two scalar ones; profiles/r03_attention_ab.json
    // holds what this kernel measured for both forms.
#if TRX_ATT_PK_FMA
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    const f32x2 sl2v = {sl2, sl2}, nrefv = {nref, nref};
#pragma unroll
    for (int t = 0; t < 16; t += 2) {
        const f32x2 a = __builtin_elementwise_fma((f32x2){s0[t], s0[t + 1]}, sl2v, nrefv);
        if (TRX_ATT_ABL & 1) { s0[t] = a.x; s0[t + 1] = a.y; continue; }
        s0[t] = __builtin_amdgcn_exp2f(a.x); s0[t + 1] = __builtin_amdgcn_exp2f(a.y);
    }
#pragma unroll
    for (int t = 0; t < 16; t += 2) {
        const f32x2 a = __builtin_elementwise_fma((f32x2){s1[t], s1[t + 1]}, sl2v, nrefv);
        if (TRX_ATT_ABL & 1) { s1[t] = a.x; s1[t + 1] = a.y; continue; }
        s1[t] = __builtin_amdgcn_exp2f(a.x); s1[t + 1] = __builtin_amdgcn_exp2f(a.y);
    }
#else
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        float a;
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(s0[t]), "v"(sl2), "v"(nref));
        s0[t] = __builtin_amdgcn_exp2f(a);
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        float a;
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(s1[t]), "v"(sl2), "v"(nref));
        s1[t] = __builtin_amdgcn_exp2f(a);
    }
#endif
    float ps = 0.f;
    if (DROP) {
        // the row sum takes every probability; the dropped ones are then zeroed for the P V product
#pragma unroll
        for (int t = 0; t < 16; ++t) ps += s0[t] + s1[t];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int t = 0; t < 16; t += 2) {   // registers t, t+1 = keys 2c, 2c+1: one hash
                const unsigned bits = lowbias32(xd + (unsigned)(hb * 16 + ((t & 3) >> 1) + 4 * (t >> 2)) * DROP_C2);
                const float e0 = hb ? s1[t] : s0[t], e1 = hb ? s1[t + 1] : s0[t + 1];
                pk[hb][t >> 1] = pack2bf(drop_keep(bits, 0, thr) ? e0 : 0.f, drop_keep(bits, 1, thr) ? e1 : 0.f);
            }
    } else {
#if TRX_ATT_ROWSUM_DOT2
        // pack to bf16 for the second product and sum THOSE values (v_dot2c_f32_bf16 with (1, 1)): one
        // instruction per pair instead of two adds, and the normaliser matches what multiplies V
        const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                const unsigned w = pack2bf(hb ? s1[t] : s0[t], hb ? s1[t + 1] : s0[t + 1]);
                pk[hb][t >> 1] = w;
                ps = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), ones, ps, false);
            }
#else
        // row sum as plain f32 adds of the probabilities (four chains, asm so that hipcc does not pack them), the bf16
        // pairs for the second product beside them (the guide's rule: v_dot2c costs ~10 cycles as a filler)
        float p4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                const float e0 = hb ? s1[t] : s0[t], e1 = hb ? s1[t + 1] : s0[t + 1];
                pk[hb][t >> 1] = pack2bf(e0, e1);
                asm("v_add_f32 %0, %0, %1" : "+v"(p4[(t >> 1) & 1]) : "v"(e0));
                asm("v_add_f32 %0, %0, %1" : "+v"(p4[2 + ((t >> 1) & 1)]) : "v"(e1));
            }
        ps = (p4[0] + p4[1]) + (p4[2] + p4[3]);
#endif
    }
    lsum = lsum * alpha + ps;
    m = mn;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {   // wave-uniform: the running maximum moved for some lane
#pragma unroll
        for (int t = 0; t < 16; ++t) { o0[t] *= alpha; o1[t] *= alpha; }
    }
}

#ifndef TRX_ATT_INTERLEAVE      // 1: the second product's MFMAs between the groups of the exponential chain (round 6)
#define TRX_ATT_INTERLEAVE 1
#endif
#ifndef TRX_ATT_AUG      // 1: scale and reference through the first product (tools/experiments/attn_aug.h; measured, no faster: lab variant)
#define TRX_ATT_AUG 0
#endif
#if TRX_ATT_AUG
#include "../../tools/experiments/attn_aug.h"
#endif

#ifdef TRX_ATT_STAMP   // tools/attn_lab.hip: per workgroup [0] realtime in, [1] realtime out, [2] cycles in, [3] after the prologue, [4..] after tile j
__device__ unsigned long long* g_att_stamp;
#define TRX_STAMP(I, V) if (g_att_stamp && threadIdx.x == 0) g_att_stamp[(size_t)blockIdx.x * 32 + (I)] = (V)
#else
#define TRX_STAMP(I, V)
#endif
// (An "extra waves" form -- for 128 < Lq <= 256 ONE workgroup of ceil(Lq / 32) waves per (batch, head) instead of two 128-query
// workgroups of which the second is mostly empty -- saved 37.5 % of the MFMAs and was SLOWER, profiles/r03_attention_ab.json:
// cross-attention 160 x 512 28.8 against 26.0 us; these launches are bound by the latency of their key-tile chain and half as
// many workgroups hide less of it.  Removed in round 5.)
// Round 5: a tile with hidden keys (out of range, or later than a lane's query under the causal mask) sets them to minus infinity
// IN PLACE and then takes the same form of attn_softmax_tile as every other tile.  With the hidden-key form instantiated
// beside the plain one the kernel needed ~40 more registers (dropout: 187, two waves per SIMD); now 151 and three waves for every
// variant but the full mask: 512 x 512 with dropout 0.1 62.7 -> 56 us (tools/r05/persist_time.py).
template <int MM, bool DROP>   // mask mode, dropout: one kernel per case keeps each one's register footprint to what it needs
#ifndef TRX_ATT_WAVES      // waves per SIMD the register budget is cut for (A/B knob; 3 = 168 registers)
#define TRX_ATT_WAVES 3
#endif
#ifndef TRX_ATT_WIDE_STORE // 1: the output row leaves in 16-byte pieces (lanes r and r ^ 32 trade halves by v_permlane32_swap); A/B in round 5
#define TRX_ATT_WIDE_STORE 1
#endif
#ifndef TRX_ATT_PRIO       // 1: s_setprio 1 around the MFMA clusters: -2.8 % at 512 x 512, -3.7 % at 160 x 512 (profiles/r03_attention_ab.json)
#define TRX_ATT_PRIO 1
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MM == TRX_NN_MASK_FULL ? 2 : TRX_ATT_WAVES, MM == TRX_NN_MASK_FULL ? 2 : TRX_ATT_WAVES))) void attention_fwd_mfma_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                 const bf16_t* __restrict__ v, const float* __restrict__ mask,
                                                                 int causal, int B, int H, int Lq, int Lk,
                                                                 float scale, bf16_t* __restrict__ out, float* __restrict__ lse, DropArgs da) {
    // key tiles of 64 in a ring of 3 (prefetch distance 2), filled by LDS-DMA: [buf][K 8 KiB | V 8 KiB];
    // then the key mask of 1024 keys (16 tiles), pre-divided by the scale.  52 KiB: 3 workgroups / CU
    // (its own array: the compiler then knows mask reads cannot alias the LDS-DMA writes and does not
    // drain vmcnt before them)
    __shared__ __attribute__((aligned(128))) char lds[3 * 16384];
    __shared__ __attribute__((aligned(16))) float ldsM[1024];
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void gbl_void;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    TRX_STAMP(0, __builtin_amdgcn_s_memrealtime()); TRX_STAMP(2, __builtin_amdgcn_s_memtime());
    const int r = lane & 31, hh = lane >> 5;
    const int nqb = (Lq + 127) / 128;
    // workgroups are dealt round-robin to the 8 XCDs; renumber so that the query blocks of one
    // (batch, head) -- which re-read the same K/V -- are neighbours on ONE XCD and share its L2
    int bid = blockIdx.x;
    int qsw = 0;
    {
        const int nwg = gridDim.x, per = nwg >> 3, main_ = per << 3;
        // Two query blocks per (batch, head), the second one short (Lq = 160): its workgroup does a quarter of the first one's
        // arithmetic.  An XCD hands its workgroups to its 32 CUs in turn, so numbered (full, short, full, short, ...) the three
        // workgroups a CU holds -- j, j + 32, j + 64 of the XCD's list -- would be of ONE kind, half the CUs loaded four times
        // as much as the others; the kinds swap every 32 places instead, and every CU gets a mix.
        if (nqb == 2 && !(per & 1) && bid < main_) qsw = (bid >> 3) >> 5;
        if (bid < main_) bid = (bid & 7) * per + (bid >> 3);
    }
    const int qb = (bid + qsw) % nqb, h = (bid / nqb) % H, b = bid / (nqb * H);
    // A query block with one or two 32-query units (the last block of Lq = 160: 128 + 32; every block of a decoder with 7
    // positions) would leave three or two of its four waves computing on padding.  Those waves take KEYS instead: the block's
    // key tiles are dealt round-robin over KS = 4 (2) waves per query unit -- every wave still stages every tile and meets
    // every barrier, it only skips the arithmetic of the tiles that are not its own -- and the partial (O, m, l) of a unit's
    // waves are merged through LDS at the end (a log-sum-exp merge: any reference common to a row is exact).  The key-tile
    // chain of such a block is a quarter (half) as long; the launch does the arithmetic of 5 units instead of 8 at Lq = 160.
    const int nuq = min(4, (Lq - qb * 128 + 31) / 32);
    int KS = nuq == 1 ? 4 : (nuq == 2 ? 2 : 1);               // block-uniform
    {   // ... when the block has key tiles to deal out: with fewer than two per wave (the 160 x 160 causal decoder block: three
        // tiles; 7 x 7: one) the merge costs more than the shorter chain saves (measured: 10.7 -> 11.1 us at causal 160 x 160)
        int nkb_ = (Lk + 63) / 64;
        if (causal) nkb_ = min(nkb_, (min(Lq - 1, qb * 128 + 127) + Lk - Lq) / 64 + 1);
        if (nkb_ < 2 * KS) KS = 1;
    }
    const int uq = KS == 4 ? 0 : (KS == 2 ? (wave & 1) : wave);
    const int kp = KS == 4 ? wave : (KS == 2 ? (wave >> 1) : 0);
    const int qidx = qb * 128 + uq * 32 + r;                // this lane's query
    const int qc = qidx < Lq ? qidx : Lq - 1;
    // The prologue's vector loads, oldest first: the key mask of the first 1,024 keys (4 floats per thread), Q, then the
    // LDS-DMA of tiles 0 and 1 -- ALL through asm, so that one s_waitcnt vmcnt(4) before the first barrier retires the mask, Q
    // and tile 0 together while tile 1 stays in flight.  (Until round 5 Q was a C++ load and the mask was loaded after the
    // staging: hipcc, which does not count the asm DMAs, retired Q with vmcnt(0) -- both tiles -- and the mask then cost a second
    // memory latency in front of the first tile: ~1 us of a 10-20 us decoder launch.)
    constexpr bool keymask = MM == TRX_NN_MASK_KEY;
    const float* mkey = keymask ? mask + (int64_t)b * Lk : nullptr;
    float mv0[4] = {0.f, 0.f, 0.f, 0.f};
    if (keymask) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* mp = mkey + min(4 * tid + i, Lk - 1);
            asm volatile("global_load_dword %0, %1, off" : "=&v"(mv0[i]) : "v"(mp) : "memory");
        }
    }
    bf16x8 qf[4];   // B operand of S^T = K Q^T: B[k = 8hh + j][col r] = Q[query r][d = 16 s + 8 hh + j]
    {
        const bf16_t* qp = q + ((int64_t)b * Lq + qc) * (da.ldq ? da.ldq : H * 64) + h * 64 + 8 * hh;
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:32\n\t"
                     "global_load_dwordx4 %2, %4, off offset:64\n\tglobal_load_dwordx4 %3, %4, off offset:96"
                     : "=&v"(qf[0]), "=&v"(qf[1]), "=&v"(qf[2]), "=&v"(qf[3]) : "v"(qp) : "memory");
    }
    f32x16 o0, o1;
#pragma unroll
    for (int t = 0; t < 16; ++t) { o0[t] = 0.f; o1[t] = 0.f; }
    const float sl2 = scale * 1.44269504088896340736f;
    // (TRX_ATT_AUG: Q is multiplied by sl2 below, so "raw-score units" are the exponent's units and inv_scale is log2 e)
    // additive mask enters as the accumulators' starting value, in raw-score units (mask / scale); the
    // clamp keeps finfo.min-style masks finite, so fully masked rows come out uniform like torch's
    const float inv_scale = TRX_ATT_AUG ? 1.44269504088896340736f : 1.0f / scale;
    // The floor of a masked score, in raw-score units: 2^28 in the exponent's units (score * scale * log2 e).  Large enough that
    // a masked key weighs exp2(-2^28) = 0 beside any other and that the q.k term is absorbed (a fully masked row comes out
    // uniform, as torch's finfo.min does), small enough that the exponent's argument fma(s, scale log2e, -reference) of a tile
    // whose keys are ALL masked stays within the rounding of 2^28 (+-16: probabilities up to 2^16, finite).  With the -1e30 of
    // rounds 1-3 that residual was ~1e21 and such a tile -- the first tile of a left-padded row; since round 4 also any tile
    // that opens a key-split wave -- produced inf and NaN.
    const float mask_floor = TRX_ATT_AUG ? -268435456.0f : -268435456.0f / sl2;
#define TRX_MASK_INIT(X) fmaxf((X) * inv_scale, mask_floor)
    float m = -__builtin_inff(), lsum = 0.f;
    const int off = Lk - Lq;
    int nkb = (Lk + 63) / 64;
    if (causal) {  // last key any query of this workgroup can see
        const int lastq = min(Lq - 1, qb * 128 + 127);
        nkb = min(nkb, (lastq + off) / 64 + 1);
    }
    // last visible key of this lane's query (also bounds the tail tile)
    const int klim = causal ? min(Lk - 1, qidx + off) : Lk - 1;
    const int klim_wave_min = causal ? min(Lk - 1, qb * 128 + uq * 32 + off) : Lk - 1;   // smallest klim in the wave
    const float* mrow = (MM == TRX_NN_MASK_FULL) ? mask + ((int64_t)b * Lq + qc) * Lk : nullptr;

    // LDS-DMA geometry: a piece is one global_load_lds_dwordx4 = 8 rows x 128 B written lane-linear
    // (row 8p + lane/8, slot lane%8); wave w moves pieces 2w, 2w+1 of K and of V.  K's bank swizzle
    // (slot = chunk ^ ((row >> 1) & 7)) is applied to the SOURCE chunk; V is stored straight.
    // Addresses are (uniform base + tile offset) + a per-lane 32-bit byte offset fixed for the kernel.
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
    // one piece: 32-bit per-lane offset against a wave-uniform 64-bit base in scalar registers (no 64-bit vector add per piece:
    // the builtin, given a pointer, spent ~20 v_lshl_add_u64 / v_readfirstlane a tile on addresses), M0 written in the same statement
    // (the base goes through readfirstlane: a wave-uniform 64-bit sum that instruction selection happened to put on the vector
    // pipe would otherwise reach the "s" operand as a VGPR pair -- an assembler error at best; it folds away when the value
    // is in scalar registers already)
#define TRX_SGPR64(P) ((((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)(P) >> 32))) << 32) | \
                       (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(P)))
#define TRX_GLDS16(SBASE, VOFF, LDSADDR)                                                                     \
    {                                                                                                        \
        const unsigned vo_ = (VOFF);                                                                         \
        const unsigned la_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(LDSADDR));                        \
        const unsigned long long sb_ = TRX_SGPR64(SBASE);                                                    \
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(vo_), "s"(sb_), "s"(la_) : "memory", "m0"); \
    }
#define TRX_GLDS4(SBASE, VOFF, LDSADDR)      /* the same with 4 bytes per lane */                            \
    {                                                                                                        \
        unsigned vo_ = (VOFF);                                                                               \
        const unsigned la_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(LDSADDR));                        \
        const unsigned long long sb_ = TRX_SGPR64(SBASE);                                                    \
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : "+v"(vo_) : "s"(sb_), "s"(la_) : "memory", "m0"); \
    }
    const unsigned ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned lds_w = (unsigned)__builtin_amdgcn_readfirstlane((int)(ldsbase + (unsigned)(2 * wave * 1024)));   // this wave's pieces, scalar
#define TRX_ATT_STAGE(KB, BUF)                                                                              \
    {                                                                                                       \
        const unsigned long long kt_ = (unsigned long long)(kbase + (int64_t)(KB) * 64 * rowbytes);         \
        const unsigned long long vt_ = (unsigned long long)(vbase + (int64_t)(KB) * 64 * rowbytes);         \
        const unsigned l_ = lds_w + (unsigned)((BUF) * 16384);                                              \
        if ((KB) * 64 + 64 <= Lk) {                                                                         \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                              \
                TRX_GLDS16(kt_, kofs[i_], l_ + i_ * 1024);                                                  \
                TRX_GLDS16(vt_, vofs[i_], l_ + 8192 + i_ * 1024);                                           \
            }                                                                                               \
        } else { /* tail tile: rows past the last key re-read the last key (they are hidden anyway) */      \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                              \
                const int rw_ = min(8 * (2 * wave + i_) + prow, Lk - 1 - (KB) * 64);                        \
                const unsigned ro_ = (unsigned)rw_ * rowbytes;                                              \
                TRX_GLDS16(kt_, ro_ + (unsigned)((pslot ^ (4 * i_ + (prow >> 1))) * 16), l_ + i_ * 1024);   \
                TRX_GLDS16(vt_, ro_ + (unsigned)((pslot ^ ((prow & 3) << 1)) * 16), l_ + 8192 + i_ * 1024); \
            }                                                                                               \
        }                                                                                                   \
    }
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    // V^T read addresses: row 4(g>>1) + qq (+16 per k-step, +8 for the second read), 16-byte chunk
    // c = 4 db + 2 (g&1) + (pp>>1) stored at slot c ^ ((row & 3) << 1) -- the four rows a 16-lane group
    // touches then sit in four different 32-byte bank groups.  (row & 3) == qq for every read.
    const int other_half = (lane ^ 32) << 2;
    const unsigned vtrA0 = ldsbase + (unsigned)(8192 + (4 * (g >> 1) + qq) * 128 + (((2 * (g & 1) + (pp >> 1)) ^ (qq << 1)) << 4) + 8 * (pp & 1));
    const int kswz = (r >> 1) & 7;
    unsigned kfa[4];   // K fragment addresses: row r (+32 by offset), chunk (2s + hh) ^ swizzle
#pragma unroll
    for (int s = 0; s < 4; ++s) kfa[s] = ldsbase + (unsigned)(r * 128 + (((2 * s + hh) ^ kswz) << 4));
    // dropout hash input of (this lane's query, key pair 0): + 2 hh because register t's key is ... + 4 hh
    const unsigned xdrop = DROP ? drop_base_da(da, (unsigned)(b * H + h)) + (unsigned)qidx * DROP_C1 + (unsigned)(2 * hh) * DROP_C2 : 0u;

    // the key mask of tiles 16j .. 16j+15 is (re)loaded when tile 16j starts: 4 keys per thread
#define TRX_MASK_FILL(KB)                                                                                   \
    {                                                                                                       \
        float mv_[4];                                                                                       \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) mv_[i_] = mkey[min((KB) * 64 + 4 * tid + i_, Lk - 1)]; \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) ldsM[4 * tid + i_] = TRX_MASK_INIT(mv_[i_]);       \
    }
    // vector-memory order per wave: D(0) D(1) | iteration j: D(j+2).  At the top of iteration j tile j
    // must have landed while D(j+1)'s four loads may still fly: vmcnt(4).
    if (!(TRX_ATT_ABL & 128)) { TRX_ATT_STAGE(0, 0); }
    if (nkb > 1 && !(TRX_ATT_ABL & 128)) TRX_ATT_STAGE(1, 1);
    // the mask and Q have landed -- they are older than the last four DMA pieces, whichever tile those belong to (with two tiles
    // staged, tile 0 has landed too; the loop's own wait decides about the tiles).  ONE statement names the registers, so that
    // nothing reads them earlier: with one statement per case hipcc copied the still-empty registers into the other
    // statement's operands in front of the wait (NaN for every Lk <= 64).
    asm volatile("s_waitcnt vmcnt(4)" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]), "+v"(mv0[0]), "+v"(mv0[1]), "+v"(mv0[2]), "+v"(mv0[3]) :: "memory");
    if (keymask) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ldsM[4 * tid + i] = TRX_MASK_INIT(mv0[i]);
    }
#if TRX_ATT_AUG
    // Q times scale * log2 e, once: the first product then yields the exponent's units (attn_softmax_tile_aug)
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
        uint4 u = __builtin_bit_cast(uint4, qf[s_]);
        unsigned* w = reinterpret_cast<unsigned*>(&u);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = pack2bf(__builtin_bit_cast(float, w[i] << 16) * sl2, __builtin_bit_cast(float, w[i] & 0xffff0000u) * sl2);
        qf[s_] = __builtin_bit_cast(bf16x8, u);
    }
    float mref = 0.f;
    bf16x8 qaug = attn_aug_operand(0.f, hh);
    const bf16x8 kones = __builtin_bit_cast(bf16x8, uint4{hh ? 0u : 0x3f803f80u, hh ? 0u : 0x00003f80u, 0u, 0u});
#endif
    int buf = 0;
    TRX_STAMP(3, __builtin_amdgcn_s_memtime());
    for (int kc = 0; kc < nkb; kc += 16) {   // chunks of 16 tiles = the 1024 keys whose mask sits in LDS
    if (keymask && kc > 0) {
        __syncthreads();                     // (rare: Lk > 1024) the previous chunk's mask is no longer read
        TRX_MASK_FILL(kc);
    }
    const int kend = min(nkb, kc + 16);
    for (int kb = kc; kb < kend; ++kb) {
        if (kb + 1 < nkb) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // tile kb is in LDS for everyone; tile kb-1 is no longer read.  Raw barrier: __syncthreads()
        // would drain vmcnt to 0 and with it the prefetch distance
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(TRX_ATT_ABL & 4)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int buf1 = buf == 2 ? 0 : buf + 1, buf2 = buf == 0 ? 2 : buf - 1;
        if (kb + 2 < nkb && !(TRX_ATT_ABL & 128)) TRX_ATT_STAGE(kb + 2, buf2);
        if ((kb & (KS - 1)) == kp) {          // wave-uniform: this wave's tile (always, unless the block splits its keys)
        // ---- S^T = K Q^T for both 32-key halves ----
        f32x16 s0, s1;
        const int key0 = kb * 64;
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
                s0[t] = TRX_MASK_INIT(mrow[min(key0 + kr_, Lk - 1)]);
                s1[t] = TRX_MASK_INIT(mrow[min(key0 + 32 + kr_, Lk - 1)]);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) { s0[t] = 0.f; s1[t] = 0.f; }
        }
        // K fragments through inline asm: a C++ read of `lds` would make the compiler drain vmcnt to 0
        // first (it cannot tell tile kb's buffer from the one tile kb+2 is being DMA'd into)
        bf16x8 ka[4][2];
        {
            const unsigned kb0 = (unsigned)(buf * 16384);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (TRX_ATT_ABL & 8) { asm volatile("" : "=v"(ka[s][0]), "=v"(ka[s][1])); continue; }
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096"
                             : "=&v"(ka[s][0]), "=&v"(ka[s][1]) : "v"(kfa[s] + kb0) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(ka[0][0]), "+v"(ka[0][1]), "+v"(ka[1][0]), "+v"(ka[1][1]),
                           "+v"(ka[2][0]), "+v"(ka[2][1]), "+v"(ka[3][0]), "+v"(ka[3][1]) :: "memory");
        }
        if (TRX_ATT_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (TRX_ATT_ABL & 32) { asm volatile("" : "+v"(s0), "+v"(s1) : "v"(ka[s][0]), "v"(ka[s][1])); continue; }
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s][0], qf[s], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s][1], qf[s], s1, 0, 0, 0);
        }
#if TRX_ATT_AUG
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kones, qaug, s0, 0, 0, 0);      // ... - mref
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kones, qaug, s1, 0, 0, 0);
#endif
        if (TRX_ATT_PRIO) __builtin_amdgcn_s_setprio(0);
        // ---- V^T fragments, first 32 keys: 8 transposed reads in flight under the softmax ----
        const unsigned vtrA = vtrA0 + (unsigned)(buf * 16384), vtrB = vtrA ^ 64u;   // d block 0 / 1
        uint2 vt[4][2][2];   // [k-step of 16 keys][d block][low / high 4 keys]
#define TRX_VT_READ(S)                                                                                        \
    if (TRX_ATT_ABL & 64) asm volatile("" : "=v"(vt[S][0][0]), "=v"(vt[S][0][1]), "=v"(vt[S][1][0]), "=v"(vt[S][1][1])); else \
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
        constexpr int ss = (S) & 1;   /* k-step S = keys 16 S .. 16 S + 15 = pairs 4 ss .. 4 ss + 3 of half HB */ \
        const bf16x8 pf = __builtin_bit_cast(bf16x8, uint4{pk[HB][4 * ss], pk[HB][4 * ss + 1], pk[HB][4 * ss + 2], pk[HB][4 * ss + 3]}); \
        uint4 v0; v0.x = vt[S][0][0].x; v0.y = vt[S][0][0].y; v0.z = vt[S][0][1].x; v0.w = vt[S][0][1].y;     \
        uint4 v1; v1.x = vt[S][1][0].x; v1.y = vt[S][1][0].y; v1.z = vt[S][1][1].x; v1.w = vt[S][1][1].y;     \
        if (TRX_ATT_ABL & 16) asm volatile("" : "+v"(o0), "+v"(o1) : "v"(v0.x), "v"(v0.w), "v"(v1.x), "v"(v1.w), "v"(pf));               \
        else {                                                                                                \
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0), pf, o0, 0, 0, 0);        \
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1), pf, o1, 0, 0, 0); }      \
    }
        TRX_VT_READ(0) TRX_VT_READ(1)
        const bool vis = key0 + 63 > klim_wave_min;   // wave-uniform: some key of the tile is hidden for some lane
        const unsigned xd = xdrop + (unsigned)(kb * 32) * DROP_C2;
        unsigned pk[2][8];   // the tile's probabilities as bf16 pairs: the B operands of the second product
        if (vis) {      // keys out of range or hidden by causality: minus infinity, in place -- then ONE form of the tile for all
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int kr_ = (t & 3) + 8 * (t >> 2) + 4 * hh;
                s0[t] = (key0 + kr_ > klim) ? -__builtin_inff() : s0[t];
                s1[t] = (key0 + 32 + kr_ > klim) ? -__builtin_inff() : s1[t];
            }
        }
#if TRX_ATT_INTERLEAVE
        // Round 6: the second product INSIDE the exponential chain.  A k-step of O^T += V^T P^T needs the probabilities of 16 keys = 8
        // score registers of this lane; so after the row maximum the tile goes in four groups -- 8 exponentials, their bf16 pairs
        // and row sum, then the group's two MFMAs -- and the matrix core works on group g while the wave issues group g + 1's
        // exponentials (64 cycles of MFMA under 64+ cycles of v_exp issue), instead of 8 MFMAs after the whole softmax.  The
        // ablations of profiles/r06_attention_ablation.json price the two products at their full matrix-core time (10.8 of 40 us):
        // nothing overlapped them.  No register is added: the V^T fragments were in flight already.  Measured (same box, interleaved,
        // bit-identical; profiles/r06_attention_interleave_ab.json): -3.6 % at 512 x 512, -1.5 % at 160 x 512, -1 % at causal 160 x 160.
        // (The MFMAs as inline asm, in exactly this order, need two wait states after the last v_cvt_pk -- hipcc pads that hazard for
        // the builtin only -- and were no faster: 42.5 against 42.7 us.)
        {
            float mb = -__builtin_inff();
#pragma unroll
            for (int t = 0; t < 16; ++t) mb = fmaxf(mb, s0[t]);
#pragma unroll
            for (int t = 0; t < 16; ++t) mb = fmaxf(mb, s1[t]);
            {
                float ma = mb, mc = mb;      // (v_permlane32_swap: see attn_softmax_tile)
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mc));
                mb = fmaxf(ma, mc) * sl2;
            }
            const float mt_ = fmaxf(m, mb);
            const float mn = (TRX_ATT_LAZY > 0 && !(mt_ > m + (float)TRX_ATT_LAZY)) ? m : mt_;
            const float mref = (mn == -__builtin_inff()) ? 0.f : mn;
            const float alpha = __builtin_amdgcn_exp2f(m - mref);
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {   // wave-uniform, rare: BEFORE the tile's first MFMA on the outputs
#pragma unroll
                for (int t = 0; t < 16; ++t) { o0[t] *= alpha; o1[t] *= alpha; }
            }
            typedef __attribute__((ext_vector_type(2))) float f32x2;
            const f32x2 sl2v = {sl2, sl2}, nrefv = {-mref, -mref};
            const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
            float ps = 0.f;
#define TRX_EXP_GROUP(S)                                                                                       \
    {                                                                                                          \
        constexpr int hb_ = (S) >> 1, t0_ = 8 * ((S) & 1);                                                     \
        _Pragma("unroll") for (int t = t0_; t < t0_ + 8; t += 2) {                                            \
            const f32x2 a_ = __builtin_elementwise_fma((f32x2){hb_ ? s1[t] : s0[t], hb_ ? s1[t + 1] : s0[t + 1]}, sl2v, nrefv); \
            const float e0_ = __builtin_amdgcn_exp2f(a_.x), e1_ = __builtin_amdgcn_exp2f(a_.y);               \
            if (DROP) {                                                                                        \
                ps += e0_ + e1_;                                                                               \
                const unsigned bits_ = lowbias32(xd + (unsigned)(hb_ * 16 + ((t & 3) >> 1) + 4 * (t >> 2)) * DROP_C2); \
                pk[hb_][t >> 1] = pack2bf(drop_keep(bits_, 0, da.thr) ? e0_ : 0.f, drop_keep(bits_, 1, da.thr) ? e1_ : 0.f); \
            } else {                                                                                           \
                const unsigned w_ = pack2bf(e0_, e1_);                                                         \
                pk[hb_][t >> 1] = w_;                                                                          \
                ps = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w_), ones, ps, false);       \
            }                                                                                                  \
        }                                                                                                      \
    }
            TRX_EXP_GROUP(0)
            TRX_VT_WAIT(0, 1, 0)
            TRX_PV_STEP(0, 0)
            TRX_VT_READ(2) TRX_VT_READ(3)
            __builtin_amdgcn_sched_barrier(0);
            TRX_EXP_GROUP(1)
            TRX_PV_STEP(1, 0)
            __builtin_amdgcn_sched_barrier(0);
            TRX_EXP_GROUP(2)
            TRX_VT_WAIT(2, 3, 0)
            TRX_PV_STEP(2, 1)
            __builtin_amdgcn_sched_barrier(0);
            TRX_EXP_GROUP(3)
            TRX_PV_STEP(3, 1)
#undef TRX_EXP_GROUP
            lsum = lsum * alpha + ps;
            m = mn;
        }
        }
#else
#if TRX_ATT_AUG
        attn_softmax_tile_aug<DROP>(s0, s1, o0, o1, m, lsum, mref, qaug, hh, xd, da.thr, pk);
#else
        attn_softmax_tile<false, DROP>(s0, s1, o0, o1, m, lsum, sl2, key0, hh, klim, xd, da.thr, pk, other_half);
#endif
        // ---- O^T += V^T P^T: second half's reads fly under the first half's MFMAs ----
        TRX_VT_READ(2) TRX_VT_READ(3)
        TRX_VT_WAIT(0, 1, 8)
        if (TRX_ATT_PRIO) __builtin_amdgcn_s_setprio(1);
        TRX_PV_STEP(0, 0) TRX_PV_STEP(1, 0)
        TRX_VT_WAIT(2, 3, 0)
        TRX_PV_STEP(2, 1) TRX_PV_STEP(3, 1)
        if (TRX_ATT_PRIO) __builtin_amdgcn_s_setprio(0);
        }
#endif
        buf = buf1;
        TRX_STAMP(4 + (kb < 26 ? kb : 26), __builtin_amdgcn_s_memtime());
    }
    }
#undef TRX_VT_READ
#undef TRX_VT_WAIT
#undef TRX_PV_STEP
#undef TRX_ATT_STAGE
#undef TRX_MASK_FILL
#undef TRX_MASK_INIT
    float ltot = lsum + __shfl_xor(lsum, 32, 64);
    if (KS > 1) {     // block-uniform: merge the key-split waves of every query unit
        // partial state of a wave: 32 + 2 floats per lane, lane-major rows of float4 (conflict-free 16-byte accesses):
        // [slot][9][64 lanes] float4 in the K / V ring, which nobody reads any more after the barrier
        __syncthreads();
        float4* part = reinterpret_cast<float4*>(lds);
        const int per = 4 / KS;                              // query units of the block
        if (kp > 0) {
            float4* w = part + ((kp - 1) * per + uq) * 9 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w[i * 64] = make_float4(o0[4 * i], o0[4 * i + 1], o0[4 * i + 2], o0[4 * i + 3]);
                w[(4 + i) * 64] = make_float4(o1[4 * i], o1[4 * i + 1], o1[4 * i + 2], o1[4 * i + 3]);
            }
            w[8 * 64] = make_float4(m, ltot, 0.f, 0.f);
        }
        __syncthreads();
        if (kp == 0) {
            const float NINF = -__builtin_inff();
            float mp[3], lp[3], M = m;
            for (int p_ = 1; p_ < KS; ++p_) {
                const float4 ml = part[((p_ - 1) * per + uq) * 9 * 64 + 8 * 64 + lane];
                mp[p_ - 1] = ml.x; lp[p_ - 1] = ml.y;
                M = fmaxf(M, ml.x);
            }
            const float ws = m == NINF ? 0.f : __builtin_amdgcn_exp2f(m - M);     // a wave without a visible key: m = -inf, O = 0, l = 0
#pragma unroll
            for (int t = 0; t < 16; ++t) { o0[t] *= ws; o1[t] *= ws; }
            ltot *= ws;
            for (int p_ = 1; p_ < KS; ++p_) {
                const float wp = mp[p_ - 1] == NINF ? 0.f : __builtin_amdgcn_exp2f(mp[p_ - 1] - M);
                const float4* rd = part + ((p_ - 1) * per + uq) * 9 * 64 + lane;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 a = rd[i * 64], c = rd[(4 + i) * 64];
                    o0[4 * i] = __builtin_fmaf(a.x, wp, o0[4 * i]); o0[4 * i + 1] = __builtin_fmaf(a.y, wp, o0[4 * i + 1]);
                    o0[4 * i + 2] = __builtin_fmaf(a.z, wp, o0[4 * i + 2]); o0[4 * i + 3] = __builtin_fmaf(a.w, wp, o0[4 * i + 3]);
                    o1[4 * i] = __builtin_fmaf(c.x, wp, o1[4 * i]); o1[4 * i + 1] = __builtin_fmaf(c.y, wp, o1[4 * i + 1]);
                    o1[4 * i + 2] = __builtin_fmaf(c.z, wp, o1[4 * i + 2]); o1[4 * i + 3] = __builtin_fmaf(c.w, wp, o1[4 * i + 3]);
                }
                ltot = __builtin_fmaf(lp[p_ - 1], wp, ltot);
            }
            m = M;
        }
    }
    if (qidx < Lq && kp == 0) {
        // natural-log LSE of the scaled, masked scores (what the backward pass recomputes against)
        if (lse && hh == 0) lse[((int64_t)b * H + h) * Lq + qidx] = (m + __builtin_amdgcn_logf(ltot)) * 0.69314718055994530942f;
    }
    // (whole 128-byte rows through LDS -- a wave parks its 32 rows in the K / V ring and reads them back 8 lanes per row -- were
    // measured beside the lane trade below: 40.83 against 41.04 us at 512 x 512, not worth a barrier and 36 lines; round 5)
    {
        const bool live = qidx < Lq && kp == 0;
        store_row_bf16(out + ((int64_t)b * Lq + (live ? qidx : 0)) * H * 64 + (int64_t)h * 64, live, TRX_ATT_WIDE_STORE && nuq == 4, hh, o0, o1,
                       (DROP ? da.inv_keep : 1.0f) / ltot);
    }
    TRX_STAMP(31, __builtin_amdgcn_s_memtime()); TRX_STAMP(1, __builtin_amdgcn_s_memrealtime());
}

#ifdef TRX_NN_LAB      // kernels that measured no faster than attention_fwd_mfma_kernel: lab builds only
#include "../../tools/experiments/attn_fwd_persist.h"
#include "../../tools/experiments/attn_fwd_pp.h"
#endif
#include "attn_fwd_f32.h"
#include "attn_bwd_f32.h"
#include "attn_bwd_mfma.h"
#include "attn_decode.h"

#ifdef TRX_NN_LAB
#define TRX_VALU_PART 2
#include "../../tools/experiments/attn_valu.h"
#undef TRX_VALU_PART
#endif

}  // namespace

extern "C" {

const char* trx_nn_last_error(void) { return g_err.c_str(); }
const char* trx_nn_version(void) { return "trxnn 0.1 (gfx950)"; }
int trx_nn_set_seed_device(const uint64_t* seed_dev) { g_seed_dev = reinterpret_cast<const unsigned long long*>(seed_dev); return TRX_NN_OK; }
int trx_add_layernorm_bwd_blocks(int64_t rows) { const int64_t b = (rows + 3) / 4; return (int)(b < 1 ? 1 : (b > BWD_MAX_BLOCKS ? BWD_MAX_BLOCKS : b)); }

int trx_add_layernorm_fwd_dropout(const void* x, const void* res, const float* gamma, const float* beta, float eps,
                                  int64_t rows, int cols, int dtype, float p, uint64_t seed, void* y, float* mean,
                                  float* rstd, void* stream) {
    if (!x || !gamma || !beta || !y || rows < 0 || cols <= 0) return fail(TRX_NN_EINVAL, "add_layernorm_fwd: bad argument");
    if (dtype != TRX_NN_F32 && dtype != TRX_NN_BF16) return fail(TRX_NN_EINVAL, "unknown dtype");
    if (!(p >= 0.f && p < 1.f)) return fail(TRX_NN_EINVAL, "dropout probability must be in [0, 1)");
    if (rows == 0) return TRX_NN_OK;
    const Drop drop = make_drop(p, seed, 0);
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int V = dtype == TRX_NN_BF16 ? 8 : 4;
    // (gamma / beta are read 16 bytes at a time by the vector kernel too: a parameter view at an odd offset takes the narrow kernel)
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res) |
                           reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0;
    const bool vec = aligned && cols % V == 0 && cols <= 64 * NCH * V;
    if (vec) {
#define TRX_LNF(NC_)                                                                                                              \
        if (dtype == TRX_NN_BF16) hipLaunchKernelGGL((add_ln_fwd_vec_kernel<true, false, NC_>), grid, block, 0, st, x, res, gamma, beta, eps, rows, cols, y, mean, rstd, drop); \
        else hipLaunchKernelGGL((add_ln_fwd_vec_kernel<false, false, NC_>), grid, block, 0, st, x, res, gamma, beta, eps, rows, cols, y, mean, rstd, drop);
        TRX_NC_SWITCH(chunks_per_lane(cols, V), TRX_LNF)
#undef TRX_LNF
    } else {
        if (dtype == TRX_NN_BF16) hipLaunchKernelGGL(add_ln_fwd_kernel<true>, grid, block, 0, st, x, res, gamma, beta, eps, rows, cols, y, mean, rstd, drop);
        else hipLaunchKernelGGL(add_ln_fwd_kernel<false>, grid, block, 0, st, x, res, gamma, beta, eps, rows, cols, y, mean, rstd, drop);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRX_NN_OK : fail(TRX_NN_EHIP, hipGetErrorString(e));
}

int trx_add_layernorm_fwd(const void* x, const void* res, const float* gamma, const float* beta, float eps,
                          int64_t rows, int cols, int dtype, void* y, float* mean, float* rstd, void* stream) {
    return trx_add_layernorm_fwd_dropout(x, res, gamma, beta, eps, rows, cols, dtype, 0.f, 0, y, mean, rstd, stream);
}

int trx_add_layernorm_bwd_dropout(const void* dy, const void* x, const void* res, const float* gamma, const float* mean,
                                  const float* rstd, int64_t rows, int cols, int dtype, float p, uint64_t seed, void* dz,
                                  void* dx, float* dgamma, float* dbeta, float* ws, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dz || !dgamma || !dbeta || !ws || rows <= 0 || cols <= 0)
        return fail(TRX_NN_EINVAL, "add_layernorm_bwd: bad argument");
    if (dtype != TRX_NN_F32 && dtype != TRX_NN_BF16) return fail(TRX_NN_EINVAL, "unknown dtype");
    if (!(p >= 0.f && p < 1.f)) return fail(TRX_NN_EINVAL, "dropout probability must be in [0, 1)");
    if (p > 0.f && !dx) return fail(TRX_NN_EINVAL, "add_layernorm_bwd: dropout needs the separate dx output");
    const Drop drop = make_drop(p, seed, 0);
    if ((size_t)cols * 8 * sizeof(float) > 160 * 1024) return fail(TRX_NN_EINVAL, "cols too large for the LDS partials");
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)cols * 8 * sizeof(float);
    const int V = dtype == TRX_NN_BF16 ? 8 : 4;
    const int nblk = trx_add_layernorm_bwd_blocks(rows);
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(res) |
                           reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(gamma)) & 15) == 0;
    const bool vec = aligned && cols % V == 0 && cols <= 64 * NCH * V;
#define TRX_LAUNCH_LNB(KERNEL)                                                                                              \
    {                                                                                                                       \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(KERNEL, dim3(nblk), dim3(256), lds, st, dy, x, res, gamma, mean, rstd, rows, cols, dz,           \
                           p > 0.f ? dx : nullptr, ws, nblk, drop);                                                         \
    }
#define TRX_LNB(NC_) { if (dtype == TRX_NN_BF16) TRX_LAUNCH_LNB((add_ln_bwd_vec_kernel<true, false, NC_>)) else TRX_LAUNCH_LNB((add_ln_bwd_vec_kernel<false, false, NC_>)) }
    if (vec) { TRX_NC_SWITCH(chunks_per_lane(cols, V), TRX_LNB) }
    else if (dtype == TRX_NN_BF16) TRX_LAUNCH_LNB(add_ln_bwd_kernel<true>)
    else TRX_LAUNCH_LNB(add_ln_bwd_kernel<false>)
#undef TRX_LNB
#undef TRX_LAUNCH_LNB
    hipLaunchKernelGGL(add_ln_bwd_reduce_kernel, dim3((cols + 15) / 16, 2), dim3(256), 0, st, ws, nblk, cols, dgamma, dbeta);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRX_NN_OK : fail(TRX_NN_EHIP, hipGetErrorString(e));
}

int trx_add_layernorm_bwd(const void* dy, const void* x, const void* res, const float* gamma, const float* mean,
                          const float* rstd, int64_t rows, int cols, int dtype, void* dz, float* dgamma,
                          float* dbeta, float* ws, void* stream) {
    return trx_add_layernorm_bwd_dropout(dy, x, res, gamma, mean, rstd, rows, cols, dtype, 0.f, 0, dz, nullptr, dgamma, dbeta, ws, stream);
}

int trx_add_layernorm_fwd_mixed(const void* x_bf16, const void* res_f32, const float* gamma, const float* beta, float eps,
                                int64_t rows, int cols, float p, uint64_t seed, void* y_f32, void* y_bf16, float* mean, float* rstd,
                                const float* x_bias, void* stream) {
    if (!x_bf16 || !res_f32 || !gamma || !beta || !y_f32 || rows < 0 || cols <= 0) return fail(TRX_NN_EINVAL, "add_layernorm_fwd_mixed: bad argument");
    if (!(p >= 0.f && p < 1.f)) return fail(TRX_NN_EINVAL, "dropout probability must be in [0, 1)");
    if (cols % 4 != 0 || cols > 64 * NCH * 4) return fail(TRX_NN_EINVAL, "add_layernorm_fwd_mixed: cols must be a multiple of 4 and <= 1024");
    if (((reinterpret_cast<uintptr_t>(res_f32) | reinterpret_cast<uintptr_t>(y_f32) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) |
          reinterpret_cast<uintptr_t>(x_bias)) & 15) ||
        ((reinterpret_cast<uintptr_t>(x_bf16) | reinterpret_cast<uintptr_t>(y_bf16)) & 7))
        return fail(TRX_NN_EINVAL, "add_layernorm_fwd_mixed: operands must be 16-byte (bf16 ones: 8-byte) aligned (gamma, beta and the bias are read 16 bytes at a time)");
    if (rows == 0) return TRX_NN_OK;
    const Drop drop = make_drop(p, seed, 0);
#define TRX_LNFM(NC_) hipLaunchKernelGGL((add_ln_fwd_vec_kernel<false, true, NC_>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, \
                                         x_bf16, res_f32, gamma, beta, eps, rows, cols, y_f32, mean, rstd, drop, y_bf16, x_bias);
    TRX_NC_SWITCH(chunks_per_lane(cols, 4), TRX_LNFM)
#undef TRX_LNFM
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRX_NN_OK : fail(TRX_NN_EHIP, hipGetErrorString(e));
}

int trx_add_layernorm_bwd_mixed(const void* dy_f32, const void* dy_bf16, const void* x_bf16, const void* res_f32, const float* gamma,
                                const float* mean, const float* rstd, int64_t rows, int cols, float p, uint64_t seed, void* dz_f32,
                                void* dx_bf16, float* dgamma, float* dbeta, const float* x_bias, float* dx_bias, float* ws,
                                void* stream) {
    // dgamma == dbeta == NULL: the first stage only -- the partial column sums stay in ws for trx_add_layernorm_bwd_reduce_many
    const bool partials_only = !dgamma && !dbeta;
    if (partials_only ? dx_bias != nullptr : (x_bias == nullptr) != (dx_bias == nullptr))
        return fail(TRX_NN_EINVAL, "add_layernorm_bwd_mixed: x_bias and dx_bias go together (partials only: no dx_bias)");
    if ((!dy_f32 && !dy_bf16) || !x_bf16 || !res_f32 || !gamma || !mean || !rstd || !dz_f32 || !dx_bf16 || (!partials_only && (!dgamma || !dbeta)) || !ws || rows <= 0 || cols <= 0)
        return fail(TRX_NN_EINVAL, "add_layernorm_bwd_mixed: bad argument");
    if (!(p >= 0.f && p < 1.f)) return fail(TRX_NN_EINVAL, "dropout probability must be in [0, 1)");
    if (cols % 4 != 0 || cols > 64 * NCH * 4) return fail(TRX_NN_EINVAL, "add_layernorm_bwd_mixed: cols must be a multiple of 4 and <= 1024");
    if (((reinterpret_cast<uintptr_t>(res_f32) | reinterpret_cast<uintptr_t>(dy_f32) | reinterpret_cast<uintptr_t>(dz_f32) | reinterpret_cast<uintptr_t>(gamma)) & 15) ||
        ((reinterpret_cast<uintptr_t>(x_bf16) | reinterpret_cast<uintptr_t>(dx_bf16) | reinterpret_cast<uintptr_t>(dy_bf16)) & 7))
        return fail(TRX_NN_EINVAL, "add_layernorm_bwd_mixed: operands must be 16-byte (x, dx: 8-byte) aligned");
    const Drop drop = make_drop(p, seed, 0);
    const int nblk = trx_add_layernorm_bwd_blocks(rows);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)cols * 8 * sizeof(float);
#define TRX_LNBM(NC_) hipLaunchKernelGGL((add_ln_bwd_vec_kernel<false, true, NC_>), dim3(nblk), dim3(256), lds, st, dy_f32, x_bf16, res_f32, gamma, mean, rstd, \
                                         rows, cols, dz_f32, dx_bf16, ws, nblk, drop, dy_bf16, x_bias);
    TRX_NC_SWITCH(chunks_per_lane(cols, 4), TRX_LNBM)
#undef TRX_LNBM
    if (!partials_only)
        hipLaunchKernelGGL(add_ln_bwd_reduce_kernel, dim3((cols + 15) / 16, x_bias ? 3 : 2), dim3(256), 0, st, ws, nblk, cols, dgamma, dbeta, dx_bias);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRX_NN_OK : fail(TRX_NN_EHIP, hipGetErrorString(e));
}

int trx_add_layernorm_bwd_reduce_many(const trx_ln_reduce_item* items, int n, int cols, void* stream) {
    if (!items || n <= 0 || cols <= 0) return fail(TRX_NN_EINVAL, "add_layernorm_bwd_reduce_many: bad argument");
    for (int i = 0; i < n; ++i)
        if (!items[i].ws || !items[i].dgamma || !items[i].dbeta || items[i].nblk <= 0)
            return fail(TRX_NN_EINVAL, "add_layernorm_bwd_reduce_many: an item without partials, dgamma / dbeta or blocks");
    for (int i0 = 0; i0 < n; i0 += TRX_LN_REDUCE_MAX) {
        LnReduceTable t{};
        const int m = n - i0 < TRX_LN_REDUCE_MAX ? n - i0 : TRX_LN_REDUCE_MAX;
        for (int i = 0; i < m; ++i) t.it[i] = items[i0 + i];
        hipLaunchKernelGGL(add_ln_bwd_reduce_many_kernel, dim3((cols + 15) / 16, 3, m), dim3(256), 0, (hipStream_t)stream, t, cols);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRX_NN_OK : fail(TRX_NN_EHIP, hipGetErrorString(e));
}

int trx_dropout_keep_mask(uint64_t seed, float p, int64_t streams, int64_t rows, int64_t cols, unsigned char* keep, void* stream) {
    if (!keep || streams <= 0 || rows <= 0 || cols <= 0 || !(p >= 0.f && p < 1.f)) return fail(TRX_NN_EINVAL, "dropout_keep_mask: bad argument");
    const int64_t n = streams * rows * cols;
    const unsigned grid = (unsigned)((n + 255) / 256 > 65536 ? 65536 : (n + 255) / 256);
    hipLaunchKernelGGL(dropout_keep_mask_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (unsigned)seed, (unsigned)(seed >> 32),
                       p > 0.f ? drop_thr(p) : 0u, streams, rows, cols, keep, g_seed_dev);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRX_NN_OK : fail(TRX_NN_EHIP, hipGetErrorString(e));
}

static DropArgs make_drop_args(float p, uint64_t seed) {
    DropArgs a;
    a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32);
    a.seed_dev = g_seed_dev;
    a.thr = p > 0.f ? drop_thr(p) : 0u;
    a.inv_keep = 1.0f / (1.0f - p);
    a.kv_bs = 0;
    a.ldq = 0; a.ldk = 0;
    return a;
}

static int attention_fwd_impl(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                              int B, int H, int Lq, int Lk, int64_t kv_bs, float scale, int dtype, float p, uint64_t seed,
                              void* out, float* lse, void* stream, int ldq = 0, int ldk = 0) {
    if ((ldq || ldk) && dtype != TRX_NN_BF16)
        return fail(TRX_NN_EINVAL, "attention_fwd: strided operands are supported by the bf16 matrix-core path only");
    if ((ldq && ldq < H * 64) || (ldk && ldk < H * 64) || ((ldq | ldk) & 7))
        return fail(TRX_NN_EINVAL, "attention_fwd: row strides must be >= H * 64 and multiples of 8 elements");
    if (!q || !k || !v || !out || B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return fail(TRX_NN_EINVAL, "attention_fwd: bad argument");
    if (kv_bs != 0 && kv_bs < (int64_t)Lk * H * 64) return fail(TRX_NN_EINVAL, "attention_fwd: key/value batch stride smaller than one batch item");
    if (mask_mode != TRX_NN_MASK_NONE && !mask) return fail(TRX_NN_EINVAL, "attention_fwd: mask is null");
    if (mask_mode < 0 || mask_mode > 2) return fail(TRX_NN_EINVAL, "attention_fwd: unknown mask mode");
    if (dtype != TRX_NN_F32 && dtype != TRX_NN_BF16) return fail(TRX_NN_EINVAL, "unknown dtype");
    if (!(p >= 0.f && p < 1.f)) return fail(TRX_NN_EINVAL, "dropout probability must be in [0, 1)");
    if (!(scale > 0.f)) return fail(TRX_NN_EINVAL, "attention_fwd: scale must be positive");
    // the matrix-core kernels move q, k, v by 16-byte loads / LDS-DMA and store whole 16-byte pieces of an output row
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(out)) & 15)
        return fail(TRX_NN_EINVAL, "attention_fwd: q, k, v and out must be 16-byte aligned");
    DropArgs da = make_drop_args(p, seed);
    da.kv_bs = kv_bs; da.ldq = ldq; da.ldk = ldk;
    hipStream_t st = (hipStream_t)stream;
#ifdef TRX_NN_LAB
    const int qblocks = (Lq + 63) / 64;
    dim3 grid((unsigned)((int64_t)B * H * qblocks)), block(64);
    static const bool force_valu = getenv("TRX_NN_ATTN_VALU") != nullptr;
#else
    constexpr bool force_valu = false;
#endif
    if (dtype == TRX_NN_BF16 && !force_valu) {
        dim3 g2((unsigned)((int64_t)B * H * ((Lq + 127) / 128))), b2(256);
#ifdef TRX_NN_LAB
        // lab builds: TRX_NN_ATTN_PP=1 / TRX_NN_ATTN_PERSIST=1 put the encoder's shape class on the kernels of
        // tools/experiments/attn_fwd_pp.h (33 % slower at 512 x 512) / attn_fwd_persist.h (39.8 against 39.9 us)
        static const bool use_pp = getenv("TRX_NN_ATTN_PP") != nullptr;
        const bool pp = use_pp && Lq >= 256 && Lk <= 1024 && mask_mode != TRX_NN_MASK_FULL;
        dim3 g3((unsigned)((int64_t)B * H * ((Lq + 255) / 256))), b3(512);
        static const bool use_persist = getenv("TRX_NN_ATTN_PERSIST") && atoi(getenv("TRX_NN_ATTN_PERSIST")) != 0;
        const int nitems = (int)((int64_t)B * H * (Lq / 128));
        const int pslots = 768;              // 3 workgroups per CU
        const bool persist = use_persist && !pp && !causal && Lq % 128 == 0 && Lk % 64 == 0 && Lk >= 128 && Lk <= 512 && mask_mode != TRX_NN_MASK_FULL && nitems > pslots;
#define TRX_LAUNCH_MFMA(MM_, DROP_)                                                                                       \
    do {                                                                                                                  \
        if (persist) {                                                                                                    \
            hipLaunchKernelGGL((attention_fwd_persist_kernel<(MM_) == TRX_NN_MASK_FULL ? TRX_NN_MASK_NONE : (MM_), DROP_>), dim3(pslots), b2, 0, st, \
                               (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, mask, B, H, Lq, Lk, scale, (bf16_t*)out, lse, da, nitems); \
        }                                                                                                                 \
        else if (pp) {                                                                                                         \
            auto kern_ = attention_fwd_pp_kernel<(MM_) == TRX_NN_MASK_FULL ? TRX_NN_MASK_NONE : (MM_), DROP_>;            \
            static std::atomic<unsigned long long> attr_devs_{0ull};   /* the LDS limit is a per-device attribute */          \
            int dev_ = 0;                                                                                                 \
            (void)hipGetDevice(&dev_);                                                                                    \
            if (!(attr_devs_.load() & (1ull << (dev_ & 63)))) {                                                           \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern_), hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES); \
                attr_devs_.fetch_or(1ull << (dev_ & 63));                                                                 \
            }                                                                                                             \
            hipLaunchKernelGGL(kern_, g3, b3, PP_LDS_BYTES, st, (const bf16_t*)q, (const bf16_t*)k,                       \
                               (const bf16_t*)v, mask, causal, B, H, Lq, Lk, scale, (bf16_t*)out, lse, da);               \
        }                                                                                                                 \
        else hipLaunchKernelGGL((attention_fwd_mfma_kernel<MM_, DROP_>), g2, b2, 0, st, (const bf16_t*)q, (const bf16_t*)k, \
                                (const bf16_t*)v, mask, causal, B, H, Lq, Lk, scale, (bf16_t*)out, lse, da);              \
    } while (0)
#else
#define TRX_LAUNCH_MFMA(MM_, DROP_)                                                                                       \
    hipLaunchKernelGGL((attention_fwd_mfma_kernel<MM_, DROP_>), g2, b2, 0, st, (const bf16_t*)q, (const bf16_t*)k,        \
                       (const bf16_t*)v, mask, causal, B, H, Lq, Lk, scale, (bf16_t*)out, lse, da)
#endif
        if (da.thr) {
            if (mask_mode == TRX_NN_MASK_NONE) TRX_LAUNCH_MFMA(TRX_NN_MASK_NONE, true);
            else if (mask_mode == TRX_NN_MASK_KEY) TRX_LAUNCH_MFMA(TRX_NN_MASK_KEY, true);
            else TRX_LAUNCH_MFMA(TRX_NN_MASK_FULL, true);
        } else {
            if (mask_mode == TRX_NN_MASK_NONE) TRX_LAUNCH_MFMA(TRX_NN_MASK_NONE, false);
            else if (mask_mode == TRX_NN_MASK_KEY) TRX_LAUNCH_MFMA(TRX_NN_MASK_KEY, false);
            else TRX_LAUNCH_MFMA(TRX_NN_MASK_FULL, false);
        }
#undef TRX_LAUNCH_MFMA
    } else if (dtype == TRX_NN_F32 && !force_valu) {
        // fp32 on the matrix cores (attn_fwd_f32.h: v_mfma_f32_16x16x4_f32, exact fp32): 64 queries per workgroup
        dim3 gf((unsigned)((int64_t)B * H * ((Lq + 63) / 64))), bf(256);
#define TRX_LAUNCH_F32(MM_, DROP_) hipLaunchKernelGGL((attention_fwd_f32_mfma_kernel<MM_, DROP_>), gf, bf, 0, st, (const float*)q, (const float*)k, \
                                                      (const float*)v, mask, causal, B, H, Lq, Lk, scale, (float*)out, lse, da)
        if (da.thr) {
            if (mask_mode == TRX_NN_MASK_NONE) TRX_LAUNCH_F32(TRX_NN_MASK_NONE, true);
            else if (mask_mode == TRX_NN_MASK_KEY) TRX_LAUNCH_F32(TRX_NN_MASK_KEY, true);
            else TRX_LAUNCH_F32(TRX_NN_MASK_FULL, true);
        } else {
            if (mask_mode == TRX_NN_MASK_NONE) TRX_LAUNCH_F32(TRX_NN_MASK_NONE, false);
            else if (mask_mode == TRX_NN_MASK_KEY) TRX_LAUNCH_F32(TRX_NN_MASK_KEY, false);
            else TRX_LAUNCH_F32(TRX_NN_MASK_FULL, false);
        }
#undef TRX_LAUNCH_F32
    }
#ifdef TRX_NN_LAB
    else {
#define TRX_LAUNCH_VALU(BF_, DROP_) hipLaunchKernelGGL((attention_fwd_kernel<BF_, DROP_>), grid, block, 0, st, q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, scale, out, lse, da)
        if (dtype == TRX_NN_BF16) { if (da.thr) TRX_LAUNCH_VALU(true, true); else TRX_LAUNCH_VALU(true, false); }
        else { if (da.thr) TRX_LAUNCH_VALU(false, true); else TRX_LAUNCH_VALU(false, false); }
#undef TRX_LAUNCH_VALU
    }
#endif
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRX_NN_OK : fail(TRX_NN_EHIP, hipGetErrorString(e));
}

int trx_attention_fwd_dropout(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                              int B, int H, int Lq, int Lk, float scale, int dtype, float p, uint64_t seed, void* out,
                              float* lse, void* stream) {
    return attention_fwd_impl(q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, 0, scale, dtype, p, seed, out, lse, stream);
}

int trx_attention_fwd_kvcache(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                              int B, int H, int Lq, int Lk, int64_t kv_batch_stride, float scale, int dtype, void* out,
                              void* stream) {
    return attention_fwd_impl(q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, kv_batch_stride, scale, dtype, 0.f, 0, out, nullptr, stream);
}

int trx_attention_fwd_lse(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                          int B, int H, int Lq, int Lk, float scale, int dtype, void* out, float* lse, void* stream) {
    return trx_attention_fwd_dropout(q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, scale, dtype, 0.f, 0, out, lse, stream);
}

int trx_attention_fwd(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                      int B, int H, int Lq, int Lk, float scale, int dtype, void* out, void* stream) {
    return trx_attention_fwd_dropout(q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, scale, dtype, 0.f, 0, out, nullptr, stream);
}

static int attention_bwd_impl(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                              int B, int H, int Lq, int Lk, float scale, int dtype, float p, uint64_t seed, const void* out,
                              const void* dout, const float* lse, void* dq, void* dk, void* dv, void* stream, int ldq, int ldk,
                              void* ws_user = nullptr) {
    if ((ldq || ldk) && dtype != TRX_NN_BF16)
        return fail(TRX_NN_EINVAL, "attention_bwd: strided operands are supported by the bf16 matrix-core path only");
    if ((ldq && ldq < H * 64) || (ldk && ldk < H * 64) || ((ldq | ldk) & 7))
        return fail(TRX_NN_EINVAL, "attention_bwd: row strides must be >= H * 64 and multiples of 8 elements");
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0)
        return fail(TRX_NN_EINVAL, "attention_bwd: bad argument");
    if (mask_mode != TRX_NN_MASK_NONE && !mask) return fail(TRX_NN_EINVAL, "attention_bwd: mask is null");
    if (mask_mode < 0 || mask_mode > 2) return fail(TRX_NN_EINVAL, "attention_bwd: unknown mask mode");
    if (dtype != TRX_NN_F32 && dtype != TRX_NN_BF16) return fail(TRX_NN_EINVAL, "unknown dtype");
    if (!(p >= 0.f && p < 1.f)) return fail(TRX_NN_EINVAL, "dropout probability must be in [0, 1)");
    if (!(scale > 0.f)) return fail(TRX_NN_EINVAL, "attention_bwd: scale must be positive");
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(dout) |
         reinterpret_cast<uintptr_t>(dq) | reinterpret_cast<uintptr_t>(dk) | reinterpret_cast<uintptr_t>(dv)) & 15)
        return fail(TRX_NN_EINVAL, "attention_bwd: q, k, v, out, dout, dq, dk and dv must be 16-byte aligned");
    DropArgs da = make_drop_args(p, seed);
    da.ldq = ldq; da.ldk = ldk;
    hipStream_t st = (hipStream_t)stream;
#ifdef TRX_NN_LAB
    static const bool force_valu = getenv("TRX_NN_ATTN_VALU") != nullptr;
#else
    constexpr bool force_valu = false;
#endif
    if (dtype == TRX_NN_BF16 && !force_valu) {
        // matrix-core path; its per-query scalars (-lse/scale, -delta) live in a stream-ordered scratch
        const size_t n = (size_t)B * H * Lq;
        float* ws = (float*)ws_user;        // caller's scratch (2 B H Lq floats: trx_attention_bwd_ws); else a stream-ordered one
        if (!ws && (hipMallocAsync((void**)&ws, 2 * n * sizeof(float), st) != hipSuccess || !ws)) {
            (void)hipGetLastError();
            return fail(TRX_NN_EHIP, "attention_bwd: scratch allocation failed");
        }
        float *negl = ws, *negd = ws + n;
        dim3 g1((unsigned)((int64_t)B * H * ((Lq + 127) / 128))), g2((unsigned)((int64_t)B * H * ((Lk + 127) / 128))), b2(256);
#define TRX_LAUNCH_BWD(MM_, DROP_)                                                                                          \
    hipLaunchKernelGGL((attention_bwd_dq_mfma_kernel<MM_, DROP_>), g1, b2, 0, st, (const bf16_t*)q, (const bf16_t*)k,       \
                       (const bf16_t*)v, mask, causal, B, H, Lq, Lk, scale, (const bf16_t*)dout, (const bf16_t*)out, lse, negl, negd, (bf16_t*)dq, da); \
    hipLaunchKernelGGL((attention_bwd_dkv_mfma_kernel<MM_, DROP_>), g2, b2, 0, st, (const bf16_t*)q, (const bf16_t*)k,      \
                       (const bf16_t*)v, mask, causal, B, H, Lq, Lk, scale, (const bf16_t*)dout, negl, negd, (bf16_t*)dk, (bf16_t*)dv, da)
        if (da.thr) {
            if (mask_mode == TRX_NN_MASK_NONE) { TRX_LAUNCH_BWD(TRX_NN_MASK_NONE, true); }
            else if (mask_mode == TRX_NN_MASK_KEY) { TRX_LAUNCH_BWD(TRX_NN_MASK_KEY, true); }
            else { TRX_LAUNCH_BWD(TRX_NN_MASK_FULL, true); }
        } else {
            if (mask_mode == TRX_NN_MASK_NONE) { TRX_LAUNCH_BWD(TRX_NN_MASK_NONE, false); }
            else if (mask_mode == TRX_NN_MASK_KEY) { TRX_LAUNCH_BWD(TRX_NN_MASK_KEY, false); }
            else { TRX_LAUNCH_BWD(TRX_NN_MASK_FULL, false); }
        }
#undef TRX_LAUNCH_BWD
        hipError_t e1 = hipGetLastError();
        if (!ws_user) (void)hipFreeAsync(ws, st);
        if (e1 != hipSuccess) return fail(TRX_NN_EHIP, hipGetErrorString(e1));
        return TRX_NN_OK;
    }
    if (dtype == TRX_NN_F32 && !force_valu) {
        // fp32 on the matrix cores (attn_bwd_f32.h): the dq pass makes delta = dO . O and hands it to the dk/dv pass
        const size_t n = (size_t)B * H * Lq;
        float* ws = (float*)ws_user;
        if (!ws && (hipMallocAsync((void**)&ws, n * sizeof(float), st) != hipSuccess || !ws)) {
            (void)hipGetLastError();
            return fail(TRX_NN_EHIP, "attention_bwd: scratch allocation failed");
        }
        dim3 gfq((unsigned)((int64_t)B * H * ((Lq + 63) / 64))), gfk((unsigned)((int64_t)B * H * ((Lk + 63) / 64))), bf(256);
#define TRX_LAUNCH_BWD_F32(MM_, DROP_)                                                                                      \
    hipLaunchKernelGGL((attention_bwd_dq_f32_mfma_kernel<MM_, DROP_>), gfq, bf, 0, st, (const float*)q, (const float*)k,    \
                       (const float*)v, mask, causal, B, H, Lq, Lk, scale, (const float*)out, (const float*)dout, lse, (float*)dq, ws, da); \
    hipLaunchKernelGGL((attention_bwd_dkv_f32_mfma_kernel<MM_, DROP_>), gfk, bf, 0, st, (const float*)q, (const float*)k,   \
                       (const float*)v, mask, causal, B, H, Lq, Lk, scale, (const float*)dout, lse, ws, (float*)dk, (float*)dv, da)
        if (da.thr) {
            if (mask_mode == TRX_NN_MASK_NONE) { TRX_LAUNCH_BWD_F32(TRX_NN_MASK_NONE, true); }
            else if (mask_mode == TRX_NN_MASK_KEY) { TRX_LAUNCH_BWD_F32(TRX_NN_MASK_KEY, true); }
            else { TRX_LAUNCH_BWD_F32(TRX_NN_MASK_FULL, true); }
        } else {
            if (mask_mode == TRX_NN_MASK_NONE) { TRX_LAUNCH_BWD_F32(TRX_NN_MASK_NONE, false); }
            else if (mask_mode == TRX_NN_MASK_KEY) { TRX_LAUNCH_BWD_F32(TRX_NN_MASK_KEY, false); }
            else { TRX_LAUNCH_BWD_F32(TRX_NN_MASK_FULL, false); }
        }
#undef TRX_LAUNCH_BWD_F32
        hipError_t e1 = hipGetLastError();
        if (!ws_user) (void)hipFreeAsync(ws, st);
        return e1 == hipSuccess ? TRX_NN_OK : fail(TRX_NN_EHIP, hipGetErrorString(e1));
    }
#ifdef TRX_NN_LAB
    dim3 gq((unsigned)((int64_t)B * H * ((Lq + 63) / 64))), gk((unsigned)((int64_t)B * H * ((Lk + 63) / 64))), block(64);
#define TRX_LAUNCH_VALU_BWD(BF_, DROP_)                                                                                     \
    hipLaunchKernelGGL((attention_bwd_dq_kernel<BF_, DROP_>), gq, block, 0, st, q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, \
                       scale, out, dout, lse, dq, da);                                                                      \
    hipLaunchKernelGGL((attention_bwd_dkv_kernel<BF_, DROP_>), gk, block, 0, st, q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, \
                       scale, out, dout, lse, dk, dv, da)
    if (dtype == TRX_NN_BF16) { if (da.thr) { TRX_LAUNCH_VALU_BWD(true, true); } else { TRX_LAUNCH_VALU_BWD(true, false); } }
    else { if (da.thr) { TRX_LAUNCH_VALU_BWD(false, true); } else { TRX_LAUNCH_VALU_BWD(false, false); } }
#undef TRX_LAUNCH_VALU_BWD
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRX_NN_OK : fail(TRX_NN_EHIP, hipGetErrorString(e));
#else
    return fail(TRX_NN_EINVAL, "attention_bwd: unknown dtype");
#endif
}

int trx_attention_bwd_dropout(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                              int B, int H, int Lq, int Lk, float scale, int dtype, float p, uint64_t seed, const void* out,
                              const void* dout, const float* lse, void* dq, void* dk, void* dv, void* stream) {
    return attention_bwd_impl(q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, scale, dtype, p, seed, out, dout, lse, dq, dk, dv, stream, 0, 0);
}

int trx_attention_fwd_strided(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                              int B, int H, int Lq, int Lk, int ldq, int ldkv, float scale, float p, uint64_t seed, void* out,
                              float* lse, void* stream) {
    return attention_fwd_impl(q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, 0, scale, TRX_NN_BF16, p, seed, out, lse, stream, ldq, ldkv);
}

int trx_attention_bwd_strided(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                              int B, int H, int Lq, int Lk, int ldq, int ldkv, float scale, float p, uint64_t seed, const void* out,
                              const void* dout, const float* lse, void* dq, void* dk, void* dv, void* stream) {
    return attention_bwd_impl(q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, scale, TRX_NN_BF16, p, seed, out, dout, lse, dq, dk, dv, stream, ldq, ldkv);
}

int64_t trx_attention_bwd_ws_bytes(int B, int H, int Lq) { return (int64_t)2 * B * H * Lq * (int64_t)sizeof(float); }

int trx_attention_bwd_ws(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                         int B, int H, int Lq, int Lk, int ldq, int ldkv, float scale, int dtype, float p, uint64_t seed, const void* out,
                         const void* dout, const float* lse, void* dq, void* dk, void* dv, void* ws, void* stream) {
    return attention_bwd_impl(q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, scale, dtype, p, seed, out, dout, lse, dq, dk, dv, stream, ldq, ldkv, ws);
}

int trx_attention_decode_gather(const void* q, int ldq, const void* kv, const int32_t* anc, const int64_t* t_dev, void* out, int n, int H,
                                int T, float scale, void* stream) {
    if (!q || !kv || !anc || !t_dev || !out || n <= 0 || H <= 0 || T <= 0 || ldq < H * 64 || ldq % 8)
        return fail(TRX_NN_EINVAL, "attention_decode_gather: bad argument");
    if (T > DEC_MAX_T) return fail(TRX_NN_EINVAL, "attention_decode_gather: at most 256 cache positions");
    const int waves = n * H;
    hipLaunchKernelGGL(attention_decode_gather_kernel, dim3((waves + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, ldq,
                       (const bf16_t*)kv, anc, (const long long*)t_dev, (bf16_t*)out, n, H, T, scale * 1.4426950408889634f);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRX_NN_OK : fail(TRX_NN_EHIP, hipGetErrorString(e));
}

int trx_attention_bwd(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                      int B, int H, int Lq, int Lk, float scale, int dtype, const void* out, const void* dout,
                      const float* lse, void* dq, void* dk, void* dv, void* stream) {
    return trx_attention_bwd_dropout(q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, scale, dtype, 0.f, 0, out, dout, lse, dq, dk, dv, stream);
}

}  // extern "C"
