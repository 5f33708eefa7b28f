// nn_ops.hip -- predictor hot spots on gfx950: fused residual-add + LayerNorm (forward, backward)
// and attention forward.  C ABI in include/trx_nn.h.  Wave = 64.
//
// Replaces (inside the Hugging Face modules the reference instantiates at textreact/model.py:21-31):
//   LayerNorm(dense(h) + residual)  -- BertSelfOutput / BertOutput / embeddings / lm_head
//   softmax(q k^T / 8 + mask) v      -- Bert/Roberta self-, cross- and causal attention, heads of 64
#include "../../include/trx_nn.h"
#include <hip/hip_runtime.h>
#include <string>
#include <cstdlib>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& m) { g_err = m; return code; }

typedef unsigned short bf16_t;
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
template <bool BF> __device__ __forceinline__ float ld(const void* p, int64_t i) {
    return BF ? bf2f(reinterpret_cast<const bf16_t*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}
template <bool BF> __device__ __forceinline__ void st(void* p, int64_t i, float v) {
    if (BF) reinterpret_cast<bf16_t*>(p)[i] = f2bf(v); else reinterpret_cast<float*>(p)[i] = v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- add + LayerNorm forward: one wave per row, the row cached in registers -----------------------
// HBM-bound: algorithmic bytes = rows * cols * (x + res + y) * sizeof(dtype).
// Fast path: 16 bytes per lane per access (8 bf16 / 4 f32), up to NCH chunks per lane in registers
// (cols <= 2048 bf16 / 1024 f32; 768 = 1.5 / 3 chunks per lane).  Other shapes: scalar fallback.
constexpr int CPL = 32;  // scalar fallback: cached values per lane
constexpr int NCH = 4;
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
        for (int i = 0; i < 4; ++i) w[i] = (unsigned)f2bf(f[2 * i]) | ((unsigned)f2bf(f[2 * i + 1]) << 16);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = __float_as_uint(f[i]);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <bool BF>
__global__ __launch_bounds__(256) void add_ln_fwd_vec_kernel(const void* x, const void* res, const float* gamma,
                                                             const float* beta, float eps, int64_t rows, int cols,
                                                             void* y, float* mean, float* rstd) {
    constexpr int V = Vec16<BF>::N;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = cols / V;
    const char* xr = reinterpret_cast<const char*>(x) + row * cols * (BF ? 2 : 4);
    const char* rr = res ? reinterpret_cast<const char*>(res) + row * cols * (BF ? 2 : 4) : nullptr;
    char* yr = reinterpret_cast<char*>(y) + row * cols * (BF ? 2 : 4);
    float v[NCH][V];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            unpack16<BF>(*reinterpret_cast<const uint4*>(xr + (size_t)c * 16), v[i]);
            if (rr) {
                float r[V];
                unpack16<BF>(*reinterpret_cast<const uint4*>(rr + (size_t)c * 16), r);
#pragma unroll
                for (int j = 0; j < V; ++j) v[i][j] += r[j];
            }
#pragma unroll
            for (int j = 0; j < V; ++j) s += v[i][j];
        }
    }
    const float mu = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (lane + 64 * i < nchunk) {
#pragma unroll
            for (int j = 0; j < V; ++j) { const float d = v[i][j] - mu; q += d * d; }
        }
    const float rs = rsqrtf(wave_sum(q) / (float)cols + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            float o[V];
#pragma unroll
            for (int j = 0; j < V; ++j) o[j] = (v[i][j] - mu) * rs * gamma[c * V + j] + beta[c * V + j];
            *reinterpret_cast<uint4*>(yr + (size_t)c * 16) = pack16<BF>(o);
        }
    }
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
}

// scalar fallback (any cols)
template <bool BF>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const void* x, const void* res, const float* gamma,
                                                         const float* beta, float eps, int64_t rows, int cols,
                                                         void* y, float* mean, float* rstd) {
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
            v[i] = c < cols ? ld<BF>(x, base + c) + (res ? ld<BF>(res, base + c) : 0.f) : 0.f;
            s += v[i];
        }
    } else {
        for (int c = lane; c < cols; c += 64) s += ld<BF>(x, base + c) + (res ? ld<BF>(res, base + c) : 0.f);
    }
    const float mu = wave_sum(s) / (float)cols;
    float q = 0.f;
    if (cached) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) { const int c = lane + 64 * i; const float d = c < cols ? v[i] - mu : 0.f; q += d * d; }
    } else {
        for (int c = lane; c < cols; c += 64) { const float d = ld<BF>(x, base + c) + (res ? ld<BF>(res, base + c) : 0.f) - mu; q += d * d; }
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
            const float z = ld<BF>(x, base + c) + (res ? ld<BF>(res, base + c) : 0.f);
            st<BF>(y, base + c, (z - mu) * rs * gamma[c] + beta[c]);
        }
    }
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
}

// ---- backward: dz, and per-block partial column sums of dgamma / dbeta ----
constexpr int BWD_ROWS_PER_WAVE = 8;
template <bool BF>
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(const void* dy, const void* x, const void* res,
                                                         const float* gamma, const float* mean, const float* rstd,
                                                         int64_t rows, int cols, void* dz, float* ws, int nblk) {
    extern __shared__ float sm[];  // [4 waves][2][cols]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* pg = sm + (wave * 2) * cols;
    float* pb = pg + cols;
    for (int c = lane; c < cols; c += 64) { pg[c] = 0.f; pb[c] = 0.f; }
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * BWD_ROWS_PER_WAVE;
    for (int r = 0; r < BWD_ROWS_PER_WAVE; ++r) {
        const int64_t row = row0 + r;
        if (row >= rows) break;
        const int64_t base = row * cols;
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < cols; c += 64) {
            const float z = ld<BF>(x, base + c) + (res ? ld<BF>(res, base + c) : 0.f);
            const float xh = (z - mu) * rs, g = ld<BF>(dy, base + c) * gamma[c];
            s1 += g; s2 += g * xh;
        }
        s1 = wave_sum(s1) / (float)cols; s2 = wave_sum(s2) / (float)cols;
        for (int c = lane; c < cols; c += 64) {
            const float z = ld<BF>(x, base + c) + (res ? ld<BF>(res, base + c) : 0.f);
            const float xh = (z - mu) * rs, d = ld<BF>(dy, base + c);
            st<BF>(dz, base + c, rs * (d * gamma[c] - s1 - xh * s2));
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
__global__ void add_ln_bwd_reduce_kernel(const float* ws, int nblk, int cols, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float a = 0.f, b = 0.f;
    for (int i = 0; i < nblk; ++i) { a += ws[(int64_t)i * cols + c]; b += ws[((int64_t)nblk + i) * cols + c]; }
    dgamma[c] = a; dbeta[c] = b;
}

// ---- attention forward, fp32 math, one lane per query row --------------------------------------
// A wave owns 64 consecutive queries of one (batch, head); K and V rows are the same for every lane,
// so hipcc fetches them through the scalar cache (s_load) and the products are v_fmac with an SGPR
// operand: no LDS, no barriers.  q and the running output stay in registers (2 x 64 VGPRs), keys are
// consumed in chunks of 8 with an online softmax.  FLOPs = 4 * B * H * Lq * Lk * 64.
constexpr int DH = 64, KC = 8;
template <bool BF>
__global__ __launch_bounds__(64) void attention_fwd_kernel(const void* __restrict__ q, const void* __restrict__ k,
                                                           const void* __restrict__ v, const float* __restrict__ mask,
                                                           int mask_mode, int causal, int B, int H, int Lq, int Lk,
                                                           float scale, void* __restrict__ out, float* __restrict__ lse) {
    const int qblocks = (Lq + 63) / 64;
    const int bid = blockIdx.x;
    const int qb = bid % qblocks, h = (bid / qblocks) % H, b = bid / (qblocks * H);
    const int i = qb * 64 + threadIdx.x;
    const bool live = i < Lq;
    const int ii = live ? i : Lq - 1;
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
            const int64_t koff = (((int64_t)b * Lk + jj) * H + h) * DH;   // wave-uniform
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
            const float pj = (j < Lk && j <= jmax_row) ? __expf(s[c] - mn) : 0.f;
            l += pj;
            const int64_t voff = (((int64_t)b * Lk + jj) * H + h) * DH;
#pragma unroll
            for (int d = 0; d < DH; ++d) o[d] = __builtin_fmaf(pj, ld<BF>(v, voff + d), o[d]);
        }
        m = mn;
    }
    if (live && lse) lse[((int64_t)b * H + h) * Lq + i] = m + __logf(l);
    if (live) {
        const float inv = 1.0f / l;
        const int64_t ooff = ((int64_t)b * Lq + i) * H * DH + (int64_t)h * DH;
#pragma unroll
        for (int d = 0; d < DH; ++d) st<BF>(out, ooff + d, o[d] * inv);
    }
}

// ---- attention forward on the matrix cores (bf16 storage, fp32 accumulate) -----------------------
// Workgroup = 4 waves = 128 queries of one (batch, head); a wave owns 32 queries and walks the keys in
// tiles of 32.  The score tile is computed TRANSPOSED, S^T = K Q^T with v_mfma_f32_32x32x16_bf16
// (A = K rows from LDS, B = Q fragments held in registers), so a lane owns ONE query (col = lane & 31)
// and its 16 accumulator registers are 16 of the tile's 32 keys: the softmax row statistics are
// lane-local plus one exchange with lane ^ 32.  The probabilities then feed the second product
// without leaving registers: O^T += V^T P^T, where registers 8s..8s+7 of the S^T accumulator,
// converted to bf16, ARE the B fragment of k-step s (cdna guide section 3, "an accumulator tile as
// the next MFMA's operand"), and the matching V^T fragments -- keys 16s + 8(j>>2) + 4h + (j&3) for
// element j of lane half h -- are read column-major from the row-major V tile with
// ds_read_b64_tr_b16.  FLOPs = 4 B H Lq Lk 64; K/V tiles are staged through LDS by all 256 threads.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ unsigned pack2bf(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }

__global__ __launch_bounds__(256) void attention_fwd_mfma_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                 const bf16_t* __restrict__ v, const float* __restrict__ mask,
                                                                 int mask_mode, int causal, int B, int H, int Lq, int Lk,
                                                                 float scale, bf16_t* __restrict__ out, float* __restrict__ lse) {
    __shared__ __attribute__((aligned(16))) char lds[4096 + 4096 + 128];
    char* ldsK = lds; char* ldsV = lds + 4096; float* ldsM = reinterpret_cast<float*>(lds + 8192);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int nqb = (Lq + 127) / 128;
    const int qb = blockIdx.x % nqb, h = (blockIdx.x / nqb) % H, b = blockIdx.x / (nqb * H);
    const int qidx = qb * 128 + wave * 32 + r;              // this lane's query
    const int qc = qidx < Lq ? qidx : Lq - 1;
    // Q fragments: B operand, B[k = 8hh + j][col r] = Q[query r][d = 16 s + 8 hh + j]
    bf16x8 qf[4];
    {
        const bf16_t* qp = q + (((int64_t)b * Lq + qc) * H + h) * 64 + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    }
    f32x16 o0, o1;
#pragma unroll
    for (int t = 0; t < 16; ++t) { o0[t] = 0.f; o1[t] = 0.f; }
    float m = -__builtin_inff(), lsum = 0.f;
    const int off = Lk - Lq;
    int nkb = (Lk + 31) / 32;
    if (causal) {  // last key any query of this workgroup can see
        const int lastq = min(Lq - 1, qb * 128 + 127);
        nkb = min(nkb, (lastq + off) / 32 + 1);
    }
    const int srow = tid >> 3, schunk = tid & 7;
    // transposed-read addressing of the V tile (see header): 16-lane group g, lane 4qq+pp of the group
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const unsigned ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;  // LDS offset of the tile buffers
    const unsigned vtr = ldsbase + (unsigned)(4096 + (4 * (g >> 1) + qq) * 128 + (16 * (g & 1) + 4 * pp) * 2);
    const unsigned ka = (unsigned)(r * 128);
    const int kswz = (r >> 1) & 7;

    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();
        {
            const int key = min(kb * 32 + srow, Lk - 1);
            const int64_t go = (((int64_t)b * Lk + key) * H + h) * 64 + schunk * 8;
            const uint4 kr = *reinterpret_cast<const uint4*>(k + go);
            const uint4 vr = *reinterpret_cast<const uint4*>(v + go);
            *reinterpret_cast<uint4*>(ldsK + srow * 128 + ((schunk ^ ((srow >> 1) & 7)) << 4)) = kr;
            *reinterpret_cast<uint4*>(ldsV + srow * 128 + schunk * 16) = vr;
            if (mask_mode == TRX_NN_MASK_KEY && tid < 32) ldsM[tid] = (kb * 32 + tid < Lk) ? mask[(int64_t)b * Lk + kb * 32 + tid] : 0.f;
        }
        __syncthreads();
        // ---- S^T = K Q^T ----
        f32x16 st;
#pragma unroll
        for (int t = 0; t < 16; ++t) st[t] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(ldsK + ka + (((2 * s + hh) ^ kswz) << 4));
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[s], st, 0, 0, 0);
        }
        // ---- scale, mask, visibility; lane-local row maximum ----
        float p[16];
        float mb = -__builtin_inff();
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int kr_ = (t & 3) + 8 * (t >> 2) + 4 * hh;      // key row inside the tile
            const int key = kb * 32 + kr_;
            float val = st[t] * scale;
            if (mask_mode == TRX_NN_MASK_KEY) val += ldsM[kr_];
            else if (mask_mode == TRX_NN_MASK_FULL) val += mask[((int64_t)b * Lq + qc) * Lk + min(key, Lk - 1)];
            const bool hidden = key >= Lk || (causal && key > qidx + off);
            p[t] = hidden ? -__builtin_inff() : val;
            mb = fmaxf(mb, p[t]);
        }
        mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
        const float mn = fmaxf(m, mb);
        const float alpha = (m == mn) ? 1.0f : __expf(m - mn);
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            p[t] = (p[t] == -__builtin_inff()) ? 0.f : __expf(p[t] - mn);
            ps += p[t];
        }
        lsum = lsum * alpha + ps;
        m = mn;
#pragma unroll
        for (int t = 0; t < 16; ++t) { o0[t] *= alpha; o1[t] *= alpha; }
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 pw;
            pw.x = pack2bf(p[8 * s + 0], p[8 * s + 1]); pw.y = pack2bf(p[8 * s + 2], p[8 * s + 3]);
            pw.z = pack2bf(p[8 * s + 4], p[8 * s + 5]); pw.w = pack2bf(p[8 * s + 6], p[8 * s + 7]);
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                uint2 lo, hi;   // keys 16s + 4hh + 0..3 and 16s + 8 + 4hh + 0..3, column d = 32 db + r
                const unsigned a0 = vtr + (unsigned)(16 * s * 128 + 64 * db);
                asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(lo), "=&v"(hi) : "v"(a0) : "memory");
                uint4 vw; vw.x = lo.x; vw.y = lo.y; vw.z = hi.x; vw.w = hi.y;
                const bf16x8 vf = __builtin_bit_cast(bf16x8, vw);
                if (db == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o0, 0, 0, 0);
                else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o1, 0, 0, 0);
            }
        }
    }
    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    if (qidx < Lq) {
        const float inv = 1.0f / ltot;
        if (lse && hh == 0) lse[((int64_t)b * H + h) * Lq + qidx] = m + __logf(ltot);
        bf16_t* op = out + ((int64_t)b * Lq + qidx) * H * 64 + (int64_t)h * 64;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {   // registers 4gq..4gq+3 = d rows 8gq + 4hh + 0..3
            uint2 w0, w1;
            w0.x = pack2bf(o0[4 * gq] * inv, o0[4 * gq + 1] * inv); w0.y = pack2bf(o0[4 * gq + 2] * inv, o0[4 * gq + 3] * inv);
            w1.x = pack2bf(o1[4 * gq] * inv, o1[4 * gq + 1] * inv); w1.y = pack2bf(o1[4 * gq + 2] * inv, o1[4 * gq + 3] * inv);
            *reinterpret_cast<uint2*>(op + 8 * gq + 4 * hh) = w0;
            *reinterpret_cast<uint2*>(op + 32 + 8 * gq + 4 * hh) = w1;
        }
    }
}

// ---- attention backward, fp32 math, probabilities recomputed from lse --------------------------
// pass 1 (a lane per query row i):  delta_i = dO_i . O_i ;  dS_ij = p_ij (dO_i . V_j - delta_i) ;
//                                   dQ_i = scale * sum_j dS_ij K_j
// pass 2 (a lane per key row j):    dV_j = sum_i p_ij dO_i ;  dK_j = scale * sum_i dS_ij Q_i
// In both passes the "other" operand row (K_j, V_j / Q_i, dO_i) is wave-uniform -> scalar loads.
template <bool BF>
__global__ __launch_bounds__(64) void attention_bwd_dq_kernel(const void* __restrict__ q, const void* __restrict__ k,
                                                              const void* __restrict__ v, const float* __restrict__ mask,
                                                              int mask_mode, int causal, int B, int H, int Lq, int Lk,
                                                              float scale, const void* __restrict__ o, const void* __restrict__ dout,
                                                              const float* __restrict__ lse, void* __restrict__ dq) {
    const int qblocks = (Lq + 63) / 64;
    const int bid = blockIdx.x;
    const int qb = bid % qblocks, h = (bid / qblocks) % H, b = bid / (qblocks * H);
    const int i = qb * 64 + threadIdx.x;
    const bool live = i < Lq;
    const int ii = live ? i : Lq - 1;
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
        const float ds = p * (dp - delta);
#pragma unroll
        for (int d = 0; d < DH; ++d) acc[d] = __builtin_fmaf(ds, ld<BF>(k, koff + d), acc[d]);
    }
    if (live) {
#pragma unroll
        for (int d = 0; d < DH; ++d) st<BF>(dq, qoff + d, acc[d] * scale);
    }
}

template <bool BF>
__global__ __launch_bounds__(64) void attention_bwd_dkv_kernel(const void* __restrict__ q, const void* __restrict__ k,
                                                               const void* __restrict__ v, const float* __restrict__ mask,
                                                               int mask_mode, int causal, int B, int H, int Lq, int Lk,
                                                               float scale, const void* __restrict__ o, const void* __restrict__ dout,
                                                               const float* __restrict__ lse, void* __restrict__ dk, void* __restrict__ dv) {
    const int kblocks = (Lk + 63) / 64;
    const int bid = blockIdx.x;
    const int kb = bid % kblocks, h = (bid / kblocks) % H, b = bid / (kblocks * H);
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
        const float ds = p * (dp - delta);
#pragma unroll
        for (int d = 0; d < DH; ++d) {
            av[d] = __builtin_fmaf(p, ld<BF>(dout, ooff + d), av[d]);
            ak[d] = __builtin_fmaf(ds, ld<BF>(q, qoff + d), ak[d]);
        }
    }
    if (live) {
#pragma unroll
        for (int d = 0; d < DH; ++d) { st<BF>(dk, koff + d, ak[d] * scale); st<BF>(dv, koff + d, av[d]); }
    }
}

}  // namespace

extern "C" {

const char* trx_nn_last_error(void) { return g_err.c_str(); }
const char* trx_nn_version(void) { return "trxnn 0.1 (gfx950)"; }
int trx_add_layernorm_bwd_blocks(int64_t rows) { return (int)((rows + 4 * BWD_ROWS_PER_WAVE - 1) / (4 * BWD_ROWS_PER_WAVE)); }

int trx_add_layernorm_fwd(const void* x, const void* res, const float* gamma, const float* beta, float eps,
                          int64_t rows, int cols, int dtype, void* y, float* mean, float* rstd, void* stream) {
    if (!x || !gamma || !beta || !y || rows < 0 || cols <= 0) return fail(TRX_NN_EINVAL, "add_layernorm_fwd: bad argument");
    if (dtype != TRX_NN_F32 && dtype != TRX_NN_BF16) return fail(TRX_NN_EINVAL, "unknown dtype");
    if (rows == 0) return TRX_NN_OK;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int V = dtype == TRX_NN_BF16 ? 8 : 4;
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res)) & 15) == 0;
    const bool vec = aligned && cols % V == 0 && cols <= 64 * NCH * V;
    if (vec) {
        if (dtype == TRX_NN_BF16) hipLaunchKernelGGL(add_ln_fwd_vec_kernel<true>, grid, block, 0, st, x, res, gamma, beta, eps, rows, cols, y, mean, rstd);
        else hipLaunchKernelGGL(add_ln_fwd_vec_kernel<false>, grid, block, 0, st, x, res, gamma, beta, eps, rows, cols, y, mean, rstd);
    } else {
        if (dtype == TRX_NN_BF16) hipLaunchKernelGGL(add_ln_fwd_kernel<true>, grid, block, 0, st, x, res, gamma, beta, eps, rows, cols, y, mean, rstd);
        else hipLaunchKernelGGL(add_ln_fwd_kernel<false>, grid, block, 0, st, x, res, gamma, beta, eps, rows, cols, y, mean, rstd);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRX_NN_OK : fail(TRX_NN_EHIP, hipGetErrorString(e));
}

int trx_add_layernorm_bwd(const void* dy, const void* x, const void* res, const float* gamma, const float* mean,
                          const float* rstd, int64_t rows, int cols, int dtype, void* dz, float* dgamma,
                          float* dbeta, float* ws, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dz || !dgamma || !dbeta || !ws || rows <= 0 || cols <= 0)
        return fail(TRX_NN_EINVAL, "add_layernorm_bwd: bad argument");
    if (dtype != TRX_NN_F32 && dtype != TRX_NN_BF16) return fail(TRX_NN_EINVAL, "unknown dtype");
    if ((size_t)cols * 8 * sizeof(float) > 160 * 1024) return fail(TRX_NN_EINVAL, "cols too large for the LDS partials");
    const int nblk = trx_add_layernorm_bwd_blocks(rows);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)cols * 8 * sizeof(float);
    if (dtype == TRX_NN_BF16) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)add_ln_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(add_ln_bwd_kernel<true>, dim3(nblk), dim3(256), lds, st, dy, x, res, gamma, mean, rstd, rows, cols, dz, ws, nblk);
    } else {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)add_ln_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(add_ln_bwd_kernel<false>, dim3(nblk), dim3(256), lds, st, dy, x, res, gamma, mean, rstd, rows, cols, dz, ws, nblk);
    }
    hipLaunchKernelGGL(add_ln_bwd_reduce_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, ws, nblk, cols, dgamma, dbeta);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRX_NN_OK : fail(TRX_NN_EHIP, hipGetErrorString(e));
}

int trx_attention_fwd_lse(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                          int B, int H, int Lq, int Lk, float scale, int dtype, void* out, float* lse, void* stream) {
    if (!q || !k || !v || !out || B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return fail(TRX_NN_EINVAL, "attention_fwd: bad argument");
    if (mask_mode != TRX_NN_MASK_NONE && !mask) return fail(TRX_NN_EINVAL, "attention_fwd: mask is null");
    if (mask_mode < 0 || mask_mode > 2) return fail(TRX_NN_EINVAL, "attention_fwd: unknown mask mode");
    if (dtype != TRX_NN_F32 && dtype != TRX_NN_BF16) return fail(TRX_NN_EINVAL, "unknown dtype");
    const int qblocks = (Lq + 63) / 64;
    dim3 grid((unsigned)((int64_t)B * H * qblocks)), block(64);
    hipStream_t st = (hipStream_t)stream;
    static const bool force_valu = getenv("TRX_NN_ATTN_VALU") != nullptr;
    if (dtype == TRX_NN_BF16 && !force_valu) {
        dim3 g2((unsigned)((int64_t)B * H * ((Lq + 127) / 128))), b2(256);
        hipLaunchKernelGGL(attention_fwd_mfma_kernel, g2, b2, 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, mask,
                           mask_mode, causal, B, H, Lq, Lk, scale, (bf16_t*)out, lse);
    } else if (dtype == TRX_NN_BF16) hipLaunchKernelGGL(attention_fwd_kernel<true>, grid, block, 0, st, q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, scale, out, lse);
    else hipLaunchKernelGGL(attention_fwd_kernel<false>, grid, block, 0, st, q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, scale, out, lse);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRX_NN_OK : fail(TRX_NN_EHIP, hipGetErrorString(e));
}

int trx_attention_fwd(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                      int B, int H, int Lq, int Lk, float scale, int dtype, void* out, void* stream) {
    return trx_attention_fwd_lse(q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, scale, dtype, out, nullptr, stream);
}

int trx_attention_bwd(const void* q, const void* k, const void* v, const float* mask, int mask_mode, int causal,
                      int B, int H, int Lq, int Lk, float scale, int dtype, const void* out, const void* dout,
                      const float* lse, void* dq, void* dk, void* dv, void* stream) {
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0)
        return fail(TRX_NN_EINVAL, "attention_bwd: bad argument");
    if (mask_mode != TRX_NN_MASK_NONE && !mask) return fail(TRX_NN_EINVAL, "attention_bwd: mask is null");
    if (mask_mode < 0 || mask_mode > 2) return fail(TRX_NN_EINVAL, "attention_bwd: unknown mask mode");
    if (dtype != TRX_NN_F32 && dtype != TRX_NN_BF16) return fail(TRX_NN_EINVAL, "unknown dtype");
    hipStream_t st = (hipStream_t)stream;
    dim3 gq((unsigned)((int64_t)B * H * ((Lq + 63) / 64))), gk((unsigned)((int64_t)B * H * ((Lk + 63) / 64))), block(64);
    if (dtype == TRX_NN_BF16) {
        hipLaunchKernelGGL(attention_bwd_dq_kernel<true>, gq, block, 0, st, q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, scale, out, dout, lse, dq);
        hipLaunchKernelGGL(attention_bwd_dkv_kernel<true>, gk, block, 0, st, q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, scale, out, dout, lse, dk, dv);
    } else {
        hipLaunchKernelGGL(attention_bwd_dq_kernel<false>, gq, block, 0, st, q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, scale, out, dout, lse, dq);
        hipLaunchKernelGGL(attention_bwd_dkv_kernel<false>, gk, block, 0, st, q, k, v, mask, mask_mode, causal, B, H, Lq, Lk, scale, out, dout, lse, dk, dv);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRX_NN_OK : fail(TRX_NN_EHIP, hipGetErrorString(e));
}

}  // extern "C"
