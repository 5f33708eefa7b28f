// knn_prep.hip -- layout kernels around the scan: classify input rows, build the bf16 GEMM
// operand (plain or 3-term split), norms / L2 bias.  All HBM-streaming, 16 bytes per lane.
//
// Replaces faiss IndexFlat::add (retrieve/retrieve_faiss.py:66: memcpy of N x d fp32 into the
// index) and the fvec_norms_L2sqr pre-pass of exhaustive_L2sqr_blas.
//
// Operand layouts, row stride Kp = round_up(K, 64), pad columns zero:
//   plain (every value representable in bf16):         [ v(d) ]                     K = d
//   split (fp32 inputs):  corpus row  [ hi(d) | lo(d) | hi(d) ]                     K = 3d
//                         query row   [ hi(d) | hi(d) | lo(d) ]
//     hi = bf16(v), lo = bf16(v - hi):  x.y ~= xh.yh + xh.yl + xl.yh, dropped terms <= 3.1 * 2^-18
//     per component (DESIGN.md "Exactness"); the MFMA sees one contraction of length 3d.
#include "knn_common.h"
#include <algorithm>

namespace trx {

struct RowStats {          // device-side accumulators (zeroed before each pass)
    u32 inexact_any;       // some value is not representable in bf16
    u32 nonint_any;        // some value is not an integer
    u32 maxabs_bits;       // max |v| as float bits (non-negative floats order like uints)
    u32 maxnorm2_bits;     // max over rows of |row|^2
    u32 nonfp4_any;        // some |value| is not one of 0, 1, 2, 3, 4, 6 (the integers E2M1 holds: the fp4 form of the scan)
    u32 maxerr2_bits;      // max over rows of |row - bf16(row)|^2 (fp32 rows; knn_common.h: round_term)
    u32 hostile_any;       // some row is hostile (knn_common.h: hostile_norm2); such rows are in none of the maxima above
};

template <bool BF>
__device__ __forceinline__ float load_val(const void* p, int64_t i) {
    if (BF) return bf16_to_f32(reinterpret_cast<const bf16_t*>(p)[i]);
    return reinterpret_cast<const float*>(p)[i];
}

// NV consecutive components [c0, c0 + NV) of a row as floats, zero beyond d: 16-byte loads when the group lies inside the row
// and the rows allow them (`vec`: base and row stride multiples of 16 bytes; c0 is a multiple of 8), else element by element.
// (Round 5: the operand builders loaded element by element always -- 8 to 32 two-byte loads per thread, each behind its own
// s_waitcnt vmcnt(0).)
template <bool BF, int NV>
__device__ __forceinline__ void load_group(const char* row, int c0, int d, bool vec, float* v) {
    if (vec && c0 + NV <= d) {
        if (BF) {
#pragma unroll
            for (int j = 0; j < NV / 8; ++j) {
                const uint4 u = *reinterpret_cast<const uint4*>(row + (size_t)(c0 + 8 * j) * 2);
                const u32 w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[8 * j + 2 * i] = __uint_as_float(w[i] << 16); v[8 * j + 2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NV / 4; ++j) {
                const uint4 u = *reinterpret_cast<const uint4*>(row + (size_t)(c0 + 4 * j) * 4);
                v[4 * j] = __uint_as_float(u.x); v[4 * j + 1] = __uint_as_float(u.y); v[4 * j + 2] = __uint_as_float(u.z); v[4 * j + 3] = __uint_as_float(u.w);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = c0 + i < d ? load_val<BF>(row, c0 + i) : 0.f;
    }
}
__device__ __forceinline__ bool rows_allow_16_byte_loads(const void* x, int64_t ld, bool bf) {
    return (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (ld * (bf ? 2 : 4)) % 16 == 0;
}

// one wave per row (grid-stride): stats + fp32 |row|^2.  16 bytes per lane per load when the rows
// allow it (d and the row stride multiples of 8 bf16 / 4 f32): 4 TB/s class instead of the
// 0.16 TB/s the 2-byte-per-lane form reached (it was 1.3 % of a C1 search).
template <bool BF>
__device__ __forceinline__ void stat_one(float v, float& s, float& maxabs, u32& inexact, u32& nonint, float& e2) {
    s = __builtin_fmaf(v, v, s);
    const float a = fabsf(v);
    maxabs = fmaxf(maxabs, a);
    if (!BF) {
        const float r = bf16_to_f32(f32_to_bf16_rn(v));
        inexact |= (r != v) ? 1u : 0u;
        const float dv = v - r;             // exact (r is within 2^-8 of v)
        e2 = __builtin_fmaf(dv, dv, e2);
    }
    nonint |= ((rintf(v) != v) ? 1u : 0u) | ((a > 4.f && a != 6.f) ? 2u : 0u);      // bit 1: beyond the integers of E2M1
}
template <bool BF>
__global__ __launch_bounds__(256) void row_stats_kernel(const void* x, int64_t n, int d, int64_t ld,
                                                        RowStats* st, float* norm2, float* err2) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    constexpr int VEC = BF ? 8 : 4;
    const bool vec = (d % VEC == 0) && (ld % VEC == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    u32 inexact = 0, nonint = 0, hostile = 0;
    float maxabs = 0.f, maxn2 = 0.f, maxe2 = 0.f;
    for (int64_t r = wave0; r < n; r += nwaves) {
        const char* row = reinterpret_cast<const char*>(x) + r * ld * (BF ? 2 : 4);
        float s = 0.f, e2 = 0.f, ra = 0.f;      // (ra: this row's max |v|, folded into the launch's only if the row is benign)
        u32 r_inexact = 0, r_nonint = 0;
        if (vec) {
            for (int c = lane * VEC; c < d; c += 64 * VEC) {
                const uint4 u = *reinterpret_cast<const uint4*>(row + (size_t)c * (BF ? 2 : 4));
                const u32 w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (BF) {
                        stat_one<BF>(__uint_as_float(w[i] << 16), s, ra, r_inexact, r_nonint, e2);
                        stat_one<BF>(__uint_as_float(w[i] & 0xffff0000u), s, ra, r_inexact, r_nonint, e2);
                    } else {
                        stat_one<BF>(__uint_as_float(w[i]), s, ra, r_inexact, r_nonint, e2);
                    }
                }
            }
        } else {
            for (int i = lane; i < d; i += 64) stat_one<BF>(load_val<BF>(row, i), s, ra, r_inexact, r_nonint, e2);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); e2 += __shfl_xor(e2, o, 64); }
        if (lane == 0 && norm2) norm2[r] = s;
        if (lane == 0 && err2) err2[r] = e2;
        if (hostile_norm2(s)) { hostile = 1u; continue; }      // (wave-uniform: s is the row's sum in every lane) -- in no maximum, no flag
        maxn2 = fmaxf(maxn2, s);
        maxe2 = fmaxf(maxe2, e2);
        maxabs = fmaxf(maxabs, ra); inexact |= r_inexact; nonint |= r_nonint;
    }
    // wave-reduce, then block-reduce through LDS: ONE set of atomics per block (the four counters
    // are single addresses; 65k waves x 2 atomics on them took longer than streaming the data)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        maxabs = fmaxf(maxabs, __shfl_xor(maxabs, o, 64));
        maxn2 = fmaxf(maxn2, __shfl_xor(maxn2, o, 64));
        maxe2 = fmaxf(maxe2, __shfl_xor(maxe2, o, 64));
        inexact |= __shfl_xor(inexact, o, 64);
        nonint |= __shfl_xor(nonint, o, 64);
    }
    __shared__ float sh_abs[4], sh_n2[4], sh_e2[4];
    __shared__ u32 sh_flags[4];
    const int w = threadIdx.x >> 6;
    if (lane == 0) { sh_abs[w] = maxabs; sh_n2[w] = maxn2; sh_e2[w] = maxe2; sh_flags[w] = (inexact ? 1u : 0u) | ((nonint & 1u) ? 2u : 0u) | ((nonint & 3u) ? 4u : 0u) | (hostile ? 8u : 0u); }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = fmaxf(fmaxf(sh_abs[0], sh_abs[1]), fmaxf(sh_abs[2], sh_abs[3]));
        float m = fmaxf(fmaxf(sh_n2[0], sh_n2[1]), fmaxf(sh_n2[2], sh_n2[3]));
        const u32 f = sh_flags[0] | sh_flags[1] | sh_flags[2] | sh_flags[3];
        // An atomic only where it would change the word: the six words are single addresses, and 2,048 blocks x 6 atomics on
        // them were 78 us of a 132 us launch over 65,536 query rows (round 5: 132 -> 55 us, tools/r05/fifteenth.sh).  The words only grow, so a value read before
        // (however stale) that already covers this block's is reason enough to skip.
        const u32 ab = __float_as_uint(a), mb = __float_as_uint(m), eb = __float_as_uint(fmaxf(fmaxf(sh_e2[0], sh_e2[1]), fmaxf(sh_e2[2], sh_e2[3])));
        if ((f & 1u) && !__atomic_load_n(&st->inexact_any, __ATOMIC_RELAXED)) atomicOr(&st->inexact_any, 1u);
        if ((f & 2u) && !__atomic_load_n(&st->nonint_any, __ATOMIC_RELAXED)) atomicOr(&st->nonint_any, 1u);
        if ((f & 4u) && !__atomic_load_n(&st->nonfp4_any, __ATOMIC_RELAXED)) atomicOr(&st->nonfp4_any, 1u);
        if ((f & 8u) && !__atomic_load_n(&st->hostile_any, __ATOMIC_RELAXED)) atomicOr(&st->hostile_any, 1u);
        if (ab > __atomic_load_n(&st->maxabs_bits, __ATOMIC_RELAXED)) atomicMax(&st->maxabs_bits, ab);
        if (mb > __atomic_load_n(&st->maxnorm2_bits, __ATOMIC_RELAXED)) atomicMax(&st->maxnorm2_bits, mb);
        if (eb > __atomic_load_n(&st->maxerr2_bits, __ATOMIC_RELAXED)) atomicMax(&st->maxerr2_bits, eb);
    }
}

// The exact class of a search -- integer inputs small enough that every fp32 partial sum (and the L2 key) is exact, so
// the approximate order IS the exact order and no certificate is needed -- decided on the device from the query
// statistics, so that a search need not read them back (knn_api.hip: the stream-ordered search).
__global__ void classify_kernel(const RowStats* qs, int q_split, int idx_nonint, float idx_maxabs, int Kp, int d, int l2, int idx_nonfp4, int* out) {
    const float qmax = __uint_as_float(qs->maxabs_bits);
    const double prod = (double)Kp * (double)qmax * (double)idx_maxabs;
    const double keymag = l2 ? 2.0 * prod + (double)d * idx_maxabs * idx_maxabs : prod;
    out[0] = (!q_split && !qs->nonint_any && !idx_nonint && qmax <= 256.f && idx_maxabs <= 256.f && keymag < 16777216.0) ? 1 : 0;
    // [1]: the form of the scan (knn_scan.hip, FMT).  1 = int8: the exact class, and both operands fit a signed byte -- the L2
    // scan stages the DOUBLED query.  2 = fp4: every value on both sides is an integer E2M1 holds (0, 1, 2, 3, 4, 6 and their
    // negatives: Morgan bit vectors are 0 / 1) -- products and fp32 sums exact as in the bf16 form, 256 components per K-step
    const int i8 = (out[0] && idx_maxabs <= 127.f && qmax * (l2 ? 2.f : 1.f) <= 127.f) ? 1 : 0;
    out[1] = (out[0] && !idx_nonfp4 && !qs->nonfp4_any) ? 2 : i8;
    // [2]: the keys of the candidates ARE their canonical scores (knn_select.hip re-scores nothing): the exact class, and for L2
    // the query norms |x|^2 (fp32 sums of squares) exact as well -- dist = |x|^2 - key, every term an integer below 2^24
    out[2] = (out[0] && (!l2 || (double)d * (double)qmax * (double)qmax < 16777216.0)) ? 1 : 0;
}
hipError_t launch_classify(const void* qstats, int q_split, int idx_nonint, float idx_maxabs, int Kp, int d, int l2, int idx_nonfp4, int* out, hipStream_t st) {
    hipLaunchKernelGGL(classify_kernel, dim3(1), dim3(1), 0, st, (const RowStats*)qstats, q_split, idx_nonint, idx_maxabs, Kp, d, l2, idx_nonfp4, out);
    return hipGetLastError();
}

// Build GEMM operand rows.  One thread per (row, 8-component group).
// SPLIT: 0 plain, 1 split-corpus [hi|lo|hi], 2 split-query [hi|hi|lo]
template <bool BF, int SPLIT>
__global__ __launch_bounds__(256) void build_operand_kernel(const void* x, int64_t n, int d, int64_t ld,
                                                            bf16_t* out, int Kp) {
    const int groups = (d + 7) / 8;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * groups) return;
    const int64_t r = t / groups;
    const int g = (int)(t - r * groups);
    const char* row = reinterpret_cast<const char*>(x) + r * ld * (BF ? 2 : 4);
    bf16_t* o = out + r * Kp;
    bf16_t hi[8], lo[8];
    const int c0 = g * 8;
    const int cnt = (d - c0) < 8 ? (d - c0) : 8;
    float vals[8];
    load_group<BF, 8>(row, c0, d, rows_allow_16_byte_loads(x, ld, BF), vals);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float v = vals[i];
        hi[i] = f32_to_bf16_rn(v);
        lo[i] = f32_to_bf16_rn(v - bf16_to_f32(hi[i]));
    }
    if ((d & 7) == 0) {
        // 16-byte stores: every segment start (0, d, 2d) + c0 is a multiple of 8 elements
        uint4 H, L;
        H.x = hi[0] | ((u32)hi[1] << 16); H.y = hi[2] | ((u32)hi[3] << 16);
        H.z = hi[4] | ((u32)hi[5] << 16); H.w = hi[6] | ((u32)hi[7] << 16);
        L.x = lo[0] | ((u32)lo[1] << 16); L.y = lo[2] | ((u32)lo[3] << 16);
        L.z = lo[4] | ((u32)lo[5] << 16); L.w = lo[6] | ((u32)lo[7] << 16);
        *reinterpret_cast<uint4*>(o + c0) = H;
        if (SPLIT != 0) {
            *reinterpret_cast<uint4*>(o + d + c0) = (SPLIT == 1) ? L : H;
            *reinterpret_cast<uint4*>(o + 2 * d + c0) = (SPLIT == 1) ? H : L;
        }
    } else {
        for (int i = 0; i < cnt; ++i) {
            o[c0 + i] = hi[i];
            if (SPLIT != 0) {
                o[d + c0 + i] = (SPLIT == 1) ? lo[i] : hi[i];
                o[2 * d + c0 + i] = (SPLIT == 1) ? hi[i] : lo[i];
            }
        }
    }
}

// The listing slack of the approximate mode (knn_api.hip): 2 eps_q in the scan's accumulator units (the L2 form carries key / 2),
// with eps_q computed exactly as the select kernel computes it (knn_select.hip: the certificate's bound), rounded up.
__global__ void slack_kernel(const float* qnorm2, int64_t nq, int64_t q_pad, float eps_rel, float eps_round, float ymax_norm2, const float* qerr2,
                             float yerr2_max, int l2, float* out) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= q_pad) return;
    float r = 0.f;
    if (q < nq) {
        const float xn2 = qnorm2[q];
        const float bq = l2 ? (2.0f * sqrtf(xn2 * ymax_norm2) + ymax_norm2) : sqrtf(xn2 * ymax_norm2);
        const float bx = l2 ? 2.0f * sqrtf(xn2 * ymax_norm2) : sqrtf(xn2 * ymax_norm2);
        const double eps = ((double)eps_rel * (double)bq + round_term(eps_round, bx, xn2, qerr2, q, ymax_norm2, yerr2_max, l2 != 0)) * 1.0001 + 1e-30;
        const double sl = l2 ? eps : 2.0 * eps;        // 2 eps in key units = eps in units of key / 2
        r = (float)sl;
        if ((double)r < sl) r = nextafterf(r, __builtin_inff());
        if (!(r < 3.0e38f)) r = 3.0e38f;              // a non-finite bound: list everything (the certificate fails anyway)
    }
    out[q] = r;
}
hipError_t launch_slack(const float* qnorm2, int64_t nq, int64_t q_pad, float eps_rel, float eps_round, float ymax_norm2, const float* qerr2,
                        float yerr2_max, int l2, float* out, hipStream_t st) {
    if (q_pad <= 0) return hipSuccess;
    hipLaunchKernelGGL(slack_kernel, dim3((unsigned)((q_pad + 255) / 256)), dim3(256), 0, st, qnorm2, nq, q_pad, eps_rel, eps_round, ymax_norm2, qerr2, yerr2_max, l2, out);
    return hipGetLastError();
}

// fp4 (E2M1) operand rows for bit vectors and other tiny counts: component c of row r -> its 4-bit code (sign, two exponent
// bits, one mantissa bit: 0 -> 0, 1 -> 2, 2 -> 4, 3 -> 5, 4 -> 6, 6 -> 7), two per byte, low nibble first, zero beyond d up to
// the row's Kp4 bytes.  One thread per (row, 16-byte group = 32 components).  A value outside the set gives garbage -- the
// launch that would read it is gated off on the device then (classify_kernel, out[1]).
template <bool BF>
__global__ __launch_bounds__(256) void build_operand_fp4_kernel(const void* x, int64_t n, int d, int64_t ld, unsigned char* out, int Kp4) {
    const int groups = Kp4 / 16;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * groups) return;
    const int64_t r = t / groups;
    const int g = (int)(t - r * groups);
    const char* row = reinterpret_cast<const char*>(x) + r * ld * (BF ? 2 : 4);
    u32 w[4] = {0u, 0u, 0u, 0u};
    float vals[32];
    load_group<BF, 32>(row, g * 32, d, rows_allow_16_byte_loads(x, ld, BF), vals);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const float v = vals[i];
        const int a = (int)fminf(fabsf(v), 7.f);
        const u32 mag = a <= 2 ? (u32)(2 * a) : (a == 3 ? 5u : (a == 4 ? 6u : 7u));      // (a == 0 -> 0, 1 -> 2, 2 -> 4)
        const u32 code = mag | (v < 0.f ? 8u : 0u);
        w[i >> 3] |= code << (4 * (i & 7));
    }
    *reinterpret_cast<uint4*>(out + r * Kp4 + g * 16) = make_uint4(w[0], w[1], w[2], w[3]);
}
hipError_t launch_build_operand_fp4(const void* x, int is_bf16, int64_t n, int d, int64_t ld, unsigned char* out, int Kp4, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int64_t total = n * (Kp4 / 16);
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (is_bf16) hipLaunchKernelGGL(build_operand_fp4_kernel<true>, grid, block, 0, st, x, n, d, ld, out, Kp4);
    else hipLaunchKernelGGL(build_operand_fp4_kernel<false>, grid, block, 0, st, x, n, d, ld, out, Kp4);
    return hipGetLastError();
}

// int8 operand rows for the integer class: component c of row r -> (int8)(scale * value) (scale 2: the queries of an L2
// search), zero beyond d up to the row's Kp8 bytes.  One thread per (row, 16-byte group).  A value that is not a small
// integer gives garbage -- the launch that would read it is gated off on the device then (classify_kernel, out[1]).
template <bool BF>
__global__ __launch_bounds__(256) void build_operand_i8_kernel(const void* x, int64_t n, int d, int64_t ld, signed char* out, int Kp8, float scale) {
    const int groups = Kp8 / 16;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * groups) return;
    const int64_t r = t / groups;
    const int g = (int)(t - r * groups);
    const char* row = reinterpret_cast<const char*>(x) + r * ld * (BF ? 2 : 4);
    u32 w[4] = {0u, 0u, 0u, 0u};
    float vals[16];
    load_group<BF, 16>(row, g * 16, d, rows_allow_16_byte_loads(x, ld, BF), vals);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float v = vals[i] * scale;
        const int q = (int)fminf(fmaxf(v, -128.f), 127.f);
        w[i >> 2] |= ((u32)q & 0xffu) << (8 * (i & 3));
    }
    *reinterpret_cast<uint4*>(out + r * Kp8 + g * 16) = make_uint4(w[0], w[1], w[2], w[3]);
}
hipError_t launch_build_operand_i8(const void* x, int is_bf16, int64_t n, int d, int64_t ld, signed char* out, int Kp8, float scale, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int64_t total = n * (Kp8 / 16);
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (is_bf16) hipLaunchKernelGGL(build_operand_i8_kernel<true>, grid, block, 0, st, x, n, d, ld, out, Kp8, scale);
    else hipLaunchKernelGGL(build_operand_i8_kernel<false>, grid, block, 0, st, x, n, d, ld, out, Kp8, scale);
    return hipGetLastError();
}
// int8 form, L2: the accumulators of a tile start from -|y|^2 (an integer below 2^24 in the exact class); pad rows from a
// value no sum can lift into any list
__global__ void fill_bias_i32_kernel(const float* norm2, int64_t n, int64_t n_pad, int* bias) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pad) return;
    bias[j] = j < n ? -(int)fminf(norm2[j], 1073741824.f) : -(1 << 30);
}
hipError_t launch_fill_bias_i32(const float* norm2, int64_t n, int64_t n_pad, int* bias, hipStream_t st) {
    if (n_pad <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_bias_i32_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, st, norm2, n, n_pad, bias);
    return hipGetLastError();
}

// bias[j] = -norm2[j] for j < n, -inf for n <= j < n_pad
__global__ void fill_bias_kernel(const float* norm2, int64_t n, int64_t n_pad, float* bias) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pad) return;
    bias[j] = j < n ? -0.5f * norm2[j] : -__builtin_inff();     // accumulator start of the L2 scan: x.y - |y|^2 / 2 = key / 2
}

// widen bf16 rows to f32 rows (plain -> split transition keeps exact values as f32)
__global__ void widen_rows_kernel(const bf16_t* in, int64_t n, int d, int64_t ld_in, float* out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * d) return;
    const int64_t r = t / d;
    const int c = (int)(t - r * d);
    out[t] = bf16_to_f32(in[r * ld_in + c]);
}

hipError_t launch_row_stats(const void* x, int is_bf16, int64_t n, int d, int64_t ld, void* stats_dev,
                            float* norm2, float* err2, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    if (is_bf16)
        hipLaunchKernelGGL(row_stats_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, x, n, d, ld,
                           reinterpret_cast<RowStats*>(stats_dev), norm2, err2);
    else
        hipLaunchKernelGGL(row_stats_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, x, n, d, ld,
                           reinterpret_cast<RowStats*>(stats_dev), norm2, err2);
    return hipGetLastError();
}

hipError_t launch_build_operand(const void* x, int is_bf16, int split_kind, int64_t n, int d, int64_t ld,
                                bf16_t* out, int Kp, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int groups = (d + 7) / 8;
    const int64_t total = n * groups;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define TRX_BO(a, b) hipLaunchKernelGGL((build_operand_kernel<a, b>), grid, block, 0, st, x, n, d, ld, out, Kp)
    if (is_bf16) {
        if (split_kind == 0) TRX_BO(true, 0); else if (split_kind == 1) TRX_BO(true, 1); else TRX_BO(true, 2);
    } else {
        if (split_kind == 0) TRX_BO(false, 0); else if (split_kind == 1) TRX_BO(false, 1); else TRX_BO(false, 2);
    }
#undef TRX_BO
    return hipGetLastError();
}

hipError_t launch_fill_bias(const float* norm2, int64_t n, int64_t n_pad, float* bias, hipStream_t st) {
    if (n_pad <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_bias_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, st, norm2, n,
                       n_pad, bias);
    return hipGetLastError();
}

hipError_t launch_widen_rows(const bf16_t* in, int64_t n, int d, int64_t ld_in, float* out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int64_t total = n * d;
    hipLaunchKernelGGL(widen_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, n, d,
                       ld_in, out);
    return hipGetLastError();
}

// int8 host data (TRX_DTYPE_I8: Morgan bit vectors, retrieve_faiss.py:36-44) crosses PCIe as bytes and becomes bf16 here --
// every int8 value is a bf16 number -- so the index and the search see the values a float32 conversion on the host would
// have given them.  The block is contiguous (ld = d): 16 values per thread, a scalar tail.
__global__ __launch_bounds__(256) void widen_i8_kernel(const signed char* in, int64_t total, bf16_t* out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t e0 = t * 16;
    if (e0 >= total) return;
    if (e0 + 16 <= total) {
        const uint4 u = *reinterpret_cast<const uint4*>(in + e0);
        const u32 w[4] = {u.x, u.y, u.z, u.w};
        u32 o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int a = (int)(signed char)((w[i >> 1] >> (16 * (i & 1))) & 0xffu);
            const int b = (int)(signed char)((w[i >> 1] >> (16 * (i & 1) + 8)) & 0xffu);
            o[i] = (u32)f32_to_bf16_rn((float)a) | ((u32)f32_to_bf16_rn((float)b) << 16);
        }
        *reinterpret_cast<uint4*>(out + e0) = make_uint4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint4*>(out + e0 + 8) = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
        for (int64_t e = e0; e < total; ++e) out[e] = f32_to_bf16_rn((float)in[e]);
    }
}
hipError_t launch_widen_i8(const signed char* in, int64_t n, int d, bf16_t* out, hipStream_t st) {
    const int64_t total = n * d;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(widen_i8_kernel, dim3((unsigned)((total + 4095) / 4096)), dim3(256), 0, st, in, total, out);
    return hipGetLastError();
}

// Hostile rows of a block just appended (knn_common.h): the operand row becomes zero, |y|^2 becomes +inf (so that fill_bias makes
// the L2 bias -inf and the scan never lists the row), and the row's id joins the index's special list, from which
// merge_special_kernel folds its canonical score into every result.  One wave per row; the count may run past MAX_SPECIAL
// (the host then sends every search to the exact scan).
__global__ __launch_bounds__(256) void sanitize_hostile_kernel(float* norm2, int64_t n, int64_t id0, bf16_t* op, int Kp, int* special, int* nspecial) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave0; r < n; r += nwaves) {
        if (!hostile_norm2(norm2[r])) continue;
        bf16_t* row = op + r * Kp;
        for (int c = lane * 8; c < Kp; c += 512) *reinterpret_cast<uint4*>(row + c) = make_uint4(0u, 0u, 0u, 0u);      // Kp is a multiple of 128
        if (lane == 0) {
            norm2[r] = __builtin_inff();
            const int pos = atomicAdd(nspecial, 1);
            if (pos < MAX_SPECIAL) special[pos] = (int)(id0 + r);
        }
    }
}
hipError_t launch_sanitize_hostile(float* norm2, int64_t n, int64_t id0, bf16_t* op, int Kp, int* special, int* nspecial, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const unsigned grid = (unsigned)std::min<int64_t>((n + 3) / 4, 2048);
    hipLaunchKernelGGL(sanitize_hostile_kernel, dim3(grid), dim3(256), 0, st, norm2, n, id0, op, Kp, special, nspecial);
    return hipGetLastError();
}

}  // namespace trx
