// tanimoto.hip -- libtrxtani.so (include/trx_tanimoto.h): brute-force Tanimoto similarity of count fingerprints,
// the scoring loop of the reference's retrieve/retrieve.py:18-40,55-62.
//
// Integer, VALU-bound work: sum_i min(a_i, b_i) = (sum a + sum b - sum |a_i - b_i|) / 2, and gfx950 has
// v_sad_u8 (four byte differences accumulated per lane and instruction).  One wave owns a block of 64 corpus
// rows (lane = row); dword j of the block is one coalesced 256-byte load (the packed layout of the header) and is
// used against 16 queries whose dword j arrives through the scalar cache (the queries are stored transposed, so
// the 16 dwords are one 64-byte scalar load) -- 16 v_sad_u8 with an SGPR operand per vector load, 16 accumulators.
// No LDS, no MFMA: there is no product to form.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/trx_tanimoto.h"

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char* msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

constexpr int QG = TRX_TANI_QUERY_GROUP;
constexpr int TILE_C = 256;                  // counts per pack tile (64 dwords)
constexpr int TILE_LD = TILE_C + 4;          // LDS row stride in bytes: +4 spreads the 64 rows over the banks

// ---- pack: [n, ld] counts -> byte magnitudes, 64-row blocks, dword-interleaved -------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const T* __restrict__ fps, int64_t n, int d, int64_t ld, int64_t first_row,
                                                   uint32_t* __restrict__ packed, int32_t* __restrict__ row_sum,
                                                   int32_t* __restrict__ flags) {
    __shared__ __attribute__((aligned(16))) unsigned char tile[64 * TILE_LD];
    const int t = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * TILE_C;
    bool big = false;
    for (int r = 0; r < 64; ++r) {           // coalesced along the row
        const int64_t row = r0 + r;
        const int c = c0 + t;
        long long v = 0;
        if (row < n && c < d) v = (long long)fps[row * ld + c];
        if (v < 0) v = -v;
        if (v > 255) { big = true; v = 255; }
        tile[r * TILE_LD + t] = (unsigned char)v;
    }
    if (__any(big) && (t & 63) == 0) atomicOr(flags, 1);
    __syncthreads();
    const int lane = t & 63, jj = t >> 6;    // lane = row of the block
    const int dw = d >> 2;
    const int64_t blk = (first_row + r0) >> 6;
    int sum = 0;
    for (int j = jj; j < TILE_C / 4; j += 4) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(&tile[lane * TILE_LD + 4 * j]);
        const int jg = c0 / 4 + j;
        if (jg < dw) packed[(blk * dw + jg) * 64 + lane] = w;
        sum += (int)(w & 255u) + (int)((w >> 8) & 255u) + (int)((w >> 16) & 255u) + (int)(w >> 24);
    }
    if (r0 + lane < n && sum) atomicAdd(&row_sum[first_row + r0 + lane], sum);
}

// fp32 stand-in of and / den used only to bound the selection: relative error < 2^-21 (v_rcp_f32 is good to 1 ulp, the
// integers are exact in fp32), the same function in both kernels
__device__ __forceinline__ float approx_sim(int a, int dn) {
    return dn > 0 ? (float)a * __builtin_amdgcn_rcpf((float)dn) : 0.0f;
}

// ---- scores ------------------------------------------------------------------------------------------------------
constexpr int UNROLL = 8;                    // vector loads in flight per wave

// NG groups of 16 queries per wave: 16 * NG accumulators, so the corpus is read once per 64 queries (NG = 4)
template <int NG>
__global__ __launch_bounds__(256) void scores_kernel(const uint32_t* __restrict__ packed, const int32_t* __restrict__ row_sum, int64_t n,
                                                     int dw, const uint32_t* __restrict__ q_t, const int32_t* __restrict__ q_sum, int q0,
                                                     int nq, int nq_pad, int16_t* __restrict__ and_out, int64_t ld_out,
                                                     float* __restrict__ block_max) {
    constexpr int NQ = NG * QG;
    const int lane = threadIdx.x & 63;
    const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk * 64 >= n) return;               // wave-uniform
    const int qbase = q0 + blockIdx.y * NQ;
    const uint32_t* p = packed + blk * dw * 64 + lane;
    const uint32_t* qp = q_t + qbase;        // wave-uniform: the query dwords travel through the scalar cache
    uint32_t acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = 0;
    int j = 0;
    for (; j + UNROLL <= dw; j += UNROLL) {
        uint32_t v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[(int64_t)(j + u) * 64];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t* qq = qp + (int64_t)(j + u) * nq_pad;
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_sad_u8(v[u], qq[q], acc[q]);
        }
    }
    for (; j < dw; ++j) {
        const uint32_t v = p[(int64_t)j * 64];
        const uint32_t* qq = qp + (int64_t)j * nq_pad;
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_sad_u8(v, qq[q], acc[q]);
    }
    const int64_t row = blk * 64 + lane;
    const bool live = row < n;
    const int rs = live ? row_sum[row] : 0;
    const int64_t nblocks = (n + 63) >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int qi = qbase + q;
        if (qi < nq) {                                     // wave-uniform
            const int s = rs + q_sum[qi];
            const int a = (s - (int)acc[q]) >> 1;          // sum of minima (< 32768)
            if (live) and_out[(int64_t)qi * ld_out + row] = (int16_t)a;
            if (block_max) {                               // best APPROXIMATE similarity of this block (selection bound, see the header)
                // similarities are >= 0, so their bit patterns order like unsigned integers: a DPP max-reduction
                // (six VALU instructions, no LDS crossbar) leaves the block maximum in lane 63
                uint32_t m = live ? __float_as_uint(approx_sim(a, s - a)) : 0u;
                m = max(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
                m = max(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
                m = max(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x141, 0xf, 0xf, false));   // row_half_mirror
                m = max(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x140, 0xf, 0xf, false));   // row_mirror
                m = max(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x142, 0xa, 0xf, false));   // row_bcast:15
                m = max(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x143, 0xc, 0xf, false));   // row_bcast:31
                if (lane == 63) block_max[(int64_t)qi * nblocks + blk] = __uint_as_float(m);
            }
        }
    }
}

// exact key of one pair: the double the reference's Python float holds, cut to 36 bits (order-preserving: header)
__device__ __forceinline__ int64_t exact_key(int a, int dn, int64_t row) {
    const double sim = dn > 0 ? (double)a / (double)dn : 0.0;
    const uint64_t fx = (uint64_t)(sim * 68719476735.0);     // 2^36 - 1
    return (int64_t)((fx << TRX_TANI_KEY_ID_BITS) | (uint64_t)row);
}

// exact keys of the pairs whose approximate similarity reaches thr[q] (thr == NULL: of every pair), appended in any
// order (keys are distinct: the row number is part of a key)
__global__ __launch_bounds__(256) void filter_kernel(const int16_t* __restrict__ and_in, int64_t ld, const int32_t* __restrict__ row_sum,
                                                     const int32_t* __restrict__ q_sum, const int32_t* __restrict__ q_ids, int64_t n,
                                                     const float* __restrict__ thr, int64_t cap, int64_t* __restrict__ out,
                                                     int32_t* __restrict__ counts) {
    const int slot = blockIdx.y;                           // position in the output; q_ids maps it to a query (or identity)
    const int q = q_ids ? q_ids[slot] : slot;
    const float t = thr ? thr[q] : -1.0f;
    const int qs = q_sum[q];
    const int16_t* aq = and_in + (int64_t)q * ld;
    const bool vec = ((reinterpret_cast<uintptr_t>(aq) | reinterpret_cast<uintptr_t>(row_sum)) & 15) == 0;
    for (int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i0 < n; i0 += (int64_t)gridDim.x * 2048) {
        int a[8], rs[8];
        if (vec && i0 + 8 <= n) {                          // 8 rows per lane: one 16-byte and two 16-byte loads
            const uint4 av = *reinterpret_cast<const uint4*>(aq + i0);
            const int4 r0 = *reinterpret_cast<const int4*>(row_sum + i0), r1 = *reinterpret_cast<const int4*>(row_sum + i0 + 4);
            const uint32_t aw[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[2 * e] = (int)(aw[e] & 0xffffu); a[2 * e + 1] = (int)(aw[e] >> 16); }
            rs[0] = r0.x; rs[1] = r0.y; rs[2] = r0.z; rs[3] = r0.w; rs[4] = r1.x; rs[5] = r1.y; rs[6] = r1.z; rs[7] = r1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool in = i0 + e < n;
                a[e] = in ? (int)aq[i0 + e] : 0;
                rs[e] = in ? row_sum[i0 + e] : -qs;        // den 0 -> approximate similarity 0; masked below anyway
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int dn = rs[e] + qs - a[e];
            if (i0 + e < n && approx_sim(a[e], dn) >= t) {
                const int pos = atomicAdd(&counts[slot], 1);
                if (pos < cap) out[(int64_t)slot * cap + pos] = exact_key(a[e], dn, i0 + e);
            }
        }
    }
}

template <int NG>
void launch_scores(const void* packed, const int32_t* row_sum, int64_t n, int d, const uint32_t* q_t, const int32_t* q_sum, int q0, int count,
                   int nq, int nq_pad, int16_t* and_out, int64_t ld_out, float* block_max, hipStream_t st) {
    const int64_t blocks = (n + 63) / 64;
    const dim3 grid((unsigned)((blocks + 3) / 4), (unsigned)count);
    hipLaunchKernelGGL(scores_kernel<NG>, grid, dim3(256), 0, st, (const uint32_t*)packed, row_sum, n, d / 4, q_t, q_sum, q0, nq, nq_pad,
                       and_out, ld_out, block_max);
}

}  // namespace

extern "C" {

const char* trx_tanimoto_last_error(void) { return g_err; }

int64_t trx_tanimoto_packed_bytes(int64_t n, int d) {
    if (n < 0 || d <= 0 || d % 4) return -1;
    return ((n + 63) / 64) * 64 * (int64_t)d;
}

int trx_tanimoto_pack(const void* fps, int dtype, int64_t n, int d, int64_t ld, int64_t first_row, void* packed, int32_t* row_sum,
                      int32_t* flags, void* stream) {
    if (n < 0 || d <= 0 || d % 4 || ld < d || first_row < 0 || first_row % 64) return fail(-1, "trx_tanimoto_pack: bad shape (d % 4, ld >= d, first_row % 64)");
    if (n == 0) return 0;
    if (!fps || !packed || !row_sum || !flags) return fail(-1, "trx_tanimoto_pack: null pointer");
    const dim3 grid((unsigned)((n + 63) / 64), (unsigned)((d + TILE_C - 1) / TILE_C));
    hipStream_t st = (hipStream_t)stream;
    uint32_t* out = (uint32_t*)packed;
    switch (dtype) {
        case TRX_TANI_I64: hipLaunchKernelGGL(pack_kernel<long long>, grid, dim3(256), 0, st, (const long long*)fps, n, d, ld, first_row, out, row_sum, flags); break;
        case TRX_TANI_I32: hipLaunchKernelGGL(pack_kernel<int>, grid, dim3(256), 0, st, (const int*)fps, n, d, ld, first_row, out, row_sum, flags); break;
        case TRX_TANI_I8: hipLaunchKernelGGL(pack_kernel<signed char>, grid, dim3(256), 0, st, (const signed char*)fps, n, d, ld, first_row, out, row_sum, flags); break;
        default: return fail(-1, "trx_tanimoto_pack: unknown dtype");
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(-2, hipGetErrorString(e));
}

int trx_tanimoto_scores(const void* packed, const int32_t* row_sum, int64_t n, int d, const uint32_t* q_t, const int32_t* q_sum, int nq,
                        int16_t* and_out, int64_t ld_out, float* block_max, void* stream) {
    if (n < 0 || nq < 0 || d <= 0 || d % 4 || ld_out < n) return fail(-1, "trx_tanimoto_scores: bad shape");
    if (n >= ((int64_t)1 << TRX_TANI_KEY_ID_BITS)) return fail(-1, "trx_tanimoto_scores: n must be < 2^27 (row numbers ride in the keys)");
    if (n == 0 || nq == 0) return 0;
    if (!packed || !row_sum || !q_t || !q_sum || !and_out) return fail(-1, "trx_tanimoto_scores: null pointer");
    const int nq_pad = (nq + QG - 1) / QG * QG;
    hipStream_t st = (hipStream_t)stream;
    const int full = nq_pad / (4 * QG);                   // passes over the corpus with 64 queries per wave
    if (full) launch_scores<4>(packed, row_sum, n, d, q_t, q_sum, 0, full, nq, nq_pad, and_out, ld_out, block_max, st);
    const int q0 = full * 4 * QG, rem = (nq_pad - q0) / QG;
    if (rem == 1) launch_scores<1>(packed, row_sum, n, d, q_t, q_sum, q0, 1, nq, nq_pad, and_out, ld_out, block_max, st);
    else if (rem == 2) launch_scores<2>(packed, row_sum, n, d, q_t, q_sum, q0, 1, nq, nq_pad, and_out, ld_out, block_max, st);
    else if (rem == 3) launch_scores<3>(packed, row_sum, n, d, q_t, q_sum, q0, 1, nq, nq_pad, and_out, ld_out, block_max, st);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(-2, hipGetErrorString(e));
}

int trx_tanimoto_filter(const int16_t* and_in, int64_t ld, const int32_t* row_sum, const int32_t* q_sum, const int32_t* q_ids, int nsel,
                        int64_t n, const float* thr, int64_t cap, int64_t* out, int32_t* counts, void* stream) {
    if (n < 0 || nsel < 0 || ld < n || cap <= 0) return fail(-1, "trx_tanimoto_filter: bad shape");
    if (n >= ((int64_t)1 << TRX_TANI_KEY_ID_BITS)) return fail(-1, "trx_tanimoto_filter: n must be < 2^27 (row numbers ride in the keys)");
    if (n == 0 || nsel == 0) return 0;
    if (!and_in || !row_sum || !q_sum || !out || !counts) return fail(-1, "trx_tanimoto_filter: null pointer");
    const int64_t want = (n + 2047) / 2048;
    const dim3 grid((unsigned)(want < 1024 ? want : 1024), (unsigned)nsel);
    hipLaunchKernelGGL(filter_kernel, grid, dim3(256), 0, (hipStream_t)stream, and_in, ld, row_sum, q_sum, q_ids, n, thr, cap, out, counts);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(-2, hipGetErrorString(e));
}

}  // extern "C"
