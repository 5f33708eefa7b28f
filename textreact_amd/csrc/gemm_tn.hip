// gemm_tn.hip -- C[N, K] = A[M, N]^T . B[M, K] in bf16 with fp32 accumulation on gfx950: the WEIGHT GRADIENT of a
// Linear layer (dW = dY^T X), the product the library runs worst in the predictor's training step: it contracts
// over the M = 16,384 token rows into only N*K / 65,536 = 9 .. 36 output tiles, and hipBLASLt does not split the
// contraction (0.48 PFLOP/s, ~24 % of the step).
//
// Here the contraction IS split: workgroup (tile, s) accumulates rows [s * M/S, (s+1) * M/S) of one 256 x 256 tile
// of C into an fp32 partial, a second kernel sums the S partials in a fixed order (deterministic) and rounds to
// bf16.  Inside a workgroup it is the retrieval scan's loop turned sideways: 8 waves, each 128 (n) x 64 (k) of the
// tile = 4 x 2 tiles of v_mfma_f32_32x32x16_bf16 (128 accumulator registers), steps of 64 rows of A and B staged
// by LDS-DMA into two 64 KiB stages.  Both operands are consumed TRANSPOSED (the contraction index m is the slow
// dimension of both): every fragment is two ds_read_b64_tr_b16 on the row-major [m][n] / [m][k] tile, exactly the
// V^T operand of the attention kernels.  A 32-lane half of such a read touches 4 rows x 64 B; rows are 512 B
// apart, so 16-byte chunk c of row m is stored at slot c ^ (4 * (m & 3)) (applied to the DMA source chunk), which
// spreads the 4 rows over the 4 quarters of a 256-byte bank row.
#include "../../include/trx_nn.h"
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>

namespace trxtn {

__device__ const uint4 g_zero16 = {0u, 0u, 0u, 0u};      // what the rows past M of a ragged last step read as


typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int TILE = 256, BM = 64, THREADS = 512;
constexpr int STAGE = 2 * BM * 512;           // A rows then B rows, 512 B each: 64 KiB
constexpr int LDS_TOTAL = 2 * STAGE;

struct Params {
    const bf16_t* A; const bf16_t* B; float* ws;
    float* ws_colsum;   // [nsplit][N] column sums of A (the bias gradient of the same Linear), or null
    int M, N, K, lda, ldb;
    int tn, tk, nsplit, steps_per_split;
};

__global__ __launch_bounds__(THREADS, 2) void gemm_tn_kernel(Params p) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void gbl_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_n = wave >> 2, wave_k = wave & 3;      // 2 x 4 waves: 128 n x 64 k each
    // Which (split, tile) a workgroup takes (TRX_TN_GROUP, default on).  The tn x tk tiles of ONE split read the same token rows: a
    // row block of A is wanted by the tk workgroups of its tile column, one of B by the tn of its tile row.  Workgroups are
    // placed on the 8 XCDs round-robin (bid & 7), each XCD with its own L2: numbered split-fastest (round 2) the ~32 workgroups of
    // an XCD are 32 different (split, tile) pairs that share nothing, and every operand byte is fetched tk or tn times from the
    // fabric -- 600 MB per FFN weight gradient, 7 TB/s: the kernel ran at the Infinity-Cache's bandwidth, not at its MFMAs'.
    // Numbered split-SLOWEST and dealt to the XCDs in contiguous runs (the scan kernel's bijective remap), an XCD holds the tiles
    // of one or two splits, which walk the same rows in step and find each other's lines in L2.
#ifndef TRX_TN_GROUP
#define TRX_TN_GROUP 1
#endif
    int bid = blockIdx.x;
    int split, kt, nt;
    if (TRX_TN_GROUP) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7;
        const int v = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);      // XCD x owns [base, base + q (+1))
        const int tiles = p.tn * p.tk;
        split = v / tiles;
        const int t = v - split * tiles;
        kt = t % p.tk; nt = t / p.tk;
    } else {
        split = bid % p.nsplit; bid /= p.nsplit;
        kt = bid % p.tk; nt = bid / p.tk;
    }
    const int n0 = nt * TILE, k0 = kt * TILE;
    const int total_steps = (p.M + BM - 1) / BM;          // the last step may be partial: rows >= M read as zeros (A) / row M-1 (B)
    const int step0 = split * p.steps_per_split;
    const int nsteps = max(0, min(p.steps_per_split, total_steps - step0));

    // ---- staging: a piece = one global_load_lds_dwordx4 = 2 rows x 512 B; wave w moves pieces 4w .. 4w+3 of A and of B
    const int prow = lane >> 5, pslot = lane & 31;
    int64_t offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 2 * (4 * wave + i) + prow;            // row of the 64-row step
        const int chunk = pslot ^ (4 * (row & 3));            // source chunk of this lane's slot
        offA[i] = (int64_t)row * p.lda + n0 + chunk * 8;
        offB[i] = (int64_t)row * p.ldb + k0 + chunk * 8;
    }
#define TRX_TN_STAGE(S, BUF)                                                                                     \
    {                                                                                                            \
        const int r0_ = (step0 + (S)) * BM;                                                                      \
        const bf16_t* a_ = p.A + (int64_t)r0_ * p.lda;                                                           \
        const bf16_t* b_ = p.B + (int64_t)r0_ * p.ldb;                                                           \
        char* l_ = smem + (BUF) * STAGE + (4 * wave) * 1024;                                                     \
        if (r0_ + BM <= p.M) {                                                                                   \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                   \
                __builtin_amdgcn_global_load_lds((gbl_void*)(a_ + offA[i_]), (lds_void*)(l_ + i_ * 1024), 16, 0, 0); \
                __builtin_amdgcn_global_load_lds((gbl_void*)(b_ + offB[i_]), (lds_void*)(l_ + BM * 512 + i_ * 1024), 16, 0, 0); \
            }                                                                                                    \
        } else { /* the ragged last step (wave-uniform): a row past M contributes nothing when its A half is zero */ \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                   \
                const int over_ = r0_ + 2 * (4 * wave + i_) + prow - (p.M - 1);      /* > 0: rows past the end */ \
                const bf16_t* pa_ = over_ > 0 ? reinterpret_cast<const bf16_t*>(&g_zero16) : a_ + offA[i_];      \
                const bf16_t* pb_ = b_ + offB[i_] - (over_ > 0 ? (int64_t)over_ * p.ldb : 0);                    \
                __builtin_amdgcn_global_load_lds((gbl_void*)pa_, (lds_void*)(l_ + i_ * 1024), 16, 0, 0);         \
                __builtin_amdgcn_global_load_lds((gbl_void*)pb_, (lds_void*)(l_ + BM * 512 + i_ * 1024), 16, 0, 0); \
            }                                                                                                    \
        }                                                                                                        \
    }

    // ---- transposed fragment addresses: lane -> row 4 hh + qq (+ 8 for the second read, + 16 per sub-step),
    // columns 16 (g & 1) + 4 pp .. + 3 of a 32-column block
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3, hh = lane >> 5;
    const unsigned ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned fa[4], fb[2];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) {
        const int c = 16 * wave_n + 4 * ib + 2 * (g & 1) + (pp >> 1);
        fa[ib] = ldsbase + (unsigned)((4 * hh + qq) * 512 + ((c ^ (4 * qq)) << 4) + 8 * (pp & 1));
    }
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
        const int c = 8 * wave_k + 4 * jb + 2 * (g & 1) + (pp >> 1);
        fb[jb] = ldsbase + (unsigned)(BM * 512 + (4 * hh + qq) * 512 + ((c ^ (4 * qq)) << 4) + 8 * (pp & 1));
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;

    // column sums of A (= db of the Linear whose dW this is) ride along in the workgroups of the first k tile: thread ->
    // 16-byte chunk tid & 31 of rows (tid >> 5) + 16 j, which all share one swizzle
    const bool colsum = p.ws_colsum != nullptr && kt == 0;
    const unsigned csaddr = ldsbase + (unsigned)((tid >> 5) * 512 + (((tid & 31) ^ (4 * ((tid >> 5) & 3))) << 4));
    float cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = 0.f;

    if (nsteps > 0) TRX_TN_STAGE(0, 0);
    __syncthreads();
    int cur = 0;
    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) TRX_TN_STAGE(s + 1, cur ^ 1);
        const unsigned sb = (unsigned)(cur * STAGE);
        u32x4 cv[4];
        if (colsum) {   // older than every fragment read below: the first counted wait retires them
#pragma unroll
            for (int j = 0; j < 4; ++j)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(cv[j]) : "v"(csaddr + sb), "n"(j * 8192) : "memory");
        }
        // 4 sub-steps of 16 rows; fragments of sub-step ss+1 are read while the MFMAs of ss run
        uint2 ra[2][4][2], rb[2][2][2];     // [slot][block][low / high 4 rows]
#define TRX_TN_READ(SLOT, SS)                                                                                    \
    _Pragma("unroll") for (int ib_ = 0; ib_ < 4; ++ib_)                                                          \
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                \
                     : "=&v"(ra[SLOT][ib_][0]), "=&v"(ra[SLOT][ib_][1]) : "v"(fa[ib_] + sb), "n"((SS) * 8192), "n"((SS) * 8192 + 4096) : "memory"); \
    _Pragma("unroll") for (int jb_ = 0; jb_ < 2; ++jb_)                                                          \
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                \
                     : "=&v"(rb[SLOT][jb_][0]), "=&v"(rb[SLOT][jb_][1]) : "v"(fb[jb_] + sb), "n"((SS) * 8192), "n"((SS) * 8192 + 4096) : "memory");
#define TRX_TN_WAIT(SLOT, CNT)                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(" #CNT ")"                                                                   \
                 : "+v"(ra[SLOT][0][0]), "+v"(ra[SLOT][0][1]), "+v"(ra[SLOT][1][0]), "+v"(ra[SLOT][1][1]),       \
                   "+v"(ra[SLOT][2][0]), "+v"(ra[SLOT][2][1]), "+v"(ra[SLOT][3][0]), "+v"(ra[SLOT][3][1]),       \
                   "+v"(rb[SLOT][0][0]), "+v"(rb[SLOT][0][1]), "+v"(rb[SLOT][1][0]), "+v"(rb[SLOT][1][1]) :: "memory");
#define TRX_TN_FRAG(R) __builtin_bit_cast(bf16x8, uint4{R[0].x, R[0].y, R[1].x, R[1].y})
#define TRX_TN_MFMA(SLOT)                                                                                        \
    _Pragma("unroll") for (int ib_ = 0; ib_ < 4; ++ib_)                                                          \
        _Pragma("unroll") for (int jb_ = 0; jb_ < 2; ++jb_)                                                      \
            acc[ib_][jb_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TRX_TN_FRAG(ra[SLOT][ib_]), TRX_TN_FRAG(rb[SLOT][jb_]), acc[ib_][jb_], 0, 0, 0);
        TRX_TN_READ(0, 0)
        TRX_TN_READ(1, 1)
        TRX_TN_WAIT(0, 12)
        TRX_TN_MFMA(0)
        if (colsum) {
            asm volatile("" : "+v"(cv[0]), "+v"(cv[1]), "+v"(cv[2]), "+v"(cv[3]));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned w[4] = {cv[j][0], cv[j][1], cv[j][2], cv[j][3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cs[2 * e] += __uint_as_float(w[e] << 16);
                    cs[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
                }
            }
        }
        TRX_TN_READ(0, 2)
        TRX_TN_WAIT(1, 12)
        TRX_TN_MFMA(1)
        TRX_TN_READ(1, 3)
        TRX_TN_WAIT(0, 12)
        TRX_TN_MFMA(0)
        TRX_TN_WAIT(1, 0)
        TRX_TN_MFMA(1)
        __syncthreads();
        cur ^= 1;
    }
#undef TRX_TN_STAGE
    if (colsum) {   // 16 row groups -> one value per column, through the (now idle) stage memory
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(tid >> 5) * 256 + (tid & 31) * 8 + e] = cs[e];
        __syncthreads();
        if (tid < 256) {
            float t = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < 16; ++g2) t += red[g2 * 256 + tid];
            p.ws_colsum[(int64_t)split * p.N + n0 + tid] = t;
        }
    }
    // ---- partial tile (fp32): ws[split][n][k]; register t of acc[ib][jb] = C[n = .. + (t&3) + 8(t>>2) + 4hh][k = .. + (lane & 31)]
    float* out = p.ws + (int64_t)split * p.N * p.K;
    const int r = lane & 31;
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const int kcol = k0 + 64 * wave_k + 32 * jb + r;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int nrow = n0 + 128 * wave_n + 32 * ib + (t & 3) + 8 * (t >> 2) + 4 * hh;
                out[(int64_t)nrow * p.K + kcol] = acc[ib][jb][t];
            }
        }
}

// OUT32: the results are written as fp32 (the gradient of an fp32 parameter: no rounding, no cast kernel afterwards)
template <bool OUT32>
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float* __restrict__ ws, int nsplit, int64_t nk, int K, int ldc,
                                                             void* __restrict__ C, const float* __restrict__ ws_colsum, int N,
                                                             void* __restrict__ colsum_out) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (ws_colsum && i < N) {   // the first N / 4 threads also finish the column sums
        f32x4 c = *reinterpret_cast<const f32x4*>(ws_colsum + i);
        for (int j = 1; j < nsplit; ++j) c += *reinterpret_cast<const f32x4*>(ws_colsum + (int64_t)j * N + i);
        if (OUT32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(colsum_out) + i) = c;
        else {
            const f32x2 lo = {c[0], c[1]}, hi = {c[2], c[3]};
            uint2 w;
            w.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2));
            w.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2));
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(colsum_out) + i) = w;
        }
    }
    if (i >= nk) return;
    // four partials in flight per thread (a plain loop waits for each one before asking for the next); the order of
    // the additions is fixed, so the result is reproducible
    f32x4 s4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) s4[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int j = 0;
    for (; j + 3 < nsplit; j += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) s4[u] += *reinterpret_cast<const f32x4*>(ws + (int64_t)(j + u) * nk + i);
    }
    for (; j < nsplit; ++j) s4[0] += *reinterpret_cast<const f32x4*>(ws + (int64_t)j * nk + i);
    const f32x4 s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    const int64_t n = i / K, k = i % K;
    if (OUT32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(C) + n * ldc + k) = s;
    else {
        const f32x2 lo = {s[0], s[1]}, hi = {s[2], s[3]};
        uint2 w;
        w.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2));
        w.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2));
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(C) + n * ldc + k) = w;
    }
}

static void plan(int M, int N, int K, int* tn, int* tk, int* nsplit, int* sps) {
    *tn = N / TILE; *tk = K / TILE;
    const int tiles = *tn * *tk, steps = (M + BM - 1) / BM;
    int s = 256 / tiles;                        // one wave of workgroups on the 256 CUs (a workgroup takes a whole CU)
    if (s < 1) s = 1;
    if (s > 24) s = 24;
    if (s > steps / 4) s = steps / 4 > 0 ? steps / 4 : 1;   // at least 4 steps per split
    *sps = (steps + s - 1) / s;
    *nsplit = (steps + *sps - 1) / *sps;
}

}  // namespace trxtn

extern "C" int64_t trx_gemm_tn_ws_bytes(int M, int N, int K) {
    using namespace trxtn;
    if (M <= 0 || N <= 0 || K <= 0 || N % TILE || K % TILE) return -1;
    int tn, tk, ns, sps;
    plan(M, N, K, &tn, &tk, &ns, &sps);
    return (int64_t)ns * ((int64_t)N * K + N) * (int64_t)sizeof(float);   // partial tiles + partial column sums
}

extern "C" int trx_gemm_tn_bf16(const void* A, int lda, const void* B, int ldb, void* ws, void* C, int ldc, void* colsum_bf16,
                                int out_f32, int M, int N, int K, void* stream) {
    using namespace trxtn;
    if (!A || !B || !C || !ws || M <= 0 || N <= 0 || K <= 0) return TRX_NN_EINVAL;
    if (colsum_bf16 && (reinterpret_cast<uintptr_t>(colsum_bf16) & 7)) return TRX_NN_EINVAL;
    if (N % TILE || K % TILE || lda < N || ldb < K || ldc < K || (lda | ldb) % 8 || ldc % 4) return TRX_NN_EINVAL;
    if (((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(ws)) & 15) ||
        (reinterpret_cast<uintptr_t>(C) & (out_f32 ? 15 : 7)) || (out_f32 && colsum_bf16 && (reinterpret_cast<uintptr_t>(colsum_bf16) & 15)))
        return TRX_NN_EINVAL;
    Params p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.ws = (float*)ws;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb;
    plan(M, N, K, &p.tn, &p.tk, &p.nsplit, &p.steps_per_split);
    p.ws_colsum = colsum_bf16 ? p.ws + (int64_t)p.nsplit * N * K : nullptr;
    // the attribute is per device: remember which devices of this process have it (bit per ordinal)
    static std::atomic<unsigned long long> attr_devs{0ull};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return TRX_NN_EHIP;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess)
            return TRX_NN_EHIP;
        attr_devs.fetch_or(bit, std::memory_order_release);
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(p.tn * p.tk * p.nsplit), dim3(THREADS), LDS_TOTAL, st, p);
    const int64_t nk = (int64_t)N * K;
    if (out_f32)
        hipLaunchKernelGGL(gemm_tn_reduce_kernel<true>, dim3((unsigned)((nk / 4 + 255) / 256)), dim3(256), 0, st, (const float*)ws, p.nsplit,
                           nk, K, ldc, C, p.ws_colsum, N, colsum_bf16);
    else
        hipLaunchKernelGGL(gemm_tn_reduce_kernel<false>, dim3((unsigned)((nk / 4 + 255) / 256)), dim3(256), 0, st, (const float*)ws, p.nsplit,
                           nk, K, ldc, C, p.ws_colsum, N, colsum_bf16);
    return hipGetLastError() == hipSuccess ? TRX_NN_OK : TRX_NN_EHIP;
}
