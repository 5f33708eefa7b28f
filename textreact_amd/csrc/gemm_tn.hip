// gemm_tn.hip -- C[N, K] = A[M, N]^T . B[M, K] in bf16 with fp32 accumulation on gfx950: the WEIGHT GRADIENT of a
// Linear layer (dW = dY^T X), the product the library runs worst in the predictor's training step: it contracts
// over the M = 16,384 token rows into only N*K / 65,536 = 9 .. 36 output tiles, and hipBLASLt does not split the
// contraction (0.48 PFLOP/s, ~24 % of the step).
//
// Here the contraction IS split: workgroup (tile, s) accumulates rows [s * M/S, (s+1) * M/S) of one 256 x 256 tile
// of C into an fp32 partial, a second kernel sums the S partials in a fixed order (deterministic) and rounds to
// bf16.  Round 4: the loop is the retrieval scan's TWO-GROUP PING-PONG (knn_scan.hip) turned sideways.
//
//   Workgroup = 8 waves.  MFMA operands: P = B (X, the k index of C) on the MFMA's row side, S = A (dY, the n index)
//   on its column side -- so a lane's 4 accumulator registers are 4 CONSECUTIVE k of one n.  Group g = wave >> 2 owns
//   k columns 128 g .. 128 g + 127 of the tile, wave wq = wave & 3 the n columns 64 wq .. 64 wq + 63:
//   8 x 4 tiles of v_mfma_f32_16x16x32_bf16 = 128 accumulator registers per wave.
//
//   A K-step is 64 token rows of both operands = four 16 KiB BANDS of 64 rows x 128 columns (X0, X1: the two groups'
//   halves of P; Y0, Y1: the two halves of S), each band double-buffered (stage u & 1 of K-step u): 128 KiB of LDS.
//   Bands arrive by LDS-DMA in 1 KiB pieces of 4 rows x 256 B (no staging registers).  Both operands are consumed
//   TRANSPOSED (the contraction index m is the slow dimension of both): a fragment is two ds_read_b64_tr_b16 --
//   16-lane group gg reads rows 4 gg .. 4 gg + 3 (+ 16 for the second read) of a 32-row half, 16 columns.  A 32-lane
//   half of such a read touches 8 rows x 32 B that are 256 B apart, so 16-byte chunk c of row m is stored at slot
//   c ^ (2 (m & 7)) (a permutation of the DMA's per-lane SOURCE address): conflict-free.
//
//   The two groups share the four SIMDs pairwise and run one barrier interval apart; K-step u, half kk (32 rows):
//     interval 4u   : G0 L(u,0) [DMA Y0 of K-step u+1]     G1 M(u-1,1)
//     interval 4u+1 : G0 M(u,0)                            G1 L(u,0) [DMA Y1 of u+1]
//     interval 4u+2 : G0 L(u,1) [DMA X1 of u+1]            G1 M(u,0)
//     interval 4u+3 : G0 M(u,1)                            G1 L(u,1) [DMA X0 of u+2]
//   L = 24 transposed fragment reads of the next half + 4 DMA pieces per wave, M = 32 MFMAs.  Write-after-read: every
//   fragment read is retired (lgkmcnt(0)) before the barrier that ends its interval and each DMA is issued at least one
//   barrier after the last read of the band it overwrites (X0 is read by group 0 only, last in interval 4u+2; the
//   other bands last in interval 4u+3).  Read-after-write: a wave waits vmcnt(4) at the end of every load phase -- the
//   pieces of its PREVIOUS load phase, two intervals old -- and every band has a barrier between that wait and its
//   first read.  Against the first form of this kernel (DMA burst, one __syncthreads per K-step, 32x32x16 MFMAs:
//   both waves of a SIMD read, waited and multiplied together) see DESIGN.md section 3.4.
//
//   Partials leave in the accumulators' own layout (ws[split][tile][wave][fragment][lane][4]: every store is 1 KiB
//   contiguous); the reduction kernel reads them the same way and writes C row-major.
#include "../../include/trx_nn.h"
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>

namespace trxtn {


typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int TILE = 256, BM = 64, THREADS = 512;
constexpr int BAND2 = 2 * BM * 256;           // a band's two stages: 32 KiB
constexpr int LDS_TOTAL = 4 * BAND2 + 16;     // X0 X1 Y0 Y1: 128 KiB (+ one word of the grouped form)

struct Params {
    const bf16_t* A; const bf16_t* B; float* ws;
    float* ws_colsum;   // [nsplit][N] column sums of A (the bias gradient of the same Linear), or null
    int M, N, K, lda, ldb;
    int tn, tk, nsplit, steps_per_split;
};

// ---- the grouped form (trx_gemm_tn_grouped_*): MANY weight gradients in one persistent launch.  A split contraction
// pays for its partials whatever the shape -- every workgroup stores its 256 KB of accumulators and a second launch reads
// them back: 64 MB each way per call, 25 of a call's 40 .. 78 us -- and one training step makes 88 such calls.  All of a
// step's weight gradients together are ~1,900 tiles of 256 x 256: enough to give every CU whole tiles, so nothing is split,
// nothing is reduced, a tile's result goes straight to the gradient.  Work list: problems are dealt to the 8 XCDs whole
// (longest first; the tiles of a problem read the same token rows and find them in their XCD's L2), the workgroups of an XCD
// (blockIdx & 7, the placement the scan kernel relies on as well: speed only) draw tiles from their XCD's list through
// an atomic counter.
struct GProblem {            // device format, 64 bytes
    const bf16_t* A; const bf16_t* B; float* C; float* colsum;
    int M, N, K, lda, ldb, ldc, tn, tk;
};
struct GHeader { unsigned magic; int nproblems, nitems, off_counters, off_begin, off_probs, off_items, bytes; };
struct GParams { const char* block; int off_counters, off_begin, off_probs, off_items; };
constexpr unsigned G_MAGIC = 0x54524e47u;      // "TRNG"
constexpr int LDS_NEXT = 4 * (2 * BM * 256);   // one word behind the stages: the next item's index (grouped form)

#define TN_SGPR64(P) ((((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)(P) >> 32))) << 32) | \
                      (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(P)))

template <bool GROUPED>
__global__ __launch_bounds__(THREADS, 2) void gemm_tn_kernel(Params p, GParams g) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void gbl_void;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const unsigned ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int xb = grp ^ 1;                      // the X band this wave's group STAGES (it reads band grp)

    // ---- transposed fragment addresses.  Lane -> row 4 gg + qq of a 32-row half (+ 16 for the second read), columns
    // 4 pp .. 4 pp + 3 of a 16-column block; the read hands lane l the 4 rows' values of column l & 15.
    unsigned fx[8], fy[4];
    {
        const int gg = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
        const int r7 = 4 * (gg & 1) + qq;                   // (row & 7) of every row this lane addresses
        const unsigned rowoff = ldsbase + (unsigned)((4 * gg + qq) * 256 + 8 * (pp & 1));
#pragma unroll
        for (int ib = 0; ib < 8; ++ib) fx[ib] = rowoff + (unsigned)(grp * BAND2 + (((2 * ib + (pp >> 1)) ^ (2 * r7)) << 4));
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) fy[jb] = rowoff + (unsigned)((2 + (wq >> 1)) * BAND2 + (((8 * (wq & 1) + 2 * jb + (pp >> 1)) ^ (2 * r7)) << 4));
    }
    u32x4 ones4 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    asm volatile("" : "+v"(ones4));     // one register quad, not a constant rebuilt per use

    // grouped form: this XCD's list and the first item
    int g_end = 0, g_item = 0;
    int* g_counter = nullptr;
    if constexpr (GROUPED) {
        const int x = blockIdx.x & 7;
        const int* begin = reinterpret_cast<const int*>(g.block + g.off_begin);
        g_end = begin[x + 1];
        g_counter = reinterpret_cast<int*>(const_cast<char*>(g.block) + g.off_counters) + x;
        int first = 0;
        if (tid == 0) first = begin[x] + __hip_atomic_fetch_add(g_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wave == 0) {
            first = __builtin_amdgcn_readfirstlane(first);
            asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(ldsbase + LDS_NEXT), "v"(first) : "memory");
        }
        __builtin_amdgcn_s_barrier();
        unsigned it;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(it) : "v"(ldsbase + LDS_NEXT) : "memory");
        g_item = __builtin_amdgcn_readfirstlane((int)it);
        __builtin_amdgcn_s_barrier();       // everybody has read the word before wave 0 writes the next one
    }
  for (;;) {
    // ---- what this workgroup computes now
    const bf16_t* pA; const bf16_t* pB;
    int pM, pN, pK, plda, pldb, ptk, split, kt, nt, step0, nsteps;
    float* gC = nullptr; float* gcolsum = nullptr; int gldc = 0;
    if constexpr (GROUPED) {
        if (g_item >= g_end) break;
        const unsigned item = (unsigned)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const unsigned*>(g.block + g.off_items)[g_item]);
        const GProblem& q = reinterpret_cast<const GProblem*>(g.block + g.off_probs)[item >> 12];
        const int t = (int)(item & 4095u);
        // (wave-uniform values that arrive through vector loads -- the table is ordinary global memory -- made scalar by hand)
        auto sp = [](const void* v) { return reinterpret_cast<const void*>(TN_SGPR64(v)); };
        auto si = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
        pA = (const bf16_t*)sp(q.A); pB = (const bf16_t*)sp(q.B); pM = si(q.M); pN = si(q.N); pK = si(q.K); plda = si(q.lda); pldb = si(q.ldb);
        ptk = si(q.tk); gC = (float*)sp(q.C); gcolsum = (float*)sp(q.colsum); gldc = si(q.ldc);
        kt = t % ptk; nt = t / ptk; split = 0; step0 = 0;
        nsteps = (pM + BM - 1) / BM;
    } else {
        // Which (split, tile) a workgroup takes.  The tn x tk tiles of ONE split read the same token rows: numbered
        // split-SLOWEST and dealt to the XCDs in contiguous runs (the scan kernel's bijective remap), an XCD holds the tiles
        // of one or two splits, which walk the same rows in step and find each other's lines in L2 (round 3: numbered
        // split-fastest every operand byte came tk or tn times over the fabric).
        const int bid = blockIdx.x, nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7;
        const int v = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);      // XCD x owns [base, base + q (+1))
        const int tiles = p.tn * p.tk;
        split = v / tiles;
        const int t = v - split * tiles;
        kt = t % p.tk; nt = t / p.tk;
        pA = p.A; pB = p.B; pM = p.M; pN = p.N; pK = p.K; plda = p.lda; pldb = p.ldb; ptk = p.tk;
        const int total_steps = (pM + BM - 1) / BM;          // the last step may be partial: rows >= M read as zeros
        step0 = split * p.steps_per_split;
        nsteps = min(p.steps_per_split, total_steps - step0);     // >= 1 (plan())
    }
    const int n0 = nt * TILE, k0 = kt * TILE;

    // ---- staging geometry: a piece = one buffer_load_dwordx4 ... lds = 4 rows x 256 B of a band, written lane-linear
    // (row 4 piece + lane / 16, slot lane % 16); slot s of row r receives global chunk s ^ (2 (r & 7)).  In a load phase
    // wave wq of the loading group moves pieces 4 wq .. 4 wq + 3 of one band: rows 16 wq + 4 i + prow.  The operands are
    // addressed through buffer descriptors (wave-uniform base and K-step offset in scalar registers, a 32-bit lane
    // offset): no 64-bit vector address arithmetic in the loop, and the range check IS the ragged edge -- a row past M lies
    // past the descriptor's last byte and arrives in LDS as zeros (tools/experiments/buffer_lds_oob.hip), so the last
    // step of a token count that is not a multiple of 64 needs no code at all, and neither do the pieces the DMA
    // cursors issue past the end of a split (rows of the next split, or zeros; their stages are dead).
    unsigned voY[4], voX[4];                     // lane offsets of this wave's four pieces (bytes)
    {
        const int prow = lane >> 4, pslot = lane & 15;
        const int c_even = pslot ^ (2 * prow);      // pieces with (piece & 1) == 0: (row & 7) == prow
        const int c_odd = c_even ^ 8;               //                          1:              4 + prow
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            voY[i] = (unsigned)(((16 * wq + 4 * i + prow) * plda + ((i & 1) ? c_odd : c_even) * 8) * 2);
            voX[i] = (unsigned)(((16 * wq + 4 * i + prow) * pldb + ((i & 1) ? c_odd : c_even) * 8) * 2);
        }
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)pA, 0, (int)(unsigned)((((int64_t)pM - 1) * plda + pN) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)pB, 0, (int)(unsigned)((((int64_t)pM - 1) * pldb + pK) * 2), 0x00020000);
#define TN_BUF16(RSRC, VOFF, SOFF, LDSADDR)                                                                      \
    {                                                                                                            \
        const unsigned la_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(LDSADDR));                            \
        const unsigned so_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(SOFF));                               \
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(VOFF), "s"(RSRC), "s"(so_), "s"(la_) : "memory", "m0"); \
    }
    // DMA cursors (wave-uniform byte offsets of the K-step whose band this wave's group stages next)
    const unsigned kstepY = (unsigned)(BM * plda * 2), kstepX = (unsigned)(BM * pldb * 2);
    unsigned soY = (unsigned)((((int64_t)step0 + 1) * BM * plda + n0 + 128 * grp) * 2);
    unsigned soX = (unsigned)((((int64_t)step0 + (grp ? 2 : 1)) * BM * pldb + k0 + 128 * xb) * 2);
#define TN_DMA_Y(STG)                                                                                            \
    {                                                                                                            \
        const unsigned l_ = ldsbase + (unsigned)((2 + grp) * BAND2 + (STG) * 16384 + wq * 4096);                 \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) TN_BUF16(rsA, voY[i_], soY, l_ + i_ * 1024);            \
        soY += kstepY;                                                                                           \
    }
    // group 0 stages X1 of the next K-step into the other stage, group 1 X0 of the one after into this stage
#define TN_DMA_X(STG)                                                                                            \
    {                                                                                                            \
        const unsigned l_ = ldsbase + (unsigned)(xb * BAND2 + ((STG) ^ xb) * 16384 + wq * 4096);                 \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) TN_BUF16(rsB, voX[i_], soX, l_ + i_ * 1024);            \
        soX += kstepX;                                                                                           \
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Column sums of A (= db of the Linear whose dW this is) ride along ON THE MATRIX PIPE: the S fragments a wave holds
    // anyway, multiplied by a P fragment of ones, are the column sums of its 64 n columns over the half K-step (all 16
    // rows of the result equal): 4 more MFMAs per phase, no LDS reads, no vector instructions.  The tk workgroups of a
    // tile row and split hold the same S bands, so they share the work: workgroup kt takes the K-steps with
    // (step % tk) == kt (group 0 only) and writes its own partial row -- every workgroup of the launch pays the same
    // 1 / (2 tk) of 12.5 %.  The four MFMAs and the scalar branch around them are ONE asm statement: with a branch it can
    // see in this loop hipcc spills (40 .. 536 bytes of scratch per lane, depending on where the condition sits).
    const int cs_tk = GROUPED ? 1 : ptk;      // grouped form: the workgroup of a tile row's first k tile takes every K-step
    const bool cs_on = grp == 0 && (GROUPED ? (gcolsum != nullptr && kt == 0) : p.ws_colsum != nullptr);
    f32x4 acs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acs[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // ---- prologue: K-step 0 in full and X0 of K-step 1 (all waves: 2 pieces of each band) ----
    {
        const int prow = lane >> 4, pslot = lane & 15;
        const unsigned so0A = (unsigned)(((int64_t)step0 * BM * plda + n0) * 2), so0B = (unsigned)(((int64_t)step0 * BM * pldb + k0) * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pc = 2 * wave + i, row = 4 * pc + prow, chunk = (pslot ^ (2 * prow)) ^ (8 * i);
            const unsigned va = (unsigned)((row * plda + chunk * 8) * 2), vb = (unsigned)((row * pldb + chunk * 8) * 2);
            const unsigned l = ldsbase + (unsigned)(pc * 1024);
            TN_BUF16(rsB, vb, so0B, l + 0 * BAND2);
            TN_BUF16(rsB, vb, so0B + 256u, l + 1 * BAND2);
            TN_BUF16(rsA, va, so0A, l + 2 * BAND2);
            TN_BUF16(rsA, va, so0A + 256u, l + 3 * BAND2);
            TN_BUF16(rsB, vb, so0B + kstepX, l + 0 * BAND2 + 16384);
        }
        // grouped form: the NEXT item's index is drawn here, under the prologue's loads (its wait is the one below), and
        // parked in LDS for the end of this item
        int nxt = 0;
        if constexpr (GROUPED) {
            if (tid == 0) nxt = __hip_atomic_fetch_add(g_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if constexpr (GROUPED) {
            if (wave == 0) {
                const int first_of_xcd = reinterpret_cast<const int*>(g.block + g.off_begin)[blockIdx.x & 7];
                nxt = __builtin_amdgcn_readfirstlane(nxt) + first_of_xcd;
                asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(ldsbase + LDS_NEXT), "v"(nxt) : "memory");
            }
        }
        __builtin_amdgcn_s_barrier();
    }
    if (grp) __builtin_amdgcn_s_barrier();       // group 1 runs one interval late

    uint2 xl[8], xh[8], yl[4], yh[4];           // fragments: low / high 4 rows of a lane's 8 contraction rows
#define TN_READ(KK, STG)                                                                                         \
    {                                                                                                            \
        _Pragma("unroll") for (int jb_ = 0; jb_ < 4; ++jb_)                                                      \
            asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"            \
                         : "=&v"(yl[jb_]), "=&v"(yh[jb_]) : "v"(fy[jb_]), "n"((STG) * 16384 + (KK) * 8192), "n"((STG) * 16384 + (KK) * 8192 + 4096) : "memory"); \
        _Pragma("unroll") for (int ib_ = 0; ib_ < 8; ++ib_)                                                      \
            asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"            \
                         : "=&v"(xl[ib_]), "=&v"(xh[ib_]) : "v"(fx[ib_]), "n"((STG) * 16384 + (KK) * 8192), "n"((STG) * 16384 + (KK) * 8192 + 4096) : "memory"); \
    }
    // end of a load phase: the fragment reads are retired, all but the four pieces just issued have landed
#define TN_WAIT_L()                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(4)" ::: "memory");                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    __builtin_amdgcn_s_barrier();                                                                                \
    __builtin_amdgcn_sched_barrier(0);
#define TN_END_M()                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    __builtin_amdgcn_s_barrier();                                                                                \
    __builtin_amdgcn_sched_barrier(0);
#define TN_FRAG(LO, HI) __builtin_bit_cast(bf16x8, (u32x4){LO.x, LO.y, HI.x, HI.y})
    // the n block index runs back and forth over consecutive rows of MFMAs: one operand changes per instruction
#define TN_MFMA()                                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                               \
    _Pragma("unroll") for (int ib_ = 0; ib_ < 8; ++ib_)                                                          \
        _Pragma("unroll") for (int j0_ = 0; j0_ < 4; ++j0_) {                                                    \
            const int jb_ = (ib_ & 1) ? 3 - j0_ : j0_;                                                           \
            acc[ib_][jb_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(TN_FRAG(xl[ib_], xh[ib_]), TN_FRAG(yl[jb_], yh[jb_]), acc[ib_][jb_], 0, 0, 0); \
        }                                                                                                        \
    __builtin_amdgcn_s_setprio(0);
#define TN_MFMA_CS(COND)                                                                                         \
    {                                                                                                            \
        /* (indexed by a loop variable: with literal indices this hipcc folds yl[0].x to undef -- it does not see the   \
           asm statements that wrote it -- and all four MFMAs read one garbage register) */                       \
        bf16x8 yy_[4];                                                                                           \
        _Pragma("unroll") for (int jb_ = 0; jb_ < 4; ++jb_) yy_[jb_] = TN_FRAG(yl[jb_], yh[jb_]);                \
        asm volatile("s_cmp_eq_u32 %9, 0\n\ts_cbranch_scc1 1f\n\t"                                              \
                     "v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %6, %1\n\t"     \
                     "v_mfma_f32_16x16x32_bf16 %2, %4, %7, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %4, %8, %3\n1:"     \
                     : "+v"(acs[0]), "+v"(acs[1]), "+v"(acs[2]), "+v"(acs[3])                                    \
                     : "v"(ones4), "v"(yy_[0]), "v"(yy_[1]), "v"(yy_[2]), "v"(yy_[3]), "s"(COND) : "scc");           \
    }
    // one K-step in stage STG (see the interval table in the header)
#define TN_KSTEP(STG)                                                                                            \
    {                                                                                                            \
        const unsigned cs_k_ = (unsigned)__builtin_amdgcn_readfirstlane((cs_on && cs_u == 0) ? 1 : 0);                                                  \
        cs_u = cs_u + 1 == cs_tk ? 0 : cs_u + 1;                                                                  \
        TN_READ(0, STG);                                                                                         \
        TN_DMA_Y((STG) ^ 1);                                                                                     \
        TN_WAIT_L();                                                                                             \
        TN_MFMA_CS(cs_k_);                                                                                       \
        TN_MFMA();                                                                                               \
        TN_END_M();                                                                                              \
        TN_READ(1, STG);                                                                                         \
        TN_DMA_X(STG);                                                                                           \
        TN_WAIT_L();                                                                                             \
        TN_MFMA_CS(cs_k_);                                                                                       \
        TN_MFMA();                                                                                               \
        TN_END_M();                                                                                              \
    }
    int cs_u = GROUPED ? 0 : (step0 % ptk + ptk - kt) % ptk;        // (step - kt) mod tk of the next K-step: 0 = this workgroup's turn
    int u = 0;
    for (; u + 1 < nsteps; u += 2) { TN_KSTEP(0); TN_KSTEP(1); }
    if (u < nsteps) TN_KSTEP(0);
    if (!grp) __builtin_amdgcn_s_barrier();      // group 0 waits for the interval group 1 is behind
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if constexpr (GROUPED) {
        // ---- the tile itself, straight into the gradient: register r of acc[ib][jb] =
        // C[n = n0 + 64 wq + 16 jb + (lane & 15)][k = k0 + 128 grp + 16 ib + 4 (lane >> 4) + r]
        // (N need not be a multiple of the tile here -- a vocabulary projection: the columns of dY past N that the last tile row
        // staged are the next token row's first values, finite, and what they produced is simply not stored)
        if (cs_on && lane < 16) {
            float* o = gcolsum + n0 + 64 * wq + lane;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
                if (n0 + 64 * wq + lane + 16 * jb < pN) o[16 * jb] = acs[jb][0];
        }
        const int nrow0 = n0 + 64 * wq + (lane & 15);
        float* out = gC + (int64_t)nrow0 * gldc + k0 + 128 * grp + 4 * (lane >> 4);
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
            if (nrow0 + 16 * jb < pN) {
#pragma unroll
                for (int ib = 0; ib < 8; ++ib) *reinterpret_cast<f32x4*>(out + (int64_t)16 * jb * gldc + 16 * ib) = acc[ib][jb];
            }
        // the next item: every wave's pieces have landed (its own vmcnt(0) above, then this barrier), so the next prologue
        // may write the stages; the index was parked by wave 0 during this item's prologue
        __builtin_amdgcn_s_barrier();
        unsigned it;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(it) : "v"(ldsbase + LDS_NEXT) : "memory");
        g_item = __builtin_amdgcn_readfirstlane((int)it);
        __builtin_amdgcn_s_barrier();       // everybody has read the word before wave 0 parks the next one
    } else {
        if (cs_on && lane < 16) {      // partial row (split, kt): all 16 rows of the result are the same sums
            float* o = p.ws_colsum + ((int64_t)split * p.tk + kt) * p.N + n0 + 64 * wq + lane;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) o[16 * jb] = acs[jb][0];
        }
        // ---- partial tile (fp32) in the accumulators' layout: ws[split][tile][wave][ib][jb][lane][4]
        float* out = p.ws + (((int64_t)split * (p.tn * p.tk) + (nt * p.tk + kt)) * 8 + wave) * (32 * 256) + lane * 4;
#pragma unroll
        for (int ib = 0; ib < 8; ++ib)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) *reinterpret_cast<f32x4*>(out + (ib * 4 + jb) * 256) = acc[ib][jb];
        break;
    }
  }
}

// OUT32: the results are written as fp32 (the gradient of an fp32 parameter: no rounding, no cast kernel afterwards).
// A block = the 512 accumulator quads of one (tile, wave, jb): 16 rows of C x 128 consecutive k.  Its eight waves read
// 1 KiB runs of every partial and, between them, write whole 512-byte (fp32) row segments -- all writers of a cache line
// sit in ONE block, hence on one XCD (numbered linearly over the quads, the two halves of a line came from two XCDs'
// L2s as partial lines: the pass took 15 .. 26 us instead of 12 .. 15).
template <bool OUT32>
__global__ __launch_bounds__(512) void gemm_tn_reduce_kernel(const float* __restrict__ ws, int nsplit, int64_t nk, int tk, int ldc,
                                                             void* __restrict__ C, const float* __restrict__ ws_colsum, int N,
                                                             void* __restrict__ colsum_out) {
    const int nmain = (int)(nk >> 11);
    if ((int)blockIdx.x >= nmain) {
        // the column sums: block -> 128 columns, thread -> (column quad t & 31, row class t >> 5): rows class, class + 16, ...
        // of the nsplit x tk partial rows, all of a thread's loads in flight at once (the rows were written by other XCDs a
        // moment ago -- each load is a trip to memory: summed one after the other by one thread per column quad, the 72 .. 84
        // rows of a 768-wide gradient took 30 us), then the 16 classes in a fixed order through LDS
        __shared__ f32x4 red[16][32];
        const int cq = threadIdx.x & 31, cls = threadIdx.x >> 5;
        const int64_t i0 = ((int64_t)(blockIdx.x - nmain) * 32 + cq) * 4;
        const int nrows = nsplit * tk;
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        for (int j = cls; j < nrows; j += 16) c += *reinterpret_cast<const f32x4*>(ws_colsum + (int64_t)j * N + i0);
        red[cls][cq] = c;
        __syncthreads();
        if (cls == 0) {
            f32x4 t = red[0][cq];
#pragma unroll
            for (int g = 1; g < 16; ++g) t += red[g][cq];
            if (OUT32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(colsum_out) + i0) = t;
            else {
                const f32x2 lo = {t[0], t[1]}, hi = {t[2], t[3]};
                uint2 w;
                w.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2));
                w.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2));
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(colsum_out) + i0) = w;
            }
        }
        return;
    }
    // this thread's quad in the accumulators' layout [tile][wave][ib][jb][lane] (see the end of gemm_tn_kernel)
    const int jb = blockIdx.x & 3, wave = (blockIdx.x >> 2) & 7, tile = blockIdx.x >> 5;
    const int ib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t i = ((((int64_t)tile * 8 + wave) * 8 + ib) * 4 + jb) * 256 + lane * 4;
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
    const int nt = tile / tk, kt = tile - nt * tk;
    const int64_t n = nt * TILE + 64 * (wave & 3) + 16 * jb + (lane & 15);
    const int k = kt * TILE + 128 * (wave >> 2) + 16 * ib + 4 * (lane >> 4);
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
    return (int64_t)ns * ((int64_t)N * K + (int64_t)tk * N) * (int64_t)sizeof(float);   // partial tiles + partial column sums (one row per split and k tile)
}

extern "C" int trx_gemm_tn_bf16(const void* A, int lda, const void* B, int ldb, void* ws, void* C, int ldc, void* colsum_bf16,
                                int out_f32, int M, int N, int K, void* stream) {
    using namespace trxtn;
    if (!A || !B || !C || !ws || M <= 0 || N <= 0 || K <= 0) return TRX_NN_EINVAL;
    if (colsum_bf16 && (reinterpret_cast<uintptr_t>(colsum_bf16) & 7)) return TRX_NN_EINVAL;
    if (N % TILE || K % TILE || lda < N || ldb < K || ldc < K || (lda | ldb) % 8 || ldc % 4) return TRX_NN_EINVAL;
    // the operands are addressed through buffer descriptors: 32-bit byte offsets, a little past the last row
    if (((int64_t)M + 4 * BM) * lda * 2 >= (1ll << 32) || ((int64_t)M + 4 * BM) * ldb * 2 >= (1ll << 32)) return TRX_NN_EINVAL;
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
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess)
            return TRX_NN_EHIP;
        attr_devs.fetch_or(bit, std::memory_order_release);
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gemm_tn_kernel<false>, dim3(p.tn * p.tk * p.nsplit), dim3(THREADS), LDS_TOTAL, st, p, GParams{});
    const int64_t nk = (int64_t)N * K;
    if (out_f32)
        hipLaunchKernelGGL(gemm_tn_reduce_kernel<true>, dim3((unsigned)(nk / 2048 + (p.ws_colsum ? N / 128 : 0))), dim3(512), 0, st, (const float*)ws, p.nsplit,
                           nk, p.tk, ldc, C, p.ws_colsum, N, colsum_bf16);
    else
        hipLaunchKernelGGL(gemm_tn_reduce_kernel<false>, dim3((unsigned)(nk / 2048 + (p.ws_colsum ? N / 128 : 0))), dim3(512), 0, st, (const float*)ws, p.nsplit,
                           nk, p.tk, ldc, C, p.ws_colsum, N, colsum_bf16);
    return hipGetLastError() == hipSuccess ? TRX_NN_OK : TRX_NN_EHIP;
}

// ---- the grouped form: plan on the host into the caller's (pinned) memory, upload by the caller, run ----
#include <algorithm>
#include <vector>

static bool tn_problem_ok(const trx_tn_problem& q) {
    using namespace trxtn;
    if (!q.A || !q.B || !q.C || q.M <= 0 || q.N <= 0 || q.K <= 0) return false;
    if (q.N % 8 || q.K % TILE || q.lda < q.N || q.ldb < q.K || q.ldc < q.K || (q.lda | q.ldb) % 8 || q.ldc % 4) return false;      // (N: any multiple of 8)
    if (((int64_t)q.M + 4 * BM) * q.lda * 2 >= (1ll << 32) || ((int64_t)q.M + 4 * BM) * q.ldb * 2 >= (1ll << 32)) return false;
    if ((reinterpret_cast<uintptr_t>(q.A) | reinterpret_cast<uintptr_t>(q.B) | reinterpret_cast<uintptr_t>(q.C)) & 15) return false;
    if (q.colsum && (reinterpret_cast<uintptr_t>(q.colsum) & 3)) return false;
    if (((q.N + TILE - 1) / TILE) * (q.K / TILE) > 4096) return false;
    return true;
}

extern "C" int64_t trx_gemm_tn_grouped_block_bytes(const trx_tn_problem* probs, int n) {
    using namespace trxtn;
    if (!probs || n <= 0 || n >= (1 << 19)) return -1;
    int64_t items = 0;
    for (int i = 0; i < n; ++i) {
        if (!tn_problem_ok(probs[i])) return -1;
        items += (int64_t)((probs[i].N + TILE - 1) / TILE) * (probs[i].K / TILE);
    }
    return (int64_t)sizeof(GHeader) + 8 * 4 + 12 * 4 + (int64_t)n * (int64_t)sizeof(GProblem) + items * 4;
}

extern "C" int trx_gemm_tn_grouped_plan(const trx_tn_problem* probs, int n, void* host_block, int64_t host_bytes) {
    using namespace trxtn;
    const int64_t need = trx_gemm_tn_grouped_block_bytes(probs, n);
    if (need < 0 || !host_block || host_bytes < need) return TRX_NN_EINVAL;
    char* base = static_cast<char*>(host_block);
    GHeader* h = reinterpret_cast<GHeader*>(base);
    h->magic = G_MAGIC; h->nproblems = n;
    h->off_counters = (int)sizeof(GHeader);
    h->off_begin = h->off_counters + 8 * 4;
    h->off_probs = h->off_begin + 12 * 4;
    h->off_items = h->off_probs + n * (int)sizeof(GProblem);
    int* counters = reinterpret_cast<int*>(base + h->off_counters);
    int* begin = reinterpret_cast<int*>(base + h->off_begin);
    GProblem* gp = reinterpret_cast<GProblem*>(base + h->off_probs);
    unsigned* items = reinterpret_cast<unsigned*>(base + h->off_items);
    for (int x = 0; x < 8; ++x) counters[x] = 0;
    // problems to the 8 XCDs whole, longest first into the least loaded (work = tiles x 64-row steps) ...
    std::vector<int> order(n);
    std::vector<int64_t> work(n);
    for (int i = 0; i < n; ++i) {
        const trx_tn_problem& q = probs[i];
        gp[i] = GProblem{(const bf16_t*)q.A, (const bf16_t*)q.B, (float*)q.C, (float*)q.colsum, q.M, q.N, q.K, q.lda, q.ldb, q.ldc, (q.N + TILE - 1) / TILE, q.K / TILE};
        order[i] = i;
        work[i] = (int64_t)gp[i].tn * gp[i].tk * ((q.M + BM - 1) / BM);
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return work[a] > work[b]; });
    std::vector<int> bin[8];
    int64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i : order) {
        int best = 0;
        for (int x = 1; x < 8; ++x) if (load[x] < load[best]) best = x;
        bin[best].push_back(i); load[best] += work[i];
    }
    // ... and inside an XCD's list the problems with the most steps per tile first, so that what is left when the list
    // runs dry are short tiles
    int pos = 0;
    for (int x = 0; x < 8; ++x) {
        begin[x] = pos;
        std::stable_sort(bin[x].begin(), bin[x].end(), [&](int a, int b) { return probs[a].M > probs[b].M; });
        for (int i : bin[x])
            for (int t = 0; t < gp[i].tn * gp[i].tk; ++t) items[pos++] = ((unsigned)i << 12) | (unsigned)t;
    }
    begin[8] = pos;
    h->nitems = pos;
    h->bytes = (int)need;
    return TRX_NN_OK;
}

extern "C" int trx_gemm_tn_grouped_run(const void* dev_block, const void* host_block, void* stream) {
    using namespace trxtn;
    if (!dev_block || !host_block) return TRX_NN_EINVAL;
    const GHeader* h = static_cast<const GHeader*>(host_block);
    if (h->magic != G_MAGIC || h->nitems <= 0) return TRX_NN_EINVAL;
    static std::atomic<unsigned long long> attr_devs{0ull};
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return TRX_NN_EHIP;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL) != hipSuccess)
            return TRX_NN_EHIP;
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return TRX_NN_EHIP;
        cus[dev & 63].store(n, std::memory_order_relaxed);
        attr_devs.fetch_or(bit, std::memory_order_release);
    }
    // one workgroup per CU (128 KiB of LDS each), a multiple of 8 so that every XCD has the same number
    int nwg = cus[dev & 63].load(std::memory_order_relaxed) & ~7;
    if (nwg < 8) nwg = 8;
    if (nwg > ((h->nitems + 7) & ~7)) nwg = (h->nitems + 7) & ~7;
    GParams g{static_cast<const char*>(dev_block), h->off_counters, h->off_begin, h->off_probs, h->off_items};
    hipLaunchKernelGGL(gemm_tn_kernel<true>, dim3(nwg), dim3(THREADS), LDS_TOTAL, (hipStream_t)stream, Params{}, g);
    return hipGetLastError() == hipSuccess ? TRX_NN_OK : TRX_NN_EHIP;
}

// The three steps in one call, with the plan staged in pinned memory the LIBRARY owns: a ring of slots, each guarded by an
// event (a slot is rewritten only after the copy that read it has run); on a capturing stream a block of its own that is
// never reused -- the captured copy reads it again at every replay (the operands' addresses in it are the graph's static
// ones).  dev_block: trx_gemm_tn_grouped_block_bytes() bytes of device memory, the caller's, alive until the launch has run.
#include <mutex>
namespace trxtn {
struct PlanSlot { void* host = nullptr; int64_t bytes = 0; hipEvent_t ev = nullptr; int ev_dev = -1; bool used = false; };
static std::mutex g_plan_mu;
static PlanSlot g_plan_ring[8];
static int g_plan_next = 0;
static void* pinned_alloc(int64_t bytes) {
    // (allocation is not a stream operation, but a capture in "global" mode rejects it: relaxed for the length of the call)
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    hipThreadExchangeStreamCaptureMode(&mode);
    void* p = nullptr;
    const hipError_t e = hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault);
    hipThreadExchangeStreamCaptureMode(&mode);
    return e == hipSuccess ? p : nullptr;
}
}  // namespace trxtn

extern "C" int trx_gemm_tn_grouped(const trx_tn_problem* probs, int n, void* dev_block, int64_t dev_bytes, void* stream) {
    using namespace trxtn;
    const int64_t need = trx_gemm_tn_grouped_block_bytes(probs, n);
    if (need < 0 || !dev_block || dev_bytes < need) return TRX_NN_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) return TRX_NN_EHIP;
    std::lock_guard<std::mutex> lock(g_plan_mu);
    void* host = nullptr;
    PlanSlot* slot = nullptr;
    if (cap != hipStreamCaptureStatusNone) {
        host = pinned_alloc(need);                    // lives as long as the process: the graph may be replayed until then
        if (!host) return TRX_NN_EHIP;
    } else {
        slot = &g_plan_ring[g_plan_next];
        g_plan_next = (g_plan_next + 1) & 7;
        if (slot->used && hipEventSynchronize(slot->ev) != hipSuccess) return TRX_NN_EHIP;
        if (slot->bytes < need) {
            if (slot->host) hipHostFree(slot->host);
            slot->host = pinned_alloc(need); slot->bytes = slot->host ? need : 0;
            if (!slot->host) return TRX_NN_EHIP;
        }
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return TRX_NN_EHIP;
        if (slot->ev && slot->ev_dev != dev) { hipEventDestroy(slot->ev); slot->ev = nullptr; }      // an event belongs to the device it was made on
        if (!slot->ev) {
            if (hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming) != hipSuccess) return TRX_NN_EHIP;
            slot->ev_dev = dev;
        }
        host = slot->host;
    }
    int rc = trx_gemm_tn_grouped_plan(probs, n, host, need);
    if (rc != TRX_NN_OK) return rc;
    if (hipMemcpyAsync(dev_block, host, (size_t)need, hipMemcpyHostToDevice, st) != hipSuccess) return TRX_NN_EHIP;
    if (slot) {
        if (hipEventRecord(slot->ev, st) != hipSuccess) return TRX_NN_EHIP;
        slot->used = true;
    }
    return trx_gemm_tn_grouped_run(dev_block, host, stream);
}
