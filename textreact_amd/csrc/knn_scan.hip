// knn_scan.hip -- the hot kernel: Q x C^T as a bf16 MFMA GEMM with the top-k selection fused into
// the epilogue, so the Q x N score matrix never exists.  gfx950 only.
//
// Replaces the inside of faiss IndexFlat::search called at retrieve/retrieve_faiss.py:71 (the
// sgemm blocks + per-query heap of SURVEY.md section 2b rows N2/N3).
//
// Shape of the work (round 2; the structure was chosen in tools/scan_lab.hip)
//   workgroup = 512 threads = 8 waves, owns ONE tile of 256 queries and walks a contiguous range
//   ("split") of 256-row corpus tiles.  Operands are swapped with respect to the usual GEMM naming:
//   A = corpus rows (MFMA M side), B = queries (MFMA N side), so in the accumulator the QUERY is on
//   the lane (col = lane & 15) and a lane's registers are consecutive corpus rows: everything the
//   selection does per query is lane-local.  Waves are laid out 2 (M) x 4 (N): a wave owns 128 corpus
//   rows x 64 queries = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16 = 128 accumulator registers.
//
// Two-group ping-pong.  Waves 0-3 (group 0, corpus rows 0-127 of every tile) and waves 4-7 (group 1,
//   rows 128-255) share the four SIMDs pairwise and run one barrier interval apart: while one group
//   issues the 32 MFMAs of a 32-deep half K-step, the other reads its 12 fragments of the next half
//   and issues its share of the LDS-DMA.  Phase (K-step u, half kk):
//     interval 4u   : G0 L(u,0) [DMA B rows 0-127 of K-step u+1]     G1 M(u-1,1)
//     interval 4u+1 : G0 M(u,0)                                      G1 L(u,0) [DMA B rows 128-255 of u+1]
//     interval 4u+2 : G0 L(u,1) [DMA A rows 128-255 of u+1]          G1 M(u,0)
//     interval 4u+3 : G0 M(u,1)                                      G1 L(u,1) [DMA A rows 0-127 of u+2]
//   K-step u lives in LDS stage u & 1 (A 32 KiB + B 32 KiB per stage).  Write-after-read: every fragment
//   read is retired (lgkmcnt(0)) before the barrier that ends its interval, and each DMA is issued at
//   least one barrier after the last read of the half-tile it overwrites (A rows 0-127 are read by group
//   0 only, last in interval 4u+2; everything else last in interval 4u+3).  Read-after-write: a wave
//   waits vmcnt(4) at the end of every load phase -- all but the four pieces it has just issued, i.e.
//   the pieces of its previous load phase, two intervals old -- and every half-tile has a barrier
//   between that wait and its first read.  The K-step sequence runs on across tiles.
//
// LDS image: 16-byte chunk c of row r sits at slot c ^ ((r >> 1) & 7) (permutation on the per-lane DMA
//   SOURCE address), which makes the ds_read_b128 fragment reads conflict-free.
//
// Selection (exact with respect to the approximate key; nothing in it needs a workgroup barrier or an atomic):
//   * filter + listing, inside the gaps of a tile's LAST 32 MFMAs.  Every finished group of 4 accumulator
//     rows (one v_mfma tile = 4 corpus rows x the lane's query) is reduced to its maximum (v_max + v_max3),
//     compared with the query's threshold straight into a scalar register pair (v_cmp), and tested one row
//     of MFMAs later by a scalar branch that is almost never taken (about 3 of the 32 groups of a tile once
//     the thresholds are warm; the untaken tests cost nothing measurable).  A group with a hit lists its
//     passing rows right there, where it is a known register: a masked store of the packed (key, id) into
//     the LANE's own list -- (query, split, wave row, row quad), 127 slots in HBM, slot counter in a byte of
//     a register of that lane, nobody else writes that list.
//   * threshold: each lane keeps the J best tile maxima it has seen for each of its 4 queries
//     (J = kprime / 8; in LDS between tiles).  They are maxima of J different tiles, and the 8 lanes of a
//     query (4 row quads x 2 wave rows) see disjoint rows: the minimum over those 8 lanes of the J-th best is
//     a key that at least 8 J = kprime corpus rows reach, so a row below it is outside the top kprime.
//     Refreshed every 8th tile (2 cross-lane steps + the partner wave's value through LDS, which may be
//     stale: stale = lower = still valid) and seeded by a bootstrap launch of this kernel over 16 tiles.
//   * shared between the splits of a query through g_thr, FOUR slots per query (slot = split & 3): a split
//     publishes (atomicMax) a key that at least kprime / 4 of ITS rows reach -- kprime 16: the smaller of its two
//     wave rows' second-best tracked maxima, 2 + 2 rows -- and the splits see disjoint rows, so the minimum over
//     the four slots is a key that kprime rows of the corpus reach (kprime 32: the fourth-best of 16, 4 + 4 rows).  That is a far tighter bound than any split
//     finds alone (its own needs all kprime rows inside the split, through per-lane J-th bests): the rows listed
//     per tile fall by more than half.  Published at tiles 7, 15 and every 16th, read back by four one-piece
//     LDS-DMAs.  The bootstrap launch publishes a bound that stands alone into all four slots.
//   * a list that could not take another tile (32 rows) is cut to its kprime best (key desc, id asc) by
//     the whole wave at the end of the tile; the packed value in kprime-th place becomes the list's floor
//     and its key the lane's threshold: tie-heavy or adversarially ordered corpora get here (random data
//     lists ~20 rows per list), and stay exact -- no row is ever lost, whatever the data order.
//   Everything else a finished tile needs (tracking, refresh, the room check) runs at the start of the wave's
//   next load phase, under the partner wave's MFMAs.
//   L2: the accumulators of a tile start from -|y|^2 / 2 of their corpus rows (C operand of the tile's first
//   MFMAs, read from a per-tile LDS copy), so an accumulator is h = x.y - |y|^2 / 2 = key / 2 and the kernel
//   filters and tracks in h; packed values and g_thr carry key = 2 h (exact).
//   At the end every list publishes its count and the bound "every row of mine that is not listed has
//   packed (key, id) <= bound"; knn_select.hip merges the lists of a query.
//
// What shaped the code (measured): the chip is POWER-limited in this loop (the clock
//   falls from 2.1 to 1.8 GHz when the LDS-DMA traffic is added to the MFMAs), so wall time follows energy,
//   not issue slots; with 128 accumulator registers live hipcc spills whatever else it is given, and a
//   spill reload costs a trip through the VMEM queue behind the DMA pieces (thousands of cycles): all
//   long-lived per-lane state is either recomputed from the lane id or parked in LDS by hand.
#include "knn_common.h"
#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace trx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) const void gbl_void;

constexpr int LDS_A0 = 0;
constexpr int LDS_B0 = 2 * TILE_M * 128;         // 65536
constexpr int S_THRW = LDS_B0 + 2 * TILE_N * 128;  // 131072: f32 [2 wave rows][256 queries] own threshold of a wave row
constexpr int S_GTHR = S_THRW + 2 * TILE_N * 4;  // u32 [4 slots][256] copy of g_thr
constexpr int S_E2 = S_GTHR + 4 * TILE_N * 4;    // f32 [2 wave rows][256] second-best tracked maximum of a wave row (what the split shares)
constexpr int S_BIAS = S_E2 + 2 * TILE_N * 4;    // f32 [2 tile parities][256 rows]  (L2)
constexpr int S_TRK = S_BIAS + 2 * TILE_M * 4;   // u32 [8][512 threads]: every lane's tracked tile maxima (see tile_end)
constexpr int S_SLK = S_TRK + 8 * SCAN_THREADS * 4;       // f32 [256 queries]: the listing slack of a query (ScanParams.slack), zeros without one
constexpr int LDS_TOTAL = S_SLK + TILE_N * 4;             // 158,720 B -> one workgroup per CU
// ds instruction offsets are 16-bit: the selection state is addressed relative to S_THRW
constexpr int R_THRW = 0, R_GTHR = S_GTHR - S_THRW, R_E2 = S_E2 - S_THRW, R_SLK = S_SLK - S_THRW;

// LDS accesses of the selection state go through asm: a C++ access to the array the LDS-DMA writes makes
// hipcc drain vmcnt to 0 (cdna_hip_programming.md section 5, "Three .s-level traps").  Addresses are
// LDS byte addresses (lds0 + offset).  Each read retires itself.
// The byte offset is an instruction immediate (OFF < 65536), so a lane needs one base register per array shape,
// not one address register per access (hipcc hoists those out of the tile loop and spills them).
template <int OFF> __device__ __forceinline__ u32 lds_ld32(u32 a) {
    u32 v;
    asm volatile("ds_read_b32 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a), "n"(OFF) : "memory");
    return v;
}
template <int OFF> __device__ __forceinline__ void lds_st32(u32 a, u32 v) { asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"(OFF) : "memory"); }
__device__ __forceinline__ u64 ld_u64_l2(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// XCD-contiguous bijective remap of the block id (blocks b and b+8 share an XCD under the
// observed round-robin placement; speed only, never correctness).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// diagnostic builds: -DTRX_ABL=1 never lists a row (sites stay), -DTRX_ABL=2 removes the sites too (results wrong)
#if defined(TRX_ABL) && TRX_ABL == 1
#define TRX_SITE_COND(M) (__builtin_expect((M) == 0x123456789abcdefull, 0))
#elif defined(TRX_ABL) && TRX_ABL == 2
#define TRX_SITE_COND(M) false
#else
#define TRX_SITE_COND(M) (__builtin_expect((M) != 0ull, 0))
#endif

// cache policy of the LDS-DMA streams (gfx940+ bits: 1 sc0, 2 nt, 16 sc1): A = corpus tiles (touched by the 8 workgroups
// of an XCD that share the split, then never again), B = query tiles (re-read for every corpus tile)
#ifndef TRX_CPOL_A
#define TRX_CPOL_A 0
#endif
#ifndef TRX_CPOL_B
#define TRX_CPOL_B 0
#endif

// the splits of a query publish their shared-threshold value every TRX_PUB_MASK + 1 tiles (a power of two >= 8: the
// thresholds themselves are refreshed every 8th tile) and read the slots back 4 tiles later
#ifndef TRX_PUB_MASK
#define TRX_PUB_MASK 15
#endif
#ifndef TRX_REFRESH_MASK
#define TRX_REFRESH_MASK 7      // thresholds refreshed every 8th tile (4th / 16th measured: see DESIGN.md 6.2)
#endif

template <int N> struct ic { static constexpr int value = N; };

// lane id, recomputed where it is needed (volatile: hipcc would otherwise hoist everything derived from the lane id out
// of the tile loop, keep it in registers the loop does not have, spill it, and reload it through the VMEM queue)
__device__ __forceinline__ u32 lane_now() {
    u32 l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// maximum of the 4 values of an accumulator tile in 2 instructions
__device__ __forceinline__ float max4f(const f32x4& a) {
    float g;
    asm("v_max_f32 %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4" : "=&v"(g) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
    return g;
}
// the same for the int8 form's int32 accumulators, converted once per group (the exact class: every sum is below 2^24)
__device__ __forceinline__ float max4f(const i32x4& a) {
    int g;
    asm("v_max_i32 %0, %1, %2\n\tv_max3_i32 %0, %0, %3, %4" : "=&v"(g) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
    return (float)g;
}
// f32 -> bf16 bits rounded toward -infinity (a tracked maximum may only get smaller: it stays a valid lower bound)
__device__ __forceinline__ u32 bf16_floor(float f) {
    const u32 u = __float_as_uint(f);
    return (u >> 16) + ((u >> 31) & ((u & 0xffffu) != 0u ? 1u : 0u));
}
// lanes with x >= y, straight into a scalar register pair
__device__ __forceinline__ u64 mask_ge(float x, float y) {
    u64 mk;
    asm("v_cmp_ge_f32 %0, %1, %2" : "=s"(mk) : "v"(x), "v"(y));
    return mk;
}

// store v to p in the lanes of `mask`, no branch (a skipped store costs a taken branch otherwise: ~30 cycles, and the
// slow path below is mostly branches)
__device__ __forceinline__ void store_masked(u64 mask, u64* p, u64 v) {
    u64 saved;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dwordx2 %2, %3, off\n\ts_mov_b64 exec, %0\n\ts_nop 0"
                 : "=&s"(saved) : "s"(mask), "v"(p), "v"(v) : "memory");
}

// A list (`cap` valid entries, all written by ONE lane of this wave) is cut to its kprime best by (key desc, id asc).
// Wave-wide; rare (see the header).  Returns the packed value in kprime-th place (0 = fewer than kprime rows were
// listed: nothing was dropped, the list gets no floor).  Entries with id >= n_valid are the pad rows of an
// inner-product index's last tile (score 0, above every row when the real scores are negative): they are not rows --
// they are dropped here, so that they can neither crowd real rows out of the kprime kept nor become the floor.
__device__ __forceinline__ u64 compact_list(u64* list, int cap, int kprime, int lane, u32 n_valid) {
    __builtin_amdgcn_s_waitcnt(0);     // this wave's own stores to the list have left
    u64 e[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        e[i] = (lane + 64 * i) < cap ? ld_u64_l2(list + lane + 64 * i) : 0ull;   // L2: never a stale L1 line
        if (comp_id(e[i]) >= n_valid) e[i] = 0ull;
    }
    u64 out = 0ull, last = 0ull;
    for (int t = 0; t < kprime; ++t) {
        u64 m = e[0] > e[1] ? e[0] : e[1];
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) { const u64 o = shfl_xor_u64(m, s); m = o > m ? o : m; }
        if (lane == t) out = m;
        last = m;
#pragma unroll
        for (int i = 0; i < 2; ++i) if (e[i] == m) e[i] = 0ull;     // packed values are unique (ids are); a row listed twice goes once
    }
    if (lane < kprime) list[lane] = out;
    return last;
}

// BOOT gives the threshold-bootstrap launch its own symbol, so that profiles list the two launches
// separately (the main scan's average duration is the roofline number).  RESCAN does the same for the launch that repeats
// the scan for a batch's still-uncertified queries with thresholds fixed at their seeds (knn_api.hip; usually no query: a
// launch of a few microseconds that halved the "average duration" of the scan in a kernel trace): the parameter changes
// nothing in the code.
// I8: the int8 form for the integer class (the reference's own workload: count fingerprints, retrieve_faiss.py:36-44).  The
// operands are int8 -- a 128-byte row of a K-step holds 128 components instead of 64, v_mfma_i32_16x16x64_i8 does twice the
// MACs of the bf16 instruction in the same time (tools/mfma_shape_lab: 4.73 Pop/s against 2.37 PFLOP/s on such data) -- the
// accumulators int32; L2 stages the DOUBLED query and starts from -|y|^2, so that an accumulator is the key itself
// (2 x.y - |y|^2, an integer).  The byte geometry of the loop is the bf16 form's (Kp counts 2-byte units); the filter takes
// a group's maximum in int32 and converts it once.  Whether a search may use it is decided on the device (p.gate).
// FMT 2, fp4: bit vectors and other tiny counts (every value one of 0, +-1, +-2, +-3, +-4, +-6: what E2M1 holds; the reference's
// Morgan fingerprints, retrieve_faiss.py:36-44, are 0 / 1).  v_mfma_scale_f32_16x16x128_f8f6f4 with fp4 operands and unit scales
// takes 128 components from the same 16-byte fragments in the time of the other two instructions (tools/mfma_shape_lab: 9.3
// PFLOP/s on such data against 4.7 int8 and 2.4 bf16); products and fp32 sums are exact, so the accumulators, the bias and
// the filter are the bf16 form's.  Which component a nibble of a fragment stands for is the hardware's business: corpus and
// query rows are packed by the same kernel, and a dot product does not care about a permutation applied to both sides.
template <bool L2, int J, bool BOOT, int NKS, bool RESCAN = false, int FMT = 0>
__global__ __launch_bounds__(SCAN_THREADS, 2) void knn_scan_kernel(ScanParams p) {
    // the exact class of a search is known on the device only (knn_prep.hip: classify_kernel): the bf16 and the int8 launch
    // are both enqueued (bootstrap and main scan alike), and the one whose turn it is not leaves here
    if (p.gate && *p.gate != p.gate_want) return;
    constexpr bool I8 = FMT == 1;      // (FMT 2, fp4: float accumulators and the bf16 form's arithmetic -- only the instruction differs)
    typedef typename std::conditional<I8, i32x4, f32x4>::type acc_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const u32 lds0 = (u32)(uintptr_t)(lds_char*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 2;  // 0..1 : group = corpus half of the tile
    const int wave_n = wave & 3;   // 0..3 : 64-query slice

    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int split = BOOT ? 0 : v % p.nsplits;
    const int qtile = BOOT ? v : v / p.nsplits;
    int tile0, tile1;
    if (BOOT) {   // a few tiles, spread over the corpus so that query tiles do not collide
        const int nb = p.boot_tiles < p.ntiles ? p.boot_tiles : p.ntiles;
        const int span = p.ntiles - nb + 1;
        tile0 = (int)(((long long)qtile * 97) % span);
        tile1 = tile0 + nb;
    } else {
        tile0 = split * p.tiles_per_split;
        tile1 = tile0 + p.tiles_per_split;
        if (tile1 > p.ntiles) tile1 = p.ntiles;
    }
    const int ntl = tile1 > tile0 ? tile1 - tile0 : 0;
    const int64_t qbase = (int64_t)qtile * TILE_N;
    // the number of queries may live on the device (the re-scan of a batch's uncertified queries is enqueued before anybody
    // knows how many there are): query tiles beyond it have nothing to do, and nobody reads their lists
    int nq_valid = p.nq_valid;
    if (!BOOT && p.nq_valid_dev) { const int nd = *p.nq_valid_dev; nq_valid = nd < nq_valid ? nd : nq_valid; if (qbase >= nq_valid) return; }
    // NKS: the number of K-steps when it is known at compile time (12 = 768 components, BERT's width and the headline
    // case: the K loop of a tile unrolls and the DMA cursors' wrap tests fold, 71.3 -> 70.6 ms), 0 = read it from Kp
    const int ksteps = NKS ? NKS : p.Kp / BK;   // even, >= 4
    const int Kp = p.Kp;

    // ---- staging geometry (LDS-DMA): a piece is one global_load_lds_dwordx4 = 8 rows x 128 B, written
    // lane-linear (row 8 * piece + lane / 8, slot lane % 8); slot s of row r receives global chunk
    // s ^ ((r >> 1) & 7).  In a load phase a wave moves pieces 4 wave_n .. 4 wave_n + 3 of one 128-row half.
    const int prow = lane >> 3, pslot = lane & 7;
    const int c_even = pslot ^ (prow >> 1);        // pieces with (piece & 1) == 0
    const int c_odd = pslot ^ (4 + (prow >> 1));
    // per-lane BYTE offsets of this wave's pieces inside a 128-row half: piece i = rows 32 wave_n + 8 i + prow; pieces
    // i and i + 2 differ by 16 rows (a wave-uniform amount), pieces i and i + 1 by 8 rows and the swizzle phase.
    // Kept as two unsigned 32-bit offsets against wave-uniform base pointers (global_load_lds saddr + voffset form).
    const u32 po0 = (u32)(((wave_n * 4 + 0) * 8 + prow) * Kp + c_even * 8) * 2u;
    const u32 po1 = (u32)(((wave_n * 4 + 1) * 8 + prow) * Kp + c_odd * 8) * 2u;
    const int pstep = 16 * Kp * 2;      // bytes between pieces i and i + 2
    const int half_elems = 128 * Kp;
    const bf16_t* gA = p.corpus + (int64_t)tile0 * TILE_M * Kp;
#ifdef TRX_SCAN_DEBUG_BUILD
    const bf16_t* gB = p.queries + qbase * Kp;
#else
    const bf16_t* const gB = p.queries + qbase * Kp;
#endif
    const int lds_piece0 = wave_n * 4 * 1024;

    // ---- fragment read geometry ----
    const int frow = lane & 15, fq = lane >> 4;
    const int swz = frow >> 1;
    const int r_off0 = frow * 128 + ((fq ^ swz) << 4);
    const int r_off1 = frow * 128 + (((4 + fq) ^ swz) << 4);
    const u32 aA0 = lds0 + LDS_A0 + wave_m * 128 * 128 + r_off0, aA1 = lds0 + LDS_A0 + wave_m * 128 * 128 + r_off1;
    const u32 aB0 = lds0 + LDS_B0 + wave_n * 64 * 128 + r_off0, aB1 = lds0 + LDS_B0 + wave_n * 64 * 128 + r_off1;

    // ---- selection state ----
    const int ql0 = wave_n * 64 + frow;           // query (inside the tile) of accumulator column nt: ql0 + 16 nt
    const float NEG_INF = -__builtin_inff();
    constexpr float KS = (L2 && !I8) ? 2.0f : 1.0f, KI = (L2 && !I8) ? 0.5f : 1.0f;   // key = KS * accumulator (see the header, L2; int8: the accumulator IS the key)
    // per-lane LDS bases of the selection state: [wave row][query] arrays of 4- and 8-byte entries; own row,
    // partner's row, row 0 (arrays without a wave-row dimension use b4_0)
    const u32 b4_0 = lds0 + S_THRW + ql0 * 4;
    const u32 b4_m = b4_0 + wave_m * 1024, b4_p = b4_0 + (wave_m ^ 1) * 1024;
    {   // tracked maxima of this thread: -inf (fp32, J = 2) / a pair of bf16 -inf (J = 4)
        const u32 a_trk = lds0 + S_TRK + tid * 4;
        const u32 ninf = J == 2 ? 0xff800000u : 0xff80ff80u;
        lds_st32<0>(a_trk, ninf); lds_st32<2048>(a_trk, ninf); lds_st32<4096>(a_trk, ninf); lds_st32<6144>(a_trk, ninf);
        lds_st32<8192>(a_trk, ninf); lds_st32<10240>(a_trk, ninf); lds_st32<12288>(a_trk, ninf); lds_st32<14336>(a_trk, ninf);
    }
    {   // g_thr: [query tile][4 slots][256 queries]
        const u32 t4 = lds0 + S_THRW + tid * 4;
        const u32* gsrc = p.g_thr + qbase * 4;
        lds_st32<R_GTHR>(t4, BOOT ? 0u : __hip_atomic_load(gsrc + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        lds_st32<R_GTHR + 2048>(t4, BOOT ? 0u : __hip_atomic_load(gsrc + 512 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        lds_st32<R_THRW>(t4, __float_as_uint(NEG_INF));      // 512 threads: both wave rows
        lds_st32<R_E2>(t4, __float_as_uint(NEG_INF));
        // the listing slack of this tile's queries (approximate operands: knn_api.hip, the approx mode): a query's threshold
        // is the bound its tracked maxima give MINUS this, so that every row within twice the key error of the bound is
        // listed and the candidates can be certified without a second scan; shared bounds (g_thr) stay raw
        if (tid < TILE_N) lds_st32<R_SLK>(t4, (!BOOT && p.slack) ? __float_as_uint(p.slack[qbase + tid]) : 0u);
    }

    acc_t acc[8][4];
    const acc_t zero4 = {0, 0, 0, 0};
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = zero4;
    if (ntl == 0) {
        // nothing to scan (only possible for an empty split): publish empty lists
        if (!BOOT) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int64_t o = (((qbase + ql0 + 16 * nt) * p.nsplits + split) * 2 + wave_m) * 4 + fq;
                p.cand_cnt[o] = 0u; p.cand_thr[o] = 0ull;
            }
        }
        return;
    }

    // ---- prologue: K-step 0 in full and A rows 0-127 of K-step 1 (all waves), bias of tile 0 ----
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pc = wave * 4 + i;    // piece 0..31 of a 256-row operand stage
            const int off = (pc * 8 + prow) * Kp + ((pc & 1) ? c_odd : c_even) * 8;
            __builtin_amdgcn_global_load_lds((gbl_void*)(gA + off), (lds_void*)(smem + LDS_A0 + pc * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(gB + off), (lds_void*)(smem + LDS_B0 + pc * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pc = wave * 2 + i;    // piece 0..15: rows 0-127
            const int off = (pc * 8 + prow) * Kp + ((pc & 1) ? c_odd : c_even) * 8;
            __builtin_amdgcn_global_load_lds((gbl_void*)(gA + off + BK), (lds_void*)(smem + LDS_A0 + 32768 + pc * 1024), 16, 0, 0);
        }
        if (L2 && wave == 2)
            __builtin_amdgcn_global_load_lds((gbl_void*)(p.cbias + (int64_t)tile0 * TILE_M + lane * 4), (lds_void*)(smem + S_BIAS), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // thresholds of this lane's four queries (identical in the 4 lanes of a query)
    float thrk[4], m[4];
    u32 cnt4 = 0u;      // this lane's four list counters, one byte each: bits 0-6 count, bit 7 = list was compacted (has a floor)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        thrk[nt] = NEG_INF;
        m[nt] = NEG_INF;
    }

    {
        auto seed = [&](auto NT) {      // the smallest of the four slots (0 = a slot nobody has published to yet)
            constexpr int nt = decltype(NT)::value;
            const u32 g01 = min(lds_ld32<R_GTHR + 64 * nt>(b4_0), lds_ld32<R_GTHR + 1024 + 64 * nt>(b4_0));
            const u32 g23 = min(lds_ld32<R_GTHR + 2048 + 64 * nt>(b4_0), lds_ld32<R_GTHR + 3072 + 64 * nt>(b4_0));
            const u32 g = min(g01, g23);
            if (g) thrk[nt] = KI * ordkey_inv(g) - __uint_as_float(lds_ld32<R_SLK + 64 * nt>(b4_0));
        };
        seed(ic<0>{}); seed(ic<1>{}); seed(ic<2>{}); seed(ic<3>{});
        // the pad rows of the last query tile (zero vectors) list nothing: under the inner product every corpus row scores
        // exactly 0 against them -- one tie group of the whole corpus, whose lists fill and are compacted at every tile
        // (a 40,000-query search, 157 tiles with 192 pad rows, took 190 ms instead of 12).  Thresholds only rise, so +inf stays.
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            if (qbase + ql0 + 16 * nt >= nq_valid) thrk[nt] = __builtin_inff();
    }

    // DMA cursors (wave-uniform).  B stream: K-steps 1, 2, ... (columns wrap per tile).  A stream: group 0
    // stages rows 128-255 of K-steps 1, 2, ...; group 1 rows 0-127 of K-steps 2, 3, ...
    const bf16_t* srcB = gB + (wave_m ? half_elems : 0);
    int ksB = 1;
    const bf16_t* srcA = gA + (wave_m ? 0 : half_elems) + (wave_m ? 2 : 1) * BK;
    int ksA = wave_m ? 2 : 1;
    if (ksA >= ksteps) { ksA -= ksteps; srcA += 255 * Kp; }     // Kp == 128: K-step 2 is the next tile's K-step 0
#ifdef TRX_SCAN_DEBUG_BUILD
    const int wrapA = (p.debug & 32) ? -Kp : 255 * Kp;
#else
    const int wrapA = 255 * Kp;       // added when a K-step cursor moves on to the next tile
#endif
#ifdef TRX_SCAN_DEBUG_BUILD   // timing-only switches (TRX_SCAN_DEBUG bits 1, 2, 8; results are wrong): compiled in on request
    const bool dbg_nodma = (p.debug & 1) != 0;
    const bool dbg_nofilter = (p.debug & 2) != 0;
    const bool dbg_norefresh = (p.debug & 8) != 0;
    // pricing the L2 misses (round 5): bit 16 -- every workgroup stages query tile 0 (the XCD's query footprint is one tile);
    // bit 32 -- the corpus stream re-reads the first tile of its split for ever (its footprint is one tile per split)
    if (p.debug & 16) { gB = p.queries; srcB = gB + (wave_m ? half_elems : 0); }
#else
    constexpr bool dbg_nodma = false, dbg_nofilter = false, dbg_norefresh = false;
#endif

    if (wave_m) __builtin_amdgcn_s_barrier();       // group 1 runs one interval late

    bf16x8 fa[8], fb[4];
    acc_t biasv[8];     // L2: -|y|^2 / 2 (int8: -|y|^2, int32) of this lane's rows of the tile that is about to start
    const u32 a_bias = lds0 + S_BIAS + (wave_m * 128 + fq * 4) * 4;
#define TRX_READ_BIAS()                                                                                    \
    if (L2) {                                                                                              \
        _Pragma("unroll") for (int mt_ = 0; mt_ < 8; ++mt_)                                                \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(biasv[mt_]) : "v"(a_bias + (tl & 1) * 1024), "n"(mt_ * 64) : "memory"); \
    }
#define TRX_READ(KK, STG)                                                                                  \
    {                                                                                                      \
        _Pragma("unroll") for (int nt_ = 0; nt_ < 4; ++nt_)                                                \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[nt_]) : "v"((KK) ? aB1 : aB0), "n"(nt_ * 2048 + (STG) * 32768) : "memory"); \
        _Pragma("unroll") for (int mt_ = 0; mt_ < 8; ++mt_)                                                \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[mt_]) : "v"((KK) ? aA1 : aA0), "n"(mt_ * 2048 + (STG) * 32768) : "memory"); \
    }
    // Diagnostic build (make stamp): where an interval's cycles go, per wave.  s_memtime stamps at the five edges of a wave's
    // L -> barrier -> M -> barrier round; they are SMEM results, so they are only READ behind the next load phase's own
    // lgkmcnt(0) (no wait is added anywhere): wt = the counter waits at the end of a load phase (LDS fragment reads + the
    // previous phase's DMA pieces), bl = the barrier behind them, mf = the MFMA phase, bm = the barrier behind it, ld = the
    // issue part of the load phase (fragment reads, DMA, a finished tile's bookkeeping).
#ifdef TRX_STAMP_BUILD
    unsigned long long tA_ = 0, tB_ = 0, tC_ = 0, tD_ = 0, tE_ = 0;
    u32 a_wt = 0, a_bl = 0, a_mf = 0, a_bm = 0, a_ld = 0, a_n = 0;      // (cycles of one wave over one launch fit 32 bits)
#define TRX_T(X) asm volatile("s_memtime %0" : "=s"(X)::"memory")
#define TRX_ROUND_ACC()                                                                                    \
    if (tE_ > tD_ && tD_ > tC_ && tC_ > tB_ && tB_ > tA_) {                                                \
        a_wt += (u32)(tB_ - tA_); a_bl += (u32)(tC_ - tB_); a_mf += (u32)(tD_ - tC_); a_bm += (u32)(tE_ - tD_); ++a_n; \
    }
#else
#define TRX_T(X)
#define TRX_ROUND_ACC()
#endif
    // end of a load phase; NV = VMEM operations this wave may leave in flight (4, or 5 with an extra piece)
#ifdef TRX_STAMP_BUILD
#define TRX_WAIT_L(NV)                                                                                     \
    {                                                                                                      \
        unsigned long long tN_;                                                                            \
        TRX_T(tN_);                                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(" #NV ")" ::: "memory");                      \
        TRX_ROUND_ACC();        /* the previous round's stamps have landed (lgkmcnt(0)) */                 \
        if (tE_ && tN_ > tE_) a_ld += (u32)(tN_ - tE_);                                                         \
        tA_ = tN_;                                                                                         \
        TRX_T(tB_);                                                                                        \
    }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    __builtin_amdgcn_s_barrier();                                                                          \
    TRX_T(tC_);                                                                                            \
    __builtin_amdgcn_sched_barrier(0);
#define TRX_END_M()                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    TRX_T(tD_);                                                                                            \
    __builtin_amdgcn_s_barrier();                                                                          \
    TRX_T(tE_);                                                                                            \
    __builtin_amdgcn_sched_barrier(0);
#else
#define TRX_WAIT_L(NV)                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(" #NV ")" ::: "memory");                          \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    __builtin_amdgcn_s_barrier();                                                                          \
    __builtin_amdgcn_sched_barrier(0);
#define TRX_END_M()                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    __builtin_amdgcn_s_barrier();                                                                          \
    __builtin_amdgcn_sched_barrier(0);
#endif
// The query block index runs back and forth over consecutive rows of MFMAs (0 1 2 3 | 3 2 1 0 | ...): one operand changes per
// instruction instead of two at every row end.  Lab (tools/scan_lab.hip -DLAB_SNAKE=1): MFMAs alone -0.9 ... -1.6 %, whole loop
// -0.5 ... -1.0 %.
#ifndef TRX_SNAKE
#define TRX_SNAKE 1
#endif
// (the instruction reads four registers of an fp4 operand; the builtin's eight-register type gets an UNDEFINED upper half, which
// the instruction selection drops -- zeros there cost four more live registers per fragment and 500 bytes of scratch)
#define TRX_F4OP(X) __builtin_shufflevector(__builtin_bit_cast(i32x4, X), __builtin_bit_cast(i32x4, X), 0, 1, 2, 3, -1, -1, -1, -1)
#define TRX_MFMA1(A, B, C) (FMT == 1 ? (acc_t)__builtin_bit_cast(acc_t, __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, A), __builtin_bit_cast(i32x4, B), __builtin_bit_cast(i32x4, C), 0, 0, 0)) \
                  : FMT == 2 ? (acc_t)__builtin_bit_cast(acc_t, __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(TRX_F4OP(A), TRX_F4OP(B), __builtin_bit_cast(f32x4, C), 4, 4, 0, 0, 0, 0)) \
                              : (acc_t)__builtin_bit_cast(acc_t, __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, __builtin_bit_cast(f32x4, C), 0, 0, 0)))
#define TRX_MFMA_ACC()                                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    _Pragma("unroll") for (int mt_ = 0; mt_ < 8; ++mt_)                                                    \
        _Pragma("unroll") for (int n0_ = 0; n0_ < 4; ++n0_) {                                              \
            const int nt_ = (TRX_SNAKE && (mt_ & 1)) ? 3 - n0_ : n0_;                                      \
            acc[mt_][nt_] = TRX_MFMA1(fa[mt_], fb[nt_], acc[mt_][nt_]);                                    \
        }                                                                                                  \
    __builtin_amdgcn_s_setprio(0);
    // first phase of a tile: the accumulators start from zero (IP) or from -|y|^2 / 2 of their rows (L2): no clearing pass
#define TRX_MFMA_ZERO()                                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    _Pragma("unroll") for (int mt_ = 0; mt_ < 8; ++mt_)                                                    \
        _Pragma("unroll") for (int n0_ = 0; n0_ < 4; ++n0_) {                                              \
            const int nt_ = (TRX_SNAKE && (mt_ & 1)) ? 3 - n0_ : n0_;                                      \
            acc[mt_][nt_] = TRX_MFMA1(fa[mt_], fb[nt_], (L2 ? biasv[mt_] : zero4));                        \
        }                                                                                                  \
    __builtin_amdgcn_s_setprio(0);
    // B pieces of this wave's half for the K-step after the current one -> stage STG
#define TRX_DMA_B(STG)                                                                                     \
    if (!dbg_nodma) {                                                                                      \
        const char* s_ = (const char*)(srcB + ksB * BK);                                                   \
        char* l_ = smem + LDS_B0 + (STG) * 32768 + wave_m * 16384 + lds_piece0;                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                   \
            __builtin_amdgcn_global_load_lds((gbl_void*)(s_ + (i_ >> 1) * pstep + ((i_ & 1) ? po1 : po0)), (lds_void*)(l_ + i_ * 1024), 16, 0, TRX_CPOL_B); \
    }                                                                                                      \
    ksB = (ksB + 1 == ksteps) ? 0 : ksB + 1;
    // A pieces: group 0 -> rows 128-255, group 1 -> rows 0-127, of the cursor's K-step -> stage STG
#define TRX_DMA_A(STG)                                                                                     \
    if (!dbg_nodma) {                                                                                      \
        char* l_ = smem + LDS_A0 + (STG) * 32768 + (wave_m ? 0 : 16384) + lds_piece0;                      \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                   \
            __builtin_amdgcn_global_load_lds((gbl_void*)((const char*)srcA + (i_ >> 1) * pstep + ((i_ & 1) ? po1 : po0)), (lds_void*)(l_ + i_ * 1024), 16, 0, TRX_CPOL_A); \
    }                                                                                                      \
    srcA += BK;                                                                                            \
    if (++ksA == ksteps) { ksA = 0; srcA += wrapA; }

    // ---- what happens once per finished tile, at the start of the wave's next load phase (its partner is
    // issuing MFMAs meanwhile): the room check of the lists, the tracked maxima, the threshold refresh.
    // TL = tile index inside the split whose per-lane maxima are in `m`.
#ifdef TRX_STAMP_BUILD
    unsigned long long st_cyc = 0, st_comp = 0;     // diagnostic build: cycles in here, lists compacted
#endif
    auto tile_end = [&](const int TL) __attribute__((always_inline)) {
#ifdef TRX_STAMP_BUILD
        const unsigned long long st_t0 = __builtin_readcyclecounter();
#endif
        const int tile_row0 = (tile0 + TL) * TILE_M;
        const bool full_tile = L2 || tile_row0 + TILE_M <= p.n_valid;    // IP pad rows score 0: they are not rows
        // lane-derived values, recomputed here (see lane_now): they shadow the kernel-scope ones
        const int lane = (int)lane_now();
        const int frow = lane & 15, fq = lane >> 4;
        const int ql0 = wave_n * 64 + frow;
        const u32 b4_0 = lds0 + S_THRW + ql0 * 4;
        const u32 b4_m = b4_0 + wave_m * 1024, b4_p = b4_0 + (wave_m ^ 1) * 1024;
        // ---- room for the next tile: a lane lists at most 32 rows of a tile per column.  A list that could overflow is
        // cut to its kprime best (key desc, id asc) NOW, by the whole wave; its lane's threshold rises to the key in
        // kprime-th place and the packed value in kprime-th place becomes the list's floor (cand_thr).  Rare: ~20 rows
        // are listed per list on random data (tie-heavy or adversarially ordered corpora get here, and stay exact).
        if (!BOOT) {
            const u32 lim = (u32)(p.cap - 32);
            bool need = false;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) need = need || (((cnt4 >> (8 * nt)) & 0x7fu) > lim);
            if (__builtin_expect(__any(need), 0)) {
                const int64_t li0 = (((qbase + ql0) * p.nsplits + split) * 2 + wave_m) * 4 + fq;
                const int64_t colstride = (int64_t)16 * p.nsplits * LISTS_PER_SPLIT;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    u64 om = __ballot(((cnt4 >> (8 * nt)) & 0x7fu) > lim);
                    while (om) {
                        const int l = __ffsll((long long)om) - 1;
                        om &= om - 1ull;
                        const int64_t li = li0 + nt * colstride;
                        const int64_t lsel = (int64_t)shfl_u64((u64)li, l);
                        const int ncur = (int)__shfl((int)((cnt4 >> (8 * nt)) & 0x7fu), l, 64);
                        const u64 floor_c = compact_list(p.cand + lsel * p.cap_alloc, ncur, p.kprime, lane, (u32)p.n_valid);
#ifdef TRX_STAMP_BUILD
                        ++st_comp;
#endif
                        if (lane == l) {
                            cnt4 = (cnt4 & ~(0xffu << (8 * nt))) | ((u32)(p.kprime | 0x80) << (8 * nt));
                            p.cand_thr[li] = floor_c;
                            if (floor_c) thrk[nt] = __builtin_fmaxf(thrk[nt], KI * comp_key(floor_c));
                        }
                    }
                }
                __builtin_amdgcn_s_waitcnt(0);
            }
        }
        // ---- the J best tile maxima of this lane (pad-row tiles of an inner-product index do not count).  They live in
        // LDS between tiles (8 dwords per thread; J = 4: sixteen values as bf16 pairs, rounded DOWN, so that they stay
        // lower bounds): as registers hipcc spilled them, and a spill reload waits out the whole VMEM queue.
        float trk[J][4];
        {
            const u32 a_trk = lds0 + S_TRK + (u32)(wave * 64 + lane) * 4;
            u32 w[8];
            asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:2048\n\tds_read_b32 %2, %8 offset:4096\n\tds_read_b32 %3, %8 offset:6144\n\t"
                         "ds_read_b32 %4, %8 offset:8192\n\tds_read_b32 %5, %8 offset:10240\n\tds_read_b32 %6, %8 offset:12288\n\tds_read_b32 %7, %8 offset:14336\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
                         : "v"(a_trk) : "memory");
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                if (J == 2) { trk[0][nt] = __uint_as_float(w[nt]); trk[J - 1][nt] = __uint_as_float(w[4 + nt]); }
                else {
#pragma unroll
                    for (int j = 0; j < J; ++j) trk[j][nt] = __uint_as_float(((j & 1) ? (w[(j >> 1) * 4 + nt] & 0xffff0000u) : (w[(j >> 1) * 4 + nt] << 16)));
                }
            }
            if (full_tile) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    float a = m[nt];
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        const float hi = __builtin_fmaxf(trk[j][nt], a);
                        a = __builtin_fminf(trk[j][nt], a);
                        trk[j][nt] = hi;
                    }
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    if (J == 2) { w[nt] = __float_as_uint(trk[0][nt]); w[4 + nt] = __float_as_uint(trk[J - 1][nt]); }
                    else {
#pragma unroll
                        for (int jp = 0; jp < J / 2; ++jp) {
                            const u32 lo = bf16_floor(trk[2 * jp][nt]), hi = bf16_floor(trk[2 * jp + 1][nt]);
                            w[jp * 4 + nt] = lo | (hi << 16);
                            trk[2 * jp][nt] = __uint_as_float(lo << 16); trk[2 * jp + 1][nt] = __uint_as_float(hi << 16);
                        }
                    }
                }
                lds_st32<0>(a_trk, w[0]); lds_st32<2048>(a_trk, w[1]); lds_st32<4096>(a_trk, w[2]); lds_st32<6144>(a_trk, w[3]);
                if (J == 2) { lds_st32<8192>(a_trk, w[4]); lds_st32<10240>(a_trk, w[5]); lds_st32<12288>(a_trk, w[6]); lds_st32<14336>(a_trk, w[7]); }
                else { lds_st32<8192>(a_trk, w[4]); lds_st32<10240>(a_trk, w[5]); lds_st32<12288>(a_trk, w[6]); lds_st32<14336>(a_trk, w[7]); }
            }
        }
        // ---- threshold refresh (every eighth tile): min over the query's 4 lanes of this wave, then the partner wave
        // row's value and the other splits' through LDS.  All LDS traffic of the 4 columns is issued together and
        // waited for once (one access at a time cost 8 ms per search).
        // (fixed_thr: the re-scan of uncertified queries keeps every threshold at its seed -- a key below which no row can reach the
        // query's top k -- so that ALL rows above it end up listed; a refresh would raise it to the k'-th best again)
        if (!dbg_norefresh && !p.fixed_thr && (BOOT ? TL == ntl - 1 : (TL & TRX_REFRESH_MASK) == TRX_REFRESH_MASK)) {      // bootstrap launch: once, for its final publish
            float g[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) g[nt] = trk[J - 1][nt];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) g[nt] = __builtin_fminf(g[nt], __shfl_xor(g[nt], 16, 64));
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) g[nt] = __builtin_fminf(g[nt], __shfl_xor(g[nt], 32, 64));
            if (fq == 0) {
                lds_st32<R_THRW + 0>(b4_m, __float_as_uint(g[0])); lds_st32<R_THRW + 64>(b4_m, __float_as_uint(g[1]));
                lds_st32<R_THRW + 128>(b4_m, __float_as_uint(g[2])); lds_st32<R_THRW + 192>(b4_m, __float_as_uint(g[3]));
            }
            u32 gp[4], gs[4][4];
            asm volatile("ds_read_b32 %0, %4 offset:%5\n\tds_read_b32 %1, %4 offset:%6\n\tds_read_b32 %2, %4 offset:%7\n\tds_read_b32 %3, %4 offset:%8"
                         : "=&v"(gp[0]), "=&v"(gp[1]), "=&v"(gp[2]), "=&v"(gp[3])
                         : "v"(b4_p), "n"(R_THRW), "n"(R_THRW + 64), "n"(R_THRW + 128), "n"(R_THRW + 192) : "memory");
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
                asm volatile("ds_read_b32 %0, %4 offset:%5\n\tds_read_b32 %1, %4 offset:%6\n\tds_read_b32 %2, %4 offset:%7\n\tds_read_b32 %3, %4 offset:%8"
                             : "=&v"(gs[sl][0]), "=&v"(gs[sl][1]), "=&v"(gs[sl][2]), "=&v"(gs[sl][3])
                             : "v"(b4_0 + sl * 1024), "n"(R_GTHR), "n"(R_GTHR + 64), "n"(R_GTHR + 128), "n"(R_GTHR + 192) : "memory");
            u32 sk[4];      // the queries' listing slack (zeros unless the operands are approximate): a threshold = a bound minus it
            // (the reads and their wait in ONE statement, and the wait names the earlier reads' registers: a wait that names
            // nothing orders nothing -- the first form of this had its subtraction scheduled in front of the wait, a quarter of
            // a batch's queries got a threshold off by whatever the register held and went through the re-scan)
            asm volatile("ds_read_b32 %0, %4 offset:%5\n\tds_read_b32 %1, %4 offset:%6\n\tds_read_b32 %2, %4 offset:%7\n\tds_read_b32 %3, %4 offset:%8\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(sk[0]), "=&v"(sk[1]), "=&v"(sk[2]), "=&v"(sk[3])
                         : "v"(b4_0), "n"(R_SLK), "n"(R_SLK + 64), "n"(R_SLK + 128), "n"(R_SLK + 192) : "memory");
            asm volatile("" : "+v"(gp[0]), "+v"(gp[1]), "+v"(gp[2]), "+v"(gp[3]));
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) asm volatile("" : "+v"(gs[sl][0]), "+v"(gs[sl][1]), "+v"(gs[sl][2]), "+v"(gs[sl][3]));
            const bool publish = !BOOT && ((TL & TRX_PUB_MASK) == TRX_PUB_MASK || TL == 7);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float slk = __uint_as_float(sk[nt]);
                const float both = __builtin_fminf(g[nt], __uint_as_float(gp[nt]));      // 8 J rows of this split reach this key
                float t = __builtin_fmaxf(thrk[nt], both - slk);
                const u32 gm = min(min(gs[0][nt], gs[1][nt]), min(gs[2][nt], gs[3][nt])); // kprime rows of the corpus reach this one
                if (gm) t = __builtin_fmaxf(t, KI * ordkey_inv(gm) - slk);
                thrk[nt] = t;
            }
            if (publish) {
                // what this split tells the others: the J-th best of the 4 J tracked maxima of this wave row (4 lanes x their
                // J best, each sorted), then the smaller of the two wave rows' values (the partner's may be one publish old:
                // lower, still valid): J + J = kprime / 4 rows of this split reach it
                float e2[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    if (J == 2) {
                        const float u0 = __shfl_xor(trk[0][nt], 16, 64), u1 = __shfl_xor(trk[1][nt], 16, 64);
                        const float c0 = __builtin_fmaxf(trk[0][nt], u0);
                        const float c1 = __builtin_fmaxf(__builtin_fminf(trk[0][nt], u0), __builtin_fmaxf(trk[1][nt], u1));
                        const float d0 = __shfl_xor(c0, 32, 64), d1 = __shfl_xor(c1, 32, 64);
                        e2[nt] = __builtin_fmaxf(__builtin_fminf(c0, d0), __builtin_fmaxf(c1, d1));
                    } else {
                        // two sorted lists of 4: the 4 largest of their union are max(a_i, b_(3-i)) -- a bitonic sequence,
                        // sorted by two rounds of compare-exchange; the second merge only needs its smallest element
                        float c[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) c[i] = __builtin_fmaxf(trk[i][nt], __shfl_xor(trk[3 - i][nt], 16, 64));
                        const float x0 = __builtin_fmaxf(c[0], c[2]), x2 = __builtin_fminf(c[0], c[2]);
                        const float x1 = __builtin_fmaxf(c[1], c[3]), x3 = __builtin_fminf(c[1], c[3]);
                        c[0] = __builtin_fmaxf(x0, x1); c[1] = __builtin_fminf(x0, x1);
                        c[2] = __builtin_fmaxf(x2, x3); c[3] = __builtin_fminf(x2, x3);
                        float e = __builtin_fmaxf(c[0], __shfl_xor(c[3], 32, 64));
#pragma unroll
                        for (int i = 1; i < 4; ++i) e = __builtin_fminf(e, __builtin_fmaxf(c[i], __shfl_xor(c[3 - i], 32, 64)));
                        e2[nt] = e;
                    }
                }
                if (fq == 0) {
                    lds_st32<R_E2 + 0>(b4_m, __float_as_uint(e2[0])); lds_st32<R_E2 + 64>(b4_m, __float_as_uint(e2[1]));
                    lds_st32<R_E2 + 128>(b4_m, __float_as_uint(e2[2])); lds_st32<R_E2 + 192>(b4_m, __float_as_uint(e2[3]));
                    u32 ep[4];
                    asm volatile("ds_read_b32 %0, %4 offset:%5\n\tds_read_b32 %1, %4 offset:%6\n\tds_read_b32 %2, %4 offset:%7\n\tds_read_b32 %3, %4 offset:%8\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(ep[0]), "=&v"(ep[1]), "=&v"(ep[2]), "=&v"(ep[3])
                                 : "v"(b4_p), "n"(R_E2), "n"(R_E2 + 64), "n"(R_E2 + 128), "n"(R_E2 + 192) : "memory");
                    const int slot = split & 3;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const float v4 = __builtin_fminf(e2[nt], __uint_as_float(ep[nt]));
                        if (v4 > NEG_INF) __hip_atomic_fetch_max(p.g_thr + (qbase * 4 + slot * 256 + ql0 + 16 * nt), ordkey(KS * v4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
#ifdef TRX_STAMP_BUILD
        st_cyc += __builtin_readcyclecounter() - st_t0;
#endif
    };

    // One pair of K-steps (stage 0, then stage 1).  FIRST: first pair of a tile -- the previous tile's bookkeeping
    // rides in its first load phase and its first MFMA phase starts the accumulators from zero.  LAST: last pair of a
    // tile -- its last MFMA phase takes the per-lane maxima in the MFMA gaps.  The three forms are separate
    // straight-line blocks (a tile = FIRST, middle pairs, LAST; Kp >= 256 so FIRST != LAST): as branches inside one
    // block they made hipcc spill half the accumulators.
    // first load phase of a tile's first pair: previous tile's bookkeeping + extra one-piece DMAs in front of the four
    // regular pieces (the other splits' thresholds every 8 tiles, wave 1; the next tile's bias, L2, wave 2)
#define TRX_PAIR_HEAD()                                                                                    \
    {                                                                                                      \
        const bool aux_g = !BOOT && (wave >> 2) == 1 && ((tl & TRX_PUB_MASK) == 3 || tl == 11);            \
        const bool aux_b = L2 && wave == 2;                                                                \
        const bool strict = tl > 0 && !dbg_nofilter && !BOOT && (((tl - 1) & TRX_PUB_MASK) == TRX_PUB_MASK || tl == 8); \
        if (strict) {                                                                                      \
            /* bookkeeping that publishes thresholds to g_thr issues atomics, which count in vmcnt like the DMA    \
               pieces: retire the previous load phase's pieces first (two intervals old) and wait for nothing at   \
               the end; the next load phase's vmcnt(4) covers everything issued here */                      \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
        }                                                                                                  \
        TRX_READ(0, 0);                                                                                    \
        TRX_READ_BIAS();                                                                                   \
        if (aux_g) __builtin_amdgcn_global_load_lds((gbl_void*)(p.g_thr + (qbase * 4 + (wave & 3) * 256 + lane * 4)), (lds_void*)(smem + S_GTHR + (wave & 3) * 1024), 16, 0, 0); \
        /* (the bias of the NEXT tile; clamped to the spare tile behind the last one: the extra head after a split's last  \
           tile asked for the tile after the spare one -- 1 KiB past the array, a fault whenever the array ended on the    \
           last mapped page: round 4's fuzzer found the layout) */                                        \
        if (aux_b) __builtin_amdgcn_global_load_lds((gbl_void*)(p.cbias + (int64_t)min(tile0 + tl + 1, p.ntiles) * TILE_M + lane * 4), \
                                                    (lds_void*)(smem + S_BIAS + ((tl + 1) & 1) * 1024), 16, 0, 0); \
        TRX_DMA_B(1);                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        if (tl > 0 && !dbg_nofilter) {                                                                     \
            /* the partner wave on this SIMD is issuing MFMAs meanwhile; the fragment reads above are in flight */ \
            tile_end(tl - 1);                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                             \
        }                                                                                                  \
        if (strict) { TRX_WAIT_L(63); } else if (aux_g || aux_b) { TRX_WAIT_L(5); } else { TRX_WAIT_L(4); } \
    }
#define TRX_PAIR_HEAD_PLAIN()                                                                              \
    {                                                                                                      \
        TRX_READ(0, 0);                                                                                    \
        TRX_DMA_B(1);                                                                                      \
        TRX_WAIT_L(4);                                                                                     \
    }
#define TRX_PAIR_REST(FIRST, LAST)                                                                         \
    {                                                                                                      \
        /* ---- M(u, 0) ---- */                                                                            \
        if (FIRST) { TRX_MFMA_ZERO(); } else { TRX_MFMA_ACC(); }                                           \
        TRX_END_M();                                                                                       \
        /* ---- L(u, 1) ----  G1: A rows 0-127 of u+2 -> stage 0; G0: A rows 128-255 of u+1 -> stage 1 */   \
        TRX_READ(1, 0);                                                                                    \
        if (wave_m) { TRX_DMA_A(0); } else { TRX_DMA_A(1); }                                               \
        TRX_WAIT_L(4);                                                                                     \
        TRX_MFMA_ACC();                                                                                    \
        TRX_END_M();                                                                                       \
        /* ================= K-step u+1 = odd (stage 1) ================= */                                \
        TRX_READ(0, 1);                                                                                    \
        TRX_DMA_B(0);                                                                                      \
        TRX_WAIT_L(4);                                                                                     \
        TRX_MFMA_ACC();                                                                                    \
        TRX_END_M();                                                                                       \
        /* ---- L(u+1, 1) ----  G1: A rows 0-127 of u+3 -> stage 1; G0: A rows 128-255 of u+2 -> stage 0 */ \
        TRX_READ(1, 1);                                                                                    \
        if (wave_m) { TRX_DMA_A(1); } else { TRX_DMA_A(0); }                                               \
        TRX_WAIT_L(4);                                                                                     \
        if (!(LAST)) {                                                                                     \
            TRX_MFMA_ACC();                                                                                \
        } else {                                                                                           \
            /* last 32 MFMAs of the tile.  In their gaps, for every finished group of 4 accumulator rows (mt, nt): \
               its maximum g (2 VALU) and the lane's tile maximum m (1); if g reaches the query's threshold in ANY   \
               lane (v_cmp + one scalar branch, rarely taken: ~3 of the 32 groups of a tile once the thresholds are   \
               warm) the group's rows that pass are listed right here, where the group is a known register: masked   \
               stores into the lane's own list, counter in a byte of cnt4.  Room for a whole tile (32 rows per list)  \
               is guaranteed by the check at the end of the previous tile's bookkeeping. */                           \
            __builtin_amdgcn_s_setprio(1);                                                                 \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) m[nt] = NEG_INF;                              \
            const u32 lane_l = lane_now();                                                                 \
            const int tile_row0_l = (tile0 + tl) * TILE_M;                                                 \
            const u32 id_l = (u32)(tile_row0_l + wave_m * 128) + (lane_l >> 4) * 4u;                       \
            u64* const lp_l = p.cand + ((((qbase + wave_n * 64 + (lane_l & 15)) * p.nsplits + split) * 2 + wave_m) * 4 + (lane_l >> 4)) * p.cap_alloc; \
            const int64_t colstride_l = (int64_t)16 * p.nsplits * LISTS_PER_SPLIT * p.cap_alloc;          \
            /* slot (mt, nt) = the gap behind MFMA (mt, nt).  Group (mt - 1, nt) -- 4 rows, one query column -- is   \
               reduced in slot (mt, nt), one row of MFMAs after it is complete (no wait for the MFMA result), and    \
               tested in slot (mt + 1, nt), one row later again (no v_cmp -> s_cmp -> s_cbranch chain inside a       \
               16-cycle gap; an untaken test costs nothing measurable). */                                           \
            u64 hA_[4] = {0ull, 0ull, 0ull, 0ull}, hB_[4] = {0ull, 0ull, 0ull, 0ull};                      \
            _Pragma("unroll") for (int mt = 0; mt < 10; ++mt) {                                            \
                _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                         \
                    if (mt < 8) acc[mt][nt] = TRX_MFMA1(fa[mt], fb[nt], acc[mt][nt]);                      \
                    hB_[nt] = hA_[nt];                                                                     \
                    if (mt >= 1 && mt < 9) {                /* reduce group mt - 1 */                        \
                        const float g = max4f(acc[mt - 1][nt]);                                            \
                        m[nt] = __builtin_fmaxf(m[nt], g);                                                 \
                        hA_[nt] = mask_ge(g, thrk[nt]);                                                    \
                    }                                                                                      \
                    if (mt >= 2 && !BOOT && TRX_SITE_COND(hB_[nt])) {       /* list rows of group mt - 2 */   \
                        const float tk_ = thrk[nt];                                                        \
                        u64* const lq_ = lp_l + nt * colstride_l + ((cnt4 >> (8 * nt)) & 0x7fu);           \
                        u32 ns_ = 0u;                                                                      \
                        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                    \
                            const float a_ = (float)acc[mt - 2][nt][r];                                    \
                            const u64 pk_ = mask_ge(a_, tk_);                                              \
                            if (pk_) {                                                                     \
                                store_masked(pk_, lq_ + ns_, make_comp(KS * a_ + 0.0f, id_l + (mt - 2) * 16 + r)); \
                                asm volatile("v_addc_co_u32 %0, vcc, 0, %0, %1" : "+v"(ns_) : "s"(pk_) : "vcc"); \
                            }                                                                              \
                        }                                                                                  \
                        cnt4 += ns_ << (8 * nt);                                                           \
                    }                                                                                      \
                    __builtin_amdgcn_sched_barrier(0);                                                     \
                }                                                                                          \
            }                                                                                              \
            __builtin_amdgcn_s_setprio(0);                                                                 \
        }                                                                                                  \
        TRX_END_M();                                                                                       \
    }

    // The bookkeeping of tile tl-1 lives in the first load phase of tile tl; after the last tile that phase runs
    // once more on its own (its reads and pieces are never used: the corpus has a spare tile behind its last row).
#ifdef TRX_STAMP_BUILD
    unsigned long long st_mid = 0ull;      // wall clock (100 MHz) at the head of the split's middle tile: how far apart the sharers of a stream run
#endif
    for (int tl = 0;; ++tl) {
#ifdef TRX_STAMP_BUILD
        if (tl == (ntl >> 1)) st_mid = __builtin_amdgcn_s_memrealtime();
#endif
        TRX_PAIR_HEAD();
        if (tl == ntl) break;
        TRX_PAIR_REST(true, false);
        for (int ks = 2; ks + 2 < ksteps; ks += 2) { TRX_PAIR_HEAD_PLAIN(); TRX_PAIR_REST(false, false); }
        TRX_PAIR_HEAD_PLAIN();
        TRX_PAIR_REST(false, true);
    }
    if (!wave_m) __builtin_amdgcn_s_barrier();      // group 0 waits for the interval group 1 is behind
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (BOOT) {
        // publish min(own, partner) -- the partner's value must be its final one here
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (J == 4 && p.kprime == 16) {
            // The bootstrap of a kprime = 16 search (round 5).  The main scan's rule -- the minimum over a query's 8 lanes of each
            // lane's second-best tile maximum -- is a key that 16 rows reach, but it sits near the 32nd best of the rows seen (the
            // minimum of eight second-order statistics), and the first 16 tiles of every split run on the bootstrap's bound: half
            // of all rows a 125,000-row shard lists are listed there.  This launch has time: it tracks FOUR maxima
            // per lane (J = 4) -- 8 lanes x 4 = 32 different rows of the query -- and publishes their 16th largest, which 16 rows
            // reach by construction and which is the 16th best of the 4,096 rows seen unless one lane holds more than four of the
            // top 16.  One thread per query selects it from an LDS image of the 32 values (the staging area is dead by now).
            float* sc = reinterpret_cast<float*>(smem);      // [256 queries][8 cells][4]
            {
                const u32 a_trk = lds0 + S_TRK + (u32)tid * 4;
                u32 w[8];
                w[0] = lds_ld32<0>(a_trk); w[1] = lds_ld32<2048>(a_trk); w[2] = lds_ld32<4096>(a_trk); w[3] = lds_ld32<6144>(a_trk);
                w[4] = lds_ld32<8192>(a_trk); w[5] = lds_ld32<10240>(a_trk); w[6] = lds_ld32<12288>(a_trk); w[7] = lds_ld32<14336>(a_trk);
                __syncthreads();                             // everybody has read its tracked maxima: S_TRK lies inside nobody's image, but the
                                                             // image overwrites the DMA stages other waves' last (unused) pieces were aimed at
                const int cell = wave_m * 4 + fq;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const u32 bits = (j & 1) ? (w[(j >> 1) * 4 + nt] & 0xffff0000u) : (w[(j >> 1) * 4 + nt] << 16);
                        sc[(ql0 + 16 * nt) * 32 + cell * 4 + j] = __uint_as_float(bits);
                    }
            }
            __syncthreads();
            if (tid < TILE_N) {
                float vals[32];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(sc + tid * 32 + 4 * i);
                    vals[4 * i] = t4[0]; vals[4 * i + 1] = t4[1]; vals[4 * i + 2] = t4[2]; vals[4 * i + 3] = t4[3];
                }
                float kth = NEG_INF;      // the value with exactly 15 others in front of it (ties broken by position)
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    int before = 0;
#pragma unroll
                    for (int j = 0; j < 32; ++j) before += (vals[j] > vals[i] || (vals[j] == vals[i] && j < i)) ? 1 : 0;
                    kth = before == 15 ? vals[i] : kth;
                }
                if (kth > NEG_INF) {
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl)
                        __hip_atomic_fetch_max(p.g_thr + (qbase * 4 + sl * 256 + tid), ordkey(KS * kth), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            return;
        }
        if (wave_m == 0 && fq == 0) {
            auto pubb = [&](auto NT) {
                constexpr int nt = decltype(NT)::value;
                const float a = __uint_as_float(lds_ld32<R_THRW + 64 * nt>(b4_m));
                const float b = __uint_as_float(lds_ld32<R_THRW + 64 * nt>(b4_p));
                const float both = __builtin_fminf(a, b);
                if (both > NEG_INF) {     // kprime rows reach it: it stands alone, so it goes into all four slots
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl)
                        __hip_atomic_fetch_max(p.g_thr + (qbase * 4 + sl * 256 + ql0 + 16 * nt), ordkey(KS * both), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            };
            pubb(ic<0>{}); pubb(ic<1>{}); pubb(ic<2>{}); pubb(ic<3>{});
        }
        return;
    }
#ifdef TRX_STAMP_BUILD
    if (p.stamp_out && lane == 0) {
        unsigned long long* o = p.stamp_out + ((size_t)blockIdx.x * 8 + wave) * 12;
        o[0] = st_cyc; o[1] = st_comp; o[2] = 0ull; o[3] = (unsigned long long)ntl;
        o[4] = a_wt; o[5] = a_bl; o[6] = a_mf; o[7] = a_bm; o[8] = a_ld; o[9] = a_n; o[10] = st_mid; o[11] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    // ---- publish count and bound of this lane's 4 lists ----
    {
        const int lane_ = (int)lane_now();
        const int fq_ = lane_ >> 4, ql_ = wave_n * 64 + (lane_ & 15);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int64_t li = (((qbase + ql_ + 16 * nt) * p.nsplits + split) * 2 + wave_m) * 4 + fq_;
            const u32 c4 = (cnt4 >> (8 * nt)) & 0xffu;
            u64 bound = thrk[nt] > NEG_INF ? ((u64)ordkey(KS * thrk[nt]) << 32) : 0ull;   // rows never listed have key < KS thrk
            if (c4 & 0x80u) { const u64 fl = ld_u64_l2(p.cand_thr + li); bound = fl > bound ? fl : bound; }
            p.cand_cnt[li] = c4 & 0x7fu;
            p.cand_thr[li] = bound;
        }
    }
}

template <bool L2, int J, bool BOOT, int NKS, bool RESCAN, int FMT = 0>
static hipError_t launch_one(const ScanParams& p, hipStream_t st) {
    // the attribute is per device (the ABI takes a device ordinal): one bit per ordinal, per instantiation
    static std::atomic<unsigned long long> attr_devs{0ull};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_scan_kernel<L2, J, BOOT, NKS, RESCAN, FMT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
        if (e != hipSuccess) return e;
        attr_devs.fetch_or(bit, std::memory_order_release);
    }
    dim3 grid(BOOT ? p.nqtiles : p.nqtiles * p.nsplits), block(SCAN_THREADS);
    hipLaunchKernelGGL((knn_scan_kernel<L2, J, BOOT, NKS, RESCAN, FMT>), grid, block, LDS_TOTAL, st, p);
    return hipGetLastError();
}

template <bool L2, int J, bool BOOT>
static hipError_t launch_ks(const ScanParams& p, hipStream_t st) {
    // (32 K-steps = 2048 components compiled in: no gain, 91.4 ms either way on the fingerprint workload)
    if (p.i8 == 1) return launch_one<L2, J, BOOT, 0, false, 1>(p, st);      // the int8 form: the bootstrap and the main scan (the re-scan of uncertified queries stays bf16)
    if (p.i8 == 2) return launch_one<L2, J, BOOT, 0, false, 2>(p, st);      // the fp4 form, likewise
    if (!BOOT && p.fixed_thr) return p.Kp == 12 * BK ? launch_one<L2, J, false, 12, true>(p, st) : launch_one<L2, J, false, 0, true>(p, st);
    return p.Kp == 12 * BK ? launch_one<L2, J, BOOT, 12, false>(p, st) : launch_one<L2, J, BOOT, 0, false>(p, st);
}

hipError_t launch_scan(const ScanParams& p, int metric, hipStream_t st) {
    const bool l2 = metric == 1, j4 = p.kprime > 16;
    if (p.bootstrap) {
        // the bootstrap always tracks four maxima per lane: for kprime = 16 it publishes the 16th largest of a query's 32 (see the
        // kernel's BOOT epilogue); TRX_BOOT_J2=1 keeps the main scan's rule (the A/B switch of round 5)
        const bool boot_j2 = getenv("TRX_BOOT_J2") != nullptr;      // (read per launch: tests switch it inside one process)
        if (boot_j2 && !j4) return l2 ? launch_ks<true, 2, true>(p, st) : launch_ks<false, 2, true>(p, st);
        return l2 ? launch_ks<true, 4, true>(p, st) : launch_ks<false, 4, true>(p, st);
    }
    if (l2) return j4 ? launch_ks<true, 4, false>(p, st) : launch_ks<true, 2, false>(p, st);
    return j4 ? launch_ks<false, 4, false>(p, st) : launch_ks<false, 2, false>(p, st);
}

}  // namespace trx
