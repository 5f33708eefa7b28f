// Timeline of attention_fwd_mfma_kernel (diagnostic build with stamps): per workgroup start / end (100 MHz clock) and
// cycles spent in the prologue and in each key tile (wave 0).   attn_lab [B H Lq Lk causal]
#define TRX_ATT_STAMP 1
#include "../textreact_amd/csrc/nn_ops.hip"
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, H = argc > 2 ? atoi(argv[2]) : 12, Lq = argc > 3 ? atoi(argv[3]) : 512,
              Lk = argc > 4 ? atoi(argv[4]) : 512, causal = argc > 5 ? atoi(argv[5]) : 0;
    const size_t nq = (size_t)B * Lq * H * 64, nk = (size_t)B * Lk * H * 64;
    std::vector<unsigned short> h(std::max(nq, nk));
    unsigned s = 1u;
    auto fill = [&](size_t n) { for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; float f = (((s >> 8) & 0xffff) / 32768.0f - 1.0f) * 1.7f; union { float f_; unsigned u_; } cv; cv.f_ = f; h[i] = (unsigned short)(cv.u_ >> 16); } };
    bf16_t *q, *k, *v, *o; float *mask, *lse; unsigned long long* st;
    CK(hipMalloc(&q, nq * 2)); CK(hipMalloc(&k, nk * 2)); CK(hipMalloc(&v, nk * 2)); CK(hipMalloc(&o, nq * 2));
    CK(hipMalloc(&mask, (size_t)B * Lk * 4)); CK(hipMemset(mask, 0, (size_t)B * Lk * 4)); CK(hipMalloc(&lse, (size_t)B * H * Lq * 4));
    fill(nq); CK(hipMemcpy(q, h.data(), nq * 2, hipMemcpyHostToDevice));
    fill(nk); CK(hipMemcpy(k, h.data(), nk * 2, hipMemcpyHostToDevice));
    fill(nk); CK(hipMemcpy(v, h.data(), nk * 2, hipMemcpyHostToDevice));
    const int nwg = B * H * ((Lq + 127) / 128);
    CK(hipMalloc(&st, (size_t)nwg * 32 * 8)); CK(hipMemset(st, 0, (size_t)nwg * 32 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_att_stamp), &st, sizeof(st)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        if (trx_attention_fwd_lse(q, k, v, mask, TRX_NN_MASK_KEY, causal, B, H, Lq, Lk, 0.125f, TRX_NN_BF16, o, lse, nullptr)) { printf("%s\n", trx_nn_last_error()); return 1; }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("rep %d: %.1f us  (%.0f TFLOP/s)\n", rep, ms * 1e3, 4.0 * B * H * Lq * (double)Lk * 64 * (causal ? 0.5 : 1.0) / ms / 1e9);
    }
    std::vector<unsigned long long> hs((size_t)nwg * 32);
    CK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < nwg; ++b) { t0 = std::min(t0, hs[b * 32]); t1 = std::max(t1, hs[b * 32 + 1]); }
    printf("kernel span by the 100 MHz clock: %.1f us, %d workgroups\n", (t1 - t0) * 0.01, nwg);
    std::vector<double> start, dur, pro, tile, epi;
    const int nkb = (Lk + 63) / 64;
    for (int b = 0; b < nwg; ++b) {
        const unsigned long long* x = &hs[(size_t)b * 32];
        start.push_back((x[0] - t0) * 0.01); dur.push_back((x[1] - x[0]) * 0.01);
        pro.push_back((double)(x[3] - x[2]));
        int last = 3;
        for (int j = 0; j < nkb && j < 27; ++j) if (x[4 + j]) { tile.push_back((double)(x[4 + j] - x[last])); last = 4 + j; }
        epi.push_back((double)(x[31] - x[last]));
    }
    auto pct = [](std::vector<double> a, double p) { std::sort(a.begin(), a.end()); return a[(size_t)(p * (a.size() - 1))]; };
    printf("workgroup start   (us): p10 %.1f p50 %.1f p90 %.1f max %.1f\n", pct(start, .1), pct(start, .5), pct(start, .9), pct(start, 1));
    printf("workgroup length  (us): p10 %.1f p50 %.1f p90 %.1f max %.1f\n", pct(dur, .1), pct(dur, .5), pct(dur, .9), pct(dur, 1));
    printf("prologue      (cycles): p10 %.0f p50 %.0f p90 %.0f\n", pct(pro, .1), pct(pro, .5), pct(pro, .9));
    printf("key tile      (cycles): p10 %.0f p50 %.0f p90 %.0f\n", pct(tile, .1), pct(tile, .5), pct(tile, .9));
    printf("epilogue      (cycles): p10 %.0f p50 %.0f p90 %.0f\n", pct(epi, .1), pct(epi, .5), pct(epi, .9));
    // how many workgroups are alive at each microsecond
    const int T = (int)((t1 - t0) * 0.01) + 1;
    std::vector<int> alive(T, 0);
    for (int b = 0; b < nwg; ++b) for (int t = (int)start[b]; t <= (int)(start[b] + dur[b]) && t < T; ++t) alive[t]++;
    printf("workgroups alive per us:"); for (int t = 0; t < T; t += 2) printf(" %d", alive[t]); printf("\n");
    return 0;
}
