// Issue rates of the vector instructions knn_select's re-scoring loop and the attention softmax are made of (gfx950), in cycles per wave instruction
// with 1 and with 4 waves per SIMD:  fp64_rate_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void rate(double* out, unsigned long long* cyc, int iters, float seed) {
    float f[8]; double d[8], acc = 0.0;
    for (int i = 0; i < 8; ++i) { f[i] = seed + threadIdx.x + i; d[i] = f[i]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) { asm volatile("v_cvt_f64_f32_e32 %0, %1" : "=v"(d[i]) : "v"(f[i])); }                       // independent converts
            else if (MODE == 1) { asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "v"(d[i]), "v"(d[(i + 1) & 7])); }   // dependent chain
            else if (MODE == 2) { asm volatile("v_fmac_f64_e32 %0, %1, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7])); }             // 8 independent chains
            else if (MODE == 3) { asm volatile("v_lshlrev_b32_e32 %0, 16, %0" : "+v"(f[i])); }                               // a full-rate reference
            else if (MODE == 5) { asm volatile("v_exp_f32_e32 %0, %0" : "+v"(f[i])); }
            else if (MODE == 6) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7])); }
            else if (MODE == 7) { asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7])); }
            else if (MODE == 8) { asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7])); }
            else if (MODE == 9) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) & 7]), "v"(f[(i + 2) & 7])); }
            else if (MODE == 4) { asm volatile("v_cvt_f64_f32_e32 %0, %1\n\tv_fmac_f64_e32 %2, %0, %0" : "=&v"(d[i]), "+v"(f[i]), "+v"(acc)); }  // the loop's pair
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = acc;
    for (int i = 0; i < 8; ++i) s += d[i] + f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    double* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 8 * 4096 * 256)); CK(hipMalloc(&cyc, 8 * 4096));
    const int iters = 20000;
    const char* names[10] = {"v_cvt_f64_f32 (independent)", "v_fmac_f64 (one dependent chain)", "v_fmac_f64 (8 chains)", "v_lshlrev_b32", "cvt + dependent fmac", "v_exp_f32", "v_pk_mul_f32", "v_cvt_pk_bf16_f32", "v_max3_f32", "v_fma_f32"};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int m = 0; m < 10; ++m) {
        const int grid = 256 * 16;          // 16 workgroups of 4 waves per CU: every SIMD runs 16 waves, 4 at a time or more
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (m == 0) hipLaunchKernelGGL(rate<0>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0f);
            if (m == 1) hipLaunchKernelGGL(rate<1>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0f);
            if (m == 2) hipLaunchKernelGGL(rate<2>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0f);
            if (m == 3) hipLaunchKernelGGL(rate<3>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0f);
            if (m == 4) hipLaunchKernelGGL(rate<4>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0f);
            if (m == 5) hipLaunchKernelGGL(rate<5>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0f);
            if (m == 6) hipLaunchKernelGGL(rate<6>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0f);
            if (m == 7) hipLaunchKernelGGL(rate<7>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0f);
            if (m == 8) hipLaunchKernelGGL(rate<8>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0f);
            if (m == 9) hipLaunchKernelGGL(rate<9>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double instr_per_simd = 16.0 * iters * 8.0 * (m == 4 ? 2 : 1);
            if (rep) printf("%-34s %7.3f ms  -> %.2f ns per wave instruction per SIMD (x 2.4 GHz = %.1f cycles)\n", names[m], ms,
                            ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
        }
    }
    return 0;
}
