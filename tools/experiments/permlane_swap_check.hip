// what v_permlane32_swap returns (gfx950) for operands x = lane and y = lane + 100: the maximum of the pair, then both results
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
    float x = (float)threadIdx.x;
    float a = x, b = x + 100.f;
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));      // (the builtin of this hipcc returns the first register twice)
    out[threadIdx.x] = fmaxf(a, b);
    out[64 + threadIdx.x] = a; out[128 + threadIdx.x] = b;
}
int main() {
    float* d; hipMalloc(&d, 192 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); float h[192]; hipMemcpy(h, d, 192 * 4, hipMemcpyDeviceToHost);
    for (int j = 0; j < 3; ++j) { for (int i = 0; i < 64; ++i) printf("%g ", h[j * 64 + i]); printf("\n"); }
    return 0;
}
