// what v_permlane32_swap returns (gfx950) for operands x = lane and y = lane + 100: the maximum of the pair, then both results
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
    float x = (float)threadIdx.x;
    // v_permlane32_swap vdst, src: lanes 32-63 of vdst swap with lanes 0-31 of src.  Through asm with the two wait states the
    // hazard rule asks for after a VALU write of an operand (this hipcc's builtin returns vdst as BOTH results and pads nothing).
    // a = b = lane  ->  a = [lo | lo], b = [hi | hi]: max(a, b) is the cross-half maximum in every lane
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    out[threadIdx.x] = fmaxf(a, b);
    out[64 + threadIdx.x] = a; out[128 + threadIdx.x] = b;
}
int main() {
    float* d; hipMalloc(&d, 192 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); float h[192]; hipMemcpy(h, d, 192 * 4, hipMemcpyDeviceToHost);
    for (int j = 0; j < 3; ++j) { for (int i = 0; i < 64; ++i) printf("%g ", h[j * 64 + i]); printf("\n"); }
    return 0;
}
