// What does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer descriptor) leave in LDS:
// zeros, or the old contents?  Build: hipcc -O3 --offload-arch=gfx950 -o buffer_lds_oob buffer_lds_oob.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const char* p, unsigned nbytes, float* o) {
    extern __shared__ char smem[];
    float* f = reinterpret_cast<float*>(smem);
    for (int i = threadIdx.x; i < 256; i += 64) f[i] = -7.0f;      // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
    unsigned voff = threadIdx.x * 16, soff = 0;
    unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(r), "s"(soff), "s"(la) : "memory", "m0");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) o[i] = f[i];
}
int main() {
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 1.0f + i;
    float *d, *o;
    hipMalloc(&d, 1024); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    for (unsigned nb : {1024u, 512u, 520u, 0u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, (const char*)d, nb, o);
        std::vector<float> r(256);
        hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        printf("num_records %4u:", nb);
        for (int i : {0, 1, 127, 128, 129, 130, 131, 132, 255}) printf(" [%d]=%g", i, r[i]);
        printf("\n");
    }
    return 0;
}
