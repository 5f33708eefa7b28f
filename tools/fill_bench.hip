// tools/fill_bench.hip -- micro-benchmark of the global -> LDS fill path on gfx950, in the access
// pattern of knn_scan_kernel (256 workgroups of 512 threads, each step brings 32 KiB of a shared
// "corpus" stream and 32 KiB of a private "query" tile, 1 KiB per wave-instruction).
//   mode 0: LDS-DMA (global_load_lds_dwordx4), counted vmcnt, 2 steps in flight
//   mode 1: register staging (global_load_dwordx4 -> ds_write_b128), 1 step in flight
//   mode 2: register loads only (no LDS write)
// build: hipcc --offload-arch=gfx950 -O3 tools/fill_bench.hip -o gpurun_out/fill_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

template <int MODE>
__global__ __launch_bounds__(512, 2) void fill_kernel(const char* corpus, const char* queries, int ntiles, int ksteps,
                                                      int64_t row_bytes, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int prow = lane >> 3, pslot = lane & 7;
    unsigned off[4];
    for (int i = 0; i < 4; ++i) off[i] = (unsigned)(((wave * 4 + i) * 8 + prow) * row_bytes + pslot * 16);
    const char* qb = queries + (int64_t)blockIdx.x * 256 * row_bytes;
    unsigned acc = 0;
    uint4 r[8];
    const int total = ntiles * ksteps;
    for (int s = 0; s < total; ++s) {
        const int tl = s / ksteps, ks = s - tl * ksteps;
        const char* a = corpus + (int64_t)tl * 256 * row_bytes + ks * 128;
        const char* b = qb + ks * 128;
        char* la = smem + (s & 1) * 65536 + wave * 4096;
        char* lb = la + 32768;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_global_load_lds((gbl_void*)(a + off[i]), (lds_void*)(la + i * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_void*)(b + off[i]), (lds_void*)(lb + i * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // previous step landed, this one in flight
            __builtin_amdgcn_s_barrier();
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r[i] = *reinterpret_cast<const uint4*>(a + off[i]);
                r[4 + i] = *reinterpret_cast<const uint4*>(b + off[i]);
            }
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    *reinterpret_cast<uint4*>(la + i * 1024 + lane * 16) = r[i];
                    *reinterpret_cast<uint4*>(lb + i * 1024 + lane * 16) = r[4 + i];
                }
                __syncthreads();
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc ^= r[i].x ^ r[i].w;
            }
        }
    }
    __syncthreads();
    acc ^= reinterpret_cast<unsigned*>(smem)[tid];
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int ntiles = argc > 2 ? atoi(argv[2]) : 1000;
    const int ksteps = 12;
    const int64_t row_bytes = 768 * 2;
    char *corpus, *queries; unsigned* sink;
    CK(hipMalloc(&corpus, (size_t)(ntiles + 1) * 256 * row_bytes));
    CK(hipMalloc(&queries, (size_t)256 * 256 * row_bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(corpus, 1, (size_t)(ntiles + 1) * 256 * row_bytes));
    CK(hipMemset(queries, 2, (size_t)256 * 256 * row_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = 131072 + 2048;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) { (void)hipFuncSetAttribute((const void*)fill_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(fill_kernel<0>, dim3(256), dim3(512), lds, 0, corpus, queries, ntiles, ksteps, row_bytes, sink); }
        if (mode == 1) { (void)hipFuncSetAttribute((const void*)fill_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(fill_kernel<1>, dim3(256), dim3(512), lds, 0, corpus, queries, ntiles, ksteps, row_bytes, sink); }
        if (mode == 2) { (void)hipFuncSetAttribute((const void*)fill_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(fill_kernel<2>, dim3(256), dim3(512), lds, 0, corpus, queries, ntiles, ksteps, row_bytes, sink); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = 256.0 * ntiles * ksteps * 65536.0;
        printf("mode %d: %.2f ms, %.1f GB filled, %.2f TB/s aggregate, %.1f GB/s per CU (%s)\n", mode, ms, bytes / 1e9,
               bytes / ms / 1e9, bytes / ms / 1e6 / 256, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
