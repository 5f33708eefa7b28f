// which counters does an LDS-DMA (global_load_lds_dwordx4) occupy on gfx950?  One wave issues N pieces from cold
// addresses and then waits on (a) lgkmcnt(0), (b) vmcnt(0); s_memtime around each wait.
// build: hipcc --offload-arch=gfx950 -O3 tools/experiments/ldsdma_counters.hip -o tools/experiments/ldsdma_counters
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
template <int N>
__global__ __launch_bounds__(64) void probe(const char* src, unsigned long long* out, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const char* p = src + (size_t)blockIdx.x * (N * 1024 * 64) + lane * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i)
        __builtin_amdgcn_global_load_lds((gbl_void*)(p + (size_t)i * 65536), (lds_void*)(smem + i * 1024), 16, 0, 0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t3 = __builtin_amdgcn_s_memtime();
    // every piece must hold its own source bytes: word w of piece i = (block, i, lane, w) as written by the host
    unsigned bad = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(lane * 16 + i * 1024)) : "memory");
        const unsigned want = ((unsigned)blockIdx.x << 20) | ((unsigned)i << 8) | (unsigned)lane;
        bad += (v != want);
    }
    if (lane == 0) { out[blockIdx.x * 4 + 0] = t1 - t0; out[blockIdx.x * 4 + 1] = t2 - t1; out[blockIdx.x * 4 + 2] = t3 - t2; }
    sink[blockIdx.x * 64 + lane] = bad;
}
int main() {
    const int blocks = 64; const size_t bytes = (size_t)blocks * 32 * 1024 * 64 + (1 << 20);
    char* src; unsigned long long* out; unsigned* sink;
    hipMalloc(&src, bytes);
 hipMalloc(&out, blocks * 32); hipMalloc(&sink, blocks * 256);
    std::vector<unsigned long long> h(blocks * 4);
#define RUN(N) { { std::vector<unsigned> hs(bytes / 4, 0xdeadbeefu); for (int b = 0; b < blocks; ++b) for (int i = 0; i < N; ++i) for (int l = 0; l < 64; ++l) hs[((size_t)b * (N * 1024 * 64) + (size_t)i * 65536 + l * 16) / 4] = ((unsigned)b << 20) | ((unsigned)i << 8) | (unsigned)l; hipMemcpy(src, hs.data(), bytes, hipMemcpyHostToDevice); } hipFuncSetAttribute((const void*)probe<N>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
    hipLaunchKernelGGL(probe<N>, dim3(blocks), dim3(64), 65536, 0, src, out, sink); hipDeviceSynchronize(); \
    hipMemcpy(h.data(), out, blocks * 32, hipMemcpyDeviceToHost); \
    std::vector<unsigned> hb(blocks * 64); hipMemcpy(hb.data(), sink, blocks * 256, hipMemcpyDeviceToHost); unsigned long long nb = 0; for (unsigned x : hb) nb += x; \
    printf("N = %2d pieces: issue %5llu, wait lgkmcnt(0) %5llu, then wait vmcnt(0) %5llu cycles (block 7); wrong words %llu\n", N, h[28], h[29], h[30], nb); }
    RUN(1) RUN(4) RUN(8) RUN(12) RUN(16) RUN(24) RUN(32)
    return 0;
}
