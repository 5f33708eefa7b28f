// Do scalar atomics and glc scalar loads work as a cross-workgroup progress counter on gfx950?  (round 5: the sharer sync of the
// scan kernel would cost no vector register and no vmcnt slot if they do.)
//   hipcc --offload-arch=gfx950 -O2 -o scalar_atomic_probe scalar_atomic_probe.hip && ./scalar_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned* counter, unsigned* out, int rounds) {
    unsigned seen = 0, spins = 0;
    for (int r = 1; r <= rounds; ++r) {
        unsigned one = 1;
        asm volatile("s_atomic_add %0, %1, 0x0" ::"s"(one), "s"(counter) : "memory");          // no return
        const unsigned need = (unsigned)r * gridDim.x;
        do {
            asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(seen) : "s"(counter) : "memory");
            ++spins;
            if (seen < need) __builtin_amdgcn_s_sleep(2);
        } while (seen < need && spins < 2000000u);
    }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = seen; out[2 * blockIdx.x + 1] = spins; }
}

int main() {
    const int blocks = 256, rounds = 200;
    unsigned *c, *o;
    hipMalloc(&c, 4); hipMalloc(&o, blocks * 8);
    hipMemset(c, 0, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, c, o, rounds);
    hipEventRecord(b);
    hipError_t e = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned> h(blocks * 2); unsigned total = 0;
    hipMemcpy(h.data(), o, blocks * 8, hipMemcpyDeviceToHost); hipMemcpy(&total, c, 4, hipMemcpyDeviceToHost);
    unsigned maxspin = 0, minseen = ~0u;
    for (int i = 0; i < blocks; ++i) { maxspin = h[2 * i + 1] > maxspin ? h[2 * i + 1] : maxspin; minseen = h[2 * i] < minseen ? h[2 * i] : minseen; }
    printf("%s: counter %u (want %u), every block saw >= %u, most polls by one block %u, %d barriers of %d workgroups in %.3f ms = %.2f us each\n",
           hipGetErrorString(e), total, blocks * rounds, minseen, maxspin, rounds, blocks, ms, ms * 1e3 / rounds);
    return 0;
}
