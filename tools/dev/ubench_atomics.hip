// Microbenchmark (development): throughput of 128-byte-line fp32 atomic adds (32 lanes x 4 B, two lines per wave instruction) into a
// region of R MiB, (a) every block anywhere in the region, (b) every XCD inside its own eighth of the region (XCC_ID read from the hardware
// register), with the cache-policy bits none / sc1 / sc0 sc1.   hipcc --offload-arch=gfx950 -O3 -o ubench_atomics ubench_atomics.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

template <int MODE>
__device__ __forceinline__ void atom(float* p, float v) {
    if (MODE == 0) asm volatile("global_atomic_add_f32 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if (MODE == 1) asm volatile("global_atomic_add_f32 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");

    if (MODE == 3) asm volatile("global_atomic_add_f32 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
}

template <int MODE, bool PART>
__global__ __launch_bounds__(256) void k(float* buf, uint32_t lines, int iters, unsigned* xmask) {
    const int l = threadIdx.x & 63, half = l >> 5, l32 = l & 31;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    if (threadIdx.x == 0) atomicOr(xmask, 1u << xcc);
    uint32_t st = (blockIdx.x * 256u + threadIdx.x / 32u * 7919u) * 2654435761u + 12345u;
    st = __builtin_amdgcn_readfirstlane(st) + half * 0x9e3779b9u;
    const uint32_t per = lines / 8;
    for (int i = 0; i < iters; i++) {
        st = st * 1664525u + 1013904223u;
        uint32_t ln = PART ? (xcc & 7u) * per + (st >> 8) % per : (st >> 8) % lines;
        atom<MODE>(buf + (size_t)ln * 32 + l32, 1.0f);
    }
}

template <int MODE, bool PART>
void run(const char* name, float* buf, uint32_t lines, unsigned* xmask) {
    const int iters = 2000, blocks = 256 * 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE, PART><<<blocks, 256>>>(buf, lines, 10, xmask);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE, PART><<<blocks, 256>>>(buf, lines, iters, xmask);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double n = (double)blocks * 8 * iters;     // wave-halves = lines
    printf("%-28s %8.3f ms  %7.2f G lines/s  %7.2f TB/s of 128-B lines\n", name, ms, n / ms / 1e6, n * 128 / ms / 1e9);
}

int main(int argc, char** argv) {
    const int mib = argc > 1 ? atoi(argv[1]) : 384;
    const uint32_t lines = (uint32_t)((size_t)mib * 1024 * 1024 / 128);
    float* buf; unsigned* xmask;
    hipMalloc(&buf, (size_t)lines * 128); hipMemset(buf, 0, (size_t)lines * 128);
    hipMalloc(&xmask, 4); hipMemset(xmask, 0, 4);
    printf("region %d MiB\n", mib);
    run<0, false>("plain, anywhere", buf, lines, xmask);
    run<0, true>("plain, per-XCD eighth", buf, lines, xmask);
    run<1, false>("sc1, anywhere", buf, lines, xmask);
    run<1, true>("sc1, per-XCD eighth", buf, lines, xmask);

    run<3, false>("nt, anywhere", buf, lines, xmask);
    run<3, true>("nt, per-XCD eighth", buf, lines, xmask);
    unsigned m; hipMemcpy(&m, xmask, 4, hipMemcpyDeviceToHost);
    printf("XCC ids seen: 0x%x\n", m);
    // correctness of the plain form across XCDs: every line must hold an integer total
    return 0;
}
