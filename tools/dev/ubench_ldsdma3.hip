// Microbenchmark (development): does a second chunk of look-ahead for the V pieces pay?  8 waves per CU (one block), per wave and chunk 36 MFMAs,
// prefetched fragment reads, 2 U pieces from an L2-resident source and 5 V pieces from a source of `vmib` MiB, one block barrier per chunk.
//   LOOK = 1: everything issued for the next chunk, vmcnt(0) at the chunk's end (two stages).
//   LOOK = 2: U for the next chunk first, then V for the chunk after; vmcnt(5) at the end lets those five V pieces fly (three V stages).
//   hipcc --offload-arch=gfx950 -O3 -o ubench_ldsdma3 ubench_ldsdma3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma(const u32x4& d, uint32_t dst, uint32_t voff, uint32_t so) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(dst), "v"(voff), "s"(d), "s"(so) : "memory");
}

template <int LOOK, int NV>
__global__ __launch_bounds__(512) void k(const float* su, uint32_t su_bytes, const float* sv, uint32_t sv_bytes, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int l = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto desc = [](const float* p, uint32_t bytes) { const uint64_t a = (uint64_t)(uintptr_t)p; return (u32x4){(uint32_t)a, (uint32_t)((a >> 32) & 0xffffu), bytes, 0x00020000u}; };
    const u32x4 du = desc(su, su_bytes), dv = desc(sv, sv_bytes);
    const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(uintptr_t)smem + 8 * 9216 + wv * 16384));
    const uint32_t voff = l * 16;
    f32x4 acc[36];
#pragma unroll
    for (int i = 0; i < 36; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint32_t vpos = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((blockIdx.x * 8u + wv) * 2654435761u) % (sv_bytes - 65536u))) & ~1023u;
    float4 fab[3], fbb[3];
    auto frag = [&](int g, int q) {
        fab[q] = *(const float4*)(smem + wv * 2304 + ((g * 64 + l) & 511) * 4);
        fbb[q] = *(const float4*)(smem + wv * 2304 + 1024 + ((g * 64 + l) & 255) * 4);
    };
    frag(0, 0); frag(1, 1);
    for (int i = 0; i < iters; i++) {
        const uint32_t uo = (uint32_t)__builtin_amdgcn_readfirstlane((int)((((blockIdx.x & 7) * 64 + (i & 63)) * 18432u + wv * 2048u) % (su_bytes - 8192u)));
        vpos = (uint32_t)__builtin_amdgcn_readfirstlane((int)((vpos + 8u * NV * 1024u * 37u) % (sv_bytes - 65536u))) & ~1023u;      // a new 5-KB stretch far from the last one
#pragma unroll
        for (int g = 0; g < 9; g++) {
            if (g + 2 < 9) frag(g + 2, (g + 2) % 3); else frag(g + 2 - 9, (g + 2) % 3);
            __builtin_amdgcn_sched_barrier(0);
            const float4 fa = fab[g % 3], fb = fbb[g % 3];
            acc[4 * g + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.x, fb.x, acc[4 * g + 0], 0, 0, 0);
            acc[4 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.y, fb.y, acc[4 * g + 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (g < 2) dma(du, lds_base + g * 1024, voff, uo + g * 1024u);
            else if (g - 2 < NV) dma(dv, lds_base + g * 1024, voff, vpos + (g - 2) * 1024u);
            __builtin_amdgcn_sched_barrier(0);
            acc[4 * g + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.z, fb.z, acc[4 * g + 2], 0, 0, 0);
            acc[4 * g + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.w, fb.w, acc[4 * g + 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (LOOK == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NV) : "memory");
        asm volatile("s_barrier" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 36; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) sink[0] = s;
}

template <int LOOK, int NV>
void run(const char* name, const float* su, uint32_t sub, const float* sv, uint32_t svb, float* sink) {
    const int iters = 1000, blocks = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void*)k<LOOK, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k<LOOK, NV><<<blocks, 512, 150 * 1024>>>(su, sub, sv, svb, 10, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<LOOK, NV><<<blocks, 512, 150 * 1024>>>(su, sub, sv, svb, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-52s %8.0f cycles per chunk (MFMA alone 2304)\n", name, ms * 1e-3 * 2.1e9 / iters);
}

int main(int argc, char** argv) {
    const uint32_t sub = 8u << 20;
    float *su, *sink; hipMalloc(&su, sub); hipMemset(su, 0, sub); hipMalloc(&sink, 64);
    for (uint32_t mib : {8u, 192u, 1024u}) {
        const uint32_t svb = mib << 20;
        float* sv; hipMalloc(&sv, svb); hipMemset(sv, 0, svb);
        char nm[96];
        snprintf(nm, sizeof nm, "V from %4u MiB, 2 stages (wait for everything)", mib); run<1, 5>(nm, su, sub, sv, svb, sink);
        snprintf(nm, sizeof nm, "V from %4u MiB, 3 V stages (five pieces in flight)", mib); run<2, 5>(nm, su, sub, sv, svb, sink);
        hipFree(sv);
    }
    return 0;
}
