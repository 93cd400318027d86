// Microbenchmark (development): what an LDS-direct load costs a wave that is busy multiplying.  One block = W waves per CU; every wave runs
// `iters` chunks of 36 v_mfma_f32_16x16x4_f32 (nine groups of four, independent accumulators) with N 1-KB LDS-direct loads spread one per
// group, a vmcnt(0) at the end of the chunk and (BAR) a block barrier -- the skeleton of conv3_wino4_kernel's K loop without its fragment reads.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_ldsdma2 ubench_ldsdma2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N, int BAR, int FR, int MID = -1, int PL = 0>
__global__ __launch_bounds__(512) void k(const float* src, uint32_t src_bytes, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int l = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t a = (uint64_t)(uintptr_t)src;
    const u32x4 d = {(uint32_t)a, (uint32_t)((a >> 32) & 0xffffu), src_bytes, 0x00020000u};
    const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(uintptr_t)smem + wv * 9216 + (blockDim.x >> 6) * 9216));
    const uint32_t voff = l * 16;
    f32x4 acc[36];
#pragma unroll
    for (int i = 0; i < 36; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float av = (float)l, bv = 1.0f;
    for (int i = 0; i < iters; i++) {
        const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(((blockIdx.x * 8 + wv) * 64 + (i & 63)) * 9216u) % (src_bytes - 16384u)));
        float4 fab[3], fbb[3];
        auto frag = [&](int g, int q) {
            fab[q] = *(const float4*)(smem + wv * 2304 + ((g * 64 + l) & 511) * 4);
            fbb[q] = *(const float4*)(smem + wv * 2304 + 1024 + ((g * 64 + l) & 255) * 4);
        };
        if (FR == 2) { frag(0, 0); frag(1, 1); }
#pragma unroll
        for (int g = 0; g < 9; g++) {
            float4 fa = make_float4(av, av, av, av), fb = make_float4(bv, bv, bv, bv);
            if (FR == 1) { fa = *(const float4*)(smem + wv * 2304 + ((g * 64 + l) & 511) * 4); fb = *(const float4*)(smem + wv * 2304 + 1024 + ((g * 64 + l) & 255) * 4); }
            if (FR == 2) { if (g + 2 < 9) frag(g + 2, (g + 2) % 3); __builtin_amdgcn_sched_barrier(0); fa = fab[g % 3]; fb = fbb[g % 3]; }
            __builtin_amdgcn_sched_barrier(0);
            acc[4 * g + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.x, fb.x, acc[4 * g + 0], 0, 0, 0);
            acc[4 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.y, fb.y, acc[4 * g + 1], 0, 0, 0);
            {
                constexpr int per = (PL == 0 || PL == 6) ? 1 : (PL == 2 ? 9 : 2);
                constexpr int g0 = PL == 3 ? 1 : (PL == 4 ? 2 : (PL == 5 ? 3 : 0));
#pragma unroll
                for (int q = 0; q < per; q++) {
                    const int pi = (g - g0) * per + q;
                    if (g >= g0 && pi < N && (PL != 2 || g == 0)) {
                        const uint32_t dst = lds_base + pi * 1024, sop = so + pi * 1024u;
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(dst), "v"(voff), "s"(d), "s"(sop) : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (MID >= 0 && g == 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(MID >= 0 ? MID : 0) : "memory");
            if (PL == 6 && g == 4) { asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); if (BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
            acc[4 * g + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.z, fb.z, acc[4 * g + 2], 0, 0, 0);
            acc[4 * g + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.w, fb.w, acc[4 * g + 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (PL == 6) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 36; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) sink[0] = s;
}

template <int N, int BAR, int FR, int MID = -1, int PL = 0>
void run(const char* name, int waves, const float* src, uint32_t bytes, float* sink, int bpc = 1) {
    const int iters = 1000, blocks = 256 * bpc;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void*)k<N, BAR, FR, MID, PL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k<N, BAR, FR, MID, PL><<<blocks, waves * 64, (150 / bpc) * 1024>>>(src, bytes, 10, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<N, BAR, FR, MID, PL><<<blocks, waves * 64, (150 / bpc) * 1024>>>(src, bytes, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double cyc = ms * 1e-3 * 2.1e9 / iters;           // cycles per chunk (at 2.1 GHz)
    printf("%-40s %d x %d waves/CU  %8.0f cycles per chunk  (MFMA alone: %d)  %5.1f B/clk/CU of LDS-direct loads\n", name, bpc, waves, cyc, 36 * 32 * waves * bpc / 4, N * 1024.0 * waves * bpc / cyc);
}

int main() {
    const uint32_t bytes = 8u << 20;
    float *src, *sink;
    hipMalloc(&src, bytes); hipMemset(src, 0, bytes); hipMalloc(&sink, 64);
    run<0, 1, 2>("no loads, prefetched frags, barrier", 4, src, bytes, sink, 2);
    run<9, 1, 2, -1, 0>("1/group, wait at chunk end", 4, src, bytes, sink, 2);
    run<9, 1, 2, -1, 1>("2/group g0-4, wait at chunk end", 4, src, bytes, sink, 2);
    run<9, 1, 2, -1, 6>("1/group, SKEWED half-chunk waits", 4, src, bytes, sink, 2);
    run<9, 1, 2, -1, 6>("1/group, SKEWED, 8 waves 1 block", 8, src, bytes, sink, 1);
    run<9, 1, 2, -1, 0>("1/group, chunk end, 8 waves 1 block", 8, src, bytes, sink, 1);
    const uint32_t mid = 192u << 20;                // beyond the L2s, inside the Infinity Cache
    float* srcm; hipMalloc(&srcm, mid); hipMemset(srcm, 0, mid);
    run<9, 1, 2, -1, 0>("1/group, chunk end, 192 MiB source", 4, srcm, mid, sink, 2);
    run<9, 1, 2, -1, 6>("1/group, SKEWED, 192 MiB source", 4, srcm, mid, sink, 2);
    return 0;
}
