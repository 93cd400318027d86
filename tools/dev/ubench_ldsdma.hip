// Microbenchmark (development): throughput of LDS-direct buffer loads (buffer_load_dword{,x4} ... offen lds) per CU, source L2-resident,
// W waves per block, one block per CU -- alone, and next to ds_read_b128 traffic from the same waves.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_ldsdma ubench_ldsdma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int WIDTH, int READS>
__global__ __launch_bounds__(1024) void k(const float* src, uint32_t src_bytes, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t a = (uint64_t)(uintptr_t)src;
    const u32x4 d = {(uint32_t)a, (uint32_t)((a >> 32) & 0xffffu), src_bytes, 0x00020000u};
    const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(uintptr_t)smem + wv * 8192));         // 8 KB of LDS per wave
    const uint32_t voff = l * (WIDTH * 4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < iters; i++) {
        const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(((blockIdx.x * 16 + wv) * 64 + (i & 63)) * 1024u) % (src_bytes - 8192u)));
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const uint32_t dst = lds_base + p * 1024, sop4 = so + p * 1024u, sop1 = so + p * 256u;
            if (WIDTH == 4) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(dst), "v"(voff), "s"(d), "s"(sop4) : "memory");
            else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" :: "s"(dst), "v"(voff), "s"(d), "s"(sop1) : "memory");
            if (READS) {
#pragma unroll
                for (int r = 0; r < READS; r++) { const float4 v = *(const float4*)(smem + wv * 2048 + ((r * 64 + l) & 511) * 4); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (acc.x == 123.456f) sink[0] = acc.x + acc.y + acc.z + acc.w;
}

template <int WIDTH, int READS>
void run(const char* name, int waves, const float* src, uint32_t bytes, float* sink) {
    const int iters = 2000, blocks = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void*)k<WIDTH, READS>, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 8192);
    k<WIDTH, READS><<<blocks, waves * 64, waves * 8192>>>(src, bytes, 10, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<WIDTH, READS><<<blocks, waves * 64, waves * 8192>>>(src, bytes, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes_moved = (double)blocks * waves * iters * 8 * 64 * WIDTH * 4;
    printf("%-44s %2d waves/CU  %7.3f ms  %6.1f B/clk/CU (at 2.1 GHz)  %6.2f TB/s chip\n", name, waves, ms, bytes_moved / blocks / (ms * 1e-3 * 2.1e9), bytes_moved / ms / 1e9);
}

int main() {
    const uint32_t bytes = 8u << 20;              // 8 MiB source: L2 / Infinity-Cache resident
    float *src, *sink;
    hipMalloc(&src, bytes); hipMemset(src, 0, bytes); hipMalloc(&sink, 64);
    for (int w : {4, 8, 16}) run<4, 0>("dwordx4 lds, alone", w, src, bytes, sink);
    for (int w : {4, 8, 16}) run<1, 0>("dword lds, alone", w, src, bytes, sink);
    for (int w : {8}) run<4, 2>("dwordx4 lds + 2 ds_read_b128 per piece", w, src, bytes, sink);
    for (int w : {8}) run<4, 8>("dwordx4 lds + 8 ds_read_b128 per piece", w, src, bytes, sink);
    return 0;
}
