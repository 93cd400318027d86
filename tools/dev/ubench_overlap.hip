// Does fp32 MFMA overlap with (packed) fp32 VALU on gfx950?  A: MFMA only, B: VALU only, C: both interleaved in one wave,
// D: half the waves run A, the other half B (same SIMDs).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    f32x2 v0 = {a, b}, v1 = {b, a}, v2 = {a, a}, v3 = {b, b}, v4 = {a, 1.f}, v5 = {b, 2.f}, v6 = {a, 3.f}, v7 = {b, 4.f};
    const f32x2 m = {a, a}, c = {b, b};
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && (blockIdx.x & 1) == 0);
    const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && (blockIdx.x & 1) == 1);
    for (int i = 0; i < iters; i++) {
        if (do_mfma) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int r = 0; r < 8; r++) {      // 64 pk_fma = 256 cycles = 4 MFMA 32x32x2 (4 x 64 cycles)
                v0 = __builtin_elementwise_fma(v0, m, c); v1 = __builtin_elementwise_fma(v1, m, c); v2 = __builtin_elementwise_fma(v2, m, c); v3 = __builtin_elementwise_fma(v3, m, c);
                v4 = __builtin_elementwise_fma(v4, m, c); v5 = __builtin_elementwise_fma(v5, m, c); v6 = __builtin_elementwise_fma(v6, m, c); v7 = __builtin_elementwise_fma(v7, m, c);
            }
        }
    }
    float s = 0;
    for (int r = 0; r < 16; r++) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    s += v0.x + v1.x + v2.x + v3.x + v4.y + v5.y + v6.y + v7.y;
    if (s == 12345.f) out[threadIdx.x] = s + wave;
}
template <int MODE> float run(int blocks, int iters) {
    float* d; hipMalloc(&d, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, iters, 0.5f, 0.25f); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters, 0.5f, 0.25f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(d); return ms;
}
int main() {
    const int iters = 20000;
    for (int bpc = 1; bpc <= 2; bpc++) {
        const int blocks = 256 * bpc;
        printf("blocks/CU %d: mfma-only %.3f ms, valu-only %.3f ms, interleaved-in-wave %.3f ms, split-across-blocks %.3f ms\n", bpc, run<0>(blocks, iters), run<1>(blocks, iters),
               run<2>(blocks, iters), run<3>(blocks * 2, iters));
    }
    // rates: MFMA flops = blocks*4 waves*iters*4*(32*32*2*2)
    return 0;
}
