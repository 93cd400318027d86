// Which vector instructions overlap with fp32 MFMA on gfx950, and at what cost?  One 512-thread block per CU = 2 waves per SIMD.
//   M : waves 0-3 run MFMAs, waves 4-7 exit            (MFMA-only, one wave per SIMD)
//   V : waves 4-7 run the VALU kind, waves 0-3 exit    (VALU-only, one wave per SIMD)
//   MV: waves 0-3 MFMA, waves 4-7 VALU                 (two DIFFERENT waves per SIMD: max(M,V) = the pipes overlap, M+V = they do not)
//   I : waves 0-3 run both, interleaved in one wave    (what a kernel's inner loop does)
// (r01's split test picked the role by blockIdx & 1, which on this chip is the parity of the XCD: the two roles never shared a SIMD.)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int VK>
__device__ __forceinline__ void valu64(float (&v)[8], f32x2 (&w)[8], float m, float c, int lds_addr, __amdgpu_buffer_rsrc_t rsrc, f32x4& w4) {
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (VK == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m), "v"(c));
            if (VK == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[j]) : "v"(w[(j + 1) & 7]), "v"(w[(j + 2) & 7]));
            if (VK == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[j]) : "v"(m));
            if (VK == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(m));
            if (VK == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[j]) : "v"(m));
            if (VK == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(v[j]) : "v"(m));
            if (VK == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(m));
            if (VK == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[j]) : "v"(w[(j + 1) & 7]));
            if (VK == 8) asm volatile("ds_read_b32 %0, %1" : "=v"(v[j]) : "v"(lds_addr));
            if (VK == 9) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(v[j]) : "v"(lds_addr), "v"(m));
            if (VK == 10) asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(v[j]) : "v"(lds_addr), "s"(rsrc));
            if (VK == 11) asm volatile("ds_write_b128 %0, %1" : : "v"(lds_addr), "v"(w4) : "memory");
            if (VK == 12) asm volatile("ds_read_b128 %0, %1" : "=v"(w4) : "v"(lds_addr));
            if ((VK >= 8) && j == 7) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
}

template <int MK, int VK, int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b) {
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    f32x4 q0 = {0}, q1 = {0}, q2 = {0}, q3 = {0}, q4 = {0}, q5 = {0}, q6 = {0}, q7 = {0};
    float v[8] = {a, b, a, b, a, b, a, b};
    f32x2 w[8];
    for (int j = 0; j < 8; j++) w[j] = (f32x2){a, b};
    __shared__ float lds[8192];
    lds[threadIdx.x] = a; lds[threadIdx.x + 512] = b;
    __syncthreads();
    const int lds_addr = (threadIdx.x & 63) * 16;
    f32x4 w4 = {a, b, a, b};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, 4096, 0x00020000);
    const int wave = threadIdx.x >> 6;
    const bool lowhalf = wave < 4;
    const bool do_m = (MODE == 0 && lowhalf) || (MODE == 2 && lowhalf) || (MODE == 3 && lowhalf) || MODE == 4;
    const bool do_v = (MODE == 1 && !lowhalf) || (MODE == 2 && !lowhalf) || (MODE == 3 && lowhalf);
    if (!do_m && !do_v) return;
    for (int i = 0; i < iters; i++) {
        if (do_m) {
            if (MK == 0) {              // 4 x 32x32x2: 4 x 64 cycles
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
            } else {                    // 8 x 16x16x4: 8 x 32 cycles
                q0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q0, 0, 0, 0); q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q1, 0, 0, 0);
                q2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q2, 0, 0, 0); q3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q3, 0, 0, 0);
                q4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q4, 0, 0, 0); q5 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q5, 0, 0, 0);
                q6 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q6, 0, 0, 0); q7 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q7, 0, 0, 0);
            }
        }
        if (do_v) valu64<VK>(v, w, a, b, lds_addr, rsrc, w4);
    }
    float s = 0;
    for (int r = 0; r < 16; r++) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    for (int r = 0; r < 4; r++) s += q0[r] + q1[r] + q2[r] + q3[r] + q4[r] + q5[r] + q6[r] + q7[r];
    for (int j = 0; j < 8; j++) s += v[j] + w[j].x + w[j].y;
    s += w4[0] + w4[3];
    if (s == 12345.f) out[threadIdx.x] = s + wave;
}
template <int MK, int VK, int MODE> float run(int iters) {
    float* d; hipMalloc(&d, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MK, VK, MODE><<<256, 512>>>(d, iters, 0.5f, 0.25f); hipDeviceSynchronize();
    hipEventRecord(e0); k<MK, VK, MODE><<<256, 512>>>(d, iters, 0.5f, 0.25f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(d); return ms;
}
template <int MK, int VK> void row(const char* name, int iters) {
    const float m = run<MK, VK, 0>(iters), v = run<MK, VK, 1>(iters), mv = run<MK, VK, 2>(iters), in = run<MK, VK, 3>(iters);
    // cycles per instruction at 2.4 GHz nominal: 64 VALU per iteration
    printf("%-14s mfma %s: M %.3f  V %.3f (%.2f cyc/inst @2.4GHz)  MV(two waves) %.3f  I(one wave) %.3f   [M+V %.3f, max %.3f]\n", name, MK == 0 ? "32x32x2" : "16x16x4", m, v,
           v * 1e-3 * 2.4e9 / (iters * 64.0), mv, in, m + v, m > v ? m : v);
}
int main() {
    const int iters = 20000;
    printf("M2 = two MFMA waves per SIMD: %.3f ms (one: %.3f)\n", run<0, 0, 4>(iters), run<0, 0, 0>(iters));
    row<0, 0>("v_fma_f32", iters); row<0, 1>("v_pk_fma_f32", iters); row<0, 2>("v_max_f32", iters); row<0, 3>("v_mul_f32", iters);
    row<0, 4>("v_add_u32", iters);
    row<1, 0>("v_fma_f32", iters); row<1, 1>("v_pk_fma_f32", iters); row<1, 4>("v_add_u32", iters);
    row<1, 8>("ds_read_b32", iters); row<1, 9>("ds_bpermute", iters); row<1, 10>("buffer_load", iters); row<1, 11>("ds_write_b128", iters); row<1, 12>("ds_read_b128", iters);
    return 0;
}
