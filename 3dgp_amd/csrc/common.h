// common.h -- shared helpers for the gfx950 kernels of libtdgp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bfloat16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/tdgp.h"

#define TDGP_API extern "C" __attribute__((visibility("default")))

// Thread-local error message (nothing throws across the ABI).
void tdgp_set_error(const char* fmt, ...);

#define TDGP_CHECK(cond, code, ...)            \
    do {                                       \
        if (!(cond)) {                         \
            tdgp_set_error(__VA_ARGS__);       \
            return (code);                     \
        }                                      \
    } while (0)

#define TDGP_LAUNCH_CHECK()                                                    \
    do {                                                                       \
        hipError_t e_ = hipGetLastError();                                     \
        if (e_ != hipSuccess) {                                                \
            tdgp_set_error("HIP launch failed: %s", hipGetErrorString(e_));    \
            return TDGP_ELAUNCH;                                               \
        }                                                                      \
    } while (0)

// Optional per-kernel timing (tdgp_profile_enable / tdgp_profile_report): HIP events recorded around every launch
// on the launch stream, the counterpart of the reference's `misc.profiled_function` ranges (misc.py:101-106).
bool tdgp_prof_on();
void tdgp_prof_begin(const char* name, hipStream_t s);
void tdgp_prof_end(hipStream_t s);
struct ProfScope {
    hipStream_t s; bool on;
    ProfScope(const char* name, hipStream_t st) : s(st), on(tdgp_prof_on()) { if (on) tdgp_prof_begin(name, s); }
    ~ProfScope() { if (on) tdgp_prof_end(s); }
};
#define TDGP_LAUNCH(NAME, KERNEL, GRID, BLOCK, LDS, STREAM, ...)              \
    do {                                                                      \
        ProfScope ps_(NAME, STREAM);                                          \
        hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__);    \
    } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Explicit fused / unfused arithmetic.  The library is compiled with -ffp-contract=off so that fp32
// chains that decide integer rows (bilinear tap indices, searchsorted, sort keys) round exactly like
// the reference's eager ops; fmaf_() is used where a fused multiply-add is wanted for speed.
__device__ __forceinline__ float fmaf_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// 64-lane wave helpers.
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ double shfl_up_f64(double v, int delta) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_up(lo, delta, 64);
    hi = __shfl_up(hi, delta, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_f64(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return __hiloint2double(hi, lo);
}
// inclusive prefix (sum or product) over the 64 lanes of a wave, fp64
template <bool PROD>
__device__ __forceinline__ double wave_scan_f64(double v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double o = shfl_up_f64(v, d);
        if (l >= d) v = PROD ? v * o : v + o;
    }
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
    return v;
}

// torch's thresholded softplus (beta 1, threshold 20), evaluated in fp64 and rounded once.
__device__ __forceinline__ float softplus20(float x) { return x > 20.f ? x : (float)log1p(exp((double)x)); }
