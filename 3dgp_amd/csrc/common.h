// common.h -- shared helpers for the gfx950 kernels of libtdgp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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

// Optional per-kernel timing (tdgp_profile_enable / tdgp_profile_report), the counterpart of the reference's
// `misc.profiled_function` ranges (misc.py:101-106).  While it is on, every launch goes through hipExtLaunchKernelGGL with a
// start and a stop event ATTACHED TO THE DISPATCH ITSELF: the pair reads the kernel's own begin / end timestamps on the launch
// stream -- the same ones rocprofv3's kernel trace reports -- instead of bracketing it with two marker packets, whose dispatch
// and marker latencies (and the cache write-back a marker triggers) inflated the r01 per-kernel figures by ~4 %.
bool tdgp_prof_on();
void tdgp_prof_events(const char* name, hipEvent_t* a, hipEvent_t* b);
#define TDGP_LAUNCH(NAME, KERNEL, GRID, BLOCK, LDS, STREAM, ...)                                      \
    do {                                                                                              \
        if (tdgp_prof_on()) {                                                                         \
            hipEvent_t pa_, pb_;                                                                      \
            tdgp_prof_events(NAME, &pa_, &pb_);                                                       \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, pa_, pb_, 0, __VA_ARGS__);        \
        } else {                                                                                      \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__);                        \
        }                                                                                             \
    } while (0)

// Per-DEVICE one-time setup: function attributes (the raised dynamic-LDS cap) and the CU count belong to a device, not to the
// process -- a `static bool` guard left the second GPU driven from one process without its attribute and sized its persistent grids by the
// first GPU's CU count.  The guard is a bit per device ordinal, set AFTER the statement ran (two racing threads both run it: idempotent).
#include <atomic>
#define TDGP_ONCE_PER_DEVICE(...)                                                              \
    do {                                                                                       \
        static std::atomic<uint64_t> done_[4];                                                 \
        int dev_ = 0;                                                                          \
        (void)hipGetDevice(&dev_);                                                             \
        const uint64_t bit_ = 1ull << (dev_ & 63);                                             \
        std::atomic<uint64_t>& w_ = done_[(dev_ >> 6) & 3];                                    \
        if (!(w_.load(std::memory_order_acquire) & bit_)) {                                    \
            __VA_ARGS__;                                                                       \
            w_.fetch_or(bit_, std::memory_order_release);                                      \
        }                                                                                      \
    } while (0)
int tdgp_cu_count();
// Device-fault word: one int in pinned, device-visible HOST memory per process.  A kernel whose bounded wait ran out (the
// producer / consumer ring of field_walk2.inc) ORs a bit into it instead of finishing silently with wrong results; every field / per-ray
// entry point reads it on the host (a plain load, no synchronisation) and fails with TDGP_ELAUNCH until tdgp_device_fault(1) clears it.
int* tdgp_fault_word();       // host == device address (pinned, mapped); nullptr if the allocation failed
#define TDGP_FAULT_CHECK(what)                                                                                                        \
    do {                                                                                                                              \
        int* fw_ = tdgp_fault_word();                                                                                                 \
        if (fw_ && __atomic_load_n(fw_, __ATOMIC_RELAXED) != 0) {                                                                     \
            tdgp_set_error("%s: an earlier launch reported a device fault (code %d: a bounded in-kernel wait ran out; its results are "   \
                           "invalid) -- tdgp_device_fault(1) reads and clears it", what, *fw_);                                       \
            return TDGP_ELAUNCH;                                                                                                      \
        }                                                                                                                             \
    } while (0)          // compute units of the CURRENT device (cached per device ordinal, thread-safe)

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
// fp64 cross-lane helpers on the DPP path (gfx9 row shifts / row broadcasts, wave shift, v_readlane): a scan is 12 VALU
// moves with no LDS round trip.  The ds_bpermute versions they replace put twelve DEPENDENT ~120-cycle shuffles on the
// critical path of every scan, which is what the per-ray kernels (4-8 waves per SIMD, nothing else to overlap) were
// waiting for.
template <int CTRL, int RMASK>
__device__ __forceinline__ double dpp_f64(double keep, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(keep), __double2loint(v), CTRL, RMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(keep), __double2hiint(v), CTRL, RMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
// inclusive prefix (sum or product) over the 64 lanes of a wave, fp64
template <bool PROD>
__device__ __forceinline__ double wave_scan_f64(double v) {
    const double id = PROD ? 1.0 : 0.0;
#define TDGP_SCAN_STEP(CTRL, RMASK) { const double o = dpp_f64<CTRL, RMASK>(id, v); v = PROD ? v * o : v + o; }
    TDGP_SCAN_STEP(0x111, 0xf)      // row_shr:1   (Hillis-Steele inside each row of 16 lanes)
    TDGP_SCAN_STEP(0x112, 0xf)      // row_shr:2
    TDGP_SCAN_STEP(0x114, 0xf)      // row_shr:4
    TDGP_SCAN_STEP(0x118, 0xf)      // row_shr:8
    TDGP_SCAN_STEP(0x142, 0xa)      // row_bcast:15 -> rows 1 and 3 take the total of the row before
    TDGP_SCAN_STEP(0x143, 0xc)      // row_bcast:31 -> rows 2 and 3 take the total of rows 0-1
#undef TDGP_SCAN_STEP
    return v;
}
// value of lane 63 in every lane (wave-uniform)
__device__ __forceinline__ double wave_last_f64(double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// lane i takes lane i-1's value, lane 0 takes `first`
__device__ __forceinline__ double wave_shr1_f64(double v, double first) { return dpp_f64<0x138, 0xf>(first, v); }
__device__ __forceinline__ double wave_sum_f64(double v) { return wave_last_f64(wave_scan_f64<false>(v)); }

// torch's thresholded softplus (beta 1, threshold 20), evaluated in fp64 and rounded once.
__device__ __forceinline__ float softplus20(float x) { return x > 20.f ? x : (float)log1p(exp((double)x)); }

// fp32 counterparts (the forward ray marchers): the same DPP scans on one register per value, and softplus as the reference's
// own fp32 formula log1p(exp(x)) on the accurate (<= 1 ulp) OCML routines.
template <int CTRL, int RMASK>
__device__ __forceinline__ float dpp_f32(float keep, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(v), CTRL, RMASK, 0xf, false));
}
template <bool PROD>
__device__ __forceinline__ float wave_scan_f32(float v) {
    const float id = PROD ? 1.0f : 0.0f;
#define TDGP_SCAN_STEP(CTRL, RMASK) { const float o = dpp_f32<CTRL, RMASK>(id, v); v = PROD ? v * o : v + o; }
    TDGP_SCAN_STEP(0x111, 0xf)
    TDGP_SCAN_STEP(0x112, 0xf)
    TDGP_SCAN_STEP(0x114, 0xf)
    TDGP_SCAN_STEP(0x118, 0xf)
    TDGP_SCAN_STEP(0x142, 0xa)
    TDGP_SCAN_STEP(0x143, 0xc)
#undef TDGP_SCAN_STEP
    return v;
}
// inclusive prefix maximum of non-negative ints over the 64 lanes (identity 0), the same DPP steps
__device__ __forceinline__ int wave_scan_max_i32(int v) {
#define TDGP_SCAN_STEP(CTRL, RMASK) { const int o = __builtin_amdgcn_update_dpp(0, v, CTRL, RMASK, 0xf, false); v = o > v ? o : v; }
    TDGP_SCAN_STEP(0x111, 0xf)
    TDGP_SCAN_STEP(0x112, 0xf)
    TDGP_SCAN_STEP(0x114, 0xf)
    TDGP_SCAN_STEP(0x118, 0xf)
    TDGP_SCAN_STEP(0x142, 0xa)
    TDGP_SCAN_STEP(0x143, 0xc)
#undef TDGP_SCAN_STEP
    return v;
}
__device__ __forceinline__ int wave_last_i32(int v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ float wave_last_f32(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ float wave_shr1_f32(float v, float first) { return dpp_f32<0x138, 0xf>(first, v); }
__device__ __forceinline__ float wave_sum_f32(float v) { return wave_last_f32(wave_scan_f32<false>(v)); }
// softplus (beta 1, threshold 20) of the forward marchers: max(x, 0) + log1p(t), t = exp(-|x|) <= 1, with log1p(t) = log(u) + (t - (u - 1)) / u
// for u = fl(1 + t) (the second term restores what the rounding of 1 + t dropped; u - 1 is exact).  About 40 vector instructions.
// The literal log1pf(expf(x)) of r02 is 131 -- OCML's log1pf alone is 121, all double-float arithmetic -- and that was a third of
// every instruction the two per-ray kernels issue (they are bound by their VALU count, profiles/r02_pmc_sq_mix.md).  Measured on
// gfx950 over [-30, 22] against the exactly rounded value (tools/dev/softplus_acc.hip): max 2.7 ulp / mean 0.33 ulp here, 1.6 / 0.26
// for log1pf(expf(x)), 1.5 / 0.28 for torch's own CPU softplus (Sleef): the same noise floor the reference itself sits on.
#ifndef TDGP_SOFTPLUS_OCML
#define TDGP_SOFTPLUS_OCML 0        // 1: the literal log1pf(expf(x)) (A/B timing and accuracy comparisons)
#endif
__device__ __forceinline__ float softplus20f(float x) {
    if (TDGP_SOFTPLUS_OCML) return x > 20.f ? x : log1pf(expf(x));
    const float t = expf(-fabsf(x));
    const float u = 1.0f + t;
    const float l1 = logf(u) + (t - (u - 1.0f)) * __builtin_amdgcn_rcpf(u);
    const float r = fmaxf(x, 0.0f) + l1;
    return x > 20.f ? x : r;
}
