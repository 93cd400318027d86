// bias_act.hip -- y = clamp(gain * act(x + b[(i / stepB) % sizeB])) and its first / second derivative forms.
//
// Replaces the reference's bias_act plugin forward (src/torch_utils/ops/bias_act.cpp:32-90,
// bias_act.cu:24-147 with grad=0); semantics follow the CPU/PyTorch path `_bias_act_ref`
// (bias_act.py:91-120): torch activations (softplus threshold 20, expm1-based elu/selu), NaN propagates
// through the clamp.
//
// HBM-bound streaming op: 8 B/element algorithmic traffic.  16 B per lane (4 x fp32 or 8 x 16-bit)
// when the base pointers are 16-B aligned; the bias index is resolved once per vector when stepB is a
// multiple of the vector width (NCHW planes), per element otherwise.
#include "common.h"

namespace {

template <int ACT>
__device__ __forceinline__ float act_fn(float x, float alpha) {
    if (ACT == 1) return x;
    if (ACT == 2) return x > 0.f ? x : 0.f;
    if (ACT == 3) return x > 0.f ? x : x * alpha;
    if (ACT == 4) return tanhf(x);
    if (ACT == 5) return 1.0f / (1.0f + expf(-x));
    if (ACT == 6) return x > 0.f ? x : expm1f(x);
    if (ACT == 7) {
        const float scale = 1.0507009873554804934193349852946f;
        const float al = 1.6732632423543772848170429916717f;
        return x > 0.f ? scale * x : (scale * al) * expm1f(x);
    }
    if (ACT == 8) return x > 20.f ? x : log1pf(expf(x));
    if (ACT == 9) return (1.0f / (1.0f + expf(-x))) * x;
    return x;
}

template <int ACT>
__device__ __forceinline__ float finish(float v, float alpha, float gain, float clamp) {
    v = act_fn<ACT>(v, alpha);
    v = v * gain;
    if (clamp >= 0.f) v = v < -clamp ? -clamp : (v > clamp ? clamp : v);   // NaN falls through (torch.clamp)
    return v;
}

template <typename T> __device__ __forceinline__ float ld(const T* p, int64_t i);
template <> __device__ __forceinline__ float ld<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld<__half>(const __half* p, int64_t i) { return __half2float(p[i]); }
template <> __device__ __forceinline__ float ld<hip_bfloat16>(const hip_bfloat16* p, int64_t i) { return (float)p[i]; }
template <typename T> __device__ __forceinline__ void st(T* p, int64_t i, float v);
template <> __device__ __forceinline__ void st<float>(float* p, int64_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st<__half>(__half* p, int64_t i, float v) { p[i] = __float2half(v); }
template <> __device__ __forceinline__ void st<hip_bfloat16>(hip_bfloat16* p, int64_t i, float v) { p[i] = hip_bfloat16(v); }

// fp32 vector path: one float4 per lane per iteration.
template <int ACT, bool VEC_BIAS>
__global__ __launch_bounds__(256) void bias_act_f32x4(const float4* __restrict__ x, const float* __restrict__ b,
                                                      float4* __restrict__ y, int n4, int sizeB, int stepB,
                                                      float alpha, float gain, float clamp) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        float4 v = x[i];
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        if (b) {
            const int e = i * 4;
            if (VEC_BIAS) {
                b0 = b1 = b2 = b3 = b[(e / stepB) % sizeB];
            } else {
                b0 = b[(e / stepB) % sizeB];
                b1 = b[((e + 1) / stepB) % sizeB];
                b2 = b[((e + 2) / stepB) % sizeB];
                b3 = b[((e + 3) / stepB) % sizeB];
            }
        }
        float4 o;
        o.x = finish<ACT>(v.x + b0, alpha, gain, clamp);
        o.y = finish<ACT>(v.y + b1, alpha, gain, clamp);
        o.z = finish<ACT>(v.z + b2, alpha, gain, clamp);
        o.w = finish<ACT>(v.w + b3, alpha, gain, clamp);
        y[i] = o;
    }
}

// generic scalar path (any dtype, any alignment, tails)
template <int ACT, typename T>
__global__ __launch_bounds__(256) void bias_act_scalar(const T* __restrict__ x, const T* __restrict__ b, T* __restrict__ y,
                                                       int64_t start, int64_t n, int sizeB, int64_t stepB,
                                                       float alpha, float gain, float clamp) {
    for (int64_t i = start + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = ld<T>(x, i);
        if (b) v = v + ld<T>(b, (i / stepB) % sizeB);
        st<T>(y, i, finish<ACT>(v, alpha, gain, clamp));
    }
}

template <int ACT>
int launch(const void* x, const void* b, void* y, int64_t n, int sizeB, int64_t stepB, float alpha, float gain,
           float clamp, int dtype, hipStream_t s) {
    const int maxBlocks = 256 * 8;
    if (dtype == TDGP_F32) {
        const bool aligned = (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
        int64_t done = 0;
        if (aligned && n >= 4) {
            const int n4 = (int)(n / 4);
            const int blocks = (int)min((int64_t)maxBlocks, cdiv64(n4, 256));
            if (b && (stepB % 4) == 0)
                TDGP_LAUNCH("bias_act_f32x4", (bias_act_f32x4<ACT, true>), dim3(blocks), dim3(256), 0, s, (const float4*)x, (const float*)b,
                                   (float4*)y, n4, sizeB, (int)stepB, alpha, gain, clamp);
            else
                TDGP_LAUNCH("bias_act_f32x4", (bias_act_f32x4<ACT, false>), dim3(blocks), dim3(256), 0, s, (const float4*)x, (const float*)b,
                                   (float4*)y, n4, sizeB, (int)stepB, alpha, gain, clamp);
            done = (int64_t)n4 * 4;
        }
        if (done < n) {
            const int blocks = (int)min((int64_t)maxBlocks, cdiv64(n - done, 256));
            TDGP_LAUNCH("bias_act_scalar", (bias_act_scalar<ACT, float>), dim3(blocks), dim3(256), 0, s, (const float*)x, (const float*)b,
                               (float*)y, done, n, sizeB, stepB, alpha, gain, clamp);
        }
    } else if (dtype == TDGP_F16) {
        const int blocks = (int)min((int64_t)maxBlocks, cdiv64(n, 256));
        TDGP_LAUNCH("bias_act_scalar", (bias_act_scalar<ACT, __half>), dim3(blocks), dim3(256), 0, s, (const __half*)x, (const __half*)b,
                           (__half*)y, (int64_t)0, n, sizeB, stepB, alpha, gain, clamp);
    } else {
        const int blocks = (int)min((int64_t)maxBlocks, cdiv64(n, 256));
        TDGP_LAUNCH("bias_act_scalar", (bias_act_scalar<ACT, hip_bfloat16>), dim3(blocks), dim3(256), 0, s, (const hip_bfloat16*)x,
                           (const hip_bfloat16*)b, (hip_bfloat16*)y, (int64_t)0, n, sizeB, stepB, alpha, gain, clamp);
    }
    return 0;
}

}  // namespace

TDGP_API int tdgp_bias_act(const void* x, const void* b, void* y, int64_t n, int sizeB, int64_t stepB, int act, float alpha,
                           float gain, float clamp, int dtype, tdgp_stream_t stream) {
    // precondition set of bias_act.cpp:35-51
    TDGP_CHECK(x && y, TDGP_EINVAL, "bias_act: x and y must be device pointers");
    TDGP_CHECK(n >= 0 && n <= INT32_MAX, TDGP_EINVAL, "bias_act: x is too large");
    TDGP_CHECK(dtype >= TDGP_F32 && dtype <= TDGP_BF16, TDGP_EINVAL, "bias_act: unsupported dtype %d", dtype);
    TDGP_CHECK(act >= 1 && act <= 9, TDGP_EUNSUPPORTED, "bias_act: no kernel found for the specified activation func (%d)", act);
    TDGP_CHECK(!b || (sizeB >= 1 && stepB >= 1), TDGP_EINVAL, "bias_act: b has wrong number of elements / stride");
    if (n == 0) return TDGP_OK;
    if (!b) { sizeB = 1; stepB = 1; }
    hipStream_t s = (hipStream_t)stream;
    switch (act) {
#define CASE(A) case A: launch<A>(x, b, y, n, sizeB, stepB, alpha, gain, clamp, dtype, s); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9)
#undef CASE
    }
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

namespace {

// ---- gradient forms (bias_act.cu:24-147 with grad = 1, 2) --------------------------------------------
// grad 1:  y = dy_in * gain * act'(xref + b)          (x carries the incoming gradient; the derivative is written in terms of
// grad 2:  y = d2_in * dy * gain * act''(xref + b)     yy = yref / gain for the activations that save y, of xref for swish)
// and, with a clamp, zero wherever the forward output yref sat outside (-clamp, clamp).
template <int ACT>
__device__ __forceinline__ float act_grad(int G, float x, float xref, float yy, float alpha) {
    const float seluScale = 1.0507009873554804934193349852946f, seluAlpha = 1.6732632423543772848170429916717f;
    if (ACT == 1) return G == 1 ? x : 0.f;
    if (ACT == 2) return G == 1 ? (yy > 0.f ? x : 0.f) : 0.f;
    if (ACT == 3) return G == 1 ? (yy > 0.f ? x : x * alpha) : 0.f;
    if (ACT == 4) return G == 1 ? x * (1.f - yy * yy) : x * (1.f - yy * yy) * (-2.f * yy);
    if (ACT == 5) return G == 1 ? x * yy * (1.f - yy) : x * yy * (1.f - yy) * (1.f - 2.f * yy);
    if (ACT == 6) return G == 1 ? (yy >= 0.f ? x : x * (yy + 1.f)) : (yy >= 0.f ? 0.f : x * (yy + 1.f));
    if (ACT == 7) return G == 1 ? (yy >= 0.f ? x * seluScale : x * (yy + seluScale * seluAlpha)) : (yy >= 0.f ? 0.f : x * (yy + seluScale * seluAlpha));
    if (ACT == 8) { const float c = expf(-yy); return G == 1 ? x * (1.f - c) : x * c * (1.f - c); }
    if (ACT == 9) {
        const float c = expf(xref), d = c + 1.f;
        if (G == 1) return xref > 40.f ? x : x * c * (xref + d) / (d * d);
        return xref > 40.f ? 0.f : x * c * (xref * (2.f - d) + 2.f * d) / (d * d * d);
    }
    return 0.f;
}

template <int ACT, typename T>
__global__ __launch_bounds__(256) void bias_act_grad_kernel(const T* __restrict__ x, const T* __restrict__ b, const T* __restrict__ xref,
                                                            const T* __restrict__ yref, const T* __restrict__ dy, T* __restrict__ y, int64_t n,
                                                            int sizeB, int64_t stepB, int G, float alpha, float gain, float clamp) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float xv = ld<T>(x, i);
        float xr = xref ? ld<T>(xref, i) : 0.f;
        float yr = yref ? ld<T>(yref, i) : 0.f;
        const float dv = dy ? ld<T>(dy, i) : 1.f;
        if (b) xr = xr + ld<T>(b, (i / stepB) % sizeB);
        const float yy = gain != 0.f ? yr / gain : 0.f;
        float v = act_grad<ACT>(G, xv, xr, yy, alpha);
        if (ACT == 9) yr = xr < -80.f ? 0.f : xr / (expf(-xr) + 1.f) * gain;        // swish saves x, not y: rebuild the forward output for the clamp
        v = v * (gain * dv);
        if (clamp >= 0.f) v = (yr > -clamp && yr < clamp) ? v : 0.f;
        st<T>(y, i, v);
    }
}

template <int ACT>
void launch_grad(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y, int64_t n, int sizeB, int64_t stepB, int G,
                 float alpha, float gain, float clamp, int dtype, hipStream_t s) {
    const int blocks = (int)min((int64_t)(256 * 8), cdiv64(n, 256));
    if (dtype == TDGP_F32)
        TDGP_LAUNCH("bias_act_grad_kernel", (bias_act_grad_kernel<ACT, float>), dim3(blocks), dim3(256), 0, s, (const float*)x, (const float*)b, (const float*)xref,
                    (const float*)yref, (const float*)dy, (float*)y, n, sizeB, stepB, G, alpha, gain, clamp);
    else if (dtype == TDGP_F16)
        TDGP_LAUNCH("bias_act_grad_kernel", (bias_act_grad_kernel<ACT, __half>), dim3(blocks), dim3(256), 0, s, (const __half*)x, (const __half*)b, (const __half*)xref,
                    (const __half*)yref, (const __half*)dy, (__half*)y, n, sizeB, stepB, G, alpha, gain, clamp);
    else
        TDGP_LAUNCH("bias_act_grad_kernel", (bias_act_grad_kernel<ACT, hip_bfloat16>), dim3(blocks), dim3(256), 0, s, (const hip_bfloat16*)x, (const hip_bfloat16*)b,
                    (const hip_bfloat16*)xref, (const hip_bfloat16*)yref, (const hip_bfloat16*)dy, (hip_bfloat16*)y, n, sizeB, stepB, G, alpha, gain, clamp);
}

}  // namespace

TDGP_API int tdgp_bias_act_grad(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y, int64_t n, int sizeB,
                                int64_t stepB, int grad, int act, float alpha, float gain, float clamp, int dtype, tdgp_stream_t stream) {
    TDGP_CHECK(x && y, TDGP_EINVAL, "bias_act_grad: x and y must be device pointers");
    TDGP_CHECK(grad == 1 || grad == 2, TDGP_EINVAL, "bias_act_grad: grad must be 1 or 2 (0 is tdgp_bias_act)");
    TDGP_CHECK(n >= 0 && n <= INT32_MAX, TDGP_EINVAL, "bias_act_grad: x is too large");
    TDGP_CHECK(dtype >= TDGP_F32 && dtype <= TDGP_BF16, TDGP_EINVAL, "bias_act_grad: unsupported dtype %d", dtype);
    TDGP_CHECK(act >= 1 && act <= 9, TDGP_EUNSUPPORTED, "bias_act_grad: no kernel found for the specified activation func (%d)", act);
    TDGP_CHECK(!b || (sizeB >= 1 && stepB >= 1), TDGP_EINVAL, "bias_act_grad: b has wrong number of elements / stride");
    if (n == 0) return TDGP_OK;
    if (!b) { sizeB = 1; stepB = 1; }
    hipStream_t s = (hipStream_t)stream;
    switch (act) {
#define CASE(A) case A: launch_grad<A>(x, b, xref, yref, dy, y, n, sizeB, stepB, grad, alpha, gain, clamp, dtype, s); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9)
#undef CASE
    }
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}
