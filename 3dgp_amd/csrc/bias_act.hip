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

// The compute type S is float for fp32 / fp16 / bf16 tensors and double for fp64 ones (bias_act.cpp:77 dispatches
// AT_DISPATCH_FLOATING_TYPES_AND_HALF; bias_act.cu:17-19: scalar_t = double for double tensors).  The overload set below picks the
// matching OCML routine; the float instantiations are the ones the generator path uses.
__device__ __forceinline__ float m_tanh(float x) { return tanhf(x); }
__device__ __forceinline__ double m_tanh(double x) { return tanh(x); }
__device__ __forceinline__ float m_exp(float x) { return expf(x); }
__device__ __forceinline__ double m_exp(double x) { return exp(x); }
__device__ __forceinline__ float m_expm1(float x) { return expm1f(x); }
__device__ __forceinline__ double m_expm1(double x) { return expm1(x); }
__device__ __forceinline__ float m_log1p(float x) { return log1pf(x); }
__device__ __forceinline__ double m_log1p(double x) { return log1p(x); }

template <int ACT, typename S>
__device__ __forceinline__ S act_fn(S x, S alpha) {
    if (ACT == 1) return x;
    if (ACT == 2) return x > S(0) ? x : S(0);
    if (ACT == 3) return x > S(0) ? x : x * alpha;
    if (ACT == 4) return m_tanh(x);
    if (ACT == 5) return S(1) / (S(1) + m_exp(-x));
    if (ACT == 6) return x > S(0) ? x : m_expm1(x);
    if (ACT == 7) {
        const S scale = (S)1.0507009873554804934193349852946;
        const S al = (S)1.6732632423543772848170429916717;
        return x > S(0) ? scale * x : (scale * al) * m_expm1(x);
    }
    if (ACT == 8) return x > S(20) ? x : m_log1p(m_exp(x));
    if (ACT == 9) return (S(1) / (S(1) + m_exp(-x))) * x;
    return x;
}

template <int ACT, typename S>
__device__ __forceinline__ S finish(S v, S alpha, S gain, S clamp) {
    v = act_fn<ACT, S>(v, alpha);
    v = v * gain;
    if (clamp >= S(0)) v = v < -clamp ? -clamp : (v > clamp ? clamp : v);   // NaN falls through (torch.clamp)
    return v;
}

template <typename T> struct Compute { typedef float type; };
template <> struct Compute<double> { typedef double type; };
template <typename T> __device__ __forceinline__ typename Compute<T>::type ld(const T* p, int64_t i);
template <> __device__ __forceinline__ double ld<double>(const double* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld<__half>(const __half* p, int64_t i) { return __half2float(p[i]); }
template <> __device__ __forceinline__ float ld<hip_bfloat16>(const hip_bfloat16* p, int64_t i) { return (float)p[i]; }
template <typename T> __device__ __forceinline__ void st(T* p, int64_t i, typename Compute<T>::type v);
template <> __device__ __forceinline__ void st<double>(double* p, int64_t i, double v) { p[i] = v; }
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
        o.x = finish<ACT, float>(v.x + b0, alpha, gain, clamp);
        o.y = finish<ACT, float>(v.y + b1, alpha, gain, clamp);
        o.z = finish<ACT, float>(v.z + b2, alpha, gain, clamp);
        o.w = finish<ACT, float>(v.w + b3, alpha, gain, clamp);
        y[i] = o;
    }
}

// generic scalar path (any dtype, any alignment, tails)
template <int ACT, typename T>
__global__ __launch_bounds__(256) void bias_act_scalar(const T* __restrict__ x, const T* __restrict__ b, T* __restrict__ y,
                                                       int64_t start, int64_t n, int sizeB, int64_t stepB,
                                                       float alpha, float gain, float clamp) {
    for (int64_t i = start + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        typedef typename Compute<T>::type S;
        S v = ld<T>(x, i);
        if (b) v = v + ld<T>(b, (i / stepB) % sizeB);
        st<T>(y, i, finish<ACT, S>(v, (S)alpha, (S)gain, (S)clamp));
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
    } else if (dtype == TDGP_F64) {
        const int blocks = (int)min((int64_t)maxBlocks, cdiv64(n, 256));
        TDGP_LAUNCH("bias_act_scalar", (bias_act_scalar<ACT, double>), dim3(blocks), dim3(256), 0, s, (const double*)x, (const double*)b,
                           (double*)y, (int64_t)0, n, sizeB, stepB, alpha, gain, clamp);
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
    TDGP_CHECK(dtype >= TDGP_F32 && dtype <= TDGP_F64, TDGP_EINVAL, "bias_act: unsupported dtype %d", dtype);
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
template <int ACT, typename S>
__device__ __forceinline__ S act_grad(int G, S x, S xref, S yy, S alpha) {
    const S seluScale = (S)1.0507009873554804934193349852946, seluAlpha = (S)1.6732632423543772848170429916717;
    const S one = 1, two = 2, zero = 0;
    if (ACT == 1) return G == 1 ? x : zero;
    if (ACT == 2) return G == 1 ? (yy > zero ? x : zero) : zero;
    if (ACT == 3) return G == 1 ? (yy > zero ? x : x * alpha) : zero;
    if (ACT == 4) return G == 1 ? x * (one - yy * yy) : x * (one - yy * yy) * (-two * yy);
    if (ACT == 5) return G == 1 ? x * yy * (one - yy) : x * yy * (one - yy) * (one - two * yy);
    if (ACT == 6) return G == 1 ? (yy >= zero ? x : x * (yy + one)) : (yy >= zero ? zero : x * (yy + one));
    if (ACT == 7) return G == 1 ? (yy >= zero ? x * seluScale : x * (yy + seluScale * seluAlpha)) : (yy >= zero ? zero : x * (yy + seluScale * seluAlpha));
    if (ACT == 8) { const S c = m_exp(-yy); return G == 1 ? x * (one - c) : x * c * (one - c); }
    if (ACT == 9) {
        const S c = m_exp(xref), d = c + one;
        if (G == 1) return xref > S(40) ? x : x * c * (xref + d) / (d * d);
        return xref > S(40) ? zero : x * c * (xref * (two - d) + two * d) / (d * d * d);
    }
    return zero;
}

template <int ACT, typename T>
__global__ __launch_bounds__(256) void bias_act_grad_kernel(const T* __restrict__ x, const T* __restrict__ b, const T* __restrict__ xref,
                                                            const T* __restrict__ yref, const T* __restrict__ dy, T* __restrict__ y, int64_t n,
                                                            int sizeB, int64_t stepB, int G, float alpha_, float gain_, float clamp_) {
    typedef typename Compute<T>::type S;
    const S alpha = alpha_, gain = gain_, clamp = clamp_;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const S xv = ld<T>(x, i);
        S xr = xref ? ld<T>(xref, i) : S(0);
        S yr = yref ? ld<T>(yref, i) : S(0);
        const S dv = dy ? ld<T>(dy, i) : S(1);
        if (b) xr = xr + ld<T>(b, (i / stepB) % sizeB);
        const S yy = gain != S(0) ? yr / gain : S(0);
        S v = act_grad<ACT, S>(G, xv, xr, yy, alpha);
        if (ACT == 9) yr = xr < S(-80) ? S(0) : xr / (m_exp(-xr) + S(1)) * gain;        // swish saves x, not y: rebuild the forward output for the clamp
        v = v * (gain * dv);
        if (clamp >= S(0)) v = (yr > -clamp && yr < clamp) ? v : S(0);
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
    else if (dtype == TDGP_F64)
        TDGP_LAUNCH("bias_act_grad_kernel", (bias_act_grad_kernel<ACT, double>), dim3(blocks), dim3(256), 0, s, (const double*)x, (const double*)b, (const double*)xref,
                    (const double*)yref, (const double*)dy, (double*)y, n, sizeB, stepB, G, alpha, gain, clamp);
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
    TDGP_CHECK(dtype >= TDGP_F32 && dtype <= TDGP_F64, TDGP_EINVAL, "bias_act_grad: unsupported dtype %d", dtype);
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
