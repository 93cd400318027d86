// upfirdn2d.hip -- zero-insert upsample -> pad/crop -> 2-D FIR -> decimate, per (n,c) plane.
//
// Replaces the reference's upfirdn2d plugin forward (src/torch_utils/ops/upfirdn2d.cpp:16-98,
// upfirdn2d.cu:29-200).  Closed form:
//   out[oy,ox] = gain * sum_{ky,kx} F[ky,kx] * U[oy*down + ky - pady0, ox*down + kx - padx0]
//   F = f (flip) or f flipped (no flip); U = zero-upsampled input.
// Three kernels: the LDS-staged 4x4 / down=1 kernel for the two forms on the generator path (F1: up 1, F2: up 2) at tileable sizes,
// its register-window sibling for small planes, and a generic fallback for any filter / up / down / strides.
// HBM-bound: ~4 B in + 4 B out per output for F1.
#include "common.h"

namespace {

struct UpfirdnParams {
    const void* x; const float* f; void* y;
    int N, C, inH, inW, outH, outW;
    int64_t xs[4], ys[4];
    int fH, fW, upx, upy, downx, downy, padx0, pady0, flip;
    float gain;
};

template <typename T> __device__ __forceinline__ float ldx(const void* p, int64_t i);
template <> __device__ __forceinline__ float ldx<float>(const void* p, int64_t i) { return ((const float*)p)[i]; }
template <> __device__ __forceinline__ float ldx<__half>(const void* p, int64_t i) { return __half2float(((const __half*)p)[i]); }
template <> __device__ __forceinline__ float ldx<hip_bfloat16>(const void* p, int64_t i) { return (float)((const hip_bfloat16*)p)[i]; }
template <typename T> __device__ __forceinline__ void stx(void* p, int64_t i, float v);
template <> __device__ __forceinline__ void stx<float>(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
template <> __device__ __forceinline__ void stx<__half>(void* p, int64_t i, float v) { ((__half*)p)[i] = __float2half(v); }
template <> __device__ __forceinline__ void stx<hip_bfloat16>(void* p, int64_t i, float v) { ((hip_bfloat16*)p)[i] = hip_bfloat16(v); }

// fp64 (upfirdn2d.cpp:63 dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF; the CUDA kernel's scalar_t is double for double tensors,
// upfirdn2d.cu:21-27: accumulation and the gain in double).  Off the hot path: generic kernel only.
template <> __device__ __forceinline__ float ldx<double>(const void* p, int64_t i) { return (float)((const double*)p)[i]; }
template <typename T> struct UpAcc { typedef float type; };
template <> struct UpAcc<double> { typedef double type; };

template <typename T>
__global__ __launch_bounds__(256) void upfirdn2d_generic(UpfirdnParams p) {
    typedef typename UpAcc<T>::type S;
    const int64_t total = (int64_t)p.N * p.C * p.outH * p.outW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int ox = (int)(i % p.outW);
        int64_t r = i / p.outW;
        int oy = (int)(r % p.outH); r /= p.outH;
        int c = (int)(r % p.C);
        int n = (int)(r / p.C);
        const int64_t xb = n * p.xs[0] + c * p.xs[1];
        S acc = 0;
        for (int ky = 0; ky < p.fH; ky++) {
            int uy = oy * p.downy + ky - p.pady0;
            if (uy < 0 || uy >= p.inH * p.upy || (uy % p.upy) != 0) continue;
            int iy = uy / p.upy;
            for (int kx = 0; kx < p.fW; kx++) {
                int ux = ox * p.downx + kx - p.padx0;
                if (ux < 0 || ux >= p.inW * p.upx || (ux % p.upx) != 0) continue;
                int ix = ux / p.upx;
                float fv = p.flip ? p.f[ky * p.fW + kx] : p.f[(p.fH - 1 - ky) * p.fW + (p.fW - 1 - kx)];
                const int64_t xi = xb + iy * p.xs[2] + ix * p.xs[3];
                if constexpr (sizeof(S) == 8) acc = __builtin_fma((double)fv, ((const double*)p.x)[xi], acc);
                else acc = fmaf_(fv, ldx<T>(p.x, xi), acc);
            }
        }
        const int64_t yi = n * p.ys[0] + c * p.ys[1] + oy * p.ys[2] + ox * p.ys[3];
        if constexpr (sizeof(S) == 8) ((double*)p.y)[yi] = acc * (double)p.gain;
        else stx<T>(p.y, yi, acc * p.gain);
    }
}

// 4x4 filter, down 1, UP in {1,2}, fp32, unit W stride on input and output.
// One thread produces 4 consecutive outputs of a row: the 4 (UP=1: 7-wide) input windows overlap, so each
// input row segment is loaded once into registers and reused across the 4 outputs and 4 filter columns.
template <int UP>
__global__ __launch_bounds__(256) void upfirdn2d_4x4(UpfirdnParams p) {
    float F[4][4];
#pragma unroll
    for (int ky = 0; ky < 4; ky++)
#pragma unroll
        for (int kx = 0; kx < 4; kx++)
            F[ky][kx] = (p.flip ? p.f[ky * 4 + kx] : p.f[(3 - ky) * 4 + (3 - kx)]) * p.gain;   // gain folded like upfirdn2d.py:196
    const int qW = (p.outW + 3) / 4;
    const int64_t total = (int64_t)p.N * p.C * p.outH * qW;
    const float* x = (const float*)p.x;
    float* y = (float*)p.y;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int oxq = (int)(i % qW);
        int64_t r = i / qW;
        int oy = (int)(r % p.outH); r /= p.outH;
        int c = (int)(r % p.C);
        int n = (int)(r / p.C);
        const int ox0 = oxq * 4;
        const float* xp = x + n * p.xs[0] + c * p.xs[1];
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        // upsampled-domain window: columns ox0 - padx0 .. ox0 + 3 + 3 - padx0  (7 wide)
        const int ux0 = ox0 - p.padx0;
#pragma unroll
        for (int ky = 0; ky < 4; ky++) {
            const int uy = oy + ky - p.pady0;
            if (uy < 0 || uy >= p.inH * UP) continue;
            if (UP == 2 && (uy & 1)) continue;
            const int iy = uy / UP;
            const float* row = xp + iy * p.xs[2];
            float win[7];
#pragma unroll
            for (int j = 0; j < 7; j++) {
                const int ux = ux0 + j;
                bool ok = ux >= 0 && ux < p.inW * UP && (UP == 1 || !(ux & 1));
                win[j] = ok ? row[ux / UP] : 0.f;
            }
#pragma unroll
            for (int o = 0; o < 4; o++)
#pragma unroll
                for (int kx = 0; kx < 4; kx++) acc[o] = fmaf_(F[ky][kx], win[o + kx], acc[o]);
        }
        float* yp = y + n * p.ys[0] + c * p.ys[1] + oy * p.ys[2] + ox0;
#pragma unroll
        for (int o = 0; o < 4; o++)
            if (ox0 + o < p.outW) yp[o] = acc[o];
    }
}

// 4x4 filter, down 1, UP in {1,2}, fp32, unit W stride: the LDS-staged form of the kernel above for planes large enough to tile
// (the two forms of the generator path at their hot sizes -- F1: [C, r+1, r+1] -> [C, r, r], F2: [96, r/2, r/2] -> [96, r, r]).
// Block = one 16 x 128 output tile of one (n, c) plane: the input window it needs ((16+3)/UP+1 rows x (128+3)/UP+1 columns, zeros
// outside the image = the padding) is fetched ONCE with coalesced loads (16 B per lane where the row pitch allows) into LDS, each
// thread then produces 2 x 4 consecutive outputs from LDS reads and stores 16 B per lane.  HBM traffic = input once + output once;
// the register-window kernel above re-reads every input row 4x (UP = 1) through L1.
template <int UP>
__global__ __launch_bounds__(256) void upfirdn2d_4x4_lds(UpfirdnParams p) {
    constexpr int TH = 16, TW = 128;
    constexpr int NR = (TH + 3) / UP + 2, NC = ((TW + 3) / UP + 2 + 3 + 3) & ~3;         // window rows / columns (columns: 16-B aligned start + slack)
    __shared__ __attribute__((aligned(16))) float tile[NR * NC];
    float F[4][4];
#pragma unroll
    for (int ky = 0; ky < 4; ky++)
#pragma unroll
        for (int kx = 0; kx < 4; kx++) F[ky][kx] = (p.flip ? p.f[ky * 4 + kx] : p.f[(3 - ky) * 4 + (3 - kx)]) * p.gain;
    const int tilesX = (p.outW + TW - 1) / TW, tilesY = (p.outH + TH - 1) / TH;
    const int64_t ntiles = (int64_t)p.N * p.C * tilesY * tilesX;
    const float* x = (const float*)p.x;
    float* y = (float*)p.y;
    const bool vec_in = (p.xs[2] & 3) == 0 && (p.xs[1] & 3) == 0 && (p.xs[0] & 3) == 0 && ((uintptr_t)x & 15) == 0;
    const bool vec_out = (p.ys[2] & 3) == 0 && (p.ys[1] & 3) == 0 && (p.ys[0] & 3) == 0 && ((uintptr_t)y & 15) == 0 && (p.outW & 3) == 0;
    auto fdiv = [](int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); };
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = (int)(t % tilesX);
        int64_t r = t / tilesX;
        const int ty = (int)(r % tilesY); r /= tilesY;
        const int c = (int)(r % p.C), n = (int)(r / p.C);
        const int oy0 = ty * TH, ox0 = tx * TW;
        const int iy0 = fdiv(oy0 - p.pady0, UP), ixa = fdiv(ox0 - p.padx0, UP) & ~3;       // first window row / 16-B aligned first window column
        const float* xp = x + n * p.xs[0] + c * p.xs[1];
        __syncthreads();
        for (int i = threadIdx.x; i < NR * (NC / 4); i += 256) {
            const int ry = i / (NC / 4), j = i % (NC / 4);
            const int iy = iy0 + ry, ix = ixa + 4 * j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < p.inH) {
                const float* row = xp + (int64_t)iy * p.xs[2];
                if (vec_in && ix >= 0 && ix + 3 < p.inW) v = *(const float4*)(row + ix);
                else {
                    if (ix >= 0 && ix < p.inW) v.x = row[ix];
                    if (ix + 1 >= 0 && ix + 1 < p.inW) v.y = row[ix + 1];
                    if (ix + 2 >= 0 && ix + 2 < p.inW) v.z = row[ix + 2];
                    if (ix + 3 >= 0 && ix + 3 < p.inW) v.w = row[ix + 3];
                }
            }
            *(float4*)&tile[ry * NC + 4 * j] = v;
        }
        __syncthreads();
        const int lx = (threadIdx.x & 31) * 4;
#pragma unroll
        for (int hrow = 0; hrow < 2; hrow++) {
            const int ly = (threadIdx.x >> 5) + hrow * 8;
            const int oy = oy0 + ly, ox = ox0 + lx;
            if (oy >= p.outH || ox >= p.outW) continue;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            if (UP == 1) {
#pragma unroll
                for (int ky = 0; ky < 4; ky++) {
                    const float* trow = &tile[(oy + ky - p.pady0 - iy0) * NC + (ox - p.padx0 - ixa)];
                    float win[7];
#pragma unroll
                    for (int j = 0; j < 7; j++) win[j] = trow[j];
#pragma unroll
                    for (int o = 0; o < 4; o++)
#pragma unroll
                        for (int kx = 0; kx < 4; kx++) acc[o] = fmaf_(F[ky][kx], win[o + kx], acc[o]);
                }
            } else {
                // zero-stuffed input: of the 4 x 4 taps only those landing on even upsampled coordinates see data -- two filter rows
                // (ky = ky0, ky0 + 2) and, per output, two filter columns
                const int ky0 = (p.pady0 - oy) & 1;
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    const int ky = ky0 + 2 * kk;
                    const float* trow = &tile[((oy + ky - p.pady0) / 2 - iy0) * NC - ixa];      // (oy + ky - pady0) is even and, where it matters, >= 2*iy0
                    const int uy = oy + ky - p.pady0;
                    if (uy < 2 * iy0) continue;                      // above the window: only ever a padding row (zeros)
#pragma unroll
                    for (int o = 0; o < 4; o++) {
                        const int kx0 = (p.padx0 - ox - o) & 1;
#pragma unroll
                        for (int k2 = 0; k2 < 2; k2++) {
                            const int kx = kx0 + 2 * k2;
                            const int ux = ox + o + kx - p.padx0;
                            const float fv = ky0 ? (kk ? (kx0 ? (k2 ? F[3][3] : F[3][1]) : (k2 ? F[3][2] : F[3][0])) : (kx0 ? (k2 ? F[1][3] : F[1][1]) : (k2 ? F[1][2] : F[1][0])))
                                                 : (kk ? (kx0 ? (k2 ? F[2][3] : F[2][1]) : (k2 ? F[2][2] : F[2][0])) : (kx0 ? (k2 ? F[0][3] : F[0][1]) : (k2 ? F[0][2] : F[0][0])));
                            const float xv = ux >= 2 * ixa ? trow[ux / 2] : 0.f;
                            acc[o] = fmaf_(fv, xv, acc[o]);
                        }
                    }
                }
            }
            float* yp = y + n * p.ys[0] + c * p.ys[1] + (int64_t)oy * p.ys[2] + ox;
            if (vec_out && ox + 3 < p.outW) *(float4*)yp = make_float4(acc[0], acc[1], acc[2], acc[3]);
            else
                for (int o = 0; o < 4 && ox + o < p.outW; o++) yp[o] = acc[o];
        }
    }
}

}  // namespace

TDGP_API int tdgp_upfirdn2d(const void* x, const float* f, void* y, int N, int C, int inH, int inW, const int64_t* x_strides,
                            int outH, int outW, const int64_t* y_strides, int fH, int fW, int upx, int upy, int downx,
                            int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain, int dtype,
                            tdgp_stream_t stream) {
    // precondition set of upfirdn2d.cpp:19-40
    TDGP_CHECK(x && f && y && x_strides && y_strides, TDGP_EINVAL, "upfirdn2d: null pointer");
    TDGP_CHECK(N > 0 && C > 0 && inH > 0 && inW > 0, TDGP_EINVAL, "upfirdn2d: x has zero size");
    TDGP_CHECK(fH >= 1 && fW >= 1, TDGP_EINVAL, "upfirdn2d: f must be at least 1x1");
    TDGP_CHECK(upx >= 1 && upy >= 1, TDGP_EINVAL, "upfirdn2d: upsampling factor must be at least 1");
    TDGP_CHECK(downx >= 1 && downy >= 1, TDGP_EINVAL, "upfirdn2d: downsampling factor must be at least 1");
    TDGP_CHECK(dtype >= TDGP_F32 && dtype <= TDGP_F64, TDGP_EINVAL, "upfirdn2d: unsupported dtype %d", dtype);
    const int ow = (inW * upx + padx0 + padx1 - fW + downx) / downx;
    const int oh = (inH * upy + pady0 + pady1 - fH + downy) / downy;
    TDGP_CHECK(ow >= 1 && oh >= 1, TDGP_EINVAL, "upfirdn2d: output must be at least 1x1");
    TDGP_CHECK(ow == outW && oh == outH, TDGP_EINVAL, "upfirdn2d: output size mismatch (%dx%d vs %dx%d)", outH, outW, oh, ow);
    TDGP_CHECK((int64_t)N * C * outH * outW <= INT32_MAX, TDGP_EINVAL, "upfirdn2d: output is too large");
    UpfirdnParams p;
    p.x = x; p.f = f; p.y = y; p.N = N; p.C = C; p.inH = inH; p.inW = inW; p.outH = outH; p.outW = outW;
    for (int i = 0; i < 4; i++) { p.xs[i] = x_strides[i]; p.ys[i] = y_strides[i]; }
    p.fH = fH; p.fW = fW; p.upx = upx; p.upy = upy; p.downx = downx; p.downy = downy; p.padx0 = padx0; p.pady0 = pady0;
    p.flip = flip; p.gain = gain;
    hipStream_t s = (hipStream_t)stream;
    const bool fast = dtype == TDGP_F32 && fH == 4 && fW == 4 && downx == 1 && downy == 1 && upx == upy && (upx == 1 || upx == 2) &&
                      x_strides[3] == 1 && y_strides[3] == 1;
    if (fast && outW >= 64 && outH >= 16 && pady0 >= 0 && padx0 >= 0 && pady0 <= 4 && padx0 <= 4) {
        // planes large enough to tile: the LDS-staged kernel (input read once, 16-B loads and stores)
        const int64_t ntiles = (int64_t)N * C * cdiv(outH, 16) * cdiv(outW, 128);
        const int blocks = (int)min((int64_t)(256 * 32), ntiles);
        if (upx == 1) TDGP_LAUNCH("upfirdn2d_4x4", (upfirdn2d_4x4_lds<1>), dim3(blocks), dim3(256), 0, s, p);
        else TDGP_LAUNCH("upfirdn2d_4x4", (upfirdn2d_4x4_lds<2>), dim3(blocks), dim3(256), 0, s, p);
    } else if (fast) {
        const int64_t total = (int64_t)N * C * outH * ((outW + 3) / 4);
        const int blocks = (int)min((int64_t)(256 * 16), cdiv64(total, 256));
        if (upx == 1) TDGP_LAUNCH("upfirdn2d_4x4", (upfirdn2d_4x4<1>), dim3(blocks), dim3(256), 0, s, p);
        else TDGP_LAUNCH("upfirdn2d_4x4", (upfirdn2d_4x4<2>), dim3(blocks), dim3(256), 0, s, p);
    } else {
        const int64_t total = (int64_t)N * C * outH * outW;
        const int blocks = (int)min((int64_t)(256 * 16), cdiv64(total, 256));
        if (dtype == TDGP_F32) TDGP_LAUNCH("upfirdn2d_generic", (upfirdn2d_generic<float>), dim3(blocks), dim3(256), 0, s, p);
        else if (dtype == TDGP_F16) TDGP_LAUNCH("upfirdn2d_generic", (upfirdn2d_generic<__half>), dim3(blocks), dim3(256), 0, s, p);
        else if (dtype == TDGP_F64) TDGP_LAUNCH("upfirdn2d_generic", (upfirdn2d_generic<double>), dim3(blocks), dim3(256), 0, s, p);
        else TDGP_LAUNCH("upfirdn2d_generic", (upfirdn2d_generic<hip_bfloat16>), dim3(blocks), dim3(256), 0, s, p);
    }
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}
