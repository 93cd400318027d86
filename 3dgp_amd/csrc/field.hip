// field.hip -- fused tri-plane feature lookup + tiny MLP (the renderer's hot kernel).
//
// Replaces, per point, the reference's eager chain (src/training/tri_plane_renderer.py:560-588
// simple_tri_plane_renderer -> F.grid_sample(bilinear, align_corners=True, zeros) on 3 planes,
// networks_epigraf.py:46-68 TriPlaneMLP.forward: mean over planes, FC(F->hid, lrelu)*sqrt2, FC(hid->4)).
// The reference materialises a [3,F,P] gather result and a [P,hid] hidden tensor per 1M-point chunk
// (403 MB + 268 MB); here neither ever leaves registers.
//
// Layout / mapping (CDNA4, wave64):
//   * planes are plane-major channel-LAST [B,3,H,W,F]: one bilinear tap = one contiguous F*4-byte line
//     (128 B for F=32) instead of F strided 4-byte reads in NCHW;
//   * a wave processes tiles of 16 points; lane l = (pt = l & 15, q = l >> 4).  The 4 lanes that share a
//     point each gather a contiguous quarter of the channels (FQ = F/4 floats, 16/32-B vector loads) of the
//     12 taps, blend them, and hold FQ features of the plane-mean g[pt][q*FQ + s];
//   * layer 1 runs on the matrix cores as h^T[hid x 16pts] = W0s[hid x F] * g^T[F x 16pts] with
//     v_mfma_f32_16x16x4_f32: B operand = g (exactly the per-lane data the gather produced: k-slot = q),
//     A operand = W0s pre-arranged in registers; FQ k-steps per 16-row tile of hid, exact fp32;
//   * the accumulator layout then gives each lane 4*MT hidden units of ITS point, so bias + lrelu are
//     lane-local, and layer 2 (hid -> 4) is 4*4*MT lane-local FMAs + two cross-lane adds (xor 16, 32);
//   * lanes q == 0 store (r,g,b,sigma) as one float4: 16 lanes * 16 B = 256 B contiguous per tile.
// Roofline: 2*(F*hid + 4*hid) = 4608 MFMA/VALU flop and 12 taps * F * 4 = 1536 B of L1/L2-served gathers per
// point; the tri-plane of one image (100.7 MB at 512^2 x 96) is read from HBM once and then lives in
// L2 / Infinity Cache.  Output traffic 16 B per point.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct FieldParams {
    const float* planes;   // [B,3,H,W,F]
    const float* coords;   // [B,P,3] or null
    const float* ray_o;    // [B*R,3]
    const float* ray_d;    // [B*R,3]
    const float* t;        // [B*R*S]
    const float* w0; const float* b0; const float* w1; const float* b1;
    float* rgbs;           // [B*P,4]
    int32_t* tap_idx;      // [B*P,3,2] or null
    int64_t total;         // B*P
    int64_t P;
    int S, H, W;
    float scale, g0, g1;
    int marcher;
};

template <int N>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float* v) {
    if constexpr (N % 4 == 0) {
#pragma unroll
        for (int i = 0; i < N / 4; i++) {
            float4 t = ((const float4*)p)[i];
            v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
        }
    } else if constexpr (N % 2 == 0) {
#pragma unroll
        for (int i = 0; i < N / 2; i++) {
            float2 t = ((const float2*)p)[i];
            v[2 * i] = t.x; v[2 * i + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; i++) v[i] = p[i];
    }
}

// correctly rounded x / 3 without the hardware division sequence (Markstein: q1 = fma(fma(-3,q0,x), r, q0))
__device__ __forceinline__ float div3(float x) {
    const float r = 0.333333343267440796f;        // RN(1/3)
    float q0 = x * r;
    float rem = fmaf_(-3.0f, q0, x);
    return fmaf_(rem, r, q0);
}

template <int FQ, int MT>
__global__ __launch_bounds__(256) void triplane_field_kernel(FieldParams p) {
    constexpr int F = FQ * 4;
    constexpr int HID = MT * 16;
    // W1s arranged [mt][q][o][r] so that lane (q) reads its 4 weights for output o with one 16-B LDS read
    __shared__ __attribute__((aligned(16))) float w1s[MT * 4 * 4 * 4];
    __shared__ float b0s[HID];
    for (int i = threadIdx.x; i < MT * 64; i += blockDim.x) {
        int r = i & 3, o = (i >> 2) & 3, q = (i >> 4) & 3, mt = i >> 6;
        w1s[i] = p.w1[o * HID + mt * 16 + 4 * q + r] * p.g1;
    }
    for (int i = threadIdx.x; i < HID; i += blockDim.x) b0s[i] = p.b0[i];
    __syncthreads();

    const int l = lane_id();
    const int pt = l & 15, q = l >> 4;
    // A operand of layer 1: lane holds W0s[mt*16 + pt][q*FQ + s]
    float a0[MT][FQ];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int s = 0; s < FQ; s++) a0[mt][s] = p.w0[(mt * 16 + pt) * F + q * FQ + s] * p.g0;
    const float b1v[4] = {p.b1[0], p.b1[1], p.b1[2], p.b1[3]};
    const float sqrt2 = 1.41421353816986083984375f;    // (float)sqrt(2)
    const float sx = (float)(p.W - 1) / 2.f, sy = (float)(p.H - 1) / 2.f;

    const int64_t ntiles = (p.total + 15) / 16;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
        const int64_t gp = tile * 16 + pt;
        const bool valid = gp < p.total;
        const int64_t gpc = valid ? gp : p.total - 1;
        const int b = (int)(gpc / p.P);
        float cx, cy, cz;
        if (p.coords) {
            cx = p.coords[gpc * 3 + 0]; cy = p.coords[gpc * 3 + 1]; cz = p.coords[gpc * 3 + 2];
        } else {
            const int64_t ray = gpc / p.S;
            const float tt = p.t[gpc];
            cx = p.ray_o[ray * 3 + 0] + tt * p.ray_d[ray * 3 + 0];      // tri_plane_renderer.py:141 (unfused mul, add)
            cy = p.ray_o[ray * 3 + 1] + tt * p.ray_d[ray * 3 + 1];
            cz = p.ray_o[ray * 3 + 2] + tt * p.ray_d[ray * 3 + 2];
        }
        const float qc[3] = {cx / p.scale, cy / p.scale, cz / p.scale};   // :576 true division

        float g[FQ];
        float pl_acc[3][FQ];
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            const float u = qc[pl == 2 ? 1 : 0];          // planes (x,y), (x,z), (y,z): width <- first coordinate (:577-581)
            const float v = qc[pl == 0 ? 1 : 2];
            const float ix = (u + 1.0f) * sx, iy = (v + 1.0f) * sy;   // align_corners=True unnormalisation
            const float fx = floorf(ix), fy = floorf(iy);
            const float tw = ix - fx, te = 1.0f - tw, tn = iy - fy, ts = 1.0f - tn;
            const float nw = ts * te, ne = ts * tw, sw = tn * te, se = tn * tw;
            const float cfx = fx < -2.f ? -2.f : (fx > (float)p.W ? (float)p.W : fx);
            const float cfy = fy < -2.f ? -2.f : (fy > (float)p.H ? (float)p.H : fy);
            const int x0 = (int)cfx, y0 = (int)cfy;
            if (p.tap_idx && q == 0 && valid) {
                p.tap_idx[(gp * 3 + pl) * 2 + 0] = x0;
                p.tap_idx[(gp * 3 + pl) * 2 + 1] = y0;
            }
            const bool vx0 = x0 >= 0 && x0 < p.W, vx1 = x0 + 1 >= 0 && x0 + 1 < p.W;
            const bool vy0 = y0 >= 0 && y0 < p.H, vy1 = y0 + 1 >= 0 && y0 + 1 < p.H;
            const float* base = p.planes + ((int64_t)(b * 3 + pl) * p.H * p.W) * F + q * FQ;
            // clamped addresses (always in bounds); invalid taps are zeroed after the load
            const int xa = min(max(x0, 0), p.W - 1), xb = min(max(x0 + 1, 0), p.W - 1);
            const int ya = min(max(y0, 0), p.H - 1), yb = min(max(y0 + 1, 0), p.H - 1);
            float t00[FQ], t01[FQ], t10[FQ], t11[FQ];
            load_vec<FQ>(base + ((int64_t)ya * p.W + xa) * F, t00);
            load_vec<FQ>(base + ((int64_t)ya * p.W + xb) * F, t01);
            load_vec<FQ>(base + ((int64_t)yb * p.W + xa) * F, t10);
            load_vec<FQ>(base + ((int64_t)yb * p.W + xb) * F, t11);
            const bool m00 = vx0 && vy0, m01 = vx1 && vy0, m10 = vx0 && vy1, m11 = vx1 && vy1;
#pragma unroll
            for (int s = 0; s < FQ; s++) {
                float acc = (m00 ? t00[s] : 0.f) * nw;
                acc = acc + (m01 ? t01[s] : 0.f) * ne;
                acc = acc + (m10 ? t10[s] : 0.f) * sw;
                acc = acc + (m11 ? t11[s] : 0.f) * se;
                pl_acc[pl][s] = acc;
            }
        }
#pragma unroll
        for (int s = 0; s < FQ; s++) g[s] = div3((pl_acc[0][s] + pl_acc[1][s]) + pl_acc[2][s]);   // x.mean(dim=1)

        // layer 1 on the matrix cores
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < FQ; s++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[mt][s], g[s], acc[mt], 0, 0, 0);

        // bias + lrelu(0.2) * sqrt(2), then layer 2: lane-local partial dot products
        float o4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            float h[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float v = acc[mt][r] + b0s[mt * 16 + 4 * q + r];
                v = v > 0.f ? v : v * 0.2f;
                h[r] = v * sqrt2;
            }
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const float4 wv = *(const float4*)&w1s[((mt * 4 + q) * 4 + o) * 4];
                o4[o] = fmaf_(h[0], wv.x, o4[o]);
                o4[o] = fmaf_(h[1], wv.y, o4[o]);
                o4[o] = fmaf_(h[2], wv.z, o4[o]);
                o4[o] = fmaf_(h[3], wv.w, o4[o]);
            }
        }
#pragma unroll
        for (int o = 0; o < 4; o++) {
            o4[o] += __shfl_xor(o4[o], 16, 64);
            o4[o] += __shfl_xor(o4[o], 32, 64);
            o4[o] = b1v[o] + o4[o];
        }
        if (q == 0 && valid) {
            if (p.marcher == 1) {
#pragma unroll
                for (int o = 0; o < 3; o++) o4[o] = (1.0f / (1.0f + expf(-o4[o]))) * (1.f + 2.f * 0.001f) - 0.001f;
            }
            ((float4*)p.rgbs)[gp] = make_float4(o4[0], o4[1], o4[2], o4[3]);
        }
    }
}

// NCHW planes [B,3F,H,W] -> [B,3,H,W,F] through an LDS tile of 64 pixels x F channels.
__global__ __launch_bounds__(256) void planes_to_hwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int F, int HW, int64_t ntiles,
                                                           int tiles_per_plane) {
    extern __shared__ float tile[];      // [F][65]
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t plane = t / tiles_per_plane;       // b*3 + pl
        const int p0 = (int)(t % tiles_per_plane) * 64;
        const int np = min(64, HW - p0);
        for (int i = threadIdx.x; i < F * 64; i += blockDim.x) {
            int c = i >> 6, px = i & 63;
            if (px < np) tile[c * 65 + px] = src[(plane * F + c) * (int64_t)HW + p0 + px];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < F * 64; i += blockDim.x) {
            int px = i / F, c = i % F;
            if (px < np) dst[(plane * (int64_t)HW + p0 + px) * F + c] = tile[c * 65 + px];
        }
        __syncthreads();
    }
}

template <int FQ, int MT>
void launch_field(const FieldParams& p, hipStream_t s) {
    const int64_t ntiles = (p.total + 15) / 16;
    const int64_t want = cdiv64(ntiles, 4);                 // one tile per wave
    const int blocks = (int)min((int64_t)(256 * 8), want);  // persistent-ish grid: 8 blocks per CU, waves stride over tiles
    TDGP_LAUNCH("triplane_field_kernel", (triplane_field_kernel<FQ, MT>), dim3(blocks), dim3(256), 0, s, p);
}

}  // namespace

TDGP_API int tdgp_planes_to_hwc(const float* planes_nchw, float* planes_hwc, int B, int F, int H, int W, tdgp_stream_t stream) {
    TDGP_CHECK(planes_nchw && planes_hwc, TDGP_EINVAL, "planes_to_hwc: null pointer");
    TDGP_CHECK(B >= 0 && F >= 1 && F <= 256 && H >= 1 && W >= 1, TDGP_EINVAL, "planes_to_hwc: bad shape");
    if (B == 0) return TDGP_OK;
    const int HW = H * W;
    const int tpp = cdiv(HW, 64);
    const int64_t ntiles = (int64_t)B * 3 * tpp;
    TDGP_LAUNCH("planes_to_hwc_kernel", planes_to_hwc_kernel, dim3((int)min((int64_t)65535, ntiles)), dim3(256), F * 65 * sizeof(float), (hipStream_t)stream,
                       planes_nchw, planes_hwc, F, HW, ntiles, tpp);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

TDGP_API int tdgp_triplane_field(const float* planes_hwc, const float* coords, const float* ray_o, const float* ray_d, const float* t,
                                 const float* w0, const float* b0, const float* w1, const float* b1, float* rgbs, int32_t* tap_idx, int B,
                                 int64_t P, int S, int F, int H, int W, int hid, float scale, int marcher, tdgp_stream_t stream) {
    TDGP_CHECK(planes_hwc && w0 && b0 && w1 && b1 && rgbs, TDGP_EINVAL, "triplane_field: null pointer");
    TDGP_CHECK(coords || (ray_o && ray_d && t && S >= 1), TDGP_EINVAL, "triplane_field: need coords, or ray_o/ray_d/t with S >= 1");
    TDGP_CHECK(B >= 0 && P >= 0 && H >= 2 && W >= 2, TDGP_EINVAL, "triplane_field: bad shape");
    TDGP_CHECK(marcher == 0 || marcher == 1, TDGP_EINVAL, "triplane_field: unknown ray marcher %d", marcher);
    TDGP_CHECK(coords || (P % S) == 0, TDGP_EINVAL, "triplane_field: P must be a multiple of S in ray mode");
    if (B == 0 || P == 0) return TDGP_OK;
    FieldParams p;
    p.planes = planes_hwc; p.coords = coords; p.ray_o = ray_o; p.ray_d = ray_d; p.t = t;
    p.w0 = w0; p.b0 = b0; p.w1 = w1; p.b1 = b1; p.rgbs = rgbs; p.tap_idx = tap_idx;
    p.total = (int64_t)B * P; p.P = P; p.S = coords ? 1 : S; p.H = H; p.W = W; p.scale = scale;
    p.g0 = (float)(1.0 / sqrt((double)F)); p.g1 = (float)(1.0 / sqrt((double)hid));    // weight_gain, layers.py:39
    p.marcher = marcher;
    hipStream_t s = (hipStream_t)stream;
    bool ok = true;
#define FIELD_CASE(FF, HH) else if (F == FF && hid == HH) launch_field<FF / 4, HH / 16>(p, s);
    if (false) {}
    FIELD_CASE(32, 64) FIELD_CASE(32, 32) FIELD_CASE(32, 128) FIELD_CASE(32, 16)
    FIELD_CASE(16, 64) FIELD_CASE(16, 32) FIELD_CASE(16, 16)
    FIELD_CASE(8, 64) FIELD_CASE(8, 32) FIELD_CASE(8, 16)
    FIELD_CASE(64, 64) FIELD_CASE(64, 128)
    else ok = false;
#undef FIELD_CASE
    TDGP_CHECK(ok, TDGP_EUNSUPPORTED, "triplane_field: no kernel for feat_dim=%d, hid_dim=%d (need feat in {8,16,32,64}, hid in {16,32,64,128})", F, hid);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}
