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
    float scale, inv_scale, g0, g1;
    int scale_is_pow2;     // scale is a power of two: x * (1/scale) == x / scale exactly
    int marcher;
    int ray_h, ray_w;      // > 0: rays form a [B, ray_h, ray_w] image -> waves walk 4x4-pixel tiles (cache locality); 0: linear point order
    int64_t R;             // rays per sample
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

// The FQ floats this lane owns of one texel (F = 4*FQ floats): piece c4 of every 64-B group (FQ % 4 == 0), else a contiguous run.
template <int FQ>
__device__ __forceinline__ void load_texel(const float* __restrict__ texel, int c4, float* v) {
    if constexpr (FQ % 4 == 0) {
#pragma unroll
        for (int j = 0; j < FQ / 4; j++) {
            const float4 t = *(const float4*)(texel + 16 * j + 4 * c4);
            v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
        }
    } else {
        load_vec<FQ>(texel + c4 * FQ, v);
    }
}

// correctly rounded x / 3 without the hardware division sequence (Markstein: q1 = fma(fma(-3,q0,x), r, q0))
__device__ __forceinline__ float div3(float x) {
    const float r = 0.333333343267440796f;        // RN(1/3)
    float q0 = x * r;
    float rem = fmaf_(-3.0f, q0, x);
    return fmaf_(rem, r, q0);
}

// Channel held in value slot s of the lane that serves k-slot q.  A texel (F floats) is fetched by 4 adjacent lanes as
// 16-B pieces: piece c4 of every 64-B group -> channels 16*j + 4*c4 + {0..3}; the lane with piece index c4 later feeds
// MFMA k-slot q = c4, so slot s = 4*j + i holds channel 16*j + 4*q + i.  (FQ < 4: scalar loads, channel = q*FQ + s.)
__host__ __device__ __forceinline__ int feat_of(int s, int q, int FQ) { return FQ % 4 == 0 ? 16 * (s >> 2) + 4 * q + (s & 3) : q * FQ + s; }

template <int FQ, int MT, bool TAPS>
__global__ __launch_bounds__(256, 2) void triplane_field_kernel(FieldParams p) {
    constexpr int F = FQ * 4;
    constexpr int HID = MT * 16;
    // MFMA A operands, one float per lane per k-step, stored [step][lane] (conflict-free ds_read_b32, shared by the 4 waves):
    //   layer 1: a0s[mt*FQ + s][lane]  = W0[mt*16 + (lane&15)][feat_of(s, lane>>4)] / sqrt(F)
    //   layer 2: a1s[mt*4 + r][lane]   = (lane&15) < 4 ? W1[lane&15][mt*16 + 4*(lane>>4) + r] * sqrt(2)/sqrt(HID) : 0
    __shared__ float a0s[MT * FQ * 64];
    __shared__ float a1s[MT * 4 * 64];
    __shared__ float b0s[HID];
    const float sqrt2 = 1.41421353816986083984375f;    // (float)sqrt(2): lrelu gain, folded into the layer-2 weights
    for (int i = threadIdx.x; i < MT * FQ * 64; i += blockDim.x) {
        const int ln = i & 63, ms = i >> 6, mt = ms / FQ, sidx = ms % FQ;
        a0s[i] = p.w0[(mt * 16 + (ln & 15)) * F + feat_of(sidx, ln >> 4, FQ)] * p.g0;
    }
    for (int i = threadIdx.x; i < MT * 4 * 64; i += blockDim.x) {
        const int ln = i & 63, ms = i >> 6, mt = ms >> 2, r = ms & 3, o = ln & 15;
        a1s[i] = o < 4 ? (p.w1[o * HID + mt * 16 + 4 * (ln >> 4) + r] * p.g1) * sqrt2 : 0.f;
    }
    for (int i = threadIdx.x; i < HID; i += blockDim.x) b0s[i] = p.b0[i];
    __syncthreads();

    const int l = lane_id();
    const int pt = l & 15, q = l >> 4;
    const float sx = (float)(p.W - 1) / 2.f, sy = (float)(p.H - 1) / 2.f;
    const int plane_elems = p.H * p.W * F;

    // Two lane layouts per 16-point tile:
    //   gather layout  lane = 4*pt + c4 : the 4 lanes of a point are ADJACENT and read adjacent 16-B pieces of each texel, so a
    //                  64-lane dwordx4 load is 16 fully used 64-B requests (lane = q*16+pt would make it 64 quarter-used ones:
    //                  the L1 tag rate, not bandwidth, was the limiter -- 1.7k of 1.9k cycles per tile);
    //   MFMA layout    lane = 16*q + pt : what v_mfma_f32_16x16x4_f32 wants (k-slot = lane >> 4).
    // The blended features hop from one to the other with FQ ds_bpermute_b32 (piece index c4 becomes k-slot q).
    const int gpt = (FQ % 4 == 0) ? (l >> 2) : pt, gc4 = (FQ % 4 == 0) ? (l & 3) : q;

    // ---- one 16-point tile.  (cx,cy,cz,gvalid,ggp) describe the point of THIS lane in the gather layout; (gp,valid) the point
    // whose result this lane stores (MFMA layout, q == 0 lanes).  `bplanes` = planes of the sample ------------------------------
    auto eval_tile = [&](float cx, float cy, float cz, const float* __restrict__ bplanes, int64_t ggp, bool gvalid, int64_t gp, bool valid) {
        float qc[3];
        if (p.scale_is_pow2) { qc[0] = cx * p.inv_scale; qc[1] = cy * p.inv_scale; qc[2] = cz * p.inv_scale; }   // exact == cx / scale
        else { qc[0] = cx / p.scale; qc[1] = cy / p.scale; qc[2] = cz / p.scale; }                                 // :576 true division

        // All 12 taps of the tile are put in flight before any of them is consumed (3 planes x 4 taps x FQ floats per lane).
        float tap[3][4][FQ];
        float wgt[3][4];
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            const float u = qc[pl == 2 ? 1 : 0];          // planes (x,y), (x,z), (y,z): width <- first coordinate (:577-581)
            const float v = qc[pl == 0 ? 1 : 2];
            const float ix = (u + 1.0f) * sx, iy = (v + 1.0f) * sy;   // align_corners=True unnormalisation
            const float fx = floorf(ix), fy = floorf(iy);
            const float tw = ix - fx, te = 1.0f - tw, tn = iy - fy, ts = 1.0f - tn;
            const float cfx = fx < -2.f ? -2.f : (fx > (float)p.W ? (float)p.W : fx);
            const float cfy = fy < -2.f ? -2.f : (fy > (float)p.H ? (float)p.H : fy);
            const int x0 = (int)cfx, y0 = (int)cfy;
            if (TAPS) {
                if (p.tap_idx && gc4 == 0 && gvalid) {
                    p.tap_idx[(ggp * 3 + pl) * 2 + 0] = x0;
                    p.tap_idx[(ggp * 3 + pl) * 2 + 1] = y0;
                }
            }
            const bool vx0 = x0 >= 0 && x0 < p.W, vx1 = x0 + 1 >= 0 && x0 + 1 < p.W;
            const bool vy0 = y0 >= 0 && y0 < p.H, vy1 = y0 + 1 >= 0 && y0 + 1 < p.H;
            // zero padding: an out-of-range tap keeps a clamped (in-bounds) address and gets weight 0
            wgt[pl][0] = (vx0 && vy0) ? ts * te : 0.f;    // nw
            wgt[pl][1] = (vx1 && vy0) ? ts * tw : 0.f;    // ne
            wgt[pl][2] = (vx0 && vy1) ? tn * te : 0.f;    // sw
            wgt[pl][3] = (vx1 && vy1) ? tn * tw : 0.f;    // se
            const float* base = bplanes + pl * plane_elems;
            const int xa = min(max(x0, 0), p.W - 1), xb = min(max(x0 + 1, 0), p.W - 1);
            const int ya = min(max(y0, 0), p.H - 1), yb = min(max(y0 + 1, 0), p.H - 1);
            const int ra = ya * p.W, rb = yb * p.W;         // 32-bit element offsets inside one plane (< 2^31)
            load_texel<FQ>(base + (ra + xa) * F, gc4, tap[pl][0]);
            load_texel<FQ>(base + (ra + xb) * F, gc4, tap[pl][1]);
            load_texel<FQ>(base + (rb + xa) * F, gc4, tap[pl][2]);
            load_texel<FQ>(base + (rb + xb) * F, gc4, tap[pl][3]);
        }
        float g[FQ];
#pragma unroll
        for (int s = 0; s < FQ; s++) {
            float pa[3];
#pragma unroll
            for (int pl = 0; pl < 3; pl++)
                pa[pl] = fmaf_(tap[pl][3][s], wgt[pl][3], fmaf_(tap[pl][2][s], wgt[pl][2], fmaf_(tap[pl][1][s], wgt[pl][1], tap[pl][0][s] * wgt[pl][0])));
            g[s] = div3((pa[0] + pa[1]) + pa[2]);          // x.mean(dim=1)
        }
        if (FQ % 4 == 0) {                                 // gather layout -> MFMA layout: lane (q, pt) takes from lane 4*pt + q
            const int src = (pt * 4 + q) * 4;
#pragma unroll
            for (int s = 0; s < FQ; s++) g[s] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(g[s])));
        }

        // layer 1 on the matrix cores: h^T[hid x 16 pts] = W0s * g^T
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < FQ; s++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0s[(mt * FQ + s) * 64 + l], g[s], acc[mt], 0, 0, 0);

        // bias + lrelu(0.2) lane-locally (each lane owns 4*MT hidden units of ITS point), then layer 2 on the matrix cores as
        // out^T[16 (4 used) x 16 pts] = W1s * h^T: k-slot = lane quarter, exactly the layout layer 1 left behind; rows 0..3
        // of the result (r,g,b,sigma) land in the four accumulator registers of the q == 0 lanes.
        f32x4 o4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float v = acc[mt][r] + b0s[mt * 16 + 4 * q + r];
                const float h = fmaxf(v, 0.2f * v);                      // leaky_relu(v, 0.2); the sqrt(2) gain lives in a1s
                o4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1s[(mt * 4 + r) * 64 + l], h, o4, 0, 0, 0);
            }
        if (q == 0 && valid) {
            float o[4] = {p.b1[0] + o4[0], p.b1[1] + o4[1], p.b1[2] + o4[2], p.b1[3] + o4[3]};
            if (p.marcher == 1) {
#pragma unroll
                for (int c = 0; c < 3; c++) o[c] = (1.0f / (1.0f + expf(-o[c]))) * (1.f + 2.f * 0.001f) - 0.001f;
            }
            ((float4*)p.rgbs)[gp] = make_float4(o[0], o[1], o[2], o[3]);
        }
    };

    if (p.ray_w > 0) {
        // Image-coherent walk: a block owns an 8x8-pixel patch (wave = one 4x4 quadrant, lane pt = pixel in it) and marches
        // all S samples of those rays, so the 16 points of a tile are neighbours in texture space (a few texels apart) and
        // the patch's footprint stays in L1.  Patches are dealt to blocks so that each XCD (block id mod 8 on gfx950) owns a
        // contiguous band of the image and therefore a compact slice of the tri-planes in its private L2.
        const int pX = (p.ray_w + 7) / 8, pY = (p.ray_h + 7) / 8;
        const int npatch = (int)(p.total / p.P) * pX * pY;
        const int nb = gridDim.x, per = nb / 8;
        const int lb = (nb % 8 == 0) ? (blockIdx.x % 8) * per + blockIdx.x / 8 : blockIdx.x;
        const int wvi = threadIdx.x >> 6;
        auto ray_of = [&](int b, int py, int px, int tpt, bool& ok) {       // pixel tpt of this wave's 4x4 quadrant -> ray index
            const int y = py * 8 + (wvi >> 1) * 4 + (tpt >> 2), x = px * 8 + (wvi & 1) * 4 + (tpt & 3);
            ok = y < p.ray_h && x < p.ray_w;
            return b * (int)p.R + (ok ? y * p.ray_w + x : 0);                   // B*R*S < 2^31 (checked on the host)
        };
        for (int patch = lb; patch < npatch; patch += nb) {
            const int px = patch % pX, py = (patch / pX) % pY, b = patch / (pX * pY);       // uniform
            bool gok, sok;
            const int gray = ray_of(b, py, px, gpt, gok);      // the ray this lane gathers for
            const int sray = ray_of(b, py, px, pt, sok);       // the ray this lane stores for (q == 0 lanes)
            const float ox = p.ray_o[gray * 3 + 0], oy = p.ray_o[gray * 3 + 1], oz = p.ray_o[gray * 3 + 2];
            const float dxr = p.ray_d[gray * 3 + 0], dyr = p.ray_d[gray * 3 + 1], dzr = p.ray_d[gray * 3 + 2];
            const float* bplanes = p.planes + (int64_t)b * 3 * plane_elems;
            const float* tp = p.t + (int64_t)gray * p.S;
            for (int k = 0; k < p.S; k++) {
                const float tt = tp[k];
                eval_tile(ox + tt * dxr, oy + tt * dyr, oz + tt * dzr, bplanes, (int64_t)gray * p.S + k, gok,      // :141 (unfused mul, add)
                          (int64_t)sray * p.S + k, sok);
            }
        }
        return;
    }
    const int64_t ntiles = (p.total + 15) / 16;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t tile = wave0; tile < ntiles; tile += nwaves) {
        const int64_t gp = tile * 16 + pt;                 // stored by this lane (MFMA layout)
        const bool valid = gp < p.total;
        const int64_t ggp = tile * 16 + gpt;               // gathered by this lane
        const bool gvalid = ggp < p.total;
        const int64_t gpc = gvalid ? ggp : p.total - 1;
        const int b = (int)(gpc / p.P);
        float cx, cy, cz;
        if (p.coords) {
            cx = p.coords[gpc * 3 + 0]; cy = p.coords[gpc * 3 + 1]; cz = p.coords[gpc * 3 + 2];
        } else {
            const int64_t ray = gpc / p.S;
            const float tt = p.t[gpc];
            cx = p.ray_o[ray * 3 + 0] + tt * p.ray_d[ray * 3 + 0];
            cy = p.ray_o[ray * 3 + 1] + tt * p.ray_d[ray * 3 + 1];
            cz = p.ray_o[ray * 3 + 2] + tt * p.ray_d[ray * 3 + 2];
        }
        eval_tile(cx, cy, cz, p.planes + (int64_t)b * 3 * plane_elems, ggp, gvalid, gp, valid);
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

template <int FQ, int MT, bool TAPS>
void launch_field_t(const FieldParams& p, hipStream_t s) {
    int64_t want;
    if (p.ray_w > 0) want = (p.total / p.P) * cdiv(p.ray_w, 8) * cdiv(p.ray_h, 8);      // one 8x8-pixel patch per block
    else want = cdiv64((p.total + 15) / 16, 4);                                          // one 16-point tile per wave
    int blocks = (int)min((int64_t)(256 * 8), want);         // persistent-ish grid: <= 8 blocks per CU, blocks stride over the work
    if (blocks > 8) blocks -= blocks % 8;                    // whole rounds of the 8 XCDs (the in-kernel XCD remap needs it)
    TDGP_LAUNCH("triplane_field_kernel", (triplane_field_kernel<FQ, MT, TAPS>), dim3(blocks), dim3(256), 0, s, p);
}

template <int FQ, int MT>
void launch_field(const FieldParams& p, hipStream_t s) {
    if (p.tap_idx) { launch_field_t<FQ, MT, true>(p, s); return; }
    launch_field_t<FQ, MT, false>(p, s);
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
                                 int64_t P, int S, int ray_w, int F, int H, int W, int hid, float scale, int marcher, tdgp_stream_t stream) {
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
    { int ex; p.scale_is_pow2 = (frexpf(scale, &ex) == 0.5f) ? 1 : 0; p.inv_scale = 1.0f / scale; }
    TDGP_CHECK((int64_t)B * P <= INT32_MAX / 4 && (int64_t)3 * H * W * F <= INT32_MAX, TDGP_EINVAL, "triplane_field: tensor too large");
    p.g0 = (float)(1.0 / sqrt((double)F)); p.g1 = (float)(1.0 / sqrt((double)hid));    // weight_gain, layers.py:39
    p.marcher = marcher;
    p.R = coords ? 0 : P / S; p.ray_w = 0; p.ray_h = 0;
    if (!coords && ray_w > 0) {
        TDGP_CHECK((p.R % ray_w) == 0, TDGP_EINVAL, "triplane_field: ray_w=%d does not divide the %lld rays", ray_w, (long long)p.R);
        p.ray_w = ray_w; p.ray_h = (int)(p.R / ray_w);
    }
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
