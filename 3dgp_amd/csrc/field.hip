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
//   * the accumulator layout then gives each lane 4*MT hidden units of ITS point, so lrelu is lane-local (the bias is
//     the accumulator's initial value), and layer 2 (hid -> 4) is 4*MT 16-block v_mfma_f32_4x4x1 steps + two cross-lane adds;
//   * ray walk: a wave marches the S samples of a 4x4-pixel quad; the taps of sample k+1 are in flight while sample k runs
//     its MLP, and results are parked in LDS and flushed as 128-B runs per ray (8 samples x 16 B).
// Roofline: 2*(F*hid + 4*hid) = 4608 MLP flop + 768 blend flop and 12 taps * F * 4 = 1536 B of L1/L2-served gathers per
// point; the tri-plane of one image (100.7 MB at 512^2 x 96) is read from HBM once and then lives in
// L2 / Infinity Cache.  Output traffic 16 B per point.
// Measured (r02, tools/dev/ubench_overlap2.hip): fp32 MFMA and every vector-ALU instruction of ANY wave of a SIMD issue through one
// port (5-7 cycles of matrix time per VALU instruction, 5.4 for a packed one), LDS and vector-memory instructions overlap.  Per
// 16-point tile this kernel issues 64 MFMAs (1152 cycles) and ~142 vector instructions (316 in r01) and takes ~2900 cycles per
// SIMD; 4.95 ms per 67 M points, matrix pipe 43 % busy.  DESIGN.md (e) lists what was tried on top and measured slower.
#include "common.h"

// Compile-time ablations for timing experiments (tools/dev/build_variant.sh): 1 = no tap loads, 2 = no layer-1 MFMAs,
// 4 = no stores, 32 = per-phase cycle counts of one wave (printed).  Always 0 in the shipped library.
#ifndef TDGP_WALK_WAVES
#define TDGP_WALK_WAVES 2      // waves per SIMD the table walk is compiled for (3: 168 registers, 24 spilled -- measured slower)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));     // v_pk_{mul,add,fma}_f32: two fp32 lanes per VALU slot, IEEE per element

struct FieldParams {
    const float* planes;   // [B,3,H,W,F]
    const float* coords;   // [B,P,3] or null
    const float* ray_o;    // [B*R,3]
    const float* ray_d;    // [B*R,3]
    const float* t;        // [B*R*S]
    const float* w0; const float* b0; const float* w1; const float* b1;
    float* rgbs;           // [B*P,4]
    const float* snoise;   // [B*P] standard-normal draws or null: sigma += snoise * snoise_std (tri_plane_renderer.py:185-186)
    float snoise_std;
    int32_t* tap_idx;      // [B*P,3,2] or null
    int64_t total;         // B*P
    int64_t P;
    int S, H, W;
    float scale, inv_scale, g0, g1;
    int scale_is_pow2;     // scale is a power of two: x * (1/scale) == x / scale exactly
    int marcher;
    int ray_h, ray_w;      // > 0: rays form a [B, ray_h, ray_w] image -> waves walk 4x4-pixel tiles (cache locality); 0: linear point order
    int64_t R;             // rays per sample
    uint32_t planes_bytes; // B*3*H*W*F*4 when it fits a buffer descriptor (< 4 GiB), else 0
    int* fault;            // the library's device-fault word (pinned host memory) or null
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

// The FQ floats this lane owns of one texel (F = 4*FQ floats), as FQ/2 register pairs: piece c4 of every 64-B group
// (FQ % 4 == 0), else a contiguous run.
template <int FQ>
__device__ __forceinline__ void load_texel(const float* __restrict__ texel, int c4, f32x2* v) {
    static_assert(FQ % 2 == 0, "feat_dim must be a multiple of 8");
    if constexpr (FQ % 4 == 0) {
#pragma unroll
        for (int j = 0; j < FQ / 4; j++) {
            const float4 t = *(const float4*)(texel + 16 * j + 4 * c4);
            v[2 * j] = (f32x2){t.x, t.y}; v[2 * j + 1] = (f32x2){t.z, t.w};
        }
    } else {
#pragma unroll
        for (int j = 0; j < FQ / 2; j++) {
            const float2 t = *(const float2*)(texel + c4 * FQ + 2 * j);
            v[j] = (f32x2){t.x, t.y};
        }
    }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// correctly rounded x / 3 without the hardware division sequence (Markstein: q1 = fma(fma(-3,q0,x), r, q0))
__device__ __forceinline__ f32x2 div3(f32x2 x) {
    const f32x2 r = {0.333333343267440796f, 0.333333343267440796f};        // RN(1/3)
    const f32x2 m3 = {-3.0f, -3.0f};
    const f32x2 q0 = x * r;
    const f32x2 rem = __builtin_elementwise_fma(m3, q0, x);
    return __builtin_elementwise_fma(rem, r, q0);
}

// Channel held in value slot s of the lane that serves k-slot q.  A texel (F floats) is fetched by 4 adjacent lanes as
// 16-B pieces: piece c4 of every 64-B group -> channels 16*j + 4*c4 + {0..3}; the lane with piece index c4 later feeds
// MFMA k-slot q = c4, so slot s = 4*j + i holds channel 16*j + 4*q + i.  (FQ < 4: scalar loads, channel = q*FQ + s.)
__host__ __device__ __forceinline__ int feat_of(int s, int q, int FQ) { return FQ % 4 == 0 ? 16 * (s >> 2) + 4 * q + (s & 3) : q * FQ + s; }

// WALK = true : the table-driven image walk alone (rays of an image, FQ % 4 == 0, planes addressable through one buffer
//                descriptor) -- the hot path, compiled as its own kernel so that its register budget is its own;
// WALK = false: every other mode (explicit coordinates, linear point order, the generic image walk).
template <int FQ, int MT, bool TAPS, bool WALK>
__device__ __forceinline__ void field_body(const FieldParams& p) {
    constexpr int F = FQ * 4;
    constexpr int HID = MT * 16;
    // MFMA A operands, one float per lane per k-step, stored [step][lane] (conflict-free ds_read_b32, shared by the 4 waves):
    //   layer 1: a0s[mt*FQ + s][lane]  = W0[mt*16 + (lane&15)][feat_of(s, lane>>4)] / sqrt(F)
    //   layer 2: a1s[mt*4 + r][lane]   = W1[lane&3][mt*16 + 4*(lane>>4) + r] * sqrt(2)/sqrt(HID)     (16-block 4x4x1 MFMA, see below)
    __shared__ float a0s[MT * FQ * 64];
    __shared__ float a1s[MT * 4 * 64];
    __shared__ __attribute__((aligned(16))) float b0s[HID];
    constexpr int NW = 4;
    constexpr int PW = 8;             // patch = 8 x 8 pixels: one 4 x 4-pixel quad per wave
    __shared__ float4 obuf_all[NW * 16 * 9];     // per wave: [16 rays][8 samples (+1 pad)] parked outputs of the ray walk
    __shared__ uint4 atab_all[(FQ % 4 == 0) ? NW * 6 * 64 : 1];   // per wave: tap table of 4 samples x 16 rays, [3 planes x {weights, texel offsets}][sample*16 + ray]
    const float sqrt2 = 1.41421353816986083984375f;    // (float)sqrt(2): lrelu gain, folded into the layer-2 weights
    for (int i = threadIdx.x; i < MT * FQ * 64; i += blockDim.x) {
        const int ln = i & 63, ms = i >> 6, mt = ms / FQ, sidx = ms % FQ;
        a0s[i] = (p.w0[(mt * 16 + (ln & 15)) * F + feat_of(sidx, ln >> 4, FQ)] * p.g0) / 3.0f;      // x.mean(dim=1) over the 3 planes folded in (see blend)
    }
    for (int i = threadIdx.x; i < MT * 4 * 64; i += blockDim.x) {
        const int ln = i & 63, ms = i >> 6, mt = ms >> 2, r = ms & 3;
        a1s[i] = (p.w1[(ln & 3) * HID + mt * 16 + 4 * (ln >> 4) + r] * p.g1) * sqrt2;
    }
    for (int i = threadIdx.x; i < HID; i += blockDim.x) b0s[i] = p.b0[i];
    __syncthreads();

    const int l = lane_id();
    const int pt = l & 15, q = l >> 4;
    // The table walk runs two waves per SIMD (256 registers each) and uses ~110: its MFMA A operands and the layer-1 bias stay in
    // registers -- 48 + 4*MT LDS reads per 16-point tile fewer, and in this loop every instruction of any kind costs ~7 cycles of issue
    // next to the MFMAs.  The generic kernel keeps reading them from LDS (its register budget is set by its other modes).
    constexpr bool REGW = WALK && MT * FQ <= 32;       // (wider MLPs would spill: they keep the LDS form)
    float a0r[REGW ? MT * FQ : 1], a1r[REGW ? MT * 4 : 1];
    f32x4 b0r[REGW ? MT : 1];
    if constexpr (REGW) {
#pragma unroll
        for (int i = 0; i < MT * FQ; i++) a0r[i] = a0s[i * 64 + l];
#pragma unroll
        for (int i = 0; i < MT * 4; i++) a1r[i] = a1s[i * 64 + l];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) b0r[mt] = *(const f32x4*)(b0s + mt * 16 + 4 * q);
    }
    auto a0 = [&](int i) -> float { if constexpr (REGW) return a0r[i]; else return a0s[i * 64 + l]; };
    auto a1 = [&](int i) -> float { if constexpr (REGW) return a1r[i]; else return a1s[i * 64 + l]; };
    const float sx = (float)(p.W - 1) / 2.f, sy = (float)(p.H - 1) / 2.f;
    // layer-2 bias: in registers, not reloaded per tile (a load there drags a vmcnt(0) into the loop); it is the initial accumulator of
    // the q == 0 lanes, so the cross-lane sum over q adds it exactly once
    const f32x4 o4init = (l >> 4) == 0 ? (f32x4){p.b1[0], p.b1[1], p.b1[2], p.b1[3]} : (f32x4){0.f, 0.f, 0.f, 0.f};
    const int plane_elems = p.H * p.W * F;

    // Two lane layouts per 16-point tile:
    //   gather layout  lane = 4*pt + c4 : the 4 lanes of a point are ADJACENT and read adjacent 16-B pieces of each texel, so a
    //                  64-lane dwordx4 load is 16 fully used 64-B requests (lane = q*16+pt would make it 64 quarter-used ones:
    //                  the L1 tag rate, not bandwidth, was the limiter -- 1.7k of 1.9k cycles per tile);
    //   MFMA layout    lane = 16*q + pt : what v_mfma_f32_16x16x4_f32 wants (k-slot = lane >> 4).
    // The blended features hop from one to the other with FQ ds_bpermute_b32 (piece index c4 becomes k-slot q).
    const int gpt = (FQ % 4 == 0) ? (l >> 2) : pt, gc4 = (FQ % 4 == 0) ? (l & 3) : q;

    // ---- one 16-point tile, in three stages so the ray walk can software-pipeline them:
    //   issue_taps : coordinates -> tap addresses + weights, all 12 taps put in flight (3 planes x 4 taps x FQ floats per lane);
    //   blend      : bilinear blend + plane mean -> g, moved to the MFMA layout;
    //   mlp_store  : both MLP layers on the matrix cores, (r,g,b,sigma) stored by the q == 0 lanes.
    // (cx,cy,cz,gvalid,ggp) describe the point of THIS lane in the gather layout; (gp,valid) the point whose result this lane
    // stores (MFMA layout).  `bplanes` = planes of the sample.
    f32x2 tap[3][4][FQ / 2];
    float wgt[3][4];
    auto issue_taps = [&](float cx, float cy, float cz, const float* __restrict__ bplanes, int64_t ggp, bool gvalid) {
        float qc[3];
        if (p.scale_is_pow2) { qc[0] = cx * p.inv_scale; qc[1] = cy * p.inv_scale; qc[2] = cz * p.inv_scale; }   // exact == cx / scale
        else { qc[0] = cx / p.scale; qc[1] = cy / p.scale; qc[2] = cz / p.scale; }                                 // :576 true division
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
    };
    // ---- the arithmetic of one 16-point tile, cut into PASSES of 16 channels (4 per gathering lane = one 16-B piece of every texel):
    //   blend_pass : g[16 channels of the pass] = sum over the 3 planes x 4 taps of weight * texel, two channels per VALU slot
    //                (v_pk_fma_f32), ONE fma chain per channel pair -- 12 instructions per pair, no separate plane sums / division: the
    //                mean's 1/3 lives in the layer-1 weights.  Vector instructions are paid for in matrix time on this chip (fp32 MFMA
    //                and VALU share the issue port), so the blend is written at its floor: 6 * FQ packed instructions per tile.  (The
    //                reference sums per plane and divides by 3: same value to fp32 rounding, different last bit; the parity tests
    //                bound it.)  The 4 blended values then hop from the gather layout to the MFMA layout (ds_bpermute);
    //   mlp_pass   : the 4 k-steps of layer 1 that consume those 16 channels, on all MT accumulator tiles;
    //   mlp_finish : lrelu, layer 2, the sum over the four hidden-unit quarters.
    // Every path of the kernel runs the same three functions in the same order, so results do not depend on the walk.
    constexpr int NP = (FQ % 4 == 0) ? FQ / 4 : 1;          // passes per tile
    constexpr int PP = (FQ % 4 == 0) ? 2 : FQ / 2;          // channel pairs per lane and pass
    auto blend_pass = [&](auto& T, int base, float* gp) {
#pragma unroll
        for (int s = 0; s < PP; s++) {
            f32x2 a = T[0][0][base + s] * (f32x2){wgt[0][0], wgt[0][0]};
#pragma unroll
            for (int j = 1; j < 12; j++) a = __builtin_elementwise_fma(T[j >> 2][j & 3][base + s], (f32x2){wgt[j >> 2][j & 3], wgt[j >> 2][j & 3]}, a);
            gp[2 * s] = a.x; gp[2 * s + 1] = a.y;
        }
        if (FQ % 4 == 0) {                                 // gather layout -> MFMA layout: lane (q, pt) takes from lane 4*pt + q
            const int src = (pt * 4 + q) * 4;
#pragma unroll
            for (int s = 0; s < 2 * PP; s++) gp[s] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(gp[s])));
        }
    };
    auto mlp_begin = [&](f32x4* acc) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) { if constexpr (REGW) acc[mt] = b0r[mt]; else acc[mt] = *(const f32x4*)(b0s + mt * 16 + 4 * q); }    // addmm(b, x, W^T): the bias (rows 4q..4q+3 of tile mt,
                                                                                             // one 16-B LDS read, not 4*MT resident registers) is the initial accumulator
    };
    auto mlp_pass = [&](f32x4* acc, int ps, const float* gp) {     // layer 1 on the matrix cores: h^T[hid x 16 pts] += W0s[:, pass] * g^T[pass]
#pragma unroll
        for (int i = 0; i < 2 * PP; i++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0(mt * FQ + ps * 2 * PP + i), gp[i], acc[mt], 0, 0, 0);
    };
    auto mlp_finish = [&](const f32x4* acc) -> float4 {
        // lrelu(0.2) lane-locally (each lane owns 4*MT hidden units of ITS point).  Layer 2 (hid -> 4) as sixteen 4x4x1
        // block MFMAs: the instruction multiplies 16 independent (4x1)*(1x4) blocks, block = lane >> 2.  With lane = 16*q + pt the
        // block is (q, pt >> 2): B = the lane's own hidden unit of point pt, A = W1[lane & 3][that unit], so after the 4*MT steps
        // lane (q, pt) holds the (r,g,b,sigma) partial sums of ITS point over the hidden units of quarter q; two cross-lane adds
        // finish them.  8 MFMA cycles per step instead of 32 for a 16x16x4 tile that would be 3/4 padding.
        f32x4 o4[2] = {o4init, (f32x4){0.f, 0.f, 0.f, 0.f}};       // the layer-2 bias rides in as the q == 0 lanes' initial accumulator
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            // leaky_relu(v, 0.2) = max(v, 0.2 v): one packed multiply per PAIR + one v_med3 per value.  The third med3 operand is
            // the largest finite float, not +inf: with +inf the compiler folds the med3 into fmaxf, which then needs the MFMA result
            // canonicalised first -- a second v_max per value (seen in the r01 ISA).  No inline asm here: a vector instruction the
            // hazard recognizer cannot see, next to MFMAs, is a wrong-result generator.  The sqrt(2) gain lives in a1s.
            const f32x2 c02 = {0.2f, 0.2f};
            const f32x2 lo = (f32x2){acc[mt][0], acc[mt][1]} * c02, hi = (f32x2){acc[mt][2], acc[mt][3]} * c02;
            const float sc[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float h = __builtin_amdgcn_fmed3f(acc[mt][r], sc[r], 3.4028234663852886e38f);
                o4[r & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1(mt * 4 + r), h, o4[r & 1], 0, 0, 0);
            }
        }
        const f32x2 s01 = (f32x2){o4[0][0], o4[0][1]} + (f32x2){o4[1][0], o4[1][1]}, s23 = (f32x2){o4[0][2], o4[0][3]} + (f32x2){o4[1][2], o4[1][3]};
        float o[4] = {s01.x, s01.y, s23.x, s23.y};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            o[c] += __shfl_xor(o[c], 16, 64);
            o[c] += __shfl_xor(o[c], 32, 64);
        }
        if (p.marcher == 1) {
#pragma unroll
            for (int c = 0; c < 3; c++) o[c] = (1.0f / (1.0f + expf(-o[c]))) * (1.f + 2.f * 0.001f) - 0.001f;
        }
        return make_float4(o[0], o[1], o[2], o[3]);        // (r,g,b,sigma) of point pt, valid in every lane of its column
    };
    auto tile_from_taps = [&]() -> float4 {                // all passes of a tile whose taps sit in tap[][][] (whole texel quarters)
        f32x4 acc[MT];
        mlp_begin(acc);
#pragma unroll
        for (int ps = 0; ps < NP; ps++) {
            float gp[2 * PP];
            blend_pass(tap, ps * PP, gp);
            mlp_pass(acc, ps, gp);
        }
        return mlp_finish(acc);
    };
    auto eval_tile = [&](float cx, float cy, float cz, const float* __restrict__ bplanes, int64_t ggp, bool gvalid, int64_t gp, bool valid) {
        issue_taps(cx, cy, cz, bplanes, ggp, gvalid);
        const float4 o = tile_from_taps();
        if (q == 0 && valid) {
            float4 on = o;
            if (p.snoise) on.w = __fadd_rn(on.w, __fmul_rn(p.snoise[gp], p.snoise_std));
            ((float4*)p.rgbs)[gp] = on;
        }
    };

    if (p.ray_w > 0) {
        // Image-coherent walk: a block owns an 8x8-pixel patch (wave = one 4x4 quadrant, lane pt = pixel in it) and marches
        // all S samples of those rays, so the 16 points of a tile are neighbours in texture space (a few texels apart) and
        // the patch's footprint stays in L1.  Patches are dealt to blocks so that each XCD (block id mod 8 on gfx950) owns a
        // contiguous band of the image and therefore a compact slice of the tri-planes in its private L2.
        const int pX = (p.ray_w + PW - 1) / PW, pY = (p.ray_h + 7) / 8;
        const int npatch = (int)(p.total / p.P) * pX * pY;
        const int nb = gridDim.x, per = nb / 8;
        const int lb = (nb % 8 == 0) ? (blockIdx.x % 8) * per + blockIdx.x / 8 : blockIdx.x;
        const int wvi = threadIdx.x >> 6;
        float4* obuf = obuf_all + wvi * (16 * 9);
        auto ray_of = [&](int b, int py, int px, int tpt, bool& ok) {       // pixel tpt of this wave's 4x4 quadrant -> ray index
            const int y = py * 8 + ((wvi >> 1) & 1) * 4 + (tpt >> 2), x = px * PW + (wvi & 1) * 4 + (tpt & 3);
            ok = y < p.ray_h && x < p.ray_w;
            return b * (int)p.R + (ok ? y * p.ray_w + x : 0);                   // B*R*S < 2^31 (checked on the host)
        };
        if constexpr (WALK) {
          {
            // Table-driven walk.  The four lanes that gather for one point would each repeat that point's coordinate -> tap
            // arithmetic (~135 VALU instructions per 16-point tile, a third of the kernel's vector work, and vector work is paid
            // in matrix time on this chip).  Instead the wave does it ONCE for four consecutive samples with lane = (ray, sample),
            // parks (4 weights, 4 texel byte offsets) x 3 planes per point in LDS, and each tile then reads its rows back (6
            // conflict-free 16-B LDS reads) and issues buffer loads: descriptor = all planes, scalar offset = the sample's plane,
            // lane offset = texel * F * 4 + piece * 16.
            uint4* atab = atab_all + wvi * (6 * 64);
            const uint32_t c4off = (uint32_t)gc4 * 16u;
            const uint32_t plane_bytes = (uint32_t)plane_elems * 4u;
            const __amdgpu_buffer_rsrc_t rpl = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.planes), 0, p.planes_bytes, 0x00020000);
            const int wslot = gc4 * 16 + gpt;                           // table slot this lane WRITES (sample j = gc4 of the group, ray gpt)
            for (int patch = lb; patch < npatch; patch += nb) {
                // patch id -> (sample, patch row, patch column).  Consecutive ids run through 8 x 8-patch (64 x 64-pixel) squares when the
                // image allows it: the 64 patches an XCD works on at a time (blocks lb = x * per .. + per - 1 of a round) then cover a compact
                // square instead of a 256 x 16-pixel band -- its footprint in the (x,z) plane is a quarter of the plane, not all of it, and the
                // eight private L2s fetch about half as much between them (FETCH_SIZE, profiles/).
                int px, py, b;
                if (((pX | pY) & 7) == 0) {
                    const int per_img = pX * pY, r = patch % per_img, sq = r >> 6, in = r & 63;
                    b = patch / per_img;
                    px = (sq % (pX >> 3)) * 8 + (in & 7); py = (sq / (pX >> 3)) * 8 + (in >> 3);
                } else {
                    px = patch % pX; py = (patch / pX) % pY; b = patch / (pX * pY);
                }
                bool gok, fok;
                const int gray = ray_of(b, py, px, gpt, gok);      // the ray this lane computes addresses / gathers for
                const int fray = ray_of(b, py, px, l >> 2, fok);   // the ray whose parked results this lane flushes
                const float ox = p.ray_o[gray * 3 + 0], oy = p.ray_o[gray * 3 + 1], oz = p.ray_o[gray * 3 + 2];
                const float dxr = p.ray_d[gray * 3 + 0], dyr = p.ray_d[gray * 3 + 1], dzr = p.ray_d[gray * 3 + 2];
                const uint32_t soff_b = (uint32_t)b * 3u * plane_bytes;
                const float* tp = p.t + (int64_t)gray * p.S;
                float t_grp = tp[min(gc4, p.S - 1)];
                auto address_phase = [&](int k0) {
                    // One (ray, sample) per lane, ~80 vector instructions per 64 points.  The four distinct (coordinate, axis) uses of
                    // the three planes -- x along W, y along H (plane xy), y along W (plane yz), z along H -- run as two packed pairs;
                    // every fp32 step is the reference's own (add, multiply, floor, subtract: unfused), so the tap indices stay bit-exact.
                    const int ks = min(k0 + gc4, p.S - 1);
                    const float tt = t_grp;
                    const f32x2 cxy = (f32x2){ox, oy} + (f32x2){tt, tt} * (f32x2){dxr, dyr};     // :141 (unfused mul, add)
                    const float cz = oz + tt * dzr;
                    f32x2 qxy; float qz;
                    if (p.scale_is_pow2) { qxy = cxy * (f32x2){p.inv_scale, p.inv_scale}; qz = cz * p.inv_scale; }
                    else { qxy = (f32x2){cxy.x / p.scale, cxy.y / p.scale}; qz = cz / p.scale; }
                    const f32x2 one2 = {1.0f, 1.0f};
                    const f32x2 iA = ((f32x2){qxy.x, qxy.y} + one2) * (f32x2){sx, sy};          // (x -> W, y -> H)
                    const f32x2 iB = ((f32x2){qxy.y, qz} + one2) * (f32x2){sx, sy};             // (y -> W, z -> H)
                    const float iv[4] = {iA.x, iA.y, iB.x, iB.y};
                    const float lim[4] = {(float)p.W, (float)p.H, (float)p.W, (float)p.H};
                    const int nn[4] = {p.W, p.H, p.W, p.H};
                    f32x2 T[4];            // per axis use: (weight of the lower tap, weight of the upper tap), 0 where the tap is outside (zero padding)
                    int i0[4], ia[4], ib[4];
#pragma unroll
                    for (int a = 0; a < 4; a++) {
                        const float f = floorf(iv[a]);
                        const float tw = iv[a] - f;
                        const f32x2 t2 = __builtin_elementwise_fma((f32x2){tw, tw}, (f32x2){-1.0f, 1.0f}, (f32x2){1.0f, 0.0f});   // (1 - tw, tw), each exactly rounded
                        const float cf = __builtin_amdgcn_fmed3f(f, -2.f, lim[a]);
                        const int x0 = (int)cf;
                        i0[a] = x0;
                        const bool v0 = (unsigned)x0 < (unsigned)nn[a], v1 = (unsigned)(x0 + 1) < (unsigned)nn[a];
                        T[a] = (f32x2){v0 ? t2.x : 0.f, v1 ? t2.y : 0.f};
                        ia[a] = min(max(x0, 0), nn[a] - 1);
                        ib[a] = min(max(x0 + 1, 0), nn[a] - 1);
                    }
                    if (TAPS) {
                        if (p.tap_idx && gok && k0 + gc4 < p.S) {
                            int32_t* tdst = p.tap_idx + ((int64_t)gray * p.S + ks) * 6;
                            tdst[0] = i0[0]; tdst[1] = i0[1]; tdst[2] = i0[0]; tdst[3] = i0[3]; tdst[4] = i0[2]; tdst[5] = i0[3];
                        }
                    }
                    const uint32_t rowb = (uint32_t)p.W * (F * 4u);                            // bytes per texel row (< 2^24: W * F * 4)
                    const int ua[3] = {0, 0, 2}, va[3] = {1, 3, 3};                           // planes (x,y), (x,z), (y,z): width <- first coordinate (:577-581)
#pragma unroll
                    for (int pl = 0; pl < 3; pl++) {
                        const f32x2 wn = T[ua[pl]] * (f32x2){T[va[pl]].x, T[va[pl]].x};       // (nw, ne) = (te, tw) * ts
                        const f32x2 ws2 = T[ua[pl]] * (f32x2){T[va[pl]].y, T[va[pl]].y};      // (sw, se) = (te, tw) * tn
                        const uint32_t ca = (uint32_t)ia[ua[pl]] * (F * 4u), cb = (uint32_t)ib[ua[pl]] * (F * 4u);
                        const uint32_t ra = __umul24((uint32_t)ia[va[pl]], rowb), rb = __umul24((uint32_t)ib[va[pl]], rowb);
                        atab[(pl * 2 + 0) * 64 + wslot] = make_uint4(__float_as_uint(wn.x), __float_as_uint(wn.y), __float_as_uint(ws2.x), __float_as_uint(ws2.y));
                        atab[(pl * 2 + 1) * 64 + wslot] = make_uint4(ra + ca, ra + cb, rb + ca, rb + cb);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    // next group's depth: issued LAST -- the compiler waits with vmcnt(0) for this group's value above, and a load placed
                    // before that wait would be waited for as well (a full memory round trip every fourth sample)
                    __builtin_amdgcn_sched_barrier(0);
                    t_grp = tp[min(k0 + 4 + gc4, p.S - 1)];
                };
                // Software pipeline over the samples: all 12 texel quarters of sample k+1 (24 x 16 B per lane at F = 32) are put in flight
                // right after sample k has been blended, and travel while the matrix cores run sample k's MLP.  (Tried and measured in
                // r02: splitting the tile into two 16-channel passes with only one pass of taps in flight -- 48 instead of 96 tap
                // registers, three / four waves per SIMD instead of two -- is 5 % FASTER on the coarse pass and 35 % SLOWER on the fine
                // pass, whose importance samples are scattered in depth: there it is the number of loads in flight per wave that hides
                // the L2 / MALL latency, not the number of waves.)
                auto issue_from_table = [&](int j) {                   // sample j of the current group: this lane's point is ray gpt
                    const int rslot = j * 16 + gpt;
#pragma unroll
                    for (int pl = 0; pl < 3; pl++) {
                        const uint4 w4 = atab[(pl * 2 + 0) * 64 + rslot], o4 = atab[(pl * 2 + 1) * 64 + rslot];
                        wgt[pl][0] = __uint_as_float(w4.x); wgt[pl][1] = __uint_as_float(w4.y); wgt[pl][2] = __uint_as_float(w4.z); wgt[pl][3] = __uint_as_float(w4.w);
                        const uint32_t so = soff_b + (uint32_t)pl * plane_bytes;
                        const uint32_t ov[4] = {o4.x + c4off, o4.y + c4off, o4.z + c4off, o4.w + c4off};
#pragma unroll
                        for (int t = 0; t < 4; t++)
#pragma unroll
                            for (int jj = 0; jj < FQ / 4; jj++) {
                                const float4 v = buf_load4(rpl, ov[t] + 64u * jj, so);
                                tap[pl][t][2 * jj] = (f32x2){v.x, v.y}; tap[pl][t][2 * jj + 1] = (f32x2){v.z, v.w};
                            }
                    }
                };
                address_phase(0);
                issue_from_table(0);
                for (int k = 0; k < p.S; k++) {
                    float g[NP][2 * PP];
#pragma unroll
                    for (int ps = 0; ps < NP; ps++) blend_pass(tap, ps * PP, g[ps]);
                    if (k + 1 < p.S) {
                        if (((k + 1) & 3) == 0) address_phase(k + 1);
                        issue_from_table((k + 1) & 3);
                    }
                    f32x4 acc[MT];
                    mlp_begin(acc);
#pragma unroll
                    for (int ps = 0; ps < NP; ps++) mlp_pass(acc, ps, g[ps]);
                    const float4 o = mlp_finish(acc);
                    if (q == 0) obuf[pt * 9 + (k & 7)] = o;
                    if ((k & 7) == 7 || k + 1 == p.S) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        const int k0 = k & ~7, nk = (k & 7) + 1;
                        const int fr = l >> 2, j = (l & 3) * 2;
                        float4 v0 = obuf[fr * 9 + j], v1 = obuf[fr * 9 + j + 1];
                        float4* dst = (float4*)p.rgbs + (int64_t)fray * p.S + k0 + j;
                        if (p.snoise && fok) {
                            const float* np = p.snoise + (int64_t)fray * p.S + k0 + j;
                            if (j < nk) v0.w = __fadd_rn(v0.w, __fmul_rn(np[0], p.snoise_std));
                            if (j + 1 < nk) v1.w = __fadd_rn(v1.w, __fmul_rn(np[1], p.snoise_std));
                        }
                        if (fok) {
                            if (j < nk) dst[0] = v0;
                            if (j + 1 < nk) dst[1] = v1;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            return;
          }
        }
        if constexpr (!WALK)
        for (int patch = lb; patch < npatch; patch += nb) {
            const int px = patch % pX, py = (patch / pX) % pY, b = patch / (pX * pY);       // uniform
            bool gok;
            const int gray = ray_of(b, py, px, gpt, gok);      // the ray this lane gathers for
            bool fok;
            const int fray = ray_of(b, py, px, l >> 2, fok);   // the ray whose parked results this lane flushes
            const float ox = p.ray_o[gray * 3 + 0], oy = p.ray_o[gray * 3 + 1], oz = p.ray_o[gray * 3 + 2];
            const float dxr = p.ray_d[gray * 3 + 0], dyr = p.ray_d[gray * 3 + 1], dzr = p.ray_d[gray * 3 + 2];
            const float* bplanes = p.planes + (int64_t)b * 3 * plane_elems;
            const float* tp = p.t + (int64_t)gray * p.S;
            // Software pipeline over the samples of the patch's rays: the taps of sample k+1 are put in flight right after sample
            // k has been blended (its tap registers are free again), so they travel while the matrix cores run sample k's MLP.
            float tt = tp[0];
            float tn = p.S > 1 ? tp[1] : 0.f;
            issue_taps(ox + tt * dxr, oy + tt * dyr, oz + tt * dzr, bplanes, (int64_t)gray * p.S, gok);          // :141 (unfused mul, add)
            for (int k = 0; k < p.S; k++) {
                float g[NP][2 * PP];
#pragma unroll
                for (int ps = 0; ps < NP; ps++) blend_pass(tap, ps * PP, g[ps]);
                if (k + 1 < p.S) {
                    tt = tn;
                    tn = k + 2 < p.S ? tp[k + 2] : 0.f;
                    issue_taps(ox + tt * dxr, oy + tt * dyr, oz + tt * dzr, bplanes, (int64_t)gray * p.S + k + 1, gok);
                }
                f32x4 acc[MT];
                mlp_begin(acc);
#pragma unroll
                for (int ps = 0; ps < NP; ps++) mlp_pass(acc, ps, g[ps]);
                const float4 o = mlp_finish(acc);
                // Results are parked in a per-wave LDS tile [16 rays][8 samples] and flushed as 128-B runs per ray: stored
                // directly, the 16 points of a step sit S*16 B apart -- sixteen 16-B fragments of sixteen different lines (the
                // PMC write traffic was 3.3x the payload).
                if (q == 0) obuf[pt * 9 + (k & 7)] = o;
                if ((k & 7) == 7 || k + 1 == p.S) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const int k0 = k & ~7, nk = (k & 7) + 1;
                    const int fr = l >> 2, j = (l & 3) * 2;
                    float4 v0 = obuf[fr * 9 + j], v1 = obuf[fr * 9 + j + 1];
                    float4* dst = (float4*)p.rgbs + (int64_t)fray * p.S + k0 + j;
                    if (p.snoise && fok) {
                        const float* np = p.snoise + (int64_t)fray * p.S + k0 + j;
                        if (j < nk) v0.w = __fadd_rn(v0.w, __fmul_rn(np[0], p.snoise_std));
                        if (j + 1 < nk) v1.w = __fadd_rn(v1.w, __fmul_rn(np[1], p.snoise_std));
                    }
                    if (fok) {
                        if (j < nk) dst[0] = v0;
                        if (j + 1 < nk) dst[1] = v1;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        return;
    }
    if constexpr (WALK) return;
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

template <int FQ, int MT, bool TAPS>
__global__ __launch_bounds__(256, 2) void triplane_field_kernel(FieldParams p) { field_body<FQ, MT, TAPS, false>(p); }

template <int FQ, int MT, bool TAPS>
__global__ __launch_bounds__(256, TDGP_WALK_WAVES) void triplane_walk_kernel(FieldParams p) { field_body<FQ, MT, TAPS, true>(p); }

#include "field_walk2.inc"

// NCHW planes [B,3F,H,W] -> [B,3,H,W,F] through an LDS tile of 64 pixels x F channels.
// The lookup alone: mean over the three planes of the bilinear samples, feats [B,P,F] -- simple_tri_plane_renderer's input to TriPlaneMLP
// (tri_plane_renderer.py:575-586, networks_epigraf.py:55).  The path of the decoders the fused kernels do not cover (tri_plane.mlp.n_layers != 2,
// has_view_cond, widths outside its table; round 6): their MLP then runs as eager tensor ops on this output, exactly as the reference's.  Built for
// correctness, not speed -- thread = (point, 4 channels) -- and written in torch's own evaluation order: per plane nw * a, + ne * b, + sw * c, + se * d
// (separate multiplies and adds), ((f0 + f1) + f2) / 3.
__global__ __launch_bounds__(256) void triplane_features_kernel(const float* __restrict__ planes, const float* __restrict__ coords, float* __restrict__ feats,
                                                                int64_t total, int64_t P, int F, int H, int W, float scale) {
    const int fq = F >> 2;
    const float sx = (float)(W - 1) / 2.f, sy = (float)(H - 1) / 2.f;
    const int64_t plane_elems = (int64_t)H * W * F;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total * fq; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t gp = i / fq;
        const int c4 = (int)(i - gp * fq) * 4;
        const int64_t b = gp / P;
        const float* cp = coords + gp * 3;
        const float qc[3] = {cp[0] / scale, cp[1] / scale, cp[2] / scale};
        float4 acc[3];
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            const float u = qc[pl == 2 ? 1 : 0], v = qc[pl == 0 ? 1 : 2];
            const float ix = (u + 1.0f) * sx, iy = (v + 1.0f) * sy;
            const float fx = floorf(ix), fy = floorf(iy);
            const float tw = ix - fx, te = 1.0f - tw, tn = iy - fy, ts = 1.0f - tn;
            const float cfx = fx < -2.f ? -2.f : (fx > (float)W ? (float)W : fx), cfy = fy < -2.f ? -2.f : (fy > (float)H ? (float)H : fy);
            const int x0 = (int)cfx, y0 = (int)cfy;
            const float w4[4] = {ts * te, ts * tw, tn * te, tn * tw};
            const float* base = planes + (b * 3 + pl) * plane_elems + c4;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int x = x0 + (k & 1), y = y0 + (k >> 1);
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);                      // zero padding: the tap VALUE is 0, its weight stays (torch's order)
                if (x >= 0 && x < W && y >= 0 && y < H) t = *(const float4*)(base + ((int64_t)y * W + x) * F);
                if (k == 0) a = make_float4(t.x * w4[0], t.y * w4[0], t.z * w4[0], t.w * w4[0]);
                else a = make_float4(a.x + t.x * w4[k], a.y + t.y * w4[k], a.z + t.z * w4[k], a.w + t.w * w4[k]);
            }
            acc[pl] = a;
        }
        *(float4*)(feats + gp * F + c4) = make_float4(((acc[0].x + acc[1].x) + acc[2].x) / 3.0f, ((acc[0].y + acc[1].y) + acc[2].y) / 3.0f,
                                                      ((acc[0].z + acc[1].z) + acc[2].z) / 3.0f, ((acc[0].w + acc[1].w) + acc[2].w) / 3.0f);
    }
}

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

#ifndef TDGP_WALK2
#define TDGP_WALK2 1           // 0: the one-role table walk (triplane_walk_kernel) everywhere -- A/B builds
#endif

template <int FQ, int MT, bool TAPS>
void launch_field_t(const FieldParams& p, hipStream_t s) {
    int64_t want;
    const bool walk = FQ % 4 == 0 && p.ray_w > 0 && p.planes_bytes != 0;
    if constexpr (FQ % 4 == 0 && FQ <= 8 && TDGP_WALK2) {
        // producer / consumer walk (field_walk2.inc): whole groups of four samples, a ring that is shorter than a patch's march
        // (the walk addresses ray_o / ray_d / t through scalar base + 32-bit byte offset)
        if (walk && (p.S & 3) == 0 && p.S >= 16 && p.total * 4 < ((int64_t)1 << 32) && (p.total / p.S) * 12 < ((int64_t)1 << 32)) {
            using L = Walk2Lds<FQ, MT>;
            const int cus = tdgp_cu_count();
            TDGP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)triplane_walk2_kernel<FQ, MT, TAPS>, hipFuncAttributeMaxDynamicSharedMemorySize, L::total));
            const int ps = ((p.S & 15) == 0 && TDGP_WALK2_DEPTHSPLIT) ? 4 : 8;       // patch side (field_walk2.inc: DEPTHSPLIT)
            const int64_t npatch = (p.total / p.P) * cdiv(p.ray_w, ps) * cdiv(p.ray_h, ps);
            int blocks = (int)min((int64_t)cus, npatch);                // one 512-thread block per CU, each striding over the patches
            if (blocks > 8) blocks -= blocks % 8;
            TDGP_LAUNCH("triplane_field_kernel", (triplane_walk2_kernel<FQ, MT, TAPS>), dim3(blocks), dim3(512), L::total, s, p);
            return;
        }
    }
    const int threads = 256;
    if (p.ray_w > 0) want = (p.total / p.P) * cdiv(p.ray_w, 8) * cdiv(p.ray_h, 8);      // one 8x8-pixel patch per block
    else want = cdiv64((p.total + 15) / 16, 4);                                          // one 16-point tile per wave
    // Persistent grid = exactly the blocks the chip holds at once (LDS-limited: 3 per CU at F = 32, hid = 64), each striding over the
    // work.  r01 launched 8 per CU: 2048 blocks over 768 slots ran as 2.67 rounds in the time of 3 (11 % of the kernel idle in the
    // last round); with 768 blocks every block gets within one patch of the same number of patches.
    static std::atomic<int> resident_pc[2];                  // blocks per CU the occupancy query allows (a property of the kernel, not of the device ordinal)
    int per_cu = resident_pc[walk].load(std::memory_order_relaxed);
    if (per_cu == 0) {
        hipError_t e;
        if constexpr (FQ % 4 == 0) e = walk ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, triplane_walk_kernel<FQ, MT, TAPS>, 256, 0)
                                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, triplane_field_kernel<FQ, MT, TAPS>, 256, 0);
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, triplane_field_kernel<FQ, MT, TAPS>, 256, 0);
        if (e != hipSuccess || per_cu < 1) per_cu = 2;
        resident_pc[walk].store(per_cu, std::memory_order_relaxed);
    }
    const int resident_blocks = per_cu * tdgp_cu_count();
    int blocks = (int)min((int64_t)resident_blocks, want);
    if (blocks > 8) blocks -= blocks % 8;                    // whole rounds of the 8 XCDs (the in-kernel XCD remap needs it)
    if constexpr (FQ % 4 == 0) {
        if (walk) { TDGP_LAUNCH("triplane_field_kernel", (triplane_walk_kernel<FQ, MT, TAPS>), dim3(blocks), dim3(threads), 0, s, p); return; }
    }
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

TDGP_API int tdgp_triplane_features(const float* planes_hwc, const float* coords, float* feats, int B, int64_t P, int F, int H, int W, float scale,
                                    tdgp_stream_t stream) {
    TDGP_CHECK(planes_hwc && coords && feats, TDGP_EINVAL, "triplane_features: null pointer");
    TDGP_CHECK(B >= 0 && P >= 0 && H >= 2 && W >= 2 && F >= 4 && (F & 3) == 0, TDGP_EINVAL, "triplane_features: bad shape (feat_dim must be a multiple of 4)");
    TDGP_CHECK((int64_t)B * P <= INT32_MAX / 4, TDGP_EINVAL, "triplane_features: tensor too large");
    if (B == 0 || P == 0) return TDGP_OK;
    const int64_t work = (int64_t)B * P * (F >> 2);
    TDGP_LAUNCH("triplane_features_kernel", triplane_features_kernel, dim3((unsigned)std::min<int64_t>(65535 * 4, cdiv64(work, 256))), dim3(256), 0, (hipStream_t)stream,
                planes_hwc, coords, feats, (int64_t)B * P, P, F, H, W, scale);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

TDGP_API int tdgp_triplane_field(const float* planes_hwc, const float* coords, const float* ray_o, const float* ray_d, const float* t,
                                 const float* w0, const float* b0, const float* w1, const float* b1, const float* sigma_noise, float density_noise,
                                 float* rgbs, int32_t* tap_idx, int B, int64_t P, int S, int ray_w, int F, int H, int W, int hid, float scale,
                                 int marcher, tdgp_stream_t stream) {
    TDGP_CHECK(planes_hwc && w0 && b0 && w1 && b1 && rgbs, TDGP_EINVAL, "triplane_field: null pointer");
    TDGP_CHECK(coords || (ray_o && ray_d && t && S >= 1), TDGP_EINVAL, "triplane_field: need coords, or ray_o/ray_d/t with S >= 1");
    TDGP_CHECK(B >= 0 && P >= 0 && H >= 2 && W >= 2, TDGP_EINVAL, "triplane_field: bad shape");
    TDGP_CHECK(marcher == 0 || marcher == 1, TDGP_EINVAL, "triplane_field: unknown ray marcher %d", marcher);
    TDGP_CHECK(coords || (P % S) == 0, TDGP_EINVAL, "triplane_field: P must be a multiple of S in ray mode");
    TDGP_CHECK(!(density_noise > 0.f) || sigma_noise, TDGP_EINVAL, "triplane_field: density_noise > 0 needs the sigma_noise draws");
    TDGP_FAULT_CHECK("triplane_field");
    if (B == 0 || P == 0) return TDGP_OK;
    FieldParams p;
    p.planes = planes_hwc; p.coords = coords; p.ray_o = ray_o; p.ray_d = ray_d; p.t = t;
    p.w0 = w0; p.b0 = b0; p.w1 = w1; p.b1 = b1; p.rgbs = rgbs; p.tap_idx = tap_idx;
    p.snoise = density_noise > 0.f ? sigma_noise : nullptr; p.snoise_std = density_noise;
    p.total = (int64_t)B * P; p.P = P; p.S = coords ? 1 : S; p.H = H; p.W = W; p.scale = scale;
    { int ex; p.scale_is_pow2 = (frexpf(scale, &ex) == 0.5f) ? 1 : 0; p.inv_scale = 1.0f / scale; }
    TDGP_CHECK((int64_t)B * P <= INT32_MAX / 4 && (int64_t)3 * H * W * F <= INT32_MAX, TDGP_EINVAL, "triplane_field: tensor too large");
    p.g0 = (float)(1.0 / sqrt((double)F)); p.g1 = (float)(1.0 / sqrt((double)hid));    // weight_gain, layers.py:39
    p.marcher = marcher;
    { const int64_t pb = (int64_t)B * 3 * H * W * F * 4; p.planes_bytes = (pb < ((int64_t)1 << 32) - 65536 && (int64_t)W * F * 4 < (1 << 24)) ? (uint32_t)pb : 0u; }
    p.R = coords ? 0 : P / S; p.ray_w = 0; p.ray_h = 0;
    if (!coords && ray_w > 0) {
        TDGP_CHECK((p.R % ray_w) == 0, TDGP_EINVAL, "triplane_field: ray_w=%d does not divide the %lld rays", ray_w, (long long)p.R);
        p.ray_w = ray_w; p.ray_h = (int)(p.R / ray_w);
    }
    p.fault = tdgp_fault_word();
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
