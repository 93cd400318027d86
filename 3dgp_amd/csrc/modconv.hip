// modconv.hip -- modulated convolution of the StyleGAN2 tri-plane backbone on the fp32 matrix cores.
//
// Replaces (eval / fused path): src/training/networks_stylegan2.py:31-88 modulated_conv2d
//   -> conv2d_resample.py:46-141 -> conv2d_gradfix -> F.conv2d / F.conv_transpose2d (cuDNN grouped conv,
//   groups = batch) -> upfirdn2d (up layers) -> `x.add_(noise)` -> bias_act; and for ToRGB the skip
//   `img = upsample2d(img) + y` (networks_stylegan2.py:265-269).
//
// Formulation (identical algebra to the reference's own non-fused branch :67-76):
//   y[b,o,p] = act( d[b,o] * sum_{c,tap} W[o,c,tap] * (s[b,c] * x[b,c,p+tap]) + noise[p] + bias[o] ) * gain
//   d[b,o]   = rsqrt( sum_c s[b,c]^2 * (sum_tap W[o,c,tap]^2) + 1e-8 )
// so every sample shares ONE weight matrix (no per-sample weight materialisation: the reference builds a
// [B,Cout,Cin,3,3] tensor per call) and the modulation rides on the activation staging.
//
// Kernels: implicit GEMM, M = Cout, N = output pixels, K = Cin * taps, v_mfma_f32_32x32x2_f32 (exact fp32, no TF32 -- the
// reference disables TF32 too: training_loop.py:76-77).  Weights are pre-packed [Cin/4][tap][Cout][4 channels].
//   conv3_mfma_kernel<KS>   3x3 / 5x5 stride-1 layers whose width is a multiple of 32 (the hot ones): mask-free, zero rows between
//                           samples, 8-byte LDS fragment reads with immediate offsets, buffer-load staging;
//   upconv_mfma_kernel      the x2 layers: stride-2 transposed 3x3 conv over a linearised grid, all four output parities in one
//                           pass, dense parity-planar intermediate; followed by
//   fir_act_kernel          FIR(4x4, gain 4) + demod + noise + bias + activation on that intermediate;
//   torgb_mfma_kernel       1x1, Cout <= 96, channel-last plane output with the fused x2-upsampled skip;
//   conv_mfma_kernel        generic fallback (any width, k in {1,3,5}, NCHW or channel-last output, per-lane tap masks, "virtual
//                           rows" so 4x4 ... 16x16 maps fill a tile with several samples): the low-resolution layers and the
//                           op-level API forms the fast paths do not cover;
//   demod_kernel, splitk_reduce_kernel, pack_kernel, style_affine_kernel.
// Every conv kernel parks its accumulator tiles in a per-wave LDS tile and runs the same fused output stage (epilogue_tile):
// demodulation, noise, bias, activation, gain, clamp, the ToRGB skip, 16-byte stores.
#include <type_traits>
#include "common.h"
#include <stdlib.h>

// A/B switches for timing experiments (tools/dev/build_variant.sh); the defaults are the shipped code.
#ifndef TDGP_AB_NO_SCALAR_WV
#define TDGP_AB_NO_SCALAR_WV 0
#endif
#ifndef TDGP_AB_NO_PACKED_EPI
#define TDGP_AB_NO_PACKED_EPI 0
#endif
#ifndef TDGP_AB_FIR_SERIAL
#define TDGP_AB_FIR_SERIAL 0
#endif
#if TDGP_AB_NO_SCALAR_WV
#define TDGP_WAVE_INDEX(tid) ((tid) >> 6)
#else
#define TDGP_WAVE_INDEX(tid) __builtin_amdgcn_readfirstlane((tid) >> 6)       // the wave index is uniform: keep it (and all tile arithmetic on it) scalar
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC3 = 4;    // packed channel chunk: weights live as [chunk][tap][Cout][4 channels]

struct TapTable {
    int ntaps, halo;    // halo = k/2
    int tap_w[25];      // tap index in the packed weight layout
    int tap_off_y[25];  // dy in [-halo, halo]
    int tap_off_x[25];
    int gridH, gridW;   // output pixels
};

// Everything the output stage needs (shared by the conv kernel's own epilogue and the split-K reduction kernel).
struct EpiParams {
    const float* __restrict__ dcoef; const float* __restrict__ noise; const float* __restrict__ bias; const float* __restrict__ skip;
    float* __restrict__ y;
    int64_t noise_bstride;
    float fir[16];      // flipped filter * gain for the fused skip upsample
    int B, Cout, Hout, Wout;    // dims of the output tensor
    int out_layout, out_feat;
    int act; float alpha, gain, clamp;
    int round_bf16;     // ToRGB of a reduced-precision block: conv output and bias_act output rounded to bf16 before the fp32 skip add
};

struct ConvParams {
    const float* x; const float* wp; const float* styles;
    float* partial;     // split-K: raw partial sums [ksplit][B,Cout,Hout,Wout]
    EpiParams e;
    int B, Cin, Cout, CoutP, Hin, Win, T;
    int tw_log2;
    int ksplit;
    TapTable ph;
};

__device__ __forceinline__ float act_apply(float v, int act, float alpha) {
    switch (act) {
    case 1: return v;
    case 2: return v > 0.f ? v : 0.f;
    case 3: return v > 0.f ? v : v * alpha;
    case 4: return tanhf(v);
    case 5: return 1.0f / (1.0f + expf(-v));
    case 6: return v > 0.f ? v : expm1f(v);
    case 7: return v > 0.f ? 1.0507009873554804934193349852946f * v
                           : (1.0507009873554804934193349852946f * 1.6732632423543772848170429916717f) * expm1f(v);
    case 8: return v > 20.f ? v : log1pf(expf(v));
    case 9: return (1.0f / (1.0f + expf(-v))) * v;
    }
    return v;
}

// x2 FIR upsample of the previous-resolution image at output pixel (oy,ox): upfirdn2d.upsample2d
// (upfirdn2d.py:313-348: up 2, pad [2,1,2,1], gain 4).  Of the 4x4 taps only a 2x2 subset hits non-zero (even) positions
// of the zero-stuffed image: rows r0 = (oy-1)>>1 and r0+1 with taps ky = (oy&1), (oy&1)+2 -- same for columns.
// Branch-free: 4 clamped loads issued together, invalid taps get weight 0.
// img points at the (b, channel) plane; `stride` = element stride between neighbouring pixels (1 for NCHW, feat for
// the channel-last plane layout).
struct SkipTaps { int i00, i01, i10, i11; float w00, w01, w10, w11; };
__device__ __forceinline__ SkipTaps skip_taps(int h2, int w2, int oy, int ox, const float* fir) {
    SkipTaps t;
    const int r0 = (oy - 1) >> 1, c0 = (ox - 1) >> 1;           // arithmetic shift: -1 for oy = 0
    const int ky0 = oy & 1, kx0 = ox & 1;
    const bool vr0 = r0 >= 0, vr1 = r0 + 1 < h2, vc0 = c0 >= 0, vc1 = c0 + 1 < w2;
    const int ra = vr0 ? r0 : 0, rb = vr1 ? r0 + 1 : h2 - 1, ca = vc0 ? c0 : 0, cb = vc1 ? c0 + 1 : w2 - 1;
    t.i00 = ra * w2 + ca; t.i01 = ra * w2 + cb; t.i10 = rb * w2 + ca; t.i11 = rb * w2 + cb;
    t.w00 = (vr0 && vc0) ? fir[ky0 * 4 + kx0] : 0.f;
    t.w01 = (vr0 && vc1) ? fir[ky0 * 4 + kx0 + 2] : 0.f;
    t.w10 = (vr1 && vc0) ? fir[(ky0 + 2) * 4 + kx0] : 0.f;
    t.w11 = (vr1 && vc1) ? fir[(ky0 + 2) * 4 + kx0 + 2] : 0.f;
    return t;
}
// The same taps with the FIR weights picked by SELECTS among 16 values held in scalar registers (a plain struct of named floats, each made opaque by the caller:
// `fir[ky0 * 4 + kx0 ..]` with per-lane parities is a vector-memory load per weight -- and so is any select the compiler can fold back into an indexed load --, which a
// pipeline of hand-counted waits cannot hold: rgb_output_skip_pipelined).
struct FirRegs { float f0, f1, f2, f3, f4, f5, f6, f7, f8, f9, f10, f11, f12, f13, f14, f15; };
__device__ __forceinline__ SkipTaps skip_taps_sel(int h2, int w2, int oy, int ox, const FirRegs& r) {
    SkipTaps t;
    const int r0 = (oy - 1) >> 1, c0 = (ox - 1) >> 1;
    const bool ky = oy & 1, kx = ox & 1;
    const bool vr0 = r0 >= 0, vr1 = r0 + 1 < h2, vc0 = c0 >= 0, vc1 = c0 + 1 < w2;
    const int ra = vr0 ? r0 : 0, rb = vr1 ? r0 + 1 : h2 - 1, ca = vc0 ? c0 : 0, cb = vc1 ? c0 + 1 : w2 - 1;
    t.i00 = ra * w2 + ca; t.i01 = ra * w2 + cb; t.i10 = rb * w2 + ca; t.i11 = rb * w2 + cb;
    auto pick = [&](float a, float b, float c, float d) { return ky ? (kx ? d : c) : (kx ? b : a); };      // fir[base], [base + 1], [base + 4], [base + 5]
    t.w00 = (vr0 && vc0) ? pick(r.f0, r.f1, r.f4, r.f5) : 0.f;
    t.w01 = (vr0 && vc1) ? pick(r.f2, r.f3, r.f6, r.f7) : 0.f;
    t.w10 = (vr1 && vc0) ? pick(r.f8, r.f9, r.f12, r.f13) : 0.f;
    t.w11 = (vr1 && vc1) ? pick(r.f10, r.f11, r.f14, r.f15) : 0.f;
    return t;
}
__device__ __forceinline__ float skip_eval(const float* __restrict__ img, const SkipTaps& t, int stride) {
    const float a = img[(int64_t)t.i00 * stride], b = img[(int64_t)t.i01 * stride], c = img[(int64_t)t.i10 * stride], d = img[(int64_t)t.i11 * stride];
    return fmaf_(t.w11, d, fmaf_(t.w10, c, fmaf_(t.w01, b, t.w00 * a)));
}

// ACT = 0: activation / clamp decided at run time (every bias_act form); ACT = 1 (linear) / 3 (lrelu): the two forms of the
// generator path with no clamp, specialised at compile time.  It matters more than its instruction count suggests: fp32 VALU
// and MFMA share the issue port, and a wave in its output stage gets one VALU instruction in per 64-cycle MFMA of its
// co-resident waves, so every instruction saved here is ~60 cycles of epilogue latency.
template <int ACT = 0>
__device__ __forceinline__ float finish_act(const EpiParams& e, float v) {
    if constexpr (ACT == 1) return v * e.gain;
    else if constexpr (ACT == 3) return (v > 0.f ? v : v * e.alpha) * e.gain;
    else if constexpr (ACT == 4) return __builtin_amdgcn_fmed3f(__builtin_fmaxf(v, v * e.alpha) * e.gain, -e.clamp, e.clamp);      // lrelu (0 <= alpha <= 1), clamp >= 0
    else {
        v = act_apply(v, e.act, e.alpha) * e.gain;
        if (e.clamp >= 0.f) v = v < -e.clamp ? -e.clamp : (v > e.clamp ? e.clamp : v);
        return v;
    }
}
// which specialisation a launch can use (3 = leaky ReLU written as max(v, alpha * v): needs 0 <= alpha <= 1)
__device__ __forceinline__ int epi_variant(const EpiParams& e) {
    if (e.clamp >= 0.f) return (e.act == 3 && e.alpha >= 0.f && e.alpha <= 1.f) ? 4 : 0;       // (4: only the launches that template on it use it)
    if (e.act == 1) return 1;
    return (e.act == 3 && e.alpha >= 0.f && e.alpha <= 1.f) ? 3 : 0;
}
// Four outputs of one channel at once: (v * d + nz) + bb -> activation -> * gain, the arithmetic of finish_act element by element.
// The specialised forms run as v_pk_{mul,add}_f32 (two elements per vector-ALU slot) and the leaky ReLU as max(v, alpha * v) --
// the same value for every input, sign of zero included, when 0 <= alpha <= 1 -- : 14 instead of 28 instructions per float4,
// and in this stage an instruction costs a ~60-cycle slot behind the other waves' MFMAs.
typedef float epi_f32x2 __attribute__((ext_vector_type(2)));
template <int ACT>
__device__ __forceinline__ float4 finish_act4(const EpiParams& e, float4 v, float d, float4 nz, float bb) {
    if constexpr ((ACT == 1 || ACT == 3) && !TDGP_AB_NO_PACKED_EPI) {
        const epi_f32x2 d2 = {d, d}, b2 = {bb, bb}, g2 = {e.gain, e.gain};
        epi_f32x2 lo = ((epi_f32x2){v.x, v.y} * d2 + (epi_f32x2){nz.x, nz.y}) + b2, hi = ((epi_f32x2){v.z, v.w} * d2 + (epi_f32x2){nz.z, nz.w}) + b2;
        if constexpr (ACT == 3) {
            const epi_f32x2 a2 = {e.alpha, e.alpha};
            const epi_f32x2 la = lo * a2, ha = hi * a2;
            lo = (epi_f32x2){__builtin_fmaxf(lo.x, la.x), __builtin_fmaxf(lo.y, la.y)};
            hi = (epi_f32x2){__builtin_fmaxf(hi.x, ha.x), __builtin_fmaxf(hi.y, ha.y)};
        }
        lo = lo * g2; hi = hi * g2;
        return make_float4(lo.x, lo.y, hi.x, hi.y);
    } else {
        return make_float4(finish_act<ACT>(e, (v.x * d + nz.x) + bb), finish_act<ACT>(e, (v.y * d + nz.y) + bb), finish_act<ACT>(e, (v.z * d + nz.z) + bb),
                           finish_act<ACT>(e, (v.w * d + nz.w) + bb));
    }
}

// demod * acc + noise + bias (+ FIR-upsampled skip) -> activation * gain -> clamp -> store (NCHW or channel-last planes);
// element-at-a-time form used by the split-K reduction.
__device__ __forceinline__ void epilogue_store(const EpiParams& e, float v, int b, int o, int oy, int ox) {
    if (e.dcoef) v = v * e.dcoef[b * e.Cout + o];
    if (e.noise) v = v + e.noise[b * e.noise_bstride + (int64_t)oy * e.Wout + ox];
    if (e.bias) v = v + e.bias[o];
    const int h2 = e.Hout / 2, w2 = e.Wout / 2;
    int64_t addr;
    float sk = 0.f;
    if (e.out_layout == 0) {
        if (e.skip) sk = skip_eval(e.skip + ((int64_t)b * e.Cout + o) * h2 * w2, skip_taps(h2, w2, oy, ox, e.fir), 1);
        addr = (((int64_t)b * e.Cout + o) * e.Hout + oy) * e.Wout + ox;
    } else {
        const int pl = o / e.out_feat, f = o % e.out_feat;
        const int64_t plane = (int64_t)b * (e.Cout / e.out_feat) + pl;
        if (e.skip) sk = skip_eval(e.skip + plane * h2 * w2 * e.out_feat + f, skip_taps(h2, w2, oy, ox, e.fir), e.out_feat);
        addr = ((plane * e.Hout + oy) * e.Wout + ox) * e.out_feat + f;
    }
    e.y[addr] = finish_act(e, v) + sk;          // img = upsample2d(img) + bias_act(conv): the skip joins AFTER activation / gain / clamp (:265-269)
}

// Output stage for one 32(channels) x 32(pixels) accumulator tile parked in a per-wave LDS tile ct[32][33].
// Called from ONE place inside a rolled loop over the wave's tiles (code size, registers).
//   layout 0 (NCHW) and raw split-K partials: lane = pixel (32 consecutive x -> 128-B row segments), 16 channels per lane;
//   layout 1 (channel-last planes):           lane = channel (32 consecutive features = one 128-B texel line), 16 pixels
//                                             per lane, pixel coordinates fetched from the owning lane by shuffle.
// Side inputs (demod, bias, noise, skip taps) are loaded in batches of 8 before the math so their latencies overlap.
// `side`: per-block LDS cache of the per-channel side inputs, [0..BM) = bias, then NSB slabs of BM demod coefficients for the
// samples b0 .. b0+NSB-1 a block tile can touch (side == nullptr: tile spans more samples -> read them from global memory).
struct SideCache { const float* lds; int b0, bm, m0; };
__device__ __forceinline__ float side_bias(const EpiParams& e, const SideCache& sc, int o, bool ok) {
    if (!e.bias || !ok) return 0.f;
    return sc.lds ? sc.lds[o - sc.m0] : e.bias[o];
}
__device__ __forceinline__ float side_demod(const EpiParams& e, const SideCache& sc, int b, int o, bool ok) {
    if (!e.dcoef || !ok) return 1.f;
    return sc.lds ? sc.lds[(1 + b - sc.b0) * sc.bm + (o - sc.m0)] : e.dcoef[b * e.Cout + o];
}

// One 32(channels) x 32(pixels) accumulator tile parked in the per-wave LDS tile `ct` (row stride CT_LD = 36 floats, so
// 4 consecutive entries are a 16-B aligned float4).  Stores are 16 B per lane wherever the geometry allows it -- 4-byte
// per-lane stores top out near 1.6 TB/s on this chip, which made the output stage 15-25 % of every conv launch:
//   NCHW / raw split-K partials, `vec`: ct is channel-major; lane = (channel row l>>3 of the pass, pixel quad l&7) stores
//         float4 = 4 consecutive x of one channel (a 32-pixel subtile is made of whole rows of >= 4 pixels);
//   NCHW scalar fallback: widths that are not a multiple of 4 and the op-level ToRGB with an NCHW skip;
//   channel-last planes (CL, ToRGB): ct is PIXEL-major; lane = (pixel l>>3 of the pass, channel quad l&7) stores float4 =
//         4 consecutive features of one texel; the x2-upsampled skip is 4 float4 taps per lane, all 4 passes in flight.
constexpr int CT_LD = 36;
template <bool CL, int ACT = 0>
__device__ __forceinline__ void epilogue_tile(const EpiParams& e, const SideCache& sc, const float* ct, int obase, int pb, int poy, int pox, int pok,
                                              float* part, bool vec) {
    const int l = lane_id(), l32 = l & 31, half = l >> 5;
    const int h2 = e.Hout / 2, w2 = e.Wout / 2;
    const bool raw = part != nullptr;
    if (!CL || raw || e.out_layout == 0) {
        float* dst = raw ? part : e.y;
        const int cstride = e.Hout * e.Wout;
        if (vec && (raw || !e.skip)) {
            const int g = l & 7, cr = l >> 3;
            const int b4 = __shfl(pb, 4 * g, 64), oy4 = __shfl(poy, 4 * g, 64), ox4 = __shfl(pox, 4 * g, 64), ok4 = __shfl(pok, 4 * g, 64);
            float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!raw && e.noise && ok4) nz = *(const float4*)(e.noise + b4 * e.noise_bstride + (int64_t)oy4 * e.Wout + ox4);
            const int pix = (b4 * e.Cout * e.Hout + oy4) * e.Wout + ox4;      // tensor < 2^31 elements (checked on the host)
#pragma unroll
            for (int pass = 0; pass < 4; pass++) {
                const int row = pass * 8 + cr, o = obase + row;
                if (ok4 && o < e.Cout) {
                    float4 v = *(const float4*)&ct[row * CT_LD + 4 * g];
                    if (!raw) v = finish_act4<ACT>(e, v, side_demod(e, sc, b4, o, true), nz, side_bias(e, sc, o, true));
                    *(float4*)(dst + pix + o * cstride) = v;
                }
            }
            return;
        }
        const int pix = (pb * e.Cout * e.Hout + poy) * e.Wout + pox;
        const float nz = (!raw && e.noise && pok) ? e.noise[pb * e.noise_bstride + (int64_t)poy * e.Wout + pox] : 0.f;
        SkipTaps st;
        if (!raw && e.skip) st = skip_taps(h2, w2, poy, pox, e.fir);
#pragma unroll 1
        for (int g = 0; g < 2; g++) {
            float sk[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int o = obase + (g * 8 + q) * 2 + half;
                sk[q] = (!raw && e.skip && pok && o < e.Cout) ? skip_eval(e.skip + ((int64_t)pb * e.Cout + o) * h2 * w2, st, 1) : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int sel = (g * 8 + q) * 2 + half;
                const int o = obase + sel;
                if (pok && o < e.Cout) {
                    float v = ct[sel * CT_LD + l32];
                    if (!raw) v = finish_act<ACT>(e, (v * side_demod(e, sc, pb, o, true) + nz) + side_bias(e, sc, o, true)) + sk[q];
                    dst[pix + o * cstride] = v;
                }
            }
        }
    } else {
        // channel-last planes; ct is pixel-major here (the kernel wrote it transposed)
        const int cg = l & 7, pr = l >> 3;
        const int o = obase + 4 * cg;
        const bool okc = o < e.Cout;                                            // Cout and out_feat are multiples of 4 (host check)
        const int pl = o / e.out_feat, f = o % e.out_feat;
        const int planes = e.Cout / e.out_feat;
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e.bias && okc) bias4 = make_float4(side_bias(e, sc, o, true), side_bias(e, sc, o + 1, true), side_bias(e, sc, o + 2, true), side_bias(e, sc, o + 3, true));
        // All 16 skip taps of the tile (4 passes x 4 texels, 16 B each) are put in flight BEFORE any of them is consumed, and
        // without per-pass branches: taken pass by pass, each pass waited out a full L2 round trip and the output stage was
        // two thirds of the ToRGB kernel (per-phase cycle counts, TDGP_RGB_ABL=16).  Invalid lanes read a clamped address.
        float4 sk[4];
        int addr[4], bq[4];
        float nzq[4];
        bool okp[4];
        SkipTaps tp[4];
        float4 ta[4], tb[4], tc[4], td[4];
        const bool has_skip = e.skip != nullptr;
#pragma unroll
        for (int pass = 0; pass < 4; pass++) {
            const int px = pass * 8 + pr;
            const int b = __shfl(pb, px, 64), oy = __shfl(poy, px, 64), ox = __shfl(pox, px, 64);
            okp[pass] = __shfl(pok, px, 64) && okc;
            bq[pass] = b;
            const int plane = b * planes + pl;
            addr[pass] = ((plane * e.Hout + oy) * e.Wout + ox) * e.out_feat + f;
            nzq[pass] = (e.noise && okp[pass]) ? e.noise[b * e.noise_bstride + (int64_t)oy * e.Wout + ox] : 0.f;
            if (has_skip) {
                tp[pass] = skip_taps(h2, w2, oy, ox, e.fir);
                const float* sp = e.skip + (int64_t)(okc ? plane : b * planes) * h2 * w2 * e.out_feat + (okc ? f : 0);
                ta[pass] = *(const float4*)(sp + (int64_t)tp[pass].i00 * e.out_feat); tb[pass] = *(const float4*)(sp + (int64_t)tp[pass].i01 * e.out_feat);
                tc[pass] = *(const float4*)(sp + (int64_t)tp[pass].i10 * e.out_feat); td[pass] = *(const float4*)(sp + (int64_t)tp[pass].i11 * e.out_feat);
            }
        }
#pragma unroll
        for (int pass = 0; pass < 4; pass++) {
            sk[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (has_skip) {
                const SkipTaps& t = tp[pass];
                const float4 a = ta[pass], bb = tb[pass], c = tc[pass], d = td[pass];
                sk[pass].x = fmaf_(t.w11, d.x, fmaf_(t.w10, c.x, fmaf_(t.w01, bb.x, t.w00 * a.x)));
                sk[pass].y = fmaf_(t.w11, d.y, fmaf_(t.w10, c.y, fmaf_(t.w01, bb.y, t.w00 * a.y)));
                sk[pass].z = fmaf_(t.w11, d.z, fmaf_(t.w10, c.z, fmaf_(t.w01, bb.z, t.w00 * a.z)));
                sk[pass].w = fmaf_(t.w11, d.w, fmaf_(t.w10, c.w, fmaf_(t.w01, bb.w, t.w00 * a.w)));
            }
        }
#pragma unroll
        for (int pass = 0; pass < 4; pass++) {
            if (okp[pass]) {
                const int px = pass * 8 + pr;
                float4 v = *(const float4*)&ct[px * CT_LD + 4 * cg];
                const float d0 = side_demod(e, sc, bq[pass], o, true), d1 = side_demod(e, sc, bq[pass], o + 1, true);
                const float d2 = side_demod(e, sc, bq[pass], o + 2, true), d3 = side_demod(e, sc, bq[pass], o + 3, true);
                v.x = finish_act<ACT>(e, (v.x * d0 + nzq[pass]) + bias4.x) + sk[pass].x;
                v.y = finish_act<ACT>(e, (v.y * d1 + nzq[pass]) + bias4.y) + sk[pass].y;
                v.z = finish_act<ACT>(e, (v.z * d2 + nzq[pass]) + bias4.z) + sk[pass].z;
                v.w = finish_act<ACT>(e, (v.w * d3 + nzq[pass]) + bias4.w) + sk[pass].w;
                *(float4*)(e.y + addr[pass]) = v;
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution.  Block = 256 threads = WM x WN waves, wave tile = (MTW*32) x (NTW*32).
// KCS = channels staged per K iteration (a multiple of the packed chunk of 4), MAXT = max taps (k*k).
//
// K loop: double-buffered LDS, ONE barrier per iteration.
//   issue global loads of chunk it+1 (packed weights as 16-B vectors, the halo'd activation patch as scalars)
//   -> MFMA over chunk it from LDS buffer `cur` (fragments double-buffered in registers: the ds_reads of step s+1
//      are issued before the MFMAs of step s, no scalar loads inside, so LDS latency hides behind the matrix pipe)
//   -> write the landed registers (x style) into LDS buffer `cur ^ 1` -> barrier.
// -------------------------------------------------------------------------------------------------
template <int MTW, int NTW, int WM, int WN, int KCS, int MAXT>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvParams p) {
    constexpr int BM = 32 * MTW * WM;
    constexpr int NT = NTW * WN;            // 32-pixel subtiles per block
    constexpr int BN = 32 * NT;
    constexpr int HALO = MAXT == 25 ? 2 : 1;      // largest halo this instantiation is launched with
    constexpr int XS_MAX = (BN / 4 + 2 * HALO) * (4 + 2 * HALO) > (BN / 32 + 2 * HALO) * (32 + 2 * HALO) ? (BN / 4 + 2 * HALO) * (4 + 2 * HALO) : (BN / 32 + 2 * HALO) * (32 + 2 * HALO);
    constexpr int AS_SZ = MAXT * KCS * BM, XS_SZ = KCS * XS_MAX, BUF_SZ = AS_SZ + XS_SZ + 64;   // +64: fragment prefetch past the last row
    constexpr int KH = KCS / 2;             // k-steps (of 2 channels) per tap; even
    static_assert(KH % 2 == 0, "KCS must be a multiple of 4");
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][As | Xs], then the tap-offset table
    int* toff_tab = (int*)(smem + 2 * BUF_SZ);                      // [MAXT + 2 (+ pad to 32)]
    float* side = smem + 2 * BUF_SZ + 32;                           // [1 + NSB][BM]: bias, demod coefficients of the tile's samples

    const int ks = blockIdx.z;
    const TapTable& ph = p.ph;
    const int TW = 1 << p.tw_log2, RPS = 32 >> p.tw_log2;
    const int TR = NT * RPS;                        // virtual rows per block tile
    const int R = ph.halo;
    const int PR = TR + 2 * R, PC = TW + 2 * R;
    const int PSZ = PR * PC;
    const int tilesX = (ph.gridW + TW - 1) / TW;
    const int VR = p.B * ph.gridH;
    const int tilesY = (VR + TR - 1) / TR;
    if ((int)blockIdx.x >= tilesX * tilesY) return;
    const int ty = blockIdx.x / tilesX, tx = blockIdx.x % tilesX;
    const int vr0 = ty * TR, n0 = tx * TW;
    const int m0 = blockIdx.y * BM;

    const int tid = threadIdx.x, l = tid & 63, wv = TDGP_WAVE_INDEX(tid);    // wave index: scalar, so is everything derived from it
    const int l32 = l & 31, half = l >> 5;
    const int wm = wv / WN, wn = wv % WN;

    if (tid < MAXT + 2) toff_tab[tid] = tid < ph.ntaps ? (ph.tap_off_y[tid] + R) * PC + (ph.tap_off_x[tid] + R) : 0;
    // per-channel side inputs of the epilogue -> LDS now, so the output stage has no dependent global loads
    constexpr int NSB = 4;
    const int sb0 = vr0 / ph.gridH, sb1 = min(p.B - 1, (vr0 + TR - 1) / ph.gridH);
    const bool side_ok = (sb1 - sb0 + 1) <= NSB && p.ksplit == 1;
    if (side_ok) {
        for (int i = tid; i < (2 + sb1 - sb0) * BM; i += 256) {
            const int slab = i / BM, o = m0 + i % BM;
            float v = slab == 0 ? 0.f : 1.f;
            if (o < p.e.Cout) {
                if (slab == 0) { if (p.e.bias) v = p.e.bias[o]; }
                else if (p.e.dcoef) v = p.e.dcoef[(sb0 + slab - 1) * p.e.Cout + o];
            }
            side[i] = v;
        }
    }

    // ---- per-thread patch positions (fixed across the K loop) -------------------------------------
    constexpr int NPOS = (XS_MAX + 255) / 256;
    int pos_off[NPOS];          // element offset of (b', iy, ix) in x for channel 0, or -1
    int pos_sb[NPOS];           // b' * Cin
#pragma unroll
    for (int k = 0; k < NPOS; k++) {
        const int pos = tid + k * 256;
        pos_off[k] = -1; pos_sb[k] = 0;
        if (pos < PSZ) {
            const int pr = pos / PC, pc = pos % PC;
            const int vi = vr0 - R + pr, ix = n0 - R + pc;
            if (vi >= 0 && ix >= 0 && ix < p.Win) {
                const int bb = vi / ph.gridH, iy = vi % ph.gridH;
                if (bb < p.B && iy < p.Hin) {
                    pos_off[k] = ((bb * p.Cin) * p.Hin + iy) * p.Win + ix;
                    pos_sb[k] = bb * p.Cin;
                }
            }
        }
    }
    const int chw = p.Hin * p.Win;

    // ---- per-lane pixel info for its NTW subtiles ------------------------------------------------
    int px_base[NTW];           // LDS offset of the pixel's (-halo,-halo) tap inside the patch
    int px_mask[NTW];           // bit t: tap t reads a row of the same sample
    int px_b[NTW], px_m[NTW], px_n[NTW];
    bool px_ok[NTW];
#pragma unroll
    for (int n = 0; n < NTW; n++) {
        const int sn = wn * NTW + n;
        const int lr = sn * RPS + (l32 >> p.tw_log2), lc = l32 & (TW - 1);
        const int vr = vr0 + lr, nn = n0 + lc;
        const int bb = vr / ph.gridH, m = vr % ph.gridH;
        px_b[n] = bb; px_m[n] = m; px_n[n] = nn;
        px_ok[n] = vr < VR && nn < ph.gridW;
        px_base[n] = lr * PC + lc + half * PSZ;     // + the lane's channel of the pair
        int mask = 0;
        for (int t = 0; t < ph.ntaps; t++) {
            const int iy = m + ph.tap_off_y[t];
            if (iy >= 0 && iy < p.Hin) mask |= 1 << t;
        }
        px_mask[n] = mask;
    }

    f32x16 acc[MTW][NTW];
#pragma unroll
    for (int m = 0; m < MTW; m++)
#pragma unroll
        for (int n = 0; n < NTW; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;

    const int G = KCS / KC3;                        // packed chunks per staged iteration
    const int nchunks_packed = (p.Cin + KC3 - 1) / KC3;
    const int niter = (nchunks_packed + G - 1) / G;
    const int arows = ph.ntaps * KCS;
    const int it_per = (niter + p.ksplit - 1) / p.ksplit;          // split-K: this block reduces iterations [it0, it1)
    const int it0 = ks * it_per, it1 = min(niter, it0 + it_per);

    // A tile of one iteration: ntaps x G packed chunks x BM columns, one float4 (the 4 channels of a column) per slot
    constexpr int GC = KCS / KC3;
    constexpr int NA = (MAXT * GC * BM + 255) / 256;
    float4 a_reg[NA];
    float x_reg[NPOS][KCS], s_reg[NPOS][KCS];
    // Everything about a thread's staging slots that does not depend on the K iteration is computed once: global offsets
    // advance by a uniform stride per iteration (the packed weights are zero-padded to whole iterations).
    int a_goff[NA], a_loff[NA];                       // global float offset at iteration 0 (-1: slot unused), LDS float offset
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int e = tid + i * 256;
        a_goff[i] = -1; a_loff[i] = 0;
        if (e < ph.ntaps * GC * BM) {
            const int col = e % BM, tg = e / BM, g = tg % GC, t = tg / GC;
            a_loff[i] = (t * KCS + g * KC3) * BM + col;
            if (m0 + col < p.CoutP) a_goff[i] = ((g * p.T + ph.tap_w[t]) * p.CoutP + m0 + col) * 4;
        }
    }
    const int a_gstride = GC * p.T * p.CoutP * 4;

    auto load_stage = [&](int it) {
        const float* wp_it = p.wp + (int64_t)it * a_gstride;
#pragma unroll
        for (int i = 0; i < NA; i++) a_reg[i] = a_goff[i] >= 0 ? *(const float4*)(wp_it + a_goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
        const int c0 = it * KCS;
        const float* x_it = p.x + (int64_t)c0 * chw;
#pragma unroll
        for (int k = 0; k < NPOS; k++) {
            const int off = pos_off[k];
#pragma unroll
            for (int ci = 0; ci < KCS; ci++) {
                const bool ok = off >= 0 && c0 + ci < p.Cin;
                x_reg[k][ci] = ok ? x_it[off + ci * chw] : 0.f;
                s_reg[k][ci] = (ok && p.styles) ? p.styles[pos_sb[k] + c0 + ci] : 1.f;
            }
        }
    };
    auto store_stage = [&](float* As, float* Xs) {
#pragma unroll
        for (int i = 0; i < NA; i++)
            if (tid + i * 256 < ph.ntaps * GC * BM) {           // [col][4 ch] -> k-major rows
                float* d = As + a_loff[i];
                d[0] = a_reg[i].x; d[BM] = a_reg[i].y; d[2 * BM] = a_reg[i].z; d[3 * BM] = a_reg[i].w;
            }
#pragma unroll
        for (int k = 0; k < NPOS; k++) {
            const int pos = tid + k * 256;
            if (pos < PSZ) {
#pragma unroll
                for (int ci = 0; ci < KCS; ci++) Xs[ci * PSZ + pos] = x_reg[k][ci] * s_reg[k][ci];     // modulation rides on the staging
            }
        }
    };

    if (it0 < it1) {
        load_stage(it0);
        store_stage(smem, smem + AS_SZ);
    }
    __syncthreads();
    int cur = 0;
    for (int it = it0; it < it1; it++) {
        const bool more = it + 1 < it1;
        if (more) load_stage(it + 1);
        const float* As = smem + cur * BUF_SZ + (wm * MTW) * 32 + l32 + half * BM;
        const float* Xs = smem + cur * BUF_SZ + AS_SZ;
        // ---- MFMA over (tap, channel pair); fragments of step s+1 are in flight while step s multiplies ------
        float fa[2][MTW], fb[2][NTW];
        bool fk[2][NTW];                                  // tap-row validity of the staged B values (applied at use, not at load)
        int toff_cur = toff_tab[0], toff_nxt = toff_tab[1];
        auto load_frag = [&](int buf, int row2, int kk, int toff, int t) {
#pragma unroll
            for (int m = 0; m < MTW; m++) fa[buf][m] = As[row2 * 2 * BM + m * 32];
#pragma unroll
            for (int n = 0; n < NTW; n++) {
                fb[buf][n] = Xs[2 * kk * PSZ + px_base[n] + toff];
                fk[buf][n] = (px_mask[n] >> t) & 1;
            }
        };
        const int ntaps = ph.ntaps;
        load_frag(0, 0, 0, toff_cur, 0);
        for (int t = 0; t < ntaps; t++) {
#pragma unroll
            for (int kk = 0; kk < KH; kk++) {
                const int cb = kk & 1, nb = cb ^ 1;
                if (kk + 1 < KH) load_frag(nb, t * KH + kk + 1, kk + 1, toff_cur, t);
                else load_frag(nb, (t + 1) * KH, 0, toff_nxt, t + 1);          // first step of the next tap (harmless past the end)
                __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ds_reads ABOVE this step's MFMAs (hipcc would sink them to their use)
                float bq[NTW];
#pragma unroll
                for (int n = 0; n < NTW; n++) bq[n] = fk[cb][n] ? fb[cb][n] : 0.f;
#pragma unroll
                for (int m = 0; m < MTW; m++)
#pragma unroll
                    for (int n = 0; n < NTW; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cb][m], bq[n], acc[m][n], 0, 0, 0);
            }
            toff_cur = toff_nxt;
            toff_nxt = toff_tab[t + 2];
        }
        if (more) store_stage(smem + (cur ^ 1) * BUF_SZ, smem + (cur ^ 1) * BUF_SZ + AS_SZ);
        __syncthreads();             // cur fully read, cur^1 fully written
        cur ^= 1;
    }

    // ---- epilogue: accumulators -> per-wave 32x33 LDS tile -> epilogue_tile() ---------------------------------------
    // The loop over the wave's tiles stays rolled; the accumulator tile is selected with a compile-time-indexed if-chain
    // so the accumulators never need dynamic register indexing.
    float* ct = smem + wv * (32 * CT_LD);
    const bool ct_pixel_major = (MAXT == 1) && p.ksplit == 1 && p.e.out_layout == 1;       // channel-last epilogue reads float4 of channels
    const bool vec = (p.e.Wout & 3) == 0;                                                 // 4 consecutive lanes = 4 consecutive x of one row
    const EpiParams& e = p.e;
    SideCache scache;
    scache.lds = side_ok ? side : nullptr; scache.b0 = sb0; scache.bm = BM; scache.m0 = m0;
    float* part = (p.ksplit > 1) ? p.partial + (int64_t)ks * e.B * e.Cout * e.Hout * e.Wout : nullptr;
#pragma unroll 1
    for (int tile = 0; tile < MTW * NTW; tile++) {
        int pb = 0, pm = 0, pn = 0, pk = 0;
#pragma unroll
        for (int k = 0; k < MTW * NTW; k++) {
            if (tile == k) {
                constexpr int dummy = 0; (void)dummy;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    ct[ct_pixel_major ? l32 * CT_LD + row : row * CT_LD + l32] = acc[k % MTW][k / MTW][r];
                }
                pb = px_b[k / MTW]; pm = px_m[k / MTW]; pn = px_n[k / MTW]; pk = px_ok[k / MTW] ? 1 : 0;
            }
        }
        const int m = tile % MTW;
        const int poy = pm, pox = pn;
        const int pok = (pk && poy < e.Hout && pox < e.Wout) ? 1 : 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        epilogue_tile<MAXT == 1>(e, scache, ct, m0 + (wm * MTW + m) * 32, pb, poy, pox, pok, part, vec);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// -------------------------------------------------------------------------------------------------
// Fast paths of the 3x3 layers.  Measured on this chip (tools/dev/ubench_overlap.hip): fp32 MFMA and fp32 VALU do NOT
// overlap -- an MFMA-only loop runs at 149-155 TFLOP/s, a v_pk_fma-only loop of the same cycle count takes the same time, and
// the two together take the SUM (or more), whether interleaved in one wave or split across the waves of a SIMD.  Every VALU
// instruction inside the K loop therefore costs matrix time (4 cycles of 64 for a 32x32x2 step), and the generic kernel above
// spends ~35 VALU per 8 MFMA on tap masks, LDS address arithmetic and fragment moves.  The two kernels below are built so the
// loop body is MFMA + ds_read_b64 with immediate offsets and nothing else:
//   * weights packed [chunk][tap][Cout][4 channels] and staged as-is: a lane's A operand for BOTH k-steps of a 4-channel chunk
//     is one aligned 8-byte LDS read (k-step kk multiplies channels {kk, kk+2}: lane half h holds channels 2h, 2h+1);
//   * activations staged [position][4 channels] (one 16-B LDS write per position), read the same way;
//   * taps and k-steps fully unrolled: every LDS address is base register + compile-time offset;
//   * no masks: padding is materialised as zeros in the staged patch (zero rows between samples / the linearised grid).
// -------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Staging loads go through buffer instructions: address = descriptor base (SGPRs) + per-lane byte offset (a VGPR that never
// changes) + per-iteration byte offset (an SGPR, advanced on the scalar unit) -- no vector ALU work per K iteration at all,
// and an offset at or beyond the descriptor's size returns 0, which is how halo positions, padded out-channel columns and
// unused slots are masked (offset kOOB) without branches or selects.
constexpr uint32_t kOOB = 0xFFFFFFF0u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// ---- KS x KS (3x3 / 5x5), stride 1, W % 32 == 0 -----------------------------------------------------------------------------
// Pixel tile = NT "virtual rows" x 32 columns, virtual row vi = b*(H+R) + m (R = KS/2) where rows m >= H of every sample are
// ZERO rows: they are the bottom padding of sample b and the top padding of sample b+1, so a tile may straddle samples without
// any per-lane tap mask (cost: R/(H+R) of the rows compute nothing useful).  Left/right padding = the zero halo columns.
struct Conv3Params {
    const float* x; const float* wp; const float* styles; float* partial;
    EpiParams e;
    int B, Cin, Cout, CoutP, H, W, ksplit;
    uint32_t x_bytes, wp_bytes, st_bytes;       // sizes for the buffer descriptors
};

#ifndef TDGP_C3_ABL
#define TDGP_C3_ABL 0      // 16: per-phase cycle counts of one wave, printed (timing experiments only)
#endif
template <int KS, int MTW, int NTW, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv3_mfma_kernel(Conv3Params p) {
    constexpr int R = KS / 2, T = KS * KS;
    constexpr int BM = 32 * MTW * WM, NT = NTW * WN, PR = NT + 2 * R, PC = 32 + 2 * R, PSZ = PR * PC;
    constexpr int AS_SZ = T * BM * 4, XS_SZ = PSZ * 4, BUF_SZ = AS_SZ + XS_SZ;          // floats
    extern __shared__ __attribute__((aligned(16))) float smem[];
#if TDGP_C3_ABL & 16
    long long tq[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define TQ(i) { const long long tn_ = __builtin_readcyclecounter(); tq[i] += tn_ - tprev; tprev = tn_; }
#else
#define TQ(i)
#endif
    float* side = smem + 2 * BUF_SZ;                                // [1 + NSB][BM]: bias, demod coefficients of the tile's samples

    const int H1 = p.H + R;
    const int tilesX = p.W >> 5, VR = p.B * H1;
    const int ty = blockIdx.x / tilesX, tx = blockIdx.x % tilesX;
    const int vr0 = ty * NT, n0 = tx * 32, m0 = blockIdx.y * BM, ks = blockIdx.z;
    const int tid = threadIdx.x, l = tid & 63, wv = TDGP_WAVE_INDEX(tid), l32 = l & 31, half = l >> 5;     // wave index: scalar
    const int wm = wv / WN, wn = wv % WN;

    constexpr int NSB = 4;
    const int sb0 = vr0 / H1, sb1 = min(p.B - 1, (vr0 + NT - 1) / H1);
    const bool side_ok = (sb1 - sb0 + 1) <= NSB && p.ksplit == 1;
    if (side_ok) {
        for (int i = tid; i < (2 + sb1 - sb0) * BM; i += 256) {
            const int slab = i / BM, o = m0 + i % BM;
            float v = slab == 0 ? 0.f : 1.f;
            if (o < p.e.Cout) {
                if (slab == 0) { if (p.e.bias) v = p.e.bias[o]; }
                else if (p.e.dcoef) v = p.e.dcoef[(sb0 + slab - 1) * p.e.Cout + o];
            }
            side[i] = v;
        }
    }

    // ---- patch slots of this thread: position (row, col) of the halo'd patch, 4 channels per K iteration; byte offsets into
    // x / styles for the buffer loads, kOOB (-> 0) for halo positions outside the image, zero rows and unused slots
    constexpr int NPOS = (PSZ + 255) / 256;
    uint32_t pos_xo[NPOS], pos_so[NPOS];
#pragma unroll
    for (int k = 0; k < NPOS; k++) {
        const int pos = tid + k * 256;
        pos_xo[k] = kOOB; pos_so[k] = kOOB;
        if (pos < PSZ) {
            const int pr = pos / PC, pc = pos % PC;
            const int vi = vr0 - R + pr, ix = n0 - R + pc;
            if (vi >= 0 && ix >= 0 && ix < p.W) {
                const int b = vi / H1, m = vi - b * H1;
                if (b < p.B && m < p.H) { pos_xo[k] = (uint32_t)(((b * p.Cin) * p.H + m) * p.W + ix) * 4u; pos_so[k] = (uint32_t)(b * p.Cin) * 4u; }
            }
        }
    }
    const uint32_t chw4 = (uint32_t)(p.H * p.W) * 4u;
    const bool cin4 = (p.Cin & 3) == 0;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes), rw = make_rsrc(p.wp, p.wp_bytes), rs = make_rsrc(p.styles ? p.styles : p.x, p.styles ? p.st_bytes : 0);

    f32x16 acc[MTW][NTW];
#pragma unroll
    for (int m = 0; m < MTW; m++)
#pragma unroll
        for (int n = 0; n < NTW; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;

    const int niter = (p.Cin + 3) >> 2;
    const int it_per = (niter + p.ksplit - 1) / p.ksplit;
    const int it0 = ks * it_per, it1 = min(niter, it0 + it_per);

    constexpr int NA = (T * BM + 255) / 256;
    // Staging registers: weights one K iteration ahead (they come from L2), activations TWO iterations ahead in alternating
    // register sets -- at 64 channels the activation tensor streams from HBM, and with one iteration (4600 cycles) of flight
    // time the LDS write waited ~8000 cycles per iteration for its loads (per-phase cycle counts, TDGP_C3_ABL=16).
    float4 a_reg[NA], x_reg[2][NPOS], s_reg[2][NPOS];
    uint32_t a_vo[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int e = tid + i * 256;
        a_vo[i] = kOOB;
        if (e < T * BM) {
            const int t = e / BM, col = e % BM;
            if (m0 + col < p.CoutP) a_vo[i] = (uint32_t)(t * p.CoutP + m0 + col) * 16u;
        }
    }
    const uint32_t a_gstride4 = (uint32_t)(T * p.CoutP) * 16u;

    // (iteration / channel indices are clamped into the tensors: the scalar offset of a buffer load must stay inside the
    //  descriptor for its range check to be meaningful; the clamped loads of the pipeline's overrun are never stored, a
    //  duplicated last channel meets zero-padded weights)
    auto load_a = [&](int it) {
        const uint32_t a_so = (uint32_t)min(it, niter - 1) * a_gstride4;
#pragma unroll
        for (int i = 0; i < NA; i++) a_reg[i] = buf_load4(rw, a_vo[i], a_so);
    };
    auto load_x = [&](int it, int set) {
        const uint32_t c0 = (uint32_t)min(it, niter - 1) * 4u, cl = (uint32_t)(p.Cin - 1);
#pragma unroll
        for (int k = 0; k < NPOS; k++) {
            x_reg[set][k] = make_float4(buf_load1(rx, pos_xo[k], min(c0, cl) * chw4), buf_load1(rx, pos_xo[k], min(c0 + 1, cl) * chw4),
                                        buf_load1(rx, pos_xo[k], min(c0 + 2, cl) * chw4), buf_load1(rx, pos_xo[k], min(c0 + 3, cl) * chw4));
            if (p.styles) {
                if (cin4) s_reg[set][k] = buf_load4(rs, pos_so[k], c0 * 4u);
                else s_reg[set][k] = make_float4(buf_load1(rs, pos_so[k], min(c0, cl) * 4u), buf_load1(rs, pos_so[k], min(c0 + 1, cl) * 4u),
                                                 buf_load1(rs, pos_so[k], min(c0 + 2, cl) * 4u), buf_load1(rs, pos_so[k], min(c0 + 3, cl) * 4u));
            }
        }
    };
    auto store_stage = [&](int set, float* As, float* Xs) {
#pragma unroll
        for (int i = 0; i < NA; i++)
            if (tid + i * 256 < T * BM) *(float4*)(As + (tid + i * 256) * 4) = a_reg[i];
#pragma unroll
        for (int k = 0; k < NPOS; k++)
            if (tid + k * 256 < PSZ) {       // modulation rides on the staging
                if (p.styles) *(float4*)(Xs + (tid + k * 256) * 4) = make_float4(x_reg[set][k].x * s_reg[set][k].x, x_reg[set][k].y * s_reg[set][k].y,
                                                                                x_reg[set][k].z * s_reg[set][k].z, x_reg[set][k].w * s_reg[set][k].w);
                else *(float4*)(Xs + (tid + k * 256) * 4) = x_reg[set][k];
            }
    };
    const int a_lane = ((wm * MTW) * 32 + l32) * 4 + 2 * half;                   // + (t*BM + m*32)*4
    const int b_lane = (((wn * NTW) + R) * PC + l32 + R) * 4 + 2 * half;        // centre tap of subtile 0; + (n*PC + dy*PC + dx)*4
    auto mma = [&](int cur) {
        const float* As = smem + cur * BUF_SZ + a_lane;
        const float* Xs = smem + cur * BUF_SZ + AS_SZ + b_lane;
        f32x2 fa[2][MTW], fb[2][NTW];
        auto load_frag = [&](int buf, int t) {
            const int dy = t / KS - R, dx = t % KS - R;
#pragma unroll
            for (int m = 0; m < MTW; m++) fa[buf][m] = *(const f32x2*)(As + (t * BM + m * 32) * 4);
#pragma unroll
            for (int n = 0; n < NTW; n++) fb[buf][n] = *(const f32x2*)(Xs + ((n + dy) * PC + dx) * 4);
        };
        load_frag(0, 0);
#pragma unroll
        for (int t = 0; t < T; t++) {
            const int cb = t & 1;
            if (t + 1 < T) load_frag(cb ^ 1, t + 1);
            __builtin_amdgcn_sched_barrier(0);          // keep the prefetch ds_reads above this tap's MFMAs
#pragma unroll
            for (int kk = 0; kk < 2; kk++)
#pragma unroll
                for (int m = 0; m < MTW; m++)
#pragma unroll
                    for (int n = 0; n < NTW; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cb][m][kk], fb[cb][n][kk], acc[m][n], 0, 0, 0);
        }
    };

    float* const L0 = smem;
    float* const L1 = smem + BUF_SZ;
    // Loads are issued unconditionally (past the last iteration they fall outside the descriptors or are never stored) and
    // the loop body has no exit in the middle: only then does the compiler wait with vmcnt(<the younger loads>) in front of the
    // LDS write; with loads under a condition it falls back to vmcnt(0), which silently removes the second stage.
    load_a(it0); load_x(it0, 0);
    load_x(it0 + 1, 1);
    if (it0 < it1) store_stage(0, L0, L0 + AS_SZ);
    __syncthreads();
    TQ(0)
    int it = it0;
    for (; it + 1 < it1; it += 2) {
        // even half: L0 holds iteration `it`; x set 1 = iteration it+1 (in flight since the previous half)
        load_a(it + 1);
        load_x(it + 2, 0);
        TQ(1)
        mma(0);
        TQ(2)
        store_stage(1, L1, L1 + AS_SZ);
        TQ(3)
        __syncthreads();
        TQ(4)
        // odd half: L1 holds iteration it+1; x set 0 = iteration it+2
        load_a(it + 2);
        load_x(it + 3, 1);
        TQ(1)
        mma(1);
        TQ(2)
        if (it + 2 < it1) store_stage(0, L0, L0 + AS_SZ);
        TQ(3)
        __syncthreads();
        TQ(4)
    }
    if (it < it1) {                     // odd iteration count: the last one sits in L0 -- and the epilogue's per-wave tiles overlay the
        mma(0);                         // stage buffers: every wave must be done reading them (without this barrier a fast wave's
        __syncthreads();                // epilogue corrupted a slow wave's last fragments: rare, wave-tile-sized errors)
    }
    TQ(2)

    // ---- epilogue: accumulators -> per-wave LDS tile -> epilogue_tile() (16-B stores, fused demod/noise/bias/act) ------------
    float* ct = smem + wv * (32 * CT_LD);
    const EpiParams& e = p.e;
    SideCache scache;
    scache.lds = side_ok ? side : nullptr; scache.b0 = sb0; scache.bm = BM; scache.m0 = m0;
    float* part = (p.ksplit > 1) ? p.partial + (int64_t)ks * e.B * e.Cout * e.Hout * e.Wout : nullptr;
    const int evar = epi_variant(e);
#pragma unroll 1
    for (int tile = 0; tile < MTW * NTW; tile++) {
#pragma unroll
        for (int k = 0; k < MTW * NTW; k++) {
            if (tile == k) {
                constexpr int dummy = 0; (void)dummy;
#pragma unroll
                for (int r = 0; r < 16; r++) ct[((r & 3) + 8 * (r >> 2) + 4 * half) * CT_LD + l32] = acc[k % MTW][k / MTW][r];
            }
        }
        const int m = tile % MTW, n = tile / MTW;
        const int vi = vr0 + wn * NTW + n;
        const int pb = vi / H1, poy = vi - pb * H1;
        const int pok = (vi < VR && poy < p.H) ? 1 : 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (evar == 3) epilogue_tile<false, 3>(e, scache, ct, m0 + (wm * MTW + m) * 32, pb, poy, n0 + l32, pok, part, true);
        else if (evar == 1) epilogue_tile<false, 1>(e, scache, ct, m0 + (wm * MTW + m) * 32, pb, poy, n0 + l32, pok, part, true);
        else epilogue_tile<false, 0>(e, scache, ct, m0 + (wm * MTW + m) * 32, pb, poy, n0 + l32, pok, part, true);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    TQ(5)
#if TDGP_C3_ABL & 16
    if (tid == 0 && (blockIdx.x == 3 || blockIdx.x == 900) && blockIdx.y == 0 && blockIdx.z == 0)
        printf("conv3 blk %d iters %d: prologue %lld load-issue %lld mma %lld store %lld barrier %lld epilogue %lld\n", (int)blockIdx.x, it1 - it0, tq[0], tq[1], tq[2], tq[3], tq[4], tq[5]);
#endif
#undef TQ
}

// ---- x2 layers: stride-2 transposed 3x3 convolution (conv2d_resample.py:108-125, unflipped weights) --------------------
//     Z[2i+a, 2j+e] += w[a,e] * x[i,j]
// as ONE implicit GEMM over a LINEARISED input grid.  Grid point v' = m*G1 + n, m in [0,H], n in [0,G1), G1 = W+2, with x := 0
// on columns >= W and row H (one zero column would do; two make the Z row pitch a multiple of 4 floats for the FIR's 16-B loads).  The four output parities (py,px) of grid point (m,n) are
//     Z[2m+py, 2n+px] = sum_{a = py (mod 2), e = px (mod 2)} w[a,e] * x[m - (a==2), n - (e==2)]
// and in linear space the four distinct taps are the offsets {0, -1, -G1, -G1-1}: the zero columns double as the
// left AND right padding (n-1 at n = 0 wraps onto the last column of the previous row), the zero row separates samples.  So
//   * a block tile is BN CONSECUTIVE grid points -- no 2-D tile edges, no "+1" column of wasted tiles (the old four-phase
//     launch lost 11-50 % of its lanes to the (W+1)-wide phase grids), no tap masks in the MFMA loop;
//   * the activation patch is two runs of BN+1 positions (the dy = -1 and dy = 0 rows);
//   * all 9 taps of a channel pair are multiplied in one pass: 9 A fragments x 4 B fragments feed the 4 parity
//     accumulators of each 32x32 tile (taps per parity 4/2/2/1), so x is staged once instead of once per phase;
//   * the epilogue interleaves the px = 0/1 accumulators through LDS and writes Z as dense 16-B-per-lane rows.
// Z layout (workspace): [ksplit][B][Cout][py][2*GS], entry 2*v' + px; GS = (H+1)*G1 rounded up to 32 so a 32-point
// subtile never straddles samples and every row segment stays 16-B aligned.  Seen as an image, parity plane py holds Z rows
// 2m+py with row pitch 2*G1; the pad columns 2W+1.. and the pad row 2H+1 come out as exact zeros.
struct UpParams {
    const float* x; const float* wp; const float* styles; float* z;
    int B, Cin, Cout, CoutP, H, W, G1, GS, ksplit;
    int64_t zslice;
    uint32_t x_bytes, wp_bytes, st_bytes;       // sizes for the buffer descriptors
};

#ifndef TDGP_UP_ABL
#define TDGP_UP_ABL 0      // timing experiments (tools/dev/build_variant.sh): 1 no Z stores, 2 no global staging loads, 4 no LDS fragment reads, 8 no barriers
#endif
constexpr int UP_CT_W = 68;     // epilogue LDS tile: 32 channels x 64 floats (+4 pad)

template <int MTW, int NTW, int WM, int WN, bool DEEP>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN > 4 ? 1 : 2)) void upconv_mfma_kernel(UpParams p) {
    constexpr int NTH = 64 * WM * WN;
    constexpr int BM = 32 * MTW * WM, BN = 32 * NTW * WN;
    constexpr int XP = BN + 2;                            // positions per run (BN + 1 used)
    constexpr int AS_SZ = 9 * BM * 4, XS_SZ = 2 * XP * 4, BUF_SZ = AS_SZ + XS_SZ;          // floats
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int v0 = blockIdx.x * BN, m0 = blockIdx.y * BM, ks = blockIdx.z;
    const int tid = threadIdx.x, l = tid & 63, wv = TDGP_WAVE_INDEX(tid), l32 = l & 31, half = l >> 5;     // wave index: scalar
    const int wm = wv / WN, wn = wv % WN;

    // ---- activation patch slots of this thread (fixed across the K loop): run 0 = grid points v-(W+1) (dy = -1), run 1 = dy = 0
    constexpr int NE = 2 * (BN + 1);
    constexpr int NPOS = (NE + NTH - 1) / NTH;
    uint32_t pos_xo[NPOS], pos_so[NPOS];        // byte offsets into x / styles (kOOB -> the load returns 0)
    int pos_lds[NPOS];
#pragma unroll
    for (int k = 0; k < NPOS; k++) {
        const int e = tid + k * NTH;
        pos_xo[k] = kOOB; pos_so[k] = kOOB; pos_lds[k] = -1;
        if (e < NE) {
            const int seg = e / (BN + 1), idx = e % (BN + 1);
            const int g = v0 + idx - 1 - (seg == 0 ? p.G1 : 0);
            pos_lds[k] = (seg * XP + idx) * 4;
            if (g >= 0) {
                const int b = g / p.GS, vp = g - b * p.GS;
                const int m = vp / p.G1, n = vp - m * p.G1;
                if (b < p.B && m < p.H && n < p.W) {
                    pos_xo[k] = (uint32_t)(((b * p.Cin) * p.H + m) * p.W + n) * 4u;
                    pos_so[k] = (uint32_t)(b * p.Cin) * 4u;
                }
            }
        }
    }
    const uint32_t chw4 = (uint32_t)(p.H * p.W) * 4u;
    const bool cin4 = (p.Cin & 3) == 0;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes), rw = make_rsrc(p.wp, p.wp_bytes), rs = make_rsrc(p.styles ? p.styles : p.x, p.styles ? p.st_bytes : 0);

    f32x16 acc[4][MTW][NTW];            // [py*2+px]
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int m = 0; m < MTW; m++)
#pragma unroll
            for (int n = 0; n < NTW; n++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[q][m][n][r] = 0.f;

    const int niter = (p.Cin + 3) >> 2;
    const int it_per = (niter + p.ksplit - 1) / p.ksplit;
    const int it0 = ks * it_per, it1 = min(niter, it0 + it_per);

    // Staging pipeline, two stages deep: the global loads of K iteration i+2 are issued while iteration i multiplies, into the
    // second of two register sets, so a load has two full iterations (~2 x 2300 cycles) to land before its LDS write.  With a
    // single set the loads of iteration i+1 had one iteration, and the vmcnt wait in front of the LDS write cost a third of the
    // kernel (ablation without the loads: 125-130 TFLOP/s against 82-88).
    constexpr int NA = (9 * BM + NTH - 1) / NTH;
    constexpr int NSET = DEEP ? 2 : 1;         // DEEP: two register sets (needs the VGPRs: the 64x128 tile with its two patch slots per thread spills)
    float4 a_reg[NSET][NA], x_reg[NSET][NPOS], s_reg[NSET][NPOS];
    uint32_t a_vo[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int e = tid + i * NTH;
        a_vo[i] = kOOB;
        if (e < 9 * BM) {
            const int t = e / BM, col = e % BM;
            if (m0 + col < p.CoutP) a_vo[i] = (uint32_t)(t * p.CoutP + m0 + col) * 16u;
        }
    }
    const uint32_t a_gstride4 = (uint32_t)(9 * p.CoutP) * 16u;

    // (iteration / channel indices are clamped into the tensors: the scalar offset of a buffer load must stay inside the
    //  descriptor for its range check to be meaningful; the clamped loads of the pipeline's overrun are never stored, a
    //  duplicated last channel meets zero-padded weights)
    auto load_stage = [&](int it, int set) {
        const int itc = min(it, niter - 1);
        const uint32_t a_so = (uint32_t)itc * a_gstride4;
#pragma unroll
        for (int i = 0; i < NA; i++) a_reg[set][i] = buf_load4(rw, a_vo[i], a_so);
        const uint32_t c0 = (uint32_t)itc * 4u, cl = (uint32_t)(p.Cin - 1);
#pragma unroll
        for (int k = 0; k < NPOS; k++) {
            x_reg[set][k] = make_float4(buf_load1(rx, pos_xo[k], min(c0, cl) * chw4), buf_load1(rx, pos_xo[k], min(c0 + 1, cl) * chw4),
                                        buf_load1(rx, pos_xo[k], min(c0 + 2, cl) * chw4), buf_load1(rx, pos_xo[k], min(c0 + 3, cl) * chw4));
            if (p.styles) {
                if (cin4) s_reg[set][k] = buf_load4(rs, pos_so[k], c0 * 4u);
                else s_reg[set][k] = make_float4(buf_load1(rs, pos_so[k], min(c0, cl) * 4u), buf_load1(rs, pos_so[k], min(c0 + 1, cl) * 4u),
                                                 buf_load1(rs, pos_so[k], min(c0 + 2, cl) * 4u), buf_load1(rs, pos_so[k], min(c0 + 3, cl) * 4u));
            }
        }
    };
    auto store_stage = [&](int set, float* As, float* Xs) {
#pragma unroll
        for (int i = 0; i < NA; i++)
            if (tid + i * NTH < 9 * BM) *(float4*)(As + (tid + i * NTH) * 4) = a_reg[set][i];
#pragma unroll
        for (int k = 0; k < NPOS; k++)
            if (pos_lds[k] >= 0) {
                if (p.styles) *(float4*)(Xs + pos_lds[k]) = make_float4(x_reg[set][k].x * s_reg[set][k].x, x_reg[set][k].y * s_reg[set][k].y,
                                                                        x_reg[set][k].z * s_reg[set][k].z, x_reg[set][k].w * s_reg[set][k].w);
                else *(float4*)(Xs + pos_lds[k]) = x_reg[set][k];
            }
    };
    const int a_lane = ((wm * MTW) * 32 + l32) * 4 + 2 * half;
    const int b_lane = ((wn * NTW) * 32 + l32) * 4 + 2 * half;      // + (run*XP + dxi + n*32)*4, run 0: dy = -1, dxi 0: dx = -1
    auto mma = [&](int cur) {
        const float* As = smem + cur * BUF_SZ + a_lane;
        const float* Xs = smem + cur * BUF_SZ + AS_SZ + b_lane;
        f32x2 fa[2][MTW], fb[4][NTW];               // fb index = run*2 + dxi
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int n = 0; n < NTW; n++) fb[q][n] = *(const f32x2*)(Xs + ((q >> 1) * XP + (q & 1) + n * 32) * 4);
        auto load_a = [&](int buf, int t) {
#pragma unroll
            for (int m = 0; m < MTW; m++) fa[buf][m] = *(const f32x2*)(As + (t * BM + m * 32) * 4);
        };
        load_a(0, 0);
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int cb = t & 1;
            if (t + 1 < 9) load_a(cb ^ 1, t + 1);
            __builtin_amdgcn_sched_barrier(0);          // keep the prefetch ds_reads above this tap's MFMAs
            const int a = t / 3, e = t % 3;
            const int q = (a & 1) * 2 + (e & 1);                    // output parity of this tap
            const int f = (a == 2 ? 0 : 2) + (e == 2 ? 0 : 1);     // which shifted run it reads
#pragma unroll
            for (int kk = 0; kk < 2; kk++)
#pragma unroll
                for (int m = 0; m < MTW; m++)
#pragma unroll
                    for (int n = 0; n < NTW; n++) acc[q][m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cb][m][kk], fb[f][n][kk], acc[q][m][n], 0, 0, 0);
        }
    };

    float* const L0 = smem;
    float* const L1 = smem + BUF_SZ;
#if TDGP_UP_ABL & 16
    long long tseg[4] = {0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#endif
    if constexpr (!DEEP) {
        // single register set: the loads of iteration i+1 fly while iteration i multiplies
        if (it0 < it1) { load_stage(it0, 0); store_stage(0, L0, L0 + AS_SZ); }
        __syncthreads();
        int cur = 0;
        for (int it = it0; it < it1; it++) {
            const bool more = it + 1 < it1;
            if (more) load_stage(it + 1, 0);
            mma(cur);
            if (more) store_stage(0, smem + (cur ^ 1) * BUF_SZ, smem + (cur ^ 1) * BUF_SZ + AS_SZ);
            __syncthreads();
            cur ^= 1;
        }
    } else {
    load_stage(it0, 0);
    load_stage(it0 + 1, 1);
    if (it0 < it1) store_stage(0, L0, L0 + AS_SZ);
    load_stage(it0 + 2, 0);
    __syncthreads();
#if TDGP_UP_ABL & 16
    tprev = __builtin_readcyclecounter();
#define TSEG(i) { const long long tn = __builtin_readcyclecounter(); tseg[i] += tn - tprev; tprev = tn; }
#else
#define TSEG(i)
#endif
    int it = it0;
    for (; it + 1 < it1; it += 2) {
        // even half: L0 holds iteration `it`; set 1 = iteration it+1, set 0 = iteration it+2 (both in flight).
        // (The loads past the last iteration are issued too -- they fall outside the descriptors or are simply never stored: with
        //  a fixed number of loads per half-iteration and no exit in the middle of the body, the compiler's s_waitcnt in front of
        //  the LDS write is vmcnt(<one stage>) instead of the vmcnt(0) it falls back to otherwise, which un-did the second stage.)
        mma(0);
        TSEG(0)
        store_stage(1, L1, L1 + AS_SZ);
        TSEG(1)
        load_stage(it + 3, 1);
        TSEG(2)
        __syncthreads();
        TSEG(3)
        // odd half: L1 holds iteration it+1; set 0 = iteration it+2, set 1 = iteration it+3
        mma(1);
        TSEG(0)
        if (it + 2 < it1) store_stage(0, L0, L0 + AS_SZ);
        TSEG(1)
        load_stage(it + 4, 0);
        TSEG(2)
        __syncthreads();
        TSEG(3)
    }
    if (it < it1) {                     // odd iteration count: the last one sits in L0 -- and the epilogue's per-wave tiles overlay the
        mma(0);                         // stage buffers: every wave must be done reading them (without this barrier a fast wave's
        __syncthreads();                // epilogue corrupted a slow wave's last fragments: rare, wave-tile-sized errors)
    }
    TSEG(0)
    }
#if TDGP_UP_ABL & 16
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 700) && blockIdx.y == 0 && blockIdx.z == 0)
        printf("upconv blk %d iters %d: mma %lld store %lld load-issue %lld barrier %lld cycles\n", (int)blockIdx.x, it1 - it0, tseg[0], tseg[1], tseg[2], tseg[3]);
#endif
#undef TSEG

    // ---- epilogue: per (channel tile, point subtile, py): interleave px = 0/1 in a per-wave LDS tile, store 16 B per lane -----
    // Vector-ALU instructions of this stage queue behind the 64-cycle MFMAs of the SIMD's other waves (~60 cycles each), and with
    // Cin = 128 the K loop is only 32 iterations: the stage was ~20 % of the 256^2 -> 512^2 layer.  Fast path (whole 32-channel
    // sub-tile inside Cout, slice < 4 GiB): buffer stores whose per-pass offset is a scalar, LDS reads with immediate offsets, one
    // integer division per wave -- no per-pass address arithmetic or bounds compares.
    float* ct = smem + wv * (32 * UP_CT_W);
    float* zout = p.z + (int64_t)ks * p.zslice;
    const bool zbuf = (uint64_t)p.zslice * 4u < 0xFFFF0000ull;
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(zout, zbuf ? (uint32_t)(p.zslice * 4) : 0u);
    const int q = l & 15, cr = l >> 4;
    const uint32_t z_vo = (uint32_t)(cr * 4 * p.GS + 4 * q) * 4u;            // channel cr of the pass: 2 parity rows x 2GS floats each
    const uint32_t z_pass = (uint32_t)(16 * p.GS) * 4u;                      // 4 channels further
    const int gbase = v0 + wn * NTW * 32;
    const int b0 = gbase / p.GS, vpb = gbase - b0 * p.GS;
#pragma unroll 1
    for (int tile = 0; tile < 2 * MTW * NTW; tile++) {
#pragma unroll
        for (int k = 0; k < 2 * MTW * NTW; k++) {
            if (tile == k) {
                constexpr int dummy = 0; (void)dummy;
                const int py = k / (MTW * NTW), mm = (k / NTW) % MTW, nn = k % NTW;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    *(float2*)&ct[row * UP_CT_W + 2 * l32] = make_float2(acc[py * 2][mm][nn][r], acc[py * 2 + 1][mm][nn][r]);
                }
            }
        }
        const int py = tile / (MTW * NTW), mm = (tile / NTW) % MTW, nn = tile % NTW;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int b = b0, vp0 = vpb + nn * 32;                 // GS is a multiple of 32: the 32 points of a sub-tile share their sample
        while (vp0 >= p.GS) { vp0 -= p.GS; b++; }
        const int obase = m0 + (wm * MTW + mm) * 32;
        if (b < p.B) {
            if (zbuf && obase + 32 <= p.Cout) {
                const uint32_t zb = (uint32_t)(((b * p.Cout + obase) * 2 + py) * (2 * p.GS) + 2 * vp0) * 4u;
#pragma unroll
                for (int pass = 0; pass < 8; pass++) {
                    const float4 v = *(const float4*)&ct[(pass * 4 + cr) * UP_CT_W + 4 * q];
                    const u32x4 bits = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                    __builtin_amdgcn_raw_buffer_store_b128(bits, rz, z_vo, zb + (uint32_t)pass * z_pass, 0);
                }
            } else {
#pragma unroll
                for (int pass = 0; pass < 8; pass++) {
                    const int ch = pass * 4 + cr, o = obase + ch;
                    if (o < p.Cout) {
                        const float4 v = *(const float4*)&ct[ch * UP_CT_W + 4 * q];
                        *(float4*)(zout + (((int64_t)b * p.Cout + o) * 2 + py) * (2 * p.GS) + 2 * vp0 + 4 * q) = v;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// =================================================================================================================================
// OPT-IN arithmetic (tdgp_set_conv_arith(1); the default path is the fp32 kernel above): 3x3 stride-1 convolution with every fp32
// operand split into three bf16 pieces (8 + 8 + 8 mantissa bits: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), round to
// nearest; all differences exact) and the product formed from the six leading piece products
//     a*b ~= a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)            dropped terms <= 3 * 2^-24 |a b|
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: fp32-grade results at 16/6 = 2.7x the fp32 MFMA rate.  Not bit-compatible with
// an fp32 FMA chain (neither is the fp32 MFMA with the CPU), hence opt-in.
// Block = 64 output channels x 8 virtual rows x 32 columns, 4 waves (1 x 4), wave tile 2 x 2; K chunk = 16 channels.  Weights are
// pre-split at pack time [chunk16][tap][piece][CoutP][16 bf16]; activations are split while they are staged (style scale first)
// into [piece][position][16 bf16].  A lane's fragment = 8 consecutive channels (16 B) of its column: channels 8*half .. +7 for
// both operands, so the hardware's order of the 16 k's inside the instruction does not matter.  LDS: weight stage 54 KB +
// activation stage 32 KB, single-buffered (one block per CU): the next chunk's loads are in flight during the 216 MFMAs of a chunk.
// =================================================================================================================================
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
struct Conv3sParams {
    const float* x; const void* wsp; const float* styles;
    EpiParams e;
    int B, Cin, Cout, CoutP, H, W;
    uint32_t x_bytes, wsp_bytes, st_bytes;
};

// Compile-time loop (the LDS fragment reads below take their offsets as instruction immediates).
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// LDS fragment read the compiler's waitcnt pass cannot see.  Why: with an LDS-direct buffer load in flight hipcc puts s_waitcnt vmcnt(0)
// in front of every ds_read of the same LDS object (may alias the load's destination) and in front of every __syncthreads() (the
// workgroup release fence) -- i.e. a tap row's weight loads could never fly under the previous row's MFMAs.  These reads, the
// lgkmcnt waits that make their results usable and the barriers are therefore spelled out; vmcnt is waited for explicitly where a
// buffer is about to be read.
template <int OFF> __device__ __forceinline__ bf16x8_t lds_frag(uint32_t addr) {
    bf16x8_t r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
typedef __attribute__((address_space(3))) void* lds_ptr_t;
// LDS addresses are carried as integers (byte offsets in the LDS aperture): a generic -> LDS pointer cast of a computed pointer drags a
// null check along (and tripped a backend assertion in one build); the array itself is a known object, its cast folds to a constant.
#define TDGP_LDS_BASE(arr) ((uint32_t)(uintptr_t)(lds_ptr_t)(arr))
__device__ __forceinline__ lds_ptr_t lds_ptr(uint32_t addr) { return (lds_ptr_t)(uintptr_t)addr; }
// this wave's LDS writes have completed, then the block barrier -- no fence (a fence would wait for the LDS-direct loads in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Two values at a time: v_cvt_pk_bf16_f32 (round to nearest even, already packed as a channel pair), the residual by an exact
// v_pk_add_f32 -- 9 vector-ALU instructions per pair.  x = hi + mid + lo + r with |r| <= 2^-25 |x|.
typedef float split_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 split_b2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(float v0, float v1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    split_f2 v = {v0, v1};
    const split_b2 h = __builtin_convertvector(v, split_b2);
    v = v - __builtin_convertvector(h, split_f2);
    const split_b2 m = __builtin_convertvector(v, split_b2);
    v = v - __builtin_convertvector(m, split_f2);
    const split_b2 l = __builtin_convertvector(v, split_b2);
    __builtin_memcpy(&hi, &h, 4); __builtin_memcpy(&mid, &m, 4); __builtin_memcpy(&lo, &l, 4);
}

__global__ __launch_bounds__(256, 2) void conv3s_mfma_kernel(Conv3sParams p) {
    constexpr int R = 1, MTW = 2, NTW = 2, WN = 4, BM = 64, NT = NTW * WN, PR = NT + 2, PC = 34, PSZ = PR * PC;
    constexpr int AROW_BYTES = 3 * 3 * BM * 32, XS_BYTES = 3 * PSZ * 32;      // one tap row of weights: [dx][piece][64 columns][32 B]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* As = (char*)smem;                                           // two tap-row buffers
    const uint32_t lds0 = TDGP_LDS_BASE(smem);
    char* Xs = As + 2 * AROW_BYTES;
    float* side = (float*)(Xs + XS_BYTES);                          // [1 + NSB][BM]
    float* sty = side + (1 + 4) * BM;                               // [2][Cin]: the styles of the two samples a block can touch
    const int H1 = p.H + R;
    const int tilesX = p.W >> 5, VR = p.B * H1;
    const int ty = blockIdx.x / tilesX, tx = blockIdx.x % tilesX;
    const int vr0 = ty * NT, n0 = tx * 32, m0 = blockIdx.y * BM;
    const int tid = threadIdx.x, l = tid & 63, wv = TDGP_WAVE_INDEX(tid), l32 = l & 31, half = l >> 5;
    const int wn = wv;

    constexpr int NSB = 4;
    const int sb0 = vr0 / H1, sb1 = min(p.B - 1, (vr0 + NT - 1) / H1);
    const bool side_ok = (sb1 - sb0 + 1) <= NSB;
    if (side_ok) {
        for (int i = tid; i < (2 + sb1 - sb0) * BM; i += 256) {
            const int slab = i / BM, o = m0 + i % BM;
            float v = slab == 0 ? 0.f : 1.f;
            if (o < p.e.Cout) {
                if (slab == 0) { if (p.e.bias) v = p.e.bias[o]; }
                else if (p.e.dcoef) v = p.e.dcoef[(sb0 + slab - 1) * p.e.Cout + o];
            }
            side[i] = v;
        }
    }

    // Staging work items = (position of the 10 x 34 patch, channel half): 680 items over 256 threads, half-major, so a wave-wide load
    // reads consecutive positions of one channel plane (few cache lines: the loads are bound by lines per clock in the CU's vector
    // memory path) and the splitting work is even across the waves (whole positions left wave 0 with twice the work of waves 2, 3).
    // The styles of the (at most two) samples a block touches sit in LDS.
    constexpr int NITEM = PSZ * 2, NSLOT = (NITEM + 255) / 256;
    const int b_lo = max(vr0 - R, 0) / H1;
    for (int i = tid; i < 2 * p.Cin; i += 256) {
        const int sb = i >= p.Cin ? 1 : 0;
        sty[i] = p.styles[min(b_lo + sb, p.B - 1) * p.Cin + (i - sb * p.Cin)];
    }
    uint32_t it_xo[NSLOT], it_st[NSLOT], it_ld[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; k++) {
        const int e = tid + k * 256, hf = e >= PSZ ? 1 : 0, pos = e - hf * PSZ;
        it_xo[k] = kOOB; it_st[k] = (uint32_t)(8 * hf); it_ld[k] = (uint32_t)(pos * 32 + hf * 16);
        if (e < NITEM) {
            const int pr = pos / PC, pc = pos % PC;
            const int vi = vr0 - R + pr, ix = n0 - R + pc;
            if (vi >= 0 && ix >= 0 && ix < p.W) {
                const int b = vi / H1, m = vi - b * H1;
                if (b < p.B && m < p.H) {
                    it_xo[k] = (uint32_t)(((b * p.Cin + 8 * hf) * p.H + m) * p.W + ix) * 4u;
                    it_st[k] = (uint32_t)((b - b_lo) * p.Cin + 8 * hf);
                }
            }
        }
    }
    const uint32_t chw4 = (uint32_t)(p.H * p.W) * 4u;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes), rw = make_rsrc(p.wsp, p.wsp_bytes);

    f32x16 acc[MTW][NTW];
#pragma unroll
    for (int m = 0; m < MTW; m++)
#pragma unroll
        for (int n = 0; n < NTW; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;

    const int niter = p.Cin >> 4;                                   // the host only takes this kernel with Cin % 16 == 0
    // Weights: one tap row (18 KB, a verbatim copy of the packed rows) per stage, fetched by LDS-direct buffer loads -- no staging
    // registers, no ds_write: vector e = tid + 256 i lands at byte 16 e of the buffer, i.e. each wave-wide load writes 1 KB at a
    // wave-uniform LDS base.  Tap row g (global index 3 * chunk + ky) lives in buffer g & 1.
    constexpr int NAV = 3 * 3 * BM * 2;                             // 16-B vectors of one tap row of weights
    constexpr int NA = (NAV + 255) / 256;
    float x_reg[NSLOT][8];
    uint32_t a_vo[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int e = tid + i * 256;
        const int tp = e / (BM * 2), rem = e % (BM * 2), col = rem >> 1, hf = rem & 1;
        a_vo[i] = (uint32_t)((tp * p.CoutP + min(m0 + col, p.CoutP - 1)) * 2 + hf) * 16u;      // (columns past CoutP: a duplicate, never stored)
    }
    const uint32_t a_gstride = (uint32_t)(9 * p.CoutP) * 32u;      // one tap row = 3 taps x 3 pieces x CoutP columns x 32 B
    const int nstage = 3 * niter;
    auto load_a = [&](int g, int buf) {                             // stages past the end re-read the last one
        const uint32_t a_so = (uint32_t)min(g, nstage - 1) * a_gstride;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            if (i * 256 + 192 < NAV || i * 256 + wv * 64 < NAV)                            // wave-uniform (NAV is a multiple of 64)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, lds_ptr(lds0 + (uint32_t)(buf * AROW_BYTES + (i * 256 + wv * 64) * 16)), 16, a_vo[i], a_so, 0, 0);
        }
    };
    auto wait_loads = [&]() { __builtin_amdgcn_s_waitcnt(0x0F70); };        // vmcnt(0): the LDS-direct loads have landed
    auto load_x = [&](int it) {
        const uint32_t c0 = (uint32_t)min(it, niter - 1) * 16u;
#pragma unroll
        for (int k = 0; k < NSLOT; k++)
#pragma unroll
            for (int j = 0; j < 8; j++) x_reg[k][j] = buf_load1(rx, it_xo[k], (c0 + j) * chw4);
    };
    auto store_x = [&](int it) {
        const uint32_t c0 = (uint32_t)it * 16u;
        float4 sq[NSLOT][2];
#pragma unroll
        for (int k = 0; k < NSLOT; k++) { sq[k][0] = *(const float4*)&sty[it_st[k] + c0]; sq[k][1] = *(const float4*)&sty[it_st[k] + c0 + 4]; }
#pragma unroll
        for (int k = 0; k < NSLOT; k++) {
            uint32_t pk[3][4];
            split3_pair(x_reg[k][0] * sq[k][0].x, x_reg[k][1] * sq[k][0].y, pk[0][0], pk[1][0], pk[2][0]);
            split3_pair(x_reg[k][2] * sq[k][0].z, x_reg[k][3] * sq[k][0].w, pk[0][1], pk[1][1], pk[2][1]);
            split3_pair(x_reg[k][4] * sq[k][1].x, x_reg[k][5] * sq[k][1].y, pk[0][2], pk[1][2], pk[2][2]);
            split3_pair(x_reg[k][6] * sq[k][1].z, x_reg[k][7] * sq[k][1].w, pk[0][3], pk[1][3], pk[2][3]);
            if (tid + k * 256 < NITEM) {
#pragma unroll
                for (int pc_ = 0; pc_ < 3; pc_++) *(uint4*)(Xs + pc_ * (PSZ * 32) + it_ld[k]) = make_uint4(pk[pc_][0], pk[pc_][1], pk[pc_][2], pk[pc_][3]);
            }
        }
    };
    auto frag = [&](const char* ptr) { const uint4 v = *(const uint4*)ptr; bf16x8_t r; __builtin_memcpy(&r, &v, 16); return r; };
    const int a_lane = l32 * 32 + half * 16;                                        // + ((dx*3 + piece)*BM + m*32) * 32
    const int b_lane = ((wn * NTW + R) * PC + l32 + R) * 32 + half * 16;           // centre tap of subtile 0; + (piece*PSZ + (n + dy)*PC + dx) * 32
    auto mma_row = [&](int row, int buf) {
        const int dy = row - R;
#pragma unroll
        for (int tx_ = 0; tx_ < 3; tx_++) {
            const int dx = tx_ - R;
            bf16x8_t fa[3][MTW], fb[3][NTW];
#pragma unroll
            for (int pc_ = 0; pc_ < 3; pc_++) {
#pragma unroll
                for (int m = 0; m < MTW; m++) fa[pc_][m] = frag(As + buf * AROW_BYTES + a_lane + ((tx_ * 3 + pc_) * BM + m * 32) * 32);
#pragma unroll
                for (int n = 0; n < NTW; n++) fb[pc_][n] = frag(Xs + b_lane + (pc_ * PSZ + (n + dy) * PC + dx) * 32);
            }
            // piece products outermost (smallest terms first), the four tiles inside: consecutive MFMAs are independent -- with the
            // six products of one tile back to back each instruction waited for the result of the one before
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; q++)
#pragma unroll
                for (int m = 0; m < MTW; m++)
#pragma unroll
                    for (int n = 0; n < NTW; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]][m], fb[PB[q]][n], acc[m][n], 0, 0, 0);
        }
    };

#if TDGP_C3_ABL & 32
    long long ts[5] = {0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define TS(i) { const long long tn_ = __builtin_readcyclecounter(); ts[i] += tn_ - tprev; tprev = tn_; }
#else
#define TS(i)
#endif
    // Two blocks per CU (LDS 75 KB each): one block's staging and barriers run under the other's MFMAs.  Per chunk of 16 channels:
    // the activation stage once, the weights one tap row at a time; one barrier per tap row, behind which the other weight buffer is
    // free for the next row's loads, which land under this row's MFMAs.
    load_x(0);
    load_a(0, 0);
    TS(0)
    for (int it = 0; it < niter; it++) {
        const int g = 3 * it, b0 = g & 1;
        wait_loads();                           // tap row g has landed (and the activation registers)
        __syncthreads();                        // ... for every wave; the previous chunk's fragments have been read
        TS(1)
        store_x(it);
        TS(2)
        load_a(g + 1, b0 ^ 1);
        load_x(it + 1);                         // in flight during this chunk's MFMAs
        TS(3)
        __syncthreads();
        TS(1)
        mma_row(0, b0);
        TS(4)
        wait_loads();
        __syncthreads();
        TS(1)
        load_a(g + 2, b0);
        TS(3)
        mma_row(1, b0 ^ 1);
        TS(4)
        wait_loads();
        __syncthreads();
        TS(1)
        load_a(g + 3, b0 ^ 1);
        TS(3)
        mma_row(2, b0);
        TS(4)
    }
    wait_loads();
    __syncthreads();
    TS(0)

    float* ct = smem + wv * (32 * CT_LD);
    const EpiParams& e = p.e;
    SideCache scache;
    scache.lds = side_ok ? side : nullptr; scache.b0 = sb0; scache.bm = BM; scache.m0 = m0;
    const int evar = epi_variant(e);
#pragma unroll 1
    for (int tile = 0; tile < MTW * NTW; tile++) {
#pragma unroll
        for (int k = 0; k < MTW * NTW; k++) {
            if (tile == k) {
                constexpr int dummy = 0; (void)dummy;
#pragma unroll
                for (int r = 0; r < 16; r++) ct[((r & 3) + 8 * (r >> 2) + 4 * half) * CT_LD + l32] = acc[k % MTW][k / MTW][r];
            }
        }
        const int m = tile % MTW, n = tile / MTW;
        const int vi = vr0 + wn * NTW + n;
        const int pb = vi / H1, poy = vi - pb * H1;
        const int pok = (vi < VR && poy < p.H) ? 1 : 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (evar == 3) epilogue_tile<false, 3>(e, scache, ct, m0 + m * 32, pb, poy, n0 + l32, pok, nullptr, true);
        else if (evar == 1) epilogue_tile<false, 1>(e, scache, ct, m0 + m * 32, pb, poy, n0 + l32, pok, nullptr, true);
        else epilogue_tile<false, 0>(e, scache, ct, m0 + m * 32, pb, poy, n0 + l32, pok, nullptr, true);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#if TDGP_C3_ABL & 32
    TS(1)
    if (tid == 0 && (blockIdx.x == 3 || blockIdx.x == 400) && blockIdx.y == 0)
        printf("conv3s blk %d iters %d: prologue %lld barrier+epilogue %lld store(+load wait) %lld load-issue %lld mma %lld\n", (int)blockIdx.x, niter, ts[0], ts[1], ts[2], ts[3], ts[4]);
#endif
#undef TS
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same arithmetic for the x2 layers: upconv_mfma_kernel's linearised-grid transposed convolution (same Z layout, so the FIR /
// output stage is shared) on the bf16 MFMA.  Block = 64 output channels x 128 consecutive grid points, 4 waves (1 x 4), a wave
// owns 32 points x 64 channels x the four output parities (8 accumulator tiles); K chunk = 16 channels, the two activation runs
// (dy = -1, dy = 0; 129 positions each) staged once per chunk, the weights one tap row (= one ky) at a time through two buffers:
// 62 KB of LDS, two blocks per CU.  No split-K (these launches have >= 256 blocks).
// ---------------------------------------------------------------------------------------------------------------------------------
struct Up3sParams {
    const float* x; const void* wsp; const float* styles; float* z;
    int B, Cin, Cout, CoutP, H, W, G1, GS;
    int64_t zslice;
    uint32_t x_bytes, wsp_bytes;
};

__global__ __launch_bounds__(256, 2) void upconv3s_mfma_kernel(Up3sParams p) {
    constexpr int MTW = 2, BM = 64, BN = 128, XP = BN + 2;
    constexpr int AROW_BYTES = 3 * 3 * BM * 32, XPIECE = 2 * XP * 32, XS_BYTES = 3 * XPIECE;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* As = (char*)smem;                                           // two tap-row buffers (the epilogue's tiles overlay them)
    const uint32_t lds0 = TDGP_LDS_BASE(smem), xraw0 = lds0 + (uint32_t)(2 * AROW_BYTES + XS_BYTES);
    char* Xs = As + 2 * AROW_BYTES;                                   // [piece][run][position][16 bf16]
    float* Xraw = (float*)(Xs + XS_BYTES);                            // the next chunk's fp32 activations as they arrive: [slot][channel][thread]
    float* styc = Xraw + (2 * 8 * 256 + 8 * 64);                      // [chunk parity][2 samples][16 channels] (+ 32 floats of slack per buffer)
    const int v0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int tid = threadIdx.x, l = tid & 63, wv = TDGP_WAVE_INDEX(tid), l32 = l & 31, half = l >> 5;
    const int wn = wv;

    const int b_lo = max(v0 - 1 - p.G1, 0) / p.GS;
    // staging items = (channel half, run, position): 2 x 2 x 129, half-major (a wave-wide load = consecutive positions of one plane)
    constexpr int NRUN = BN + 1, NITEM = 2 * 2 * NRUN, NSLOT = (NITEM + 255) / 256;
    uint32_t it_xo[NSLOT], it_st[NSLOT], it_ld[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; k++) {
        const int e = tid + k * 256;
        const int hf = e / (2 * NRUN), rem = e - hf * (2 * NRUN), seg = rem / NRUN, idx = rem - seg * NRUN;
        it_xo[k] = kOOB; it_st[k] = (uint32_t)(8 * (hf & 1)); it_ld[k] = (uint32_t)((seg * XP + idx) * 32 + (hf & 1) * 16);
        if (e < NITEM) {
            const int g = v0 + idx - 1 - (seg == 0 ? p.G1 : 0);
            if (g >= 0) {
                const int b = g / p.GS, vp = g - b * p.GS;
                const int m = vp / p.G1, n = vp - m * p.G1;
                if (b < p.B && m < p.H && n < p.W) {
                    it_xo[k] = (uint32_t)(((b * p.Cin + 8 * hf) * p.H + m) * p.W + n) * 4u;
                    it_st[k] = (uint32_t)((b - b_lo) * 16 + 8 * hf);
                }
            }
        }
    }
    const uint32_t chw4 = (uint32_t)(p.H * p.W) * 4u;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes), rw = make_rsrc(p.wsp, p.wsp_bytes), rs = make_rsrc(p.styles, (uint32_t)(p.B * p.Cin) * 4u);
    const uint32_t s_vo = (uint32_t)(min(b_lo + ((l >> 4) & 1), p.B - 1) * p.Cin + (l & 15)) * 4u;       // styles of the chunk: lane = (sample, channel)

    f32x16 acc[4][MTW];                                               // [py*2+px][channel tile]
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int m = 0; m < MTW; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[q][m][r] = 0.f;

    const int niter = p.Cin >> 4;                                   // the host only takes this kernel with Cin % 16 == 0
    // Weights: one tap row (18 KB, a verbatim copy of the packed rows) per stage, fetched by LDS-direct buffer loads (no staging
    // registers -- with 128 accumulator registers there are none to spare -- and no ds_write): vector e = tid + 256 i lands at
    // byte 16 e of the buffer, i.e. each wave-wide load writes 1 KB at a wave-uniform LDS base.
    constexpr int NAV = 3 * 3 * BM * 2, NA = (NAV + 255) / 256;
    uint32_t a_vo[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int e = tid + i * 256;
        const int tp = e / (BM * 2), rem = e % (BM * 2), col = rem >> 1, hf = rem & 1;
        a_vo[i] = (uint32_t)((tp * p.CoutP + min(m0 + col, p.CoutP - 1)) * 2 + hf) * 16u;
    }
    const uint32_t a_gstride = (uint32_t)(9 * p.CoutP) * 32u;
    const int nstage = 3 * niter;
    auto load_a = [&](int g, int buf) {
        const uint32_t a_so = (uint32_t)min(g, nstage - 1) * a_gstride;
#pragma unroll
        for (int i = 0; i < NA; i++) {
            if (i * 256 + 192 < NAV || i * 256 + wv * 64 < NAV)        // wave-uniform (NAV is a multiple of 64)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, lds_ptr(lds0 + (uint32_t)(buf * AROW_BYTES + (i * 256 + wv * 64) * 16)), 16, a_vo[i], a_so, 0, 0);
        }
    };
    auto wait_loads = [&]() { __builtin_amdgcn_s_waitcnt(0x0F70); };        // vmcnt(0): the LDS-direct loads have landed
    // The activations never sit in registers across the MFMAs (24 loop-carried registers next to the 128 accumulators made the compiler
    // spill accumulator tiles): LDS-direct loads drop the next chunk's fp32 values into Xraw, the split reads them back from there.
    // Out-of-range positions carry an out-of-bounds offset and arrive as zeros.
    auto load_x = [&](int it) {
        const uint32_t c0 = (uint32_t)min(it, niter - 1) * 16u;
#pragma unroll
        for (int k = 0; k < NSLOT; k++) {
            if (k * 256 + 192 < NITEM || k * 256 + wv * 64 < NITEM) {                        // wave-uniform: slot 2 has items in wave 0 only
                const uint32_t dst = xraw0 + (uint32_t)(k < 2 ? (k * 8) * 256 + wv * 64 : 2 * 8 * 256) * 4u;
#pragma unroll
                for (int j = 0; j < 8; j++)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, lds_ptr(dst + (uint32_t)(j * (k < 2 ? 256 : 64)) * 4u), 4, it_xo[k], (c0 + j) * chw4, 0, 0);
            }
        }
        if (wv == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lds_ptr(xraw0 + (uint32_t)(2 * 8 * 256 + 8 * 64 + (it & 1) * 64) * 4u), 4, s_vo, c0 * 4u, 0, 0);
    };
    auto store_x = [&](int it) {
        const float* sc = styc + (it & 1) * 64;
#pragma unroll
        for (int k = 0; k < NSLOT; k++) {
            if (k * 256 + 192 < NITEM || k * 256 + wv * 64 < NITEM) {
                const float* src = Xraw + (k < 2 ? (k * 8) * 256 + tid : 2 * 8 * 256 + l);
                float xv[8];
#pragma unroll
                for (int j = 0; j < 8; j++) xv[j] = src[j * (k < 2 ? 256 : 64)];
                const float4 s0 = *(const float4*)&sc[it_st[k]], s1 = *(const float4*)&sc[it_st[k] + 4];
                uint32_t pk[3][4];
                split3_pair(xv[0] * s0.x, xv[1] * s0.y, pk[0][0], pk[1][0], pk[2][0]);
                split3_pair(xv[2] * s0.z, xv[3] * s0.w, pk[0][1], pk[1][1], pk[2][1]);
                split3_pair(xv[4] * s1.x, xv[5] * s1.y, pk[0][2], pk[1][2], pk[2][2]);
                split3_pair(xv[6] * s1.z, xv[7] * s1.w, pk[0][3], pk[1][3], pk[2][3]);
                if (tid + k * 256 < NITEM) {
#pragma unroll
                    for (int pc_ = 0; pc_ < 3; pc_++) *(uint4*)(Xs + pc_ * XPIECE + it_ld[k]) = make_uint4(pk[pc_][0], pk[pc_][1], pk[pc_][2], pk[pc_][3]);
                }
            }
        }
    };
    const uint32_t a_addr0 = lds0 + (uint32_t)(l32 * 32 + half * 16);               // + buf*AROW_BYTES; immediates: ((e*3 + piece)*BM + m*32) * 32
    const uint32_t x_addr = lds0 + (uint32_t)(2 * AROW_BYTES) + (uint32_t)((wn * 32 + l32) * 32 + half * 16);    // immediates: piece*XPIECE + (run*XP + dxi) * 32
    // tap row a = ky: Z[2m+py, 2n+px] += w[a,e] x[m - (a==2), n - (e==2)].  The next tap's nine fragments are requested before this
    // tap's twelve MFMAs are issued (lgkmcnt(9): this tap's have arrived, the next tap's may still be on their way).
    auto mma_row = [&](auto A_, int buf) {
        constexpr int a = decltype(A_)::value;
        const uint32_t a_addr = a_addr0 + (uint32_t)(buf * AROW_BYTES);
        bf16x8_t fr[2][9];                                           // [tap parity][fa(piece, m) = 2*piece + m | fb(piece) = 6 + piece]
        auto request = [&](auto E_, auto S_) {
            constexpr int e = decltype(E_)::value, st = decltype(S_)::value;
            constexpr int run = a == 2 ? 0 : 1, dxi = e == 2 ? 0 : 1;
            static_for<0, 3>([&](auto P_) {
                constexpr int pc = decltype(P_)::value;
                fr[st][2 * pc] = lds_frag<((e * 3 + pc) * BM) * 32>(a_addr);
                fr[st][2 * pc + 1] = lds_frag<((e * 3 + pc) * BM + 32) * 32>(a_addr);
                fr[st][6 + pc] = lds_frag<pc * XPIECE + (run * XP + dxi) * 32>(x_addr);
            });
        };
        request(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        static_for<0, 3>([&](auto E_) {
            constexpr int e = decltype(E_)::value, st = e & 1;
            constexpr int q = (a & 1) * 2 + (e & 1);
            if constexpr (e < 2) {
                request(std::integral_constant<int, e + 1>{}, std::integral_constant<int, st ^ 1>{});
                asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(fr[st][0]), "+v"(fr[st][1]), "+v"(fr[st][2]), "+v"(fr[st][3]), "+v"(fr[st][4]), "+v"(fr[st][5]),
                             "+v"(fr[st][6]), "+v"(fr[st][7]), "+v"(fr[st][8]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[st][0]), "+v"(fr[st][1]), "+v"(fr[st][2]), "+v"(fr[st][3]), "+v"(fr[st][4]), "+v"(fr[st][5]),
                             "+v"(fr[st][6]), "+v"(fr[st][7]), "+v"(fr[st][8]));
            }
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int pp = 0; pp < 6; pp++)
#pragma unroll
                for (int m = 0; m < MTW; m++) acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[st][2 * PA[pp] + m], fr[st][6 + PB[pp]], acc[q][m], 0, 0, 0);
        });
    };
    constexpr std::integral_constant<int, 0> R0{}; constexpr std::integral_constant<int, 1> R1{}; constexpr std::integral_constant<int, 2> R2{};

    // Tap row g (global index 3 * chunk + ky) lives in buffer g & 1.  One barrier per tap row: behind it every wave has finished the
    // previous row, so the other buffer is free for the LDS-direct loads of the next row, which land under this row's MFMAs.
#if TDGP_C3_ABL & 32
    long long ts[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define TS(i) { const long long tn_ = __builtin_readcyclecounter(); ts[i] += tn_ - tprev; tprev = tn_; }
#else
#define TS(i)
#endif
    load_a(0, 0);
    load_x(0);
    TS(0)
    for (int it = 0; it < niter; it++) {
        const int g = 3 * it, b0 = g & 1;
        wait_loads();                           // tap row g, the chunk's activations and styles have landed
        lds_barrier();                          // ... for every wave; the previous chunk's fragments have been read
        TS(1)
        store_x(it);
        TS(2)
        load_a(g + 1, b0 ^ 1);                  // lands under tap row g's MFMAs
        TS(3)
        lds_barrier();                          // the split activations are in place, Xraw is free again
        TS(1)
        mma_row(R0, b0);
        TS(4)
        wait_loads();
        lds_barrier();
        TS(1)
        load_a(g + 2, b0);
        load_x(it + 1);                         // younger than the weight loads: the wait below leaves them in flight
        TS(3)
        mma_row(R1, b0 ^ 1);
        TS(4)
        if (wv == 0) __builtin_amdgcn_s_waitcnt(0x4F79); else __builtin_amdgcn_s_waitcnt(0x4F70);       // vmcnt(25 | 16): tap row g+2 has landed
        lds_barrier();
        TS(1)
        load_a(g + 3, b0 ^ 1);
        TS(3)
        mma_row(R2, b0);
        TS(4)
    }
    wait_loads();
    __syncthreads();                                                // every wave is done with the stage buffers the tiles below overlay

    // ---- epilogue (as in upconv_mfma_kernel): interleave px = 0/1 through a per-wave LDS tile, 16-B buffer stores into Z ----
    float* ct = smem + wv * (32 * UP_CT_W);
    const bool zbuf = (uint64_t)p.zslice * 4u < 0xFFFF0000ull;
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(p.z, zbuf ? (uint32_t)(p.zslice * 4) : 0u);
    const int q4 = l & 15, cr = l >> 4;
    const uint32_t z_vo = (uint32_t)(cr * 4 * p.GS + 4 * q4) * 4u;
    const uint32_t z_pass = (uint32_t)(16 * p.GS) * 4u;
    const int gbase = v0 + wn * 32;
    const int b = gbase / p.GS, vp0 = gbase - b * p.GS;                // GS is a multiple of 32: the 32 points of a wave share their sample
#pragma unroll 1
    for (int tile = 0; tile < 2 * MTW; tile++) {
#pragma unroll
        for (int k = 0; k < 2 * MTW; k++) {
            if (tile == k) {
                constexpr int dummy = 0; (void)dummy;
                const int py = k / MTW, mm = k % MTW;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    *(float2*)&ct[row * UP_CT_W + 2 * l32] = make_float2(acc[py * 2][mm][r], acc[py * 2 + 1][mm][r]);
                }
            }
        }
        const int py = tile / MTW, mm = tile % MTW;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int obase = m0 + mm * 32;
        if (b < p.B) {
            if (zbuf && obase + 32 <= p.Cout) {
                const uint32_t zb = (uint32_t)(((b * p.Cout + obase) * 2 + py) * (2 * p.GS) + 2 * vp0) * 4u;
#pragma unroll
                for (int pass = 0; pass < 8; pass++) {
                    const float4 v = *(const float4*)&ct[(pass * 4 + cr) * UP_CT_W + 4 * q4];
                    const u32x4 bits = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                    __builtin_amdgcn_raw_buffer_store_b128(bits, rz, z_vo, zb + (uint32_t)pass * z_pass, 0);
                }
            } else {
#pragma unroll
                for (int pass = 0; pass < 8; pass++) {
                    const int ch = pass * 4 + cr, o = obase + ch;
                    if (o < p.Cout) {
                        const float4 v = *(const float4*)&ct[ch * UP_CT_W + 4 * q4];
                        *(float4*)(p.z + (((int64_t)b * p.Cout + o) * 2 + py) * (2 * p.GS) + 2 * vp0 + 4 * q4) = v;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#if TDGP_C3_ABL & 32
    TS(5)
    if (tid == 0 && (blockIdx.x == 3 || blockIdx.x == 400) && blockIdx.y == 0)
        printf("upconv3s blk %d iters %d: prologue %lld wait+barrier %lld store %lld load-issue %lld mma %lld epilogue %lld\n", (int)blockIdx.x, niter, ts[0], ts[1], ts[2], ts[3], ts[4], ts[5]);
#endif
#undef TS
}

// weight [Cout,Cin,3,3] -> split pack [chunk16][tap][piece][CoutP][16 bf16] (zero beyond Cout / Cin)
__global__ __launch_bounds__(256) void pack_split_kernel(const float* __restrict__ w, uint32_t* __restrict__ wsp, int Cout, int Cin, int CoutP, int niter) {
    const int64_t total = (int64_t)niter * 9 * 3 * CoutP * 8;                 // u32 words (2 bf16 each)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int jp = (int)(i % 8);                                          // channel pair inside the column's 16 channels
        int64_t r = i / 8;
        const int col = (int)(r % CoutP); r /= CoutP;
        const int piece = (int)(r % 3); r /= 3;
        const int tap = (int)(r % 9);
        const int it = (int)(r / 9);
        float v[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int c = it * 16 + 2 * jp + q;
            v[q] = (col < Cout && c < Cin) ? w[((int64_t)col * Cin + c) * 9 + tap] : 0.f;
        }
        uint32_t pc[3];
        split3_pair(v[0], v[1], pc[0], pc[1], pc[2]);
        wsp[i] = pc[piece];
    }
}

// Output stage of the ToRGB kernel for one 32(pixels) x 32(channels) tile parked PIXEL-major in `ct`: bias + x2-FIR-upsampled skip
// (upfirdn2d.py:313-348), * gain, clamp, channel-last float4 stores.  lane = (pixel l>>3 of the pass, channel quad l&7).
// One straight-line block -- 16 tap loads, 4 LDS reads, the math, 4 stores -- with selects instead of branches: with the
// run-time activation switch of the generic epilogue in here the compiler put a full `s_waitcnt vmcnt(0)` in front of every
// store, i.e. each of the 12 stores of a 128-pixel tile waited for the acknowledgement of the one before (per-phase counters,
// TDGP_RGB_ABL=16: output stage 24-36 k cycles per tile against 6 k of MFMA).  ToRGB is linear and has no noise / demodulation.
// What is left is vector-ALU instruction count: while the SIMD's other wave is in its MFMA phase an instruction of this stage
// gets in once per 64-cycle MFMA, so the blend runs as v_pk_fma_f32 (two channels per slot, the same fma chain per element),
// gain / clamp are compiled out when they are 1 / off (PLAIN), and for power-of-two images (POW2: lw = log2 W, lhw = log2 HW)
// the pixel coordinates come from shifts of the pixel index instead of 16 cross-lane reads.
typedef float rgb_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int rgb_u32x2 __attribute__((ext_vector_type(2)));
// G = passes handled together (4: all 16 taps in flight; 2: half the registers, for the multi-tile kernel that keeps the next
// tile's activations in registers across this stage).
template <bool SKIP, bool PLAIN, bool POW2, int G>
__device__ __forceinline__ void rgb_output_tile(const EpiParams& e, const float* __restrict__ bias_lds, const float* ct, int obase, int pb, int poy, int pox,
                                                int pok, int64_t pix0, int64_t P, int lw, int lhw) {
    const int l = lane_id(), cg = l & 7, pr = l >> 3;
    const int o = obase + 4 * cg;
    const bool okc = o < e.Cout;                                                // Cout and out_feat are multiples of 4 (host check)
    const int oc = okc ? o : 0;
    const int pl = oc / e.out_feat, f = oc - pl * e.out_feat;
    const int planes = e.Cout / e.out_feat;
    const int h2 = e.Hout / 2, w2 = e.Wout / 2;
    const float4 bias4 = *(const float4*)(bias_lds + oc);
    const rgb_f32x2 b01 = {bias4.x, bias4.y}, b23 = {bias4.z, bias4.w};
    const float lim = e.clamp >= 0.f ? e.clamp : __builtin_inff();
#pragma unroll
  for (int g0 = 0; g0 < 4; g0 += G) {
    float4 ta[G], tb[G], tc[G], td[G];
    SkipTaps tp[G];
    int addr[G];
    bool okp[G];
#pragma unroll
    for (int pass = 0; pass < G; pass++) {
        const int px = (g0 + pass) * 8 + pr;
        int b, oy, ox;
        if (POW2) {                                                              // pixel px of this wave's tile: pix0 + 4 * px
            const int pix = (int)pix0 + 4 * px;                                  // B*H*W < 2^31 - 512 (host check)
            const bool ok = pix < (int)P;
            const int pc = ok ? pix : 0;
            b = pc >> lhw;
            const int inner = pc & ((1 << lhw) - 1);
            oy = inner >> lw; ox = inner & ((1 << lw) - 1);
            okp[pass] = ok && okc;
        } else {
            b = __shfl(pb, px, 64); oy = __shfl(poy, px, 64); ox = __shfl(pox, px, 64);
            okp[pass] = __shfl(pok, px, 64) && okc;
        }
        const int plane = b * planes + pl;
        addr[pass] = ((plane * e.Hout + oy) * e.Wout + ox) * e.out_feat + f;
        if (SKIP) {
            tp[pass] = skip_taps(h2, w2, oy, ox, e.fir);
            const float* sp = e.skip + (int64_t)plane * h2 * w2 * e.out_feat + f;
#if TDGP_RGB_ABL & 32                   // timing experiment: the taps from LDS (whatever sits there) -- what an LDS-staged texel tile could cost at best
            const float4* lt = (const float4*)bias_lds;
            ta[pass] = lt[(tp[pass].i00 * 3 + cg) & 63]; tb[pass] = lt[(tp[pass].i01 * 3 + cg) & 63];
            tc[pass] = lt[(tp[pass].i10 * 3 + cg) & 63]; td[pass] = lt[(tp[pass].i11 * 3 + cg) & 63];
#else
            ta[pass] = *(const float4*)(sp + (int64_t)tp[pass].i00 * e.out_feat); tb[pass] = *(const float4*)(sp + (int64_t)tp[pass].i01 * e.out_feat);
            tc[pass] = *(const float4*)(sp + (int64_t)tp[pass].i10 * e.out_feat); td[pass] = *(const float4*)(sp + (int64_t)tp[pass].i11 * e.out_feat);
#endif
        }
    }
    float4 v[G];
#pragma unroll
    for (int pass = 0; pass < G; pass++) {
        const float4 c4 = *(const float4*)&ct[((g0 + pass) * 8 + pr) * CT_LD + 4 * cg];
        rgb_f32x2 r01, r23;
        if (!PLAIN) {                           // conv -> (bf16) -> + bias -> * gain -> clamp -> (bf16), then the skip (networks_stylegan2.py:170-171, 265-269)
            float q[4] = {c4.x, c4.y, c4.z, c4.w};
            const float bq[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float t = q[i];
                if (e.round_bf16) t = (float)(__bf16)t;
                t = (t + (e.round_bf16 ? (float)(__bf16)bq[i] : bq[i])) * e.gain;
                t = t < -lim ? -lim : (t > lim ? lim : t);
                q[i] = e.round_bf16 ? (float)(__bf16)t : t;
            }
            r01 = (rgb_f32x2){q[0], q[1]}; r23 = (rgb_f32x2){q[2], q[3]};
        } else {
            r01 = (rgb_f32x2){c4.x, c4.y} + b01; r23 = (rgb_f32x2){c4.z, c4.w} + b23;
        }
        if (SKIP) {
            const SkipTaps& t = tp[pass];
            const rgb_f32x2 w00 = {t.w00, t.w00}, w01 = {t.w01, t.w01}, w10 = {t.w10, t.w10}, w11 = {t.w11, t.w11};
            const rgb_f32x2 a01 = {ta[pass].x, ta[pass].y}, a23 = {ta[pass].z, ta[pass].w}, bb01 = {tb[pass].x, tb[pass].y}, bb23 = {tb[pass].z, tb[pass].w};
            const rgb_f32x2 cc01 = {tc[pass].x, tc[pass].y}, cc23 = {tc[pass].z, tc[pass].w}, d01 = {td[pass].x, td[pass].y}, d23 = {td[pass].z, td[pass].w};
            r01 = r01 + __builtin_elementwise_fma(w11, d01, __builtin_elementwise_fma(w10, cc01, __builtin_elementwise_fma(w01, bb01, w00 * a01)));
            r23 = r23 + __builtin_elementwise_fma(w11, d23, __builtin_elementwise_fma(w10, cc23, __builtin_elementwise_fma(w01, bb23, w00 * a23)));
        }
        v[pass] = make_float4(r01.x, r01.y, r23.x, r23.y);
    }
#pragma unroll
    for (int pass = 0; pass < G; pass++)
        if (okp[pass]) *(float4*)(e.y + addr[pass]) = v[pass];
  }
}

// The skip form of the stage above as a PIPELINE over the MT channel tiles x 2 half tiles (round 6).  On this ISA stores count in vmcnt and complete
// out of order with loads, so a batch of tap loads issued BEHIND the previous batch's stores is answered with a wait that also waits for those stores to be
// acknowledged: the stage ran as MT x (tap round trip + store round trip), 24 k cycles per 128-pixel tile against 6 k of multiply (and twice as many, smaller
// batches were 13 % slower still: round 3).  Here a batch is ONE pass (4 tap loads, one store) and the loads of batches b + 1, b + 2 are issued BEFORE batch b's
// store -- three register sets of 4 taps, 48 registers where the 4-pass form held 64 --, so the wait for a batch never has a store to wait for and every tap round
// trip but the first runs under two batches of arithmetic and stores.  Same taps, same fma chains, same bits.
typedef float rgb_f32x4 __attribute__((ext_vector_type(4)));
// PARKED: all MT channel tiles already sit in LDS, tile k at ct + k * 32 * CT_LD (torgb_ws_kernel: another wave put them there); else `acc` is parked tile by tile into `ct`.
#ifndef TDGP_RGB_PRE_STREAMED
#define TDGP_RGB_PRE_STREAMED 1        // per-pass geometry once per tile in the streamed one-role kernels too (A/B builds)
#endif
#ifndef TDGP_RGB_NSET_STREAMED
#define TDGP_RGB_NSET_STREAMED 5       // register sets of taps in the one-role kernels that stream their weights (they have registers to spare; A/B builds)
#endif
// B0, B1 (PARKED only): the batches [B0, B1) of the tile's 4 MT -- the two roles of torgb_ws_kernel share a tile's stage.
template <int MT, bool PLAIN, bool PARKED = false, int NSET1 = 3, bool PRE1 = false, int B0 = 0, int B1 = 4 * MT>
__device__ __forceinline__ void rgb_output_skip_pipelined(const EpiParams& e, const float* __restrict__ bias_lds, float* ct, f32x16 (&acc)[MT], int64_t pix0, int64_t P, int lw, int lhw) {
    const int l = lane_id(), cg = l & 7, pr = l >> 3, l32 = l & 31, half = l >> 5;
    const int planes = e.Cout / e.out_feat, h2 = e.Hout / 2, w2 = e.Wout / 2;
    const float lim = e.clamp >= 0.f ? e.clamp : __builtin_inff();
    // The tap loads are issued BY HAND (as the field kernel's): written as C++ loads the compiler answered the loop-carried register set with `vmcnt(0)` in front of
    // every use (ISA of the first build) -- its wait counts cannot see across the back edge what is still in flight.  The waits are hand-counted: when batch b is
    // finished the wave has, in issue order, the 4 loads of b, the stores of earlier batches, and the 4 A loads of the A batches ahead (A = NSET - 1) in flight; loads
    // complete in order among themselves, stores at any time, so `vmcnt(4 A)` -- at most the loads of the batches ahead left -- guarantees b's taps (with slow stores it
    // also waits for a few loads of the next batch; it never waits for a store to be acknowledged when the loads are there).  The last A batches wait with fewer, the last
    // with `vmcnt(0)`.  3dgp_amd/isa_check.py: check_torgb_asm holds the generated code to exactly this pattern.
    // the FIR taps as 16 scalar registers: left in memory the compiler folds the parity selects back into one indexed load per weight
    FirRegs fr = {e.fir[0], e.fir[1], e.fir[2], e.fir[3], e.fir[4], e.fir[5], e.fir[6], e.fir[7], e.fir[8], e.fir[9], e.fir[10], e.fir[11], e.fir[12], e.fir[13], e.fir[14], e.fir[15]};
    asm volatile("" : "+s"(fr.f0), "+s"(fr.f1), "+s"(fr.f2), "+s"(fr.f3), "+s"(fr.f4), "+s"(fr.f5), "+s"(fr.f6), "+s"(fr.f7));
    asm volatile("" : "+s"(fr.f8), "+s"(fr.f9), "+s"(fr.f10), "+s"(fr.f11), "+s"(fr.f12), "+s"(fr.f13), "+s"(fr.f14), "+s"(fr.f15));
#ifndef TDGP_RGB_WS_NSET
#define TDGP_RGB_WS_NSET 6
#endif
    // a batch = one pass (8 pixels x 32 channels per wave): 4 tap loads, one store; NSET register sets = NSET - 1 batches of look-ahead (three sets in the one-role
    // kernels, which sit at their register limit; six for the memory waves of the two-role kernel, whose whole stage is these round trips)
    constexpr int NSET = PARKED ? (B1 - B0 < TDGP_RGB_WS_NSET ? (B1 - B0 > 1 ? B1 - B0 : 2) : TDGP_RGB_WS_NSET) : NSET1, NB = B1;
    static_assert(PARKED || (B0 == 0 && B1 == 4 * MT), "a batch sub-range needs the parked form");
    rgb_f32x4 tA[NSET][4];
    // (pixel geometry, tap indices and weights are RECOMPUTED where the batch is finished -- two dozen scalar-ish vector instructions per pass -- instead of
    //  carried beside the taps: registers the kernel does not have)
    struct Geo { int addr; bool ok; SkipTaps tp; const float* sp; };
    auto geo = [&](int tile, int pass) {
        Geo g;
        const int o = tile * 32 + 4 * cg;
        const bool okc = o < e.Cout;
        const int oc = okc ? o : 0;
        const int pl = oc / e.out_feat, f = oc - pl * e.out_feat;
        const int px = pass * 8 + pr;
        const int pix = (int)pix0 + 4 * px;                                      // B*H*W < 2^31 - 512 (host check)
        const bool ok = pix < (int)P;
        const int pc = ok ? pix : 0;
        const int b = pc >> lhw, inner = pc & ((1 << lhw) - 1), oy = inner >> lw, ox = inner & ((1 << lw) - 1);
        g.ok = ok && okc;
        const int plane = b * planes + pl;
        g.addr = ((plane * e.Hout + oy) * e.Wout + ox) * e.out_feat + f;
        g.tp = skip_taps_sel(h2, w2, oy, ox, fr);
        g.sp = e.skip + (int64_t)plane * h2 * w2 * e.out_feat + f;
        return g;
    };
    auto ld = [&](rgb_f32x4& dst, const float* ptr) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory"); };
    // PARKED (the two-role kernel's memory waves, which have the registers): the per-pass geometry -- sample, tap indices and weights, pixel offset -- ONCE per tile; a
    // batch then is a handful of integer operations.  (Recomputed per batch the stage was ~1400 vector instructions per wave and tile on the FMA units the
    // multiplying wave of the same SIMD needs: the roles did not overlap.)
    constexpr bool PRE = PARKED || PRE1;            // (PRE1: the one-role kernels that stream their weights have the registers too)
    int pg_b[4], pg_pix[4], pg_i[4][4];
    float pg_w[4][4];
    bool pg_ok[4];
    if constexpr (PRE) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int pix = (int)pix0 + 4 * (q * 8 + pr);
            const bool ok = pix < (int)P;
            const int pc = ok ? pix : 0;
            const int b = pc >> lhw, inner = pc & ((1 << lhw) - 1), oy = inner >> lw, ox = inner & ((1 << lw) - 1);
            const SkipTaps tp = skip_taps_sel(h2, w2, oy, ox, fr);
            pg_b[q] = b * planes; pg_pix[q] = oy * e.Wout + ox; pg_ok[q] = ok;
            pg_i[q][0] = tp.i00; pg_i[q][1] = tp.i01; pg_i[q][2] = tp.i10; pg_i[q][3] = tp.i11;
            pg_w[q][0] = tp.w00; pg_w[q][1] = tp.w01; pg_w[q][2] = tp.w10; pg_w[q][3] = tp.w11;
        }
    }
    auto issue = [&](int bi) {
        rgb_f32x4 (&t)[4] = tA[bi % NSET];
        if constexpr (PRE) {
            const int tile = bi >> 2, q = bi & 3;
            const int o = tile * 32 + 4 * cg, oc = o < e.Cout ? o : 0;
            const int pl = oc / e.out_feat, f = oc - pl * e.out_feat;
            const float* sp = e.skip + (int64_t)(pg_b[q] + pl) * (h2 * w2) * e.out_feat + f;
#pragma unroll
            for (int j = 0; j < 4; j++) ld(t[j], sp + (int64_t)pg_i[q][j] * e.out_feat);
        } else {
            const Geo g = geo(bi >> 2, bi & 3);
            ld(t[0], g.sp + (int64_t)g.tp.i00 * e.out_feat); ld(t[1], g.sp + (int64_t)g.tp.i01 * e.out_feat);
            ld(t[2], g.sp + (int64_t)g.tp.i10 * e.out_feat); ld(t[3], g.sp + (int64_t)g.tp.i11 * e.out_feat);
        }
    };
    auto finish = [&](int bi) {
        const int tile = bi >> 2, pass = bi & 3;
        const int o = tile * 32 + 4 * cg;
        const int oc = o < e.Cout ? o : 0;
        const float4 bias4 = *(const float4*)(bias_lds + oc);
        const rgb_f32x2 b01 = {bias4.x, bias4.y}, b23 = {bias4.z, bias4.w};
        Geo g;
        if constexpr (PRE) {
            const int pl = oc / e.out_feat, f = oc - pl * e.out_feat;
            g.ok = pg_ok[pass] && o < e.Cout;
            g.addr = ((pg_b[pass] + pl) * (e.Hout * e.Wout) + pg_pix[pass]) * e.out_feat + f;
            g.tp.w00 = pg_w[pass][0]; g.tp.w01 = pg_w[pass][1]; g.tp.w10 = pg_w[pass][2]; g.tp.w11 = pg_w[pass][3];
        } else {
            int tl = tile, ps = pass;
            asm volatile("" : "+s"(tl), "+s"(ps));                               // (opaque: the geometry is recomputed here, not kept from issue())
            g = geo(tl, ps);
        }
        const float4 c4 = *(const float4*)&ct[(PARKED ? tile * (32 * CT_LD) : 0) + (pass * 8 + pr) * CT_LD + 4 * cg];
        // in flight, in issue order: this batch's 4 loads, the stores of earlier batches, the 4 loads of each batch ahead (NSET - 1 of them, fewer at the end)
        {
            constexpr int kAheadMax = NSET - 1;
            const int ahead = NB - 1 - bi < kAheadMax ? NB - 1 - bi : kAheadMax;      // compile-time: the batch loop is unrolled
            if (ahead == 5) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            else if (ahead == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (ahead == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (ahead == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        rgb_f32x2 r01, r23;
        if (!PLAIN) {                               // conv -> (bf16) -> + bias -> * gain -> clamp -> (bf16), then the skip (networks_stylegan2.py:170-171, 265-269)
            float qv[4] = {c4.x, c4.y, c4.z, c4.w};
            const float bq[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float t = qv[i];
                if (e.round_bf16) t = (float)(__bf16)t;
                t = (t + (e.round_bf16 ? (float)(__bf16)bq[i] : bq[i])) * e.gain;
                t = t < -lim ? -lim : (t > lim ? lim : t);
                qv[i] = e.round_bf16 ? (float)(__bf16)t : t;
            }
            r01 = (rgb_f32x2){qv[0], qv[1]}; r23 = (rgb_f32x2){qv[2], qv[3]};
        } else {
            r01 = (rgb_f32x2){c4.x, c4.y} + b01; r23 = (rgb_f32x2){c4.z, c4.w} + b23;
        }
        const SkipTaps& tp = g.tp;
        const rgb_f32x2 w00 = {tp.w00, tp.w00}, w01 = {tp.w01, tp.w01}, w10 = {tp.w10, tp.w10}, w11 = {tp.w11, tp.w11};
        const rgb_f32x4 ta = tA[bi % NSET][0], tb = tA[bi % NSET][1], tc = tA[bi % NSET][2], td = tA[bi % NSET][3];
        const rgb_f32x2 a01 = {ta[0], ta[1]}, a23 = {ta[2], ta[3]}, bb01 = {tb[0], tb[1]}, bb23 = {tb[2], tb[3]};
        const rgb_f32x2 cc01 = {tc[0], tc[1]}, cc23 = {tc[2], tc[3]}, d01 = {td[0], td[1]}, d23 = {td[2], td[3]};
        r01 = r01 + __builtin_elementwise_fma(w11, d01, __builtin_elementwise_fma(w10, cc01, __builtin_elementwise_fma(w01, bb01, w00 * a01)));
        r23 = r23 + __builtin_elementwise_fma(w11, d23, __builtin_elementwise_fma(w10, cc23, __builtin_elementwise_fma(w01, bb23, w00 * a23)));
        if (g.ok) *(float4*)(e.y + g.addr) = make_float4(r01.x, r01.y, r23.x, r23.y);
    };
    static_assert(NSET >= 2 && NSET <= 6, "wait counts are spelled out for up to five batches of look-ahead");
#pragma unroll
    for (int bi = B0; bi < B0 + NSET - 1 && bi < NB; bi++) issue(bi);
#pragma unroll
    for (int bi = B0; bi < NB; bi++) {
        if (!PARKED && (bi & 3) == 0) {
            // this channel tile's accumulators -> the wave's pixel-major LDS tile (wave-private: a wave barrier orders it against the previous tile's reads)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int k = bi >> 2;
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) {
                *(float4*)&ct[l32 * CT_LD + 8 * r4 + 4 * half] = make_float4(acc[k][4 * r4], acc[k][4 * r4 + 1], acc[k][4 * r4 + 2], acc[k][4 * r4 + 3]);
                acc[k][4 * r4] = 0.f; acc[k][4 * r4 + 1] = 0.f; acc[k][4 * r4 + 2] = 0.f; acc[k][4 * r4 + 3] = 0.f;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);          // (nothing of one batch moves into another: hoisted address arithmetic of later batches is what spilled the first build)
        if (bi + NSET - 1 < NB) issue(bi + NSET - 1);
        __builtin_amdgcn_sched_barrier(0);
        finish(bi);
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- ToRGB: 1x1 modulated conv without demodulation, Cout <= 96, channel-last plane output with the fused x2 skip ---------
// (networks_stylegan2.py:252-273).  12 kFLOP against 736 B per pixel at 64 channels: the layer sits on the ridge between the
// matrix cores and HBM, so the kernel is built to keep both busy -- one block = 128 consecutive pixels (1x1: no halo, tiles
// are linear over B*H*W), 64 channels per K iteration held in registers one iteration ahead, activations fetched as 16-B
// vectors along x and transposed in registers to the [pixel][4 channels] LDS layout while the style scale is applied,
// weights [chunk][Cout][4], fragments = 8-byte LDS reads with immediate offsets, no VALU in the MFMA loop.
struct RgbParams {
    const float* x;     // activations: fp32, or (XBF) bf16 -- the reduced-precision blocks, modconv_bf16.inc
    const float* wp; const float* styles;
    EpiParams e;
    int B, Cin, Cout, CoutP, HW, W;
    int64_t P;          // B*H*W
    uint32_t x_bytes, wp_bytes, st_bytes;       // sizes for the buffer descriptors
    int tpb;            // consecutive 128-pixel tiles per block (activations of tile t+1 are in flight while tile t is multiplied and stored)
    int lw, lhw;        // log2 W, log2 H*W when both are powers of two, else -1
};

#ifndef TDGP_FIR_ADJ
#define TDGP_FIR_ADJ 1      // FIR pass of the wide x2 layers: 4 adjacent rows per thread (0: two rows eight apart, the r02 form)
#endif
#ifndef TDGP_RGB_ABL
#define TDGP_RGB_ABL 0     // 16: per-phase cycle counts of one wave, printed; 32: skip taps from LDS (timing experiments only)
#endif
// RESIDENT: Cin <= 64 -- the whole weight matrix is one LDS stage, loaded once per block, and the block walks `tpb` consecutive
// tiles with the activations of tile t+1 in flight while tile t is multiplied and stored.
// FAST: power-of-two image, no clamp, gain 1 (the generator's ToRGB layers) -- the trimmed output stage.
template <int MT, bool RESIDENT, bool FAST, bool XBF = false>
__global__ __launch_bounds__(256, 2) void torgb_mfma_kernel(RgbParams p) {
    constexpr int BM = 32 * MT, BN = 128, NCH = 16;                 // 16 packed chunks = 64 channels per iteration
    constexpr int AS_SZ = NCH * BM * 4, XS_SZ = NCH * BN * 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Xs = smem + AS_SZ;
    float* side = smem + AS_SZ + XS_SZ;                             // [BM] bias
    const int tid = threadIdx.x, l = tid & 63, wv = TDGP_WAVE_INDEX(tid), l32 = l & 31, half = l >> 5;     // wave index: scalar
    const int64_t ntiles = (p.P + BN - 1) / BN;
#ifndef TDGP_RGB_XCD_COMPACT
#define TDGP_RGB_XCD_COMPACT 1          // 0: block b takes tiles b * tpb .. (A/B builds)
#endif
    // Blocks b, b + 8, ... share an XCD (round-robin dispatch: a locality hint, not a contract).  XCD-compact order: XCD x walks the x-th EIGHTH of the
    // tiles, so blocks that run next to each other in time on one XCD -- one L2 -- take vertically adjacent rows of the image: the two / three skip rows
    // an output row pair blends are then fetched into that L2 once instead of once per XCD that happens to hold a neighbour (round 5 counters: the
    // skip image came through the memory side 4.4 times, fetch 1.93 x the algorithmic bytes).  Same tiles, same arithmetic per tile.
    const int64_t nb_ = gridDim.x;
    const int64_t lb_ = (TDGP_RGB_XCD_COMPACT && (nb_ & 7) == 0 && nb_ >= 64) ? (int64_t)(blockIdx.x & 7) * (nb_ >> 3) + (blockIdx.x >> 3) : (int64_t)blockIdx.x;
    const int64_t t_begin = lb_ * p.tpb, t_end = min(t_begin + p.tpb, ntiles);
#if TDGP_RGB_ABL & 16
    long long tq[5] = {0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define TR(i) { const long long tn_ = __builtin_readcyclecounter(); tq[i] += tn_ - tprev; tprev = tn_; }
#else
#define TR(i)
#endif

    for (int i = tid; i < BM; i += 256) side[i] = (i < p.Cout && p.e.bias) ? p.e.bias[i] : 0.f;

    // two 4-channel x 4-pixel micro-tiles per thread and iteration (byte offsets for the buffer loads; kOOB -> 0)
    uint32_t x_vo[2], s_vo[2];
    const int cq = tid >> 5, pq = tid & 31;                        // micro-tile k: channels 4*(cq + 8k) .. +3 of the iteration, pixels 4*pq .. +3
    auto tile_offsets = [&](int64_t t) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int64_t pix = t * BN + 4 * pq;
            x_vo[k] = kOOB; s_vo[k] = kOOB;
            if (pix < p.P) {
                const int b = (int)(pix / p.HW), inner = (int)(pix - (int64_t)b * p.HW);
                const int sb = b * p.Cin + 4 * (cq + 8 * k);
                s_vo[k] = (uint32_t)sb * 4u;
                x_vo[k] = (uint32_t)(sb * p.HW + inner) * (XBF ? 2u : 4u);
            }
        }
    };
    const uint32_t hw4 = (uint32_t)p.HW * (XBF ? 2u : 4u);
    const bool cin4 = (p.Cin & 3) == 0;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes), rw = make_rsrc(p.wp, p.wp_bytes), rs = make_rsrc(p.styles ? p.styles : p.x, p.styles ? p.st_bytes : 0);
    auto load_x4 = [&](uint32_t vo, uint32_t so) -> float4 {        // 4 consecutive pixels of one channel
        if constexpr (XBF) {
            const rgb_u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(rx, vo, so, 0);          // 4 bf16
            return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
        } else {
            return buf_load4(rx, vo, so);
        }
    };

    constexpr int NA = (NCH * BM + 255) / 256;
    uint32_t a_vo[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int e = tid + i * 256;
        a_vo[i] = kOOB;
        if (e < NCH * BM) {
            const int chunk = e / BM, col = e % BM;
            if (col < p.CoutP) a_vo[i] = (uint32_t)(chunk * p.CoutP + col) * 16u;
        }
    }
    const uint32_t a_gstride4 = (uint32_t)(NCH * p.CoutP) * 16u;

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[m][r] = 0.f;

    const int niter = (p.Cin + 63) >> 6;
    constexpr bool a_resident = RESIDENT;           // all weights fit one stage: loaded for the block's first tile only
    float4 a_reg[NA], xr[2][4], sr[2];
    auto load_stage = [&](int it, bool with_a) {
        if (with_a) {
            const uint32_t a_so = (uint32_t)it * a_gstride4;
#pragma unroll
            for (int i = 0; i < NA; i++) a_reg[i] = buf_load4(rw, a_vo[i], a_so);
        }
        const uint32_t c0 = (uint32_t)it * 64u;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            // micro-tile channels beyond Cin: the x loads fall into the next sample (finite) or beyond the buffer (0) and meet zero weights
#pragma unroll
            for (int j = 0; j < 4; j++) xr[k][j] = load_x4(x_vo[k], (c0 + j) * hw4);
            if (p.styles) {
                if (cin4) sr[k] = buf_load4(rs, s_vo[k], c0 * 4u);
                else sr[k] = make_float4(buf_load1(rs, s_vo[k], c0 * 4u), buf_load1(rs, s_vo[k], c0 * 4u + 4u), buf_load1(rs, s_vo[k], c0 * 4u + 8u), buf_load1(rs, s_vo[k], c0 * 4u + 12u));
            } else {
                sr[k] = make_float4(1.f, 1.f, 1.f, 1.f);
            }
        }
    };
    auto store_stage = [&](bool with_a) {
        if (with_a) {
#pragma unroll
            for (int i = 0; i < NA; i++)
                if (tid + i * 256 < NCH * BM) *(float4*)(As + (tid + i * 256) * 4) = a_reg[i];
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            // 4x4 register transpose: xr[k][j] = channel j over 4 pixels  ->  per pixel, 4 channels (x style)
            // Pixel 4*pq + j lands in slot j*32 + pq: consecutive lanes write consecutive 16-B slots (conflict-free), and wave j's
            // MFMA column l32 (slot j*32 + l32) is pixel 4*l32 + j -- a permutation the channel-last epilogue does not care about.
            float* d = Xs + (((cq + 8 * k) * BN) + pq) * 4;
            *(float4*)(d + 0 * 128) = make_float4(xr[k][0].x * sr[k].x, xr[k][1].x * sr[k].y, xr[k][2].x * sr[k].z, xr[k][3].x * sr[k].w);
            *(float4*)(d + 1 * 128) = make_float4(xr[k][0].y * sr[k].x, xr[k][1].y * sr[k].y, xr[k][2].y * sr[k].z, xr[k][3].y * sr[k].w);
            *(float4*)(d + 2 * 128) = make_float4(xr[k][0].z * sr[k].x, xr[k][1].z * sr[k].y, xr[k][2].z * sr[k].z, xr[k][3].z * sr[k].w);
            *(float4*)(d + 3 * 128) = make_float4(xr[k][0].w * sr[k].x, xr[k][1].w * sr[k].y, xr[k][2].w * sr[k].z, xr[k][3].w * sr[k].w);
        }
    };

    const float* Al = As + l32 * 4 + 2 * half;
    const float* Xl = Xs + (wv * 32 + l32) * 4 + 2 * half;
    float* ct = Xs + wv * (32 * CT_LD);
    const bool has_skip = p.e.skip != nullptr;
    if (t_begin >= t_end) return;
    tile_offsets(t_begin);
    load_stage(0, true);
    if (RESIDENT) {                                 // weights: once per block, outside the tile loop (nothing of it stays live in there)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NA; i++)
            if (tid + i * 256 < NCH * BM) *(float4*)(As + (tid + i * 256) * 4) = a_reg[i];
    }
    TR(0)
    for (int64_t t = t_begin; t < t_end; t++) {
        for (int it = 0; it < niter; it++) {
            __syncthreads();                        // the previous stage's fragments (and the previous tile's parked outputs) have been read
            store_stage(!RESIDENT);
            __syncthreads();
            TR(1)
            // the next stage -- of this tile, or the first one of the next tile -- is in flight during the multiply and the output stage
            if (it + 1 < niter) load_stage(it + 1, true);
            TR(2)
            f32x2 fa[2][MT], fb[2];
            auto load_frag = [&](int buf, int ch) {
#pragma unroll
                for (int m = 0; m < MT; m++) fa[buf][m] = *(const f32x2*)(Al + (ch * BM + m * 32) * 4);
                fb[buf] = *(const f32x2*)(Xl + ch * BN * 4);
            };
            load_frag(0, 0);
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) {
                const int cb = ch & 1;
                if (ch + 1 < NCH) load_frag(cb ^ 1, ch + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 2; kk++)
#pragma unroll
                    for (int m = 0; m < MT; m++) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cb][m][kk], fb[cb][kk], acc[m], 0, 0, 0);
            }
            TR(3)
        }
        // the next tile's activations are put in flight here, after the multiply (issued before it, the loads land in registers
        // the fragment reads want and the compiler parks them with a full wait) and ahead of the output stage that hides them
        // (unconditional -- the block's last tile re-reads itself: a load under a condition makes the compiler wait for it at the join)
        if (RESIDENT) { tile_offsets(t + 1 < t_end ? t + 1 : t); load_stage(0, false); }
        __syncthreads();

        // ---- output stage: bias + FIR-upsampled skip, channel-last float4 stores (epilogue_tile<true>, pixel-major LDS tile) ----
        const int64_t pix = t * BN + 4 * l32 + wv;
        int pok = 0, pb = 0, poy = 0, pox = 0;
        constexpr bool LANE_GEOM = !FAST || RESIDENT;   // this lane's pixel, handed to the output stage by cross-lane reads (the
        if (LANE_GEOM) {                                 // shift form inside the output stage costs the multi-tile kernel ~25 registers)
            pok = pix < p.P ? 1 : 0;
            const int64_t pc = pok ? pix : 0;
            if (FAST) {
                pb = (int)(pc >> p.lhw);
                const int inner = (int)pc & ((1 << p.lhw) - 1);
                poy = inner >> p.lw; pox = inner & ((1 << p.lw) - 1);
            } else {
                pb = (int)(pc / p.HW);
                const int inner = (int)(pc - (int64_t)pb * p.HW);
                poy = inner / p.W; pox = inner - poy * p.W;
            }
        }
        const int64_t pix0 = t * BN + wv;
#ifndef TDGP_RGB_PIPELINED
#define TDGP_RGB_PIPELINED 1           // 0: the stage per channel tile (rounds 2-6; A/B builds, same bits)
#endif
        if (TDGP_RGB_PIPELINED && FAST && has_skip) {
            rgb_output_skip_pipelined<MT, true, false, RESIDENT ? 3 : TDGP_RGB_NSET_STREAMED, !RESIDENT && TDGP_RGB_PRE_STREAMED>(p.e, side, ct, acc, pix0, p.P, p.lw, p.lhw);
        } else
#pragma unroll 1
        for (int tile = 0; tile < MT; tile++) {
#pragma unroll
            for (int k = 0; k < MT; k++) {
                if (tile == k) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; r4++) {
                        *(float4*)&ct[l32 * CT_LD + 8 * r4 + 4 * half] = make_float4(acc[k][4 * r4], acc[k][4 * r4 + 1], acc[k][4 * r4 + 2], acc[k][4 * r4 + 3]);
                        acc[k][4 * r4] = 0.f; acc[k][4 * r4 + 1] = 0.f; acc[k][4 * r4 + 2] = 0.f; acc[k][4 * r4 + 3] = 0.f;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            constexpr int G = 4;
            if constexpr (!(TDGP_RGB_PIPELINED && FAST)) {
                if (has_skip) rgb_output_tile<true, FAST, !LANE_GEOM, G>(p.e, side, ct, tile * 32, pb, poy, pox, pok, pix0, p.P, p.lw, p.lhw);
                else rgb_output_tile<false, FAST, !LANE_GEOM, G>(p.e, side, ct, tile * 32, pb, poy, pox, pok, pix0, p.P, p.lw, p.lhw);
            } else {
                rgb_output_tile<false, FAST, !LANE_GEOM, G>(p.e, side, ct, tile * 32, pb, poy, pox, pok, pix0, p.P, p.lw, p.lhw);      // (FAST with a skip took the pipelined stage)
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        TR(4)
        if (!RESIDENT) break;                       // one tile per block (tpb = 1): nothing stays live across the output stage
    }
#if TDGP_RGB_ABL & 16
    if (tid == 0 && (blockIdx.x == 5 || blockIdx.x == 1000))
        printf("torgb blk %d tiles %d iters %d: prologue+issue %lld wait+store %lld issue-next %lld mma %lld epilogue %lld\n", (int)blockIdx.x, (int)(t_end - t_begin), niter, tq[0], tq[1], tq[2], tq[3], tq[4]);
#endif
#undef TR
}

// ---- ToRGB, two roles per block (round 6): Cin <= 64 (weights resident), power-of-two image, gain 1, no clamp, with a skip -- the 512^2 layer of C3 / C4 ----
// torgb_mfma_kernel runs a tile's three resources one after the other -- activation staging (global -> registers -> style, transpose -> LDS), 96 MFMAs per wave, the
// output stage's 48 tap loads + 12 stores per wave -- and two resident blocks per CU overlap them only partly (ablations of round 2: multiply + staging alone 0.56 ms of
// the layer's 1.05, the MFMAs alone 0.33).  Here a block is 8 waves, one per CU: waves 0..3 MULTIPLY tile t (fragments from LDS, nothing else), waves 4..7 do everything
// of the output stage of tile t - 1 -- 48 tap loads, 12 stores per wave, from the LDS tile the multipliers parked, with the per-pass geometry computed once per tile --; the
// multipliers also carry tile t + 1's activations (requested before the multiply, staged into the other X buffer behind it).  One of each role per SIMD, two block barriers per tile.  Same fragments, same K order, same fma chains as torgb_mfma_kernel: same bits.
#ifndef TDGP_RGB_WS_SHARE
#define TDGP_RGB_WS_SHARE 0            // batches of a tile's output stage (of 4 MT) the MULTIPLYING waves take, in front of their multiply.  Measured (512^2, B = 16): 0 -> 1.00-1.01 ms, 2 -> 1.03-1.04, 3 -> 1.05-1.06: the memory waves keep all of it
#endif
#ifndef TDGP_RGB_WS_ABL
#define TDGP_RGB_WS_ABL 0              // timing experiments (wrong results): 1 = no output stage, 2 = no multiply
#endif
template <int MT>
__global__ __launch_bounds__(512, 1) void torgb_ws_kernel(RgbParams p) {
    constexpr int BM = 32 * MT, BN = 128, NCH = 16;
    constexpr int AS_SZ = NCH * BM * 4, XS_SZ = NCH * BN * 4, CT_SZ = 4 * MT * 32 * CT_LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Xs = smem + AS_SZ;                                       // two buffers
    float* Ct = smem + AS_SZ + 2 * XS_SZ;                           // [4 pixel groups][MT channel tiles][32 pixels][CT_LD]
    float* side = Ct + CT_SZ;                                       // [BM] bias
    const int tid = threadIdx.x, l = tid & 63, wv = TDGP_WAVE_INDEX(tid), role = wv >> 2, wj = wv & 3, l32 = l & 31, half = l >> 5;
    const int mt = tid & 255;                                       // thread index inside the role
    const int64_t ntiles = (p.P + BN - 1) / BN;
    const int64_t nb_ = gridDim.x;
    const int64_t lb_ = ((nb_ & 7) == 0 && nb_ >= 64) ? (int64_t)(blockIdx.x & 7) * (nb_ >> 3) + (blockIdx.x >> 3) : (int64_t)blockIdx.x;       // XCD-compact, as torgb_mfma_kernel
    const int64_t t_begin = lb_ * p.tpb, t_end = min(t_begin + p.tpb, ntiles);
    if (t_begin >= t_end) return;
    for (int i = tid; i < BM; i += 512) side[i] = (i < p.Cout && p.e.bias) ? p.e.bias[i] : 0.f;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes), rw = make_rsrc(p.wp, p.wp_bytes), rs = make_rsrc(p.styles ? p.styles : p.x, p.styles ? p.st_bytes : 0);
    // weights: one stage, once per block (every thread of both roles)
    for (int e = tid; e < NCH * BM; e += 512) {
        const int chunk = e / BM, col = e % BM;
        *(float4*)(As + e * 4) = col < p.CoutP ? buf_load4(rw, (uint32_t)(chunk * p.CoutP + col) * 16u, 0u) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // memory role: two 4-channel x 4-pixel micro-tiles per thread and tile
    uint32_t x_vo[2], s_vo[2];
    const int cq = mt >> 5, pq = mt & 31;
    auto tile_offsets = [&](int64_t t) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int64_t pix = t * BN + 4 * pq;
            x_vo[k] = kOOB; s_vo[k] = kOOB;
            if (pix < p.P) {
                const int b = (int)(pix >> p.lhw), inner = (int)pix & ((1 << p.lhw) - 1);
                const int sb = b * p.Cin + 4 * (cq + 8 * k);
                s_vo[k] = (uint32_t)sb * 4u;
                x_vo[k] = (uint32_t)(sb * p.HW + inner) * 4u;
            }
        }
    };
    const uint32_t hw4 = (uint32_t)p.HW * 4u;
    const bool cin4 = (p.Cin & 3) == 0;
    float4 xr[2][4], sr[2];
    auto load_x = [&]() {
#pragma unroll
        for (int k = 0; k < 2; k++) {
#pragma unroll
            for (int j = 0; j < 4; j++) xr[k][j] = buf_load4(rx, x_vo[k], (uint32_t)j * hw4);
            if (p.styles) {
                if (cin4) sr[k] = buf_load4(rs, s_vo[k], 0u);
                else sr[k] = make_float4(buf_load1(rs, s_vo[k], 0u), buf_load1(rs, s_vo[k], 4u), buf_load1(rs, s_vo[k], 8u), buf_load1(rs, s_vo[k], 12u));
            } else {
                sr[k] = make_float4(1.f, 1.f, 1.f, 1.f);
            }
        }
    };
    auto store_x = [&](int buf) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            float* d = Xs + buf * XS_SZ + (((cq + 8 * k) * BN) + pq) * 4;        // (pixel 4 pq + j -> slot j * 32 + pq: torgb_mfma_kernel::store_stage)
            *(float4*)(d + 0 * 128) = make_float4(xr[k][0].x * sr[k].x, xr[k][1].x * sr[k].y, xr[k][2].x * sr[k].z, xr[k][3].x * sr[k].w);
            *(float4*)(d + 1 * 128) = make_float4(xr[k][0].y * sr[k].x, xr[k][1].y * sr[k].y, xr[k][2].y * sr[k].z, xr[k][3].y * sr[k].w);
            *(float4*)(d + 2 * 128) = make_float4(xr[k][0].z * sr[k].x, xr[k][1].z * sr[k].y, xr[k][2].z * sr[k].z, xr[k][3].z * sr[k].w);
            *(float4*)(d + 3 * 128) = make_float4(xr[k][0].w * sr[k].x, xr[k][1].w * sr[k].y, xr[k][2].w * sr[k].z, xr[k][3].w * sr[k].w);
        }
    };
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[m][r] = 0.f;
    float* ctw = Ct + wj * (MT * 32 * CT_LD);
    // prologue: tile t_begin staged (by the multiplying waves: they carry the activations, the memory waves carry the taps)
    if (role == 0) {
        tile_offsets(t_begin);
        load_x();
        store_x(0);
    }
    __syncthreads();
    const int n = (int)(t_end - t_begin);
    for (int i = 0; i < n; i++) {
        const int64_t t = t_begin + i;
        constexpr int NBT = 4 * MT, SPLIT = NBT - (TDGP_RGB_WS_SHARE < MT ? TDGP_RGB_WS_SHARE : MT);       // (at most a quarter of the batches)
        if (role == 0) {
            // the last batches of tile t - 1's output stage (the memory waves' 0.76 ms against the multipliers' 0.58: the multipliers wait at the barrier otherwise)
            if (SPLIT < NBT && i > 0 && !(TDGP_RGB_WS_ABL & 1)) {
                f32x16 none[MT];
                rgb_output_skip_pipelined<MT, true, true, 3, false, SPLIT, NBT>(p.e, side, ctw, none, (t - 1) * BN + wj, p.P, p.lw, p.lhw);
            }
            // tile t + 1's activations travel under this tile's multiply (the block's last tile re-reads itself: no load under a condition)
            tile_offsets(t + 1 < t_end ? t + 1 : t);
            load_x();
            const float* Al = As + l32 * 4 + 2 * half;
            const float* Xl = Xs + (i & 1) * XS_SZ + (wj * 32 + l32) * 4 + 2 * half;
            f32x2 fa[2][MT], fb[2];
            auto load_frag = [&](int buf, int ch) {
#pragma unroll
                for (int m = 0; m < MT; m++) fa[buf][m] = *(const f32x2*)(Al + (ch * BM + m * 32) * 4);
                fb[buf] = *(const f32x2*)(Xl + ch * BN * 4);
            };
            load_frag(0, 0);
#pragma unroll
            for (int ch = 0; ch < ((TDGP_RGB_WS_ABL & 2) ? 1 : NCH); ch++) {
                const int cb = ch & 1;
                if (ch + 1 < NCH) load_frag(cb ^ 1, ch + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 2; kk++)
#pragma unroll
                    for (int m = 0; m < MT; m++) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cb][m][kk], fb[cb][kk], acc[m], 0, 0, 0);
            }
            store_x((i + 1) & 1);                   // (nobody reads that buffer before the barriers below: tile t - 1's multiply ended an iteration ago)
        } else {
            if (i > 0 && !(TDGP_RGB_WS_ABL & 1)) {
                f32x16 none[MT];
                rgb_output_skip_pipelined<MT, true, true, 3, false, 0, SPLIT>(p.e, side, ctw, none, (t - 1) * BN + wj, p.P, p.lw, p.lhw);
            }
        }
        __syncthreads();                            // the multipliers are through with X[i & 1] and have written X[(i + 1) & 1]; tile t - 1's parked outputs have been read
        if (role == 0) {
#pragma unroll
            for (int k = 0; k < MT; k++)
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) {
                    *(float4*)&ctw[k * (32 * CT_LD) + l32 * CT_LD + 8 * r4 + 4 * half] = make_float4(acc[k][4 * r4], acc[k][4 * r4 + 1], acc[k][4 * r4 + 2], acc[k][4 * r4 + 3]);
                    acc[k][4 * r4] = 0.f; acc[k][4 * r4 + 1] = 0.f; acc[k][4 * r4 + 2] = 0.f; acc[k][4 * r4 + 3] = 0.f;
                }
        }
        __syncthreads();                            // tile t's outputs are parked
    }
    {
        constexpr int NBT = 4 * MT, SPLIT = NBT - (TDGP_RGB_WS_SHARE < MT ? TDGP_RGB_WS_SHARE : MT);       // (at most a quarter of the batches)
        f32x16 none[MT];
        if (role == 1) rgb_output_skip_pipelined<MT, true, true, 3, false, 0, SPLIT>(p.e, side, ctw, none, (t_end - 1) * BN + wj, p.P, p.lw, p.lhw);
        else if (SPLIT < NBT) rgb_output_skip_pipelined<MT, true, true, 3, false, SPLIT, NBT>(p.e, side, ctw, none, (t_end - 1) * BN + wj, p.P, p.lw, p.lhw);
    }
}

// Split-K reduction: y = epilogue(sum_ks partial[ks]) ; one thread per output element, ks summed in order (deterministic).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int ksplit, EpiParams e) {
    const int64_t slice = (int64_t)e.B * e.Cout * e.Hout * e.Wout;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < slice; i += (int64_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int k = 0; k < ksplit; k++) v += partial[k * slice + i];
        const int ox = (int)(i % e.Wout);
        int64_t r = i / e.Wout;
        const int oy = (int)(r % e.Hout); r /= e.Hout;
        const int o = (int)(r % e.Cout);
        const int b = (int)(r / e.Cout);
        epilogue_store(e, v, b, o, oy, ox);
    }
}

// -------------------------------------------------------------------------------------------------
// FIR (4x4, pad 1, gain folded into the taps) + demod + noise + bias + activation on the transposed-conv intermediate
// (upconv_mfma_kernel's parity-planar Z, split-K slices summed on the fly) -> y [B,C,2H,2W].
// (conv2d_resample.py:126 + networks_stylegan2.py:86-87,144)
// Block = one 32x64 output tile of one (b,c) plane: the (32+3) x (64+8) input window is staged in LDS with aligned 16-byte
// loads, each thread then produces 2 x 4 consecutive outputs from 4x7 windows.  HBM traffic = Z read once + y written once.
// -------------------------------------------------------------------------------------------------
struct FirParams {
    const float* z; const float* dcoef; const float* noise; const float* bias; float* y;
    uint16_t* y16;      // bf16 output (fir_act_kernel<.., true>): the reduced-precision blocks, modconv_bf16.inc
    int64_t noise_bstride, zslice;
    float fir[16];
    int B, C, ZROWS, P2, GS2, ksplit, OH, OW;      // ZROWS = 2H+2 rows of pitch P2 = 2*G1 (a multiple of 4), alternating between the parity planes
    int act; float alpha, gain, clamp;
    int noise_vec;      // host: noise may be fetched as 16-byte vectors (pointer 16-byte aligned, batch stride a multiple of 4 floats)
};

// Tile FIR_TH x FIR_TW outputs per block: 32 x 64, or 16 x 128 for wide images (longer contiguous runs per row: 544-B reads,
// 512-B writes instead of 288 / 256).
typedef float fir_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 fir_b2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fir_round_bf16(float v) { return (float)(__bf16)v; }
__device__ __forceinline__ uint32_t fir_pack_bf16(float a, float b) {
    const fir_b2 h = __builtin_convertvector((fir_f2){a, b}, fir_b2);
    uint32_t u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}
// YBF: bf16 output with the reference's rounding points behind the (fp32, unrounded) transposed-convolution intermediate: FIR output,
// + noise, bias (itself rounded) / activation / gain / clamp.
// LRELU: the generator's form (leaky ReLU with 0 <= alpha <= 1; the clamp, if any, one v_med3) decided at compile time -- the pass is instruction-bound,
// and the run-time activation switch + clamp test per output were a fifth of its output stage.
template <int FIR_TH, int FIR_TW, bool YBF = false, bool LRELU = false, bool ADJ = false>
__global__ __launch_bounds__(256) void fir_act_kernel(FirParams p) {
    constexpr int CW = FIR_TW / 4, RPP = 256 / CW;                 // threads per output row, rows per pass
    constexpr int ZP = FIR_TW + 8;                  // window columns ox0-4 .. ox0+67, fetched as aligned 16-B vectors (P2 % 4 == 0)
    __shared__ __attribute__((aligned(16))) float zt[(FIR_TH + 3) * ZP];
    const uint32_t tilesX = (uint32_t)(p.OW + FIR_TW - 1) / FIR_TW, tilesY = (uint32_t)(p.OH + FIR_TH - 1) / FIR_TH;
    const uint32_t ntiles = (uint32_t)p.B * p.C * tilesY * tilesX;          // < 2^31 (host check)
    // tile index -> (sample, channel, tile row, tile column) in 32-bit unsigned arithmetic on the scalar unit: the index is wave-uniform, and as
    // 64-bit divisions this decomposition was the largest single item of the pass's instruction count (SQ counters: 48 vector instructions per output)
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint32_t tu = __builtin_amdgcn_readfirstlane(t);
        const int tx = (int)(tu % tilesX);
        uint32_t r = tu / tilesX;
        const int ty = (int)(r % tilesY); r /= tilesY;
        const int c = (int)(r % (uint32_t)p.C);
        const int b = (int)(r / (uint32_t)p.C);
        const int oy0 = ty * FIR_TH, ox0 = tx * FIR_TW;
        const float* zp = p.z + ((int64_t)b * p.C + c) * 2 * p.GS2;
        const int nrows = min(FIR_TH, p.OH - oy0) + 3;
        // side inputs of this thread's outputs, requested together with the window so that their latency is not paid after the barrier
        const int lx = (threadIdx.x % CW) * 4;
        const float d = p.dcoef ? p.dcoef[b * p.C + c] : 1.f;
        const float bv = p.bias ? (YBF ? fir_round_bf16(p.bias[c]) : p.bias[c]) : 0.f;
        const bool nz_vec = p.noise_vec && ox0 + lx + 3 < p.OW && (p.OW & 3) == 0;
        float4 nzv[FIR_TH / RPP];
#pragma unroll
        for (int hrow = 0; hrow < FIR_TH / RPP; hrow++) {
            const int oy = oy0 + (ADJ ? (threadIdx.x / CW) * (FIR_TH / RPP) + hrow : threadIdx.x / CW + hrow * RPP);
            nzv[hrow] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (nz_vec && oy < p.OH) nzv[hrow] = *(const float4*)(p.noise + b * p.noise_bstride + (int64_t)oy * p.OW + ox0 + lx);
        }
        __syncthreads();
        if (p.ksplit == 1 && !TDGP_AB_FIR_SERIAL) {
            // no split-K slices to add: all (at most 3) window vectors of a thread are put in flight before the first is written to
            // LDS.  One load per thread at a time left 8 blocks x 4 KB in flight per CU -- by Little's law ~4 TB/s, which is where
            // the kernel sat.
            constexpr int NV = ((FIR_TH + 3) * (ZP / 4) + 255) / 256;
            float4 wv4[NV];
#pragma unroll
            for (int u = 0; u < NV; u++) {
                const int i = threadIdx.x + u * 256;
                const int ry = i / (ZP / 4), j = i % (ZP / 4);
                const int zy = oy0 - 1 + ry, zx = ox0 - 4 + 4 * j;
                wv4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < nrows * (ZP / 4) && zy >= 0 && zy < p.ZROWS && zx >= 0 && zx < p.P2)
                    wv4[u] = *(const float4*)(zp + (zy & 1) * p.GS2 + (zy >> 1) * p.P2 + zx);
            }
#pragma unroll
            for (int u = 0; u < NV; u++) {
                const int i = threadIdx.x + u * 256;
                if (i < nrows * (ZP / 4)) *(float4*)&zt[(i / (ZP / 4)) * ZP + 4 * (i % (ZP / 4))] = wv4[u];
            }
        } else
        for (int i = threadIdx.x; i < nrows * (ZP / 4); i += 256) {
            const int ry = i / (ZP / 4), j = i % (ZP / 4);
            const int zy = oy0 - 1 + ry, zx = ox0 - 4 + 4 * j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (zy >= 0 && zy < p.ZROWS && zx >= 0 && zx < p.P2) {
                const float* q = zp + (zy & 1) * p.GS2 + (zy >> 1) * p.P2 + zx;
                v = *(const float4*)q;
                for (int k0 = 1; k0 < p.ksplit; k0 += 4) {      // split-K slices, summed in order (deterministic); 4 loads in flight
                    float4 w[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) w[k] = (k0 + k < p.ksplit) ? *(const float4*)(q + (k0 + k) * p.zslice) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int k = 0; k < 4; k++) { v.x += w[k].x; v.y += w[k].y; v.z += w[k].z; v.w += w[k].w; }
                }
            }
            *(float4*)&zt[ry * ZP + 4 * j] = v;
        }
        __syncthreads();
        // ADJ: a thread's rows are ADJACENT (a (RPT+3) x 7 window serves RPT x 4 outputs: 3 (RPT + 3) window reads instead of 12 RPT), each
        // output still summing its 16 taps in the order ky, kx of the other form
        constexpr int RPT = FIR_TH / RPP;
        float accs[ADJ ? RPT : 1][4];
        if constexpr (ADJ) {
#pragma unroll
            for (int rr = 0; rr < RPT; rr++)
#pragma unroll
                for (int o = 0; o < 4; o++) accs[rr][o] = 0.f;
            const int ly0 = (threadIdx.x / CW) * RPT;
#pragma unroll
            for (int wy = 0; wy < RPT + 3; wy++) {
                const float4 wa = *(const float4*)&zt[(ly0 + wy) * ZP + lx], wb = *(const float4*)&zt[(ly0 + wy) * ZP + lx + 4],
                             wc = *(const float4*)&zt[(ly0 + wy) * ZP + lx + 8];
                const float win[7] = {wa.w, wb.x, wb.y, wb.z, wb.w, wc.x, wc.y};
#pragma unroll
                for (int rr = 0; rr < RPT; rr++) {
                    const int ky = wy - rr;
                    if (ky >= 0 && ky < 4) {
#pragma unroll
                        for (int o = 0; o < 4; o++)
#pragma unroll
                            for (int kx = 0; kx < 4; kx++) accs[rr][o] = fmaf_(p.fir[ky * 4 + kx], win[o + kx], accs[rr][o]);
                    }
                }
            }
        }
#pragma unroll
        for (int hrow = 0; hrow < FIR_TH / RPP; hrow++) {
            const int ly = ADJ ? (threadIdx.x / CW) * RPT + hrow : threadIdx.x / CW + hrow * RPP;
            const int oy = oy0 + ly;
            if (oy < p.OH && ox0 + lx < p.OW) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (ADJ) {
#pragma unroll
                    for (int o = 0; o < 4; o++) acc[o] = accs[hrow][o];
                } else
#pragma unroll
                for (int ky = 0; ky < 4; ky++) {
                    // window columns lx+3 .. lx+9 out of three aligned 16-B LDS reads (lx .. lx+11): scalar reads at a 4-float lane
                    // stride hit 8 of the 32 banks (4-way conflict, 7 reads); the vector reads are conflict-free
                    const float4 wa = *(const float4*)&zt[(ly + ky) * ZP + lx], wb = *(const float4*)&zt[(ly + ky) * ZP + lx + 4],
                                 wc = *(const float4*)&zt[(ly + ky) * ZP + lx + 8];
                    const float win[7] = {wa.w, wb.x, wb.y, wb.z, wb.w, wc.x, wc.y};
#pragma unroll
                    for (int o = 0; o < 4; o++)
#pragma unroll
                        for (int kx = 0; kx < 4; kx++) acc[o] = fmaf_(p.fir[ky * 4 + kx], win[o + kx], acc[o]);
                }
                const int64_t yoff = (((int64_t)b * p.C + c) * p.OH + oy) * p.OW + ox0 + lx;
                float out[4];
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    float v = acc[o] * d;
                    if (YBF) v = fir_round_bf16(v);
                    if (p.noise && ox0 + lx + o < p.OW) {
                        const float nq[4] = {nzv[hrow].x, nzv[hrow].y, nzv[hrow].z, nzv[hrow].w};
                        v = v + (nz_vec ? nq[o] : p.noise[b * p.noise_bstride + (int64_t)oy * p.OW + ox0 + lx + o]);
                        if (YBF) v = fir_round_bf16(v);
                    }
                    v = v + bv;
                    if constexpr (LRELU) {
                        v = __builtin_fmaxf(v, v * p.alpha) * p.gain;        // == (v > 0 ? v : alpha v) * gain for 0 <= alpha <= 1, sign of zero included
                        if (p.clamp >= 0.f) v = __builtin_amdgcn_fmed3f(v, -p.clamp, p.clamp);
                    }
                    else {
                        v = act_apply(v, p.act, p.alpha) * p.gain;
                        if (p.clamp >= 0.f) v = v < -p.clamp ? -p.clamp : (v > p.clamp ? p.clamp : v);
                    }
                    out[o] = v;
                }
                if constexpr (YBF) {
                    uint16_t* yp = p.y16 + yoff;
                    if (ox0 + lx + 3 < p.OW && (p.OW & 3) == 0) *(uint2*)yp = make_uint2(fir_pack_bf16(out[0], out[1]), fir_pack_bf16(out[2], out[3]));
                    else
                        for (int o = 0; o < 4 && ox0 + lx + o < p.OW; o++) yp[o] = (uint16_t)(fir_pack_bf16(out[o], 0.f) & 0xffffu);
                } else {
                    float* yp = p.y + yoff;
                    if (ox0 + lx + 3 < p.OW && (p.OW & 3) == 0) *(float4*)yp = make_float4(out[0], out[1], out[2], out[3]);
                    else
                        for (int o = 0; o < 4 && ox0 + lx + o < p.OW; o++) yp[o] = out[o];
                }
            }
        }
    }
}

// Plain x2 transposed convolution: the parity-planar intermediate Z of upconv_mfma_kernel (split-K slices summed in order) gathered
// into the NCHW tensor [B,C,2H+1,2W+1] -- conv_transpose2d(stride 2) without the FIR pass, for the input gradient of the stride-2
// convolutions (conv2d_gradfix.py:126-129).
__global__ __launch_bounds__(256) void z_gather_kernel(const float* __restrict__ z, float* __restrict__ y, int64_t rows, int OHt, int OWt, int P2, int GS2,
                                                       int ksplit, int64_t zslice) {
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const int i = (int)(r % OHt);
        const int64_t bc = r / OHt;
        const float* zp = z + bc * 2 * GS2 + (i & 1) * GS2 + (int64_t)(i >> 1) * P2;
        float* yp = y + r * OWt;
        for (int j = threadIdx.x; j < OWt; j += blockDim.x) {
            float v = zp[j];
            for (int k = 1; k < ksplit; k++) v += zp[k * zslice + j];
            yp[j] = v;
        }
    }
}

#include "modconv_bf16.inc"
#include "modconv_wino.inc"
#include "modconv_wino4.inc"
#include "modconv_wino4f.inc"

// d[b,o] = rsqrt(sum_c s[b,c]^2 * wsq[c][o] + 1e-8)      (networks_stylegan2.py:62)
// block (64 out-channels x 16 channel slices): coalesced wsq rows, 16-way split of the Cin loop, LDS tree at the end.
__global__ __launch_bounds__(1024) void demod_kernel(const float* __restrict__ styles, const float* __restrict__ wsq, float* __restrict__ d, int B,
                                                    int Cin, int Cout, int CoutP) {
    __shared__ float red[16][64];
    const int ol = threadIdx.x & 63, cs = threadIdx.x >> 6;
    const int o = blockIdx.x * 64 + ol, b = blockIdx.y;
    float acc = 0.f;
    if (o < Cout)
        for (int c = cs; c < Cin; c += 16) {
            const float s = styles[b * Cin + c];
            acc = fmaf_(s * s, wsq[(int64_t)c * CoutP + o], acc);
        }
    red[cs][ol] = acc;
    __syncthreads();
    if (cs == 0 && o < Cout) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) t += red[k][ol];
        d[b * Cout + o] = 1.0f / sqrtf(t + 1e-8f);
    }
}

// All demodulation coefficients of a forward in ONE launch (15 separate launches of this much work are 15 launch latencies):
// meta[l] = (wsq pointer, styles offset, Cin, Cout, CoutP, output offset), offsets in floats into styles_all / dcoef_all.
__global__ __launch_bounds__(1024) void demod_batch_kernel(const float* __restrict__ styles_all, const int64_t* __restrict__ meta, float* __restrict__ dcoef_all, int B) {
    __shared__ float red[16][64];
    const int64_t* m = meta + (int64_t)blockIdx.z * 6;
    const float* wsq = (const float*)m[0];
    const int Cin = (int)m[2], Cout = (int)m[3], CoutP = (int)m[4];
    const float* styles = styles_all + m[1];
    float* d = dcoef_all + m[5];
    const int ol = threadIdx.x & 63, cs = threadIdx.x >> 6;
    const int o = blockIdx.x * 64 + ol, b = blockIdx.y;
    if (blockIdx.x * 64 >= Cout) return;
    float acc = 0.f;
    if (o < Cout)
        for (int c = cs; c < Cin; c += 16) {
            const float s = styles[b * Cin + c];
            acc = fmaf_(s * s, wsq[(int64_t)c * CoutP + o], acc);
        }
    red[cs][ol] = acc;
    __syncthreads();
    if (cs == 0 && o < Cout) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) t += red[k][ol];
        d[b * Cout + o] = 1.0f / sqrtf(t + 1e-8f);
    }
}

// weight [Cout,Cin,k,k] -> packed (zero padded) [nchunks][k*k][CoutP][4] followed by wsq [Cin][CoutP]
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w, float* __restrict__ wp, float* __restrict__ wsq, int Cout, int Cin, int T,
                                                  int KC, int CoutP, int nchunks) {
    const int64_t total = (int64_t)nchunks * T * KC * CoutP;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % KC);               // [chunk][tap][CoutP][KC]: the KC channels of a column are one 16-B vector
        int64_t r = i / KC;
        const int o = (int)(r % CoutP); r /= CoutP;
        const int t = (int)(r % T);
        const int cc = (int)(r / T);
        const int c = cc * KC + c8;
        wp[i] = (o < Cout && c < Cin) ? w[((int64_t)o * Cin + c) * T + t] : 0.f;
    }
    const int64_t total2 = (int64_t)Cin * CoutP;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total2; i += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(i % CoutP), c = (int)(i / CoutP);
        float acc = 0.f;
        if (o < Cout)
            for (int t = 0; t < T; t++) { const float v = w[((int64_t)o * Cin + c) * T + t]; acc = fmaf_(v, v, acc); }
        wsq[i] = acc;
    }
}

// One wave per (b,row): out[obase + b*olen + oidx] = (ws[b,widx,:] . (A[row,:] * wgain) + abias[row]) * scale[row]
// row_meta[row] = (widx, obase, olen, oidx): every layer gets its own contiguous [B, Cin_l] block.
__global__ __launch_bounds__(256) void style_affine_kernel(const float* __restrict__ ws, const float* __restrict__ A, const float* __restrict__ abias,
                                                          const int32_t* __restrict__ meta, const float* __restrict__ scale, float* __restrict__ styles,
                                                          int B, int num_ws, int w_dim, int rows, float wgain) {
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (gw >= (int64_t)B * rows) return;
    const int b = (int)(gw / rows), row = (int)(gw % rows);
    const int widx = meta[row * 4 + 0], obase = meta[row * 4 + 1], olen = meta[row * 4 + 2], oidx = meta[row * 4 + 3];
    const float* x = ws + ((int64_t)b * num_ws + widx) * w_dim;
    const float* a = A + (int64_t)row * w_dim;
    float acc = 0.f;
    for (int j = lane_id(); j < w_dim; j += 64) acc = fmaf_(x[j], a[j] * wgain, acc);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane_id() == 0) styles[(int64_t)obase + (int64_t)b * olen + oidx] = (acc + abias[row]) * scale[row];
}

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

struct PackInfo { int T, KC, CoutP, nchunks, niter16, nch32, nch8, nch4, nsl64; int64_t wp_floats, wsq_floats, wsplit_floats, wbf_floats, wino_floats, wino4_floats; };
inline PackInfo pack_info(int Cout, int Cin, int k) {
    PackInfo pi;
    pi.T = k * k;
    pi.KC = KC3;
    pi.CoutP = round_up(Cout, 4);
    pi.nchunks = round_up((Cin + pi.KC - 1) / pi.KC, k == 1 ? 16 : 2);      // zero-padded to whole K iterations (1x1 kernels stage up to 16 chunks)
    pi.wp_floats = (int64_t)pi.nchunks * pi.T * pi.KC * pi.CoutP;
    pi.wsq_floats = (int64_t)Cin * pi.CoutP;
    pi.niter16 = (Cin + 15) / 16;
    pi.wsplit_floats = k == 3 ? (int64_t)pi.niter16 * 9 * 3 * pi.CoutP * 8 : 0;      // split-bf16 copy of the 3x3 weights (opt-in arithmetic)
    pi.nch32 = (Cin + 31) / 32;
    pi.wbf_floats = k == 3 ? (int64_t)pi.nch32 * 9 * 2 * pi.CoutP * 8 : 0;          // bf16 copy of the 3x3 weights (reduced-precision blocks)
    pi.nch8 = (Cin + 7) / 8;
    pi.wino_floats = k == 3 ? (int64_t)pi.nch8 * 16 * 2 * pi.CoutP * 4 : 0;          // Winograd-domain weights G g G^T (modconv_wino.inc)
    pi.nch4 = (Cin + 3) / 4; pi.nsl64 = (Cout + 63) / 64;
    pi.wino4_floats = (k == 3 && Cin >= TDGP_WINO4_MIN_C && Cout >= TDGP_WINO4_MIN_C) ? (int64_t)cdiv(Cout, W4_BM) * pi.nch4 * W4_UCH : 0;      // F(4x4,3x3)-domain weights (modconv_wino4.inc)
    return pi;
}

// pixel tiles of a launch whose blocks cover NT 32-pixel subtiles
inline int px_tiles(const ConvParams& p, int NT) {
    const int TW = 1 << p.tw_log2, RPS = 32 >> p.tw_log2, TR = NT * RPS;
    return cdiv(p.ph.gridW, TW) * cdiv(p.B * p.ph.gridH, TR);
}

// Split-K factor: low-resolution layers have K = Cin*9 = 4608 but only a handful of output tiles, so a plain launch
// leaves most of the 256 CUs idle behind a 64-iteration serial K loop.  Split K until ~2 blocks per CU exist.
inline int pick_ksplit(int blocks, int niter) {
    if (blocks >= 256 || niter < 4) return 1;
    int ks = cdiv(512, blocks);
    if (ks > niter / 2) ks = niter / 2;
    if (ks > 32) ks = 32;
    return ks < 1 ? 1 : ks;
}

template <int MTW, int NTW, int WM, int WN, int KCS, int MAXT>
int launch_conv(ConvParams& p, float* partial, int64_t partial_floats, hipStream_t s) {
    constexpr int BM = 32 * MTW * WM, NT = NTW * WN, BN = 32 * NT;
    constexpr int HALO = MAXT == 25 ? 2 : 1;
    constexpr int XS_MAX = (BN / 4 + 2 * HALO) * (4 + 2 * HALO) > (BN / 32 + 2 * HALO) * (32 + 2 * HALO) ? (BN / 4 + 2 * HALO) * (4 + 2 * HALO) : (BN / 32 + 2 * HALO) * (32 + 2 * HALO);
    const size_t lds = (size_t)(2 * (MAXT * KCS * BM + KCS * XS_MAX + 64) + 32 + 5 * BM) * sizeof(float);
    TDGP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)conv_mfma_kernel<MTW, NTW, WM, WN, KCS, MAXT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds););   // raise the dynamic-LDS cap once per instantiation and device
    const int gx = px_tiles(p, NT), gy = cdiv(p.Cout, BM);
    const int niter = cdiv(cdiv(p.Cin, KC3), KCS / KC3);
    const int64_t slice = (int64_t)p.e.B * p.e.Cout * p.e.Hout * p.e.Wout;
    int ks = pick_ksplit(gx * gy, niter);
    while (ks > 1 && ks * slice > partial_floats) ks--;          // never exceed the caller's workspace
    p.ksplit = ks;
    p.partial = partial;
    dim3 grid(gx, gy, ks);
    TDGP_LAUNCH("conv_mfma_kernel", (conv_mfma_kernel<MTW, NTW, WM, WN, KCS, MAXT>), grid, dim3(256), lds, s, p);
    if (ks > 1)
        TDGP_LAUNCH("splitk_reduce_kernel", splitk_reduce_kernel, dim3((int)min((int64_t)2048, cdiv64(slice, 256))), dim3(256), 0, s, partial, ks, p.e);
    return 0;
}

// Split-K for launches that already fill the chip: tail balancing.  512 block slots (2 blocks per CU); a launch of n blocks
// runs n / 512 full rounds plus a tail that costs ~0.7 of a round when <= 256 blocks are left (one block per CU runs faster)
// -- e.g. 560 blocks are 1.7 rounds for 1.09 rounds of work.  Splitting K by ks shrinks the rounds; each extra slice costs
// one more pass over the partial sums (~4 TB/s, mostly served by the 256 MB MALL).  Times in microseconds at ~100 TFLOP/s of
// block throughput.
inline int tail_ksplit(int blocks, int niter, double flop, int64_t slice_floats, int max_ks) {
    double best = 1e30;
    int ks = 1;
    for (int k = 1; k <= max_ks && k <= 4 && niter / k >= 16; k++) {
        const int n = blocks * k, full = n / 512, rem = n % 512;
        const double rounds = full + (rem == 0 ? 0.0 : (rem <= 256 ? 0.7 : 1.0));
        const double t = rounds * (flop / n) / (100e12 / 512) * 1e6 + (k - 1) * (double)slice_floats * 4.0 / 4e12 * 1e6;
        if (t < best * 0.97) { best = t; ks = k; }
    }
    return ks;
}

// x2 layers: tile configuration, split-K factor and Z geometry -- shared by the workspace query and the launch.
struct UpPlan { int cfg, BM, BN, G1, GS, ksplit; int64_t zslice; };
inline UpPlan up_plan(int B, int Cin, int Cout, int H, int W) {
    UpPlan u;
    u.cfg = Cout > 64 ? 0 : 1;
    u.BM = Cout > 64 ? 128 : 64; u.BN = Cout > 64 ? 64 : 128;
    u.G1 = W + 2;                  // grid pitch: W + 1 would do (one zero column); W + 2 makes the Z row pitch 2*G1 a multiple of 4 floats -> 16-B FIR loads
    u.GS = round_up((H + 1) * (W + 2), 32);
    u.zslice = (int64_t)B * Cout * 4 * u.GS;
    const int blocks = cdiv(B * u.GS, u.BN) * cdiv(Cout, u.BM);
    const int niter = cdiv(Cin, 4);
    int ks;
    if (blocks < 256) {
        ks = pick_ksplit(blocks, niter);
        const int64_t cap = ((int64_t)64 << 20) / 4;             // low-resolution layers: at most 64 MiB of slices
        while (ks > 1 && ks * u.zslice > cap) ks--;
    } else {
        ks = tail_ksplit(blocks, niter, 2.0 * Cin * Cout * 9.0 * H * W * B, u.zslice, 4);
    }
    u.ksplit = ks;
    return u;
}

template <int MTW, int NTW, int WM, int WN, bool DEEP>
void launch_upconv(const UpParams& u, hipStream_t s) {
    constexpr int BM = 32 * MTW * WM, BN = 32 * NTW * WN, NW = WM * WN;
    constexpr int stage = 2 * (9 * BM * 4 + 2 * (BN + 2) * 4), epi = NW * 32 * UP_CT_W;
    const size_t lds = (size_t)(stage > epi ? stage : epi) * sizeof(float);
    TDGP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)upconv_mfma_kernel<MTW, NTW, WM, WN, DEEP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds););
    dim3 grid(cdiv(u.B * u.GS, BN), cdiv(u.Cout, BM), u.ksplit);
    TDGP_LAUNCH("upconv_mfma_kernel", (upconv_mfma_kernel<MTW, NTW, WM, WN, DEEP>), grid, dim3(64 * NW), lds, s, u);
}

// timing experiment (tools/dev/build_variant.sh): extra dynamic LDS per block = fewer resident blocks per CU.  0 in the shipped library.
#ifndef TDGP_RGB_LDS_PAD
#define TDGP_RGB_LDS_PAD 0
#endif
template <int MT, bool RESIDENT, bool FAST, bool XBF>
void launch_torgb_v(const RgbParams& r, hipStream_t s) {
    constexpr int BM = 32 * MT;
    const size_t lds = (size_t)(16 * BM * 4 + 16 * 128 * 4 + BM) * sizeof(float) + TDGP_RGB_LDS_PAD;
    TDGP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)torgb_mfma_kernel<MT, RESIDENT, FAST, XBF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds););
    // RESIDENT: several consecutive tiles per block once there are more tiles than ~4 rounds of the 512 resident blocks
    RgbParams rr = r;
    const int64_t ntiles = cdiv64(r.P, 128);
#ifndef TDGP_RGB_WS
#define TDGP_RGB_WS 1                  // 0: torgb_mfma_kernel for every layer (A/B builds, same bits)
#endif
    if constexpr (RESIDENT && FAST && !XBF) {
        // the two-role form: enough tiles for >= 16 per block on a grid of four blocks per CU (C3 / C4: the 512^2 layer; at B = 1 the 512^2 layer still has 2048 tiles)
        const int cus = tdgp_cu_count();
        if (TDGP_RGB_WS && r.e.skip && ntiles >= (int64_t)cus * 8) {
            constexpr size_t lds_ws = (size_t)(16 * BM * 4 + 2 * 16 * 128 * 4 + 4 * MT * 32 * CT_LD + BM) * sizeof(float);
            TDGP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)torgb_ws_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ws););
            const int64_t blocks = (int64_t)cus * 4;
            rr.tpb = (int)cdiv64(ntiles, blocks);
            TDGP_LAUNCH("torgb_mfma_kernel", (torgb_ws_kernel<MT>), dim3((unsigned)cdiv64(ntiles, rr.tpb)), dim3(512), lds_ws, s, rr);
            return;
        }
    }
    rr.tpb = RESIDENT ? (int)max((int64_t)1, min((int64_t)8, ntiles / 2048)) : 1;
    TDGP_LAUNCH("torgb_mfma_kernel", (torgb_mfma_kernel<MT, RESIDENT, FAST, XBF>), dim3((unsigned)cdiv64(ntiles, rr.tpb)), dim3(256), lds, s, rr);
}

template <int MT, bool XBF = false>
void launch_torgb(const RgbParams& r, hipStream_t s) {
    const bool fast = r.lw >= 0 && r.e.clamp < 0.f && r.e.gain == 1.f;
    if constexpr (XBF) {                                    // the reduced-precision blocks always clamp: one output stage
        if (r.Cin <= 64) launch_torgb_v<MT, true, false, true>(r, s); else launch_torgb_v<MT, false, false, true>(r, s);
    } else {
        if (r.Cin <= 64) { if (fast) launch_torgb_v<MT, true, true, false>(r, s); else launch_torgb_v<MT, true, false, false>(r, s); }
        else { if (fast) launch_torgb_v<MT, false, true, false>(r, s); else launch_torgb_v<MT, false, false, false>(r, s); }
    }
}

// KS x KS stride-1 fast path (W % 32 == 0)
template <int KS, int MTW, int NTW, int WM, int WN>
int launch_conv3(Conv3Params& p, float* partial, int64_t partial_floats, hipStream_t s) {
    constexpr int BM = 32 * MTW * WM, NT = NTW * WN, R = KS / 2;
    constexpr int stage = 2 * (KS * KS * BM * 4 + (NT + 2 * R) * (32 + 2 * R) * 4) + 5 * BM, epi = 4 * 32 * CT_LD;
    const size_t lds = (size_t)(stage > epi ? stage : epi) * sizeof(float);
    TDGP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)conv3_mfma_kernel<KS, MTW, NTW, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds););
    const int gx = (p.W >> 5) * cdiv(p.B * (p.H + R), NT), gy = cdiv(p.Cout, BM);
    const int niter = cdiv(p.Cin, 4);
    const int64_t slice = (int64_t)p.B * p.Cout * p.H * p.W;
    const int max_ks = (int)min((int64_t)32, partial_floats / slice);
    int ks = gx * gy < 256 ? pick_ksplit(gx * gy, niter) : tail_ksplit(gx * gy, niter, 2.0 * p.Cin * p.Cout * KS * KS * p.H * p.W * p.B, 2 * slice, max_ks);
    if (ks > max_ks) ks = max_ks;
    if (ks < 1) ks = 1;
    p.ksplit = ks;
    p.partial = partial;
    TDGP_LAUNCH("conv_mfma_kernel", (conv3_mfma_kernel<KS, MTW, NTW, WM, WN>), dim3(gx, gy, ks), dim3(256), lds, s, p);
    if (ks > 1)
        TDGP_LAUNCH("splitk_reduce_kernel", splitk_reduce_kernel, dim3((int)min((int64_t)2048, cdiv64(slice, 256))), dim3(256), 0, s, partial, ks, p.e);
    return 0;
}

inline int pick_tw_log2(int gridW) {
    int tw = 4, lg = 2;
    while (tw < 32 && tw < gridW) { tw <<= 1; lg++; }
    return lg;
}

// F(4x4,3x3) Winograd (modconv_wino4.inc): which stride-1 3x3 layers take it -- by shape only (the workspace is sized from the same test).
// whole tile groups of 512 pixels (64 x 8, or 32 x 16 for the 32-pixel-wide layers); Cin, Cout >= 128: below, the separate input-transform pass (2.25 x the input, written and
// read back) costs more than it saves; at least one item per CU; V addressed through a 4 GiB buffer descriptor.
inline int64_t wino4_v_bytes(int B, int Cin, int H, int W) { return (int64_t)B * ((H * W) >> 9) * ((Cin + 3) / 4) * (9 * 4 * 32 * 4) * 4; }       // 512 pixels per tile group
inline int wino4_txl(int H, int W) { return (W & 63) == 0 && (H & 7) == 0 ? 4 : ((W & 31) == 0 && (H & 15) == 0 ? 3 : 0); }
// The batch goes through the two kernels in sub-batches whose V fits one buffer descriptor (< 4 GiB; TDGP_WINO4_VCAP_MB lowers the cap for
// experiments): C4's 512^2 x 128 layer at B = 16 has 4.8 GB of V.  The sub-batches share one V buffer (stream order).
#ifndef TDGP_WINO4_VCAP_MB
#define TDGP_WINO4_VCAP_MB 4095
#endif
// Layers with few channels move 2.25 x their input through V for little arithmetic: there a sub-batch whose V stays inside the 256 MB
// Infinity Cache between the two kernels (TDGP_WINO4_VSMALL_MB) beats one big launch (measured B = 16: 256^2 x 128 1.38 (one sample per launch) /
// 1.22 (two) / 1.25 (three) ms; 128^2 x 256 0.93 -> 1.03 ms -- hence Cin <= 128 only, and at least two samples per launch).
#ifndef TDGP_WINO4_VSMALL_MB
#define TDGP_WINO4_VSMALL_MB 192
#endif
#ifndef TDGP_WINO4_VSMALL_MINB
#define TDGP_WINO4_VSMALL_MINB 2          // fewest samples per launch on that path (1: A/B builds)
#endif
inline int64_t wino4_items(int bs, int Cout, int H, int W) { return (int64_t)bs * ((H * W) >> 9) * cdiv(Cout, 64); }
inline int wino4_sub_batch(int B, int Cin, int Cout, int H, int W) {
    const int64_t per = wino4_v_bytes(1, Cin, H, W);
    if (Cin <= 128 && TDGP_WINO4_VSMALL_MB > 0) {
        const int bs = (int)std::min<int64_t>(B, ((int64_t)TDGP_WINO4_VSMALL_MB << 20) / per);
        if (bs >= TDGP_WINO4_VSMALL_MINB && wino4_items(bs, Cout, H, W) >= 256) return bs;      // (one sample per launch -- 512^2 x 64: 151 MB of V next to 134 MB of x and y -- does not stay in the cache anyway: 1.84 vs 1.77 ms for the whole batch at once)
    }
    return (int)std::min<int64_t>(B, ((int64_t)TDGP_WINO4_VCAP_MB << 20) / per);
}
// Launches with too few items for the chip (the 32^2 layers at batch 4 .. 8) split the input channels of every item 2 or 4 ways (plain layers
// only; >= 16 chunks per split; the whole batch in one launch); the splits' raw sums go through the split-K buffer of the direct kernels.
// 0 = not a split-K shape.  B = 4, 32^2 x 512: direct sums 0.235 ms -> 4 splits (measured).
#ifndef TDGP_WINO4_SPLITK
#define TDGP_WINO4_SPLITK 1
#endif
inline int wino4_ksplit_log2(int B, int Cin, int Cout, int H, int W) {
    if (!TDGP_WINO4_SPLITK || wino4_v_bytes(B, Cin, H, W) > ((int64_t)TDGP_WINO4_VSMALL_MB << 20)) return 0;
    const int64_t base = wino4_items(B, Cout, H, W);
    const int nch = Cin >> 2;
    for (int l = 1; l <= 2; l++)
        if ((base << l) >= 256 && (nch & ((1 << l) - 1)) == 0 && (nch >> l) >= 16) return l;
    return 0;
}
inline bool wino4_shape_ok(int B, int Cin, int Cout, int H, int W, int k, int up, bool plain = false) {
    if (!(k == 3 && up == 1 && wino4_txl(H, W) != 0 && (Cin & 3) == 0 && Cin >= TDGP_WINO4_MIN_C && Cout >= TDGP_WINO4_MIN_C)) return false;
    const int bs = wino4_sub_batch(B, Cin, Cout, H, W);
    if (bs >= 1 && wino4_items(bs, Cout, H, W) >= 256) return true;
    return plain && wino4_items(B, Cout, H, W) < 256 && wino4_ksplit_log2(B, Cin, Cout, H, W) > 0;
}

// Workspace layout: [demod coefficients B*Cout] [transposed-conv intermediate, up=2 only] [split-K partial sums] [Winograd-domain input V]
struct WsLayout { int64_t dco, z, partial, partial_floats, wino_v, total; };
WsLayout ws_layout(int B, int Cin, int Cout, int H, int W, int k, int up) {
    WsLayout w;
    auto al = [](int64_t v) { return (v + 255) / 256 * 256; };
    w.dco = 0;
    w.z = al((int64_t)B * Cout * sizeof(float));
    if (up == 2) {
        // the parity-planar transposed-conv intermediate (x its split-K slices); the FIR kernel reduces the slices itself
        const UpPlan u = up_plan(B, Cin, Cout, H, W);
        w.partial = w.z + al(u.zslice * u.ksplit * (int64_t)sizeof(float));
        w.partial_floats = 0;
        w.wino_v = w.total = w.partial;
        return w;
    }
    const int64_t out_elems = (int64_t)B * Cout * H * W;
    w.partial = w.z;
    // split-K is only chosen for launches with < 256 blocks: bound its buffer at 32 slices and 64 MiB
    int64_t pf = out_elems * 32;
    const int64_t cap = ((int64_t)64 << 20) / 4;
    if (pf > cap) pf = (cap / out_elems) * out_elems;
    w.partial_floats = pf;
    w.wino_v = w.partial + al(pf * (int64_t)sizeof(float));
    w.total = w.wino_v + (wino4_shape_ok(B, Cin, Cout, H, W, k, up, true) ? al(wino4_v_bytes(wino4_sub_batch(B, Cin, Cout, H, W), Cin, H, W)) + 256 : 0);      // + the item counters
    return w;
}

}  // namespace

TDGP_API int64_t tdgp_modconv_pack_bytes(int Cout, int Cin, int k) {
    if (Cout < 1 || Cin < 1 || (k != 1 && k != 3 && k != 5)) return -1;
    const PackInfo pi = pack_info(Cout, Cin, k);
    return (pi.wp_floats + pi.wsq_floats + pi.wsplit_floats + pi.wbf_floats + pi.wino_floats + pi.wino4_floats) * (int64_t)sizeof(float);
}

TDGP_API int tdgp_modconv_pack(const float* weight, void* wpack, int Cout, int Cin, int k, tdgp_stream_t stream) {
    TDGP_CHECK(weight && wpack, TDGP_EINVAL, "modconv_pack: null pointer");
    TDGP_CHECK(Cout >= 1 && Cin >= 1, TDGP_EINVAL, "modconv_pack: bad channel counts");
    TDGP_CHECK(k == 1 || k == 3 || k == 5, TDGP_EUNSUPPORTED, "modconv_pack: kernel size %d not on the generator path (1, 3 or 5)", k);
    const PackInfo pi = pack_info(Cout, Cin, k);
    float* wp = (float*)wpack;
    TDGP_LAUNCH("pack_kernel", pack_kernel, dim3((int)min((int64_t)4096, cdiv64(pi.wp_floats, 256))), dim3(256), 0, (hipStream_t)stream, weight, wp,
                       wp + pi.wp_floats, Cout, Cin, pi.T, pi.KC, pi.CoutP, pi.nchunks);
    if (pi.wsplit_floats > 0)
        TDGP_LAUNCH("pack_kernel", pack_split_kernel, dim3((int)min((int64_t)4096, cdiv64(pi.wsplit_floats, 256))), dim3(256), 0, (hipStream_t)stream, weight,
                    (uint32_t*)(wp + pi.wp_floats + pi.wsq_floats), Cout, Cin, pi.CoutP, pi.niter16);
    if (pi.wbf_floats > 0)
        TDGP_LAUNCH("pack_kernel", pack_bf16_kernel, dim3((int)min((int64_t)4096, cdiv64(pi.wbf_floats, 256))), dim3(256), 0, (hipStream_t)stream, weight,
                    (uint32_t*)(wp + pi.wp_floats + pi.wsq_floats + pi.wsplit_floats), Cout, Cin, pi.CoutP, pi.nch32);
    if (pi.wino_floats > 0)
        TDGP_LAUNCH("pack_kernel", pack_wino_kernel, dim3((int)min((int64_t)4096, cdiv64(pi.wino_floats, 256))), dim3(256), 0, (hipStream_t)stream, weight,
                    wp + pi.wp_floats + pi.wsq_floats + pi.wsplit_floats + pi.wbf_floats, Cout, Cin, pi.CoutP, pi.nch8);
    if (pi.wino4_floats > 0)
        TDGP_LAUNCH("pack_kernel", pack_wino4_kernel, dim3((int)min((int64_t)4096, cdiv64(pi.wino4_floats, 256))), dim3(256), 0, (hipStream_t)stream, weight,
                    wp + pi.wp_floats + pi.wsq_floats + pi.wsplit_floats + pi.wbf_floats + pi.wino_floats, Cout, Cin, cdiv(Cout, W4_BM), pi.nch4);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

static int g_conv_arith = 0;
static inline bool arith_wino4() { return g_conv_arith == 0 || g_conv_arith == 4; }
TDGP_API int tdgp_set_conv_arith(int mode) {
    TDGP_CHECK(mode >= 0 && mode <= 4, TDGP_EINVAL, "set_conv_arith: mode %d (0 = fp32 MFMA, Winograd F(4x4,3x3) / F(2x2,3x3) where they pay; 1 = split-bf16 MFMA with fp32 accumulation; 2 = fp32 MFMA, direct sums only; 3 = as 0 without F(4x4); 4 = as 0 with the F(4x4) input transform always as a pass of its own)", mode);
    const int old = g_conv_arith;
    g_conv_arith = mode;
    return old;
}


// Cheap shape query (no launch, no allocation): does tdgp_modconv2d take a FOLDED x2 layer (out_layout 2: four parity 3x3 kernels, Cout4 = 4 x the
// layer's output channels) of this shape on the Winograd F(4x4) kernels under the current arithmetic mode?  The binding asks BEFORE it folds and
// packs the [4 Cout, Cin, 3, 3] weights (small launches paid the fold, the pack and an exception to learn the answer).
TDGP_API int tdgp_modconv2d_takes_folded_up2(int B, int Cin, int Cout4, int H, int W) {
    return (arith_wino4() && B >= 1 && (Cout4 & 3) == 0 && wino4_shape_ok(B, Cin, Cout4, H, W, 3, 1)) ? 1 : 0;
}

TDGP_API int64_t tdgp_modconv2d_workspace_bytes(int B, int Cin, int Cout, int H, int W, int k, int up) {
    return ws_layout(B, Cin, Cout, H, W, k, up).total;
}

// Winograd kernel (modconv_wino.inc): which stride-1 3x3 layers take it.  TDGP_WINO_MIN_CIN: below it the direct kernel's staging
// economy wins (K = Cin per position instead of 9 Cin); measured per layer, see DESIGN.md.
#ifndef TDGP_WINO_MIN_CIN
#define TDGP_WINO_MIN_CIN 64
#endif
inline size_t wino_lds_bytes(int) { return (size_t)(2 * 8192 + 2 * 8192 + 2 * 8 * 10 * 48 + 128) * 4; }
inline bool wino_ok(int B, int Cin, int Cout, int H, int W) {
    // (fewer than one block per CU: the direct kernel's split-K fills the chip better)
    return (W & 31) == 0 && (H & 7) == 0 && (Cin & 7) == 0 && Cin >= TDGP_WINO_MIN_CIN && wino_lds_bytes(Cin) <= 160 * 1024 &&
           (int64_t)(W >> 5) * (H >> 3) * B * cdiv(Cout, 64) >= 256;
}

TDGP_API int tdgp_modconv2d(const float* x, const void* wpack, const float* styles, const float* dcoef_in, const float* noise, int64_t noise_bstride,
                            const float* bias, const float* fir4x4, const float* skip, float* y, int B, int Cin, int Cout, int H, int W,
                            int k, int up, int demodulate, int act, float alpha, float gain, float clamp, int out_layout, int out_feat,
                            void* workspace, int64_t workspace_bytes, tdgp_stream_t stream) {
    TDGP_CHECK(x && wpack && y, TDGP_EINVAL, "modconv2d: null pointer");
    TDGP_CHECK(B >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1, TDGP_EINVAL, "modconv2d: bad shape");
    TDGP_CHECK(k == 1 || k == 3 || k == 5, TDGP_EUNSUPPORTED, "modconv2d: kernel size %d not supported (1, 3 or 5)", k);
    TDGP_CHECK(up == 1 || (up == 2 && k == 3), TDGP_EUNSUPPORTED, "modconv2d: up=%d with k=%d not supported", up, k);
    TDGP_CHECK(up == 1 || fir4x4, TDGP_EINVAL, "modconv2d: up=2 needs the 4x4 resample filter");
    TDGP_CHECK(!skip || (up == 1 && k == 1 && fir4x4 && (H % 2) == 0 && (W % 2) == 0), TDGP_EINVAL, "modconv2d: skip needs k=1, up=1, even H/W and the filter");
    TDGP_CHECK(!demodulate || styles, TDGP_EINVAL, "modconv2d: demodulate needs styles");
    TDGP_CHECK(act >= 1 && act <= 9, TDGP_EUNSUPPORTED, "modconv2d: unknown activation %d", act);
    TDGP_CHECK(out_layout == 0 || out_layout == 2 || (out_layout == 1 && out_feat >= 4 && (out_feat % 4) == 0 && (Cout % out_feat) == 0 && up == 1 && k == 1), TDGP_EINVAL,
               "modconv2d: the channel-last plane layout is a ToRGB (k=1, up=1) output");
    // out_layout 2: the x2 layers with the FIR folded into four 3x3 parity kernels (Cout = 4 x the real channels, channel 4 o + 2 py + px;
    // y is [B, Cout / 4, 2H, 2W], noise a 2H x 2W map, bias / dcoef_in replicated per parity by the caller).  Only the F(4x4) kernels write it.
    TDGP_CHECK(out_layout != 2 || (k == 3 && up == 1 && (Cout & 3) == 0 && !skip && (!demodulate || dcoef_in)), TDGP_EINVAL,
               "modconv2d: out_layout 2 (folded x2 layer) needs k=3, up=1, Cout %% 4 == 0, no skip and precomputed demodulation coefficients");
    TDGP_CHECK(out_layout != 2 || (arith_wino4() && wino4_shape_ok(B, Cin, Cout, H, W, k, up) && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
                                   (!noise || (((uintptr_t)noise & 15) == 0 && (noise_bstride & 3) == 0))), TDGP_EUNSUPPORTED,
               "modconv2d: out_layout 2 is written by the Winograd F(4x4) kernels only (shape %dx%d, %d -> %d channels, batch %d does not take them)", H, W, Cin, Cout, B);
    TDGP_CHECK((int64_t)B * Cin * H * W < ((int64_t)1 << 30) && (int64_t)B * Cout * (H * up + 1) * (W * up + 1) <= INT32_MAX, TDGP_EINVAL,
               "modconv2d: tensor too large (activations are addressed through 4 GiB buffer descriptors)");
    const WsLayout wl = ws_layout(B, Cin, Cout, H, W, k, up);
    TDGP_CHECK(workspace && workspace_bytes >= wl.total, TDGP_EWORKSPACE, "modconv2d: workspace %lld < %lld bytes", (long long)workspace_bytes,
               (long long)wl.total);
    hipStream_t s = (hipStream_t)stream;
    const PackInfo pi = pack_info(Cout, Cin, k);
    const float* wp = (const float*)wpack;
    const float* wsq = wp + pi.wp_floats;
    float* dco = (float*)((char*)workspace + wl.dco);
    float* z = (float*)((char*)workspace + wl.z);
    float* partial = (float*)((char*)workspace + wl.partial);
    if (demodulate && dcoef_in) dco = const_cast<float*>(dcoef_in);             // precomputed by tdgp_demod_batch
    else if (demodulate) TDGP_LAUNCH("demod_kernel", demod_kernel, dim3(cdiv(Cout, 64), B), dim3(1024), 0, s, styles, wsq, dco, B, Cin, Cout, pi.CoutP);
    else dco = nullptr;

    ConvParams p;
    p.x = x; p.wp = wp; p.styles = styles; p.B = B; p.Cin = Cin; p.Cout = Cout; p.CoutP = pi.CoutP; p.Hin = H; p.Win = W;
    p.T = pi.T; p.ksplit = 1; p.partial = nullptr;
    EpiParams& e = p.e;
    e.B = B; e.Cout = Cout; e.round_bf16 = 0;
    for (int i = 0; i < 16; i++) e.fir[i] = 0.f;
    if (fir4x4) {
        // fir4x4 is a HOST pointer (the filter is a static 64-byte buffer; reading it on the host keeps the call async)
        for (int ky = 0; ky < 4; ky++)
            for (int kx = 0; kx < 4; kx++) e.fir[ky * 4 + kx] = fir4x4[(3 - ky) * 4 + (3 - kx)] * 4.0f;   // no flip_filter: taps = flipped f; gain up^2
    }
    if (up == 1) {
        e.dcoef = dco; e.noise = noise; e.noise_bstride = noise_bstride; e.bias = bias; e.skip = skip; e.y = y;
        e.Hout = H; e.Wout = W; e.out_layout = out_layout; e.out_feat = out_feat > 0 ? out_feat : 1;
        e.act = act; e.alpha = alpha; e.gain = gain; e.clamp = clamp;
        TapTable& ph = p.ph;
        ph.ntaps = k * k; ph.halo = k / 2;
        for (int t = 0; t < k * k; t++) {
            ph.tap_w[t] = t;
            ph.tap_off_y[t] = t / k - k / 2;                 // correlation, padding k/2 (conv2d_resample.py:132-134)
            ph.tap_off_x[t] = t % k - k / 2;
        }
        ph.gridH = H; ph.gridW = W;
        p.tw_log2 = pick_tw_log2(W);
        if (k >= 3 && (W & 31) == 0) {
            Conv3Params c;
            c.x = x; c.wp = wp; c.styles = styles; c.partial = nullptr; c.e = e;
            c.B = B; c.Cin = Cin; c.Cout = Cout; c.CoutP = pi.CoutP; c.H = H; c.W = W; c.ksplit = 1;
            c.x_bytes = (uint32_t)((int64_t)B * Cin * H * W * 4); c.wp_bytes = (uint32_t)(pi.wp_floats * 4); c.st_bytes = (uint32_t)((int64_t)B * Cin * 4);
            const int s_blocks = (W >> 5) * cdiv(B * (H + 1), 8) * cdiv(Cout, 64);
            if (k == 3 && g_conv_arith == 1 && s_blocks >= 256 && styles && (Cin & 15) == 0 && Cin <= 2048 && H >= 16) {
                Conv3sParams q;
                q.x = x; q.wsp = wp + pi.wp_floats + pi.wsq_floats; q.styles = styles; q.e = e;
                q.B = B; q.Cin = Cin; q.Cout = Cout; q.CoutP = pi.CoutP; q.H = H; q.W = W;
                q.x_bytes = c.x_bytes; q.wsp_bytes = (uint32_t)(pi.wsplit_floats * 4); q.st_bytes = c.st_bytes;
                const size_t lds = (size_t)(2 * 3 * 3 * 64 * 32 + 3 * 10 * 34 * 32 + 5 * 64 * 4 + 2 * Cin * 4);
                TDGP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)conv3s_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds - 2 * Cin * 4 + 2 * 2048 * 4)););
                TDGP_LAUNCH("conv_mfma_kernel", conv3s_mfma_kernel, dim3((W >> 5) * cdiv(B * (H + 1), 8), cdiv(Cout, 64)), dim3(256), lds, s, q);
            } else if (g_conv_arith == 0 && TDGP_WINO4F_MAXCIN > 0 && out_layout == 0 && Cin <= TDGP_WINO4F_MAXCIN && (Cin & 15) == 0 && (Cout & 63) == 0 && wino4_txl(H, W) == 4 &&
                       wino4_shape_ok(B, Cin, Cout, H, W, k, up) && pi.wino4_floats > 0 && (int64_t)B * ((H * W) >> 9) * (Cout >> 6) >= 256 && !skip && ((uintptr_t)x & 15) == 0 &&
                       ((uintptr_t)y & 15) == 0 && (!noise || (((uintptr_t)noise & 15) == 0 && (noise_bstride & 3) == 0))) {
                // few input channels (the 256^2 / 512^2 blocks): the input transform runs inside the GEMM kernel, V never leaves the CU (modconv_wino4f.inc)
                float* vbuf = (float*)((char*)workspace + wl.wino_v);
                int* ticket = (int*)((char*)vbuf + ((wino4_v_bytes(wino4_sub_batch(B, Cin, Cout, H, W), Cin, H, W) + 255) / 256 * 256));
                const int cus = tdgp_cu_count(), nxcd = (cus % 8 == 0 && cus >= 64) ? 8 : 1, per = cus / nxcd, nsl = Cout >> 6;
                int rs = 1;
                while (rs * 2 <= nsl && (per % (rs * 2)) == 0 && (rs * 2) * 64 + per / (rs * 2) * 32 < rs * 64 + per / rs * 32) rs *= 2;
                Wino4fParams q;
                q.x = x; q.styles = styles; q.u = wp + pi.wp_floats + pi.wsq_floats + pi.wsplit_floats + pi.wbf_floats + pi.wino_floats; q.e = e;
                q.B = B; q.Cin = Cin; q.Cout = Cout; q.H = H; q.W = W; q.u_bytes = (uint32_t)(pi.wino4_floats * 4); q.x_bytes = c.x_bytes; q.st_bytes = c.st_bytes;
                q.nz_bytes = noise ? (uint32_t)(((noise_bstride ? (int64_t)(B - 1) * noise_bstride : 0) + (int64_t)H * W) * 4) : 0u;
                q.gxn = W / 64; q.gyn = H / 8; q.rs = rs; q.rt = per / rs; q.nxcd = nxcd; q.ticket = ticket;
                const size_t lds = (size_t)W4F_LDS_FLOATS * 4;
                TDGP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)conv3_wino4f_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds););
                TDGP_CHECK(hipMemsetAsync(ticket, 0, 32, s) == hipSuccess, TDGP_ELAUNCH, "modconv2d: clearing the item counters failed");
                TDGP_LAUNCH("conv_wino4f_kernel", conv3_wino4f_kernel, dim3((unsigned)(nxcd * per)), dim3(512), lds, s, q);
            } else if (arith_wino4() && wino4_shape_ok(B, Cin, Cout, H, W, k, up, out_layout == 0) && pi.wino4_floats > 0 && (out_layout == 0 || out_layout == 2) && !skip && ((uintptr_t)x & 15) == 0 &&
                       ((uintptr_t)y & 15) == 0 && (!noise || (((uintptr_t)noise & 15) == 0 && (noise_bstride & 3) == 0))) {
                float* vbuf = (float*)((char*)workspace + wl.wino_v);
                // persistent grid; the blocks of an XCD (b, b + 8, ...) take a rectangle of rs slices x rt tile groups per pass: per pass an XCD's L2 then
                // fetches rs U slices + rt V tile groups instead of one of each per block (bytes ~ rs * BM + rt * 32: a slice's U chunk : a tile group's V chunk)
#ifndef TDGP_WINO4_PAIR
#define TDGP_WINO4_PAIR 1          // 1: the 8-wave form (a slice x a pair of tile groups per block, three V stages; modconv_wino4.inc)
#endif
                constexpr bool pairk = TDGP_WINO4_PAIR && W4_BM == 32;
                const int bpc = (W4_BM == 64 || pairk) ? 1 : 2;                 // resident blocks per CU
                // K split (too few items for the chip): only when the unsplit shape does not qualify, the split-K buffer holds the slices, plain layers
                int ksl = (out_layout == 0 && !wino4_shape_ok(B, Cin, Cout, H, W, k, up)) ? wino4_ksplit_log2(B, Cin, Cout, H, W) : 0;
                const int64_t kslice = (int64_t)B * Cout * H * W;
                if ((kslice << ksl) > wl.partial_floats) ksl = 0;               // (cannot happen for shapes under 256 items: 4 splits x 255 items x 32768 floats < the 64 MiB buffer; unsplit is still correct)
                const int cus = tdgp_cu_count(), nxcd = (cus % 8 == 0 && cus >= 64) ? 8 : 1, per = cus / nxcd * bpc, nsl = cdiv(Cout, W4_BM) << ksl;
                int rs = 1;
                const int tpi = pairk ? 64 : 32;                                // tiles per item
                while (rs * 2 <= nsl && (per % (rs * 2)) == 0 && (rs * 2) * W4_BM + per / (rs * 2) * tpi < rs * W4_BM + per / rs * tpi) rs *= 2;
                const size_t lds = pairk ? (size_t)(8 * W4_UCH + 3 * W4_BM + 4) * 4 : (size_t)(2 * W4_STAGE + 2 * W4_BM + 4) * 4;          // the stages, bias + demodulation of the slice, the ticket
                TDGP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)conv3_wino4_kernel<false, pairk>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); (void)hipFuncSetAttribute((const void*)conv3_wino4_kernel<true, pairk>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds););
                const int bsub = ksl ? B : wino4_sub_batch(B, Cin, Cout, H, W);
                int* ticket = (int*)((char*)vbuf + ((wino4_v_bytes(wino4_sub_batch(B, Cin, Cout, H, W), Cin, H, W) + 255) / 256 * 256));
                for (int b0 = 0; b0 < B; b0 += bsub) {
                    const int bn = std::min(bsub, B - b0);
                    Wino4Params q;
                    q.v = vbuf; q.u = wp + pi.wp_floats + pi.wsq_floats + pi.wsplit_floats + pi.wbf_floats + pi.wino_floats; q.e = e;
                    q.e.y = y + (int64_t)b0 * Cout * H * W;                        // (out_layout 2: Cout / 4 channels of 4 H W pixels -- the same count)
                    q.ups = out_layout == 2 ? 1 : 0;
                    if (e.dcoef) q.e.dcoef = e.dcoef + (int64_t)b0 * Cout;
                    if (e.noise) q.e.noise = e.noise + (int64_t)b0 * noise_bstride;
                    q.e.B = bn;
                    q.B = bn; q.Cin = Cin; q.Cout = Cout; q.CoutP = pi.CoutP; q.H = H; q.W = W;
                    q.v_bytes = (uint32_t)wino4_v_bytes(bn, Cin, H, W); q.u_bytes = (uint32_t)(pi.wino4_floats * 4);
                    q.txl = wino4_txl(H, W); q.gxn = W / (4 << q.txl); q.gyn = H / (128 >> q.txl);
                    q.rs = rs; q.rt = per / rs; q.nxcd = nxcd; q.ticket = ticket; q.ksl = ksl; q.partial = partial;
                    const int ntg = q.gxn * q.gyn * bn;
                    TDGP_LAUNCH("wino4_input_kernel", wino4_input_kernel, dim3((unsigned)(ntg * pi.nch4)), dim3(128), 0, s, x + (int64_t)b0 * Cin * H * W,
                                styles ? styles + (int64_t)b0 * Cin : nullptr, vbuf, bn, Cin, H, W, q.gxn, q.gyn, pi.nch4, q.txl, ticket);
                    if (q.ups) TDGP_LAUNCH("upconv_wino4_kernel", (conv3_wino4_kernel<true, pairk>), dim3((unsigned)(nxcd * per)), dim3(pairk ? 512 : W4_NW * 64), lds, s, q);
                    else TDGP_LAUNCH("conv_wino4_kernel", (conv3_wino4_kernel<false, pairk>), dim3((unsigned)(nxcd * per)), dim3(pairk ? 512 : W4_NW * 64), lds, s, q);
                    if (ksl) TDGP_LAUNCH("splitk_reduce_kernel", splitk_reduce_kernel, dim3((int)min((int64_t)2048, cdiv64(kslice, 256))), dim3(256), 0, s, partial, 1 << ksl, e);
                }
            } else if (k == 3 && (arith_wino4() || g_conv_arith == 3) && wino_ok(B, Cin, Cout, H, W) && out_layout == 0 && !skip && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0 &&
                       (!noise || (((uintptr_t)noise & 7) == 0 && (noise_bstride & 1) == 0))) {        // 16-byte activation loads, 8-byte noise loads / stores
                WinoParams q;
                q.x = x; q.u = wp + pi.wp_floats + pi.wsq_floats + pi.wsplit_floats + pi.wbf_floats; q.styles = styles; q.e = e;
                q.B = B; q.Cin = Cin; q.Cout = Cout; q.CoutP = pi.CoutP; q.H = H; q.W = W;
                q.x_bytes = c.x_bytes; q.u_bytes = (uint32_t)(pi.wino_floats * 4);
                const size_t lds = wino_lds_bytes(Cin);
                TDGP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)conv3_wino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024););
                if (TDGP_WINO_PERSIST) {
                    const int cus = tdgp_cu_count();
                    const int64_t nitems = (int64_t)(W >> 5) * (H >> 3) * B * cdiv(Cout, 64);
                    TDGP_LAUNCH("conv_wino_kernel", conv3_wino_kernel, dim3((unsigned)min((int64_t)cus, nitems)), dim3(512), lds, s, q);
                } else
                TDGP_LAUNCH("conv_wino_kernel", conv3_wino_kernel, dim3((W >> 5) * (H >> 3) * B, cdiv(Cout, 64)), dim3(512), lds, s, q);
            } else if (k == 3) {
                if (Cout > 64) launch_conv3<3, 2, 2, 2, 2>(c, partial, wl.partial_floats, s);
                else launch_conv3<3, 2, 2, 1, 4>(c, partial, wl.partial_floats, s);
            } else {
                if (Cout > 64) launch_conv3<5, 2, 2, 2, 2>(c, partial, wl.partial_floats, s);
                else launch_conv3<5, 2, 2, 1, 4>(c, partial, wl.partial_floats, s);
            }
        } else if (k == 3) {
            if (Cout > 64) launch_conv<2, 2, 2, 2, 4, 9>(p, partial, wl.partial_floats, s);
            else launch_conv<2, 2, 1, 4, 4, 9>(p, partial, wl.partial_floats, s);
        } else if (k == 5) {
            if (Cout > 64) launch_conv<2, 2, 2, 2, 4, 25>(p, partial, wl.partial_floats, s);
            else launch_conv<2, 2, 1, 4, 4, 25>(p, partial, wl.partial_floats, s);
        } else if (out_layout == 1 && Cout <= 96 && !demodulate && !noise && e.act == 1 && ((H * W) & 3) == 0) {
            RgbParams r;
            r.x = x; r.wp = wp; r.styles = styles; r.e = e;
            r.B = B; r.Cin = Cin; r.Cout = Cout; r.CoutP = pi.CoutP; r.HW = H * W; r.W = W; r.P = (int64_t)B * H * W;
            r.lw = r.lhw = -1;
            if ((W & (W - 1)) == 0 && (H & (H - 1)) == 0 && (int64_t)B * H * W < ((int64_t)1 << 31) - 512) {
                r.lw = 0; while ((1 << r.lw) < W) r.lw++;
                r.lhw = 0; while ((1 << r.lhw) < H * W) r.lhw++;
            }
            r.x_bytes = (uint32_t)((int64_t)B * Cin * H * W * 4); r.wp_bytes = (uint32_t)(pi.wp_floats * 4); r.st_bytes = (uint32_t)((int64_t)B * Cin * 4);
            if (Cout <= 32) launch_torgb<1>(r, s);
            else if (Cout <= 64) launch_torgb<2>(r, s);
            else launch_torgb<3>(r, s);
        } else {
            if (Cout > 64 && Cout <= 96) launch_conv<3, 1, 1, 4, 32, 1>(p, partial, wl.partial_floats, s);
            else if (Cout > 64) launch_conv<2, 2, 2, 2, 16, 1>(p, partial, wl.partial_floats, s);
            else launch_conv<2, 2, 1, 4, 16, 1>(p, partial, wl.partial_floats, s);
        }
    } else {
        // transposed conv, stride 2, UNFLIPPED weights (conv2d_resample.py:108-125) -> parity-planar Z -> FIR + output stage
        const UpPlan pl = up_plan(B, Cin, Cout, H, W);
        UpParams u;
        u.x = x; u.wp = wp; u.styles = styles; u.z = z;
        u.B = B; u.Cin = Cin; u.Cout = Cout; u.CoutP = pi.CoutP; u.H = H; u.W = W; u.G1 = pl.G1; u.GS = pl.GS; u.ksplit = pl.ksplit; u.zslice = pl.zslice;
        u.x_bytes = (uint32_t)((int64_t)B * Cin * H * W * 4); u.wp_bytes = (uint32_t)(pi.wp_floats * 4); u.st_bytes = (uint32_t)((int64_t)B * Cin * 4);
        // opt-in split-bf16 arithmetic (tdgp_set_conv_arith(1)): same Z, one slice
        const int s_blocks = cdiv(B * pl.GS, 128) * cdiv(Cout, 64);
        const bool split = g_conv_arith == 1 && s_blocks >= 256 && styles && (Cin & 15) == 0 && Cin <= 2048 && pl.GS >= 2 * (128 + pl.G1 + 2);
        if (split) {
            Up3sParams q;
            q.x = x; q.wsp = wp + pi.wp_floats + pi.wsq_floats; q.styles = styles; q.z = z;
            q.B = B; q.Cin = Cin; q.Cout = Cout; q.CoutP = pi.CoutP; q.H = H; q.W = W; q.G1 = pl.G1; q.GS = pl.GS; q.zslice = pl.zslice;
            q.x_bytes = u.x_bytes; q.wsp_bytes = (uint32_t)(pi.wsplit_floats * 4);
            const size_t lds = (size_t)(2 * 3 * 3 * 64 * 32 + 3 * 2 * 130 * 32 + (2 * 8 * 256 + 8 * 64) * 4 + 2 * 64 * 4);
            TDGP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)upconv3s_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds););
            TDGP_LAUNCH("upconv_mfma_kernel", upconv3s_mfma_kernel, dim3(cdiv(B * pl.GS, 128), cdiv(Cout, 64)), dim3(256), lds, s, q);
        }
        // (Round 3, measured and not kept: 8-wave blocks -- 128 x 128 for Cout > 64, 64 x 256 below -- so that one staged weight chunk serves twice the
        //  MFMAs: -0.7 % / -1.3 % on the whole step; the staging instructions are not what holds this kernel at 0.65 of the MFMA peak.)
        else if (pl.cfg == 0) launch_upconv<2, 1, 2, 2, true>(u, s);
        else launch_upconv<2, 1, 1, 4, false>(u, s);
        FirParams f;
        f.z = z; f.dcoef = dco; f.noise = noise; f.noise_bstride = noise_bstride; f.bias = bias; f.y = y; f.y16 = nullptr;
        f.noise_vec = noise && (((uintptr_t)noise) & 15) == 0 && (noise_bstride & 3) == 0;
        for (int i = 0; i < 16; i++) f.fir[i] = e.fir[i];
        f.B = B; f.C = Cout; f.ZROWS = 2 * H + 2; f.P2 = 2 * pl.G1; f.GS2 = 2 * pl.GS; f.ksplit = split ? 1 : pl.ksplit; f.zslice = pl.zslice;
        f.OH = 2 * H; f.OW = 2 * W;
        f.act = act; f.alpha = alpha; f.gain = gain; f.clamp = clamp;
        if (TDGP_FIR_ADJ && f.OW >= 128 && (f.OH & 31) == 0 && f.ksplit == 1 && f.act == 3 && f.alpha >= 0.f && f.alpha <= 1.f) {
            const int64_t ntiles = (int64_t)B * Cout * cdiv(f.OH, 32) * cdiv(f.OW, 128);
            TDGP_LAUNCH("fir_act_kernel", (fir_act_kernel<32, 128, false, true, true>), dim3((int)min((int64_t)(256 * 32), ntiles)), dim3(256), 0, s, f);
        } else if (f.OW >= 128 && !TDGP_AB_FIR_SERIAL) {
            const int64_t ntiles = (int64_t)B * Cout * cdiv(f.OH, 16) * cdiv(f.OW, 128);
            if (f.act == 3 && f.alpha >= 0.f && f.alpha <= 1.f)
                TDGP_LAUNCH("fir_act_kernel", (fir_act_kernel<16, 128, false, true>), dim3((int)min((int64_t)(256 * 32), ntiles)), dim3(256), 0, s, f);
            else
                TDGP_LAUNCH("fir_act_kernel", (fir_act_kernel<16, 128>), dim3((int)min((int64_t)(256 * 32), ntiles)), dim3(256), 0, s, f);
        } else {
            const int64_t ntiles = (int64_t)B * Cout * cdiv(f.OH, 32) * cdiv(f.OW, 64);
            TDGP_LAUNCH("fir_act_kernel", (fir_act_kernel<32, 64>), dim3((int)min((int64_t)(256 * 32), ntiles)), dim3(256), 0, s, f);
        }
    }
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

// Reduced-precision blocks (BASELINE configs[4], modconv_bf16.inc): x is bf16 NCHW; y is bf16 NCHW (3x3 layers) or, for the ToRGB
// form (k = 1, channel-last planes + fused skip), fp32.  Forms the bf16 MFMA kernels take: 3x3 with Cin % 32 == 0 and, for up = 1,
// W % 32 == 0; the channel-last ToRGB.  Anything else returns TDGP_EUNSUPPORTED (the caller widens to fp32 and uses tdgp_modconv2d).
TDGP_API int tdgp_modconv2d_bf16(const void* x, const void* wpack, const float* styles, const float* dcoef_in, const float* noise, int64_t noise_bstride,
                                 const float* bias, const float* fir4x4, const float* skip, void* y, int B, int Cin, int Cout, int H, int W, int k, int up,
                                 int demodulate, int act, float alpha, float gain, float clamp, int out_layout, int out_feat, void* workspace,
                                 int64_t workspace_bytes, tdgp_stream_t stream) {
    TDGP_CHECK(x && wpack && y, TDGP_EINVAL, "modconv2d_bf16: null pointer");
    TDGP_CHECK(B >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1, TDGP_EINVAL, "modconv2d_bf16: bad shape");
    TDGP_CHECK(act >= 1 && act <= 9, TDGP_EUNSUPPORTED, "modconv2d_bf16: unknown activation %d", act);
    TDGP_CHECK(!demodulate || styles, TDGP_EINVAL, "modconv2d_bf16: demodulate needs styles");
    TDGP_CHECK((int64_t)B * Cin * H * W < ((int64_t)1 << 30) && (int64_t)B * Cout * (H * up + 1) * (W * up + 1) <= INT32_MAX, TDGP_EINVAL,
               "modconv2d_bf16: tensor too large (activations are addressed through 4 GiB buffer descriptors)");
    const bool rgb = k == 1 && up == 1 && out_layout == 1 && Cout <= 96 && !demodulate && !noise && act == 1 && ((H * W) & 3) == 0 && out_feat >= 4 &&
                     (out_feat % 4) == 0 && (Cout % out_feat) == 0 && (!skip || (fir4x4 && (H % 2) == 0 && (W % 2) == 0));
    const bool c3 = k == 3 && up == 1 && out_layout == 0 && !skip && (W & 31) == 0 && (Cin & 31) == 0 && Cin <= 2048;
    bool u3 = k == 3 && up == 2 && out_layout == 0 && !skip && fir4x4 && (Cin & 31) == 0 && (W & 1) == 0;
    size_t u3_lds = 0;
    if (u3) {                                     // every acceptance test before the first launch: the x2 kernel keeps the styles of all samples a block's
        const UpPlan pl0 = up_plan(B, Cin, Cout, H, W);                                   // grid points can touch in LDS
        const int nsb0 = std::min(B, (130 + pl0.G1) / pl0.GS + 2);
        u3_lds = (size_t)(9 * 2 * 64 * 32 + 2 * 2 * 130 * 32) + (size_t)nsb0 * Cin * 4;
        TDGP_CHECK(u3_lds <= 80 * 1024, TDGP_EUNSUPPORTED, "modconv2d_bf16: x2 layer with Cin=%d, B=%d, H=%d needs %zu bytes of LDS (> 80 KiB)", Cin, B, H, u3_lds);
    }
    TDGP_CHECK(rgb || c3 || u3, TDGP_EUNSUPPORTED, "modconv2d_bf16: no bf16 kernel for k=%d up=%d Cin=%d W=%d layout=%d", k, up, Cin, W, out_layout);
    const bool nz16 = !noise || ((((uintptr_t)noise) & 15) == 0 && (noise_bstride & 3) == 0);
    TDGP_CHECK(nz16, TDGP_EINVAL, "modconv2d_bf16: noise must be 16-byte aligned with a batch stride that is a multiple of 4 floats");
    const WsLayout wl = ws_layout(B, Cin, Cout, H, W, k, up);
    TDGP_CHECK(workspace && workspace_bytes >= wl.total, TDGP_EWORKSPACE, "modconv2d_bf16: workspace %lld < %lld bytes", (long long)workspace_bytes,
               (long long)wl.total);
    hipStream_t s = (hipStream_t)stream;
    const PackInfo pi = pack_info(Cout, Cin, k);
    const float* wp = (const float*)wpack;
    const float* wsq = wp + pi.wp_floats;
    const void* wb = wp + pi.wp_floats + pi.wsq_floats + pi.wsplit_floats;
    float* dco = (float*)((char*)workspace + wl.dco);
    float* z = (float*)((char*)workspace + wl.z);
    if (demodulate && dcoef_in) dco = const_cast<float*>(dcoef_in);
    else if (demodulate) TDGP_LAUNCH("demod_kernel", demod_kernel, dim3(cdiv(Cout, 64), B), dim3(1024), 0, s, styles, wsq, dco, B, Cin, Cout, pi.CoutP);
    else dco = nullptr;
    EpiParams e;
    e.B = B; e.Cout = Cout; e.round_bf16 = 1;
    for (int i = 0; i < 16; i++) e.fir[i] = 0.f;
    if (fir4x4)
        for (int ky = 0; ky < 4; ky++)
            for (int kx = 0; kx < 4; kx++) e.fir[ky * 4 + kx] = fir4x4[(3 - ky) * 4 + (3 - kx)] * 4.0f;
    e.dcoef = dco; e.noise = noise; e.noise_bstride = noise_bstride; e.bias = bias; e.skip = skip; e.y = (float*)y;
    e.Hout = H * up; e.Wout = W * up; e.out_layout = out_layout; e.out_feat = out_feat > 0 ? out_feat : 1;
    e.act = act; e.alpha = alpha; e.gain = gain; e.clamp = clamp;
    const uint32_t x_bytes = (uint32_t)((int64_t)B * Cin * H * W * 2);
    if (rgb) {
        RgbParams r;
        r.x = (const float*)x; r.wp = wp; r.styles = styles; r.e = e;
        r.B = B; r.Cin = Cin; r.Cout = Cout; r.CoutP = pi.CoutP; r.HW = H * W; r.W = W; r.P = (int64_t)B * H * W;
        r.lw = r.lhw = -1;
        if ((W & (W - 1)) == 0 && (H & (H - 1)) == 0 && (int64_t)B * H * W < ((int64_t)1 << 31) - 512) {
            r.lw = 0; while ((1 << r.lw) < W) r.lw++;
            r.lhw = 0; while ((1 << r.lhw) < H * W) r.lhw++;
        }
        r.x_bytes = x_bytes; r.wp_bytes = (uint32_t)(pi.wp_floats * 4); r.st_bytes = (uint32_t)((int64_t)B * Cin * 4);
        if (Cout <= 32) launch_torgb<1, true>(r, s);
        else if (Cout <= 64) launch_torgb<2, true>(r, s);
        else launch_torgb<3, true>(r, s);
    } else if (c3) {
        ConvBfParams q;
        q.x = (const uint16_t*)x; q.wb = wb; q.styles = styles; q.e = e; q.y16 = (uint16_t*)y;
        q.B = B; q.Cin = Cin; q.Cout = Cout; q.CoutP = pi.CoutP; q.H = H; q.W = W;
        q.x_bytes = x_bytes; q.wb_bytes = (uint32_t)(pi.wbf_floats * 4);
        const size_t lds = (size_t)(9 * 2 * 64 * 32 + 2 * 10 * 34 * 32 + 5 * 64 * 4 + 2 * Cin * 4);
        TDGP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)conv3_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds - 2 * Cin * 4 + 2 * 2048 * 4)););
        TDGP_LAUNCH("conv_bf16_kernel", conv3_bf16_kernel<true>, dim3((W >> 5) * cdiv(B * (H + 1), 8), cdiv(Cout, 64)), dim3(256), lds, s, q);
    } else {
        const UpPlan pl = up_plan(B, Cin, Cout, H, W);
        UpBfParams q;
        q.x = (const uint16_t*)x; q.wb = wb; q.styles = styles; q.z = z;
        q.B = B; q.Cin = Cin; q.Cout = Cout; q.CoutP = pi.CoutP; q.H = H; q.W = W; q.G1 = pl.G1; q.GS = pl.GS; q.zslice = pl.zslice;
        q.x_bytes = x_bytes; q.wb_bytes = (uint32_t)(pi.wbf_floats * 4); q.st_bytes = (uint32_t)((int64_t)B * Cin * 4);
        const size_t lds = u3_lds;                                              // checked against the 80 KiB budget above
        TDGP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)upconv_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024););
        TDGP_LAUNCH("upconv_bf16_kernel", upconv_bf16_kernel, dim3(cdiv(B * pl.GS, 128), cdiv(Cout, 64)), dim3(256), lds, s, q);
        FirParams f;
        f.z = z; f.dcoef = dco; f.noise = noise; f.noise_bstride = noise_bstride; f.bias = bias; f.y = nullptr; f.y16 = (uint16_t*)y;
        f.noise_vec = noise != nullptr;          // alignment checked at entry
        for (int i = 0; i < 16; i++) f.fir[i] = e.fir[i];
        f.B = B; f.C = Cout; f.ZROWS = 2 * H + 2; f.P2 = 2 * pl.G1; f.GS2 = 2 * pl.GS; f.ksplit = 1; f.zslice = pl.zslice;
        f.OH = 2 * H; f.OW = 2 * W;
        f.act = act; f.alpha = alpha; f.gain = gain; f.clamp = clamp;
        if (f.OW >= 128) {
            const int64_t ntiles = (int64_t)B * Cout * cdiv(f.OH, 16) * cdiv(f.OW, 128);
            if (f.act == 3 && f.alpha >= 0.f && f.alpha <= 1.f)
                TDGP_LAUNCH("fir_act_kernel", (fir_act_kernel<16, 128, true, true>), dim3((int)min((int64_t)(256 * 32), ntiles)), dim3(256), 0, s, f);
            else
                TDGP_LAUNCH("fir_act_kernel", (fir_act_kernel<16, 128, true>), dim3((int)min((int64_t)(256 * 32), ntiles)), dim3(256), 0, s, f);
        } else {
            const int64_t ntiles = (int64_t)B * Cout * cdiv(f.OH, 32) * cdiv(f.OW, 64);
            TDGP_LAUNCH("fir_act_kernel", (fir_act_kernel<32, 64, true>), dim3((int)min((int64_t)(256 * 32), ntiles)), dim3(256), 0, s, f);
        }
    }
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

TDGP_API int tdgp_cast_f32_bf16(const float* x, void* y_bf16, int64_t n, tdgp_stream_t stream) {
    TDGP_CHECK((x && y_bf16) || n == 0, TDGP_EINVAL, "cast_f32_bf16: null pointer");
    if (n <= 0) return TDGP_OK;
    TDGP_LAUNCH("cast_f32_bf16_kernel", cast_f32_bf16_kernel, dim3((int)min((int64_t)8192, cdiv64(n, 512))), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)y_bf16, n);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

TDGP_API int tdgp_conv_transpose2d_x2(const float* x, const void* wpack, const float* styles, float* y, int B, int Cin, int Cout, int H, int W,
                                      void* workspace, int64_t workspace_bytes, tdgp_stream_t stream) {
    TDGP_CHECK(x && wpack && y, TDGP_EINVAL, "conv_transpose2d_x2: null pointer");
    TDGP_CHECK(B >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1, TDGP_EINVAL, "conv_transpose2d_x2: bad shape");
    TDGP_CHECK((int64_t)B * Cin * H * W < ((int64_t)1 << 30) && (int64_t)B * Cout * (H * 2 + 1) * (W * 2 + 1) <= INT32_MAX, TDGP_EINVAL,
               "conv_transpose2d_x2: tensor too large (activations are addressed through 4 GiB buffer descriptors)");
    const WsLayout wl = ws_layout(B, Cin, Cout, H, W, 3, 2);
    TDGP_CHECK(workspace && workspace_bytes >= wl.total, TDGP_EWORKSPACE, "conv_transpose2d_x2: workspace %lld < %lld bytes", (long long)workspace_bytes,
               (long long)wl.total);
    hipStream_t s = (hipStream_t)stream;
    const PackInfo pi = pack_info(Cout, Cin, 3);
    float* z = (float*)((char*)workspace + wl.z);
    const UpPlan pl = up_plan(B, Cin, Cout, H, W);
    UpParams u;
    u.x = x; u.wp = (const float*)wpack; u.styles = styles; u.z = z;
    u.B = B; u.Cin = Cin; u.Cout = Cout; u.CoutP = pi.CoutP; u.H = H; u.W = W; u.G1 = pl.G1; u.GS = pl.GS; u.ksplit = pl.ksplit; u.zslice = pl.zslice;
    u.x_bytes = (uint32_t)((int64_t)B * Cin * H * W * 4); u.wp_bytes = (uint32_t)(pi.wp_floats * 4); u.st_bytes = (uint32_t)((int64_t)B * Cin * 4);
    if (pl.cfg == 0) launch_upconv<2, 1, 2, 2, true>(u, s);
    else launch_upconv<2, 1, 1, 4, false>(u, s);
    const int64_t rows = (int64_t)B * Cout * (2 * H + 1);
    TDGP_LAUNCH("z_gather_kernel", z_gather_kernel, dim3((unsigned)min((int64_t)65535 * 8, rows)), dim3(256), 0, s, (const float*)z, y, rows, 2 * H + 1, 2 * W + 1,
                2 * pl.G1, 2 * pl.GS, pl.ksplit, pl.zslice);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

TDGP_API int64_t tdgp_modconv_wsq_offset(int Cout, int Cin, int k) {
    if (Cout < 1 || Cin < 1 || (k != 1 && k != 3 && k != 5)) return -1;
    return pack_info(Cout, Cin, k).wp_floats * (int64_t)sizeof(float);
}

TDGP_API int tdgp_demod_batch(const float* styles_all, const int64_t* meta, float* dcoef_all, int B, int num_layers, int max_cout, tdgp_stream_t stream) {
    TDGP_CHECK(styles_all && meta && dcoef_all, TDGP_EINVAL, "demod_batch: null pointer");
    TDGP_CHECK(B >= 1 && num_layers >= 1 && max_cout >= 1, TDGP_EINVAL, "demod_batch: bad shape");
    TDGP_LAUNCH("demod_kernel", demod_batch_kernel, dim3(cdiv(max_cout, 64), B, num_layers), dim3(1024), 0, (hipStream_t)stream, styles_all, meta, dcoef_all, B);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

TDGP_API int tdgp_style_affine(const float* ws, const float* A, const float* abias, const int32_t* row_meta, const float* row_scale,
                               float* styles, int B, int num_ws, int w_dim, int rows_total, tdgp_stream_t stream) {
    TDGP_CHECK(ws && A && abias && row_meta && row_scale && styles, TDGP_EINVAL, "style_affine: null pointer");
    TDGP_CHECK(B >= 1 && num_ws >= 1 && w_dim >= 1 && rows_total >= 1, TDGP_EINVAL, "style_affine: bad shape");
    const int64_t waves = (int64_t)B * rows_total;
    const float wgain = (float)(1.0 / sqrt((double)w_dim));
    TDGP_LAUNCH("style_affine_kernel", style_affine_kernel, dim3((int)cdiv64(waves, 4)), dim3(256), 0, (hipStream_t)stream, ws, A, abias, row_meta, row_scale, styles, B,
                       num_ws, w_dim, rows_total, wgain);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}
