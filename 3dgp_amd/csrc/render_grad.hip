// render_grad.hip -- gradients of the volume-rendering quadrature.
//
// Reference: autograd through src/training/tri_plane_renderer.py:353-398 (ClassicalRayMarcher) and :299-349 (MipRayMarcher2).
// Per ray, with interval quantities (classical: the S samples themselves; mip: midpoints of neighbouring samples, plus the
// last sample when use_inf_depth):
//     sigma_i = softplus(s_i [+ bias]) | relu(s_i)        a_i = 1 - exp(-delta_i sigma_i)        q_i = 1 - a_i + 1e-10
//     T_i = prod_{j<i} q_j                                 w_i = a_i T_i                          rgb = sum w_i c_i, depth = sum w_i z_i
// Given G_i = dL/dw_i (= d_rgb . c_i + d_depth z_i + d_weights_i, with the white-back / last-back terms folded in):
//     dL/dc_i = w_i d_rgb
//     dL/da_i = T_i (G_i - U_i),        U_i = sum_{k>i} G_k a_k prod_{i<j<k} q_j   (backward recurrence U_i = G_{i+1} a_{i+1} + q_{i+1} U_{i+1}:
//                                        no division by q_i, which is 1e-10 behind an opaque sample)
//     dL/dsigma_i = dL/da_i delta_i (1 - a_i),   dL/ds_i = dL/dsigma_i softplus'(s_i)
// One thread per ray: a forward sweep keeps a_i and T_i in thread-private arrays (scratch memory is lane-interleaved, so the
// sweeps stay coalesced), a backward sweep produces the gradients.  Depths carry no gradient (sample positions are data).
#include "common.h"

namespace {

constexpr int RG_MAXS = 256;

struct MarchGradParams {
    const float* colors;      // [rays,S,C]
    const float* dens;        // [rays,S]
    const float* depths;      // [rays,S]
    const float* d_rgb;       // [rays,C]
    const float* d_depth;     // [rays] or null
    const float* d_weights;   // [rays,M] or null
    float* d_colors;          // [rays,S,C]
    float* d_dens;            // [rays,S]
    int64_t rays;
    int S, C, marcher, flags;
    float density_bias;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int C>
__global__ __launch_bounds__(64) void ray_march_grad_kernel(MarchGradParams p) {
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r >= p.rays) return;
    const int S = p.S;
    const bool mip = p.marcher == 1, inf = p.flags & 1, last_back = (p.flags & 2) && !mip, white = (p.flags & 4) && mip, relu = p.flags & 8;
    const int M = mip ? (inf ? S : S - 1) : S;              // intervals
    const float* col = p.colors + r * S * C;
    const float* den = p.dens + r * S;
    const float* dep = p.depths + r * S;
    float drgb[C];
#pragma unroll
    for (int c = 0; c < C; c++) drgb[c] = p.d_rgb[r * C + c] * (mip ? 2.0f : 1.0f);      // mip: rgb * 2 - 1
    const float ddep = p.d_depth ? p.d_depth[r] : 0.f;
    float drgb_sum = 0.f;
#pragma unroll
    for (int c = 0; c < C; c++) drgb_sum += drgb[c];

    float a[RG_MAXS], T[RG_MAXS];
    // ---- forward sweep ------------------------------------------------------------------------------------------------------
    float Tc = 1.f, wsum = 0.f;
    for (int i = 0; i < M; i++) {
        const bool tail = i == S - 1;                                           // the appended far interval
        const float delta = tail ? (inf ? 1e10f : 1e-3f) : dep[i + 1] - dep[i];
        float s = mip ? (tail ? den[i] : (den[i] + den[i + 1]) * 0.5f) + p.density_bias : den[i];
        const float sigma = relu ? fmaxf(s, 0.f) : softplus20(s);
        const float ai = 1.0f - expf(-delta * sigma);
        a[i] = ai; T[i] = Tc;
        wsum += ai * Tc;
        Tc *= (1.0f - ai) + 1e-10f;
    }
    // ---- backward sweep -----------------------------------------------------------------------------------------------------
    // G_i = dL/dw_i; last_back: w'_{S-1} = 1 - sum_{j<S-1} w_j, so G_j -= G_{S-1} (and the last weight itself has no gradient);
    // white_back: rgb += 1 - sum w  ->  G_i -= sum_c d_rgb_c
    auto g_of = [&](int i) {
        const bool tail = i == S - 1;
        float g = 0.f;
        if (mip) {
#pragma unroll
            for (int c = 0; c < C; c++) g += drgb[c] * (tail ? col[i * C + c] : (col[i * C + c] + col[(i + 1) * C + c]) * 0.5f);
            g += ddep * (tail ? dep[i] : (dep[i] + dep[i + 1]) * 0.5f);
            if (white) g -= drgb_sum;
        } else {
#pragma unroll
            for (int c = 0; c < C; c++) g += drgb[c] * col[i * C + c];
            g += ddep * dep[i];
        }
        if (p.d_weights) g += p.d_weights[r * M + i];
        return g;
    };
    const float g_last = last_back ? g_of(S - 1) : 0.f;
    const float w_last_extra = last_back ? 1.0f - wsum : 0.f;
    for (int i = 0; i < S; i++) {
        p.d_dens[r * S + i] = 0.f;
#pragma unroll
        for (int c = 0; c < C; c++) p.d_colors[(r * S + i) * C + c] = 0.f;
    }
    float U = 0.f;                     // U_i of the interval being visited
    for (int i = M - 1; i >= 0; i--) {
        const bool tail = i == S - 1;
        float G = g_of(i);
        if (last_back) G = (i == S - 1) ? 0.f : G - g_last;
        const float ai = a[i], Ti = T[i];
        const float w = ai * Ti + ((last_back && i == S - 1) ? w_last_extra : 0.f);
        // colours
        if (mip) {
#pragma unroll
            for (int c = 0; c < C; c++) {
                const float dc = w * drgb[c];
                if (tail) p.d_colors[(r * S + i) * C + c] += dc;
                else { p.d_colors[(r * S + i) * C + c] += 0.5f * dc; p.d_colors[(r * S + i + 1) * C + c] += 0.5f * dc; }
            }
        } else {
#pragma unroll
            for (int c = 0; c < C; c++) p.d_colors[(r * S + i) * C + c] = w * drgb[c];
        }
        // densities
        const float da = Ti * (G - U);
        const float delta = tail ? (inf ? 1e10f : 1e-3f) : dep[i + 1] - dep[i];
        const float dsigma = da * delta * (1.0f - ai);
        const float s = mip ? (tail ? den[i] : (den[i] + den[i + 1]) * 0.5f) + p.density_bias : den[i];
        const float ds = relu ? (s > 0.f ? dsigma : 0.f) : (s > 20.f ? dsigma : dsigma * sigmoidf_(s));
        if (mip && !tail) { p.d_dens[r * S + i] += 0.5f * ds; p.d_dens[r * S + i + 1] += 0.5f * ds; }
        else p.d_dens[r * S + i] += ds;
        U = G * ai + ((1.0f - ai) + 1e-10f) * U;            // U_{i-1}
    }
}


// The same gradients with one WAVE per ray (round 4; the thread-per-ray kernel above walks S samples with a stride of S floats between
// lanes and keeps two S-long arrays in scratch: 5.6 ms for 262 k rays x 128 samples, 0.6 GB of traffic).  lane l owns the EPL consecutive
// intervals i = l EPL + e.  T_i is an exclusive prefix product (in-lane, then a 6-step wave scan of the lane totals); the backward
// recurrence U_{i-1} = G_i a_i + q_i U_i is a suffix composition of affine maps u -> B + A u (in-lane, then a 6-step wave scan of (A, B)
// pairs: (A1, B1) o (A2, B2) = (A1 A2, B1 + A1 B2)); the half-and-half deposits of the mip marcher take the left neighbour's interval from
// the previous element / lane.  Same formulas as above; only the association of the products / sums differs (tests: the reference's
// autograd goldens and the double-precision oracle, same bounds as before).  Loads and stores are EPL (x C) consecutive floats per lane.
template <int C, int EPL>
__global__ __launch_bounds__(256) void ray_march_grad_wave_kernel(MarchGradParams p) {
    const int l = lane_id();
    const int S = p.S;
    const bool mip = p.marcher == 1, inf = p.flags & 1, last_back = (p.flags & 2) && !mip, white = (p.flags & 4) && mip, relu = p.flags & 8;
    const int M = mip ? (inf ? S : S - 1) : S;              // intervals
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r = wave; r < p.rays; r += nwaves) {
        const float* col = p.colors + r * S * C;
        const float* den = p.dens + r * S;
        const float* dep = p.depths + r * S;
        float drgb[C], drgb_sum = 0.f;
#pragma unroll
        for (int c = 0; c < C; c++) { drgb[c] = p.d_rgb[r * C + c] * (mip ? 2.0f : 1.0f); drgb_sum += drgb[c]; }
        const float ddep = p.d_depth ? p.d_depth[r] : 0.f;
        // samples base .. base + EPL of this lane (the extra one closes its last interval), clamped to the ray
        const int base = l * EPL;
        float dn[EPL + 1], dp[EPL + 1], cl[EPL + 1][C];
#pragma unroll
        for (int e = 0; e <= EPL; e++) {
            const int j = min(base + e, S - 1);
            dn[e] = den[j]; dp[e] = dep[j];
#pragma unroll
            for (int c = 0; c < C; c++) cl[e][c] = col[j * C + c];
        }
        auto g_at = [&](const float (&c0)[C], const float (&c1)[C], float z0, float z1, bool tail, float dw) {
            float g = 0.f;
            if (mip) {
#pragma unroll
                for (int c = 0; c < C; c++) g += drgb[c] * (tail ? c0[c] : (c0[c] + c1[c]) * 0.5f);
                g += ddep * (tail ? z0 : (z0 + z1) * 0.5f);
                if (white) g -= drgb_sum;
            } else {
#pragma unroll
                for (int c = 0; c < C; c++) g += drgb[c] * c0[c];
                g += ddep * z0;
            }
            return g + dw;
        };
        float g_last = 0.f;
        if (last_back) {
            float cs[C];
#pragma unroll
            for (int c = 0; c < C; c++) cs[c] = col[(S - 1) * C + c];
            g_last = g_at(cs, cs, dep[S - 1], dep[S - 1], true, p.d_weights ? p.d_weights[r * M + S - 1] : 0.f);
        }
        // ---- per interval: a, q, G, s ---------------------------------------------------------------------------------------------
        float a[EPL], q[EPL], G[EPL], sv[EPL], dl[EPL];
        bool ok[EPL];
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const int i = base + e;
            ok[e] = i < M;
            const bool tail = i == S - 1;
            const float delta = tail ? (inf ? 1e10f : 1e-3f) : dp[e + 1] - dp[e];
            const float s_ = mip ? (tail ? dn[e] : (dn[e] + dn[e + 1]) * 0.5f) + p.density_bias : dn[e];
            const float sigma = relu ? fmaxf(s_, 0.f) : softplus20(s_);
            const float ai = ok[e] ? 1.0f - expf(-delta * sigma) : 0.f;
            a[e] = ai; q[e] = ok[e] ? (1.0f - ai) + 1e-10f : 1.0f; sv[e] = s_; dl[e] = delta;
            float g = ok[e] ? g_at(cl[e], cl[e + 1], dp[e], dp[e + 1], tail, p.d_weights ? p.d_weights[r * M + min(i, M - 1)] : 0.f) : 0.f;
            if (last_back) g = (tail || !ok[e]) ? 0.f : g - g_last;
            G[e] = g;
        }
        // ---- T_i: exclusive prefix product ------------------------------------------------------------------------------------------
        float T[EPL], run = 1.f;
#pragma unroll
        for (int e = 0; e < EPL; e++) { T[e] = run; run *= q[e]; }
        float inc = run;                                     // inclusive scan of the lane totals
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const float o = __shfl_up(inc, d, 64); if (l >= d) inc *= o; }
        float exc = __shfl_up(inc, 1, 64);
        if (l == 0) exc = 1.f;
#pragma unroll
        for (int e = 0; e < EPL; e++) T[e] *= exc;
        float w_extra = 0.f;
        if (last_back) {
            float ws = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; e++) ws += a[e] * T[e];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) ws += __shfl_xor(ws, d, 64);
            w_extra = 1.0f - ws;
        }
        // ---- U_i: suffix composition of u -> G a + q u ------------------------------------------------------------------------------
        float A = 1.f, Bc = 0.f;                             // this lane's intervals, first applied last: f_first o ... o f_last
#pragma unroll
        for (int e = EPL - 1; e >= 0; e--) { Bc = G[e] * a[e] + q[e] * Bc; A *= q[e]; }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float oA = __shfl_down(A, d, 64), oB = __shfl_down(Bc, d, 64);
            if (l + d < 64) { Bc = Bc + A * oB; A *= oA; }
        }
        float U = __shfl_down(Bc, 1, 64);                    // value behind this lane's last interval: the lanes to the right, applied to 0
        if (l == 63) U = 0.f;
        // ---- gradients per interval, last element first ------------------------------------------------------------------------------
        float wv[EPL], ds[EPL];
#pragma unroll
        for (int e = EPL - 1; e >= 0; e--) {
            const int i = base + e;
            const bool tail = i == S - 1;
            wv[e] = ok[e] ? a[e] * T[e] + ((last_back && tail) ? w_extra : 0.f) : 0.f;
            const float da = T[e] * (G[e] - U);
            const float dsigma = da * dl[e] * (1.0f - a[e]);
            const float d_ = relu ? (sv[e] > 0.f ? dsigma : 0.f) : (sv[e] > 20.f ? dsigma : dsigma * sigmoidf_(sv[e]));
            ds[e] = ok[e] ? d_ : 0.f;
            U = G[e] * a[e] + q[e] * U;
        }
        // ---- deposits per SAMPLE j = base + e: classical: its own interval; mip: half of interval j (all of it for the far one) + half of j - 1
        float wl = __shfl_up(wv[EPL - 1], 1, 64), dsl = __shfl_up(ds[EPL - 1], 1, 64);     // the left neighbour's last interval (never the far one)
        if (l == 0) { wl = 0.f; dsl = 0.f; }
        float od[EPL], oc[EPL][C];
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const int j = base + e;
            const bool tail = j == S - 1;
            float wj, dj;
            if (mip) {
                const float wp = e == 0 ? wl : wv[e - 1], dpv = e == 0 ? dsl : ds[e - 1];
                wj = (tail ? wv[e] : 0.5f * wv[e]) + 0.5f * wp;
                dj = (tail ? ds[e] : 0.5f * ds[e]) + 0.5f * dpv;
            } else { wj = wv[e]; dj = ds[e]; }
            od[e] = dj;
#pragma unroll
            for (int c = 0; c < C; c++) oc[e][c] = mip ? ((tail ? wv[e] : 0.5f * wv[e]) * drgb[c]) + (0.5f * (e == 0 ? wl : wv[e - 1])) * drgb[c] : wj * drgb[c];
            (void)wj;
        }
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            const int j = base + e;
            if (j < S) {
                p.d_dens[r * S + j] = od[e];
#pragma unroll
                for (int c = 0; c < C; c++) p.d_colors[(r * S + j) * C + c] = oc[e][c];
            }
        }
    }
}

}  // namespace

TDGP_API int tdgp_ray_march_grad(const float* colors, const float* densities, const float* depths, const float* d_rgb, const float* d_depth,
                                 const float* d_weights, float* d_colors, float* d_densities, int64_t rays, int S, int C, int marcher, int flags,
                                 float density_bias, tdgp_stream_t stream) {
    TDGP_CHECK(colors && densities && depths && d_rgb && d_colors && d_densities, TDGP_EINVAL, "ray_march_grad: null pointer");
    TDGP_CHECK(S >= 2 && S <= RG_MAXS, TDGP_EUNSUPPORTED, "ray_march_grad: S=%d outside [2,%d]", S, RG_MAXS);
    TDGP_CHECK(marcher == 0 || marcher == 1, TDGP_EINVAL, "ray_march_grad: unknown ray marcher %d", marcher);
    if (rays == 0) return TDGP_OK;
    MarchGradParams p;
    p.colors = colors; p.dens = densities; p.depths = depths; p.d_rgb = d_rgb; p.d_depth = d_depth; p.d_weights = d_weights;
    p.d_colors = d_colors; p.d_dens = d_densities; p.rays = rays; p.S = S; p.C = C; p.marcher = marcher; p.flags = flags; p.density_bias = density_bias;
    hipStream_t s = (hipStream_t)stream;
    TDGP_CHECK(C == 1 || C == 3 || C == 4, TDGP_EUNSUPPORTED, "ray_march_grad: C=%d (1, 3 or 4 colour channels)", C);
#ifndef TDGP_MARCH_GRAD_WAVE
#define TDGP_MARCH_GRAD_WAVE 1      // 0: the thread-per-ray kernel (A/B builds)
#endif
    if (TDGP_MARCH_GRAD_WAVE) {
        // a wave per ray, 4 waves per block, persistent over the rays (a few blocks per CU's worth of waves cover the load latencies)
        const dim3 grid((unsigned)std::min<int64_t>(cdiv64(rays, 4), (int64_t)tdgp_cu_count() * 8)), block(256);
#define TDGP_MG(CC, EE) TDGP_LAUNCH("ray_march_grad_kernel", (ray_march_grad_wave_kernel<CC, EE>), grid, block, 0, s, p)
#define TDGP_MG_C(EE) do { if (C == 3) TDGP_MG(3, EE); else if (C == 1) TDGP_MG(1, EE); else TDGP_MG(4, EE); } while (0)
        if (S <= 64) TDGP_MG_C(1); else if (S <= 128) TDGP_MG_C(2); else TDGP_MG_C(4);
#undef TDGP_MG_C
#undef TDGP_MG
    } else {
        const dim3 grid((unsigned)cdiv64(rays, 64)), block(64);
        if (C == 3) TDGP_LAUNCH("ray_march_grad_kernel", ray_march_grad_kernel<3>, grid, block, 0, s, p);
        else if (C == 1) TDGP_LAUNCH("ray_march_grad_kernel", ray_march_grad_kernel<1>, grid, block, 0, s, p);
        else TDGP_LAUNCH("ray_march_grad_kernel", ray_march_grad_kernel<4>, grid, block, 0, s, p);
    }
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

// =================================================================================================================================
// Tri-plane field backward: gradients of simple_tri_plane_renderer + TriPlaneMLP (tri_plane_renderer.py:560-588,
// networks_epigraf.py:46-68) w.r.t. the planes (grid_sample backward: a scatter of 4 taps x 3 planes per point) and the four MLP
// tensors.  Forward values are recomputed, nothing is saved by the forward pass.
//
// One wave = tiles of 32 points; lane = (point l32, half).  Per tile, on v_mfma_f32_32x32x2_f32:
//     h_pre [hid x 32] = W0s [hid x F] . g [F x 32]                 (g = mean of the three bilinear samples, staged in LDS)
//     dW0s  [hid x F] += dh_pre [hid x 32] . g^T                     K = the 32 points
//     dW1s^T[hid x 4] += h [hid x 32] . do^T [32 x 4]                (N padded to 32)
//     dg    [F x 32]   = W0s^T [F x hid] . dh_pre [hid x 32]
// with W0s = w0 / sqrt(F), W1s = w1 / sqrt(hid) (layers.py:39-51), h = lrelu(h_pre) sqrt2, o = W1s h + b1, and
// do = d_out (mip: through sigmoid * 1.002 - 0.001), dh = W1s^T do, dh_pre = dh sqrt2 (h > 0 ? 1 : 0.2) on the vector ALU in the
// accumulator layout.  dg / 3 goes to the 12 taps of the point with fp32 atomics (like torch's grid_sampler backward, the
// summation order -- and so the last bits -- of d_planes vary run to run); the weight gradients are accumulated in registers
// over all tiles of a wave, reduced wave -> block -> grid in a fixed order (deterministic).
// =================================================================================================================================
namespace {

typedef float fg_f32x16 __attribute__((ext_vector_type(16)));
constexpr int FG_PITCH = 33;

struct FieldGradParams {
    const float* planes;       // [B,3,H,W,F]
    const float* coords;       // [B,P,3]
    const float* w0; const float* b0; const float* w1; const float* b1;
    const float* d_out;        // [B,P,4]
    float* d_planes;           // [B,3,H,W,F], accumulated into (caller zeroes) or null
    float* d_coords;           // [B,P,3] gradient w.r.t. the sample positions, written; or null
    float* partial;            // [gridDim.x][npart]
    int64_t total, P;
    int F, hid, H, W, marcher, npart;
    float scale, g0, g1;
};

// NW waves per block: 4 for hid <= 32, 2 for hid <= 64 (LDS: ~21.6 KB per wave there; 2 waves keep three blocks on a CU)
template <int MT, int NW>
__global__ __launch_bounds__(64 * NW) void triplane_field_grad_kernel(FieldGradParams p) {
    constexpr int NT = 64 * NW;
    constexpr int HP = 32 * MT;                               // hid padded to whole MFMA tiles
    extern __shared__ __attribute__((aligned(16))) float fg_smem[];
    float* W0s = fg_smem;                                     // [HP][33]   w0 * g0, zero padded
    float* W1s = W0s + HP * FG_PITCH;                         // [4][HP]    w1 * g1
    float* B0 = W1s + 4 * HP;                                 // [HP]
    float* wave_mem = B0 + HP;
    const int tid = threadIdx.x, l = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), l32 = l & 31, half = l >> 5;
    constexpr int WAVE_FLOATS = 32 * FG_PITCH + 2 * HP * FG_PITCH + 4 * 32;
    float* gl = wave_mem + wv * WAVE_FLOATS;                  // [32 pts][33]    features
    float* hl = gl + 32 * FG_PITCH;                           // [HP][33]        h
    float* dl = hl + HP * FG_PITCH;                           // [HP][33]        dh_pre
    float* dol = dl + HP * FG_PITCH;                          // [4][32]         do

    for (int i = tid; i < HP * FG_PITCH; i += NT) {
        const int m = i / FG_PITCH, f = i % FG_PITCH;
        W0s[i] = (m < p.hid && f < p.F) ? p.w0[m * p.F + f] * p.g0 : 0.f;
    }
    for (int i = tid; i < 4 * HP; i += NT) { const int j = i / HP, m = i % HP; W1s[i] = m < p.hid ? p.w1[j * p.hid + m] * p.g1 : 0.f; }
    for (int i = tid; i < HP; i += NT) B0[i] = i < p.hid ? p.b0[i] : 0.f;
    __syncthreads();

    fg_f32x16 dW0a[MT], dW1a[MT];
    float db0a[MT][16], db1a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) { dW0a[mt][r] = 0.f; dW1a[mt][r] = 0.f; db0a[mt][r] = 0.f; }
    const float b1v[4] = {p.b1[0], p.b1[1], p.b1[2], p.b1[3]};
    const float sx = (float)(p.W - 1) / 2.f, sy = (float)(p.H - 1) / 2.f;
    const int fh = p.F / 2;                                   // channels of this lane in the gather: [half * fh, half * fh + fh)
    const float sqrt2 = 1.41421356237309515f;

    const int64_t ntiles = (p.total + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * NW + wv; tile < ntiles; tile += (int64_t)gridDim.x * NW) {
        const int64_t gp = tile * 32 + l32;
        const bool valid = gp < p.total;
        const int64_t gpc = valid ? gp : 0;
        const int b = (int)(gpc / p.P);
        // ---- 1. geometry + gather ------------------------------------------------------------------------------------------
        const float* cp = p.coords + gpc * 3;
        const float q[3] = {cp[0] / p.scale, cp[1] / p.scale, cp[2] / p.scale};
        float tw_[3][4];
        int to_[3][4];
        float fr_[3][2];                                      // (tw, tn): fractional position inside the texel cell, per plane
        int in_[3];                                           // in-range mask of the four taps (bit t), per plane
        float gsum[16];
#pragma unroll
        for (int c = 0; c < 16; c++) gsum[c] = 0.f;
        float acc3[3][16];
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            const float ix = (q[pl == 2 ? 1 : 0] + 1.0f) * sx, iy = (q[pl == 0 ? 1 : 2] + 1.0f) * sy;
            const float fx = floorf(ix), fy = floorf(iy);
            const float twx = ix - fx, te = 1.0f - twx, tn = iy - fy, ts = 1.0f - tn;
            const float cfx = fminf(fmaxf(fx, -2.f), (float)p.W), cfy = fminf(fmaxf(fy, -2.f), (float)p.H);
            const int x0 = (int)cfx, y0 = (int)cfy;
            const float wgt[4] = {ts * te, ts * twx, tn * te, tn * twx};
            const float* plane = p.planes + ((int64_t)b * 3 + pl) * p.H * p.W * p.F;
            fr_[pl][0] = twx; fr_[pl][1] = tn; in_[pl] = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int x = x0 + (t & 1), y = y0 + (t >> 1);
                const bool in = valid && x >= 0 && x < p.W && y >= 0 && y < p.H;
                tw_[pl][t] = in ? wgt[t] : 0.f;
                to_[pl][t] = in ? (y * p.W + x) * p.F : 0;
                in_[pl] |= in ? (1 << t) : 0;
            }
#pragma unroll
            for (int c = 0; c < 16; c++) acc3[pl][c] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float* texel = plane + to_[pl][t] + half * fh;
#pragma unroll
                for (int c4 = 0; c4 < 4; c4++) {
                    if (4 * c4 < fh) {
                        const float4 v = *(const float4*)(texel + 4 * c4);
                        acc3[pl][4 * c4 + 0] += v.x * tw_[pl][t]; acc3[pl][4 * c4 + 1] += v.y * tw_[pl][t];
                        acc3[pl][4 * c4 + 2] += v.z * tw_[pl][t]; acc3[pl][4 * c4 + 3] += v.w * tw_[pl][t];
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 16; c++) gsum[c] = ((acc3[0][c] + acc3[1][c]) + acc3[2][c]) / 3.0f;
        // features of the point: [pt][f], zero beyond F
#pragma unroll
        for (int c = 0; c < 16; c++) {
            if (c < fh) gl[l32 * FG_PITCH + half * fh + c] = gsum[c];
        }
        for (int f = p.F + half; f < 32; f += 2) gl[l32 * FG_PITCH + f] = 0.f;
        const float4 dout4 = valid ? *(const float4*)(p.d_out + gpc * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- 2. h_pre = W0s g + b0 (MFMA), h = lrelu * sqrt2 -> LDS ----------------------------------------------------------
        fg_f32x16 hacc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 16; r++) hacc[mt][r] = B0[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
        for (int ks = 0; ks < 16; ks++) {
            const float bf = gl[l32 * FG_PITCH + 2 * ks + half];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) hacc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(W0s[(mt * 32 + l32) * FG_PITCH + 2 * ks + half], bf, hacc[mt], 0, 0, 0);
        }
        float oj[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float hp = hacc[mt][r];
                const float h = (hp > 0.f ? hp : hp * 0.2f) * sqrt2;
                hacc[mt][r] = h;
                hl[m * FG_PITCH + l32] = h;
#pragma unroll
                for (int j = 0; j < 4; j++) oj[j] = fmaf_(W1s[j * HP + m], h, oj[j]);
            }
        // ---- 3. outputs, incoming gradient ------------------------------------------------------------------------------------
        float dj[4] = {dout4.x, dout4.y, dout4.z, dout4.w};
#pragma unroll
        for (int j = 0; j < 4; j++) oj[j] = (oj[j] + __shfl_xor(oj[j], 32, 64)) + b1v[j];
        if (p.marcher == 1) {
#pragma unroll
            for (int j = 0; j < 3; j++) { const float sg = 1.0f / (1.0f + expf(-oj[j])); dj[j] = dj[j] * 1.002f * sg * (1.0f - sg); }
        }
        if (half == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) { db1a[j] += dj[j]; dol[j * 32 + l32] = dj[j]; }
        }
        // ---- 4. dh_pre in the accumulator layout -> LDS ----------------------------------------------------------------------------
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float dh = 0.f;
#pragma unroll
                for (int j = 0; j < 4; j++) dh = fmaf_(W1s[j * HP + m], dj[j], dh);
                const float dhp = dh * sqrt2 * (hacc[mt][r] > 0.f ? 1.0f : 0.2f);
                db0a[mt][r] += dhp;
                dl[m * FG_PITCH + l32] = dhp;
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- 5. weight-gradient GEMMs over the 32 points; dg = W0s^T dh_pre ----------------------------------------------------
#pragma unroll
        for (int ks = 0; ks < 16; ks++) {
            const int k = 2 * ks + half;                                          // point index of this K step
            const float bg = gl[k * FG_PITCH + l32];                              // B[k = pt][n = f]
            const float bd = l32 < 4 ? dol[l32 * 32 + k] : 0.f;                   // B[k = pt][n = j]
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                dW0a[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(dl[(mt * 32 + l32) * FG_PITCH + k], bg, dW0a[mt], 0, 0, 0);
                dW1a[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(hl[(mt * 32 + l32) * FG_PITCH + k], bd, dW1a[mt], 0, 0, 0);
            }
        }
        fg_f32x16 dg;
#pragma unroll
        for (int r = 0; r < 16; r++) dg[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < HP / 2; ks++) {
            const int k = 2 * ks + half;                                          // hidden unit of this K step
            dg = __builtin_amdgcn_mfma_f32_32x32x2f32(W0s[k * FG_PITCH + l32], dl[k * FG_PITCH + l32], dg, 0, 0, 0);   // A[m = f][k], B[k][n = pt]
        }
        // ---- 6. scatter: d_plane[tap] += w_tap * dg / 3 ------------------------------------------------------------------------
        // Transposed through LDS so that ONE atomic instruction adds the 32 channels of one tap (a 128-B line) per half-wave:
        // issued from the accumulator layout (lane = point) every instruction touched 64 different lines and the kernel ran at
        // the L2's atomic line rate (82 ms for 8.4 M points).  What bounds it now (round 4, tools/dev/ubench_atomics.hip): the chip
        // retires 10.3 G such lines per second = 0.6 lane-atomics per clock and CU -- the same for a 24-MiB and a 1.6-GiB target, with
        // or without sc1 / nt, and with every XCD kept inside its own eighth of the target -- so 12 lines per point are 1.17 ms per
        // million points whatever the access pattern; only fewer lines (an LDS window that merges the taps of neighbouring rays before
        // they leave the CU) can lower it.
        if (p.d_planes || p.d_coords) {
#pragma unroll
            for (int r = 0; r < 16; r++) gl[l32 * FG_PITCH + (r & 3) + 8 * (r >> 2) + 4 * half] = dg[r];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // ---- 6a. gradient w.r.t. the sample position (grid_sampler's grid gradient; what a camera is trained through, loss.py:69-83):
        //     d ix = sum_c dg_c/3 [(ne - nw) ts + (se - sw) tn],   d iy = sum_c dg_c/3 [(sw - nw) te + (se - ne) tw]
        // taps outside the plane read as zero (torch grid_sampler_2d_backward), then x (size - 1) / 2 (align_corners) and / scale.
        // lane = (point, channel half): the taps are fetched again (cache hits: this wave gathered them a moment ago).
        if (p.d_coords) {
            float dq[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                const float* plane = p.planes + ((int64_t)b * 3 + pl) * p.H * p.W * p.F;
                const float twx = fr_[pl][0], tn = fr_[pl][1], te = 1.0f - twx, ts = 1.0f - tn;
                float gix = 0.f, giy = 0.f;
#pragma unroll
                for (int c4 = 0; c4 < 4; c4++) {
                    if (4 * c4 < fh) {
                        float4 tv[4];
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            tv[t] = *(const float4*)(plane + to_[pl][t] + half * fh + 4 * c4);
                            if (!((in_[pl] >> t) & 1)) tv[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                        const float* gq = gl + l32 * FG_PITCH + half * fh + 4 * c4;
                        const float gv[4] = {gq[0], gq[1], gq[2], gq[3]};
                        const float nw[4] = {tv[0].x, tv[0].y, tv[0].z, tv[0].w}, ne[4] = {tv[1].x, tv[1].y, tv[1].z, tv[1].w};
                        const float sw[4] = {tv[2].x, tv[2].y, tv[2].z, tv[2].w}, se[4] = {tv[3].x, tv[3].y, tv[3].z, tv[3].w};
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            gix = fmaf_(gv[i], (ne[i] - nw[i]) * ts + (se[i] - sw[i]) * tn, gix);
                            giy = fmaf_(gv[i], (sw[i] - nw[i]) * te + (se[i] - ne[i]) * twx, giy);
                        }
                    }
                }
                gix += __shfl_xor(gix, 32, 64);
                giy += __shfl_xor(giy, 32, 64);
                dq[pl == 2 ? 1 : 0] += gix * sx;                 // planes (x,y), (x,z), (y,z): width <- first coordinate
                dq[pl == 0 ? 1 : 2] += giy * sy;
            }
            if (half == 0 && valid) {
                const float k = 1.0f / (3.0f * p.scale);
                float* dc = p.d_coords + gpc * 3;
                dc[0] = dq[0] * k; dc[1] = dq[1] * k; dc[2] = dq[2] * k;
            }
        }
        if (p.d_planes) {
            int* tab_off = (int*)hl;                             // [32 pts][12 taps] float offset into d_planes (h / dh_pre are dead now)
            float* tab_w = hl + 32 * 12;
#pragma unroll
            for (int pl = 0; pl < 3; pl++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    if (((pl * 4 + t) & 1) == half) {
                        tab_off[l32 * 12 + pl * 4 + t] = (int)((((int64_t)b * 3 + pl) * p.H * p.W) * p.F) + to_[pl][t];
                        tab_w[l32 * 12 + pl * 4 + t] = tw_[pl][t] / 3.0f;
                    }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int e2 = 0; e2 < 32 * 12 / 2; e2++) {
                const int e = 2 * e2 + half;                    // (point, tap) handled by this half-wave; lane = channel
                const float wt = tab_w[e];
                if (wt != 0.f && l32 < p.F) unsafeAtomicAdd(p.d_planes + tab_off[e] + l32, wt * gl[(e / 12) * FG_PITCH + l32]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // ---- reduction of the weight gradients: lanes -> wave slot in LDS -> block (fixed order) -> partial[blockIdx] ------------
    // layout of a slot / of `partial`: dW0 [hid][F] | db0 [hid] | dW1 [4][hid] | db1 [4]
    __syncthreads();
    float* slot = wave_mem + wv * WAVE_FLOATS;                 // >= npart floats (host check)
    for (int i = l; i < p.npart; i += 64) slot[i] = 0.f;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int o_db0 = p.hid * p.F, o_dw1 = o_db0 + p.hid, o_db1 = o_dw1 + 4 * p.hid;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < p.hid && l32 < p.F) slot[m * p.F + l32] = dW0a[mt][r];                 // C[m][n = f]
            if (m < p.hid && l32 < 4) slot[o_dw1 + l32 * p.hid + m] = dW1a[mt][r];          // C[m][n = j]
            float s = db0a[mt][r];
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 16, 64);
            if (m < p.hid && l32 == 0) slot[o_db0 + m] = s;
        }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float s = db1a[j];
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 16, 64);
        if (l == 0) slot[o_db1 + j] = s;
    }
    __syncthreads();
    for (int i = tid; i < p.npart; i += NT) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w++) s += wave_mem[w * WAVE_FLOATS + i];
        p.partial[(int64_t)blockIdx.x * p.npart + i] = s;
    }
}

// d_w0 = g0 * sum dW0s, d_b0, d_w1 = g1 * sum dW1s, d_b1: blocks summed in block order
__global__ __launch_bounds__(256) void field_grad_reduce_kernel(const float* __restrict__ partial, int nblocks, int npart, int n_w0, int n_b0, int n_w1,
                                                               float g0, float g1, float* __restrict__ d_w0, float* __restrict__ d_b0,
                                                               float* __restrict__ d_w1, float* __restrict__ d_b1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npart) return;
    float s = 0.f;
    for (int k = 0; k < nblocks; k++) s += partial[(int64_t)k * npart + i];
    if (i < n_w0) d_w0[i] = s * g0;
    else if (i < n_w0 + n_b0) d_b0[i - n_w0] = s;
    else if (i < n_w0 + n_b0 + n_w1) d_w1[i - n_w0 - n_b0] = s * g1;
    else d_b1[i - n_w0 - n_b0 - n_w1] = s;
}

int field_grad_blocks(int64_t total) { return (int)min((int64_t)1536, max((int64_t)1, cdiv64(total, 128))); }

}  // namespace

TDGP_API int64_t tdgp_triplane_field_grad_workspace_bytes(int B, int64_t P, int F, int hid) {
    const int npart = hid * F + hid + 4 * hid + 4;
    return (int64_t)field_grad_blocks((int64_t)B * P) * npart * (int64_t)sizeof(float);
}

TDGP_API int tdgp_triplane_field_grad(const float* planes_hwc, const float* coords, const float* w0, const float* b0, const float* w1, const float* b1,
                                      const float* d_out, float* d_planes_hwc, float* d_w0, float* d_b0, float* d_w1, float* d_b1, float* d_coords,
                                      void* workspace, int64_t workspace_bytes, int B, int64_t P, int F, int H, int W, int hid, float scale, int marcher,
                                      tdgp_stream_t stream) {
    TDGP_CHECK(planes_hwc && coords && w0 && b0 && w1 && b1 && d_out && d_w0 && d_b0 && d_w1 && d_b1, TDGP_EINVAL, "triplane_field_grad: null pointer");
    TDGP_CHECK(B >= 1 && P >= 1 && H >= 2 && W >= 2, TDGP_EINVAL, "triplane_field_grad: bad shape");
    TDGP_CHECK(F % 8 == 0 && F >= 8 && F <= 32, TDGP_EUNSUPPORTED, "triplane_field_grad: feat_dim=%d (8, 16, 24 or 32)", F);
    TDGP_CHECK(hid >= 1 && hid <= 64, TDGP_EUNSUPPORTED, "triplane_field_grad: hid_dim=%d > 64", hid);
    TDGP_CHECK(marcher == 0 || marcher == 1, TDGP_EINVAL, "triplane_field_grad: unknown ray marcher %d", marcher);
    TDGP_CHECK((int64_t)B * 3 * H * W * F <= INT32_MAX, TDGP_EINVAL, "triplane_field_grad: plane tensor too large");
    const int64_t need = tdgp_triplane_field_grad_workspace_bytes(B, P, F, hid);
    TDGP_CHECK(workspace && workspace_bytes >= need, TDGP_EINVAL, "triplane_field_grad: workspace of %lld bytes needed", (long long)need);
    FieldGradParams p;
    p.planes = planes_hwc; p.coords = coords; p.w0 = w0; p.b0 = b0; p.w1 = w1; p.b1 = b1; p.d_out = d_out; p.d_planes = d_planes_hwc; p.d_coords = d_coords;
    p.partial = (float*)workspace; p.total = (int64_t)B * P; p.P = P; p.F = F; p.hid = hid; p.H = H; p.W = W; p.marcher = marcher;
    p.npart = hid * F + hid + 4 * hid + 4;
    p.scale = scale; p.g0 = (float)(1.0 / sqrt((double)F)); p.g1 = (float)(1.0 / sqrt((double)hid));
    const int nb = field_grad_blocks(p.total);
    const int MT = hid <= 32 ? 1 : 2, HP = 32 * MT;
    const int wave_floats = 32 * FG_PITCH + 2 * HP * FG_PITCH + 4 * 32;
    TDGP_CHECK(p.npart <= wave_floats, TDGP_EUNSUPPORTED, "triplane_field_grad: reduction slot too small");
    const int NW = MT == 1 ? 4 : 2;
    const size_t lds = (size_t)(HP * FG_PITCH + 4 * HP + HP + NW * wave_floats) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    if (MT == 1) {
        TDGP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)triplane_field_grad_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds););
        TDGP_LAUNCH("triplane_field_grad_kernel", (triplane_field_grad_kernel<1, 4>), dim3(nb), dim3(256), lds, s, p);
    } else {
        TDGP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)triplane_field_grad_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds););
        TDGP_LAUNCH("triplane_field_grad_kernel", (triplane_field_grad_kernel<2, 2>), dim3(nb), dim3(128), lds, s, p);
    }
    TDGP_LAUNCH("field_grad_reduce_kernel", field_grad_reduce_kernel, dim3(cdiv(p.npart, 256)), dim3(256), 0, s, (const float*)workspace, nb, p.npart, hid * F,
                hid, 4 * hid, p.g0, p.g1, d_w0, d_b0, d_w1, d_b1);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}
