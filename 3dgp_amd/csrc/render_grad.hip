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
    const dim3 grid((unsigned)cdiv64(rays, 64)), block(64);
    hipStream_t s = (hipStream_t)stream;
    if (C == 3) TDGP_LAUNCH("ray_march_grad_kernel", ray_march_grad_kernel<3>, grid, block, 0, s, p);
    else if (C == 1) TDGP_LAUNCH("ray_march_grad_kernel", ray_march_grad_kernel<1>, grid, block, 0, s, p);
    else if (C == 4) TDGP_LAUNCH("ray_march_grad_kernel", ray_march_grad_kernel<4>, grid, block, 0, s, p);
    else TDGP_CHECK(false, TDGP_EUNSUPPORTED, "ray_march_grad: C=%d (1, 3 or 4 colour channels)", C);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}
