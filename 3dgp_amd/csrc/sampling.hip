// sampling.hip -- per-ray stages of the volumetric renderer: stratified samples, ray marchers,
// hierarchical importance resampling, depth merge + alpha compositing.
//
// Replaces ~40 eager PyTorch ops per chunk of src/training/tri_plane_renderer.py:
//   :208-235 sample_stratified      :353-405 ClassicalRayMarcher     :300-348 MipRayMarcher2
//   :237-255 sample_importance      :257-295 sample_pdf              :196-206 unify_samples
//
// Mapping: ONE 64-lane wavefront per ray (4 rays per 256-thread block), samples across lanes, per-wave LDS
// scratch, no block-level synchronisation.
// Arithmetic (r02): the marchers are the reference's own fp32 chain -- delta, softplus = log1p(exp(x)), alpha = 1 - exp(-delta*sigma),
// transmittance = cumprod, weights, weighted sums -- on fp32 OCML transcendentals (<= 1 ulp), fp32 DPP wave scans and fp32 wave
// sums.  r01 evaluated the transcendentals, scans and sums in fp64 to be bit-identical to the CPU oracle (which rounds the exact
// value once); that cost 2-3x the instructions (fp64 exp / log1p, two-register DPP moves, half-rate adds) for an agreement that
// is not the reference's either (torch's expf is Sleef's 1-ulp routine, its sums fp32 cascades).  What has to be EXACT is kept
// exact: sample_pdf's normaliser follows torch's summation order, the cdf is torch's sequential-double cumsum rounded per prefix,
// so for given weights the searchsorted indices and the fine samples equal the reference's bit for bit (the stage-level tests).
#include "common.h"

#ifndef TDGP_RAY_ABL
#define TDGP_RAY_ABL 0      // 1: per-phase cycle counts of one wave, printed (timing experiments only)
#endif

namespace {

constexpr int MAXS = 256;          // max samples per ray in one pass (2*S for the merged pass)
constexpr int RAYS_PER_BLOCK = 4;

// Per-wave LDS scratch.  MS = capacity in samples; the fused kernels pick 128 when the ray fits (4.1 KB per wave -> 8 waves
// per SIMD instead of 4: these kernels are chains of dependent LDS / cross-lane steps, occupancy is what hides them).
template <int MS>
struct alignas(16) WaveScratchT {
    float z[MS];        // depths (s- or t-space)
    float sig[MS];      // densities
    float w[MS + 4];    // weights / pdf scratch
    float cdf[MS];
    float bins[MS];
    float col[3][MS];   // colours (merged pass)
};
using WaveScratch = WaveScratchT<MAXS>;

__device__ __forceinline__ void wave_sync() {
    // all LDS traffic of a wave is issued in order; this only stops the compiler from reordering across it
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float s2t(float s, float t_near, float t_far) { return s * t_far + (1.0f - s) * t_near; }

// ------------------------------------------------------------------------------------------------
// classical marcher on LDS-resident depths/densities: writes weights w[0..S), returns (final_T, sum w).
// tri_plane_renderer.py:355-387.
// flags: bit0 use_inf_depth, bit1 last_back, bit3 relu clamp
// ------------------------------------------------------------------------------------------------
__device__ void march_classical_lds(const float* z, const float* sig, float* w, int S, int flags, float cut_thr, float& final_T, float& wagg) {
    const int l = lane_id();
    double carry = 1.0;
    float wsum = 0.0f;
    if (S > 64 && S <= 128) {
        // Two chunks of 64 samples (the merged 64 + 64 list): both chunks' softplus / exp / prefix products are evaluated SIDE BY SIDE -- two independent
        // dependency chains the scheduler interleaves -- and joined by the one multiplication that couples them.  The loop below runs them one after the
        // other, and these kernels are bound by the length of a wave's dependent chain at the SIMD's 8-wave occupancy.  Same operations on
        // the same values (the first chunk's `* carry` is `* 1.0`): same bits.
        float alpha[2], fac[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int i = 64 * c + l;
            alpha[c] = 0.f; fac[c] = 1.0f;
            if (i < S) {
                const float delta = (i < S - 1) ? (z[i + 1] - z[i]) : ((flags & 1) ? 1e10f : 1e-3f);
                float sp = (flags & 8) ? (sig[i] > 0.f ? sig[i] : 0.f) : softplus20f(sig[i]);
                if (sp < cut_thr) sp = 0.f;
                alpha[c] = 1.0f - expf(-(delta * sp));
                fac[c] = (1.0f - alpha[c]) + 1e-10f;
            }
        }
        const double s0 = wave_scan_f64<true>((double)fac[0]), s1 = wave_scan_f64<true>((double)fac[1]);
        const double c1 = wave_last_f64(s0);
        const double i1 = s1 * c1;
        const float incl0 = (float)s0, incl1 = (float)i1;
        const float wi0 = alpha[0] * wave_shr1_f32(incl0, 1.0f), wi1 = alpha[1] * wave_shr1_f32(incl1, (float)c1);
        w[l] = wi0; wsum += wi0;
        if (64 + l < S) { w[64 + l] = wi1; wsum += wi1; }
        carry = wave_last_f64(i1);
    } else
    for (int base = 0; base < S; base += 64) {
        const int i = base + l;
        float alpha = 0.f, fac = 1.0f;
        if (i < S) {
            float delta = (i < S - 1) ? (z[i + 1] - z[i]) : ((flags & 1) ? 1e10f : 1e-3f);
            float sp = (flags & 8) ? (sig[i] > 0.f ? sig[i] : 0.f) : softplus20f(sig[i]);
            if (sp < cut_thr) sp = 0.f;                        // cut_quantile (:366-368); cut_thr = 0 never cuts (sp >= 0)
            alpha = 1.0f - expf(-(delta * sp));
            fac = (1.0f - alpha) + 1e-10f;
        }
        // transmittance = torch.cumprod on the CPU: prefixes accumulated in fp64, each rounded to fp32.  The fp64 DPP
        // scan (6 steps of 2 moves + 1 v_mul_f64) rounds to the same fp32 prefix as the sequential fp64 product except with probability
        // ~2^-29 per element.  Round 5: the fp32 scan used since round 2 left the composited depth 2.4e-7 (2 ulp) from the reference's
        // float64 image on average where the reference's own fp32 run sits at 0.9e-7 (e2e_full_c3) -- with this it is level.
        const double incl64 = wave_scan_f64<true>((double)fac) * carry;
        const float incl = (float)incl64;
        const float excl = wave_shr1_f32(incl, (float)carry);
        const float wi = alpha * excl;
        if (i < S) { w[i] = wi; wsum += wi; }
        carry = wave_last_f64(incl64);
    }
    wagg = wave_sum_f32(wsum);
    final_T = (float)carry;
    wave_sync();
    if ((flags & 2) && l == 0) w[S - 1] += (1.0f - wagg);
    wave_sync();
}

// mip marcher weights on LDS-resident data: M = S (inf depth) or S-1 mid-point samples.
// tri_plane_renderer.py:305-334.
__device__ void march_mip_lds(const float* z, const float* sig, float* w, int S, int flags, float density_bias, float cut_thr, float& final_T,
                              float& wagg) {
    const int l = lane_id();
    const int M = (flags & 1) ? S : S - 1;
    double carry = 1.0;
    float wsum = 0.0f;
    for (int base = 0; base < M; base += 64) {
        const int i = base + l;
        float alpha = 0.f, fac = 1.0f;
        if (i < M) {
            float delta, smid;
            if (i < S - 1) { delta = z[i + 1] - z[i]; smid = (sig[i] + sig[i + 1]) / 2.f; }
            else { delta = 1e10f; smid = sig[S - 1]; }
            float sp = softplus20f(smid + density_bias);
            if (sp < cut_thr) sp = 0.f;                        // cut_quantile (:324-326)
            float dd = sp * delta;
            alpha = 1.0f - expf(-dd);
            fac = (1.0f - alpha) + 1e-10f;
        }
        const double incl64 = wave_scan_f64<true>((double)fac) * carry;        // fp64 prefixes rounded to fp32: see march_classical_lds
        const float incl = (float)incl64;
        const float excl = wave_shr1_f32(incl, (float)carry);
        const float wi = alpha * excl;
        if (i < M) { w[i] = wi; wsum += wi; }
        carry = wave_last_f64(incl64);
    }
    wagg = wave_sum_f32(wsum);
    final_T = (float)carry;
    wave_sync();
}

// ------------------------------------------------------------------------------------------------
// torch.sum(x, -1) of the reference's CPU path over x[i] = w1[i] + eps, i < n, IN TORCH'S OWN ORDER: the value is
// sample_pdf's normaliser (tri_plane_renderer.py:272) and its last bit decides which cdf knot a draw falls behind,
// i.e. the searchsorted indices.  ATen's sum kernel (cpu/SumKernel.cpp, AVX2 dispatch: 8 fp32 lanes) cuts the row
// into nv = n/8 vectors, adds them into 4 interleaved vector accumulators (acc[k] += V[4i+k]), the nv%4 left-over
// vectors into accumulator 0, then acc0 += acc1, acc2, acc3, and finally a scalar takes the n%8 tail elements in
// order followed by the 8 lanes of acc0 in order (no cascade level is reached below 512 elements; n <= 254 here).
// Rows shorter than 8 run the same scheme on scalars.  Lane (a = (l>>3)&3, col = l&7) plays lane col of accumulator a;
// every lane returns the result.
// ------------------------------------------------------------------------------------------------
__device__ float torch_order_sum_lds(const float* w1, int n, float eps) {
    const int l = lane_id();
    if (n >= 8) {
        const int nv = n >> 3, nilp = nv >> 2;
        const int col = l & 7, a = (l >> 3) & 3;
        float acc = 0.f;
        for (int i = 0; i < nilp; i++) acc = __fadd_rn(acc, __fadd_rn(w1[(4 * i + a) * 8 + col], eps));
        for (int j = 4 * nilp; j < nv; j++) {
            const float v = __fadd_rn(w1[j * 8 + col], eps);
            if (a == 0) acc = __fadd_rn(acc, v);
        }
        float p = __shfl(acc, col, 64);
        p = __fadd_rn(p, __shfl(acc, col + 8, 64));
        p = __fadd_rn(p, __shfl(acc, col + 16, 64));
        p = __fadd_rn(p, __shfl(acc, col + 24, 64));
        float fin = 0.f;
        for (int k = nv * 8; k < n; k++) fin = __fadd_rn(fin, __fadd_rn(w1[k], eps));
#pragma unroll
        for (int c = 0; c < 8; c++) fin = __fadd_rn(fin, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), c)));      // lanes 0..7: scalar reads, no LDS permute
        return fin;
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int nilp = n >> 2;                       // 0 or 1
    if (nilp)
        for (int k = 0; k < 4; k++) acc[k] = __fadd_rn(acc[k], __fadd_rn(w1[k], eps));
    for (int r = 4 * nilp; r < n; r++) acc[0] = __fadd_rn(acc[0], __fadd_rn(w1[r], eps));
    for (int k = 1; k < 4; k++) acc[0] = __fadd_rn(acc[0], acc[k]);
    return acc[0];
}

// ------------------------------------------------------------------------------------------------
// sample_importance + sample_pdf on LDS-resident z[S] (s-space) and weights w[Wn].
// tri_plane_renderer.py:237-295.  Clobbers sc.w / sc.cdf / sc.bins.
// Lane j produces sample j (strided by 64).  `emit(j, sample, ind, below, above)` consumes the results.
// ------------------------------------------------------------------------------------------------
template <typename SC, typename GetU, typename Emit>
__device__ void importance_lds(SC& sc, int S, int Wn, GetU getu, int N, int mip, Emit emit) {
    const int l = lane_id();
    const float eps = 1e-5f;
    float* w = sc.w;
    // smoothed / offset weights, in place (two passes through registers)
    if (mip) {
        // max_pool1d(2,1,pad 1) -> Wn+1, avg_pool1d(2,1) -> Wn, + 0.01
        float nv[MAXS / 64];
#pragma unroll
        for (int c = 0; c < MAXS / 64; c++) {
            const int i = l + 64 * c;
            if (i < Wn) {
                float a = (i - 1 >= 0) ? w[i - 1] : -INFINITY, b = w[i], d = (i + 1 < Wn) ? w[i + 1] : -INFINITY;
                float t0 = a > b ? a : b;      // tmp[i]   = max(w[i-1], w[i])
                float t1 = b > d ? b : d;      // tmp[i+1] = max(w[i], w[i+1])
                nv[c] = (t0 + t1) / 2.f + 0.01f;
            }
        }
        wave_sync();
#pragma unroll
        for (int c = 0; c < MAXS / 64; c++) {
            const int i = l + 64 * c;
            if (i < Wn) w[i] = nv[c];
        }
    } else {
        for (int i = l; i < Wn; i += 64) w[i] = w[i] + 1e-5f;
    }
    wave_sync();
    const int nb = S - 1, ns = Wn - 2, nc = ns + 1;
    for (int i = l; i < nb; i += 64) sc.bins[i] = 0.5f * (sc.z[i] + sc.z[i + 1]);
    // pdf normaliser: torch's fp32 accumulation order, not the exactly rounded sum (the integer rows depend on it)
    const float totf = torch_order_sum_lds(w + 1, ns, eps);
    // cdf = [0, cumsum(pdf)]
    double carry = 0.0;
    if (l == 0) sc.cdf[0] = 0.f;
    for (int base = 0; base < ns; base += 64) {
        const int i = base + l;
        float pdf = (i < ns) ? (w[1 + i] + eps) / totf : 0.f;
        double incl = wave_scan_f64<false>((double)pdf) + carry;
        if (i < ns) sc.cdf[i + 1] = (float)incl;
        carry = wave_last_f64(incl);
    }
    wave_sync();
    for (int j = l; j < N; j += 64) {
        const float uu = getu(j);
        int lo = 0, hi = nc;                       // searchsorted(right=True): first index with cdf > u
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (sc.cdf[mid] <= uu) lo = mid + 1; else hi = mid;
        }
        const int ind = lo;
        const int below = ind - 1 < 0 ? 0 : ind - 1;
        const int above = ind > ns ? ns : ind;
        const float cb = sc.cdf[below], ca = sc.cdf[above];
        const float bb = sc.bins[below < nb ? below : nb - 1], ba = sc.bins[above < nb ? above : nb - 1];
        float denom = ca - cb;
        if (denom < eps) denom = 1.f;
        const float smp = bb + (uu - cb) / denom * (ba - bb);
        emit(j, smp, ind, below, above);
    }
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// One stratified sample (tri_plane_renderer.py:225-233): lin = torch.linspace(0, 1, S) on the CPU, one fused multiply-add per element
// (camera_rays.hip:linspace_f).
__device__ __forceinline__ float strat_one(float u, int k, int S, float step, int marcher) {
    auto lin = [&](int q) { return (S == 1) ? 0.f : ((q < S / 2) ? __fmaf_rn(step, (float)q, 0.f) : __fmaf_rn(-step, (float)(S - 1 - q), 1.f)); };
    if (marcher == 0) {
        const float lower = (k == 0) ? lin(0) : 0.5f * (lin(k) + lin(k - 1));
        const float upper = (k == S - 1) ? lin(S - 1) : 0.5f * (lin(k + 1) + lin(k));
        return lower + (upper - lower) * u;
    }
    const float delta = (float)((1.0 - 0.0) / (double)(S - 1));
    return lin(k) + u * delta;
}

__global__ __launch_bounds__(256) void stratified_kernel(const float* __restrict__ u, float* __restrict__ sdist, float* __restrict__ tdist,
                                                        int64_t n, int S, int marcher, float t_near, float t_far) {
    const float step = (1.f - 0.f) / (float)(S - 1);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float s = strat_one(u[i], (int)(i % S), S, step, marcher);
        sdist[i] = s;
        if (tdist) tdist[i] = s2t(s, t_near, t_far);
    }
}

// The same, four consecutive samples of a ray per thread as 16-byte loads / stores (S % 4 == 0, 16-byte aligned pointers: the generator's shapes).
// Round 5: the scalar form moved 4 bytes per lane and instruction and spent a 64-bit modulo per sample -- 4.3 of the 6.3 TB/s a copy reaches.
__global__ __launch_bounds__(256) void stratified4_kernel(const float4* __restrict__ u, float4* __restrict__ sdist, float4* __restrict__ tdist,
                                                         int64_t n4, int S, int marcher, float t_near, float t_far) {
    const float step = (1.f - 0.f) / (float)(S - 1);
    const int S4 = S >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % S4) * 4;
        const float4 uu = u[i];
        const float4 s = make_float4(strat_one(uu.x, k, S, step, marcher), strat_one(uu.y, k + 1, S, step, marcher), strat_one(uu.z, k + 2, S, step, marcher),
                                     strat_one(uu.w, k + 3, S, step, marcher));
        sdist[i] = s;
        if (tdist) tdist[i] = make_float4(s2t(s.x, t_near, t_far), s2t(s.y, t_near, t_far), s2t(s.z, t_near, t_far), s2t(s.w, t_near, t_far));
    }
}

__global__ __launch_bounds__(256) void density_activation_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int relu, float bias) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = relu ? (x[i] > 0.f ? x[i] : 0.f) : softplus20f(x[i] + bias);
}

// generic marcher: colours [rays,S,C], densities [rays,S], depths [rays,S]
__global__ __launch_bounds__(256) void ray_march_kernel(const float* __restrict__ colors, const float* __restrict__ dens,
                                                       const float* __restrict__ depths, float* __restrict__ rgb, float* __restrict__ depth_o,
                                                       float* __restrict__ weights, float* __restrict__ final_T, int64_t rays, int S, int C,
                                                       int marcher, int flags, float density_bias, float cut_thr) {
    __shared__ WaveScratch scratch[RAYS_PER_BLOCK];
    const int wv = threadIdx.x >> 6, l = lane_id();
    const int64_t r = (int64_t)blockIdx.x * RAYS_PER_BLOCK + wv;
    if (r >= rays) return;
    WaveScratch& sc = scratch[wv];
    for (int i = l; i < S; i += 64) { sc.z[i] = depths[r * S + i]; sc.sig[i] = dens[r * S + i]; }
    wave_sync();
    float fT, wagg;
    const int M = (marcher == 0) ? S : ((flags & 1) ? S : S - 1);
    if (marcher == 0) march_classical_lds(sc.z, sc.sig, sc.w, S, flags, cut_thr, fT, wagg);
    else march_mip_lds(sc.z, sc.sig, sc.w, S, flags, density_bias, cut_thr, fT, wagg);
    if (weights) for (int i = l; i < M; i += 64) weights[r * M + i] = sc.w[i];
    // composite: sum_i w_i * c_i
    for (int c = 0; c <= C; c++) {          // c == C: depth
        float acc = 0.0f;
        for (int i = l; i < M; i += 64) {
            float v;
            if (c < C) {
                v = colors[(r * S + i) * C + c];
                if (marcher == 1 && i < S - 1) v = (v + colors[(r * S + i + 1) * C + c]) / 2.f;
            } else {
                v = sc.z[i];
                if (marcher == 1 && i < S - 1) v = (v + sc.z[i + 1]) / 2.f;
            }
            acc += sc.w[i] * v;
        }
        float out = wave_sum_f32(acc);
        if (marcher == 1 && c < C) {
            if (flags & 4) out = out + 1.0f - wagg;
            out = out * 2.0f - 1.0f;
        }
        if (l == 0) { if (c < C) rgb[r * C + c] = out; else depth_o[r] = out; }
    }
    if (l == 0) final_T[r] = fT;
}

__global__ __launch_bounds__(256) void sample_importance_kernel(const float* __restrict__ z, const float* __restrict__ weights,
                                                               const float* __restrict__ u, float* __restrict__ samples, int32_t* __restrict__ inds,
                                                               int32_t* __restrict__ below, int32_t* __restrict__ above, float* __restrict__ cdf_o,
                                                               int64_t rays, int S, int Wn, int N, int mip) {
    __shared__ WaveScratch scratch[RAYS_PER_BLOCK];
    const int wv = threadIdx.x >> 6, l = lane_id();
    const int64_t r = (int64_t)blockIdx.x * RAYS_PER_BLOCK + wv;
    if (r >= rays) return;
    WaveScratch& sc = scratch[wv];
    for (int i = l; i < S; i += 64) sc.z[i] = z[r * S + i];
    for (int i = l; i < Wn; i += 64) sc.w[i] = weights[r * Wn + i];
    wave_sync();
    importance_lds(sc, S, Wn, [&](int j) { return u[r * N + j]; }, N, mip, [&](int j, float smp, int ind, int bl, int ab) {
        samples[r * N + j] = smp;
        if (inds) { inds[r * N + j] = ind; below[r * N + j] = bl; above[r * N + j] = ab; }
    });
    if (cdf_o) for (int i = l; i < Wn - 1; i += 64) cdf_o[r * (Wn - 1) + i] = sc.cdf[i];
}

// stable rank of every element of key[0..M) under (key, index) order -> rank[]; brute force from LDS broadcasts.
// NS = ceil(M / 64) value slots per lane actually in use (compile time, so no work is spent on empty slots).
template <int NS>
__device__ __forceinline__ void stable_ranks_n(const float* key, int M, int* rank) {
    const int l = lane_id();
    float k[NS];
#pragma unroll
    for (int c = 0; c < NS; c++) { const int i = l + 64 * c; k[c] = i < M ? key[i] : 0.f; rank[c] = 0; }
#pragma unroll 4
    for (int m = 0; m < M; m++) {
        const float km = key[m];                   // same address on every lane: LDS broadcast
#pragma unroll
        for (int c = 0; c < NS; c++) {
            const int i = l + 64 * c;
            rank[c] += (km < k[c] || (km == k[c] && m < i)) ? 1 : 0;
        }
    }
}
__device__ __forceinline__ void stable_ranks(const float* key, int M, int* rank /* per-lane, MAXS/64 entries */) {
#pragma unroll
    for (int c = 0; c < MAXS / 64; c++) rank[c] = 0;
    if (M <= 64) stable_ranks_n<1>(key, M, rank);
    else if (M <= 128) stable_ranks_n<2>(key, M, rank);
    else if (M <= 192) stable_ranks_n<3>(key, M, rank);
    else stable_ranks_n<4>(key, M, rank);
}

// Bitonic sort of one (key, index) pair per lane across the 64 lanes of a wave, ascending by (key, index): 21 compare-exchange
// stages, each two ds_bpermute (LDS pipe) + ~5 VALU -- a third of the vector work of the brute-force rank above for 64 keys.
// Indices are unique, so the order is total and the result is the stable sort.
__device__ __forceinline__ void wave_bitonic_sort(float& key, int& idx) {
    const int l = lane_id();
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const float pk = __shfl_xor(key, j, 64);
            const int pi = __shfl_xor(idx, j, 64);
            const bool keep_min = ((l & j) == 0) == ((l & k) == 0);        // lower lane of an ascending pair / upper lane of a descending one
            const bool partner_less = pk < key || (pk == key && pi < idx);
            if (partner_less == keep_min) { key = pk; idx = pi; }
        }
    }
}

// The same network over 128 (key, index) pairs, two per lane (element e = 64 c + lane): 28 stages, the j = 64 stage is a compare-exchange
// inside the lane.  For 64 < N <= 128 fine samples (BASELINE configs[4]: 96) instead of the brute-force rank (N broadcasts x 2 slots).
__device__ __forceinline__ void wave_bitonic_sort2(float (&key)[2], int (&idx)[2]) {
    const int l = lane_id();
#pragma unroll
    for (int k = 2; k <= 128; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j == 64) {                                  // k = 128: ascending everywhere; slot 0 keeps the smaller
                const bool less1 = key[1] < key[0] || (key[1] == key[0] && idx[1] < idx[0]);
                if (less1) { const float tk = key[0]; key[0] = key[1]; key[1] = tk; const int ti = idx[0]; idx[0] = idx[1]; idx[1] = ti; }
            } else {
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const float pk = __shfl_xor(key[c], j, 64);
                    const int pi = __shfl_xor(idx[c], j, 64);
                    const bool asc = k == 128 ? true : (k == 64 ? c == 0 : ((l & k) == 0));     // (e & k) == 0: bit 6 of e is the slot, bit 7 is never set
                    const bool keep_min = ((l & j) == 0) == asc;
                    const bool partner_less = pk < key[c] || (pk == key[c] && pi < idx[c]);
                    if (partner_less == keep_min) { key[c] = pk; idx[c] = pi; }
                }
            }
        }
    }
}

// fused: coarse march (s-space) -> importance sampling -> fine depths (t-space), WRITTEN IN ASCENDING DEPTH ORDER.
// The reference leaves the fine samples in draw order and sorts coarse+fine together later (unify_samples); sorting the
// fine list here (stable, by (t, draw index)) changes nothing in that final order but makes the j-th fine sample of
// neighbouring rays neighbours in space (texture locality of the second field pass) and turns the later merge into a
// merge of two sorted lists.  fine_perm[pos] = draw index of the sample stored at pos.
template <int MS>
__global__ __launch_bounds__(256) void importance_from_coarse_kernel(const float* __restrict__ rgbs, const float* __restrict__ sdist,
                                                                    const float* __restrict__ u_fine, float* __restrict__ tfine,
                                                                    float* __restrict__ sfine, int32_t* __restrict__ inds,
                                                                    int32_t* __restrict__ fine_perm, int64_t rays, int S, int N,
                                                                    int marcher, int flags, float density_bias, float cut_thr, float t_near, float t_far) {
    __shared__ WaveScratchT<MS> scratch[RAYS_PER_BLOCK];
    const int wv = threadIdx.x >> 6, l = lane_id();
    const int64_t r = (int64_t)blockIdx.x * RAYS_PER_BLOCK + wv;
    if (r >= rays) return;
    WaveScratchT<MS>& sc = scratch[wv];
#if TDGP_RAY_ABL & 1
    long long tph[6], t0_ = __builtin_readcyclecounter();
#define TPH(i) { const long long t_ = __builtin_readcyclecounter(); tph[i] = t_ - t0_; t0_ = t_; }
#else
#define TPH(i)
#endif
    for (int i = l; i < S; i += 64) { sc.z[i] = sdist[r * S + i]; sc.sig[i] = rgbs[(r * S + i) * 4 + 3]; }
    // the draws are not needed before the cdf exists: their loads go out now and land during the march
    float upre[MS / 64];
#pragma unroll
    for (int c = 0; c < MS / 64; c++) upre[c] = (l + 64 * c < N) ? u_fine[r * N + l + 64 * c] : 0.f;
    wave_sync();
    TPH(0)
    float fT, wagg;
    int Wn = S;
    if (marcher == 0) march_classical_lds(sc.z, sc.sig, sc.w, S, flags, cut_thr, fT, wagg);
    else { march_mip_lds(sc.z, sc.sig, sc.w, S, flags, density_bias, cut_thr, fT, wagg); Wn = (flags & 1) ? S : S - 1; }
    TPH(1)
    float* tkey = sc.col[0];
    importance_lds(sc, S, Wn, [&](int j) { float v = upre[0];
#pragma unroll
                                          for (int c = 1; c < MS / 64; c++) v = (j >> 6) == c ? upre[c] : v;
                                          return v; }, N, marcher, [&](int j, float smp, int ind, int, int) {
        tkey[j] = s2t(smp, t_near, t_far);
        if (sfine) sfine[r * N + j] = smp;
        if (inds) inds[r * N + j] = ind;
    });
    wave_sync();
    TPH(2)
    if (N <= 64) {                  // one sample per lane: sort across the lanes, store coalesced
        float key = l < N ? tkey[l] : INFINITY;
        // Distinct keys (all but measure-zero rays): the sorted slot of a key is the number of smaller keys -- 16 broadcast LDS reads
        // and 128 compare / add-carry instructions with no dependent chain, where the network below is 21 stages of two dependent
        // cross-lane permutes each (it was 30 % of this kernel's time).  Equal keys collide on a slot, which the read-back sees: those
        // rays take the network, whose (key, draw index) order is the stable sort.
        {
            if (l >= N) tkey[l] = INFINITY;
            wave_sync();
            int cnt = 0;
#pragma unroll
            for (int jj = 0; jj < 16; jj++) {
                const float4 k4 = *(const float4*)&tkey[4 * jj];
                cnt += (k4.x < key ? 1 : 0) + (k4.y < key ? 1 : 0) + (k4.z < key ? 1 : 0) + (k4.w < key ? 1 : 0);
            }
            int* slot = (int*)sc.bins;                      // free since importance_lds returned
            if (l < N) slot[cnt] = l;
            wave_sync();
            const bool mine = l >= N || slot[cnt] == l;
            if (__all(mine)) {
                if (l < N) {
                    const int src = slot[l];
                    tfine[r * N + l] = tkey[src];
                    if (fine_perm) fine_perm[r * N + l] = src;
                }
                return;
            }
        }
        int idx = l;
        wave_bitonic_sort(key, idx);
        TPH(3)
        if (l < N) {
            tfine[r * N + l] = key;
            if (fine_perm) fine_perm[r * N + l] = idx;
        }
        TPH(4)
#if TDGP_RAY_ABL & 1
        if (l == 0 && (r == 1000 || r == 200000)) printf("importance ray %lld: load %lld march %lld importance %lld sort %lld store %lld\n", (long long)r, tph[0], tph[1], tph[2], tph[3], tph[4]);
#endif
        return;
    }
    if (N <= 128) {                 // two samples per lane
        float key[2] = {tkey[l], l + 64 < N ? tkey[l + 64] : INFINITY};
        {                           // distinct keys: slot = number of smaller keys, as above (MS >= 128 entries behind tkey)
            if (l + 64 >= N) tkey[l + 64] = INFINITY;
            wave_sync();
            int cnt[2] = {0, 0};
#pragma unroll 8
            for (int jj = 0; jj < 32; jj++) {
                const float4 k4 = *(const float4*)&tkey[4 * jj];
#pragma unroll
                for (int c = 0; c < 2; c++)
                    cnt[c] += (k4.x < key[c] ? 1 : 0) + (k4.y < key[c] ? 1 : 0) + (k4.z < key[c] ? 1 : 0) + (k4.w < key[c] ? 1 : 0);
            }
            int* slot = (int*)sc.bins;
            slot[cnt[0]] = l;
            if (l + 64 < N) slot[cnt[1]] = l + 64;
            wave_sync();
            const bool mine = slot[cnt[0]] == l && (l + 64 >= N || slot[cnt[1]] == l + 64);
            if (__all(mine)) {
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const int pos = l + 64 * c;
                    if (pos < N) {
                        const int src = slot[pos];
                        tfine[r * N + pos] = tkey[src];
                        if (fine_perm) fine_perm[r * N + pos] = src;
                    }
                }
                return;
            }
        }
        int idx[2] = {l, l + 64};
        wave_bitonic_sort2(key, idx);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int pos = l + 64 * c;
            if (pos < N) {
                tfine[r * N + pos] = key[c];
                if (fine_perm) fine_perm[r * N + pos] = idx[c];
            }
        }
        return;
    }
    int rank[MAXS / 64];
    stable_ranks(tkey, N, rank);
#pragma unroll
    for (int c = 0; c < MAXS / 64; c++) {
        const int j = l + 64 * c;
        if (j < N) {
            tfine[r * N + rank[c]] = tkey[j];
            if (fine_perm) fine_perm[r * N + rank[c]] = j;
        }
    }
}

__global__ __launch_bounds__(256) void unify_kernel(const float* __restrict__ d1, const float* __restrict__ c1, const float* __restrict__ s1, int S1,
                                                   const float* __restrict__ d2, const float* __restrict__ c2, const float* __restrict__ s2, int S2,
                                                   float* __restrict__ d, float* __restrict__ c, float* __restrict__ s, int32_t* __restrict__ perm,
                                                   int64_t rays, int C) {
    __shared__ WaveScratch scratch[RAYS_PER_BLOCK];
    const int wv = threadIdx.x >> 6, l = lane_id();
    const int64_t r = (int64_t)blockIdx.x * RAYS_PER_BLOCK + wv;
    if (r >= rays) return;
    WaveScratch& sc = scratch[wv];
    const int M = S1 + S2;
    for (int i = l; i < M; i += 64) sc.z[i] = i < S1 ? d1[r * S1 + i] : d2[r * S2 + (i - S1)];
    wave_sync();
    int rank[MAXS / 64];
    stable_ranks(sc.z, M, rank);
#pragma unroll
    for (int cc = 0; cc < MAXS / 64; cc++) {
        const int i = l + 64 * cc;
        if (i >= M) continue;
        const int pos = rank[cc];
        d[r * M + pos] = sc.z[i];
        s[r * M + pos] = i < S1 ? s1[r * S1 + i] : s2[r * S2 + (i - S1)];
        for (int k = 0; k < C; k++) c[(r * M + pos) * C + k] = i < S1 ? c1[(r * S1 + i) * C + k] : c2[(r * S2 + (i - S1)) * C + k];
        if (perm) perm[r * M + pos] = i;
    }
}

// fused: merge coarse + fine by depth (stable), march in t-space, composite.
template <int MS>
__global__ __launch_bounds__(256) void merge_composite_kernel(const float* __restrict__ rgbs1, const float* __restrict__ t1, int S1,
                                                             const float* __restrict__ rgbs2, const float* __restrict__ t2, int S2,
                                                             float* __restrict__ rgb, float* __restrict__ depth_o, float* __restrict__ wsum_o,
                                                             float* __restrict__ final_T, int32_t* __restrict__ perm, const int32_t* __restrict__ perm2, int64_t rays,
                                                             int marcher, int flags, float density_bias, float cut_thr) {
    __shared__ WaveScratchT<MS> scratch[RAYS_PER_BLOCK];
    const int wv = threadIdx.x >> 6, l = lane_id();
    const int64_t r = (int64_t)blockIdx.x * RAYS_PER_BLOCK + wv;
    if (r >= rays) return;
    WaveScratchT<MS>& sc = scratch[wv];
    const int M = S1 + S2;
#if TDGP_RAY_ABL & 1
    long long tph[6], t0_ = __builtin_readcyclecounter();
#endif
    // keys -> sc.cdf (scratch), ranks, scatter into sorted sc.z / sc.sig / sc.col.  The colours do not depend on the ranks: their
    // loads go out with the keys' and land while the ranks are searched.
    float4 cval[MS / 64];
#pragma unroll
    for (int cc = 0; cc < MS / 64; cc++) {
        const int i = l + 64 * cc;
        if (i < M) {
            sc.cdf[i] = i < S1 ? t1[r * S1 + i] : t2[r * S2 + (i - S1)];
            cval[cc] = i < S1 ? ((const float4*)rgbs1)[r * S1 + i] : ((const float4*)rgbs2)[r * S2 + (i - S1)];
        }
    }
    wave_sync();
    // both lists already ascending (stratified coarse samples; fine samples sorted by importance_from_coarse)?  Then the
    // stable merge position is the element's own index plus a binary search in the OTHER list; otherwise (arbitrary caller
    // data, or the ~1e-10-probability ulp inversion of s -> t) fall back to the brute-force stable rank.
    TPH(0)
    bool bad = false;
    for (int i = l; i < M; i += 64)
        if (i + 1 < M && i + 1 != S1 && sc.cdf[i + 1] < sc.cdf[i]) bad = true;
    int rank[MAXS / 64];
    if (__any(bad)) {
        stable_ranks(sc.cdf, M, rank);
    } else {
        // Ranks of a stable merge of two ascending lists (coarse wins ties).
        //   fine element j:    its own index + #coarse <= f_j  =: j + cnt_j     -- a binary search in the coarse list;
        //   coarse element i:  its own index + #fine < c_i.  No second search: c_i <= f_j  <=>  cnt_j >= i + 1, so #fine < c_i = #{j : cnt_j <= i}, and
        //                      cnt_j does not decrease with j, so that count is (the largest j with cnt_j <= i) + 1 = a PREFIX MAXIMUM over
        //                      last[c] = max{j + 1 : cnt_j = c} -- one LDS max-scatter by the fine elements and one wave scan.
        // (Round 5.  Before, every element searched the other list: 7 lock-step rounds of ~14 vector instructions on both slots of a lane, a third of
        //  this kernel's instructions; now the slots that hold only coarse elements skip the search.)  The searches of a lane's slots still advance
        //  together, a fixed number of branch-free steps: one chain of dependent LDS reads instead of one per slot.
        constexpr int NSL = MS / 64;
        float v[NSL];
        int lo[NSL], hi[NSL];
        bool fine[NSL];
        int* const last = (int*)sc.bins;                 // [S1 + 1] (S1 + 1 <= MS: the list lengths are checked on the host)
        for (int c = l; c <= S1 && c < MS; c += 64) last[c] = 0;        // (c == MS only when there is no fine list at all)
#pragma unroll
        for (int cc = 0; cc < NSL; cc++) {
            const int i = l + 64 * cc;
            v[cc] = i < M ? sc.cdf[i] : 0.f;
            fine[cc] = i >= S1 && i < M;
            lo[cc] = 0;
            hi[cc] = fine[cc] ? S1 : 0;                  // coarse elements (and slots past the end) never open a search
        }
        const int steps = 32 - __clz(S1);
        for (int it = 0; it < steps; it++) {
#pragma unroll
            for (int cc = 0; cc < NSL; cc++) {           // #coarse <= v (coarse wins ties)
                if (64 * cc + 64 <= S1) continue;        // a slot of coarse elements only (wave-uniform)
                const bool open = lo[cc] < hi[cc];
                const int mid = (lo[cc] + hi[cc]) >> 1;
                const float o = sc.cdf[open ? mid : 0];
                const bool right = o <= v[cc];
                lo[cc] = (open && right) ? mid + 1 : lo[cc];
                hi[cc] = (open && !right) ? mid : hi[cc];
            }
        }
        wave_sync();                                     // last[] is zeroed before the scatter below
#pragma unroll
        for (int cc = 0; cc < NSL; cc++) {
            const int i = l + 64 * cc;
            if (fine[cc]) atomicMax(&last[lo[cc]], i - S1 + 1);
        }
        wave_sync();
#pragma unroll
        for (int cc = 0; cc < MAXS / 64; cc++) rank[cc] = 0;
        int carry = 0;
#pragma unroll
        for (int cc = 0; cc < NSL; cc++) {
            const int i = l + 64 * cc;
            if (64 * cc < S1) {                          // this slot holds coarse elements: F(i) = max over c <= i of last[c]
                const int incl = max(wave_scan_max_i32((i <= S1 && i < MS) ? last[i] : 0), carry);
                carry = wave_last_i32(incl);
                rank[cc] = fine[cc] ? (i - S1) + lo[cc] : i + incl;
            } else {
                rank[cc] = (i - S1) + lo[cc];
            }
        }
    }
    TPH(1)
#pragma unroll
    for (int cc = 0; cc < MS / 64; cc++) {
        const int i = l + 64 * cc;
        if (i >= M) continue;
        const int pos = rank[cc];
        const float4 v = cval[cc];
        sc.z[pos] = sc.cdf[i];
        sc.col[0][pos] = v.x; sc.col[1][pos] = v.y; sc.col[2][pos] = v.z; sc.sig[pos] = v.w;
        if (perm) perm[r * M + pos] = i < S1 ? i : S1 + (perm2 ? perm2[r * S2 + (i - S1)] : i - S1);
    }
    wave_sync();
    TPH(2)
    float fT, wagg;
    const int Mm = (marcher == 0) ? M : ((flags & 1) ? M : M - 1);
    if (marcher == 0) march_classical_lds(sc.z, sc.sig, sc.w, M, flags, cut_thr, fT, wagg);
    else march_mip_lds(sc.z, sc.sig, sc.w, M, flags, density_bias, cut_thr, fT, wagg);
    TPH(3)
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, wacc = 0.f;
    for (int i = l; i < Mm; i += 64) {
        const float wi = sc.w[i];
        wacc += wi;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float v = c < 3 ? sc.col[c][i] : sc.z[i];
            if (marcher == 1 && i < M - 1) v = (v + (c < 3 ? sc.col[c][i + 1] : sc.z[i + 1])) / 2.f;
            acc[c] += wi * v;
        }
    }
    float out[4];
#pragma unroll
    for (int c = 0; c < 4; c++) out[c] = wave_sum_f32(acc[c]);
    const float wtot = wave_sum_f32(wacc);            // weights.sum(2): final weights incl. last_back
    if (marcher == 1) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (flags & 4) out[c] = out[c] + 1.0f - wagg;
            out[c] = out[c] * 2.0f - 1.0f;
        }
    }
    TPH(4)
#if TDGP_RAY_ABL & 1
    if (l == 0 && (r == 1000 || r == 200000)) printf("merge ray %lld: load %lld rank %lld gather %lld march %lld composite %lld\n", (long long)r, tph[0], tph[1], tph[2], tph[3], tph[4]);
#endif
    if (l == 0) {
        rgb[r * 3 + 0] = out[0]; rgb[r * 3 + 1] = out[1]; rgb[r * 3 + 2] = out[2];
        depth_o[r] = out[3];
        if (wsum_o) wsum_o[r] = wtot;
        if (final_T) final_T[r] = fT;
    }
}

inline int ray_blocks(int64_t rays) { return (int)cdiv64(rays, RAYS_PER_BLOCK); }

}  // namespace

TDGP_API int tdgp_sample_stratified(const float* u, float* sdist, float* tdist, int64_t rays, int S, int marcher, float t_near,
                                    float t_far, tdgp_stream_t stream) {
    TDGP_CHECK(u && sdist, TDGP_EINVAL, "sample_stratified: null pointer");
    TDGP_CHECK(S >= 2 && rays >= 0, TDGP_EINVAL, "sample_stratified: need S >= 2");
    TDGP_CHECK(marcher == 0 || marcher == 1, TDGP_EINVAL, "sample_stratified: unknown ray marcher %d", marcher);
    const int64_t n = rays * S;
    if (n == 0) return TDGP_OK;
    if ((S & 3) == 0 && (((uintptr_t)u | (uintptr_t)sdist | (uintptr_t)tdist) & 15) == 0) {
        TDGP_LAUNCH("stratified_kernel", stratified4_kernel, dim3((int)min((int64_t)8192, cdiv64(n / 4, 256))), dim3(256), 0, (hipStream_t)stream, (const float4*)u,
                    (float4*)sdist, (float4*)tdist, n / 4, S, marcher, t_near, t_far);
        TDGP_LAUNCH_CHECK();
        return TDGP_OK;
    }
    TDGP_LAUNCH("stratified_kernel", stratified_kernel, dim3((int)min((int64_t)8192, cdiv64(n, 256))), dim3(256), 0, (hipStream_t)stream, u, sdist, tdist, n, S,
                       marcher, t_near, t_far);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

TDGP_API int tdgp_density_activation(const float* sigma, float* out, int64_t n, int flags, float density_bias, tdgp_stream_t stream) {
    TDGP_CHECK(sigma && out, TDGP_EINVAL, "density_activation: null pointer");
    TDGP_CHECK(n >= 0, TDGP_EINVAL, "density_activation: negative length");
    if (n == 0) return TDGP_OK;
    TDGP_LAUNCH("density_activation_kernel", density_activation_kernel, dim3((int)min((int64_t)8192, cdiv64(n, 256))), dim3(256), 0, (hipStream_t)stream, sigma, out, n,
                (flags & 8) ? 1 : 0, density_bias);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

TDGP_API int tdgp_ray_march(const float* colors, const float* densities, const float* depths, float* rgb, float* depth, float* weights,
                            float* final_T, int64_t rays, int S, int C, int marcher, int flags, float density_bias, float cut_threshold, tdgp_stream_t stream) {
    TDGP_CHECK(colors && densities && depths && rgb && depth && final_T, TDGP_EINVAL, "ray_march: null pointer");
    TDGP_CHECK(S >= 2 && S <= MAXS, TDGP_EUNSUPPORTED, "ray_march: S=%d outside [2,%d]", S, MAXS);
    TDGP_CHECK(C >= 1 && C <= 8, TDGP_EUNSUPPORTED, "ray_march: C=%d outside [1,8]", C);
    TDGP_CHECK(marcher == 0 || marcher == 1, TDGP_EINVAL, "ray_march: unknown ray marcher %d", marcher);
    TDGP_FAULT_CHECK("ray_march");
    if (rays == 0) return TDGP_OK;
    TDGP_LAUNCH("ray_march_kernel", ray_march_kernel, dim3(ray_blocks(rays)), dim3(256), 0, (hipStream_t)stream, colors, densities, depths, rgb, depth, weights,
                       final_T, rays, S, C, marcher, flags, density_bias, cut_threshold);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

TDGP_API int tdgp_sample_importance(const float* z, const float* weights, const float* u, float* samples, int32_t* inds, int32_t* below,
                                    int32_t* above, float* cdf, int64_t rays, int S, int Wn, int N, int marcher, tdgp_stream_t stream) {
    TDGP_CHECK(z && weights && u && samples, TDGP_EINVAL, "sample_importance: null pointer");
    TDGP_CHECK(!inds || (below && above), TDGP_EINVAL, "sample_importance: inds/below/above come together");
    TDGP_CHECK(S >= 4 && S <= MAXS && Wn >= 3 && Wn <= S && N >= 1, TDGP_EUNSUPPORTED, "sample_importance: bad S=%d Wn=%d N=%d", S, Wn, N);
    if (rays == 0) return TDGP_OK;
    TDGP_LAUNCH("sample_importance_kernel", sample_importance_kernel, dim3(ray_blocks(rays)), dim3(256), 0, (hipStream_t)stream, z, weights, u, samples, inds, below,
                       above, cdf, rays, S, Wn, N, marcher);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

TDGP_API int tdgp_unify_samples(const float* d1, const float* c1, const float* s1, int S1, const float* d2, const float* c2, const float* s2,
                                int S2, float* d, float* c, float* s, int32_t* perm, int64_t rays, int C, tdgp_stream_t stream) {
    TDGP_CHECK(d1 && c1 && s1 && d2 && c2 && s2 && d && c && s, TDGP_EINVAL, "unify_samples: null pointer");
    TDGP_CHECK(S1 >= 1 && S2 >= 1 && S1 + S2 <= MAXS, TDGP_EUNSUPPORTED, "unify_samples: S1+S2=%d > %d", S1 + S2, MAXS);
    if (rays == 0) return TDGP_OK;
    TDGP_LAUNCH("unify_kernel", unify_kernel, dim3(ray_blocks(rays)), dim3(256), 0, (hipStream_t)stream, d1, c1, s1, S1, d2, c2, s2, S2, d, c, s, perm, rays, C);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

TDGP_API int tdgp_importance_from_coarse(const float* rgbs_coarse, const float* sdist, const float* u_fine, float* tdist_fine,
                                         float* sdist_fine, int32_t* inds, int32_t* fine_perm, int64_t rays, int S, int N, int marcher, int flags,
                                         float density_bias, float cut_threshold, float t_near, float t_far, tdgp_stream_t stream) {
    TDGP_CHECK(rgbs_coarse && sdist && u_fine && tdist_fine, TDGP_EINVAL, "importance_from_coarse: null pointer");
    TDGP_CHECK(S >= 4 && S <= MAXS && N >= 1 && N <= MAXS, TDGP_EUNSUPPORTED, "importance_from_coarse: bad S=%d N=%d (both at most %d)", S, N, MAXS);
    TDGP_CHECK(marcher == 0 || marcher == 1, TDGP_EINVAL, "importance_from_coarse: unknown ray marcher %d", marcher);
    TDGP_FAULT_CHECK("importance_from_coarse");
    if (rays == 0) return TDGP_OK;
    if (S <= 128 && N <= 128)
        TDGP_LAUNCH("importance_from_coarse_kernel", importance_from_coarse_kernel<128>, dim3(ray_blocks(rays)), dim3(256), 0, (hipStream_t)stream, rgbs_coarse, sdist,
                           u_fine, tdist_fine, sdist_fine, inds, fine_perm, rays, S, N, marcher, flags, density_bias, cut_threshold, t_near, t_far);
    else
        TDGP_LAUNCH("importance_from_coarse_kernel", importance_from_coarse_kernel<MAXS>, dim3(ray_blocks(rays)), dim3(256), 0, (hipStream_t)stream, rgbs_coarse, sdist,
                           u_fine, tdist_fine, sdist_fine, inds, fine_perm, rays, S, N, marcher, flags, density_bias, cut_threshold, t_near, t_far);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

TDGP_API int tdgp_merge_composite(const float* rgbs_coarse, const float* t_coarse, int S1, const float* rgbs_fine, const float* t_fine, int S2,
                                  float* rgb, float* depth, float* wsum, float* final_T, int32_t* perm, const int32_t* fine_perm, int64_t rays,
                                  int marcher, int flags, float density_bias, float cut_threshold, tdgp_stream_t stream) {
    TDGP_CHECK(rgbs_coarse && t_coarse && rgbs_fine && t_fine && rgb && depth, TDGP_EINVAL, "merge_composite: null pointer");
    TDGP_CHECK(S1 >= 1 && S2 >= 1 && S1 + S2 <= MAXS, TDGP_EUNSUPPORTED, "merge_composite: S1+S2=%d > %d", S1 + S2, MAXS);
    TDGP_CHECK(marcher == 0 || marcher == 1, TDGP_EINVAL, "merge_composite: unknown ray marcher %d", marcher);
    TDGP_FAULT_CHECK("merge_composite");
    if (rays == 0) return TDGP_OK;
    if (S1 + S2 <= 128)
        TDGP_LAUNCH("merge_composite_kernel", merge_composite_kernel<128>, dim3(ray_blocks(rays)), dim3(256), 0, (hipStream_t)stream, rgbs_coarse, t_coarse, S1,
                           rgbs_fine, t_fine, S2, rgb, depth, wsum, final_T, perm, fine_perm, rays, marcher, flags, density_bias, cut_threshold);
    else if (S1 + S2 <= 192)        // BASELINE configs[4]: 96 + 96 samples -- three lane slots and 6.2 KB of scratch per wave instead of four and 8.2 KB
        TDGP_LAUNCH("merge_composite_kernel", merge_composite_kernel<192>, dim3(ray_blocks(rays)), dim3(256), 0, (hipStream_t)stream, rgbs_coarse, t_coarse, S1,
                           rgbs_fine, t_fine, S2, rgb, depth, wsum, final_T, perm, fine_perm, rays, marcher, flags, density_bias, cut_threshold);
    else
        TDGP_LAUNCH("merge_composite_kernel", merge_composite_kernel<MAXS>, dim3(ray_blocks(rays)), dim3(256), 0, (hipStream_t)stream, rgbs_coarse, t_coarse, S1,
                           rgbs_fine, t_fine, S2, rgb, depth, wsum, final_T, perm, fine_perm, rays, marcher, flags, density_bias, cut_threshold);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}
