// conv_grad.hip -- weight gradient of a 2-D convolution on the gfx950 matrix cores.
//
// Reference: src/torch_utils/ops/conv2d_gradfix.py:141-150 (`Conv2dGradWeight.forward` = aten::convolution_backward with
// output_mask [0,1,0], non-transposed, groups 1, dilation 1):
//
//     dW[o,c,ky,kx] = sum_{b,oy,ox} dy[b,o,oy,ox] * x[b,c, oy*stride + ky - pad, ox*stride + kx - pad]       (zero outside x)
//
// Per tap (ky,kx) this is a GEMM  dW_tap[Cout x Cin] = DY[Cout x P] * Xshift_tap[P x Cin]  whose long dimension is the pixel
// index P = B*OH*OW (up to 2M), while the output is small (<= 2.4 M floats): the launch is split over the K dimension --
// block (z) owns a slice of (b,oy) rows, writes its partial [Cout,Cin,k,k] and a second kernel sums the slices in slice order
// (deterministic; fp32 atomics would make the result depend on the block schedule).
//
// Block = 256 threads = 2 x 2 waves, each wave one 32(o) x 32(c) MFMA tile (v_mfma_f32_32x32x2_f32, K = 2 pixels per
// instruction).  Per 32-pixel chunk of an output row the block stages dy[64 o][32 px] and the tap-shifted x[64 c][32 px] in LDS
// (row pitch 33 floats: the fragment reads of 32 lanes hit 32 banks), the next chunk's global loads are in flight during the
// 16 MFMAs of the current one.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WGradParams {
    const float* x;        // [B,Cin,H,W]
    const float* dy;       // [B,Cout,OH,OW]
    float* partial;        // [nslice][Cout][Cin][k*k]
    int B, Cin, Cout, H, W, OH, OW, k, stride, pad;
    int tiles_c;           // cdiv(Cin, 64)
    int rows, rows_per_slice;      // rows = B*OH
    uint32_t x_bytes, dy_bytes;
};

constexpr int WG_PITCH = 33;

__global__ __launch_bounds__(256) void conv_wgrad_mfma_kernel(WGradParams p) {
    __shared__ float As[64 * WG_PITCH];         // dy tile [o][px]
    __shared__ float Bs[64 * WG_PITCH];         // x tile  [c][px]
    const int tid = threadIdx.x, l = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), l32 = l & 31, half = l >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    const int o0 = (blockIdx.x / p.tiles_c) * 64, c0 = (blockIdx.x % p.tiles_c) * 64;
    const int tap = blockIdx.y, ky = tap / p.k, kx = tap - ky * p.k;
    const int row_begin = blockIdx.z * p.rows_per_slice, row_end = min(row_begin + p.rows_per_slice, p.rows);
    const int chunks_per_row = (p.OW + 31) >> 5;
    const int nchunks = max(row_end - row_begin, 0) * chunks_per_row;

    // this thread's 8 + 8 staged elements: tile row = (tid >> 5) + 8 * i, pixel = tid & 31.  Channel part of the byte offset fixed per
    // thread (rows beyond Cout / Cin -> beyond the descriptor -> 0), position part one vector value per chunk, row/sample part scalar:
    // the chunk -> (b, oy, ox) arithmetic runs on the scalar unit and each load is one instruction.
    const int px = tid & 31, r0 = tid >> 5;
    constexpr uint32_t OOB = 0xFFFFFFF0u;
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    uint32_t vo[8], vc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int o = o0 + r0 + 8 * i, c = c0 + r0 + 8 * i;
        vo[i] = o < p.Cout ? (uint32_t)(o * p.OH * p.OW) * 4u : OOB;
        vc[i] = c < p.Cin ? (uint32_t)(c * p.H * p.W) * 4u : OOB;
    }
    float ra[8], rb[8];
    auto load_chunk = [&](int ch) {
        const int rr = __builtin_amdgcn_readfirstlane(ch / chunks_per_row);
        const int row = row_begin + rr, ox = (ch - rr * chunks_per_row) * 32 + px;
        const int b = row / p.OH, oy = row - b * p.OH;                                   // uniform (scalar unit)
        const int iy = oy * p.stride + ky - p.pad, ix = ox * p.stride + kx - p.pad;
        const bool oka = ox < p.OW, okb = oka && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const uint32_t pa = oka ? (uint32_t)ox * 4u : OOB, pb = okb ? (uint32_t)ix * 4u : OOB;       // OOB + channel offset stays out of range (wraps past 4 GiB are excluded on the host)
        const uint32_t sa = (uint32_t)((b * p.Cout * p.OH + oy) * p.OW) * 4u, sb = (uint32_t)((b * p.Cin * p.H + (okb ? iy : 0)) * p.W) * 4u;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t oa = (pa == OOB || vo[i] == OOB) ? OOB : vo[i] + pa, ob = (pb == OOB || vc[i] == OOB) ? OOB : vc[i] + pb;
            ra[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rdy, oa, sa, 0));
            rb[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, ob, sb, 0));
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    if (nchunks > 0) load_chunk(0);
    const float* al = As + (wm * 32 + l32) * WG_PITCH + half;
    const float* bl = Bs + (wn * 32 + l32) * WG_PITCH + half;
    for (int ch = 0; ch < nchunks; ch++) {
        __syncthreads();                                   // the previous chunk's fragments have been read
#pragma unroll
        for (int i = 0; i < 8; i++) {
            As[(r0 + 8 * i) * WG_PITCH + px] = ra[i];
            Bs[(r0 + 8 * i) * WG_PITCH + px] = rb[i];
        }
        __syncthreads();
        if (ch + 1 < nchunks) load_chunk(ch + 1);
#pragma unroll
        for (int ks = 0; ks < 16; ks++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(al[2 * ks], bl[2 * ks], acc, 0, 0, 0);
    }

    // acc[r] = C[m = (r & 3) + 8 * (r >> 2) + 4 * half][n = l32]
    const int kk = p.k * p.k;
    float* dst = p.partial + (int64_t)blockIdx.z * p.Cout * p.Cin * kk;
    const int c = c0 + wn * 32 + l32;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int o = o0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (o < p.Cout && c < p.Cin) dst[((int64_t)o * p.Cin + c) * kk + tap] = acc[r];
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int64_t n, int nslice) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < nslice; s++) v += partial[(int64_t)s * n + i];
        dw[i] = v;
    }
}

int wgrad_slices(int B, int Cin, int Cout, int OH, int k) {
    const int blocks_xy = cdiv(Cout, 64) * cdiv(Cin, 64) * k * k;
    int ns = cdiv(2048, blocks_xy);                   // ~8 blocks per CU
    const int rows = B * OH;
    if (ns > rows) ns = rows;
    if (ns > 256) ns = 256;
    return ns < 1 ? 1 : ns;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// General strided convolution (correlation), groups 1, dilation 1:   y[b,o,oy,ox] = bias[o] + sum_{c,ky,kx} w[o,c,ky,kx] x[b,c,oy*st+ky-pad,ox*st+kx-pad]
// The forms the fused generator kernels do not cover: stride 2 (the input gradient of the x2 transposed convolution, i.e. the
// adjoint of conv_transpose2d(stride 2) -- conv2d_gradfix.py:126-129 -- and the discriminator's down-sampling convolutions,
// conv2d_resample.py:104-107) and paddings other than k // 2.  Block = 64 output channels x 64 consecutive pixels of one output
// row, 2 x 2 waves of 32 x 32 MFMA tiles; K loop over chunks of 8 input channels x all taps, weights as [tap][c][o] and the
// strided activation patch [c][ky][column] staged in LDS, the next chunk's loads in flight during the MFMAs of the current one.
// ---------------------------------------------------------------------------------------------------------------------------------
struct ConvFwdParams {
    const float* x; const float* w; const float* bias; float* y;
    int B, Cin, Cout, H, W, OH, OW, k, stride, pad;
    int tiles_x;       // cdiv(OW, 64)
    uint32_t x_bytes, w_bytes;
};

constexpr int CF_KC = 8, CF_APITCH = 65;

template <int K>
__global__ __launch_bounds__(256) void conv_strided_mfma_kernel(ConvFwdParams p) {
    constexpr int T = K * K;
    constexpr int IWP = 64 * 2 + K - 1 + 1;                      // patch columns for stride <= 2 (+1: odd pitch)
    __shared__ float As[T * CF_KC * CF_APITCH];                  // [tap][c][o]
    __shared__ float Bs[CF_KC * K * IWP];                        // [c][ky][col]
    const int tid = threadIdx.x, l = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), l32 = l & 31, half = l >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    const int o0 = blockIdx.y * 64;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int oy = t % p.OH, b = t / p.OH;
    const int ox0 = tx * 64;
    const int iy0 = oy * p.stride - p.pad, ix0 = ox0 * p.stride - p.pad;
    const int ncols = 63 * p.stride + K;                         // patch columns in use
    constexpr int NA = (T * CF_KC * 64 + 255) / 256, NB = (CF_KC * K * IWP + 255) / 256;
    float ra[NA], rb[NB];
    // Loop-invariant per-thread byte offsets (out-of-range elements -> an offset beyond the descriptor: the load returns 0); per chunk
    // only a SCALAR offset changes.  Computing the addresses inside the K loop cost ~300 vector-ALU instructions per chunk, each
    // queued behind the other waves' 64-cycle MFMAs: several times the 36 MFMAs of the chunk themselves.
    constexpr uint32_t OOB = 0xFFFFFFF0u;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    uint32_t va[NA], vb[NB];
    int ca[NA], cb[NB];                                          // channel of the element inside a chunk (for the ragged last chunk)
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int e = tid + i * 256;                             // e = (o * KC + c) * T + tap: taps of one (o,c) are contiguous in memory
        const int tap = e % T, oc = e / T, c = oc % CF_KC, o = oc / CF_KC;
        ca[i] = c;
        va[i] = (e < T * CF_KC * 64 && o0 + o < p.Cout) ? (uint32_t)(((o0 + o) * p.Cin + c) * T + tap) * 4u : OOB;
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const int e = tid + i * 256;
        const int col = e % IWP, r = e / IWP, ky = r % K, c = r / K;
        const int iy = iy0 + ky, ix = ix0 + col;
        cb[i] = c;
        vb[i] = (e < CF_KC * K * IWP && col < ncols && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? (uint32_t)(((b * p.Cin + c) * p.H + iy) * p.W + ix) * 4u : OOB;
    }
    const uint32_t hw4 = (uint32_t)(p.H * p.W) * 4u;
    auto load_chunk = [&](int c0) {
        const uint32_t sa = (uint32_t)(c0 * T) * 4u, sb = (uint32_t)c0 * hw4;
        const bool ragged = c0 + CF_KC > p.Cin;                  // uniform: only the last chunk of a Cin that is not a multiple of 8
#pragma unroll
        for (int i = 0; i < NA; i++) {
            ra[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, va[i], sa, 0));
            if (ragged && c0 + ca[i] >= p.Cin) ra[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            rb[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vb[i], sb, 0));
            if (ragged && c0 + cb[i] >= p.Cin) rb[i] = 0.f;
        }
    };
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    load_chunk(0);
    const int colbase = (wn * 32 + l32) * p.stride;
    for (int c0 = 0; c0 < p.Cin; c0 += CF_KC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int e = tid + i * 256;
            if (e < T * CF_KC * 64) { const int tap = e % T, oc = e / T, c = oc % CF_KC, o = oc / CF_KC; As[(tap * CF_KC + c) * CF_APITCH + o] = ra[i]; }
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int e = tid + i * 256;
            if (e < CF_KC * K * IWP) Bs[e] = rb[i];
        }
        __syncthreads();
        if (c0 + CF_KC < p.Cin) load_chunk(c0 + CF_KC);
#pragma unroll
        for (int ky = 0; ky < K; ky++)
#pragma unroll
            for (int kx = 0; kx < K; kx++)
#pragma unroll
                for (int ks = 0; ks < CF_KC / 2; ks++) {
                    const int c = 2 * ks + half;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[((ky * K + kx) * CF_KC + c) * CF_APITCH + wm * 32 + l32],
                                                               Bs[(c * K + ky) * IWP + colbase + kx], acc, 0, 0, 0);
                }
    }
    const int ox = ox0 + wn * 32 + l32;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int o = o0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (o < p.Cout && ox < p.OW) p.y[(((int64_t)b * p.Cout + o) * p.OH + oy) * p.OW + ox] = acc[r] + (p.bias ? p.bias[o] : 0.f);
    }
}

}  // namespace

TDGP_API int tdgp_conv2d(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int H, int W, int OH, int OW, int k,
                         int stride, int pad, tdgp_stream_t stream) {
    TDGP_CHECK(x && w && y, TDGP_EINVAL, "conv2d: null pointer");
    TDGP_CHECK(B >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1 && pad >= 0, TDGP_EINVAL, "conv2d: bad shape");
    TDGP_CHECK(stride == 1 || stride == 2, TDGP_EUNSUPPORTED, "conv2d: stride %d (1 or 2)", stride);
    TDGP_CHECK(k == 1 || k == 3, TDGP_EUNSUPPORTED, "conv2d: kernel size %d (1 or 3)", k);
    TDGP_CHECK(OH == (H + 2 * pad - k) / stride + 1 && OW == (W + 2 * pad - k) / stride + 1 && OH >= 1 && OW >= 1, TDGP_EINVAL,
               "conv2d: output %dx%d does not match input %dx%d, k=%d, stride=%d, pad=%d", OH, OW, H, W, k, stride, pad);
    ConvFwdParams p;
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.k = k; p.stride = stride; p.pad = pad;
    p.tiles_x = cdiv(OW, 64);
    TDGP_CHECK((int64_t)B * Cin * H * W < ((int64_t)1 << 30) && (int64_t)Cout * Cin * k * k < ((int64_t)1 << 30), TDGP_EINVAL,
               "conv2d: tensor too large (addressed through 4 GiB buffer descriptors)");
    p.x_bytes = (uint32_t)((int64_t)B * Cin * H * W * 4); p.w_bytes = (uint32_t)((int64_t)Cout * Cin * k * k * 4);
    const int64_t gx = (int64_t)p.tiles_x * OH * B;
    TDGP_CHECK(gx <= 2147483647LL, TDGP_EINVAL, "conv2d: too many tiles");
    const dim3 grid((unsigned)gx, cdiv(Cout, 64)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (k == 1) TDGP_LAUNCH("conv_strided_mfma_kernel", conv_strided_mfma_kernel<1>, grid, block, 0, s, p);
    else TDGP_LAUNCH("conv_strided_mfma_kernel", conv_strided_mfma_kernel<3>, grid, block, 0, s, p);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}

TDGP_API int64_t tdgp_conv2d_weight_grad_workspace_bytes(int B, int Cin, int Cout, int OH, int k) {
    return (int64_t)wgrad_slices(B, Cin, Cout, OH, k) * Cout * Cin * k * k * (int64_t)sizeof(float);
}

TDGP_API int tdgp_conv2d_weight_grad(const float* x, const float* dy, float* dw, void* workspace, int64_t workspace_bytes, int B, int Cin,
                                     int Cout, int H, int W, int OH, int OW, int k, int stride, int pad, tdgp_stream_t stream) {
    TDGP_CHECK(x && dy && dw, TDGP_EINVAL, "conv2d_weight_grad: null pointer");
    TDGP_CHECK(B >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1 && k >= 1 && stride >= 1 && pad >= 0, TDGP_EINVAL, "conv2d_weight_grad: bad shape");
    TDGP_CHECK(OH == (H + 2 * pad - k) / stride + 1 && OW == (W + 2 * pad - k) / stride + 1, TDGP_EINVAL,
               "conv2d_weight_grad: output %dx%d does not match input %dx%d, k=%d, stride=%d, pad=%d", OH, OW, H, W, k, stride, pad);
    TDGP_CHECK(k <= 7, TDGP_EUNSUPPORTED, "conv2d_weight_grad: kernel size %d > 7", k);
    const int64_t need = tdgp_conv2d_weight_grad_workspace_bytes(B, Cin, Cout, OH, k);
    TDGP_CHECK(workspace && workspace_bytes >= need, TDGP_EINVAL, "conv2d_weight_grad: workspace of %lld bytes needed", (long long)need);
    WGradParams p;
    p.x = x; p.dy = dy; p.partial = (float*)workspace;
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.k = k; p.stride = stride; p.pad = pad;
    p.tiles_c = cdiv(Cin, 64);
    TDGP_CHECK((int64_t)B * Cin * H * W < ((int64_t)1 << 30) && (int64_t)B * Cout * OH * OW < ((int64_t)1 << 30), TDGP_EINVAL,
               "conv2d_weight_grad: tensor too large (addressed through 4 GiB buffer descriptors)");
    p.x_bytes = (uint32_t)((int64_t)B * Cin * H * W * 4); p.dy_bytes = (uint32_t)((int64_t)B * Cout * OH * OW * 4);
    p.rows = B * OH;
    const int ns = wgrad_slices(B, Cin, Cout, OH, k);
    p.rows_per_slice = cdiv(p.rows, ns);
    hipStream_t s = (hipStream_t)stream;
    TDGP_LAUNCH("conv_wgrad_mfma_kernel", conv_wgrad_mfma_kernel, dim3(cdiv(Cout, 64) * p.tiles_c, k * k, ns), dim3(256), 0, s, p);
    const int64_t n = (int64_t)Cout * Cin * k * k;
    TDGP_LAUNCH("wgrad_reduce_kernel", wgrad_reduce_kernel, dim3((int)min((int64_t)1024, cdiv64(n, 256))), dim3(256), 0, s, (const float*)workspace, dw, n, ns);
    TDGP_LAUNCH_CHECK();
    return TDGP_OK;
}
