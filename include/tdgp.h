/*
 * tdgp.h -- C ABI of libtdgp_hip.so: the MI355X (gfx950) generator-forward hot path of 3DGP.
 *
 * Conventions:
 *   - plain pointers and sizes only; every data pointer is a DEVICE pointer unless marked "host";
 *   - the library never allocates, frees or synchronises: the caller owns all memory (outputs and
 *     workspaces are caller-allocated) and passes the HIP stream to launch on;
 *   - every function returns 0 on success or a negative TDGP_E* code; tdgp_last_error() returns a
 *     thread-local message.  Nothing throws across the ABI;
 *   - fp32 unless a `dtype` argument says otherwise (TDGP_F32 / TDGP_F16 / TDGP_BF16; TDGP_F64 for the two plugin ops);
 *   - re-entrant.  Process-wide state is limited to: the init-once kernel tables, the two switches tdgp_set_conv_arith (algorithm of
 *     the large 3x3 layers; default 0 = fp32 MFMA) and tdgp_profile_enable (per-kernel event timing; default off), and the launch
 *     geometry cached per kernel instantiation and device (resident blocks per CU, CU count, raised LDS caps), and the device-fault
 *     word (tdgp_device_fault).  Nothing else is remembered between calls.
 *
 * Each entry point cites the reference interface it replaces (file:line under the reference tree).
 * The reference binds its two native ops through pybind (bias_act.cpp:94, upfirdn2d.cpp:102); the
 * modulated convolution and the renderer have no native boundary in the reference (plain PyTorch
 * functions) -- their entry points here take the same operands as those Python functions.
 */
#ifndef TDGP_H
#define TDGP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDGP_OK            0
#define TDGP_EINVAL       -1   /* bad argument (shape / range / null)         */
#define TDGP_EUNSUPPORTED -2   /* valid request this build has no kernel for  */
#define TDGP_ELAUNCH      -3   /* HIP launch error                            */
#define TDGP_EWORKSPACE   -4   /* workspace too small                         */

#define TDGP_F32  0
#define TDGP_F16  1
#define TDGP_BF16 2
#define TDGP_F64  3   /* tdgp_bias_act / tdgp_bias_act_grad / tdgp_upfirdn2d only (bias_act.cpp:77, upfirdn2d.cpp:63: ..._FLOATING_TYPES_AND_HALF); compute type double */

typedef void* tdgp_stream_t;   /* hipStream_t */

int         tdgp_version(void);
const char* tdgp_last_error(void);

/* Per-kernel timing (the reference wraps its ops in torch.autograd.profiler ranges: src/torch_utils/misc.py:101-106).
 * tdgp_profile_enable(1) makes every launch record a HIP event pair on its stream; tdgp_profile_report() waits for
 * the recorded events (the ONLY entry point that blocks) and writes "<kernel> <launches> <total_ms> <min_ms> <max_ms>"
 * lines into a HOST buffer, returning the bytes needed.  tdgp_profile_enable(0/1) also clears previous records. */
int     tdgp_profile_enable(int on);
int64_t tdgp_profile_report(char* buf, int64_t cap);

/* Device-fault word.  The producer / consumer field kernel bounds its in-kernel waits (a protocol bug or a multi-millisecond stall under
 * a debugger / thread-trace profiler must never hang the GPU); a wait that runs out ORs a code into one int of pinned HOST memory (the
 * single allocation this library makes: 64 bytes, hipHostMalloc, on first use) instead of completing silently with wrong numbers.
 * tdgp_device_fault(0) returns the word, tdgp_device_fault(1) returns and clears it -- meaningful after the stream has been synchronised
 * by the caller.  While it is non-zero tdgp_triplane_field / tdgp_importance_from_coarse / tdgp_merge_composite / tdgp_ray_march return
 * TDGP_ELAUNCH (checked on the host at entry, no synchronisation).  No reference counterpart (its ops cannot time out).
 * codes: 1 = field ring, producer side; 2 = field ring, consumer side. */
int     tdgp_device_fault(int clear);

/* ---------------------------------------------------------------------------------------------
 * bias_act forward:  y = clamp(gain * act(x + b[(i / stepB) % sizeB]))
 * replaces: src/torch_utils/ops/bias_act.cpp:32 `bias_act(x,b,xref,yref,dy,grad=0,dim,act,alpha,gain,clamp)`
 * act = the reference's cuda_idx 1..9 (bias_act.py:21-31); clamp < 0 disables; b may be NULL.
 * n <= INT_MAX (bias_act.cpp:40).  x and y dense with identical layout.
 * --------------------------------------------------------------------------------------------- */
int tdgp_bias_act(const void* x, const void* b, void* y, int64_t n, int sizeB, int64_t stepB,
                  int act, float alpha, float gain, float clamp, int dtype, tdgp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * bias_act first / second derivative (the training-side forms of the same plugin call).
 * replaces: src/torch_utils/ops/bias_act.cpp:32 `bias_act(x,b,xref,yref,dy,grad,dim,act,alpha,gain,clamp)` with grad = 1, 2
 *   grad 1:  y = x * dy * gain * act'(xref + b)       x = incoming gradient        (bias_act.py:172-178)
 *   grad 2:  y = x * dy * gain * act''(xref + b)      x = second-order gradient    (bias_act.py:190-197)
 * derivatives are written through yref / gain (xref for swish) exactly as bias_act.cu:60-133; with clamp >= 0 the result is 0
 * where the forward output yref lay outside (-clamp, clamp).  xref / yref / dy / b may be NULL (read as 0 / 0 / 1 / 0).
 * --------------------------------------------------------------------------------------------- */
int tdgp_bias_act_grad(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y, int64_t n,
                       int sizeB, int64_t stepB, int grad, int act, float alpha, float gain, float clamp, int dtype,
                       tdgp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * upfirdn2d forward on [N,C,H,W] with arbitrary element strides.
 * replaces: src/torch_utils/ops/upfirdn2d.cpp:16
 *   `upfirdn2d(x,f,upx,upy,downx,downy,padx0,padx1,pady0,pady1,flip,gain)`
 * f: fp32 [fH,fW] contiguous.  outH/outW must equal (in*up + pad0 + pad1 - f + down) / down.
 * x_strides / y_strides: host arrays of 4 element strides (N,C,H,W).
 * --------------------------------------------------------------------------------------------- */
int tdgp_upfirdn2d(const void* x, const float* f, void* y, int N, int C, int inH, int inW,
                   const int64_t* x_strides, int outH, int outW, const int64_t* y_strides,
                   int fH, int fW, int upx, int upy, int downx, int downy,
                   int padx0, int padx1, int pady0, int pady1, int flip, float gain,
                   int dtype, tdgp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Modulated convolution (new native boundary; the reference runs cuDNN grouped convs from Python).
 * replaces: src/training/networks_stylegan2.py:31 `modulated_conv2d(x, weight, styles, noise, up,
 *           padding=k//2, resample_filter, demodulate, flip_weight=(up==1), fused_modconv=True)`
 *           + conv2d_resample.py:46 + the bias_act that follows it in SynthesisLayer/ToRGBLayer
 *           (networks_stylegan2.py:139-144,170-171), + for ToRGB the skip-connection
 *           `img = upsample2d(img) + y` (networks_stylegan2.py:265-269).
 *
 * Weights are static across calls, so they are packed once:
 *   tdgp_modconv_pack_bytes  -> bytes of the packed buffer for (Cout,Cin,k)
 *   tdgp_modconv_pack        -> wpack ([Cin/4][k*k][Cout][4 channels], zero padded, + sum_taps w^2 per (c,o))
 * Forward:
 *   y[b,o] = clamp( act( d[b,o] * conv(x[b]*s[b,:], W)[o] + noise + bias[o] ) * gain )  (+ skip term, added after the clamp)
 *   d[b,o] = rsqrt(sum_c s[b,c]^2 * wsq[o,c] + 1e-8) if demodulate else 1
 *   up=2: transposed conv (stride 2) followed by the 4x4 FIR `fir4x4` with gain 4 (pad 1,1,1,1).
 *   fir4x4: HOST pointer to the 16 filter taps (a static 64-byte buffer: upfirdn2d.setup_filter([1,3,3,1])).
 * noise: NULL, or [H,W] (noise_bstride = 0), or [B,1,H,W] (noise_bstride = H*W); already multiplied
 *        by noise_strength by the caller.
 * skip: NULL or the previous-resolution image (same layout as the output: [B,Cout,H/2,W/2] for out_layout 0,
 *       [B,Cout/feat,H/2,W/2,feat] for out_layout 1) to be FIR-upsampled x2 with `fir4x4` (gain 4) and added
 *       (only with up=1, k=1).
 * out_layout: 0 = NCHW, 1 = plane-major channel-last [B, Cout/feat, H, W, feat] (the renderer's layout;
 *       `feat` = out_feat).
 * dcoef: NULL (the call computes d itself when demodulate != 0) or the [B,Cout] coefficients precomputed by
 *       tdgp_demod_batch for this layer.
 * workspace: tdgp_modconv2d_workspace_bytes(...) bytes (0 allowed when it returns 0).
 * k in {1, 3, 5} with padding k/2; up=2 needs k=3.  styles = NULL: plain convolution -- this is how the 5x5
 *       `Conv2dLayer`s of the depth adaptor run (src/training/layers.py:221-236: conv2d_resample + bias_act).
 * --------------------------------------------------------------------------------------------- */
int64_t tdgp_modconv_pack_bytes(int Cout, int Cin, int k);
int     tdgp_modconv_pack(const float* weight, void* wpack, int Cout, int Cin, int k, tdgp_stream_t stream);
int64_t tdgp_modconv2d_workspace_bytes(int B, int Cin, int Cout, int H, int W, int k, int up);
/* 1 when tdgp_modconv2d would take a folded x2 layer (out_layout 2, Cout4 = 4 x Cout parity channels) of this shape on the F(4x4) kernels
 * under the current arithmetic mode, else 0 -- a host-side query (no launch) the binding makes before folding and packing the weights. */
int     tdgp_modconv2d_takes_folded_up2(int B, int Cin, int Cout4, int H, int W);
int     tdgp_modconv2d(const float* x, const void* wpack, const float* styles, const float* dcoef, const float* noise,
                       int64_t noise_bstride, const float* bias, const float* fir4x4, const float* skip,
                       float* y, int B, int Cin, int Cout, int H, int W, int k, int up, int demodulate,
                       int act, float alpha, float gain, float clamp, int out_layout, int out_feat,
                       void* workspace, int64_t workspace_bytes, tdgp_stream_t stream);

/* The reduced-precision blocks of the backbone -- BASELINE configs[4]; replaces the reference's `use_fp16` path
 * (networks_stylegan2.py:50-53, :237 `dtype = float16 if use_fp16`, :286-304 / networks_epigraf.py:99-108 `num_fp16_res`, conv_clamp 256)
 * with bfloat16 in the place of float16.  Same arguments as tdgp_modconv2d except:
 *   x: bf16 NCHW [B,Cin,H,W];
 *   y: bf16 NCHW [B,Cout,H*up,W*up] for the 3x3 layers; for the ToRGB form (k = 1, up = 1, out_layout = 1, optional skip) the fp32
 *      channel-last planes, as in tdgp_modconv2d (the skip image stays fp32, :268).
 * bf16 weights x bf16 (style-scaled) activations on v_mfma_f32_32x32x16_bf16, fp32 accumulation; demodulation, noise, bias,
 * activation, gain, clamp in fp32; the reference's rounding points behind the accumulator (conv output, + noise, bias_act output; the
 * bias itself is rounded as in `self.bias.to(x.dtype)`), and the skip is added AFTER the clamp.
 * Forms taken: 3x3 with Cin % 32 == 0 (and W % 32 == 0 for up = 1); the channel-last ToRGB with Cout <= 96.  Anything else returns
 * TDGP_EUNSUPPORTED -- decided before anything is launched: widen x to fp32 and call tdgp_modconv2d.  `noise` must be 16-byte aligned with
 * a batch stride that is a multiple of 4 floats (TDGP_EINVAL otherwise; tdgp_modconv2d takes any float* and falls back to scalar
 * loads).  Workspace: tdgp_modconv2d_workspace_bytes of the same shape.
 * tdgp_cast_f32_bf16: x.to(bf16) (round to nearest even) at the first reduced-precision block (:250). */
int     tdgp_modconv2d_bf16(const void* x_bf16, const void* wpack, const float* styles, const float* dcoef, const float* noise,
                            int64_t noise_bstride, const float* bias, const float* fir4x4, const float* skip,
                            void* y, int B, int Cin, int Cout, int H, int W, int k, int up, int demodulate,
                            int act, float alpha, float gain, float clamp, int out_layout, int out_feat,
                            void* workspace, int64_t workspace_bytes, tdgp_stream_t stream);
int     tdgp_cast_f32_bf16(const float* x, void* y_bf16, int64_t n, tdgp_stream_t stream);

/* Plain x2 transposed convolution (the adjoint of a 3x3 stride-2 convolution, conv2d_gradfix.py:126-129):
 *   y [B,Cout,2H+1,2W+1] = conv_transpose2d(x * styles[:, :, None, None], w.transpose(0,1), stride=2),   w [Cout,Cin,3,3] packed by
 * tdgp_modconv_pack (the weights of the up-sampling synthesis layers; for the input gradient of y' = conv2d(a, W, stride 2) pass
 * x = dy' and w = W.transpose(0,1)).  styles may be NULL.  workspace: tdgp_modconv2d_workspace_bytes(B, Cin, Cout, H, W, 3, 2). */
int     tdgp_conv_transpose2d_x2(const float* x, const void* wpack, const float* styles, float* y, int B, int Cin, int Cout,
                                 int H, int W, void* workspace, int64_t workspace_bytes, tdgp_stream_t stream);

/* Arithmetic of the large 3x3 convolutions of tdgp_modconv2d.  All modes are fp32 in, fp32 accumulate, fp32 out:
 *   0 (default) = fp32 MFMA; stride-1 layers with W % 32 == 0, H % 8 == 0, Cin % 8 == 0, Cin >= 64 whose launch has at least one
 *       64-channel x 64-tile block per CU run as Winograd F(2x2,3x3) -- the convolution's exact algebra with 16 instead of 36
 *       multiplies per 2x2 outputs, transforms and products in fp32 (what cuDNN selects for the reference's fp32 3x3 convolutions,
 *       training_loop.py:76-77 keeps TF32 off); measured as close to the exactly rounded result as the direct sum (<= 5e-6 of the
 *       tensor scale).  Every other layer: direct sums;
 *   2 = fp32 MFMA, direct sums in every layer (the summation structure of a plain convolution; results differ from mode 0 in the last
 *       bits only);
 *   4 = as 0, but the F(4x4) layers with few input channels (Cin <= 128, the 256^2 / 512^2 blocks) keep their input transform as a pass
 *       of its own instead of inside the GEMM kernel (round 6, modconv_wino4f.inc): the SAME bits as mode 0 (same transform arithmetic,
 *       same summation order) -- an A/B and test switch;
 *   1 = (stride 1 with W % 32 == 0 and the x2 transposed form, launches of >= 256 tiles with styles present and Cin % 16 == 0) every
 *       fp32 operand split into three bf16 pieces, six piece products per multiply on the bf16 MFMA with fp32 accumulation
 *       (fp32-grade results, <= 4e-6 of the exact layer output; layers outside those conditions keep the fp32 kernels).
 * Process-wide; returns the previous mode (negative on error). */
int     tdgp_set_conv_arith(int mode);

/* Demodulation coefficients d[b,o] = rsqrt(sum_c s[b,c]^2 * sum_tap W[o,c,tap]^2 + 1e-8) (networks_stylegan2.py:62) of SEVERAL
 * layers in one launch.  meta: int64 [num_layers, 6] on the device = (address of the layer's sum_tap W^2 table, i.e. its wpack +
 * tdgp_modconv_wsq_offset(...) bytes; float offset of its [B,Cin] styles block in styles_all; Cin; Cout; Cout rounded up to 4;
 * float offset of its [B,Cout] block in dcoef_all).  max_cout = the largest Cout among them. */
int64_t tdgp_modconv_wsq_offset(int Cout, int Cin, int k);
int     tdgp_demod_batch(const float* styles_all, const int64_t* meta, float* dcoef_all, int B, int num_layers,
                         int max_cout, tdgp_stream_t stream);

/* Batched style affines for all layers of the backbone in one launch:
 *   out[obase + b*olen + oidx] = ((ws[b, widx, :] . A[row, :]) * (1/sqrt(w_dim)) + abias[row]) * row_scale[row]
 * replaces: the per-layer `self.affine(w)` FullyConnectedLayer calls (networks_stylegan2.py:130,169;
 * layers.py:42-58) and ToRGB's `* weight_gain` (:169).
 * A: concatenated affine weights [rows_total, w_dim]; abias [rows_total]; row_scale [rows_total] (1, or 1/sqrt(Cin)
 * for ToRGB); row_meta: int32 [rows_total,4] = (widx, obase, olen, oidx) so that layer l owns the contiguous block
 * styles[obase_l : obase_l + B*Cin_l] viewed as [B, Cin_l]. */
int tdgp_style_affine(const float* ws, const float* A, const float* abias, const int32_t* row_meta,
                      const float* row_scale, float* styles, int B, int num_ws, int w_dim, int rows_total,
                      tdgp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Convolution weight gradient.
 * replaces: src/torch_utils/ops/conv2d_gradfix.py:141-150 `Conv2dGradWeight.forward` (aten::convolution_backward, output_mask
 *           [0,1,0]; non-transposed, groups 1, dilation 1):
 *   dw[o,c,ky,kx] = sum_{b,oy,ox} dy[b,o,oy,ox] * x[b,c, oy*stride + ky - pad, ox*stride + kx - pad]      (zero outside x)
 * x [B,Cin,H,W], dy [B,Cout,OH,OW] with OH = (H + 2 pad - k) / stride + 1 (likewise OW), dw [Cout,Cin,k,k], k <= 7.
 * The pixel sum is split over blocks; the slices are added in slice order (run-to-run deterministic).
 * workspace: tdgp_conv2d_weight_grad_workspace_bytes(B, Cin, Cout, OH, k) bytes.
 * (The input gradient of a stride-1 'same' convolution is tdgp_modconv2d on dy with the flipped, transposed weights.)
 * --------------------------------------------------------------------------------------------- */
int64_t tdgp_conv2d_weight_grad_workspace_bytes(int B, int Cin, int Cout, int OH, int k);
/* Plain strided convolution (correlation; groups 1, dilation 1; k in {1,3}, stride in {1,2}, any padding):
 *   y[b,o,oy,ox] = bias[o] + sum_{c,ky,kx} w[o,c,ky,kx] * x[b,c, oy*stride + ky - pad, ox*stride + kx - pad]
 * replaces: torch.nn.functional.conv2d as reached from conv2d_gradfix.py:34-39 for the forms the fused kernels do not cover -- the
 *           adjoint of the x2 transposed convolution (conv2d_gradfix.py:126-129) and the down-sampling convolutions of
 *           conv2d_resample.py:104-107.  w [Cout,Cin,k,k] as stored by the modules; bias may be NULL. */
int     tdgp_conv2d(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int H, int W,
                    int OH, int OW, int k, int stride, int pad, tdgp_stream_t stream);
int     tdgp_conv2d_weight_grad(const float* x, const float* dy, float* dw, void* workspace, int64_t workspace_bytes, int B,
                                int Cin, int Cout, int H, int W, int OH, int OW, int k, int stride, int pad,
                                tdgp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Camera + rays.
 * replaces: src/training/rendering_utils.py:194 `compute_cam2world_matrix(camera_params)`,
 *           src/training/tri_plane_renderer.py:487 `sample_rays(c2w, fov, resolution, patch_params)`
 * fov in degrees; fov_stride 1 = per-sample tensor, 0 = one shared scalar. patch_* may be NULL.
 * Ray r of sample b is pixel (row r / w, col r % w).
 * --------------------------------------------------------------------------------------------- */
int tdgp_cam2world(const float* angles, const float* radius, const float* look_at, float* c2w, int B,
                   tdgp_stream_t stream);
int tdgp_sample_rays(const float* c2w, const float* fov, int fov_stride, const float* patch_scales,
                     const float* patch_offsets, float* ray_o, float* ray_d, int B, int h, int w,
                     tdgp_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Renderer pieces.
 * replaces (all src/training/tri_plane_renderer.py unless noted):
 *   :208  ImportanceRenderer.sample_stratified            -> tdgp_sample_stratified
 *   :560  simple_tri_plane_renderer + networks_epigraf.py:46 TriPlaneMLP.forward -> tdgp_triplane_field
 *   :353 / :300  ClassicalRayMarcher / MipRayMarcher2      -> tdgp_ray_march
 *   :237,:257  sample_importance / sample_pdf              -> tdgp_sample_importance
 *   :196  unify_samples                                    -> tdgp_unify_samples
 *   :126  ImportanceRenderer.forward (whole chain)         -> the fused pair
 *                 tdgp_importance_from_coarse + tdgp_merge_composite
 * marcher: 0 = classical, 1 = mip.
 * --------------------------------------------------------------------------------------------- */

/* u [rays,S] uniforms -> sdist [rays,S] (s-space) and, if tdist != NULL, t = s*t_far + (1-s)*t_near. */
int tdgp_sample_stratified(const float* u, float* sdist, float* tdist, int64_t rays, int S, int marcher,
                           float t_near, float t_far, tdgp_stream_t stream);

/* The marchers' density activation as a function of its own: out[i] = max(x, 0) when flags bit 3 (relu clamp) is set, else the softplus
 * of x = sigma[i] + density_bias -- the routine the march kernels apply, so that a threshold taken over `out` (the cut_quantile option,
 * tri_plane_renderer.py:324-326, :366-368: `x < quantile(x)` on the activated densities) compares like with like. */
int tdgp_density_activation(const float* sigma, float* out, int64_t n, int flags, float density_bias, tdgp_stream_t stream);

/* NCHW planes [B,3F,H,W] -> plane-major channel-last [B,3,H,W,F] (the field kernel's layout). */
int tdgp_planes_to_hwc(const float* planes_nchw, float* planes_hwc, int B, int F, int H, int W,
                       tdgp_stream_t stream);

/* The lookup alone (round 6): feats [B,P,F] = mean over the three planes of the bilinear samples (align_corners, zero padding) at coords [B,P,3] / scale
 * -- what simple_tri_plane_renderer hands to TriPlaneMLP (tri_plane_renderer.py:575-586, networks_epigraf.py:55: `x.mean(dim=1)`), in torch's own
 * evaluation order.  For decoders the fused kernel below does not cover (tri_plane.mlp.n_layers != 2, has_view_cond: networks_epigraf.py:35-43), whose
 * layers the binding then runs as eager tensor ops like the reference.  planes_hwc [B,3,H,W,F], F % 4 == 0. */
int tdgp_triplane_features(const float* planes_hwc, const float* coords, float* feats, int B, int64_t P, int F, int H, int W,
                           float scale, tdgp_stream_t stream);

/* Tri-plane bilinear lookup (align_corners, zero padding) + mean + tiny MLP, fused.
 * planes_hwc: [B,3,H,W,F].  Points are given either as coords [B,P,3] (ray_o = NULL), or as rays:
 * ray_o/ray_d [B,R,3] and t [B,R,S] with P = R*S, point p = ray p/S at depth t[p].  In ray mode `ray_w` > 0 declares that
 * the R rays of a sample form a [R/ray_w, ray_w] image (ray r = pixel (r / ray_w, r % ray_w)): the kernel then walks 4x4-pixel
 * tiles for cache locality (results are identical); ray_w = 0 keeps the linear point order.
 * w0 [hid,F], b0 [hid], w1 [4,hid], b1 [4] are the RAW module parameters (gains 1/sqrt(F), 1/sqrt(hid),
 * lrelu 0.2 * sqrt(2) applied inside, layers.py:39-58).
 * out: rgbs [B,P,4] = (r,g,b,sigma); marcher=1 applies sigmoid*1.002-0.001 to rgb (networks_epigraf.py:61-62).
 * tap_idx: optional int32 [B,P,3,2] = (floor ix, floor iy) per plane, for integer-row parity tests.
 * sigma_noise / density_noise: training-time density noise (tri_plane_renderer.py:185-186): sigma += sigma_noise[b,p] * density_noise
 *          with sigma_noise [B,P] standard-normal draws; density_noise = 0 (sigma_noise may be NULL) turns it off.
 * Requires F % 4 == 0, F <= 64, hid % 16 == 0, hid <= 128. */
int tdgp_triplane_field(const float* planes_hwc, const float* coords, const float* ray_o, const float* ray_d,
                        const float* t, const float* w0, const float* b0, const float* w1, const float* b1,
                        const float* sigma_noise, float density_noise, float* rgbs, int32_t* tap_idx, int B, int64_t P,
                        int S, int ray_w, int F, int H, int W, int hid, float scale, int marcher, tdgp_stream_t stream);

/* Gradient of tdgp_triplane_field in coords mode: autograd through simple_tri_plane_renderer + TriPlaneMLP
 * (tri_plane_renderer.py:560-588 -- grid_sample backward --, networks_epigraf.py:46-68) w.r.t. the planes and the four MLP tensors.
 * d_out [B,P,4] = gradient w.r.t. (r,g,b,sigma) as tdgp_triplane_field returns them.  d_planes_hwc [B,3,H,W,F] is ACCUMULATED
 * into with fp32 atomics (zero it first; NULL skips it; like torch's grid_sampler backward its last bits vary run to run);
 * d_w0 [hid,F], d_b0 [hid], d_w1 [4,hid], d_b1 [4] are written, deterministically.  d_coords [B,P,3] (NULL skips it) receives the
 * gradient w.r.t. the sample positions -- grid_sampler's grid gradient (taps outside a plane read as zero, x (size - 1) / 2 for
 * align_corners) through the plane mean and `coords / scale`: what the camera parameters are trained through (loss.py:69-83 applies the
 * camera adaptor inside run_G; rendering_utils.py:194-218 and tri_plane_renderer.py:487-527 are differentiable in the reference).
 * Requires F in {8,16,24,32}, hid <= 64.
 * workspace: tdgp_triplane_field_grad_workspace_bytes(B, P, F, hid) bytes. */
int64_t tdgp_triplane_field_grad_workspace_bytes(int B, int64_t P, int F, int hid);
int     tdgp_triplane_field_grad(const float* planes_hwc, const float* coords, const float* w0, const float* b0,
                                 const float* w1, const float* b1, const float* d_out, float* d_planes_hwc, float* d_w0,
                                 float* d_b0, float* d_w1, float* d_b1, float* d_coords, void* workspace, int64_t workspace_bytes,
                                 int B, int64_t P, int F, int H, int W, int hid, float scale, int marcher, tdgp_stream_t stream);

/* Generic marcher on [rays,S,C] colours, [rays,S] densities/depths (any S <= 256, C <= 8).
 * weights: [rays,S] (classical, or mip with inf depth) / [rays,S-1] (mip without); may be NULL.
 * flags: bit0 use_inf_depth, bit1 last_back (classical), bit2 white_back (mip), bit3 clamp_mode relu.
 * cut_threshold: the `cut_quantile` option (tri_plane_renderer.py:324-326, 366-368): activated densities below it are set to 0
 * before alpha; the caller computes it (the reference takes torch.quantile over the whole activated-density tensor); 0 = off. */
int tdgp_ray_march(const float* colors, const float* densities, const float* depths, float* rgb,
                   float* depth, float* weights, float* final_T, int64_t rays, int S, int C, int marcher,
                   int flags, float density_bias, float cut_threshold, tdgp_stream_t stream);

/* Gradient of tdgp_ray_march: autograd through ClassicalRayMarcher / MipRayMarcher2 (:353-398, :299-349).
 * d_rgb [rays,C], d_depth [rays] (may be NULL), d_weights [rays,M] (may be NULL; M as in tdgp_ray_march) ->
 * d_colors [rays,S,C], d_densities [rays,S] (gradient w.r.t. the RAW densities, through softplus / relu).  Depths carry no
 * gradient.  C in {1,3,4}, S <= 256; marcher / flags / density_bias as in tdgp_ray_march. */
int tdgp_ray_march_grad(const float* colors, const float* densities, const float* depths, const float* d_rgb,
                        const float* d_depth, const float* d_weights, float* d_colors, float* d_densities,
                        int64_t rays, int S, int C, int marcher, int flags, float density_bias, tdgp_stream_t stream);

/* sample_importance: z [rays,S] (s-space), weights [rays,Wn], u [rays,N] -> samples [rays,N].
 * Optional outputs: inds/below/above int32 [rays,N] (searchsorted right=True, clamped), cdf [rays,Wn-1]. */
int tdgp_sample_importance(const float* z, const float* weights, const float* u, float* samples,
                           int32_t* inds, int32_t* below, int32_t* above, float* cdf,
                           int64_t rays, int S, int Wn, int N, int marcher, tdgp_stream_t stream);

/* unify_samples: concat + stable sort by depth + gather; perm (int32 [rays,S1+S2]) optional. */
int tdgp_unify_samples(const float* d1, const float* c1, const float* s1, int S1,
                       const float* d2, const float* c2, const float* s2, int S2,
                       float* d, float* c, float* s, int32_t* perm, int64_t rays, int C,
                       tdgp_stream_t stream);

/* Fused chain, step 1: coarse march (s-space depths) -> importance sampling -> fine depths in t-space.
 * rgbs_coarse [rays,S,4], sdist [rays,S], u_fine [rays,N] -> tdist_fine [rays,N] WRITTEN IN ASCENDING DEPTH ORDER (stable):
 * the reference keeps draw order and sorts coarse+fine together afterwards, so the final composite is unchanged, but the
 * second field pass becomes spatially coherent and the merge below is a merge of two sorted lists.
 * Optional: sdist_fine [rays,N] and inds int32 [rays,N] in DRAW order; fine_perm int32 [rays,N]: draw index of sorted slot.
 * cut_threshold as in tdgp_ray_march (here and in tdgp_merge_composite). */
int tdgp_importance_from_coarse(const float* rgbs_coarse, const float* sdist, const float* u_fine,
                                float* tdist_fine, float* sdist_fine, int32_t* inds, int32_t* fine_perm,
                                int64_t rays, int S, int N, int marcher, int flags, float density_bias,
                                float cut_threshold, float t_near, float t_far, tdgp_stream_t stream);

/* Fused chain, step 2: merge coarse+fine by depth (stable, coarse before fine on ties), march in t-space.
 * rgbs_* [rays,S*,4], t_* [rays,S*] (any order; ascending lists take a fast path) -> rgb [rays,3], depth [rays],
 * wsum [rays], final_T [rays]; perm optional int32 [rays,S1+S2] = index into the concatenation [coarse ; fine in draw
 * order] (fine_perm, optional, maps the fine list's slots back to draw order; NULL = identity). */
int tdgp_merge_composite(const float* rgbs_coarse, const float* t_coarse, int S1,
                         const float* rgbs_fine, const float* t_fine, int S2,
                         float* rgb, float* depth, float* wsum, float* final_T, int32_t* perm,
                         const int32_t* fine_perm, int64_t rays, int marcher, int flags, float density_bias,
                         float cut_threshold, tdgp_stream_t stream);

/* ImportanceRenderer.forward (tri_plane_renderer.py:126-170) as ONE call: stratified samples -> field -> coarse march -> importance samples ->
 * field -> merge + march.  planes_hwc [B,3,H,W,F]; w0 [hid,F], b0 [hid], w1 [4,hid], b1 [4] raw module parameters; ray_o / ray_d [B,R,3];
 * u_coarse [B*R,S], u_fine [B*R,N] the two uniform draws (tri_plane_renderer.py:225,279) -> rgb [B*R,3], depth [B*R], wsum [B*R] (may be NULL),
 * final_T [B*R] (may be NULL).  ray_w > 0: the R rays are a [R/ray_w, ray_w] image (4x4-pixel tiles, same results); scale = box_size / 2;
 * flags as in tdgp_ray_march; no cut_quantile / density noise / intermediates (those go through the staged entry points above).
 * Same kernels, same bits as tdgp_sample_stratified + tdgp_triplane_field + tdgp_importance_from_coarse + tdgp_triplane_field +
 * tdgp_merge_composite.  workspace: tdgp_render_fused_workspace_bytes(B, R, S, N) bytes, 16-byte aligned, caller-owned. */
int64_t tdgp_render_fused_workspace_bytes(int B, int64_t R, int S, int N);
int     tdgp_render_fused(const float* planes_hwc, const float* w0, const float* b0, const float* w1, const float* b1,
                          const float* ray_o, const float* ray_d, const float* u_coarse, const float* u_fine,
                          float* rgb, float* depth, float* wsum, float* final_T,
                          int B, int64_t R, int ray_w, int S, int N, int F, int H, int W, int hid, float scale,
                          float t_near, float t_far, int marcher, int flags, float density_bias,
                          void* workspace, int64_t workspace_bytes, tdgp_stream_t stream);

/* [B, h*w, 3] ray colours -> [B,3,h,w] image (networks_epigraf.py:242). */
int tdgp_rays_to_image(const float* rgb, float* img, int B, int hw, tdgp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TDGP_H */
