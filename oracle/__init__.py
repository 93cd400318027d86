"""CPU oracle for the 3DGP generator-forward hot path -- TEST INFRASTRUCTURE ONLY.

This package is the parity checker: a plain-C restatement (``tdgp_oracle.c``)
of what the reference computes on its CPU/PyTorch path, plus numpy glue that
chains the primitives into the reference's call order.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package (``3dgp_amd/``) never does.

Pinned by ``tests/golden/*.npz`` -- vectors produced by importing the reference
itself in the build container (``tools/gen_goldens.py``): op level for every
row of SURVEY.md section 8a, end to end at three small configurations, and
(round 5) at the REAL size of BASELINE configs[0..4] (``e2e_full_c1..c4``,
``bf16_full_c5``: image, depth, the reference's own float64 run, sampled
tri-plane texels, rays and the integer rows of a strip of image rows).  The reference ships
no tests/golden vectors of its own for this path and its
native code is CUDA-only (not buildable here: no nvcc), so there is no
``oracle/_ref`` build.

Nothing here reads ``/root/reference``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libtdgp_oracle.so')
_lib = None

ACT_IDS = dict(linear=1, relu=2, lrelu=3, tanh=4, sigmoid=5, elu=6, selu=7, softplus=8, swish=9)
ACT_DEF_ALPHA = dict(linear=0, relu=0, lrelu=0.2, tanh=0, sigmoid=0, elu=0, selu=0, softplus=0, swish=0)
ACT_DEF_GAIN = dict(linear=1, relu=np.sqrt(2), lrelu=np.sqrt(2), tanh=1, sigmoid=1, elu=1, selu=1, softplus=1, swish=np.sqrt(2))


def build(force=False):
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    src = os.path.join(_HERE, 'tdgp_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-B', '_build/libtdgp_oracle.so'],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_upfirdn2d_out_size.restype = ctypes.c_int
    return _lib


def set_threads(n):
    """OpenMP thread count used by the oracle (for the timed cpu_baseline)."""
    os.environ['OMP_NUM_THREADS'] = str(int(n))
    try:
        ctypes.CDLL('libgomp.so.1').omp_set_num_threads(int(n))
    except OSError:
        pass


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


c_int, c_i64, c_float = ctypes.c_int, ctypes.c_int64, ctypes.c_float


# ---------------------------------------------------------------------------- ops

def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    x = _f(x)
    alpha = float(ACT_DEF_ALPHA[act] if alpha is None else alpha)
    gain = float(ACT_DEF_GAIN[act] if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    y = np.empty_like(x)
    if b is not None:
        b = _f(b)
        assert b.ndim == 1 and b.shape[0] == x.shape[dim]
        step = int(np.prod(x.shape[dim + 1:], dtype=np.int64))
        size_b = b.shape[0]
    else:
        step, size_b = 1, 1
    lib().orc_bias_act(_p(x), _p(b), _p(y), c_i64(x.size), c_int(size_b), c_i64(step), c_int(ACT_IDS[act]),
                       c_float(alpha), c_float(gain), c_float(clamp))
    return y


def bias_act_grad(x, b, xref, yref, dy, grad, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """bias_act plugin call with grad = 1 / 2 (bias_act.cpp:32, bias_act.cu:60-147): x = incoming (second-order) gradient."""
    x = _f(x)
    y = np.empty_like(x)
    alpha = ACT_DEF_ALPHA[act] if alpha is None else alpha
    gain = ACT_DEF_GAIN[act] if gain is None else gain
    clamp = -1.0 if clamp is None else clamp
    opt = [None if t is None else _f(t) for t in (b, xref, yref, dy)]
    step = int(np.prod(x.shape[dim + 1:])) if opt[0] is not None else 1
    lib().orc_bias_act_grad(_p(x), _p(opt[0]), _p(opt[1]), _p(opt[2]), _p(opt[3]), _p(y), c_i64(x.size), opt[0].size if opt[0] is not None else 1, c_i64(step),
                            int(grad), c_int(ACT_IDS[act]), c_float(alpha), c_float(gain), c_float(clamp))
    return y


def setup_filter(f=(1, 3, 3, 1), normalize=True, flip_filter=False, gain=1):
    """upfirdn2d.py:70-114 for the non-separable (<8 taps) case."""
    f = np.asarray(f, dtype=np.float32)
    if f.ndim == 0:
        f = f[None]
    if f.ndim == 1:
        f = np.outer(f, f).astype(np.float32)
    if normalize:
        f = (f / f.sum(dtype=np.float32)).astype(np.float32)
    if flip_filter:
        f = f[::-1, ::-1]
    f = (f * np.float32(gain ** (f.ndim / 2))).astype(np.float32)
    return np.ascontiguousarray(f)


def _parse_padding(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    padding = list(padding)
    if len(padding) == 2:
        padding = [padding[0], padding[0], padding[1], padding[1]]
    return padding


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    x = _f(x)
    if f is None:
        f = np.ones([1, 1], dtype=np.float32)
    f = _f(f)
    assert x.ndim == 4 and f.ndim == 2
    upx, upy = (up, up) if isinstance(up, int) else up
    dx, dy = (down, down) if isinstance(down, int) else down
    px0, px1, py0, py1 = _parse_padding(padding)
    n, c, h, w = x.shape
    fh, fw = f.shape
    L = lib()
    ow = L.orc_upfirdn2d_out_size(w, upx, dx, px0, px1, fw)
    oh = L.orc_upfirdn2d_out_size(h, upy, dy, py0, py1, fh)
    assert ow >= 1 and oh >= 1
    y = np.empty([n, c, oh, ow], dtype=np.float32)
    L.orc_upfirdn2d(_p(x), _p(f), _p(y), n, c, h, w, fh, fw, upx, upy, dx, dy, px0, px1, py0, py1,
                    int(bool(flip_filter)), c_float(gain))
    return y


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1):
    """upfirdn2d.py:313-348."""
    px0, px1, py0, py1 = _parse_padding(padding)
    fh, fw = f.shape
    p = [px0 + (fw + up - 1) // 2, px1 + (fw - up) // 2, py0 + (fh + up - 1) // 2, py1 + (fh - up) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * up * up)


def fc(x, weight, bias=None, act='linear', lr_multiplier=1.0):
    """FullyConnectedLayer.forward, layers.py:42-58."""
    x = _f(x)
    weight = _f(weight)
    out_f, in_f = weight.shape
    rows = x.size // in_f
    y = np.empty(list(x.shape[:-1]) + [out_f], dtype=np.float32)
    wg = np.float32(lr_multiplier / np.sqrt(in_f))
    b = None if bias is None else _f(bias)
    lib().orc_fc(_p(x), _p(weight), _p(b), _p(y), c_i64(rows), in_f, out_f, c_float(wg), c_float(lr_multiplier),
                 c_int(ACT_IDS[act]), c_float(ACT_DEF_ALPHA[act]), c_float(ACT_DEF_GAIN[act] if act != 'linear' else 1.0))
    return y


def normalize_2nd_moment(x):
    x = _f(x)
    y = np.empty_like(x)
    lib().orc_normalize_2nd_moment(_p(x), _p(y), c_i64(x.shape[0]), x.shape[1])
    return y


def round_bf16(x):
    """fp32 array -> the nearest bfloat16 values (round to nearest even), kept as fp32."""
    x = _f(x)
    y = np.empty_like(x)
    lib().orc_round_bf16(_p(x), _p(y), c_i64(x.size))
    return y


def bias_act_bf16(x, b=None, act='linear', alpha=None, gain=None, clamp=None):
    """`_bias_act_ref` (bias_act.py:91-120) on a bfloat16 tensor as torch's CPU path evaluates it: every elementwise op computes in fp32
    and rounds its result to bf16 -- bias add (the bias itself rounded first, networks_stylegan2.py:144 `self.bias.to(x.dtype)`),
    activation, gain, clamp.  (The CUDA plugin rounds once at the end; the fixtures come from the CPU path.)"""
    x = round_bf16(x)
    alpha = float(ACT_DEF_ALPHA[act] if alpha is None else alpha)
    gain = float(ACT_DEF_GAIN[act] if gain is None else gain)
    if b is not None:
        x = round_bf16(x + round_bf16(b).reshape(1, -1, *([1] * (x.ndim - 2))))
    if act != 'linear':
        x = round_bf16(bias_act(x, None, act=act, alpha=alpha, gain=1.0))
    if gain != 1:
        x = round_bf16(x * np.float32(gain))
    if clamp is not None and clamp >= 0:
        x = np.clip(x, -np.float32(clamp), np.float32(clamp))
    return x.astype(np.float32)


def modulated_conv2d(x, weight, styles, noise=None, up=1, demodulate=True, resample_filter=None, prec='f32'):
    """Eval/fused modulated conv (networks_stylegan2.py:31-88) for the layer forms on the
    path: odd k (1, 3; 5 for the depth adaptor), padding=k//2, up in {1,2}, flip_weight=(up==1).
    prec='bf16': the reduced-precision path with bfloat16 (x must hold bf16 values); see orc_modconv2d."""
    x, weight, styles = _f(x), _f(weight), _f(styles)
    B, cin, H, W = x.shape
    cout, cin2, k, k2 = weight.shape
    assert cin == cin2 and k == k2 and styles.shape == (B, cin)
    y = np.empty([B, cout, H * up, W * up], dtype=np.float32)
    mode = 0
    if noise is not None:
        noise = _f(noise)
        mode = 1 if noise.ndim == 2 else 2
    f = None if resample_filter is None else _f(resample_filter)
    if up == 2:
        assert f is not None and f.shape == (4, 4)
    lib().orc_modconv2d(_p(x), _p(weight), _p(styles), _p(noise), mode, _p(f), _p(y),
                        B, cin, cout, H, W, k, up, int(bool(demodulate)), 1 if prec == 'bf16' else 0)
    return y


def triplane_field_grad(planes, coords, w0, b0, w1, b1, d_rgb, d_sigma, scale=1.0, mlp_mode='classical', return_coords=False):
    """(d_planes [B,3F,H,W], d_w0, d_b0, d_w1, d_b1) of triplane_field for incoming d_rgb [B,P,3], d_sigma [B,P,1] (autograd through
    tri_plane_renderer.py:560-588 + networks_epigraf.py:46-68), double arithmetic.  return_coords: also d_coords [B,P,3], the gradient
    w.r.t. the sample positions (what the camera parameters are trained through, loss.py:69-83)."""
    planes, coords, w0, b0, w1, b1, d_rgb, d_sigma = (_f(a) for a in (planes, coords, w0, b0, w1, b1, d_rgb, d_sigma))
    B, c3, H, W = planes.shape
    F, hid, P = c3 // 3, w0.shape[0], coords.shape[1]
    dp, dw0, db0, dw1, db1 = np.empty_like(planes), np.empty_like(w0), np.empty_like(b0), np.empty_like(w1), np.empty_like(b1)
    dc = np.empty([B, P, 3], dtype=np.float32) if return_coords else None
    lib().orc_triplane_field_grad(_p(planes), _p(coords), _p(w0), _p(b0), _p(w1), _p(b1), _p(d_rgb), _p(d_sigma), _p(dp), _p(dw0), _p(db0), _p(dw1),
                                  _p(db1), _p(dc) if return_coords else None, B, c_i64(P), F, H, W, hid, c_float(scale), 1 if mlp_mode == 'mip' else 0)
    return (dp, dw0, db0, dw1, db1, dc) if return_coords else (dp, dw0, db0, dw1, db1)


def ray_march_grad(colors, densities, depths, d_rgb, d_depth=None, d_weights=None, mode='classical', use_inf_depth=True, last_back=False,
                   white_back=False, clamp_mode='softplus', density_bias=0.0):
    """(d_colors, d_densities) of march_classical / march_mip (autograd through tri_plane_renderer.py:299-398), double arithmetic.
    colors [B,R,S,C], densities / depths [B,R,S,1]; d_rgb [B,R,C], d_depth [B,R,1], d_weights [B,R,M,1]."""
    colors, densities, depths, d_rgb = _f(colors), _f(densities), _f(depths), _f(d_rgb)
    B, R, S, C = colors.shape
    assert S <= 512 and C <= 4
    d_depth = None if d_depth is None else _f(d_depth)
    d_weights = None if d_weights is None else _f(d_weights)
    flags = (1 if use_inf_depth else 0) | (2 if last_back else 0) | (4 if white_back else 0) | (8 if clamp_mode == 'relu' else 0)
    dc, dd = np.empty_like(colors), np.empty_like(densities)
    lib().orc_ray_march_grad(_p(colors), _p(densities), _p(depths), _p(d_rgb), _p(d_depth), _p(d_weights), _p(dc), _p(dd),
                             c_i64(B * R), S, C, 1 if mode == 'mip' else 0, flags, c_float(density_bias))
    return dc, dd


def conv2d_weight_grad(x, dy, k, stride=1, padding=0):
    """dw [Cout,Cin,k,k] of y = conv2d(x, w, stride, padding) given dy (conv2d_gradfix.py:141-150), double accumulation."""
    x, dy = _f(x), _f(dy)
    B, cin, H, W = x.shape
    _, cout, OH, OW = dy.shape
    assert OH == (H + 2 * padding - k) // stride + 1 and OW == (W + 2 * padding - k) // stride + 1
    dw = np.empty([cout, cin, k, k], dtype=np.float32)
    lib().orc_conv2d_weight_grad(_p(x), _p(dy), _p(dw), B, cin, cout, H, W, OH, OW, k, stride, padding)
    return dw


def conv2d_same(x, w):
    """Stride-1 correlation with padding k // 2 (the unmodulated form of orc_modconv2d)."""
    x = _f(x)
    return modulated_conv2d(x, w, np.ones([x.shape[0], x.shape[1]], np.float32), demodulate=False)


def conv2d_input_grad(dy, w):
    """dx of y = conv2d_same(x, w): correlation of dy with the flipped, in/out-transposed weights (conv2d_gradfix.py:126-129)."""
    wt = np.ascontiguousarray(_f(w)[:, :, ::-1, ::-1].transpose(1, 0, 2, 3))
    return conv2d_same(dy, wt)


def cam2world(angles, radius, look_at):
    angles, radius, look_at = _f(angles), _f(radius), _f(look_at)
    B = angles.shape[0]
    m = np.empty([B, 4, 4], dtype=np.float32)
    lib().orc_cam2world(_p(angles), _p(radius), _p(look_at), _p(m), B)
    return m


def sample_rays(c2w, fov, h, w, patch_scales=None, patch_offsets=None):
    c2w = _f(c2w)
    B = c2w.shape[0]
    fov = _f(np.broadcast_to(np.asarray(fov, dtype=np.float32), [B]))
    ps = None if patch_scales is None else _f(patch_scales)
    po = None if patch_offsets is None else _f(patch_offsets)
    o = np.empty([B, h * w, 3], dtype=np.float32)
    d = np.empty([B, h * w, 3], dtype=np.float32)
    lib().orc_sample_rays(_p(c2w), _p(fov), _p(ps), _p(po), _p(o), _p(d), B, h, w)
    return o, d


def sample_stratified(u, mode='classical'):
    """u: [..., S] uniforms -> s-space coarse samples (tri_plane_renderer.py:208-235)."""
    u = _f(u)
    S = u.shape[-1]
    s = np.empty_like(u)
    lib().orc_sample_stratified(_p(u), _p(s), c_i64(u.size // S), S, 0 if mode == 'classical' else 1)
    return s


def s_to_t(s, t_near, t_far):
    s = _f(s)
    t = np.empty_like(s)
    lib().orc_s_to_t(_p(s), _p(t), c_i64(s.size), c_float(t_near), c_float(t_far))
    return t


def ray_points(ray_o, ray_d, t):
    """[B,R,3],[B,R,3],[B,R,S] -> coords [B,R*S,3]."""
    ray_o, ray_d, t = _f(ray_o), _f(ray_d), _f(t)
    B, R, S = t.shape
    c = np.empty([B, R * S, 3], dtype=np.float32)
    lib().orc_ray_points(_p(ray_o), _p(ray_d), _p(t), _p(c), c_i64(B * R), S)
    return c


def triplane_field(planes, coords, w0, b0, w1, b1, scale, mlp_mode='classical', return_taps=False, return_feats=False):
    """simple_tri_plane_renderer + TriPlaneMLP.  planes [B,3F,H,W], coords [B,P,3]."""
    planes, coords = _f(planes), _f(coords)
    w0, b0, w1, b1 = _f(w0), _f(b0), _f(w1), _f(b1)
    B, c3, H, W = planes.shape
    F = c3 // 3
    P = coords.shape[1]
    hid = w0.shape[0]
    assert w0.shape == (hid, F) and w1.shape == (4, hid)
    rgb = np.empty([B, P, 3], dtype=np.float32)
    sigma = np.empty([B, P, 1], dtype=np.float32)
    taps = np.empty([B, P, 3, 2], dtype=np.int32) if return_taps else None
    feats = np.empty([B, P, F], dtype=np.float32) if return_feats else None
    lib().orc_triplane_field(_p(planes), _p(coords), _p(w0), _p(b0), _p(w1), _p(b1), _p(rgb), _p(sigma),
                             _p(taps), _p(feats), B, c_i64(P), F, H, W, hid, c_float(scale),
                             0 if mlp_mode == 'classical' else 1)
    out = dict(rgb=rgb, sigma=sigma)
    if return_taps:
        out['taps'] = taps
    if return_feats:
        out['feats'] = feats
    return out


def _softplus32(x):
    """F.softplus (threshold 20) as the C marchers evaluate it: log1p(exp(x)) in double, rounded to fp32."""
    x = np.asarray(x, np.float32)
    return np.where(x > 20.0, x, np.log1p(np.exp(np.minimum(x, 20.0).astype(np.float64))).astype(np.float32)).astype(np.float32)


def cut_threshold(densities, cut_quantile, mode='classical', use_inf_depth=True, clamp_mode='softplus', density_bias=0.0):
    """The threshold of the marchers' `cut_quantile` option (tri_plane_renderer.py:324-326, 366-368): torch.quantile (linear
    interpolation) over ALL activated densities of the call; densities [B,R,S,1] raw, in march order.  0.0 = off."""
    if not cut_quantile > 0.0:
        return 0.0
    assert cut_quantile <= 1.0
    d = np.asarray(densities, np.float32)[..., 0]
    if mode == 'classical':
        act = np.maximum(d, 0) if clamp_mode == 'relu' else _softplus32(d)
    else:
        mid = ((d[..., :-1] + d[..., 1:]) / np.float32(2)).astype(np.float32)
        if use_inf_depth:
            mid = np.concatenate([mid, d[..., -1:]], axis=-1)
        act = _softplus32((mid + np.float32(density_bias)).astype(np.float32))
    return float(np.quantile(act.reshape(-1).astype(np.float32), np.float32(cut_quantile)))


def march_classical(colors, densities, depths, use_inf_depth=True, clamp_mode='softplus', last_back=False, cut_quantile=0.0):
    """[B,R,S,C],[B,R,S,1],[B,R,S,1] -> rgb [B,R,C], depth [B,R,1], weights [B,R,S,1], final_T [B,R]."""
    colors, densities, depths = _f(colors), _f(densities), _f(depths)
    thr = cut_threshold(densities, cut_quantile, 'classical', use_inf_depth, clamp_mode)
    B, R, S, C = colors.shape
    rgb = np.empty([B, R, C], dtype=np.float32)
    dep = np.empty([B, R, 1], dtype=np.float32)
    wts = np.empty([B, R, S, 1], dtype=np.float32)
    fT = np.empty([B, R], dtype=np.float32)
    lib().orc_march_classical(_p(colors), _p(densities), _p(depths), _p(rgb), _p(dep), _p(wts), _p(fT),
                              c_i64(B * R), S, C, int(bool(use_inf_depth)), int(clamp_mode == 'relu'), int(bool(last_back)), c_float(thr))
    return rgb, dep, wts, fT


def march_mip(colors, densities, depths, use_inf_depth=True, density_bias=0.0, white_back=False, cut_quantile=0.0):
    colors, densities, depths = _f(colors), _f(densities), _f(depths)
    thr = cut_threshold(densities, cut_quantile, 'mip', use_inf_depth, 'softplus', density_bias)
    B, R, S, C = colors.shape
    M = S if use_inf_depth else S - 1
    rgb = np.empty([B, R, C], dtype=np.float32)
    dep = np.empty([B, R, 1], dtype=np.float32)
    wts = np.empty([B, R, M, 1], dtype=np.float32)
    fT = np.empty([B, R], dtype=np.float32)
    lib().orc_march_mip(_p(colors), _p(densities), _p(depths), _p(rgb), _p(dep), _p(wts), _p(fT),
                        c_i64(B * R), S, C, int(bool(use_inf_depth)), c_float(density_bias), int(bool(white_back)), c_float(thr))
    return rgb, dep, wts, fT


def torch_sum(x):
    """torch.sum(x, -1) of the CPU path for a 2-D fp32 array, in torch's own accumulation order (orc_torch_sum_f32)."""
    x = _f(x)
    L = lib()
    L.orc_torch_sum_f32.restype = ctypes.c_float
    rows = x.reshape(-1, x.shape[-1])
    return np.array([L.orc_torch_sum_f32(_p(r), c_i64(r.size)) for r in rows], dtype=np.float32).reshape(x.shape[:-1])


def sample_importance(z_vals, weights, u, mode='classical', return_aux=False):
    """z_vals [B,R,S,1], weights [B,R,Wn,1], u [B*R,N] -> sdist_fine [B,R,N,1] (+ inds/below/above/cdf)."""
    z_vals, weights, u = _f(z_vals), _f(weights), _f(u)
    B, R, S, _ = z_vals.shape
    Wn = weights.shape[2]
    N = u.shape[-1]
    rays = B * R
    out = np.empty([B, R, N, 1], dtype=np.float32)
    if return_aux:
        inds = np.empty([rays, N], dtype=np.int64)
        below = np.empty([rays, N], dtype=np.int64)
        above = np.empty([rays, N], dtype=np.int64)
        cdf = np.empty([rays, Wn - 1], dtype=np.float32)
    else:
        inds = below = above = cdf = None
    lib().orc_sample_importance(_p(z_vals), _p(weights), _p(u), _p(out), _p(inds), _p(below), _p(above), _p(cdf),
                                c_i64(rays), S, Wn, N, 0 if mode == 'classical' else 1)
    if return_aux:
        return out, dict(inds=inds, below=below, above=above, cdf=cdf)
    return out


def unify_samples(d1, c1, s1, d2, c2, s2, return_perm=False):
    d1, c1, s1, d2, c2, s2 = map(_f, (d1, c1, s1, d2, c2, s2))
    B, R, S1, C = c1.shape
    S2 = c2.shape[2]
    M = S1 + S2
    d = np.empty([B, R, M, 1], dtype=np.float32)
    c = np.empty([B, R, M, C], dtype=np.float32)
    s = np.empty([B, R, M, 1], dtype=np.float32)
    perm = np.empty([B, R, M], dtype=np.int64) if return_perm else None
    lib().orc_unify_samples(_p(d1), _p(c1), _p(s1), S1, _p(d2), _p(c2), _p(s2), S2, _p(d), _p(c), _p(s), _p(perm),
                            c_i64(B * R), C)
    return (d, c, s, perm) if return_perm else (d, c, s)


from .pipeline import (mapping_forward, synthesis_backbone, importance_render, importance_render_grad, synthesis_forward,  # noqa: E402,F401
                       generator_forward, block_resolutions, channels_dict, conv2d_layer, depth_adaptor_forward,
                       camera_adaptor_forward)
