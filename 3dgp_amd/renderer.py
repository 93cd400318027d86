"""Volumetric renderer of the generator forward: camera, rays, tri-plane field, importance sampling, compositing.

Same callables, argument order and return tuples as the reference's Python layer
(src/training/tri_plane_renderer.py, src/training/rendering_utils.py, src/training/networks_epigraf.py):
  compute_cam2world_matrix(camera_params)                                   rendering_utils.py:194
  sample_rays(c2w, fov, resolution, patch_params=None, device=None)          tri_plane_renderer.py:487
  simple_tri_plane_renderer(x, coords, mlp, scale=1.0) -> {'rgb','sigma'}    tri_plane_renderer.py:560
  ImportanceRenderer(ray_marcher_type).forward(planes, decoder, ray_origins, ray_directions, rendering_options)
      -> (rgb [B,R,C], depth [B,R,1], weights.sum(2) [B,R,1], final_transmittance [B,R])      :126-170
  ClassicalRayMarcher / MipRayMarcher2 .forward(colors, densities, depths, rendering_options)   :353 / :300
  TriPlaneMLP                                                                 networks_epigraf.py:29-68
The reference runs these as ~40 eager PyTorch ops per 1M-point chunk; here each stage is one HIP kernel through
the C ABI (include/tdgp.h) and the run_batchwise chunking (training_utils.py:171) disappears (results do not
depend on it: every op is per-point / per-ray).

Random numbers: the reference draws `torch.rand_like([B,R,S,1])` (:225) and `torch.rand(B*R, S)` (:279) inside the
renderer.  Here they may be supplied explicitly as `rendering_options['u_coarse']` / `['u_fine']` (that is how parity
is defined); when absent they are drawn on the device in the same order and shapes.
"""
import numpy as np
import torch

from . import _lib

MARCHER_IDS = {'classical': 0, 'mip': 1}


def _marcher_flags(opts, marcher):
    flags = 0
    if opts.get('use_inf_depth', True):
        flags |= 1
    if opts.get('last_back', False) and marcher == 'classical':
        flags |= 2
    if opts.get('white_back', False) and marcher == 'mip':
        flags |= 4
    mode = opts.get('clamp_mode', 'softplus')
    if mode == 'relu':
        if marcher == 'mip':
            raise AssertionError('MipRayMarcher only supports `clamp_mode`=`softplus`!')
        flags |= 8
    elif mode != 'softplus':
        raise NotImplementedError(f'Uknown clamp mode: {mode}')
    q = opts.get('cut_quantile', 0.0)
    assert q <= 1.0, f'Wrong cut_quantile argument: {q}'
    if opts.get('sp_beta', 1.0) != 1.0:
        raise NotImplementedError('softplus beta != 1 is not on the generator path')
    if opts.get('white_back_end_idx', 0) > 0 or opts.get('fill_mode') is not None:
        raise NotImplementedError('white_back_end_idx / fill_mode debugging options are not on the generator path')
    return flags


def _quantile(x, q):
    """torch.quantile(x, q) over all elements (linear interpolation between order statistics), also past torch.quantile's
    16M-element input limit (a 256^2 x 128-sample batch of 2 already exceeds it)."""
    x = x.reshape(-1).float()
    n = x.numel()
    if n <= (1 << 24):
        return torch.quantile(x, float(q))
    xs = torch.sort(x).values
    pos = float(q) * (n - 1)
    lo = int(pos)
    return torch.lerp(xs[lo], xs[min(lo + 1, n - 1)], pos - lo)


def _cut_threshold(sigma, opts, marcher, flags):
    """The `cut_quantile` option of both marchers (tri_plane_renderer.py:324-326, 366-368; used by the non-flatness score with
    0.5): activated densities below the GLOBAL quantile -- over every ray and sample of the call -- are zeroed before alpha.
    sigma [rays, S] raw densities in march order -> the threshold the kernels apply (0.0 = off: activated densities are >= 0)."""
    q = float(opts.get('cut_quantile', 0.0))
    if q <= 0.0:
        return 0.0
    # the activated densities come from the marchers' own routine (tdgp_density_activation): the reference thresholds the very tensor it
    # took the quantile of (`x < quantile(x)`), and a softplus that differs from the kernels' by an ulp would flip the samples that sit at
    # the quantile
    bias = 0.0
    if marcher == 'mip':
        mid = (sigma[:, :-1] + sigma[:, 1:]) / 2
        sigma = torch.cat([mid, sigma[:, -1:]], dim=1) if flags & 1 else mid
        bias = float(opts.get('density_bias', 0.0))
    sigma = _lib.f32c(sigma)
    dens = torch.empty_like(sigma)
    with torch.cuda.device(sigma.device):
        _lib.call('tdgp_density_activation', sigma.data_ptr(), dens.data_ptr(), sigma.numel(), flags, bias, _lib.stream_of(sigma))
    return float(_quantile(dens, q))


# ------------------------------------------------------------------------------------------------ camera + rays

def compute_cam2world_matrix(camera_params):
    """camera_params: mapping/attribute bag with angles [B,3], radius [B], look_at [B,3] -> c2w [B,4,4]."""
    get = (lambda k: camera_params[k]) if isinstance(camera_params, dict) else (lambda k: getattr(camera_params, k))
    angles, radius, look_at = (_lib.f32c(get(k)) for k in ('angles', 'radius', 'look_at'))
    _lib.require_cuda(angles, 'camera_params.angles')
    B = angles.shape[0]
    c2w = torch.empty([B, 4, 4], dtype=torch.float32, device=angles.device)
    with torch.cuda.device(angles.device):
        _lib.call('tdgp_cam2world', angles.data_ptr(), radius.data_ptr(), look_at.data_ptr(), c2w.data_ptr(), B, _lib.stream_of(angles))
    return c2w


def camera_rays_autograd(camera_params, resolution, patch_params=None):
    """(ray_o, ray_d) [B, h*w, 3] as a differentiable function of the camera parameters: the arithmetic of tdgp_cam2world +
    tdgp_sample_rays (reference: rendering_utils.py:194-218, tri_plane_renderer.py:487-527) written with tensor ops, so that autograd
    carries d(rays) back to angles / radius / look_at / fov -- the path a trained camera adaptor needs (loss.py:76-77; the fused kernels
    are used whenever no camera parameter requires a gradient).  Cameras sit on a sphere: position = radius * (sin(pitch) sin(-yaw),
    cos(pitch), sin(pitch) cos(yaw)), the look-at point likewise from `look_at` = (yaw, pitch, radius); the camera looks down -z with the
    world's +y as its up hint; pixel (row, col) of an h x w image shoots through x = -1 + 2 col / (w - 1), y = 1 - 2 row / (h - 1),
    z = -1 / tan(fov / 2), optionally restricted to a patch (scale, offset in [0, 1] units)."""
    get = (lambda k: camera_params[k]) if isinstance(camera_params, dict) else (lambda k: getattr(camera_params, k))
    angles, radius, look_at, fov = get('angles').float(), get('radius').float(), get('look_at').float(), get('fov')
    B, dev = angles.shape[0], angles.device
    w, h = resolution

    def on_sphere(yaw, pitch, r):
        sp = torch.sin(pitch)
        return torch.stack([r * sp * torch.sin(-yaw), r * torch.cos(pitch), r * sp * torch.cos(yaw)], dim=-1)

    def unit(v):
        return v / torch.norm(v, dim=-1, keepdim=True)

    eye = on_sphere(angles[:, 0], angles[:, 1], radius)                        # [B,3]
    target = on_sphere(look_at[:, 0], look_at[:, 1], look_at[:, 2])
    fwd = unit(unit(target - eye))                                             # (the reference normalises twice)
    hint = torch.tensor([0.0, 1.0, 0.0], device=dev).expand_as(fwd)
    left = unit(torch.cross(hint, fwd, dim=-1))
    up = unit(torch.cross(fwd, left, dim=-1))
    rot = torch.stack([-left, up, -fwd], dim=-1)                               # [B,3,3]: camera axes as columns
    xs = torch.linspace(-1, 1, w, device=dev).repeat(h).unsqueeze(0)           # [1, h*w], column index fastest
    ys = torch.linspace(1, -1, h, device=dev).repeat_interleave(w).unsqueeze(0)
    if patch_params is not None:
        sc, of = patch_params['scales'].float(), patch_params['offsets'].float()
        xs = (xs + 1.0) * sc[:, 0:1] - 1.0 + of[:, 0:1] * 2.0
        ys = (ys + 1.0) * sc[:, 1:2] - 1.0 + of[:, 1:2] * 2.0
    fov_t = fov.float().reshape(-1, 1) if isinstance(fov, torch.Tensor) else torch.full([1, 1], float(fov), device=dev)
    z = -torch.ones([fov_t.shape[0], h * w], device=dev) / torch.tan(fov_t / 360 * 2 * np.pi * 0.5)
    d_cam = unit(torch.stack([xs.expand(B, -1), ys.expand(B, -1), z.expand(B, -1)], dim=2))       # [B, h*w, 3]
    ray_d = torch.bmm(d_cam, rot.transpose(1, 2))                              # rot @ d for every pixel
    ray_o = eye.unsqueeze(1).expand(B, h * w, 3)
    return ray_o, ray_d


def sample_rays(c2w, fov, resolution, patch_params=None, device=None):
    """-> (ray_o_world, ray_d_world), each [B, h*w, 3]; ray r <-> pixel (r // w, r % w).
    `w, h = resolution` exactly as the reference unpacks it (tri_plane_renderer.py:496)."""
    _lib.require_cuda(c2w, 'c2w')
    c2w = _lib.f32c(c2w)
    B = c2w.shape[0]
    w, h = resolution
    if isinstance(fov, torch.Tensor):
        fov_t, stride = _lib.f32c(fov.to(c2w.device)), 1
        if fov_t.numel() != B:
            raise RuntimeError(f'sample_rays: fov must have {B} elements')
    else:
        fov_t, stride = torch.tensor([float(fov)], dtype=torch.float32, device=c2w.device), 0
    ps = po = None
    if patch_params is not None:
        ps, po = _lib.f32c(patch_params['scales']), _lib.f32c(patch_params['offsets'])
        if tuple(ps.shape) != (B, 2) or tuple(po.shape) != (B, 2):
            raise AssertionError(f'Wrong shape: patch scales/offsets must be [{B}, 2]')
        if stride == 0:
            fov_t, stride = fov_t.expand(B).contiguous(), 1
    ray_o = torch.empty([B, h * w, 3], dtype=torch.float32, device=c2w.device)
    ray_d = torch.empty_like(ray_o)
    with torch.cuda.device(c2w.device):
        _lib.call('tdgp_sample_rays', c2w.data_ptr(), fov_t.data_ptr(), stride, _lib.ptr(ps), _lib.ptr(po), ray_o.data_ptr(), ray_d.data_ptr(),
                  B, h, w, _lib.stream_of(c2w))
    return ray_o, ray_d


# ------------------------------------------------------------------------------------------------ tri-plane field

class TriPlaneMLP(torch.nn.Module):
    """The reference's decoder (networks_epigraf.py:29-68, layers.py:22-61) with its parameter layout model.{i}.{weight,bias}:
      n_layers == 0 : `nn.Identity` (feat_dim must be out_dim + 1: the planes carry rgb + sigma themselves);
      n_layers >= 2 : FC(feat -> hid, lrelu) x (n_layers - 1), FC(hid -> backbone_out_dim, linear), backbone_out_dim = 1 + (hid if has_view_cond else out_dim)
                      (with has_view_cond the reference's forward asserts backbone_out_dim == out_dim + 1, i.e. only hid_dim == out_dim runs: kept as is).
    The 2-layer rgb + sigma form with widths from the kernel's table runs inside the fused field kernel (`fused_form`); every other form runs as the
    reference does -- eager tensor ops on the looked-up features (tdgp_triplane_features) -- correct, not fast (VERDICT r05 missing #2)."""

    class _FC(torch.nn.Module):
        def __init__(self, i, o, activation='linear'):
            super().__init__()
            self.weight = torch.nn.Parameter(torch.randn([o, i]))
            self.bias = torch.nn.Parameter(torch.zeros([o]))
            self.activation = activation
            self.weight_gain = 1.0 / float(i) ** 0.5

    def __init__(self, feat_dim, hid_dim, out_dim=3, ray_marcher_type='classical', n_layers=2, has_view_cond=False):
        super().__init__()
        self.ray_marcher_type = ray_marcher_type
        self.out_dim = out_dim
        self.feat_dim = feat_dim
        if n_layers == 0:
            assert feat_dim == out_dim + 1, f'Wrong dims: {feat_dim}, {out_dim}'
            self.backbone_out_dim = feat_dim
            self.model = torch.nn.Identity()
        else:
            self.backbone_out_dim = 1 + (hid_dim if has_view_cond else out_dim)
            dims = [feat_dim] + [hid_dim] * (n_layers - 1) + [self.backbone_out_dim]
            assert len(dims) > 2, f'We cant have just a linear layer here: nothing to modulate. Dims: {dims}'
            acts = ['lrelu'] * (len(dims) - 2) + ['linear']
            self.model = torch.nn.Sequential(*[self._FC(dims[i], dims[i + 1], a) for i, a in enumerate(acts)])

    def forward(self, feats):
        """feats [B,P,feat_dim] = the plane MEAN of the looked-up features (the reference takes [B,3,P,F] and means inside; the lookup entry point
        returns the mean) -> {'rgb': [B,P,out_dim], 'sigma': [B,P,1]} by eager tensor ops (networks_epigraf.py:46-68)."""
        from .ops import bias_act as _bias_act
        B, P, F = feats.shape
        x = feats.reshape(B * P, F)
        if not isinstance(self.model, torch.nn.Identity):
            for fc in self.model:
                w = fc.weight.to(x.dtype) * fc.weight_gain
                if fc.activation == 'linear':
                    x = torch.addmm(fc.bias.to(x.dtype).unsqueeze(0), x, w.t())
                else:
                    x = _bias_act.bias_act(x.matmul(w.t()), fc.bias.to(x.dtype), act=fc.activation)
        x = x.view(B, P, self.backbone_out_dim)
        assert x.shape[2] == self.out_dim + 1, f'Wrong shape: {tuple(x.shape)} (networks_epigraf.py:57)'
        rgb = x[..., :-1]
        if self.ray_marcher_type == 'mip':
            rgb = torch.sigmoid(rgb) * (1 + 2 * 0.001) - 0.001
        elif self.ray_marcher_type != 'classical':
            raise NotImplementedError(f'Unknown ray marcher: {self.ray_marcher_type}')
        return {'rgb': rgb, 'sigma': x[:, :, [-1]]}


_FUSED_FEAT, _FUSED_HID = (8, 16, 32, 64), (16, 32, 64, 128)          # the (feat_dim, hid_dim) table of tdgp_triplane_field (csrc/field.hip: FIELD_CASE)
_FUSED_PAIRS = {(32, 64), (32, 32), (32, 128), (32, 16), (16, 64), (16, 32), (16, 16), (8, 64), (8, 32), (8, 16), (64, 64), (64, 128)}


def fused_form(mlp):
    """Does the fused field kernel evaluate this decoder?  (2 layers, rgb + sigma, widths in its table.)"""
    model = getattr(mlp, 'model', None)
    if model is None or isinstance(model, torch.nn.Identity) or len(model) != 2:
        return False
    w0, w1 = model[0].weight, model[1].weight
    return w1.shape[0] == 4 and (w0.shape[1], w0.shape[0]) in _FUSED_PAIRS


def _mlp_params(mlp):
    """(w0, b0, w1, b1, marcher) from our TriPlaneMLP or any module shaped like the reference's -- the fused kernel's operands."""
    model = getattr(mlp, 'model', None)
    if not fused_form(mlp):
        raise NotImplementedError('the fused field kernel takes the 2-layer rgb + sigma TriPlaneMLP with (feat_dim, hid_dim) from its table; other decoders run '
                                  'through the eager path (renderer._field_eager)')
    marcher = getattr(mlp, 'ray_marcher_type', None)
    if marcher is None:
        marcher = getattr(getattr(mlp, 'cfg', None), 'ray_marcher_type', 'classical')
    ps = [_lib.f32c(t.detach()) for t in (model[0].weight, model[0].bias, model[1].weight, model[1].bias)]
    return (*ps, marcher)


def _decoder_marcher(mlp):
    marcher = getattr(mlp, 'ray_marcher_type', None)
    return marcher if marcher is not None else getattr(getattr(mlp, 'cfg', None), 'ray_marcher_type', 'classical')


class HWCPlanes:
    """Tri-planes already in the field kernel's layout [B,3,H,W,F] (what the fused backbone emits)."""

    def __init__(self, tensor):
        assert tensor.ndim == 5
        self.t = tensor


def planes_to_hwc(x):
    """NCHW [B,3F,H,W] (or the reference's 5-D view [B,3,F,H,W]) -> HWCPlanes [B,3,H,W,F]."""
    if isinstance(x, HWCPlanes):
        return x
    _lib.require_cuda(x, 'planes')
    if x.ndim == 5:
        x = x.reshape(x.shape[0], x.shape[1] * x.shape[2], x.shape[3], x.shape[4])
    assert x.shape[1] % 3 == 0, f'We use 3 planes: {x.shape}'
    x = _lib.f32c(x)
    B, c3, H, W = x.shape
    out = torch.empty([B, 3, H, W, c3 // 3], dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call('tdgp_planes_to_hwc', x.data_ptr(), out.data_ptr(), B, c3 // 3, H, W, _lib.stream_of(x))
    return HWCPlanes(out)


def _field(planes, mlp_params, scale, coords=None, ray_o=None, ray_d=None, t=None, tap_idx=None, ray_w=0, sigma_noise=None, density_noise=0.0):
    w0, b0, w1, b1, marcher = mlp_params
    p = planes.t
    B, _, H, W, F = p.shape
    if coords is not None:
        coords = _lib.f32c(coords)
        P, S = coords.shape[1], 1
    else:
        P, S = t.shape[1] * t.shape[2], t.shape[2]
    rgbs = torch.empty([B, P, 4], dtype=torch.float32, device=p.device)
    if B * P == 0:
        return rgbs
    if density_noise > 0.0:
        sigma_noise = torch.randn([B, P], device=p.device) if sigma_noise is None else _lib.f32c(sigma_noise.to(p.device))
        if sigma_noise.numel() != B * P:
            raise RuntimeError(f'sigma noise must have {B}x{P} elements')
    with torch.cuda.device(p.device):
        _lib.call('tdgp_triplane_field', p.data_ptr(), _lib.ptr(coords), _lib.ptr(ray_o), _lib.ptr(ray_d), _lib.ptr(t), w0.data_ptr(),
                  b0.data_ptr(), w1.data_ptr(), b1.data_ptr(), _lib.ptr(sigma_noise) if density_noise > 0.0 else None, float(density_noise),
                  rgbs.data_ptr(), _lib.ptr(tap_idx), B, P, S, int(ray_w), F, H, W, w0.shape[0], float(scale), MARCHER_IDS[marcher],
                  _lib.stream_of(p))
    return rgbs


def triplane_features(planes, coords, scale):
    """Mean over the three planes of the bilinear samples at coords / scale: [B,P,F] (tdgp_triplane_features; tri_plane_renderer.py:575-586 + `x.mean(dim=1)`)."""
    planes = planes_to_hwc(planes)
    coords = _lib.f32c(coords)
    p = planes.t
    B, _, H, W, F = p.shape
    P = coords.shape[1]
    feats = torch.empty([B, P, F], dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        _lib.call('tdgp_triplane_features', p.data_ptr(), coords.data_ptr(), feats.data_ptr(), B, P, F, H, W, float(scale), _lib.stream_of(p))
    return feats


def _decode_eager(mlp, feats):
    """`mlp(feats)` for our TriPlaneMLP; a module shaped like the reference's takes its own [B,3,P,F] input (three identical planes mean to the same values)."""
    if isinstance(mlp, TriPlaneMLP):
        return mlp(feats)
    return mlp(feats.unsqueeze(1).expand(-1, 3, -1, -1))


def _field_eager(planes, mlp, scale, coords=None, ray_o=None, ray_d=None, t=None, sigma_noise=None, density_noise=0.0):
    """The field for decoders outside the fused kernel's form, op for op as the reference evaluates it: points = origin + t * direction
    (tri_plane_renderer.py:141,154), lookup + plane mean, the decoder's layers as eager tensor ops, sigma noise (:183-185).  -> [B,P,4] (rgb, sigma)."""
    if coords is None:
        B, R, S = t.shape
        coords = (ray_o.unsqueeze(-2) + t.unsqueeze(-1) * ray_d.unsqueeze(-2)).reshape(B, R * S, 3)
    out = _decode_eager(mlp, triplane_features(planes, coords, scale))
    sigma = out['sigma']
    if density_noise > 0.0:
        n = torch.randn_like(sigma) if sigma_noise is None else _lib.f32c(sigma_noise.to(sigma.device)).reshape(sigma.shape)
        sigma = sigma + n * density_noise
    return torch.cat([out['rgb'], sigma], dim=-1).contiguous()


def simple_tri_plane_renderer(x, coords, mlp, scale=1.0, return_taps=False, sigma_noise=None, density_noise=0.0):
    """x: [B, 3*feat, H, W] (or HWCPlanes); coords: [B, P, 3] -> {'rgb': [B,P,3], 'sigma': [B,P,1]}.  `density_noise` > 0 adds
    `sigma_noise * density_noise` to sigma inside the kernel (`sigma_noise` [B,P,1] standard-normal draws, drawn here when None)."""
    planes = planes_to_hwc(x)
    _lib.require_cuda(coords, 'coords')
    B = planes.t.shape[0]
    assert coords.ndim == 3 and coords.shape[0] == B and coords.shape[2] == 3, f'Wrong shape: coords {tuple(coords.shape)}'
    taps = torch.empty([B, coords.shape[1], 3, 2], dtype=torch.int32, device=coords.device) if return_taps else None
    if not fused_form(mlp):
        if return_taps:
            raise NotImplementedError('tap indices are an output of the fused field kernel (2-layer decoder)')
        rgbs = _field_eager(planes, mlp, scale, coords=coords, sigma_noise=sigma_noise, density_noise=density_noise)
        return {'rgb': rgbs[..., :-1], 'sigma': rgbs[..., -1:]}
    rgbs = _field(planes, _mlp_params(mlp), scale, coords=coords, tap_idx=taps, sigma_noise=sigma_noise, density_noise=density_noise)
    out = {'rgb': rgbs[..., :3], 'sigma': rgbs[..., 3:4]}
    if return_taps:
        out['taps'] = taps
    return out


def simple_tri_plane_renderer_backward(x, coords, mlp, d_rgb, d_sigma, scale=1.0, planes_grad=True, coords_grad=False):
    """Gradients of `simple_tri_plane_renderer` (tri_plane_renderer.py:560-588 + TriPlaneMLP) for incoming d_rgb [B,P,3], d_sigma
    [B,P,1]: returns (d_planes in the field layout [B,3,H,W,F] or None, d_w0, d_b0, d_w1, d_b1) and, with `coords_grad`, d_coords
    [B,P,3] (grid_sampler's grid gradient through the plane mean and `coords / scale`).  tdgp_triplane_field_grad: forward values are
    recomputed; the plane gradient is a scatter with fp32 atomics (as torch's grid_sampler backward)."""
    planes = planes_to_hwc(x)
    _lib.require_cuda(coords, 'coords')
    coords = _lib.f32c(coords)
    w0, b0, w1, b1, marcher = _mlp_params(mlp)
    p = planes.t
    B, _, H, W, F = p.shape
    P = coords.shape[1]
    d_out = torch.cat([_lib.f32c(d_rgb), _lib.f32c(d_sigma)], dim=-1).contiguous()
    d_planes = torch.zeros_like(p) if planes_grad else None
    hid = w0.shape[0]
    d_w0, d_b0, d_w1, d_b1 = torch.empty_like(w0), torch.empty_like(b0), torch.empty_like(w1), torch.empty_like(b1)
    d_coords = torch.empty([B, P, 3], dtype=torch.float32, device=p.device) if coords_grad else None
    nbytes = int(_lib.load().tdgp_triplane_field_grad_workspace_bytes(B, P, F, hid))
    ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        _lib.call('tdgp_triplane_field_grad', p.data_ptr(), coords.data_ptr(), w0.data_ptr(), b0.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                  d_out.data_ptr(), _lib.ptr(d_planes), d_w0.data_ptr(), d_b0.data_ptr(), d_w1.data_ptr(), d_b1.data_ptr(), _lib.ptr(d_coords),
                  ws.data_ptr(), nbytes, B, P, F, H, W, hid, float(scale), MARCHER_IDS[marcher], _lib.stream_of(p))
    return (d_planes, d_w0, d_b0, d_w1, d_b1, d_coords) if coords_grad else (d_planes, d_w0, d_b0, d_w1, d_b1)


def planes_from_hwc(t):
    """Field layout [B,3,H,W,F] -> NCHW [B,3F,H,W] (the layout of the reference's plane tensor and of its gradient)."""
    B, _, H, W, F = t.shape
    return t.permute(0, 1, 4, 2, 3).reshape(B, 3 * F, H, W).contiguous()


# ------------------------------------------------------------------------------------------------ marchers

def ray_march_backward(colors, densities, depths, opts, marcher, d_rgb, d_depth=None, d_weights=None):
    """Gradients of `_march` w.r.t. colours and raw densities (autograd through tri_plane_renderer.py:299-398), tdgp_ray_march_grad.
    colors [B,R,S,C], densities / depths [B,R,S,1]; d_rgb [B,R,C], d_depth [B,R,1], d_weights [B,R,M,1] -> (d_colors, d_densities)."""
    _lib.require_cuda(colors, 'colors')
    colors, densities, depths, d_rgb = _lib.f32c(colors), _lib.f32c(densities), _lib.f32c(depths), _lib.f32c(d_rgb)
    d_depth = None if d_depth is None else _lib.f32c(d_depth)
    d_weights = None if d_weights is None else _lib.f32c(d_weights)
    B, R, S, C = colors.shape
    d_colors = torch.empty_like(colors)
    d_dens = torch.empty_like(densities)
    with torch.cuda.device(colors.device):
        _lib.call('tdgp_ray_march_grad', colors.data_ptr(), densities.data_ptr(), depths.data_ptr(), d_rgb.data_ptr(), _lib.ptr(d_depth),
                  _lib.ptr(d_weights), d_colors.data_ptr(), d_dens.data_ptr(), B * R, S, C, MARCHER_IDS[marcher], _marcher_flags(opts, marcher),
                  float(opts.get('density_bias', 0.0)), _lib.stream_of(colors))
    return d_colors, d_dens


def _march(colors, densities, depths, opts, marcher):
    _lib.require_cuda(colors, 'colors')
    colors, densities, depths = _lib.f32c(colors), _lib.f32c(densities), _lib.f32c(depths)
    B, R, S, C = colors.shape
    flags = _marcher_flags(opts, marcher)
    M = S if (marcher == 'classical' or (flags & 1)) else S - 1
    rgb = torch.empty([B, R, C], dtype=torch.float32, device=colors.device)
    depth = torch.empty([B, R, 1], dtype=torch.float32, device=colors.device)
    weights = torch.empty([B, R, M, 1], dtype=torch.float32, device=colors.device)
    final_T = torch.empty([B, R], dtype=torch.float32, device=colors.device)
    with torch.cuda.device(colors.device):
        _lib.call('tdgp_ray_march', colors.data_ptr(), densities.data_ptr(), depths.data_ptr(), rgb.data_ptr(), depth.data_ptr(),
                  weights.data_ptr(), final_T.data_ptr(), B * R, S, C, MARCHER_IDS[marcher], flags, float(opts.get('density_bias', 0.0)),
                  _cut_threshold(densities.reshape(B * R, S), opts, marcher, flags), _lib.stream_of(colors))
    return rgb, depth, weights, final_T


class ClassicalRayMarcher(torch.nn.Module):
    def forward(self, colors, densities, depths, rendering_options):
        """[B,R,S,C], [B,R,S,1], [B,R,S,1] -> (rgb [B,R,C], depth [B,R,1], weights [B,R,S,1], final_transmittance [B,R])."""
        return _march(colors, densities, depths, rendering_options, 'classical')


class MipRayMarcher2(torch.nn.Module):
    def forward(self, colors, densities, depths, rendering_options):
        return _march(colors, densities, depths, rendering_options, 'mip')


# ------------------------------------------------------------------------------------------------ importance renderer

class ImportanceRenderer(torch.nn.Module):
    fused_entry = True        # forward() through the single C-ABI call tdgp_render_fused when no intermediates / cut_quantile / density noise are asked for

    def __init__(self, ray_marcher_type: str):
        super().__init__()
        assert ray_marcher_type in ['classical', 'mip']
        self.ray_marcher_type = ray_marcher_type
        self.ray_marcher = ClassicalRayMarcher() if ray_marcher_type == 'classical' else MipRayMarcher2()

    # -- stages, exposed with the reference's method names -------------------------------------------------------
    def sample_stratified(self, ray_origins, ray_start, ray_end, num_proposal_steps, disparity_sampling=False, noise=None):
        """-> sdist_coarse [B,R,S,1] (tri_plane_renderer.py:208-235; the scalar ray_start/ray_end branch -- the tensor and
        disparity branches are never taken by the generator)."""
        if disparity_sampling or isinstance(ray_start, torch.Tensor):
            raise NotImplementedError('disparity / per-ray-limit stratified sampling is dead code on the generator path')
        if not (ray_start == 0.0 and ray_end == 1.0):
            raise NotImplementedError('stratified sampling happens in s-space [0, 1] (tri_plane_renderer.py:133-136)')
        B, R, _ = ray_origins.shape
        S = num_proposal_steps
        u = torch.rand([B, R, S, 1], device=ray_origins.device) if noise is None else _lib.f32c(noise)
        sdist = torch.empty([B, R, S, 1], dtype=torch.float32, device=ray_origins.device)
        with torch.cuda.device(u.device):
            _lib.call('tdgp_sample_stratified', u.data_ptr(), sdist.data_ptr(), None, B * R, S, MARCHER_IDS[self.ray_marcher_type], 0.0, 1.0,
                      _lib.stream_of(u))
        return sdist

    def sample_importance(self, z_vals, weights, N_importance, u=None, return_aux=False):
        """z_vals [B,R,S,1], weights [B,R,Wn,1] -> importance_z_vals [B,R,N,1] (tri_plane_renderer.py:237-295)."""
        _lib.require_cuda(z_vals, 'z_vals')
        B, R, S, _ = z_vals.shape
        z, w = _lib.f32c(z_vals), _lib.f32c(weights)
        Wn = w.shape[2]
        u = torch.rand([B * R, N_importance], device=z.device) if u is None else _lib.f32c(u)
        out = torch.empty([B, R, N_importance, 1], dtype=torch.float32, device=z.device)
        aux = None
        if return_aux:
            aux = {k: torch.empty([B * R, N_importance], dtype=torch.int32, device=z.device) for k in ('inds', 'below', 'above')}
            aux['cdf'] = torch.empty([B * R, Wn - 1], dtype=torch.float32, device=z.device)
        with torch.cuda.device(z.device):
            _lib.call('tdgp_sample_importance', z.data_ptr(), w.data_ptr(), u.data_ptr(), out.data_ptr(),
                      _lib.ptr(aux['inds']) if aux else None, _lib.ptr(aux['below']) if aux else None, _lib.ptr(aux['above']) if aux else None,
                      _lib.ptr(aux['cdf']) if aux else None, B * R, S, Wn, N_importance, MARCHER_IDS[self.ray_marcher_type], _lib.stream_of(z))
        return (out, aux) if return_aux else out

    def unify_samples(self, depths1, colors1, densities1, depths2, colors2, densities2, return_perm=False):
        """Concatenate, sort by depth (stable), gather (tri_plane_renderer.py:196-206)."""
        _lib.require_cuda(depths1, 'depths1')
        d1, c1, s1, d2, c2, s2 = (_lib.f32c(t) for t in (depths1, colors1, densities1, depths2, colors2, densities2))
        B, R, S1, C = c1.shape
        S2 = c2.shape[2]
        M = S1 + S2
        dev = d1.device
        d = torch.empty([B, R, M, 1], dtype=torch.float32, device=dev)
        c = torch.empty([B, R, M, C], dtype=torch.float32, device=dev)
        s = torch.empty([B, R, M, 1], dtype=torch.float32, device=dev)
        perm = torch.empty([B, R, M], dtype=torch.int32, device=dev) if return_perm else None
        with torch.cuda.device(dev):
            _lib.call('tdgp_unify_samples', d1.data_ptr(), c1.data_ptr(), s1.data_ptr(), S1, d2.data_ptr(), c2.data_ptr(), s2.data_ptr(), S2,
                      d.data_ptr(), c.data_ptr(), s.data_ptr(), _lib.ptr(perm), B * R, C, _lib.stream_of(d1))
        return (d, c, s, perm) if return_perm else (d, c, s)

    def run_model(self, planes, decoder, sample_coordinates, rendering_options):
        """Field evaluation at explicit coordinates (tri_plane_renderer.py:172-187); `density_noise` > 0 perturbs sigma with
        `rendering_options['sigma_noise']` (explicit draws, [B,P,1]) or fresh device-side normal draws."""
        return simple_tri_plane_renderer(planes, sample_coordinates, decoder, scale=rendering_options['box_size'] / 2,
                                         sigma_noise=rendering_options.get('sigma_noise'), density_noise=float(rendering_options.get('density_noise', 0.0)))

    # -- gradient of the whole chain -----------------------------------------------------------------------------------
    def backward(self, planes, decoder, ray_origins, ray_directions, rendering_options, d_rgb, d_depth=None, rays_grad=False):
        """Gradients of `forward` (tri_plane_renderer.py:126-170 under autograd) w.r.t. the planes and the decoder's tensors for
        incoming d_rgb [B,R,3] / d_depth [B,R,1]: returns dict(planes=[B,3F,H,W], w0, b0, w1, b1) and, with `rays_grad`, ray_o / ray_d
        [B,R,3] -- the gradient that reaches the cameras (points = origin + t * direction, :141; the field kernel's coordinate gradient
        summed over a ray's samples).

        The importance samples carry no gradient (`sample_importance` runs under no_grad, :241), so the path is
        march(unified, sorted) -> gather by the sort permutation -> {coarse, fine} field -> planes / MLP: the forward is replayed
        stage by stage (same draws: pass `u_coarse / u_fine`, and `n_coarse / n_fine` with density noise), then
        tdgp_ray_march_grad, an un-sort, and tdgp_triplane_field_grad once per pass (accumulating into the same plane gradient)."""
        opts = rendering_options
        marcher = self.ray_marcher_type
        if float(opts.get('cut_quantile', 0.0)) > 0.0:
            raise NotImplementedError('cut_quantile is an inference-only option (non-flatness score); it has no gradient path here')
        hw = planes_to_hwc(planes)
        ray_o, ray_d = _lib.f32c(ray_origins), _lib.f32c(ray_directions)
        B, R, _ = ray_o.shape
        S, N = int(opts['num_proposal_steps']), int(opts['num_fine_steps'])
        scale = opts['box_size'] / 2
        dnoise = float(opts.get('density_noise', 0.0))
        s2t = lambda s_: s_ * opts['ray_end'] + (1 - s_) * opts['ray_start']      # noqa: E731
        sd = self.sample_stratified(ray_o, 0.0, 1.0, S, noise=opts.get('u_coarse'))
        td = s2t(sd)
        pts_c = (ray_o.unsqueeze(-2) + td * ray_d.unsqueeze(-2)).reshape(B, -1, 3)
        out = simple_tri_plane_renderer(hw, pts_c, decoder, scale=scale, sigma_noise=opts.get('n_coarse'), density_noise=dnoise)
        cc, dc = out['rgb'].reshape(B, R, S, 3), out['sigma'].reshape(B, R, S, 1)
        if N > 0:
            _, _, w, _ = self.ray_marcher(cc, dc, sd, opts)
            sf = self.sample_importance(sd, w, N, u=opts.get('u_fine'))
            tf = s2t(sf)
            pts_f = (ray_o.unsqueeze(-2) + tf * ray_d.unsqueeze(-2)).reshape(B, -1, 3)
            out = simple_tri_plane_renderer(hw, pts_f, decoder, scale=scale, sigma_noise=opts.get('n_fine'), density_noise=dnoise)
            cf, df = out['rgb'].reshape(B, R, N, 3), out['sigma'].reshape(B, R, N, 1)
            d_all, c_all, s_all, perm = self.unify_samples(td, cc, dc, tf, cf, df, return_perm=True)
            g_c, g_s = ray_march_backward(c_all, s_all, d_all, opts, marcher, d_rgb, d_depth)
            idx = perm.long().unsqueeze(-1)                       # sorted slot k <- concatenated sample perm[k]
            g_c = torch.zeros_like(g_c).scatter_(2, idx.expand(-1, -1, -1, 3), g_c)
            g_s = torch.zeros_like(g_s).scatter_(2, idx, g_s)
            passes = [(pts_c, g_c[:, :, :S], g_s[:, :, :S]), (pts_f, g_c[:, :, S:], g_s[:, :, S:])]
        else:
            g_c, g_s = ray_march_backward(cc, dc, sd, opts, marcher, d_rgb, d_depth)
            passes = [(pts_c, g_c, g_s)]
        total = None
        d_ro = d_rd = None
        for (pts, gc, gs), tt in zip(passes, (td, tf) if N > 0 else (td,)):
            res = simple_tri_plane_renderer_backward(hw, pts, decoder, gc.reshape(B, -1, 3), gs.reshape(B, -1, 1), scale=scale, coords_grad=rays_grad)
            if rays_grad:
                # points = origin + t * direction (tri_plane_renderer.py:141): the depths are data (stratified draws / detached importance samples)
                dpts = res[5].reshape(B, R, -1, 3)
                d_ro = dpts.sum(2) if d_ro is None else d_ro + dpts.sum(2)
                d_rd = (dpts * tt).sum(2) if d_rd is None else d_rd + (dpts * tt).sum(2)
                res = res[:5]
            total = list(res) if total is None else [a + b for a, b in zip(total, res)]
        out = dict(planes=planes_from_hwc(total[0]), w0=total[1], b0=total[2], w1=total[3], b1=total[4])
        if rays_grad:
            out.update(ray_o=d_ro, ray_d=d_rd)
        return out

    # -- the whole chain ---------------------------------------------------------------------------------------------
    def forward(self, planes, decoder, ray_origins, ray_directions, rendering_options, return_intermediates=False):
        opts = rendering_options
        marcher = self.ray_marcher_type
        _lib.require_cuda(ray_origins, 'ray_origins')
        if isinstance(opts.get('ray_start'), str):
            raise NotImplementedError("ray_start='auto' (use_full_box) is never resolved by the reference renderer either (SURVEY.md 8a)")
        dnoise = float(opts.get('density_noise', 0.0))
        # explicit sigma-noise draws (parity tests): n_coarse [B,R*S,1], n_fine [B,R*N,1] in the reference's point order = DRAW order
        n_coarse, n_fine = opts.get('n_coarse'), opts.get('n_fine')
        planes = planes_to_hwc(planes)
        fused_mlp = fused_form(decoder)
        mlp = _mlp_params(decoder) if fused_mlp else None
        if _decoder_marcher(decoder) != marcher:
            raise RuntimeError(f'decoder was built for ray_marcher_type={_decoder_marcher(decoder)}, renderer for {marcher}')
        if fused_mlp:
            field = lambda tt, nz: _field(planes, mlp, scale, ray_o=ray_o, ray_d=ray_d, t=tt, ray_w=ray_w, sigma_noise=nz, density_noise=dnoise)      # noqa: E731
        else:         # tri_plane.mlp.n_layers != 2 / has_view_cond / widths outside the kernel's table: the reference's own op sequence (eager), same chain around it
            if getattr(decoder, 'out_dim', 3) != 3:
                raise NotImplementedError('the march kernels composite rgb + sigma (out_dim == 3)')
            field = lambda tt, nz: _field_eager(planes, decoder, scale, ray_o=ray_o, ray_d=ray_d, t=tt, sigma_noise=nz, density_noise=dnoise)              # noqa: E731
        ray_o, ray_d = _lib.f32c(ray_origins), _lib.f32c(ray_directions)
        B, R, _ = ray_o.shape
        S, N = int(opts['num_proposal_steps']), int(opts['num_fine_steps'])
        t_near, t_far = float(opts['ray_start']), float(opts['ray_end'])
        scale = opts['box_size'] / 2
        flags = _marcher_flags(opts, marcher)
        mid, dbias = MARCHER_IDS[marcher], float(opts.get('density_bias', 0.0))
        dev = ray_o.device
        u_coarse = opts.get('u_coarse')
        u_coarse = torch.rand([B, R, S, 1], device=dev) if u_coarse is None else _lib.f32c(u_coarse.to(dev))
        if u_coarse.numel() != B * R * S:
            raise RuntimeError(f'u_coarse must have {B}x{R}x{S} elements')
        # rays of an h x w image in row-major order (sample_rays): let the field kernel walk 4x4-pixel tiles
        ray_w = opts.get('ray_grid_w')
        if ray_w is None:
            side = int(round(R ** 0.5))
            ray_w = side if side * side == R else 0
        stream = _lib.stream_of(ray_o)
        cut_on = float(opts.get('cut_quantile', 0.0)) > 0.0
        if (self.fused_entry and fused_mlp and N > 0 and not return_intermediates and not cut_on and dnoise == 0.0 and B * R > 0):
            # the plain forward: ONE C-ABI call (tdgp_render_fused) -- the same five kernels the staged path below issues, same bits
            u_fine = opts.get('u_fine')
            u_fine = torch.rand([B * R, N], device=dev) if u_fine is None else _lib.f32c(u_fine.to(dev))
            if u_fine.numel() != B * R * N:
                raise RuntimeError(f'u_fine must have {B * R}x{N} elements')
            p_ = planes.t
            _, _, H, W, F = p_.shape
            w0, b0, w1, b1, _ = mlp
            rgb = torch.empty([B, R, 3], dtype=torch.float32, device=dev)
            depth = torch.empty([B, R, 1], dtype=torch.float32, device=dev)
            wsum = torch.empty([B, R, 1], dtype=torch.float32, device=dev)
            final_T = torch.empty([B, R], dtype=torch.float32, device=dev)
            nbytes = _lib.load().tdgp_render_fused_workspace_bytes(B, R, S, N)
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.call('tdgp_render_fused', p_.data_ptr(), w0.data_ptr(), b0.data_ptr(), w1.data_ptr(), b1.data_ptr(), ray_o.data_ptr(), ray_d.data_ptr(),
                          u_coarse.data_ptr(), u_fine.data_ptr(), rgb.data_ptr(), depth.data_ptr(), wsum.data_ptr(), final_T.data_ptr(), B, R, int(ray_w), S, N,
                          F, H, W, w0.shape[0], float(scale), t_near, t_far, mid, flags, dbias, ws.data_ptr(), nbytes, stream)
            return rgb, depth, wsum, final_T
        with torch.cuda.device(dev):
            sdist = torch.empty([B, R, S], dtype=torch.float32, device=dev)
            tdist = torch.empty([B, R, S], dtype=torch.float32, device=dev)
            _lib.call('tdgp_sample_stratified', u_coarse.data_ptr(), sdist.data_ptr(), tdist.data_ptr(), B * R, S, mid, t_near, t_far, stream)
            rgbs_c = field(tdist, n_coarse)
            if N > 0:
                u_fine = opts.get('u_fine')
                u_fine = torch.rand([B * R, N], device=dev) if u_fine is None else _lib.f32c(u_fine.to(dev))
                if u_fine.numel() != B * R * N:
                    raise RuntimeError(f'u_fine must have {B * R}x{N} elements')
                tfine = torch.empty([B, R, N], dtype=torch.float32, device=dev)
                sfine = torch.empty([B, R, N], dtype=torch.float32, device=dev) if return_intermediates else None
                inds = torch.empty([B * R, N], dtype=torch.int32, device=dev) if return_intermediates else None
                fperm = torch.empty([B * R, N], dtype=torch.int32, device=dev) if (return_intermediates or (dnoise > 0.0 and n_fine is not None)) else None
                cut = float(opts.get('cut_quantile', 0.0)) > 0.0
                thr_c = _cut_threshold(rgbs_c[..., 3].reshape(B * R, S), opts, marcher, flags) if cut else 0.0
                _lib.call('tdgp_importance_from_coarse', rgbs_c.data_ptr(), sdist.data_ptr(), u_fine.data_ptr(), tfine.data_ptr(),
                          _lib.ptr(sfine), _lib.ptr(inds), _lib.ptr(fperm), B * R, S, N, mid, flags, dbias, thr_c, t_near, t_far, stream)
                if dnoise > 0.0 and n_fine is not None:      # the kernel evaluates the fine samples depth-sorted: carry each draw to its slot
                    n_fine = _lib.f32c(n_fine.to(dev)).reshape(B * R, N).gather(1, fperm.long())
                rgbs_f = field(tfine, n_fine)
                rgb = torch.empty([B, R, 3], dtype=torch.float32, device=dev)
                depth = torch.empty([B, R, 1], dtype=torch.float32, device=dev)
                wsum = torch.empty([B, R, 1], dtype=torch.float32, device=dev)
                final_T = torch.empty([B, R], dtype=torch.float32, device=dev)
                perm = torch.empty([B, R, S + N], dtype=torch.int32, device=dev) if (return_intermediates or (cut and marcher == 'mip')) else None
                thr_f = 0.0
                if cut:
                    sig_all = torch.cat([rgbs_c[..., 3].reshape(B * R, S), rgbs_f[..., 3].reshape(B * R, N)], dim=1)
                    if marcher == 'mip':
                        # mid-point densities depend on the merged order: one merge pass for the permutation (the fine list is stored
                        # depth-sorted, so the permutation is taken without fine_perm), then the thresholded pass
                        _lib.call('tdgp_merge_composite', rgbs_c.data_ptr(), tdist.data_ptr(), S, rgbs_f.data_ptr(), tfine.data_ptr(), N, rgb.data_ptr(),
                                  depth.data_ptr(), wsum.data_ptr(), final_T.data_ptr(), perm.data_ptr(), None, B * R, mid, flags, dbias, 0.0, stream)
                        sig_all = sig_all.gather(1, perm.reshape(B * R, S + N).long())
                    thr_f = _cut_threshold(sig_all, opts, marcher, flags)
                _lib.call('tdgp_merge_composite', rgbs_c.data_ptr(), tdist.data_ptr(), S, rgbs_f.data_ptr(), tfine.data_ptr(), N, rgb.data_ptr(),
                          depth.data_ptr(), wsum.data_ptr(), final_T.data_ptr(), _lib.ptr(perm), _lib.ptr(fperm), B * R, mid, flags, dbias, thr_f, stream)
            else:
                rgbs4 = rgbs_c.reshape(B, R, S, 4)
                rgb, depth, w, final_T = _march(rgbs4[..., :3], rgbs4[..., 3:4], sdist.reshape(B, R, S, 1), opts, marcher)
                wsum = w.sum(2)
                rgbs_f = tfine = sfine = inds = perm = fperm = None
        out = (rgb, depth, wsum, final_T)
        if return_intermediates:
            return out, dict(sdist_coarse=sdist, tdist_coarse=tdist, rgbs_coarse=rgbs_c, tdist_fine=tfine, sdist_fine=sfine, inds=inds,
                             rgbs_fine=rgbs_f, perm=perm, fine_perm=fperm)
        return out


# ------------------------------------------------------------------------------------------------ the renderer under autograd
class _RenderFunction(torch.autograd.Function):
    """ImportanceRenderer as one autograd node: forward = the fused chain, backward = `ImportanceRenderer.backward` (replayed forward +
    tdgp_ray_march_grad + tdgp_triplane_field_grad).  The stochastic draws of the forward (stratification, inverse CDF, density
    noise) are made here and kept, so the backward differentiates the very image the forward produced."""

    @staticmethod
    def forward(ctx, planes, w0, b0, w1, b1, renderer, decoder, ray_o, ray_d, opts):
        opts = dict(opts)
        B, R, _ = ray_o.shape
        S, N = int(opts['num_proposal_steps']), int(opts['num_fine_steps'])
        dev = ray_o.device
        if opts.get('u_coarse') is None:
            opts['u_coarse'] = torch.rand([B, R, S, 1], device=dev)
        if N > 0 and opts.get('u_fine') is None:
            opts['u_fine'] = torch.rand([B * R, N], device=dev)
        if float(opts.get('density_noise', 0.0)) > 0.0:
            if opts.get('n_coarse') is None:
                opts['n_coarse'] = torch.randn([B, R * S, 1], device=dev)
            if N > 0 and opts.get('n_fine') is None:
                opts['n_fine'] = torch.randn([B, R * N, 1], device=dev)
        rgb, depth, _w, _t = renderer.forward(planes.detach(), decoder, ray_o.detach(), ray_d.detach(), opts)
        ctx.save_for_backward(planes)
        ctx.state = (renderer, decoder, ray_o.detach(), ray_d.detach(), opts)
        return rgb, depth

    @staticmethod
    def backward(ctx, d_rgb, d_depth):
        planes, = ctx.saved_tensors
        renderer, decoder, ray_o, ray_d, opts = ctx.state
        if d_rgb is None:
            d_rgb = torch.zeros([ray_o.shape[0], ray_o.shape[1], 3], device=ray_o.device)
        rays_grad = ctx.needs_input_grad[7] or ctx.needs_input_grad[8]           # cameras being trained (loss.py:76-77 with learn_camera_dist)
        res = renderer.backward(planes.detach(), decoder, ray_o, ray_d, opts, d_rgb.contiguous(), None if d_depth is None else d_depth.contiguous(),
                                rays_grad=rays_grad)
        return res['planes'], res['w0'], res['b0'], res['w1'], res['b1'], None, None, res.get('ray_o'), res.get('ray_d'), None


def render_autograd(renderer, planes, decoder, ray_origins, ray_directions, rendering_options):
    """(rgb [B,R,3], depth [B,R,1]) with gradients flowing to `planes` ([B,3F,H,W]), the decoder's four tensors and -- when they carry
    a graph (camera_rays_autograd) -- the ray origins / directions."""
    m = decoder.model
    return _RenderFunction.apply(planes, m[0].weight, m[0].bias, m[1].weight, m[1].bias, renderer, decoder, ray_origins, ray_directions,
                                 rendering_options)
