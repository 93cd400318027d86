"""Depth and camera adaptors of the 3DGP generator (forward / eval), SURVEY.md section 8f rank 1.

Reference: `src/training/networks_depth_adaptor.py:21-99` (DepthAdaptor), `src/training/networks_camera_adaptor.py:23-134`
(ParamsAdaptor, CameraAdaptor), `src/training/layers.py:181-241` (Conv2dLayer).  Module and parameter names follow the reference so
that `G.synthesis.depth_adaptor.*` / `G.synthesis.camera_adaptor.*` entries of a reference state dict load unchanged.

* DepthAdaptor: three 5x5 convolutions (1 -> 64 -> 64 -> 64, lrelu) at image resolution and a 1x1 head: ~27 GFLOP per 256^2 image.
  They run through `tdgp_modconv2d` (styles = NULL: plain convolution, bias + activation fused in the epilogue) -- the same fp32
  MFMA kernels as the synthesis backbone, 5x5 instantiation.
* CameraAdaptor: five [B, <=280] x [<=280, 256] fully connected layers; tiny GEMMs on rocBLAS (plumbing, like the mapping network)
  with the softplus / sigmoid epilogues on `tdgp_bias_act`.
"""
import numpy as np
import torch

from .config import CameraAdaptorConfig, CameraRanges, DepthAdaptorConfig
from .generator import FullyConnectedLayer, TensorGroup, normalize_2nd_moment
from .ops import bias_act as _bias_act
from .ops import modconv as _modconv
from .ops import upfirdn2d as _upfirdn2d


class Conv2dLayer(torch.nn.Module):
    """layers.py:181-241, the form the depth adaptor uses: up = down = 1, padding k // 2, bias + activation; no hyper-modulation.
    One `tdgp_modconv2d` call: weight gain folded into the packed weights, bias and activation in the kernel epilogue."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='linear', conv_clamp=None):
        super().__init__()
        self.in_channels, self.out_channels, self.activation, self.conv_clamp = in_channels, out_channels, activation, conv_clamp
        self.padding = kernel_size // 2
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))
        self.act_gain = _bias_act.activation_funcs[activation].def_gain
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels])) if bias else None
        self.register_buffer('resample_filter', _upfirdn2d.setup_filter([1, 3, 3, 1]))      # unused at up = down = 1; kept for state-dict parity (:201)
        self._packed = None

    def _pack(self):
        stamp = (self.weight.data_ptr(), self.weight._version)
        if self._packed is None or self._packed[0] != stamp:
            self._packed = (stamp, _modconv.PackedConv((self.weight.detach().float() * self.weight_gain).contiguous()))     # w = weight * weight_gain (:223)
        return self._packed[1]

    def forward(self, x, c=None, gain=1):
        act_clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            # differentiable path (training): the same layer as conv2d_gradfix.conv2d + bias_act, both autograd functions on the HIP kernels
            from .ops import conv2d_gradfix as _cg
            y = _cg.conv2d(x, self.weight * self.weight_gain, None, padding=self.padding)
            return _bias_act.bias_act(y, self.bias, act=self.activation, gain=self.act_gain * gain, clamp=act_clamp)
        return _modconv.modconv_forward(x, self._pack(), None, bias=self.bias, up=1, demodulate=False, act=self.activation,
                                        gain=self.act_gain * gain, clamp=act_clamp)


def linear_schedule(step, val_start, val_end, period, start_step=0):
    """training_utils.py:8-18: linear ramp from val_start to val_end over `period` steps, clamped at both ends."""
    if step >= start_step + period:
        return val_end
    if step <= start_step:
        return val_start
    return val_start + (val_end - val_start) * (step - start_step) / period


class DepthAdaptor(torch.nn.Module):
    """networks_depth_adaptor.py:21-99 (forward, eval)."""

    def __init__(self, cfg: DepthAdaptorConfig, min_depth, max_depth):
        super().__init__()
        self.cfg = cfg
        self.min_depth, self.max_depth = min_depth, max_depth
        self.depth_range = max_depth - min_depth
        dims = [1] + [cfg.hid_dim] * cfg.num_hid_layers
        self.layers = torch.nn.ModuleList([Conv2dLayer(i, o, cfg.kernel_size, activation='lrelu') for i, o in zip(dims[:-1], dims[1:])])
        self.head = Conv2dLayer(dims[-1], 1, 1, activation='linear') if len(self.layers) > 0 else None
        self.register_buffer('progress_coef', torch.tensor([0.0]))
        self.near_plane_offset_raw = torch.nn.Parameter(torch.tensor([cfg.near_plane_offset_bias]).float())

    def progressive_update(self, cur_kimg):
        """:61-62: the selection probability of the raw depth anneals from uniform to selection_start_p over anneal_kimg."""
        self.progress_coef.data = torch.tensor(linear_schedule(cur_kimg, 0.0, 1.0, self.cfg.anneal_kimg)).to(self.progress_coef.device)

    @property
    def start_p(self):
        return (1.0 / (self.cfg.num_hid_layers + 1) * (1 - self.progress_coef) + self.cfg.selection_start_p * self.progress_coef).item()

    def get_near_plane_offset(self, w):
        raw = self.near_plane_offset_raw.repeat(len(w))
        return raw.sigmoid() * self.cfg.near_plane_offset_max_fraction * self.depth_range      # :44-47

    def normalize(self, x, w):
        """Depth map -> [-1, 1] with the near plane re-positioned by the learned offset (:49-60)."""
        near_shifted = (self.min_depth + self.get_near_plane_offset(w)).view(len(x), 1, 1, 1)
        mid_depth_shifted = 0.5 * (self.max_depth + near_shifted)
        depth_range_contracted = self.max_depth - near_shifted
        return (x - mid_depth_shifted) / (depth_range_contracted + 1e-12) * 2.0

    def forward(self, depth_map, w, all_outs=False):
        """depth_map [B,1,h,w] -> adapted depth [B,1,h,w].  Eval semantics: 'last' and 'random' return the head of the last
        layer (:83,:97-99), 'mean' the mean over the input and every head.  The reference adds `0.0 * outs.max()` (a
        DataParallel workaround that only matters for non-finite values); the intermediate heads are therefore only
        evaluated when they are needed ('mean', or `all_outs=True` for the parity tests)."""
        x = self.normalize(depth_map.float(), w)
        if self.head is None:
            return x
        pick_random = self.training and self.cfg.out_strategy == 'random'
        need_all = all_outs or self.cfg.out_strategy == 'mean' or pick_random
        outs = [x]
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if need_all or i == len(self.layers) - 1:
                outs.append(self.head(x))
        if all_outs:
            return torch.stack(outs).transpose(0, 1)                                       # [B, num_outs, 1, h, w]
        if pick_random:
            # :86-92,96-97: one head per sample, drawn (numpy RNG) with probabilities rising linearly from start_p at the raw depth
            num_outs = len(outs)
            idx = np.arange(num_outs)
            slope = (1 - num_outs * self.start_p) * 2 / (num_outs * (num_outs - 1))
            random_idx = torch.from_numpy(np.random.choice(idx, size=(len(x),), p=idx * slope + self.start_p))
            stacked = torch.stack(outs).transpose(0, 1)
            return stacked[torch.arange(len(x)), random_idx]
        if self.cfg.out_strategy in ('last', 'random'):
            return outs[-1]
        if self.cfg.out_strategy == 'mean':
            return torch.stack(outs).transpose(0, 1).mean(dim=1)
        raise NotImplementedError(f'Unknown out strategy: {self.cfg.out_strategy}')


class ParamsAdaptor(torch.nn.Module):
    """networks_camera_adaptor.py:23-53."""

    def __init__(self, cfg: CameraAdaptorConfig, z_dim, c_dim, in_channels, out_channels, use_z=True):
        super().__init__()
        self.cfg = cfg
        lr = cfg.lr_multiplier
        self.project_params = FullyConnectedLayer(in_channels, cfg.hid_dim, activation='softplus', lr_multiplier=lr)
        self.project_z = FullyConnectedLayer(z_dim, cfg.embed_dim, activation='softplus', lr_multiplier=lr) if use_z else None
        self.project_c = FullyConnectedLayer(c_dim, cfg.embed_dim, activation='softplus', lr_multiplier=lr) if c_dim > 0 else None
        main_in = cfg.hid_dim + (cfg.embed_dim if use_z else 0) + (cfg.embed_dim if c_dim > 0 else 0)
        self.main = torch.nn.Sequential(FullyConnectedLayer(main_in, cfg.hid_dim, activation='softplus', lr_multiplier=lr),
                                        FullyConnectedLayer(cfg.hid_dim, out_channels, activation='linear', lr_multiplier=lr))

    def forward(self, x, z=None, c=None):
        x = self.project_params(x)
        if self.project_z is not None:
            x = torch.cat([x, normalize_2nd_moment(self.project_z(z))], dim=1)
        if self.project_c is not None:
            x = torch.cat([x, normalize_2nd_moment(self.project_c(c))], dim=1)
        return self.main(x)


class CameraAdaptor(torch.nn.Module):
    """networks_camera_adaptor.py:55-134: prior camera parameters -> posterior camera parameters."""

    def __init__(self, cfg: CameraAdaptorConfig, z_dim, c_dim):
        super().__init__()
        self.cfg = cfg
        self.num_origin_cam_params = 4      # yaw, pitch, roll, radius
        self.num_look_at_cam_params = 4     # fov, look-at yaw, pitch, radius
        self.num_cam_params = 8
        self.origin_adaptor = ParamsAdaptor(cfg, z_dim, c_dim, 4, 4, use_z=False)
        self.look_at_adaptor = ParamsAdaptor(cfg, z_dim, c_dim, 8, 4)

    @staticmethod
    def unroll_camera_params(cp):
        return torch.cat([cp['angles'], cp['fov'].unsqueeze(1), cp['radius'].unsqueeze(1), cp['look_at']], dim=1)     # [N, 8]

    @staticmethod
    def roll_camera_params(cp):
        # (slices, not `cp[:, [0, 1, 2]]`: indexing with a Python list builds an index tensor on the host and copies it to the device --
        #  a synchronising transfer per call, and the adaptor runs once per generator sub-batch in the FID loop)
        return TensorGroup(angles=cp[:, 0:3], fov=cp[:, 3], radius=cp[:, 4], look_at=cp[:, 5:8])

    # Column order of the unrolled parameters: yaw, pitch, roll, fov, radius, look-at yaw, pitch, radius.  The per-column expressions of the
    # reference (:74-97) are evaluated as ONE elementwise expression over the [N, 8] block with per-column constants -- the same fp32
    # operations per element in the same order (x - lo, / (hi - lo + eps); sigmoid * scale, + lo, + 1e-5 for the pitch), so the results are the
    # reference's bits; columns that pass through use lo = 0, range = 1 (x - 0 and x / 1 are exact).  Two dozen kernel launches become four.
    _CONST_CACHE = {}

    @classmethod
    def _columns(cls, cam: CameraRanges, device, eps=1e-8):
        key = (id(cam), tuple(map(tuple, (cam.yaw, cam.pitch, cam.fov, cam.look_at_yaw, cam.look_at_pitch, cam.look_at_radius))), str(device), eps)
        hit = cls._CONST_CACHE.get(key)
        if hit is None:
            f = lambda v: torch.tensor(v, dtype=torch.float64).to(torch.float32).to(device)       # noqa: E731  python floats rounded to fp32 once, like a scalar operand
            rng = lambda r: r[1] - r[0]                                                            # noqa: E731
            n_lo = f([cam.yaw[0], cam.pitch[0], 0.0, cam.fov[0], 0.0, cam.look_at_yaw[0], cam.look_at_pitch[0], cam.look_at_radius[0]])
            n_den = f([rng(cam.yaw) + eps, rng(cam.pitch) + eps, 1.0, rng(cam.fov) + eps, 1.0, rng(cam.look_at_yaw) + eps, rng(cam.look_at_pitch) + eps,
                       rng(cam.look_at_radius) + eps])
            # :86-97 incl. the reference's look-at radius expression (it mixes the radius max with the look-at PITCH min)
            d_scale = f([rng(cam.yaw), rng(cam.pitch) - 2e-5, 0.0, rng(cam.fov), 0.0, rng(cam.look_at_yaw), rng(cam.look_at_pitch), cam.look_at_radius[1] - cam.look_at_pitch[0]])
            d_lo = f([cam.yaw[0], cam.pitch[0], 0.0, cam.fov[0], 0.0, cam.look_at_yaw[0], cam.look_at_pitch[0], cam.look_at_pitch[0]])
            d_add = f([0.0, 1e-5, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
            is_sig = torch.tensor([True, True, False, True, False, True, True, True], device=device)
            passthrough = f([0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0])                              # roll * 0.0, radius * 1.0
            hit = cls._CONST_CACHE[key] = (n_lo, n_den, d_scale, d_lo, d_add, is_sig, passthrough)
        return hit

    @staticmethod
    def normalize_camera_params(cam: CameraRanges, cp, eps=1e-8):
        """:74-84 -- yaw, pitch, fov and the look-at triple are mapped to [0, 1] by their prior ranges; roll and radius pass."""
        x = CameraAdaptor.unroll_camera_params(cp)
        n_lo, n_den = CameraAdaptor._columns(cam, x.device, eps)[:2]
        return CameraAdaptor.roll_camera_params((x - n_lo) / n_den)

    @staticmethod
    def denormalize_camera_params(cam: CameraRanges, cp):
        """:86-97: sigmoid onto the prior ranges (pitch kept 1e-5 inside its range), roll zeroed, radius passed."""
        x = CameraAdaptor.unroll_camera_params(cp)
        _, _, d_scale, d_lo, d_add, is_sig, passthrough = CameraAdaptor._columns(cam, x.device)
        y = x.sigmoid() * d_scale + d_lo + d_add
        return CameraAdaptor.roll_camera_params(torch.where(is_sig, y, x * passthrough))

    def adjust_for_prior(self, old, new):
        """:99-108: components that are not learned keep their prior value."""
        if not self.cfg.adjust_angles:
            new.angles = old['angles'] + 0.0 * new.angles
        if not self.cfg.adjust_radius:
            new.radius = old['radius'] + 0.0 * new.radius
        if not self.cfg.adjust_fov:
            new.fov = old['fov'] + 0.0 * new.fov
        if not self.cfg.adjust_look_at:
            new.look_at = old['look_at'] + 0.0 * new.look_at
        return new

    def compute_new_camera_params(self, old_norm, z, c):
        """:110-124."""
        origin_params = torch.cat([old_norm.angles, old_norm.radius.unsqueeze(1)], dim=1)
        origin_new = self.origin_adaptor(origin_params, c=c)
        look_at_in = torch.cat([origin_new[:, :3], old_norm.fov.unsqueeze(1), origin_new[:, 3:4], old_norm.look_at], dim=1)
        look_at_new = self.look_at_adaptor(look_at_in, z, c)
        new_norm = self.roll_camera_params(torch.cat([origin_new[:, :3], look_at_new[:, 0:1], origin_new[:, 3:4], look_at_new[:, 1:4]], dim=1))
        if self.cfg.residual:
            new_norm = TensorGroup(**{k: old_norm[k] + new_norm[k] for k in new_norm})
        return new_norm

    def forward(self, camera_params_old, z, c=None):
        """:126-134."""
        old = TensorGroup(**{k: camera_params_old[k].float() for k in ('angles', 'fov', 'radius', 'look_at')})
        old_norm = self.normalize_camera_params(self.cfg.camera, old)
        new_norm = self.compute_new_camera_params(old_norm, z, c)
        new = self.denormalize_camera_params(self.cfg.camera, new_norm)
        return self.adjust_for_prior(old, new)
