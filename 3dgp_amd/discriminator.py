"""Discriminator of 3DGP (last row): forward + first-order gradients on the HIP ops.

Reference: `src/training/networks_discriminator.py:19-294` (DiscriminatorBlock, MinibatchStdLayer, DiscriminatorEpilogue,
Discriminator), `src/training/layers.py:181-360` (Conv2dLayer, ScalarEncoder1d, FourierEncoder1d, construct_log_spaced_freqs).
Module and parameter names follow the reference, so its state dict loads with strict=True.  fp32 only (`fp32_only: true` ->
`num_fp16_res = 0`, `conv_clamp = None`, train.py:275-276).

Every convolution goes through `ops.conv2d_resample` -> `ops.upfirdn2d` + `ops.conv2d_gradfix.conv2d` (gfx950 kernels forward and
backward: stride-1 'same' and the stride-2 down-sampling forms), every activation through `ops.bias_act`; the fully connected
heads, the minibatch-stddev statistic and the Fourier features are eager tensor ops as in the reference.  Second-order gradients
(R1) work: the convolution backward passes are differentiable functions over the same kernels (ops/conv2d_gradfix.py).
"""
from dataclasses import dataclass

import numpy as np
import torch

from .encoders import FourierEncoder1d, ScalarEncoder1d, construct_log_spaced_freqs  # noqa: F401  (re-exported: the reference keeps them in layers.py)
from .generator import FullyConnectedLayer, MappingNetwork
from .ops import bias_act as _bias_act
from .ops import conv2d_resample as _conv2d_resample
from .ops import upfirdn2d as _upfirdn2d


@dataclass
class DiscriminatorConfig:
    """The keys of configs/model/base.yaml:54-67 (+ training.patch.patch_params_cond) the discriminator reads."""
    c_dim: int = 0
    cbase: int = 32768
    cmax: int = 512
    fmaps: float = 1.0
    num_additional_start_blocks: int = 0
    patch_params_cond: bool = False
    hyper_mod: bool = False
    mbstd_group_size: int = 4
    map_depth: int = 2          # MappingNetwork default num_layers is 8 in StyleGAN2; 3dgp passes no mapping_kwargs -> layers.py default


class Conv2dLayer(torch.nn.Module):
    """layers.py:181-241, the general form (down-sampling, hyper-modulation) as a chain of differentiable ops."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='linear', up=1, down=1, resample_filter=(1, 3, 3, 1),
                 conv_clamp=None, trainable=True, c_dim=0, hyper_mod=False):
        super().__init__()
        self.in_channels, self.out_channels, self.activation, self.up, self.down, self.conv_clamp = in_channels, out_channels, activation, up, down, conv_clamp
        self.register_buffer('resample_filter', _upfirdn2d.setup_filter(list(resample_filter)))
        self.padding = kernel_size // 2
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))
        self.act_gain = _bias_act.activation_funcs[activation].def_gain
        weight = torch.randn([out_channels, in_channels, kernel_size, kernel_size])
        b = torch.zeros([out_channels]) if bias else None
        if trainable:
            self.weight = torch.nn.Parameter(weight)
            self.bias = torch.nn.Parameter(b) if b is not None else None
        else:
            self.register_buffer('weight', weight)
            if b is not None:
                self.register_buffer('bias', b)
            else:
                self.bias = None
        self.affine = FullyConnectedLayer(c_dim, in_channels, bias_init=0) if hyper_mod else None
        if hyper_mod:
            assert c_dim > 0

    def forward(self, x, c=None, gain=1):
        w = self.weight * self.weight_gain
        if self.affine is not None:
            x = x * (1.0 + self.affine(c).tanh().unsqueeze(2).unsqueeze(3))
        x = _conv2d_resample.conv2d_resample(x=x, w=w.to(x.dtype), f=self.resample_filter, up=self.up, down=self.down, padding=self.padding,
                                             flip_weight=(self.up == 1))
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        b = self.bias.to(x.dtype) if self.bias is not None else None
        return _bias_act.bias_act(x, b, act=self.activation, gain=self.act_gain * gain, clamp=clamp)


class DiscriminatorBlock(torch.nn.Module):
    """networks_discriminator.py:19-93 (residual architecture, fp32)."""

    def __init__(self, in_channels, tmp_channels, out_channels, resolution, img_channels, first_layer_idx, activation='lrelu',
                 resample_filter=(1, 3, 3, 1), conv_clamp=None, freeze_layers=0, down=2, c_dim=0, hyper_mod=False):
        assert in_channels in [0, tmp_channels]
        super().__init__()
        self.in_channels, self.resolution, self.img_channels, self.first_layer_idx = in_channels, resolution, img_channels, first_layer_idx
        self.register_buffer('resample_filter', _upfirdn2d.setup_filter(list(resample_filter)))
        self.num_layers = 0

        def trainable_gen():
            while True:
                layer_idx = self.first_layer_idx + self.num_layers
                self.num_layers += 1
                yield layer_idx >= freeze_layers
        it = trainable_gen()
        self.fromrgb = Conv2dLayer(img_channels, tmp_channels, kernel_size=1, activation=activation, c_dim=c_dim, hyper_mod=False, trainable=next(it),
                                   conv_clamp=conv_clamp)
        self.conv0 = Conv2dLayer(tmp_channels, tmp_channels, kernel_size=3, activation=activation, c_dim=c_dim, hyper_mod=False, trainable=next(it),
                                 conv_clamp=conv_clamp)
        self.conv1 = Conv2dLayer(tmp_channels, out_channels, kernel_size=3, activation=activation, down=down, c_dim=c_dim, hyper_mod=hyper_mod,
                                 trainable=next(it), resample_filter=resample_filter, conv_clamp=conv_clamp)
        self.skip = Conv2dLayer(tmp_channels, out_channels, kernel_size=1, bias=False, down=down, c_dim=c_dim, hyper_mod=False, trainable=next(it),
                                resample_filter=resample_filter)

    def forward(self, x, img, c=None, force_fp32=False):
        if x is not None:
            x = x.to(torch.float32)
        if self.in_channels == 0:
            y = self.fromrgb(img.to(torch.float32), c=c)
            x = x + y if x is not None else y
        y = self.skip(x, c=c, gain=np.sqrt(0.5))
        x = self.conv0(x, c=c)
        x = self.conv1(x, c=c, gain=np.sqrt(0.5))
        return y.add(x)


class MinibatchStdLayer(torch.nn.Module):
    """networks_discriminator.py:98-124."""

    def __init__(self, group_size, num_channels=1):
        super().__init__()
        self.group_size, self.num_channels = group_size, num_channels

    def forward(self, x):
        N, C, H, W = x.shape
        G = min(self.group_size, N) if self.group_size is not None else N
        F = self.num_channels
        c = C // F
        y = x.reshape(G, -1, F, c, H, W)
        y = y - y.mean(dim=0)
        y = y.square().mean(dim=0)
        y = (y + 1e-8).sqrt()
        y = y.mean(dim=[2, 3, 4])
        y = y.reshape(-1, F, 1, 1).repeat(G, 1, H, W)
        return torch.cat([x, y], dim=1)


class DiscriminatorEpilogue(torch.nn.Module):
    """networks_discriminator.py:128-184."""

    def __init__(self, in_channels, cmap_dim, resolution, img_channels, mbstd_group_size=4, mbstd_num_channels=1, activation='lrelu', conv_clamp=None,
                 feat_predict_dim=0):
        super().__init__()
        self.in_channels, self.cmap_dim, self.resolution, self.img_channels = in_channels, cmap_dim, resolution, img_channels
        self.mbstd = MinibatchStdLayer(group_size=mbstd_group_size, num_channels=mbstd_num_channels) if mbstd_num_channels > 0 else None
        self.conv = Conv2dLayer(in_channels + mbstd_num_channels, in_channels, kernel_size=3, activation=activation, conv_clamp=conv_clamp)
        self.fc = FullyConnectedLayer(in_channels * (resolution ** 2), in_channels, activation=activation)
        self.out = FullyConnectedLayer(in_channels, 1 if cmap_dim == 0 else cmap_dim)
        self.feat_out = None
        if feat_predict_dim > 0:
            self.feat_out = torch.nn.Sequential(FullyConnectedLayer(in_channels * (resolution ** 2), in_channels, activation=activation),
                                                FullyConnectedLayer(in_channels, feat_predict_dim))

    def forward(self, x, cmap, force_fp32=False, predict_feat=False):
        assert x.shape[1:] == (self.in_channels, self.resolution, self.resolution), f'Wrong shape: {tuple(x.shape)}'
        x = x.to(torch.float32)
        if self.mbstd is not None:
            x = self.mbstd(x)
        x = self.conv(x)
        x = x.flatten(1)
        f = self.feat_out(x) if predict_feat else None
        x = self.out(self.fc(x))
        if self.cmap_dim > 0:
            assert cmap.shape[1] == self.cmap_dim
            x = (x * cmap).sum(dim=1, keepdim=True) * (1 / np.sqrt(self.cmap_dim))
        return x, f


class Discriminator(torch.nn.Module):
    """networks_discriminator.py:188-290: forward(img, c, patch_params, camera_angles, update_emas, predict_feat) -> (logits [B], feats)."""

    def __init__(self, cfg: DiscriminatorConfig, input_resolution, img_channels, conv_clamp=None, cmap_dim=None, block_kwargs={}, epilogue_kwargs={}):
        super().__init__()
        self.cfg = cfg
        assert cfg.num_additional_start_blocks >= 0
        self.img_resolution = input_resolution * (2 ** cfg.num_additional_start_blocks)
        self.img_resolution_log2 = int(np.log2(self.img_resolution))
        self.block_resolutions = [2 ** i for i in range(self.img_resolution_log2, 2, -1)]
        self.img_channels = img_channels
        ch = {res: min(int(cfg.cbase * cfg.fmaps) // res, cfg.cmax) for res in self.block_resolutions + [4]}
        if cmap_dim is None:
            cmap_dim = ch[4]
        self.scalar_enc = ScalarEncoder1d(coord_dim=3, x_multiplier=1000.0, const_emb_dim=256) if cfg.patch_params_cond else None
        if cfg.c_dim == 0 and self.scalar_enc is None:
            cmap_dim = 0
        hyper_mod_dim = 0
        self.hyper_mod_mapping = None
        if cfg.hyper_mod:
            hyper_mod_dim = 512
            self.hyper_mod_mapping = MappingNetwork(z_dim=0, c_dim=self.scalar_enc.get_dim(), w_dim=hyper_mod_dim, num_ws=None, w_avg_beta=None,
                                                    num_layers=cfg.map_depth)
        total_c = cfg.c_dim + (0 if self.scalar_enc is None else self.scalar_enc.get_dim())
        cur = 0
        for i, res in enumerate(self.block_resolutions):
            block = DiscriminatorBlock(ch[res] if res < self.img_resolution else 0, ch[res], ch[res // 2], resolution=res, first_layer_idx=cur,
                                       down=1 if i < cfg.num_additional_start_blocks else 2, c_dim=hyper_mod_dim, hyper_mod=cfg.hyper_mod,
                                       img_channels=img_channels, conv_clamp=conv_clamp, **block_kwargs)
            setattr(self, f'b{res}', block)
            cur += block.num_layers
        self.head_mapping = None
        if cfg.c_dim > 0 or self.scalar_enc is not None:
            self.head_mapping = MappingNetwork(z_dim=0, c_dim=total_c, w_dim=cmap_dim, num_ws=None, w_avg_beta=None, num_layers=cfg.map_depth)
        self.b4 = DiscriminatorEpilogue(ch[4], cmap_dim=cmap_dim, resolution=4, img_channels=img_channels, conv_clamp=conv_clamp,
                                        **{'mbstd_group_size': cfg.mbstd_group_size, **epilogue_kwargs})

    def forward(self, img, c, patch_params=None, camera_angles=None, update_emas=False, predict_feat=False, **block_kwargs):
        B = img.shape[0]
        # networks_discriminator.py:256-281: `camera_angles` is only forwarded to the head mapping's camera encoder (camera_cond, off
        # in every 3dgp config and not built here); the reference's callers always pass it, so it is accepted and ignored.
        patch_embs = None
        if self.scalar_enc is not None:
            cond = torch.cat([patch_params['scales'][:, [0]], patch_params['offsets']], dim=1)
            assert cond.shape == (B, 3)
            patch_embs = self.scalar_enc(cond)
            c = torch.cat([c, patch_embs], dim=1)
        hyper_c = self.hyper_mod_mapping(None, patch_embs) if self.hyper_mod_mapping is not None else None
        x = None
        for res in self.block_resolutions:
            x = getattr(self, f'b{res}')(x, img, c=hyper_c, **block_kwargs)
        cmap = self.head_mapping(None, c) if self.head_mapping is not None else None
        x, f = self.b4(x, cmap, predict_feat=predict_feat)
        return x.squeeze(1), f


def seeded_discriminator(cfg, input_resolution, img_channels, seed, epilogue_kwargs={}):
    """Deterministic random weights (CPU generator): the module's own initialisation under `seed`, biases moved off zero.  The
    golden vectors of tools/gen_goldens.py:gen_discriminator are computed with exactly these weights loaded into the reference."""
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    D = Discriminator(cfg, input_resolution=input_resolution, img_channels=img_channels, epilogue_kwargs=dict(epilogue_kwargs))
    with torch.no_grad():
        for n, p in D.named_parameters():
            if n.endswith('bias'):
                p.copy_(torch.randn_like(p) * 0.2)
    torch.random.set_rng_state(state)
    return D
