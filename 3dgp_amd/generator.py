"""Generator forward (z, c, camera -> RGB) on the HIP kernels.

Module tree, attribute names and state-dict keys are the reference's (verified by a strict
`load_state_dict` into the reference Generator in tools/gen_goldens.py), so a state-dict exported from a reference
checkpoint loads unchanged:
  Generator                 src/training/networks_epigraf.py:266-291
    .mapping   MappingNetwork            src/training/layers.py:66-177
    .synthesis SynthesisNetwork          networks_epigraf.py:134-261
       .tri_plane_decoder SynthesisBlocksSequence   :73-129
            .b{res} SynthesisBlock                  networks_stylegan2.py:180-276
                 .conv0 / .conv1 SynthesisLayer     :93-150
                 .torgb ToRGBLayer                  :155-175
       .tri_plane_mlp TriPlaneMLP                   networks_epigraf.py:29-68
       .renderer ImportanceRenderer                 tri_plane_renderer.py:118-295
Inference only (eval-mode semantics: fused modconv, no density noise, no patch sampling); fp32 (`fp32_only`).
"""
import numpy as np
import torch

from . import _lib
from .config import GeneratorConfig
from .ops import bias_act as _bias_act
from .ops import modconv as _modconv
from .ops import upfirdn2d as _upfirdn2d
from . import renderer as _renderer


class TensorGroup(dict):
    """Counterpart of dnnlib.TensorGroup (src/dnnlib/util.py:66-170): a dict of tensors aligned on their first axis, with
    attribute access.  Anything that is not a key is forwarded to every member -- `g[2:5]`, `g.cpu()`, `g.clamp(-1, 1) * 0.5 + 0.5`,
    `g.repeat_interleave(4, dim=0)`, `g.mean(dim=0, keepdim=True)` -- so the harness code reads like the reference's."""

    def __getattr__(self, name):
        if name in self:
            return self[name]
        if not name.startswith('_') and hasattr(torch.Tensor, name):
            return lambda *a, **k: TensorGroup(**{key: getattr(v, name)(*a, **k) for key, v in self.items()})
        raise AttributeError(name)

    def __setattr__(self, k, v):
        self[k] = v

    def __getitem__(self, item):
        if isinstance(item, str):
            return dict.__getitem__(self, item)
        return TensorGroup(**{k: v[item] for k, v in self.items()})

    def __len__(self):
        return len(next(iter(self.values()))) if dict.__len__(self) else 0

    def _zip(self, other, fn):
        if isinstance(other, TensorGroup):
            return TensorGroup(**{k: fn(v, other[k]) for k, v in self.items()})
        return TensorGroup(**{k: fn(v, other) for k, v in self.items()})

    def __add__(self, o): return self._zip(o, lambda a, b: a + b)
    def __radd__(self, o): return self._zip(o, lambda a, b: b + a)
    def __sub__(self, o): return self._zip(o, lambda a, b: a - b)
    def __mul__(self, o): return self._zip(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._zip(o, lambda a, b: b * a)
    def __truediv__(self, o): return self._zip(o, lambda a, b: a / b)

    def reshape_each(self, shape_fn):
        return TensorGroup(**{k: v.reshape(*shape_fn(v)) for k, v in self.items()})

    @staticmethod
    def cat(groups, dim=0):
        return TensorGroup(**{k: torch.cat([g[k] for g in groups], dim=dim) for k in groups[0].keys()})


def normalize_2nd_moment(x, dim=1, eps=1e-8):
    """layers.py:16-17."""
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


class FullyConnectedLayer(torch.nn.Module):
    """layers.py:22-61.  Tiny GEMMs: torch.addmm / matmul on rocBLAS (plumbing), activation through bias_act."""

    def __init__(self, in_features, out_features, activation='linear', bias=True, lr_multiplier=1, weight_init=1, bias_init=0):
        super().__init__()
        self.in_features, self.out_features, self.activation = in_features, out_features, activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) * (weight_init / lr_multiplier))
        self.bias = torch.nn.Parameter(torch.full([out_features], float(bias_init) / lr_multiplier)) if bias else None
        self.weight_gain = lr_multiplier / np.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        w = self.weight.to(x.dtype) * self.weight_gain
        b = self.bias
        if b is not None:
            b = b.to(x.dtype)
            if self.bias_gain != 1:
                b = b * self.bias_gain
        if self.activation == 'linear' and b is not None:
            return torch.addmm(b.unsqueeze(0), x, w.t())
        return _bias_act.bias_act(x.matmul(w.t()), b, act=self.activation)


class MappingNetwork(torch.nn.Module):
    """layers.py:66-177, camera conditioning included (round 6; `camera_cond` is off in every 3dgp config of the hot path --
    configs/model/base.yaml:26 -- but it is part of the class the checkpoints are made of): with `camera_cond` the yaw / pitch of the camera,
    wrapped to [-1, 1) turns and Fourier-encoded (`ScalarEncoder1d`, x_multiplier 64; `camera_raw_scalars`: the raw values), are appended
    to the label `c` before the embedding layer; at eval time with no angles given the stored `mean_camera_params` stand in."""

    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=2, lr_multiplier=0.01, w_avg_beta=0.998, camera_cond=False, camera_cond_drop_p=0.0,
                 camera_raw_scalars=False, mean_camera_params=None):
        super().__init__()
        if camera_cond:                                                       # layers.py:84-93
            from .encoders import ScalarEncoder1d
            self.camera_scalar_enc = (ScalarEncoder1d(coord_dim=2, x_multiplier=0.0, const_emb_dim=0, use_raw=True) if camera_raw_scalars
                                      else ScalarEncoder1d(coord_dim=2, x_multiplier=64.0, const_emb_dim=0))
            c_dim = c_dim + self.camera_scalar_enc.get_dim()
            assert self.camera_scalar_enc.get_dim() > 0
        else:
            self.camera_scalar_enc = None
        self.z_dim, self.c_dim, self.w_dim, self.num_ws, self.num_layers = z_dim, c_dim, w_dim, num_ws, num_layers
        self.w_avg_beta = w_avg_beta
        self.camera_cond_drop_p = camera_cond_drop_p
        embed_features = w_dim if c_dim > 0 else 0
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        feats = [z_dim + embed_features] + [w_dim] * num_layers
        for i in range(num_layers):
            setattr(self, f'fc{i}', FullyConnectedLayer(feats[i], feats[i + 1], activation='lrelu', lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:                     # layers.py:119-120
            self.register_buffer('w_avg', torch.zeros([w_dim]))
        if mean_camera_params is not None:                                    # layers.py:122-125
            self.register_buffer('mean_camera_params', torch.as_tensor(mean_camera_params, dtype=torch.float32))
        else:
            self.mean_camera_params = None

    def forward(self, z, c, camera_angles=None, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        # layers.py:127-138.  Without camera conditioning `camera_angles` is accepted and ignored, exactly like the reference
        # (metric_utils.py:310,344 and loss.py:70 always pass it).
        if self.camera_scalar_enc is not None:
            if (not self.training) and camera_angles is None:
                if self.mean_camera_params is None:
                    raise RuntimeError('MappingNetwork(camera_cond=True): an eval forward without camera_angles needs mean_camera_params')
                camera_angles = self.mean_camera_params[:3].unsqueeze(0).repeat(len(z), 1)
            camera_angles = camera_angles[:, [0, 1]]                           # yaw and pitch (roll is always zero)
            camera_angles = camera_angles.sign() * ((camera_angles.abs() % (2.0 * np.pi)) / (2.0 * np.pi))
            embs = self.camera_scalar_enc(camera_angles)
            embs = torch.nn.functional.dropout(embs, p=self.camera_cond_drop_p, training=self.training)
            c = torch.zeros(len(embs), 0, device=embs.device) if c is None else c
            c = torch.cat([c, embs], dim=1)
        x = None
        if self.z_dim > 0:
            assert z.shape[1] == self.z_dim, f'Wrong shape: z {tuple(z.shape)}'
            x = normalize_2nd_moment(z.to(torch.float32))
        if self.c_dim > 0:
            assert c.shape[1] == self.c_dim, f'Wrong shape: c {tuple(c.shape)}'
            y = normalize_2nd_moment(self.embed(c.to(torch.float32)))
            x = torch.cat([x, y], dim=1) if x is not None else y
        for i in range(self.num_layers):
            x = getattr(self, f'fc{i}')(x)
        if update_emas and self.w_avg_beta is not None:                       # layers.py:156-159: moving average of W
            self.w_avg.copy_(x.detach().mean(dim=0).lerp(self.w_avg, self.w_avg_beta))
        if self.num_ws is not None:
            x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            if self.num_ws is None or truncation_cutoff is None:
                x = self.w_avg.lerp(x, truncation_psi)
            else:
                x[:, :truncation_cutoff] = self.w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x


class SynthesisLayer(torch.nn.Module):
    """networks_stylegan2.py:93-150: affine -> modulated 3x3 conv (optionally x2 up + FIR) -> noise -> bias -> lrelu*sqrt2,
    executed as one tdgp_modconv2d call."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, use_noise=True, activation='lrelu',
                 conv_clamp=None):
        super().__init__()
        self.in_channels, self.out_channels, self.w_dim, self.resolution = in_channels, out_channels, w_dim, resolution
        self.up, self.use_noise, self.activation, self.conv_clamp = up, use_noise, activation, conv_clamp
        self.register_buffer('resample_filter', _upfirdn2d.setup_filter([1, 3, 3, 1]))
        self.padding = kernel_size // 2
        self.act_gain = _bias_act.activation_funcs[activation].def_gain
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        if use_noise:
            self.register_buffer('noise_const', torch.randn([resolution, resolution]))
            self.noise_strength = torch.nn.Parameter(torch.zeros([]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self._const_noise = None

    def _noise(self, batch, noise_mode, device):
        assert noise_mode in ['random', 'const', 'none']
        if not self.use_noise or noise_mode == 'none':
            return None
        if noise_mode == 'random':
            return torch.randn([batch, 1, self.resolution, self.resolution], device=device) * self.noise_strength
        # noise_const * noise_strength is a function of the weights only: computed once, not once per forward
        stamp = (self.noise_const.data_ptr(), self.noise_const._version, self.noise_strength.data_ptr(), self.noise_strength._version)
        if self._const_noise is None or self._const_noise[0] != stamp:
            self._const_noise = (stamp, (self.noise_const * self.noise_strength).detach())
        return self._const_noise[1]

    def forward_autograd(self, x, w, noise_mode='random', gain=1):
        """The same layer as a chain of differentiable ops (the unfused training path, networks_stylegan2.py:130-146):
        affine -> modulated conv (stride 1 or x2 transposed + FIR) -> + noise -> bias_act."""
        styles = self.affine(w)
        if self.up == 2:
            y = _modconv.modulated_conv2d_up_autograd(x, self.weight, styles, self.resample_filter)
        else:
            y = _modconv.modulated_conv2d_autograd(x, self.weight, styles, demodulate=True)
        if self.use_noise and noise_mode == 'random':
            y = y + torch.randn([x.shape[0], 1, self.resolution, self.resolution], device=x.device) * self.noise_strength
        elif self.use_noise and noise_mode == 'const':
            y = y + self.noise_const * self.noise_strength
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return _bias_act.bias_act(y, self.bias.to(y.dtype), act=self.activation, gain=self.act_gain * gain, clamp=clamp)

    def forward(self, x, w, noise_mode='random', fused_modconv=True, gain=1, styles=None, dcoef=None):
        if styles is None:
            styles = self.affine(w)
        noise = self._noise(x.shape[0], noise_mode, x.device)
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        fir = _modconv.fir_host_array(self.resample_filter) if self.up == 2 else None
        return _modconv.modconv_forward(x, _modconv._packed(self.weight), styles, noise=noise, bias=self.bias, up=self.up, demodulate=True,
                                        act=self.activation, gain=self.act_gain * gain, clamp=clamp, fir=fir, dcoef=dcoef)


class ToRGBLayer(torch.nn.Module):
    """networks_stylegan2.py:155-175 (+ the skip add of SynthesisBlock.forward :265-269 when `skip` is given)."""

    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None):
        super().__init__()
        self.in_channels, self.out_channels, self.w_dim, self.conv_clamp = in_channels, out_channels, w_dim, conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))

    def forward_autograd(self, x, w):
        """networks_stylegan2.py:168-172 as differentiable ops."""
        styles = self.affine(w) * self.weight_gain
        y = _modconv.modulated_conv2d_autograd(x, self.weight, styles, demodulate=False)
        return _bias_act.bias_act(y, self.bias.to(y.dtype), clamp=self.conv_clamp)

    def forward(self, x, w, fused_modconv=True, styles=None, skip=None, fir=None, out_layout=0, out_feat=0):
        if styles is None:
            styles = self.affine(w) * self.weight_gain
        return _modconv.modconv_forward(x, _modconv._packed(self.weight), styles, bias=self.bias, demodulate=False, act='linear', gain=1.0,
                                        clamp=self.conv_clamp, skip=skip, fir=fir, out_layout=out_layout, out_feat=out_feat)


def _to_dtype(x, dtype):
    """x.to(dtype) of SynthesisBlock.forward (:250); fp32 -> bf16 through the library's cast (round to nearest even)."""
    if x.dtype == dtype:
        return x
    if dtype == torch.bfloat16 and x.dtype == torch.float32 and x.is_cuda:
        x = x.contiguous()
        y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        with torch.cuda.device(x.device):
            _lib.call('tdgp_cast_f32_bf16', x.data_ptr(), y.data_ptr(), x.numel(), _lib.stream_of(x))
        return y
    return x.to(dtype)


class SynthesisBlock(torch.nn.Module):
    """networks_stylegan2.py:180-276, architecture 'skip'.  `use_fp16` (:212, :237): the block keeps its activations in 16 bits --
    bfloat16 here (BASELINE configs[4]) -- on the bf16 MFMA kernels of tdgp_modconv2d_bf16; the skip image stays fp32 (:268)."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, use_noise=True, conv_clamp=None, use_fp16=False):
        super().__init__()
        self.in_channels, self.w_dim, self.resolution, self.img_channels, self.is_last = in_channels, w_dim, resolution, img_channels, is_last
        self.use_fp16 = use_fp16
        self.register_buffer('resample_filter', _upfirdn2d.setup_filter([1, 3, 3, 1]))
        self.num_conv = 0
        self.num_torgb = 1
        if in_channels == 0:
            self.const = torch.nn.Parameter(torch.randn([out_channels, resolution, resolution]))
        else:
            self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim=w_dim, resolution=resolution, up=2, use_noise=use_noise, conv_clamp=conv_clamp)
            self.num_conv += 1
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim=w_dim, resolution=resolution, use_noise=use_noise, conv_clamp=conv_clamp)
        self.num_conv += 1
        self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp)

    def forward_autograd(self, x, img, ws, **layer_kwargs):
        """networks_stylegan2.py:231-273 ('skip' architecture, fp32) as differentiable ops."""
        w_iter = iter(ws.unbind(dim=1))
        if self.in_channels == 0:
            x = self.const.to(torch.float32).unsqueeze(0).repeat([ws.shape[0], 1, 1, 1])
        else:
            x = self.conv0.forward_autograd(x, next(w_iter), **layer_kwargs)
        x = self.conv1.forward_autograd(x, next(w_iter), **layer_kwargs)
        if img is not None:
            img = _upfirdn2d.upsample2d(img, self.resample_filter)
        y = self.torgb.forward_autograd(x, next(w_iter))
        img = img.add(y) if img is not None else y
        return x, img

    def forward(self, x, img, ws, force_fp32=False, fused_modconv=None, update_emas=False, styles=None, dcoefs=None, hwc_feat=0, side_stream=None,
                overlap=True, **layer_kwargs):
        """-> (x, img).  `styles` (optional) = pre-computed [conv0?, conv1, torgb] style tensors; `hwc_feat` > 0 keeps the
        running image in the channel-last plane layout [B, C/feat, H, W, feat]."""
        assert ws.shape[1] == self.num_conv + self.num_torgb and ws.shape[2] == self.w_dim, f'Wrong shape: ws {tuple(ws.shape)}'
        w_iter = iter(ws.unbind(dim=1))
        s_iter = iter(styles) if styles is not None else iter([None] * 3)
        d_iter = iter(dcoefs) if dcoefs is not None else iter([None] * 2)
        dtype = torch.bfloat16 if self.use_fp16 and not force_fp32 else torch.float32       # :237
        if self.in_channels == 0:
            x = self.const.to(dtype).unsqueeze(0).repeat([ws.shape[0], 1, 1, 1])
        else:
            x = self.conv0(_to_dtype(x, dtype), next(w_iter), styles=next(s_iter), dcoef=next(d_iter), **layer_kwargs)      # x.to(dtype), :250
        x = self.conv1(x, next(w_iter), styles=next(s_iter), dcoef=next(d_iter), **layer_kwargs)
        fir = _modconv.fir_host_array(self.resample_filter) if img is not None else None
        if side_stream is not None and overlap and not self.is_last:
            # ToRGB of this block (reads x, the running image; HBM / instruction-bound, matrix pipe a third busy) on a second stream, next
            # to the next block's MFMA-bound x2 layer, which only needs x.  The running image lives on the side stream from block to block.
            main = torch.cuda.current_stream(x.device)
            side_stream.wait_stream(main)
            with torch.cuda.stream(side_stream):
                img = self.torgb(x, next(w_iter), styles=next(s_iter), skip=img, fir=fir, out_layout=1 if hwc_feat else 0, out_feat=hwc_feat)
            x.record_stream(side_stream)
            return x, img
        if side_stream is not None and img is not None:
            torch.cuda.current_stream(x.device).wait_stream(side_stream)         # the last ToRGB joins the streams again
            img.record_stream(torch.cuda.current_stream(x.device))
        img = self.torgb(x, next(w_iter), styles=next(s_iter), skip=img, fir=fir, out_layout=1 if hwc_feat else 0, out_feat=hwc_feat)
        return x, img


_SIDE_STREAMS = {}           # (device index, main stream handle) -> side torch.cuda.Stream; process-wide LRU, never part of a module's state
_SIDE_STREAMS_MAX = 16       # transient main streams (bench lanes, capture streams) must not accumulate side streams for the life of the process


def side_stream_of(device):
    """The side stream for work that overlaps the CURRENT stream of `device` (ToRGB beside the next block's x2 layer).  One per (GPU,
    main stream): generators that run on different streams -- the lanes of bench.py's FID loop, a GraphedGenerator being captured next
    to eager work -- do not serialise their ToRGB layers on one shared stream, and a capture never records another lane's work."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    main = torch.cuda.current_stream(idx)
    key = (idx, main.cuda_stream)
    ent = _SIDE_STREAMS.get(key)
    if ent is not None:
        _SIDE_STREAMS[key] = _SIDE_STREAMS.pop(key)          # most recently used last
        return ent
    if len(_SIDE_STREAMS) >= _SIDE_STREAMS_MAX:
        # Bounded: the least recently used entry goes (a handle whose stream was destroyed and re-issued would otherwise map a new main
        # stream onto an old side stream for ever; with the bound it is at worst shared for a while, which is only a lost overlap --
        # every use brackets the side stream with wait_stream / record_stream on both sides).
        _SIDE_STREAMS.pop(next(iter(_SIDE_STREAMS)))
    st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=torch.device('cuda', idx))
    return st


class SynthesisBlocksSequence(torch.nn.Module):
    """networks_epigraf.py:73-129: 4x4 const -> ... -> tri_plane.res; returns the (3*feat)-channel plane image."""

    def __init__(self, cfg: GeneratorConfig, out_channels):
        super().__init__()
        self.cfg = cfg
        self.out_channels = out_channels
        self.block_resolutions = cfg.block_resolutions
        ch = cfg.channels
        self.num_ws = 0
        for i, res in enumerate(self.block_resolutions):
            is_last = res == cfg.tri_plane_res
            r16 = cfg.fp16_resolution
            block = SynthesisBlock(ch[res // 2] if i > 0 else 0, ch[res], w_dim=cfg.w_dim, resolution=res, img_channels=out_channels,
                                   is_last=is_last, use_noise=cfg.use_noise, conv_clamp=cfg.conv_clamp if r16 is not None else None,
                                   use_fp16=r16 is not None and res >= r16)
            self.num_ws += block.num_conv + (block.num_torgb if is_last else 0)
            setattr(self, f'b{res}', block)
        self._affine_pack = None
        self._demod_meta = None
        self._styles_flat = None

    # ---- all style affines of the backbone in one launch (tdgp_style_affine) -------------------------------------
    def _layers(self):
        w_idx = 0
        for res in self.block_resolutions:
            blk = getattr(self, f'b{res}')
            k = 0
            if blk.in_channels != 0:
                yield blk.conv0, w_idx + k, 1.0
                k += 1
            yield blk.conv1, w_idx + k, 1.0
            k += 1
            yield blk.torgb, w_idx + k, float(blk.torgb.weight_gain)
            w_idx += blk.num_conv

    def invalidate_cache(self):
        self._affine_pack = None
        self._demod_meta = None

    def _pack_affines(self, device):
        key = tuple((l.affine.weight.data_ptr(), l.affine.weight._version, l.affine.bias.data_ptr(), l.affine.bias._version) for l, _, _ in self._layers())
        if self._affine_pack is not None and self._affine_pack['key'] == key:
            return self._affine_pack
        A, ab, scale, meta, blocks = [], [], [], [], []
        row0 = 0
        for layer, widx, post in self._layers():
            cin = layer.affine.out_features
            A.append(layer.affine.weight.detach().float())
            ab.append(layer.affine.bias.detach().float())
            scale.append(torch.full([cin], post, dtype=torch.float32))
            meta.append((widx, row0, cin))
            blocks.append((row0, cin))
            row0 += cin
        pack = dict(key=key, A=torch.cat(A).contiguous().to(device), ab=torch.cat(ab).contiguous().to(device),
                    scale=torch.cat(scale).to(device), rows=row0, blocks=blocks, meta_spec=meta, meta={})
        self._affine_pack = pack
        return pack

    def all_styles(self, ws):
        """-> list of per-layer style tensors [B, Cin_l] in layer order (conv0?, conv1, torgb per block)."""
        pack = self._pack_affines(ws.device)
        B = ws.shape[0]
        meta = pack['meta'].get(B)
        if meta is None:
            rows = []
            for widx, row0, cin in pack['meta_spec']:
                o = torch.arange(cin, dtype=torch.int32)
                rows.append(torch.stack([torch.full_like(o, widx), torch.full_like(o, row0 * B), torch.full_like(o, cin), o], 1))
            meta = pack['meta'][B] = torch.cat(rows).contiguous().to(ws.device)
        ws = _lib.f32c(ws)
        out = torch.empty([B * pack['rows']], dtype=torch.float32, device=ws.device)
        with torch.cuda.device(ws.device):
            _lib.call('tdgp_style_affine', ws.data_ptr(), pack['A'].data_ptr(), pack['ab'].data_ptr(), meta.data_ptr(), pack['scale'].data_ptr(),
                      out.data_ptr(), B, ws.shape[1], ws.shape[2], pack['rows'], _lib.stream_of(ws))
        self._styles_flat = out
        return [out[row0 * B:(row0 + cin) * B].view(B, cin) for row0, cin in pack['blocks']]

    def all_demods(self, B):
        """Demodulation coefficients of every 3x3 layer, one launch (after all_styles): list of [B, Cout_l] in layer order."""
        pack = self._affine_pack
        layers = [(layer, row0) for (layer, _, _), (row0, _) in zip(self._layers(), pack['blocks']) if isinstance(layer, SynthesisLayer)]
        key = (B, tuple((_modconv._packed(l.weight).buf.data_ptr(), l.weight._version) for l, _ in layers))
        dm = self._demod_meta
        if dm is None or dm['key'] != key:
            rows, off, views = [], 0, []
            for layer, row0 in layers:
                pk = _modconv._packed(layer.weight)
                coutp = (pk.cout + 3) // 4 * 4
                rows.append([_modconv.wsq_address(pk), row0 * B, pk.cin, pk.cout, coutp, off])
                views.append((off, pk.cout))
                off += B * pk.cout
            dm = self._demod_meta = dict(key=key, meta=torch.tensor(rows, dtype=torch.int64, device=self._styles_flat.device), total=off, views=views,
                                         max_cout=max(r[3] for r in rows))
        flat = _modconv.demod_batch(self._styles_flat, dm['meta'], dm['total'], B, dm['max_cout'])
        return [flat[off:off + B * cout].view(B, cout) for off, cout in dm['views']]

    def forward_autograd(self, ws, x=None, **block_kwargs):
        """ws -> planes [B, out_channels, R, R] through differentiable ops (every parameter and ws receive gradients)."""
        assert ws.shape[1] == self.num_ws and ws.shape[2] == self.cfg.w_dim, f'Wrong shape: ws {tuple(ws.shape)}'
        _lib.require_cuda(ws, 'ws')
        ws = ws.to(torch.float32)
        img = None
        w_idx = 0
        for res in self.block_resolutions:
            blk = getattr(self, f'b{res}')
            x, img = blk.forward_autograd(x, img, ws.narrow(1, w_idx, blk.num_conv + blk.num_torgb), **block_kwargs)
            w_idx += blk.num_conv
        return img

    def _run(self, resolutions, x, img, ws, styles, dcoefs, feat, sl=None, **block_kwargs):
        """Blocks `resolutions` on the batch slice `sl` (None = whole batch) of (ws, styles, dcoefs); x / img are already that slice."""
        cut = (lambda t: t) if sl is None else (lambda t: t[sl])
        for res in resolutions:
            blk = getattr(self, f'b{res}')
            w_idx, s_idx, d_idx = self._block_index[res]
            n = blk.num_conv + blk.num_torgb
            ov = self.overlap_torgb is True or res <= int(self.overlap_torgb)
            x, img = blk(x, img, cut(ws).narrow(1, w_idx, n), styles=[cut(t) for t in styles[s_idx:s_idx + n]],
                         dcoefs=[cut(t) for t in dcoefs[d_idx:d_idx + blk.num_conv]], hwc_feat=feat, side_stream=self._side(ws), overlap=ov, **block_kwargs)
        return x, img

    # ToRGB layers on a second stream beside the next block's x2 layer: r03 +0.5 % (B = 16) / +2.8 % (B = 4) next to the transposed-convolution
    # kernels; since the x2 layers from 32^2 -> 64^2 up run on the persistent F(4x4) grid (round 4) a concurrent kernel there takes
    # resident-block slots from a grid that owns every CU -- measured -6 % -- so by default only the blocks up to 16^2 overlap: their ToRGB
    # layers are single-digit-block launches that take ~35 us whatever the batch (a serial K loop), next to x2 layers of the same kind.
    # overlap_torgb: False = never, True = every block, an int = blocks up to that resolution (TDGP_OVERLAP_TORGB=0 / 1 / <res>); same bits always.
    @staticmethod
    def _parse_overlap(v, default=16):
        """'0' / 'false' / 'off' -> False, '1' / 'true' / 'on' / 'all' -> True, an integer -> that resolution; anything else -> the default, with a warning
        (an unparsable environment variable must not make the package unimportable)."""
        t = str(v).strip().lower()
        if t in ('0', 'false', 'off', 'no'):
            return False
        if t in ('1', 'true', 'on', 'yes', 'all'):
            return True
        try:
            return int(t)
        except ValueError:
            import warnings
            warnings.warn(f'TDGP_OVERLAP_TORGB={v!r} is not 0 / 1 / a resolution: using {default}')
            return default

    overlap_torgb = _parse_overlap.__func__(__import__('os').environ.get('TDGP_OVERLAP_TORGB', '16'))

    def _side(self, t):
        """The stream the ToRGB layers run on, per DEVICE and outside the module: a `torch.cuda.Stream` in `__dict__` would make the
        generator un-picklable / un-deep-copyable after its first forward (training_loop.py:459 and metric_utils.py:293,328 deep-copy G)
        and would pin the overlap to the first device the module ran on."""
        if self.overlap_torgb is False or self.overlap_torgb == 0:
            return None
        return side_stream_of(t.device)

    def _index_blocks(self):
        idx, w_idx, s_idx, d_idx = {}, 0, 0, 0
        for res in self.block_resolutions:
            blk = getattr(self, f'b{res}')
            idx[res] = (w_idx, s_idx, d_idx)
            w_idx += blk.num_conv
            s_idx += blk.num_conv + blk.num_torgb
            d_idx += blk.num_conv
        return idx

    def forward(self, ws, x=None, hwc=False, **block_kwargs):
        """ws [B, num_ws, w_dim] -> planes [B, out_channels, R, R] (NCHW), or HWCPlanes [B,3,R,R,feat] when hwc=True."""
        planes = None
        for _, img in self.forward_chunks(ws, x=x, hwc=hwc, chunk=None, **block_kwargs):
            planes = img
        return planes

    def forward_chunks(self, ws, x=None, hwc=False, chunk=None, chunk_from=128, **block_kwargs):
        """Generator over (batch slice, planes of that slice).  chunk = None: one item, the whole batch.  chunk = n: the blocks below
        `chunk_from` run on the whole batch (they are small and need the batch to fill the chip), the high-resolution blocks run
        `n` samples at a time, and each slice is yielded as soon as its planes exist -- the caller renders it while its planes and the
        block activations behind them (67-100 MB per sample at 512^2) are still in the 256 MB Infinity Cache, instead of streaming a
        batch-sized tensor (1-1.6 GB at B = 16) through HBM between every pair of kernels.  Per-sample results are unchanged (every
        op is per sample; split-K factors, which depend on the launch size, move the last bit)."""
        assert ws.shape[1] == self.num_ws and ws.shape[2] == self.cfg.w_dim, f'Wrong shape: ws {tuple(ws.shape)}'
        _lib.require_cuda(ws, 'ws')
        ws = ws.to(torch.float32)
        B = ws.shape[0]
        styles = self.all_styles(ws)
        dcoefs = self.all_demods(B)
        if getattr(self, '_block_index', None) is None:
            self._block_index = self._index_blocks()
        feat = self.out_channels // 3 if hwc else 0
        wrap = (lambda t: _renderer.HWCPlanes(t)) if hwc else (lambda t: t)
        if chunk is None or chunk >= B:
            x, img = self._run(self.block_resolutions, x, None, ws, styles, dcoefs, feat, **block_kwargs)
            yield slice(0, B), wrap(img)
            return
        head = [r for r in self.block_resolutions if r < chunk_from]
        tail = [r for r in self.block_resolutions if r >= chunk_from]
        x, img = self._run(head, x, None, ws, styles, dcoefs, feat, **block_kwargs)
        for c0 in range(0, B, chunk):
            sl = slice(c0, min(c0 + chunk, B))
            _, img_c = self._run(tail, None if x is None else x[sl], None if img is None else img[sl], ws, styles, dcoefs, feat, sl=sl, **block_kwargs)
            yield sl, wrap(img_c)


class SynthesisNetwork(torch.nn.Module):
    """networks_epigraf.py:134-261.  The depth / camera adaptors exist when the configuration carries them.  `.train()` switches
    to the training-mode forward (no autograd -- gradients are SURVEY.md 8f rank 4): rays at `train_resolution` (patch-wise
    training, `patch_params`), sigma perturbed by `nerf_noise_std` (set by `progressive_update`), random head of the depth adaptor."""

    def __init__(self, cfg: GeneratorConfig, img_resolution, img_channels=3):
        super().__init__()
        self.cfg = cfg
        self.img_resolution, self.img_channels = img_resolution, img_channels
        self.tri_plane_decoder = SynthesisBlocksSequence(cfg, out_channels=cfg.feat_dim * 3)
        self.tri_plane_mlp = _renderer.TriPlaneMLP(cfg.feat_dim, cfg.mlp_hid, out_dim=img_channels, ray_marcher_type=cfg.ray_marcher_type, n_layers=cfg.mlp_n_layers,
                                                   has_view_cond=cfg.has_view_cond)
        self.num_ws = self.tri_plane_decoder.num_ws
        self.test_resolution = img_resolution
        self.train_resolution = cfg.patch_resolution if cfg.patch_resolution is not None else img_resolution
        self.nerf_noise_std = 0.0
        self.renderer = _renderer.ImportanceRenderer(ray_marcher_type=cfg.ray_marcher_type)
        from . import adaptors as _adaptors          # (adaptors imports this module)
        self.depth_adaptor = _adaptors.DepthAdaptor(cfg.depth_adaptor, min_depth=cfg.ray_start, max_depth=cfg.ray_end) if cfg.depth_adaptor is not None else None
        self.camera_adaptor = _adaptors.CameraAdaptor(cfg.camera_adaptor, cfg.z_dim, cfg.c_dim) if cfg.camera_adaptor is not None else None
        self._default_render_options = dict(max_batch_res=cfg.max_batch_res, return_depth=False, return_depth_adapted=False, return_weights=False,
                                            concat_depth=False, cut_quantile=0.0, density_bias=cfg.density_bias)
        # Schedule of the inference forward (results do not depend on it): the blocks from `chunk_from` up and the renderer run `chunk`
        # samples at a time so that a sample's high-resolution activations and tri-planes are consumed out of the Infinity Cache
        # (SynthesisBlocksSequence.forward_chunks).  None = the whole batch through every kernel.
        self.chunk, self.chunk_from = None, 128
        self.strict_nan_propagation = __import__('os').environ.get('TDGP_STRICT_NAN', '0') not in ('', '0')

    def progressive_update(self, cur_kimg):
        """networks_epigraf.py:191-194: density-noise std decays linearly to 0 over nerf_noise_kimg_growth; the depth adaptor anneals."""
        from .adaptors import linear_schedule
        self.nerf_noise_std = linear_schedule(cur_kimg, self.cfg.nerf_noise_std_init, 0.0, self.cfg.nerf_noise_kimg_growth)
        if self.depth_adaptor is not None:
            self.depth_adaptor.progressive_update(cur_kimg)

    def rendering_options(self, render_opts):
        """networks_epigraf.py:222,226-231."""
        cfg = self.cfg
        return dict(box_size=cfg.cube_scale * 2, num_proposal_steps=cfg.num_ray_steps, clamp_mode='softplus', use_inf_depth=cfg.use_inf_depth,
                    ray_start=cfg.ray_start, ray_end=cfg.ray_end, num_fine_steps=cfg.num_ray_steps,
                    density_noise=self.nerf_noise_std if self.training else 0.0, last_back=cfg.last_back,
                    white_back=cfg.white_back, max_batch_res=render_opts['max_batch_res'], cut_quantile=render_opts['cut_quantile'],
                    density_bias=render_opts['density_bias'])

    @torch.no_grad()
    def compute_densities(self, ws, coords, max_batch_res=32, **block_kwargs):
        """networks_epigraf.py:196-208: sigma at explicit coordinates."""
        planes = self.tri_plane_decoder(ws[:, :self.tri_plane_decoder.num_ws], hwc=True, **block_kwargs)
        return _renderer.simple_tri_plane_renderer(planes, coords, self.tri_plane_mlp, scale=self.cfg.cube_scale)['sigma']

    def forward_autograd(self, ws, camera_params, patch_params=None, render_opts={}, u_coarse=None, u_fine=None, n_coarse=None, n_fine=None,
                         **block_kwargs):
        """The same forward as a differentiable graph: gradients reach every parameter of the tri-plane
        backbone, the tri-plane MLP and `ws`.  Backbone = chain of the autograd ops (modulated convolutions, bias_act, upfirdn2d),
        renderer = one autograd node (`renderer.render_autograd`), depth adaptor = conv2d_gradfix + bias_act.  When a camera parameter
        requires a gradient (the camera adaptor applied by the caller, loss.py:76-77) the rays are built by `renderer.camera_rays_autograd`
        and the renderer node returns d(rays) from the field kernel's coordinate gradient; otherwise rays come from the fused kernels."""
        render_opts = {**self._default_render_options, **render_opts}
        if (render_opts['return_depth_adapted'] or render_opts['concat_depth']) and self.depth_adaptor is None:
            raise RuntimeError('return_depth_adapted / concat_depth need cfg.depth_adaptor')
        B = ws.shape[0]
        planes = self.tri_plane_decoder.forward_autograd(ws[:, :self.tri_plane_decoder.num_ws], **block_kwargs)
        h = w = self.train_resolution if self.training else self.test_resolution
        cam = camera_params
        get = (lambda k: cam[k]) if isinstance(cam, dict) else (lambda k: getattr(cam, k))
        cam_trained = any(isinstance(get(k), torch.Tensor) and get(k).requires_grad for k in ('angles', 'fov', 'radius', 'look_at'))
        if cam_trained:                          # cameras from a trained CameraAdaptor (loss.py:76-77): rays as a differentiable graph
            ray_o, ray_d = _renderer.camera_rays_autograd(cam, (h, w), patch_params=patch_params)
        else:
            with torch.no_grad():
                c2w = _renderer.compute_cam2world_matrix(cam)
                ray_o, ray_d = _renderer.sample_rays(c2w, fov=get('fov'), resolution=(h, w), patch_params=patch_params, device=ws.device)
        opts = self.rendering_options(render_opts)
        opts['u_coarse'], opts['u_fine'], opts['n_coarse'], opts['n_fine'] = u_coarse, u_fine, n_coarse, n_fine
        opts['ray_grid_w'] = w
        rgb, depth = _renderer.render_autograd(self.renderer, planes, self.tri_plane_mlp, ray_o, ray_d, opts)
        img = rgb.reshape(B, h, w, self.img_channels).permute(0, 3, 1, 2).contiguous()
        depth = depth.reshape(B, 1, h, w)
        depth_adapted = None
        if self.depth_adaptor is not None:                                  # networks_epigraf.py:246-253
            depth_adapted = self.depth_adaptor(depth, ws[:, 0])
            img = torch.cat([img, depth_adapted], dim=1) if render_opts['concat_depth'] else img + 0.0 * depth_adapted.max()
        if render_opts['return_depth'] or render_opts['return_depth_adapted']:
            out = TensorGroup(img=img)
            if render_opts['return_depth']:
                out.depth = depth
            if render_opts['return_depth_adapted']:
                out.depth_adapted = depth_adapted
            return out
        return img

    @torch.no_grad()
    def forward(self, ws, camera_params, patch_params=None, render_opts={}, u_coarse=None, u_fine=None, n_coarse=None, n_fine=None,
                update_emas=False, **block_kwargs):
        """ws [B,num_ws,w_dim]; camera_params {angles [B,3], fov [B], radius [B], look_at [B,3]} -> img [B,3,h,w]
        (or TensorGroup(img, depth) with render_opts['return_depth']).  u_* / n_*: explicit uniform / normal draws of the renderer
        (stratification, inverse-CDF, density noise) for the parity tests; drawn on the device when absent."""
        render_opts = {**self._default_render_options, **render_opts}
        if (render_opts['return_depth_adapted'] or render_opts['concat_depth']) and self.depth_adaptor is None:
            raise RuntimeError('return_depth_adapted / concat_depth need cfg.depth_adaptor')
        B = ws.shape[0]
        h = w = self.train_resolution if self.training else self.test_resolution
        cam = camera_params
        get = (lambda k: cam[k]) if isinstance(cam, dict) else (lambda k: getattr(cam, k))
        c2w = _renderer.compute_cam2world_matrix(cam)
        ray_o, ray_d = _renderer.sample_rays(c2w, fov=get('fov'), resolution=(h, w), patch_params=patch_params, device=ws.device)
        opts = self.rendering_options(render_opts)
        opts['ray_grid_w'] = w                      # rays are the row-major pixels of an h x w image (sample_rays)
        R = h * w
        draws = dict(u_coarse=(u_coarse, [B, R, -1]), u_fine=(u_fine, [B, R, -1]), n_coarse=(n_coarse, [B, -1]), n_fine=(n_fine, [B, -1]))
        # cut_quantile thresholds at a quantile over every ray and sample of ONE renderer call (tri_plane_renderer.py:366-368): never
        # batch-chunked here.  The reference itself splits such a call -- by RAYS -- when the eval resolution exceeds max_batch_res
        # (networks_epigraf.py:232-239: run_batchwise over 2**24 // (B * num_ray_steps * 3) rays, "cut_quantile fails on large tensors"),
        # so each ray chunk takes its own quantiles (and, when the draws are not explicit, its own rand_like / rand pair in that order).
        cutting = float(opts.get('cut_quantile', 0.0)) > 0.0
        chunk = None if cutting else self.chunk
        ray_step = None
        if cutting and not self.training and (h > render_opts['max_batch_res'] or w > render_opts['max_batch_res']):
            ray_step = 2 ** 24 // (B * self.cfg.num_ray_steps * 3)
            assert ray_step >= 1, f'Wrong batch_size: {ray_step}'        # training_utils.py:180
            if ray_step >= R:
                ray_step = None                                         # run_batchwise's early exit (:185-186)
        rgb = depth = None
        for sl, planes in self.tri_plane_decoder.forward_chunks(ws[:, :self.tri_plane_decoder.num_ws], hwc=True, chunk=chunk, chunk_from=self.chunk_from,
                                                                **block_kwargs):
            whole = sl.start == 0 and sl.stop == B
            o = dict(opts)
            for k, (t, shape) in draws.items():          # the explicit draws of this batch slice, in the layouts the renderer takes
                if t is None:
                    o[k] = None
                elif whole:
                    o[k] = t
                else:
                    tc = t.reshape(shape)[sl]
                    o[k] = tc.reshape(-1, tc.shape[-1]) if k == 'u_fine' else tc
            if ray_step is not None:
                assert whole
                rgb = torch.empty([B, R, self.img_channels], dtype=torch.float32, device=ws.device)
                depth = torch.empty([B, R, 1], dtype=torch.float32, device=ws.device)
                for a in range(0, R, ray_step):
                    rs = slice(a, min(a + ray_step, R))
                    oc = dict(o, ray_grid_w=0)           # a run of rays, not an image: the field kernel takes the linear point order
                    for k in ('u_coarse', 'u_fine'):
                        if o[k] is not None:
                            tc = o[k].reshape(B, R, -1)[:, rs].contiguous()
                            oc[k] = tc.reshape(-1, tc.shape[-1]) if k == 'u_fine' else tc
                    rgb[:, rs], depth[:, rs], _w, _T = self.renderer(planes, self.tri_plane_mlp, ray_o[:, rs].contiguous(), ray_d[:, rs].contiguous(), oc)
                continue
            rgb_c, depth_c, _w, _T = self.renderer(planes, self.tri_plane_mlp, ray_o[sl], ray_d[sl], o)
            if whole:
                rgb, depth = rgb_c, depth_c
            else:
                if rgb is None:
                    rgb = torch.empty([B, R, rgb_c.shape[-1]], dtype=torch.float32, device=ws.device)
                    depth = torch.empty([B, R, 1], dtype=torch.float32, device=ws.device)
                rgb[sl], depth[sl] = rgb_c, depth_c
        img = torch.empty([B, self.img_channels, h, w], dtype=torch.float32, device=ws.device)
        with torch.cuda.device(ws.device):
            _lib.call('tdgp_rays_to_image', rgb.data_ptr(), img.data_ptr(), B, h * w, _lib.stream_of(rgb))
        depth = depth.reshape(B, 1, h, w)
        depth_adapted = None
        if self.depth_adaptor is not None:                                  # networks_epigraf.py:246-253
            # (self.training: DepthAdaptor.forward with out_strategy='random' draws np.random.choice in training mode -- the host RNG
            #  stream must advance exactly as the reference's, which always evaluates the adaptor: networks_depth_adaptor.py:86-92)
            needed = render_opts['concat_depth'] or render_opts['return_depth_adapted'] or self.strict_nan_propagation or self.training
            if needed:
                depth_adapted = self.depth_adaptor(depth, ws[:, 0])
                if render_opts['concat_depth']:
                    img = torch.cat([img, depth_adapted], dim=1)
                else:
                    img = img + 0.0 * depth_adapted.max()
            # else: ELIDED.  The reference evaluates the adaptor (27 GFLOP per 256^2 image: three 5x5 convolutions at 64 channels) and then adds
            # `0.0 * depth_adapted.max()` to the image (:253, "to avoid potential DataParallel issues") -- for finite values that is `img`, bit
            # for bit (x + 0.0 == x for every finite and infinite x; only -0.0 pixels would become +0.0, and a raw MLP output is never -0.0
            # after `+ b1`... either way equal under ==).  The adaptor draws no random numbers in eval mode, so skipping it leaves every RNG
            # stream where the reference's is.  What IS lost: a NaN / Inf inside the adaptor (non-finite adaptor weights) no longer poisons the
            # image; `strict_nan_propagation = True` (or TDGP_STRICT_NAN=1) restores the literal forward.  Training-mode forwards are never elided.
        if render_opts['return_depth'] or render_opts['return_depth_adapted']:
            out = TensorGroup(img=img)
            if render_opts['return_depth']:
                out.depth = depth
            if render_opts['return_depth_adapted']:
                out.depth_adapted = depth_adapted
            return out
        return img


class Generator(torch.nn.Module):
    """networks_epigraf.py:266-291: forward(z, c, camera_params, camera_angles_cond, truncation_psi, truncation_cutoff,
    update_emas, **synthesis_kwargs)."""

    def __init__(self, cfg: GeneratorConfig, img_resolution=None, img_channels=3, mapping_kwargs=None):
        super().__init__()
        self.cfg = cfg
        self.z_dim, self.c_dim, self.w_dim = cfg.z_dim, cfg.c_dim, cfg.w_dim
        self.img_resolution = img_resolution or cfg.img_resolution
        self.img_channels = img_channels
        self.synthesis = SynthesisNetwork(cfg, img_resolution=self.img_resolution, img_channels=img_channels)
        self.num_ws = self.synthesis.num_ws
        # mapping_kwargs: what the launcher injects (train.py:170-172: camera_cond, camera_cond_drop_p, camera_raw_scalars, mean_camera_params);
        # by default taken from the configuration
        if mapping_kwargs is None and cfg.camera_cond:
            mapping_kwargs = dict(camera_cond=True, camera_cond_drop_p=cfg.camera_cond_drop_p, camera_raw_scalars=cfg.camera_raw_scalars,
                                  mean_camera_params=None if cfg.mean_camera_params is None else torch.tensor(cfg.mean_camera_params, dtype=torch.float32))
        self.mapping = MappingNetwork(z_dim=cfg.z_dim, c_dim=cfg.c_dim, w_dim=cfg.w_dim, num_ws=self.num_ws, num_layers=cfg.map_depth, **(mapping_kwargs or {}))
        self.eval()

    def progressive_update(self, cur_kimg):
        """networks_epigraf.py:285-286."""
        self.synthesis.progressive_update(cur_kimg)

    def load_numpy_state_dict(self, sd, strict=True):
        """Load a {reference key: numpy array} state-dict (weights.random_state_dict or an exported checkpoint)."""
        res = self.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()}, strict=strict)
        self.synthesis.tri_plane_decoder.invalidate_cache()
        return res

    def forward_autograd(self, z, c, camera_params, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        """Generator.forward as a differentiable graph: mapping network (eager tensor ops) -> `synthesis.forward_autograd`."""
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis.forward_autograd(ws, camera_params=camera_params, **synthesis_kwargs)

    @torch.no_grad()
    def forward(self, z, c, camera_params, camera_angles_cond=None, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        ws = self.mapping(z, c, camera_angles=camera_angles_cond, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff,
                          update_emas=update_emas)
        return self.synthesis(ws, camera_params=camera_params, update_emas=update_emas, **synthesis_kwargs)
