"""State-dict layout of the generator and deterministic synthetic weights.

Key names and shapes are the reference's (probed in SURVEY.md section 8a):
  synthesis.tri_plane_decoder.b{r}.{const | conv0|conv1|torgb}.{weight,bias,affine.weight,affine.bias,
      noise_const,noise_strength,resample_filter}
  synthesis.tri_plane_mlp.model.{0..n-1}.{weight,bias}        (n = tri_plane.mlp.n_layers; none for n = 0)
  mapping.{embed,fc0,fc1}.{weight,bias}, mapping.w_avg
so a state-dict exported from a reference checkpoint loads unchanged.

There is no network access for checkpoints: `random_state_dict` draws every tensor from a
numpy RandomState keyed by (seed, crc32(name)) with the reference's init distributions
(networks_stylegan2.py:120-126,162-165; layers.py:36-38).  `exercise_all=True` additionally
randomises biases / noise strengths / w_avg (zero at reference init) so every term of the
forward pass is numerically live in parity tests.
"""
import zlib
from collections import OrderedDict

import numpy as np

from .config import GeneratorConfig


def state_dict_spec(cfg: GeneratorConfig):
    """Ordered name -> shape map of every tensor the generator forward reads."""
    spec = OrderedDict()
    ch = cfg.channels
    root = 'synthesis.tri_plane_decoder'
    img_c = cfg.plane_channels

    def layer(pfx, cin, cout, res, k, noise, synth=True):
        spec[pfx + '.weight'] = (cout, cin, k, k)
        if noise:
            spec[pfx + '.noise_strength'] = ()
        spec[pfx + '.bias'] = (cout,)
        if synth:                                          # SynthesisLayer registers the filter with or without noise; ToRGBLayer has none
            spec[pfx + '.resample_filter'] = (4, 4)
        if noise:
            spec[pfx + '.noise_const'] = (res, res)
        spec[pfx + '.affine.weight'] = (cin, cfg.w_dim)
        spec[pfx + '.affine.bias'] = (cin,)

    for i, r in enumerate(cfg.block_resolutions):
        pfx = f'{root}.b{r}'
        cout = ch[r]
        spec[pfx + '.resample_filter'] = (4, 4)
        if i == 0:
            spec[pfx + '.const'] = (cout, r, r)
        else:
            layer(pfx + '.conv0', ch[r // 2], cout, r, 3, cfg.use_noise)
        layer(pfx + '.conv1', cout, cout, r, 3, cfg.use_noise)
        layer(pfx + '.torgb', cout, img_c, r, 1, False, synth=False)
    if cfg.mlp_n_layers > 0:               # networks_epigraf.py:35-43 (0 layers: nn.Identity, no parameters)
        dims = [cfg.feat_dim] + [cfg.mlp_hid] * (cfg.mlp_n_layers - 1) + [1 + (cfg.mlp_hid if cfg.has_view_cond else 3)]
        for i in range(len(dims) - 1):
            spec[f'synthesis.tri_plane_mlp.model.{i}.weight'] = (dims[i + 1], dims[i])
            spec[f'synthesis.tri_plane_mlp.model.{i}.bias'] = (dims[i + 1],)
    spec['mapping.w_avg'] = (cfg.w_dim,)
    c_dim = cfg.c_dim
    if cfg.camera_cond:                    # layers.py:84-93: yaw / pitch encodings appended to the label
        enc = 2 * (1 if cfg.camera_raw_scalars else 2 * 6)          # raw: 1 value per angle; Fourier: sin + cos of ceil(log2(64)) = 6 frequencies
        c_dim += enc
        if not cfg.camera_raw_scalars:
            spec['mapping.camera_scalar_enc.fourier_encoder.fourier_coefs'] = (6,)
        if cfg.mean_camera_params is not None:
            spec['mapping.mean_camera_params'] = (len(cfg.mean_camera_params),)
    if c_dim > 0:
        spec['mapping.embed.weight'] = (cfg.w_dim, c_dim)
        spec['mapping.embed.bias'] = (cfg.w_dim,)
    feats = [cfg.z_dim + (cfg.w_dim if c_dim > 0 else 0)] + [cfg.w_dim] * cfg.map_depth
    for i in range(cfg.map_depth):
        spec[f'mapping.fc{i}.weight'] = (feats[i + 1], feats[i])
        spec[f'mapping.fc{i}.bias'] = (feats[i + 1],)
    da = cfg.depth_adaptor
    if da is not None:                     # networks_depth_adaptor.py:28-40
        pfx = 'synthesis.depth_adaptor'
        dims = [1] + [da.hid_dim] * da.num_hid_layers
        for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
            spec[f'{pfx}.layers.{i}.weight'] = (cout, cin, da.kernel_size, da.kernel_size)
            spec[f'{pfx}.layers.{i}.bias'] = (cout,)
            spec[f'{pfx}.layers.{i}.resample_filter'] = (4, 4)
        if da.num_hid_layers > 0:
            spec[f'{pfx}.head.weight'] = (1, dims[-1], 1, 1)
            spec[f'{pfx}.head.bias'] = (1,)
            spec[f'{pfx}.head.resample_filter'] = (4, 4)
        spec[f'{pfx}.progress_coef'] = (1,)
        spec[f'{pfx}.near_plane_offset_raw'] = (1,)
    ca = cfg.camera_adaptor
    if ca is not None:                     # networks_camera_adaptor.py:24-64
        for name, nin, use_z in (('origin_adaptor', 4, False), ('look_at_adaptor', 8, True)):
            pfx = f'synthesis.camera_adaptor.{name}'
            spec[f'{pfx}.project_params.weight'] = (ca.hid_dim, nin)
            spec[f'{pfx}.project_params.bias'] = (ca.hid_dim,)
            main_in = ca.hid_dim
            if use_z:
                spec[f'{pfx}.project_z.weight'] = (ca.embed_dim, cfg.z_dim)
                spec[f'{pfx}.project_z.bias'] = (ca.embed_dim,)
                main_in += ca.embed_dim
            if cfg.c_dim > 0:
                spec[f'{pfx}.project_c.weight'] = (ca.embed_dim, cfg.c_dim)
                spec[f'{pfx}.project_c.bias'] = (ca.embed_dim,)
                main_in += ca.embed_dim
            spec[f'{pfx}.main.0.weight'] = (ca.hid_dim, main_in)
            spec[f'{pfx}.main.0.bias'] = (ca.hid_dim,)
            spec[f'{pfx}.main.1.weight'] = (4, ca.hid_dim)
            spec[f'{pfx}.main.1.bias'] = (4,)
    return spec


def _rng(seed, name):
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def resample_filter():
    """upfirdn2d.setup_filter([1,3,3,1]) (upfirdn2d.py:70-114): outer product / 64."""
    f = np.array([1, 3, 3, 1], dtype=np.float32)
    f = np.outer(f, f).astype(np.float32)
    return (f / f.sum(dtype=np.float32)).astype(np.float32)


def random_state_dict(cfg: GeneratorConfig, seed=0, exercise_all=False):
    sd = OrderedDict()
    for name, shape in state_dict_spec(cfg).items():
        g = _rng(seed, name)
        leaf = name.rsplit('.', 1)[-1]
        if leaf == 'resample_filter':
            v = resample_filter()
        elif leaf == 'progress_coef':
            v = np.zeros(shape)
        elif leaf == 'fourier_coefs':                        # construct_log_spaced_freqs(64): 2^k / 64 * pi, k = 0..5 (layers.py:346-358)
            v = (2.0 ** np.arange(shape[0]) / 2.0 ** shape[0]).astype(np.float32) * np.float32(np.pi)
        elif leaf == 'mean_camera_params':
            v = np.asarray(cfg.mean_camera_params, np.float32)
        elif leaf == 'near_plane_offset_raw':
            v = np.full(shape, cfg.depth_adaptor.near_plane_offset_bias + (0.5 * g.randn() if exercise_all else 0.0))
        elif name.startswith('synthesis.camera_adaptor') and leaf == 'weight':
            v = g.randn(*shape) / cfg.camera_adaptor.lr_multiplier      # weight_init / lr_multiplier, layers.py:36
        elif name.startswith('synthesis.camera_adaptor') and leaf == 'bias':
            v = (0.1 * g.randn(*shape) / cfg.camera_adaptor.lr_multiplier) if exercise_all else np.zeros(shape)
        elif name.startswith('mapping.fc') and leaf == 'weight':
            v = g.randn(*shape) / 0.01                       # weight_init / lr_multiplier, layers.py:36
        elif leaf in ('weight', 'const', 'noise_const'):
            v = g.randn(*shape)
        elif leaf == 'noise_strength':
            v = np.asarray(0.1 * g.randn() if exercise_all else 0.0)
        elif name.endswith('affine.bias'):
            v = 1.0 + (0.1 * g.randn(*shape) if exercise_all else 0.0)
        elif name.startswith('mapping.fc') and leaf == 'bias':
            v = (0.1 * g.randn(*shape) / 0.01) if exercise_all else np.zeros(shape)
        elif leaf in ('bias', 'w_avg'):
            v = 0.1 * g.randn(*shape) if exercise_all else np.zeros(shape)
        else:
            raise KeyError(name)
        sd[name] = np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.float32), shape)).astype(np.float32)
    return sd


def synthetic_inputs(cfg: GeneratorConfig, batch, seed=0):
    """Seeded synthetic (z, c, camera, u_coarse, u_fine) as in SURVEY.md section 8d.

    RNG tensors are explicit inputs: the reference's eval forward with noise_mode='const' draws exactly
    `rand_like [B,R,S,1]` (tri_plane_renderer.py:225) then `rand [B*R,S]` (:279); parity is defined on
    identical values of those tensors, not on identical generator state.
    """
    g = np.random.RandomState(seed + 1000003)
    z = g.randn(batch, cfg.z_dim).astype(np.float32)
    c = np.zeros((batch, cfg.c_dim), dtype=np.float32)
    if cfg.c_dim > 0:
        c[np.arange(batch), (seed + np.arange(batch)) % cfg.c_dim] = 1.0
    yaw = g.uniform(-1.57, 1.57, batch)                       # configs/camera/uniform.yaml:7
    pitch = g.uniform(np.pi / 4, 3 * np.pi / 4, batch)        # :8
    angles = np.stack([yaw, pitch, np.zeros(batch)], 1).astype(np.float32)
    fov = g.uniform(10.0, 45.0, batch).astype(np.float32)     # configs/camera/base.yaml:4
    radius = np.ones(batch, dtype=np.float32)
    la = np.stack([g.uniform(0, 2 * np.pi, batch), g.uniform(0.1, np.pi - 0.1, batch), g.uniform(0, 0.2, batch)], 1)
    R = cfg.img_resolution ** 2
    S = cfg.num_ray_steps
    u_coarse = g.rand(batch, R, S).astype(np.float32)
    u_fine = g.rand(batch * R, S).astype(np.float32)
    camera = dict(angles=angles, fov=fov, radius=radius, look_at=la.astype(np.float32))
    return dict(z=z, c=c, camera=camera, u_coarse=u_coarse, u_fine=u_fine)


def config_from_json(d):
    """The dict written by tools/export_reference_checkpoint.py (or GeneratorConfig.to_dict()) -> GeneratorConfig."""
    from .config import CameraAdaptorConfig, CameraRanges, DepthAdaptorConfig
    d = dict(d)
    chk = d.pop('checked_options', None)          # written by the exporter: options that change the forward and are not implemented here
    if chk:
        bad = [k for k in ('use_full_box', 'ray_start_is_auto') if chk.get(k)]
        if not chk.get('fp32_only', True) and chk.get('num_fp16_res', 0) > 0:
            bad.append('num_fp16_res')
        if bad:
            raise NotImplementedError(f'exported generator uses options this package does not implement: {bad}')
    da, ca = d.pop('depth_adaptor', None), d.pop('camera_adaptor', None)
    cfg = GeneratorConfig(**d)
    if da is not None:
        cfg.depth_adaptor = DepthAdaptorConfig(**da)
    if ca is not None:
        ca = dict(ca)
        cam = ca.pop('camera', None)
        cfg.camera_adaptor = CameraAdaptorConfig(**ca)
        if cam is not None:
            cfg.camera_adaptor.camera = CameraRanges(**{k: tuple(v) for k, v in cam.items()})
    return cfg


def load_exported(path):
    """(GeneratorConfig, state dict) from the directory written by tools/export_reference_checkpoint.py.  Tensors the forward
    does not read (optimizer state is never there; `*.w_avg`-like buffers are) are kept; `Generator.load_numpy_state_dict` is
    strict about the keys of `state_dict_spec(cfg)` only."""
    import json
    import os
    cfg = config_from_json(json.load(open(os.path.join(path, 'generator.json'))))
    with np.load(os.path.join(path, 'generator.npz')) as z:
        sd = OrderedDict((k, z[k]) for k in z.files)
    spec = state_dict_spec(cfg)
    missing = [k for k in spec if k not in sd]
    if missing:
        raise KeyError(f'exported checkpoint lacks {len(missing)} tensors the generator forward reads, e.g. {missing[:3]}')
    return cfg, OrderedDict((k, sd[k]) for k in spec)
