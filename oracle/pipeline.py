"""Oracle glue: chains the C primitives in the reference's call order.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Plain numpy, no torch.

cfg is a plain dict:
  z_dim, w_dim, c_dim, map_depth, cbase, cmax, fmaps, use_noise,
  tri_plane_res, feat_dim, mlp_hid, ray_marcher_type ('classical'|'mip'),
  num_ray_steps, ray_start, ray_end, cube_scale, use_inf_depth, last_back,
  white_back, density_bias, img_resolution
sd is a dict name -> np.float32 array using the reference's state-dict names
(SURVEY.md section 8a): synthesis.tri_plane_decoder.b{r}.{const|conv0|conv1|torgb}...,
synthesis.tri_plane_mlp.model.{0,1}.{weight,bias}, mapping.{embed,fc0,fc1}.{weight,bias}, mapping.w_avg.
"""
import numpy as np

import oracle as O


def block_resolutions(cfg):
    """networks_epigraf.py:94-96 with in_resolution=0."""
    return [2 ** i for i in range(2, int(np.log2(cfg['tri_plane_res'])) + 1)]


def channels_dict(cfg):
    """networks_epigraf.py:98."""
    return {r: min(int(cfg['cbase'] * cfg['fmaps']) // r, cfg['cmax']) for r in block_resolutions(cfg)}


def num_ws(cfg):
    """networks_epigraf.py:101-112: one w per conv, plus one torgb w for the last block."""
    n = 0
    for i, _ in enumerate(block_resolutions(cfg)):
        n += 1 if i == 0 else 2
    return n + 1


def camera_angle_embeddings(camera_angles, raw_scalars=False, x_multiplier=64.0):
    """layers.py:132-135 + ScalarEncoder1d / FourierEncoder1d (layers.py:251-340): yaw and pitch wrapped to signed turns, then either raw or
    sin / cos of (2^k / 64 * pi) * (64 * angle), k = 0..5 -- per angle [raw | sin x 6 | cos x 6], the two angles concatenated."""
    a = np.asarray(camera_angles, np.float32)[:, [0, 1]]
    a = (np.sign(a) * (np.mod(np.abs(a), np.float32(2.0 * np.pi)) / np.float32(2.0 * np.pi))).astype(np.float32)
    if raw_scalars:
        return a.reshape(len(a), 2)
    nf = int(np.ceil(np.log2(x_multiplier)))
    coefs = ((np.float32(2.0) ** np.arange(nf, dtype=np.float32)) / np.float32(2 ** nf)).astype(np.float32) * np.float32(np.pi)
    raw = coefs[None, None, :] * (a * np.float32(x_multiplier))[:, :, None]
    return np.concatenate([np.sin(raw), np.cos(raw)], axis=2).astype(np.float32).reshape(len(a), -1)


def mapping_forward(sd, cfg, z, c, truncation_psi=1.0, truncation_cutoff=None, camera_angles=None):
    """MappingNetwork.forward, layers.py:127-174.  With cfg['camera_cond'] (layers.py:84-93,128-138) the camera's yaw / pitch encodings are appended to the
    label; without angles (eval) `mapping.mean_camera_params` stands in."""
    x = None
    if cfg.get('camera_cond'):
        if camera_angles is None:
            camera_angles = np.repeat(np.asarray(sd['mapping.mean_camera_params'], np.float32)[None, :3], len(z), axis=0)
        emb = camera_angle_embeddings(camera_angles, raw_scalars=cfg.get('camera_raw_scalars', False))
        c = emb if (c is None or cfg['c_dim'] == 0) else np.concatenate([np.asarray(c, np.float32), emb], axis=1)
    if cfg['z_dim'] > 0:
        x = O.normalize_2nd_moment(z)
    if cfg['c_dim'] > 0 or cfg.get('camera_cond'):
        y = O.normalize_2nd_moment(O.fc(c, sd['mapping.embed.weight'], sd['mapping.embed.bias']))
        x = np.concatenate([x, y], axis=1) if x is not None else y
    for i in range(cfg['map_depth']):
        x = O.fc(x, sd[f'mapping.fc{i}.weight'], sd[f'mapping.fc{i}.bias'], act='lrelu', lr_multiplier=0.01)
    nws = num_ws(cfg)
    ws = np.repeat(x[:, None, :], nws, axis=1).astype(np.float32)
    if truncation_psi != 1:
        w_avg = sd['mapping.w_avg'].astype(np.float32)
        psi = np.float32(truncation_psi)
        cut = nws if truncation_cutoff is None else truncation_cutoff
        # torch.lerp(start, end, weight): weight < 0.5 ? start + w*(end-start) : end - (end-start)*(1-w)
        seg = ws[:, :cut]
        diff = seg - w_avg
        if psi < 0.5:
            ws[:, :cut] = w_avg + psi * diff
        else:
            ws[:, :cut] = seg - diff * (np.float32(1) - psi)
    return ws


def triplane_decode(feats, weights, biases, marcher='classical'):
    """TriPlaneMLP.forward on the plane-mean features (networks_epigraf.py:46-68) for ANY layer count: FullyConnectedLayer x n (lrelu ... linear; none:
    nn.Identity), the last value is sigma, the others rgb ('mip': sigmoid * 1.002 - 0.001).  feats [B,P,F] -> (rgb [B,P,3], sigma [B,P,1])."""
    B, P, F = feats.shape
    x = np.asarray(feats, np.float32).reshape(B * P, F)
    for i, (w, b) in enumerate(zip(weights, biases)):
        x = O.fc(x, w, b, act='lrelu' if i + 1 < len(weights) else 'linear')
    x = x.reshape(B, P, -1)
    rgb = x[..., :-1]
    if marcher == 'mip':
        rgb = ((1.0 / (1.0 + np.exp(-rgb.astype(np.float64)))).astype(np.float32) * np.float32(1 + 2 * 0.001) - np.float32(0.001)).astype(np.float32)
    return rgb.astype(np.float32), x[..., -1:].astype(np.float32)


def _layer(sd, pfx, x, w, up, noise_mode, f, use_noise, bf16=False, clamp=None):
    """SynthesisLayer.forward, networks_stylegan2.py:128-145.  bf16: the block runs in reduced precision (x holds bf16 values)."""
    styles = O.fc(w, sd[pfx + '.affine.weight'], sd[pfx + '.affine.bias'])
    noise = None
    if use_noise and noise_mode == 'const':
        noise = (sd[pfx + '.noise_const'] * sd[pfx + '.noise_strength']).astype(np.float32)
    elif use_noise and isinstance(noise_mode, dict):        # explicit 'random' noise tensors keyed by layer
        noise = (noise_mode[pfx] * sd[pfx + '.noise_strength']).astype(np.float32)
    x = O.modulated_conv2d(x, sd[pfx + '.weight'], styles, noise=noise, up=up, demodulate=True, resample_filter=f, prec='bf16' if bf16 else 'f32')
    if bf16:
        return O.bias_act_bf16(x, sd[pfx + '.bias'], act='lrelu', clamp=clamp)
    return O.bias_act(x, sd[pfx + '.bias'], act='lrelu', clamp=clamp)


def _torgb(sd, pfx, x, w, bf16=False, clamp=None):
    """ToRGBLayer.forward, networks_stylegan2.py:168-172."""
    cin = sd[pfx + '.weight'].shape[1]
    styles = O.fc(w, sd[pfx + '.affine.weight'], sd[pfx + '.affine.bias'])
    styles = (styles * np.float32(1 / np.sqrt(cin))).astype(np.float32)
    x = O.modulated_conv2d(x, sd[pfx + '.weight'], styles, demodulate=False, prec='bf16' if bf16 else 'f32')
    if bf16:
        return O.bias_act_bf16(x, sd[pfx + '.bias'], clamp=clamp)       # y.to(float32) afterwards: exact
    return O.bias_act(x, sd[pfx + '.bias'], clamp=clamp)


def fp16_resolution(cfg):
    """networks_epigraf.py:99: blocks at or above this resolution run in reduced precision (None: fp32_only, train.py:271-273)."""
    n = int(cfg.get('num_fp16_res', 0) or 0)
    if n <= 0:
        return None
    return max(2 ** (int(np.log2(cfg['tri_plane_res'])) + 1 - n), 8)


def synthesis_backbone(sd, cfg, ws, noise_mode='const', return_intermediates=False):
    """SynthesisBlocksSequence.forward (networks_epigraf.py:114-129) over SynthesisBlock.forward
    (networks_stylegan2.py:231-273), architecture 'skip'.  fp32 (`fp32_only`), or -- cfg['num_fp16_res'] > 0, BASELINE configs[4] --
    the reference's reduced-precision blocks (:237 `dtype = float16 if use_fp16`) with bfloat16, conv_clamp = cfg['conv_clamp']: x is
    rounded at the block input (:250), every layer follows orc_modconv2d's / bias_act_bf16's rounding points, the ToRGB output is
    widened to fp32 (:268) and the skip image stays fp32."""
    f = O.setup_filter([1, 3, 3, 1])
    B = ws.shape[0]
    x = img = None
    w_idx = 0
    inter = {}
    root = 'synthesis.tri_plane_decoder'
    r16 = fp16_resolution(cfg)
    for i, r in enumerate(block_resolutions(cfg)):
        pfx = f'{root}.b{r}'
        bf16 = r16 is not None and r >= r16
        clamp = cfg.get('conv_clamp') if r16 is not None else None
        if i == 0:
            x = np.repeat(sd[pfx + '.const'][None], B, axis=0).astype(np.float32)
            if bf16:
                x = O.round_bf16(x)
            x = _layer(sd, pfx + '.conv1', x, ws[:, w_idx], 1, noise_mode, f, cfg['use_noise'], bf16, clamp)
            nconv = 1
        else:
            if bf16:
                x = O.round_bf16(x)                   # x.to(dtype), :250
            x = _layer(sd, pfx + '.conv0', x, ws[:, w_idx], 2, noise_mode, f, cfg['use_noise'], bf16, clamp)
            x = _layer(sd, pfx + '.conv1', x, ws[:, w_idx + 1], 1, noise_mode, f, cfg['use_noise'], bf16, clamp)
            nconv = 2
        if img is not None:
            img = O.upsample2d(img, f)
        y = _torgb(sd, pfx + '.torgb', x, ws[:, w_idx + nconv], bf16, clamp)
        img = img + y if img is not None else y
        w_idx += nconv
        if return_intermediates:
            inter[f'x{r}'] = x
            inter[f'img{r}'] = img
    return (img, inter) if return_intermediates else img


def importance_render(planes, mlp, ray_o, ray_d, opts, u_coarse, u_fine, return_intermediates=False, n_coarse=None, n_fine=None):
    """ImportanceRenderer.forward, tri_plane_renderer.py:126-170.
    planes [B,3F,H,W]; mlp = (w0,b0,w1,b1); u_coarse [B,R,S]; u_fine [B*R,S].  opts['density_noise'] > 0 (training):
    sigma += n * density_noise after each field evaluation (:185-186), n_coarse [B,R*S,1] / n_fine [B,R*N,1] = the randn draws."""
    dnoise = np.float32(opts.get('density_noise', 0.0))

    def add_noise(sigma, n):
        return sigma if not dnoise > 0 else (sigma + np.asarray(n, np.float32).reshape(sigma.shape) * dnoise).astype(np.float32)

    mode = opts['ray_marcher_type']
    B, R, _ = ray_o.shape
    S = opts['num_proposal_steps']
    scale = opts['box_size'] / 2
    sdist = O.sample_stratified(u_coarse.reshape(B, R, S), mode)
    tdist = O.s_to_t(sdist, opts['ray_start'], opts['ray_end'])
    out = O.triplane_field(planes, O.ray_points(ray_o, ray_d, tdist), *mlp, scale=scale, mlp_mode=mode)
    col_c = out['rgb'].reshape(B, R, S, 3)
    den_c = add_noise(out['sigma'], n_coarse).reshape(B, R, S, 1)

    def march(c, d, z):
        if mode == 'classical':
            return O.march_classical(c, d, z, use_inf_depth=opts['use_inf_depth'], clamp_mode=opts.get('clamp_mode', 'softplus'),
                                     last_back=opts.get('last_back', False), cut_quantile=opts.get('cut_quantile', 0.0))
        return O.march_mip(c, d, z, use_inf_depth=opts['use_inf_depth'], density_bias=opts.get('density_bias', 0.0),
                           white_back=opts.get('white_back', False), cut_quantile=opts.get('cut_quantile', 0.0))

    inter = dict(sdist_coarse=sdist, colors_coarse=col_c, densities_coarse=den_c)
    N = opts['num_fine_steps']
    if N > 0:
        _, _, w_c, _ = march(col_c, den_c, sdist[..., None])              # s-space depths: :152
        sfine = O.sample_importance(sdist[..., None], w_c, u_fine, mode)   # [B,R,N,1]
        tfine = O.s_to_t(sfine[..., 0], opts['ray_start'], opts['ray_end'])
        out = O.triplane_field(planes, O.ray_points(ray_o, ray_d, tfine), *mlp, scale=scale, mlp_mode=mode)
        col_f = out['rgb'].reshape(B, R, N, 3)
        den_f = add_noise(out['sigma'], n_fine).reshape(B, R, N, 1)
        d_all, c_all, s_all = O.unify_samples(tdist[..., None], col_c, den_c, tfine[..., None], col_f, den_f)
        rgb, depth, wts, fT = march(c_all, s_all, d_all)
        inter.update(weights_coarse=w_c, sdist_fine=sfine, colors_fine=col_f, densities_fine=den_f, all_depths=d_all)
    else:
        rgb, depth, wts, fT = march(col_c, den_c, sdist[..., None])
    res = (rgb, depth, wts.sum(axis=2, dtype=np.float64).astype(np.float32), fT)
    return (res, inter) if return_intermediates else res


def importance_render_grad(planes, mlp, ray_o, ray_d, opts, u_coarse, u_fine, d_rgb, d_depth=None):
    """Gradients of importance_render w.r.t. planes and the MLP tensors (autograd through tri_plane_renderer.py:126-170; the
    importance samples are constants, :241): march gradient on the unified samples, un-sort, field gradient of both passes.
    -> (d_planes, d_w0, d_b0, d_w1, d_b1)"""
    mode = opts['ray_marcher_type']
    B, R, _ = ray_o.shape
    S, N = opts['num_proposal_steps'], opts['num_fine_steps']
    scale = opts['box_size'] / 2
    _, inter = importance_render(planes, mlp, ray_o, ray_d, opts, u_coarse, u_fine, return_intermediates=True)
    sdist = inter['sdist_coarse']
    tdist = O.s_to_t(sdist, opts['ray_start'], opts['ray_end'])
    kw = dict(mode=mode, use_inf_depth=opts['use_inf_depth'])
    if mode == 'classical':
        kw.update(clamp_mode=opts.get('clamp_mode', 'softplus'), last_back=opts.get('last_back', False))
    else:
        kw.update(density_bias=opts.get('density_bias', 0.0), white_back=opts.get('white_back', False))
    pts_c = O.ray_points(ray_o, ray_d, tdist)
    if N > 0:
        tfine = O.s_to_t(inter['sdist_fine'][..., 0], opts['ray_start'], opts['ray_end'])
        d_all, c_all, s_all, perm = O.unify_samples(tdist[..., None], inter['colors_coarse'], inter['densities_coarse'], tfine[..., None],
                                                    inter['colors_fine'], inter['densities_fine'], return_perm=True)
        g_c, g_s = O.ray_march_grad(c_all, s_all, d_all, d_rgb, d_depth, **kw)
        u_c, u_s = np.zeros_like(g_c), np.zeros_like(g_s)
        np.put_along_axis(u_c, np.broadcast_to(perm[..., None], g_c.shape), g_c, axis=2)
        np.put_along_axis(u_s, perm[..., None], g_s, axis=2)
        passes = [(pts_c, u_c[:, :, :S], u_s[:, :, :S]), (O.ray_points(ray_o, ray_d, tfine), u_c[:, :, S:], u_s[:, :, S:])]
    else:
        g_c, g_s = O.ray_march_grad(inter['colors_coarse'], inter['densities_coarse'], sdist[..., None], d_rgb, d_depth, **kw)
        passes = [(pts_c, g_c, g_s)]
    total = None
    for pts, gc, gs in passes:
        res = O.triplane_field_grad(planes, pts, *mlp, np.ascontiguousarray(gc).reshape(B, -1, 3), np.ascontiguousarray(gs).reshape(B, -1, 1), scale=scale,
                                    mlp_mode=mode)
        total = list(res) if total is None else [a + b for a, b in zip(total, res)]
    return tuple(total)


def render_options(cfg):
    """networks_epigraf.py:226-231."""
    return dict(box_size=cfg['cube_scale'] * 2, num_proposal_steps=cfg['num_ray_steps'], num_fine_steps=cfg['num_ray_steps'],
                clamp_mode='softplus', use_inf_depth=cfg['use_inf_depth'], ray_start=cfg['ray_start'], ray_end=cfg['ray_end'],
                last_back=cfg.get('last_back', False), white_back=cfg.get('white_back', False),
                density_bias=cfg.get('density_bias', 0.0), ray_marcher_type=cfg['ray_marcher_type'], cut_quantile=cfg.get('cut_quantile', 0.0))


def synthesis_forward(sd, cfg, ws, camera, u_coarse, u_fine, noise_mode='const', return_intermediates=False, training=None):
    """SynthesisNetwork.forward, networks_epigraf.py:210-261 (no adaptors).
    camera: dict angles [B,3], fov [B], radius [B], look_at [B,3].
    training = None (eval) or dict(resolution, patch_scales [B,2], patch_offsets [B,2], density_noise, n_coarse, n_fine):
    the training-mode forward (:220-222) -- rays of a patch at train_resolution, sigma perturbed by the density noise."""
    planes, inter = synthesis_backbone(sd, cfg, ws, noise_mode, return_intermediates=True)
    tr = training or {}
    h = w = tr.get('resolution', cfg['img_resolution'])
    c2w = O.cam2world(camera['angles'], camera['radius'], camera['look_at'])
    ray_o, ray_d = O.sample_rays(c2w, camera['fov'], h, w, tr.get('patch_scales'), tr.get('patch_offsets'))
    mlp = tuple(sd[f'synthesis.tri_plane_mlp.model.{i}.{n}'] for i in (0, 1) for n in ('weight', 'bias'))
    ropts = dict(render_options(cfg), density_noise=tr.get('density_noise', 0.0))
    B = ws.shape[0]
    mbr = cfg.get('max_batch_res', 128)
    if training is None and ropts['cut_quantile'] > 0 and (h > mbr or w > mbr) and 2 ** 24 // (B * cfg['num_ray_steps'] * 3) < h * w:
        # networks_epigraf.py:232-239: above max_batch_res the eval forward renders through run_batchwise over ray chunks of
        # 2**24 // (B * num_ray_steps * 3) rays (training_utils.py:171-203) -- each chunk takes its OWN quantiles
        step, R, S = 2 ** 24 // (B * cfg['num_ray_steps'] * 3), h * w, cfg['num_ray_steps']
        uc, uf = np.asarray(u_coarse).reshape(B, R, S), np.asarray(u_fine).reshape(B, R, S)
        parts = [importance_render(planes, mlp, ray_o[:, a:a + step], ray_d[:, a:a + step], ropts, np.ascontiguousarray(uc[:, a:a + step]),
                                   np.ascontiguousarray(uf[:, a:a + step]).reshape(-1, S)) for a in range(0, R, step)]
        rgb, depth, wsum, fT = (np.concatenate([p[i] for p in parts], axis=1) for i in range(4))
        rinter = {}
    else:
        (rgb, depth, wsum, fT), rinter = importance_render(planes, mlp, ray_o, ray_d, ropts, u_coarse, u_fine, return_intermediates=True,
                                                           n_coarse=tr.get('n_coarse'), n_fine=tr.get('n_fine'))
    img = np.ascontiguousarray(rgb.reshape(B, h, w, 3).transpose(0, 3, 1, 2))
    depth = depth.reshape(B, 1, h, w)
    if return_intermediates:
        inter.update(rinter)
        inter.update(planes=planes, c2w=c2w, ray_o=ray_o, ray_d=ray_d)
        return img, depth, inter
    return img, depth


def generator_forward(sd, cfg, z, c, camera, u_coarse, u_fine, noise_mode='const', truncation_psi=1.0):
    """Generator.forward, networks_epigraf.py:288-291."""
    ws = mapping_forward(sd, cfg, z, c, truncation_psi)
    return synthesis_forward(sd, cfg, ws, camera, u_coarse, u_fine, noise_mode)


# ---------------------------------------------------------------------------------------------------------------------
# depth / camera adaptors (forward, eval)
# ---------------------------------------------------------------------------------------------------------------------
def conv2d_layer(x, weight, bias, act='linear'):
    """Conv2dLayer.forward with up = down = 1 (layers.py:221-236): correlation with padding k // 2, bias_act with the
    activation's default gain.  The plain convolution is the modulated-conv restatement with unit styles, no demodulation."""
    weight = np.asarray(weight, dtype=np.float32)
    cout, cin, k, _ = weight.shape
    w = (weight * np.float32(1 / np.sqrt(cin * k * k))).astype(np.float32)
    y = O.modulated_conv2d(x, w, np.ones((x.shape[0], cin), dtype=np.float32), up=1, demodulate=False)
    return O.bias_act(y, bias, act=act)


def _sigmoid32(x):
    x = np.asarray(x, dtype=np.float32)
    with np.errstate(over='ignore'):
        return (np.float32(1.0) / (np.float32(1.0) + np.exp(-x, dtype=np.float32))).astype(np.float32)


def depth_adaptor_forward(sd, cfg, depth, w, return_all=False):
    """DepthAdaptor.forward in eval mode (networks_depth_adaptor.py:49-99).  depth [B,1,h,w], w [B,w_dim] (only its batch
    size is used, :44).  Returns the adapted depth [B,1,h,w] (and the stacked per-layer heads [B,n,1,h,w])."""
    da, pfx = cfg['depth_adaptor'], 'synthesis.depth_adaptor'
    f32 = np.float32
    B = depth.shape[0]
    raw = np.repeat(np.asarray(sd[pfx + '.near_plane_offset_raw'], dtype=f32), B)
    off = (_sigmoid32(raw) * f32(da['near_plane_offset_max_fraction'])) * f32(cfg['ray_end'] - cfg['ray_start'])     # :46
    near = (f32(cfg['ray_start']) + off).reshape(B, 1, 1, 1)
    mid = f32(0.5) * (f32(cfg['ray_end']) + near)
    rng = f32(cfg['ray_end']) - near
    x = ((np.asarray(depth, dtype=f32) - mid) / (rng + f32(1e-12)) * f32(2.0)).astype(f32)
    outs = [x]
    for i in range(da['num_hid_layers']):
        x = conv2d_layer(x, sd[f'{pfx}.layers.{i}.weight'], sd[f'{pfx}.layers.{i}.bias'], 'lrelu')
        outs.append(conv2d_layer(x, sd[pfx + '.head.weight'], sd[pfx + '.head.bias'], 'linear'))
    stack = np.stack(outs, axis=1)
    if da['out_strategy'] in ('last', 'random'):
        res = stack[:, -1] + f32(0.0) * stack.max()
    elif da['out_strategy'] == 'mean':
        res = stack.mean(axis=1, dtype=f32)
    else:
        raise NotImplementedError(da['out_strategy'])
    return (res, stack) if return_all else res


def _params_adaptor(sd, pfx, lr, x, z=None, c=None):
    """ParamsAdaptor.forward (networks_camera_adaptor.py:44-52)."""
    fc = lambda name, v, act: O.fc(v, sd[f'{pfx}.{name}.weight'], sd[f'{pfx}.{name}.bias'], act=act, lr_multiplier=lr)     # noqa: E731
    x = fc('project_params', x, 'softplus')
    if z is not None and f'{pfx}.project_z.weight' in sd:
        x = np.concatenate([x, O.normalize_2nd_moment(fc('project_z', z, 'softplus'))], axis=1)
    if c is not None and f'{pfx}.project_c.weight' in sd:
        x = np.concatenate([x, O.normalize_2nd_moment(fc('project_c', c, 'softplus'))], axis=1)
    return fc('main.1', fc('main.0', x, 'softplus'), 'linear')


def camera_adaptor_forward(sd, cfg, camera, z, c=None):
    """CameraAdaptor.forward (networks_camera_adaptor.py:74-134): prior camera parameters -> posterior."""
    ca, pfx = cfg['camera_adaptor'], 'synthesis.camera_adaptor'
    cam, f32, eps = ca['camera'], np.float32, np.float32(1e-8)
    col = lambda a: np.asarray(a, dtype=f32).reshape(len(a), -1)        # noqa: E731
    yaw, pitch, roll = (col(camera['angles'])[:, [i]] for i in range(3))
    fov, radius = col(camera['fov']), col(camera['radius'])
    la_yaw, la_pitch, la_radius = (col(camera['look_at'])[:, [i]] for i in range(3))
    nrm = lambda v, r: (v - f32(r[0])) / (f32(r[1] - r[0]) + eps)       # noqa: E731
    n_yaw, n_pitch, n_fov = nrm(yaw, cam['yaw']), nrm(pitch, cam['pitch']), nrm(fov, cam['fov'])
    n_la = [nrm(la_yaw, cam['look_at_yaw']), nrm(la_pitch, cam['look_at_pitch']), nrm(la_radius, cam['look_at_radius'])]
    lr = ca['lr_multiplier']
    origin_new = _params_adaptor(sd, pfx + '.origin_adaptor', lr, np.concatenate([n_yaw, n_pitch, roll, radius], axis=1), c=c)
    la_in = np.concatenate([origin_new[:, :3], n_fov, origin_new[:, [3]]] + n_la, axis=1)
    la_new = _params_adaptor(sd, pfx + '.look_at_adaptor', lr, la_in, z=z, c=c)
    new = [origin_new[:, [0]], origin_new[:, [1]], origin_new[:, [2]], la_new[:, [0]], origin_new[:, [3]], la_new[:, [1]], la_new[:, [2]], la_new[:, [3]]]
    if ca['residual']:
        old = [n_yaw, n_pitch, roll, n_fov, radius] + n_la
        new = [o + n for o, n in zip(old, new)]
    rng = lambda r: f32(r[1] - r[0])                                     # noqa: E731
    d_yaw = _sigmoid32(new[0]) * rng(cam['yaw']) + f32(cam['yaw'][0])
    d_pitch = _sigmoid32(new[1]) * f32(cam['pitch'][1] - cam['pitch'][0] - 2e-5) + f32(cam['pitch'][0]) + f32(1e-5)
    d_roll = new[2] * f32(0.0)
    d_fov = _sigmoid32(new[3]) * rng(cam['fov']) + f32(cam['fov'][0])
    d_radius = new[4]
    d_la_yaw = _sigmoid32(new[5]) * rng(cam['look_at_yaw']) + f32(cam['look_at_yaw'][0])
    d_la_pitch = _sigmoid32(new[6]) * rng(cam['look_at_pitch']) + f32(cam['look_at_pitch'][0])
    d_la_radius = _sigmoid32(new[7]) * f32(cam['look_at_radius'][1] - cam['look_at_pitch'][0]) + f32(cam['look_at_pitch'][0])   # sic (:95)
    out = dict(angles=np.concatenate([d_yaw, d_pitch, d_roll], axis=1), fov=d_fov[:, 0], radius=d_radius[:, 0],
               look_at=np.concatenate([d_la_yaw, d_la_pitch, d_la_radius], axis=1))
    if not ca['adjust_angles']:
        out['angles'] = col(camera['angles']) + f32(0.0) * out['angles']
    if not ca['adjust_radius']:
        out['radius'] = col(camera['radius'])[:, 0] + f32(0.0) * out['radius']
    if not ca['adjust_fov']:
        out['fov'] = col(camera['fov'])[:, 0] + f32(0.0) * out['fov']
    if not ca['adjust_look_at']:
        out['look_at'] = col(camera['look_at']) + f32(0.0) * out['look_at']
    return {k: v.astype(f32) for k, v in out.items()}
