"""The CPU oracle against golden vectors captured from the reference (tools/gen_goldens.py).

Floating point: tolerance stated per test (reductions are order-dependent in torch).
Integer rows (searchsorted indices, sort permutation): bit-exact on identical inputs.
"""
import os

import numpy as np
import pytest

from conftest import assert_close, assert_close_up_to_threshold_flips, assert_image_parity, load_golden

ACTS = ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish']


@pytest.mark.parametrize('act', ACTS)
def test_bias_act(oracle, act):
    g = load_golden('bias_act')
    y = oracle.bias_act(g['x'], g['b'], act=act)
    assert_close(y, g[f'y_{act}'], 2e-6, act)


def test_bias_act_variants(oracle):
    g = load_golden('bias_act')
    assert_close(oracle.bias_act(g['x'], g['b'], act='lrelu', gain=1.0, clamp=0.5), g['y_lrelu_gain1_clamp'], 1e-6)
    assert_close(oracle.bias_act(g['x'], g['b'], act='lrelu', alpha=0.01, gain=2.0), g['y_lrelu_alpha'], 1e-6)
    assert_close(oracle.bias_act(g['x'], None, act='linear', gain=0.5), g['y_linear_nobias'], 1e-7)
    assert_close(oracle.bias_act(g['x'], g['b_dim3'], dim=3, act='relu'), g['y_relu_dim3'], 1e-6)
    assert_close(oracle.bias_act(g['x2'], g['b'], act='lrelu'), g['y2_lrelu'], 1e-6)
    assert_close(oracle.bias_act(g['x'], g['b'], act='swish', clamp=2.0), g['y_swish_channels_last'], 2e-6)


def test_setup_filter(oracle):
    g = load_golden('upfirdn2d')
    np.testing.assert_array_equal(oracle.setup_filter([1, 3, 3, 1]), g['f1331'])
    assert_close(oracle.setup_filter([1, 2, 3, 4], flip_filter=True, gain=2.0), g['f1331_flip_gain'], 1e-7)


def test_upfirdn2d(oracle):
    g = load_golden('upfirdn2d')
    f = g['f1331']
    tol = 2e-6
    assert_close(oracle.upfirdn2d(g['x_f1'], f, padding=[1, 1, 1, 1], gain=4), g['y_f1'], tol, 'F1', 1.0)
    assert_close(oracle.upsample2d(g['x_f2'], f), g['y_f2'], tol, 'F2', 1.0)
    assert_close(oracle.upfirdn2d(g['x_odd'], f, padding=[2, 1, 2, 1]), g['y_filter2d'], tol, 'filter2d', 1.0)
    assert_close(oracle.upfirdn2d(g['x_odd'], f, down=2, padding=[1, 1, 1, 1]), g['y_downsample2d'], tol, 'downsample2d', 1.0)
    assert_close(oracle.upfirdn2d(g['x_odd'], f, padding=[-1, 2, 3, -1]), g['y_negpad'], tol, 'negpad', 1.0)
    assert_close(oracle.upfirdn2d(g['x_odd'], g['f_asym'], padding=2), g['y_asym_noflip'], tol, 'asym', 1.0)
    assert_close(oracle.upfirdn2d(g['x_odd'], g['f_asym'], padding=2, flip_filter=True), g['y_asym_flip'], tol, 'asym flip', 1.0)
    assert_close(oracle.upfirdn2d(g['x_odd'], g['f_rect'], up=[3, 2], down=[2, 1], padding=[2, 1, 0, 3], gain=1.5),
                 g['y_rect_up3_down2'], tol, 'rect', 1.0)
    assert_close(oracle.upfirdn2d(g['x_odd'], None), g['y_identity'], 0, 'identity')
    assert_close(oracle.upfirdn2d(g['x_f1_33'], f, padding=[1, 1, 1, 1], gain=4), g['y_f1_33'], tol, 'F1 33', 1.0)


@pytest.mark.parametrize('tag', ['c3_up1', 'c3_up2', 'c3_up2_b1', 'c1_rgb', 'c3_up1_b1_nonoise'])
def test_modconv(oracle, tag):
    g = load_golden('modconv')
    k, up, demod = g[f'{tag}_meta']
    y = oracle.modulated_conv2d(g[f'{tag}_x'], g[f'{tag}_w'], g[f'{tag}_s'], noise=g.get(f'{tag}_noise'), up=int(up),
                                demodulate=bool(demod), resample_filter=g['f'])
    assert_close(y, g[f'{tag}_y'], 5e-6, tag, 1.0)


@pytest.mark.parametrize('marcher', ['classical', 'mip'])
def test_field(oracle, marcher):
    g = load_golden('field')
    out = oracle.triplane_field(g['planes'], g['coords'], g[f'{marcher}_w0'], g[f'{marcher}_b0'], g[f'{marcher}_w1'],
                                g[f'{marcher}_b1'], scale=0.5, mlp_mode=marcher, return_feats=True)
    assert_close(out['feats'], g['feats_mean'], 2e-6, 'bilinear+mean', 1.0)
    assert_close(out['rgb'], g[f'{marcher}_rgb'], 5e-6, 'rgb', 1.0)
    assert_close(out['sigma'], g[f'{marcher}_sigma'], 5e-6, 'sigma', 1.0)


@pytest.mark.parametrize('marcher', ['classical', 'mip'])
def test_stratified_and_importance(oracle, marcher):
    g = load_golden('sampling')
    sd = oracle.sample_stratified(g[f'{marcher}_u_coarse'][..., 0], marcher)
    np.testing.assert_array_equal(sd, g[f'{marcher}_sdist'][..., 0])          # pure fp32 elementwise: bit-exact
    sf, aux = oracle.sample_importance(g[f'{marcher}_sdist'], g[f'{marcher}_weights'], g[f'{marcher}_u_fine'], marcher,
                                       return_aux=True)
    # INT rows bit-exact vs the REFERENCE: the pdf normaliser follows torch's CPU sum order (orc_torch_sum_f32), the cdf is torch's
    # sequential-double cumsum, so every searchsorted decision -- and the fine samples themselves -- reproduce exactly
    np.testing.assert_array_equal(aux['inds'], g[f'{marcher}_inds'])
    np.testing.assert_array_equal(aux['below'], g[f'{marcher}_below'])
    np.testing.assert_array_equal(aux['above'], g[f'{marcher}_above'])
    np.testing.assert_array_equal(sf, g[f'{marcher}_sdist_fine'])


@pytest.mark.parametrize('marcher', ['classical', 'mip'])
@pytest.mark.parametrize('S', [32, 48, 64, 96])
def test_importance_hot_sizes(oracle, marcher, S):
    """sample_importance at the ray-step counts of BASELINE configs[0..4]: pdf rows of 30 / 46 / 62 / 94 elements walk every
    branch of torch's sum order (interleaved vector accumulators, left-over vectors, scalar tail).  0 integer mismatches."""
    g = load_golden('sampling_hot')
    tag = f'{marcher}{S}'
    sf, aux = oracle.sample_importance(g[f'{tag}_sdist'], g[f'{tag}_weights'], g[f'{tag}_u_fine'], marcher, return_aux=True)
    np.testing.assert_array_equal(aux['inds'], g[f'{tag}_inds'].astype(np.int64))
    np.testing.assert_array_equal(sf, g[f'{tag}_sdist_fine'])


def test_importance_stage_of_e2e(oracle):
    """The importance-sampling stage exactly as the reference's forward called it (captured arguments of sample_importance,
    tri_plane_renderer.py:153): indices and fine samples bit-exact."""
    g = load_golden('e2e_tiny')
    sf, aux = oracle.sample_importance(g['imp_sdist'], g['imp_weights'], g['u_fine'], 'classical', return_aux=True)
    np.testing.assert_array_equal(aux['inds'], g['inds'])
    np.testing.assert_array_equal(sf, g['imp_sdist_fine'])


def test_torch_sum_order(oracle):
    """orc_torch_sum_f32 against torch.sum itself (the torch of this image is test infrastructure too): every row length that
    changes the kernel's path, incl. the 512-element cascade level."""
    torch = pytest.importorskip('torch')
    if torch.backends.cpu.get_cpu_capability() not in ('AVX2', 'AVX512'):
        pytest.skip('the goldens pin the 8-lane (AVX2-dispatch) order of ATen\'s sum kernel')
    rs = np.random.RandomState(3)
    for n in (1, 3, 4, 7, 8, 9, 15, 30, 46, 62, 94, 126, 254, 511, 512, 600, 2100):
        x = (rs.rand(50, n).astype(np.float32) ** 3 + np.float32(1e-5)).astype(np.float32)
        np.testing.assert_array_equal(oracle.torch_sum(x), torch.sum(torch.from_numpy(x), -1).numpy())


def test_unify(oracle):
    g = load_golden('sampling')
    d, c, s, perm = oracle.unify_samples(g['un_d1'], g['un_c1'], g['un_s1'], g['un_d2'], g['un_c2'], g['un_s2'], return_perm=True)
    np.testing.assert_array_equal(perm, g['un_perm'])
    np.testing.assert_array_equal(d, g['un_d'])
    np.testing.assert_array_equal(c, g['un_c'])
    np.testing.assert_array_equal(s, g['un_s'])


@pytest.mark.parametrize('tag,kw', [('cl_inf', dict(use_inf_depth=True)), ('cl_noinf', dict(use_inf_depth=False)),
                                    ('cl_lastback', dict(use_inf_depth=True, last_back=True)),
                                    ('cl_relu', dict(use_inf_depth=True, clamp_mode='relu')),
                                    ('cl_cut', dict(use_inf_depth=True, cut_quantile=0.5))])
def test_march_classical(oracle, tag, kw):
    g = load_golden('marchers')
    rgb, dep, w, fT = oracle.march_classical(g['colors'], g['densities'], g['depths'], **kw)
    assert_close(w, g[f'{tag}_weights'], 1e-6, 'weights', 1.0)
    assert_close(rgb, g[f'{tag}_rgb'], 5e-6, 'rgb', 1.0)
    assert_close(dep, g[f'{tag}_depth'], 5e-6, 'depth', 1.0)
    assert_close(fT, g[f'{tag}_T'], 2e-6, 'T')


@pytest.mark.parametrize('tag,kw', [('mip_inf', dict(use_inf_depth=True)), ('mip_noinf_white', dict(use_inf_depth=False, white_back=True)),
                                    ('mip_bias', dict(use_inf_depth=True, density_bias=-1.0)),
                                    ('mip_cut', dict(use_inf_depth=True, cut_quantile=0.3))])
def test_march_mip(oracle, tag, kw):
    g = load_golden('marchers')
    rgb, dep, w, fT = oracle.march_mip(g['colors01'], g['densities'], g['depths'], **kw)
    assert_close(w, g[f'{tag}_weights'], 1e-6, 'weights', 1.0)
    assert_close(rgb, g[f'{tag}_rgb'], 5e-6, 'rgb', 1.0)
    assert_close(dep, g[f'{tag}_depth'], 5e-6, 'depth', 1.0)
    assert_close(fT, g[f'{tag}_T'], 2e-6, 'T')


def test_camera_and_rays(oracle):
    g = load_golden('camera')
    c2w = oracle.cam2world(g['angles'], g['radius'], g['look_at'])
    assert_close(c2w, g['c2w'], 2e-6, 'c2w')
    for hw in [(8, 8), (5, 7), (16, 16)]:
        # the reference unpacks `w, h = resolution` (tri_plane_renderer.py:496): golden key AxB <=> w=A, h=B
        o, d = oracle.sample_rays(g['c2w'], g['fov'], hw[1], hw[0])
        assert_close(o, g['ray_o_%dx%d' % hw], 1e-7, 'ray_o')
        assert_close(d, g['ray_d_%dx%d' % hw], 2e-6, 'ray_d', 1.0)
    o, d = oracle.sample_rays(g['c2w'], g['fov'], 6, 6, g['patch_scales'], g['patch_offsets'])
    assert_close(d, g['ray_d_patch'], 2e-6, 'ray_d patch', 1.0)
    o, d = oracle.sample_rays(g['c2w'], 18.0, 4, 4)
    assert_close(d, g['ray_d_scalar_fov'], 2e-6, 'ray_d scalar fov', 1.0)


def test_camera_rays_autograd_reproduces_the_rays(tdgp):
    """renderer.camera_rays_autograd (pure tensor ops, runs on the CPU): the differentiable restatement of cam2world + sample_rays gives
    the reference's rays (goldens from rendering_utils.py:194-218 / tri_plane_renderer.py:487-527), incl. patch rays and a scalar fov."""
    import torch
    g = load_golden('camera')
    cam = {k: torch.from_numpy(g[k]) for k in ('angles', 'radius', 'look_at', 'fov')}
    R = tdgp.renderer
    for hw in [(8, 8), (5, 7), (16, 16)]:
        o, d = R.camera_rays_autograd(cam, hw)                 # `w, h = resolution` as the reference unpacks it
        assert_close(o.numpy(), g['ray_o_%dx%d' % hw], 2e-6, 'ray_o')
        assert_close(d.numpy(), g['ray_d_%dx%d' % hw], 2e-6, 'ray_d', 1.0)
    pp = dict(scales=torch.from_numpy(g['patch_scales']), offsets=torch.from_numpy(g['patch_offsets']))
    o, d = R.camera_rays_autograd(cam, (6, 6), patch_params=pp)
    assert_close(d.numpy(), g['ray_d_patch'], 2e-6, 'ray_d patch', 1.0)
    o, d = R.camera_rays_autograd(dict(cam, fov=18.0), (4, 4))
    assert_close(d.numpy(), g['ray_d_scalar_fov'], 2e-6, 'ray_d scalar fov', 1.0)
    # ... and carries gradients to every camera parameter
    camg = {k: v.clone().requires_grad_(True) for k, v in cam.items()}
    o, d = R.camera_rays_autograd(camg, (5, 7))
    grads = torch.autograd.grad((o * 0.3).sum() + (d * torch.linspace(-1, 1, d.numel()).reshape(d.shape)).sum(), list(camg.values()))
    assert all(torch.isfinite(x).all() and float(x.abs().sum()) > 0 for x in grads)


def test_field_grad_wrt_coords_oracle(oracle):
    """oracle.triplane_field_grad(return_coords=True) against autograd through the reference's simple_tri_plane_renderer."""
    g = load_golden('field_grad')
    for tag in ('small', 'hot'):
        for marcher in ('classical', 'mip'):
            k = f'{tag}_{marcher}_'
            r = oracle.triplane_field_grad(g[f'{tag}_planes'], g[f'{tag}_coords'], g[k + 'w0'], g[k + 'b0'], g[k + 'w1'], g[k + 'b1'], g[f'{tag}_d_rgb'],
                                           g[f'{tag}_d_sigma'], scale=0.5, mlp_mode=marcher, return_coords=True)
            assert_close(r[5], g[k + 'd_coords'], 2e-6, 'd_coords', 1.0)


def test_mapping(oracle, tdgp):
    g = load_golden('mapping')
    for tag, cfg in [('c0', tdgp.config.config_tiny()), ('c10', tdgp.config.config_mid())]:
        sd = tdgp.weights.random_state_dict(cfg, seed=11, exercise_all=True)
        ws = oracle.mapping_forward(sd, cfg.to_dict(), g[f'{tag}_z'], g[f'{tag}_c'])
        assert ws.shape == (3, cfg.num_ws, cfg.w_dim)
        assert_close(ws, g[f'{tag}_ws'], 1e-5, 'ws', 1.0)
        assert_close(oracle.mapping_forward(sd, cfg.to_dict(), g[f'{tag}_z'], g[f'{tag}_c'], 0.7), g[f'{tag}_ws_psi07'], 1e-5, 'psi', 1.0)
        assert_close(oracle.mapping_forward(sd, cfg.to_dict(), g[f'{tag}_z'], g[f'{tag}_c'], 0.3, 3), g[f'{tag}_ws_psi03_cut3'], 1e-5, 'psi cut', 1.0)


def _e2e(oracle, tdgp, tag, cfg, seed):
    g = load_golden(tag)
    sd = tdgp.weights.random_state_dict(cfg, seed=seed, exercise_all=True)
    cam = {k[4:]: v for k, v in g.items() if k.startswith('cam_')}
    ws = oracle.mapping_forward(sd, cfg.to_dict(), g['z'], g['c'])
    assert_close(ws, g['ws'], 1e-5, 'ws', 1.0)
    img, depth, inter = oracle.synthesis_forward(sd, cfg.to_dict(), g['ws'], cam, g['u_coarse'], g['u_fine'], 'const',
                                                 return_intermediates=True)
    return g, img, depth, inter


def test_e2e_tiny(oracle, tdgp):
    cfg = tdgp.config.config_tiny()
    g, img, depth, inter = _e2e(oracle, tdgp, 'e2e_tiny', cfg, 21)
    for r in cfg.block_resolutions:
        assert_close(inter[f'x{r}'], g[f'x{r}'], 5e-6, f'x{r}', 1.0)
    assert_close(inter['planes'], g['planes'], 5e-6, 'planes', 1.0)
    assert_close(inter['c2w'], g['c2w'], 2e-6, 'c2w', 1.0)
    assert_close(inter['ray_d'], g['ray_d'], 2e-6, 'ray_d', 1.0)
    # the stated north-star tolerance: <= 1e-4 max-rel RGB vs the reference CPU path
    assert_image_parity(img, g, 'oracle e2e_tiny img')
    assert_image_parity(depth, g, 'oracle e2e_tiny depth', 'depth')


def test_e2e_mid(oracle, tdgp):
    g, img, depth, _ = _e2e(oracle, tdgp, 'e2e_mid', tdgp.config.config_mid(), 31)
    assert_image_parity(img, g, 'oracle e2e_mid img')
    assert_image_parity(depth, g, 'oracle e2e_mid depth', 'depth')


@pytest.mark.parametrize('tag', ['c1', 'c2', 'c3', 'c4', 'c2mip'])
def test_e2e_full_size(oracle, tdgp, tag):
    """BASELINE configs[0..3] at their REAL size (512^2 tri-planes, 512-channel backbone -- 1024 for configs[3] --, 64^2/32 - 128^2/48 - 256^2/64 rays x steps):
    the oracle against ONE image from the reference itself (tools/gen_goldens.py:gen_e2e_full) -- this is what pins "oracle == reference"
    at the shapes the GPU tests then hold the HIP path to (VERDICT r04 missing #2).  Image and depth through assert_image_parity (range
    <= 1e-5, per-pixel bound against the reference's own float64 run); 4096 sampled texels of the 100 MB tri-planes; and the integer
    rows of the importance stage on a strip of image rows: stratified samples bit-exact, searchsorted indices exact up to draws inside a
    knot window (each one explained against both cdfs), sort permutation."""
    from conftest import assert_inds_mismatches_in_window, full_golden_case, load_full_golden, report_parity
    g = load_full_golden(tag)
    cfg, sd, inp = full_golden_case(tdgp, tag)
    oracle.set_threads(os.cpu_count() or 1)
    ws = oracle.mapping_forward(sd, cfg.to_dict(), inp['z'], inp['c'])
    assert_close(ws, g['ws'], 1e-5, 'ws', 1.0)
    img, depth, inter = oracle.synthesis_forward(sd, cfg.to_dict(), g['ws'], inp['camera'], inp['u_coarse'], inp['u_fine'], 'const', return_intermediates=True)
    assert_image_parity(img, g, f'oracle {tag} full size img', full_size=True)
    assert_image_parity(depth, g, f'oracle {tag} full size depth', 'depth', full_size=True)
    pl = inter['planes'].reshape(-1)[g['planes_pick']]
    e_pl = float(np.abs(pl - g['planes_vals']).max() / g['planes_absmax'])
    report_parity(f'oracle {tag} full size tri-planes (4096 sampled texels vs the reference)', range_err=e_pl)
    assert e_pl <= 1e-5, e_pl
    h, S = cfg.img_resolution, cfg.num_ray_steps
    sel = np.concatenate([np.arange(r * h, (r + 1) * h) for r in g['rows']])
    np.testing.assert_array_equal(inter['sdist_coarse'][0, sel], g['strip_sdist_coarse'])                 # INT row (2): one sample per bin, bit-exact
    # rays: bit-identical to the reference's (norm / cross / bmm restated as the fused chains torch's CPU kernels execute; the cam2world
    # matrix is bit-identical for these cameras -- its sin / cos are 1-ulp routines in torch, so that part is held to 2e-7 in general)
    assert_close(inter['c2w'], g['c2w'], 2e-7, 'c2w', 1.0)
    if np.array_equal(inter['c2w'], g['c2w']):                # (c1 / c2 / c3; c4's camera has one matrix entry an ulp off: torch's sin / cos are 1-ulp routines)
        np.testing.assert_array_equal(inter['ray_d'][0, sel], g['strip_ray_d'])
        np.testing.assert_array_equal(inter['ray_o'][0, sel], g['strip_ray_o'])
    else:
        assert_close(inter['ray_d'][0, sel], g['strip_ray_d'], 2e-7, 'ray_d', 1.0)
        assert_close(inter['ray_o'][0, sel], g['strip_ray_o'], 2e-7, 'ray_o', 1.0)
    w_c = inter['weights_coarse'][0, sel]
    assert_close(w_c[..., 0], g['strip_weights_coarse'], 1e-5, 'coarse weights of the strip', 1.0)
    _, aux = oracle.sample_importance(inter['sdist_coarse'][:1, sel, :, None], w_c[None], inp['u_fine'].reshape(h * h, S)[sel], cfg.ray_marcher_type, return_aux=True)
    n, _ = assert_inds_mismatches_in_window(aux['inds'], g['strip_inds'], inp['u_fine'].reshape(h * h, S)[sel], g['strip_cdf'], aux['cdf'], what=f'oracle {tag} strip')
    d = np.abs(inter['sdist_fine'][0, sel, :, 0] - g['strip_sdist_fine'])
    assert np.quantile(d, 0.999) <= 2e-5 and d.max() <= 1e-3, (float(np.quantile(d, 0.999)), float(d.max()))
    assert n <= 8, n


def test_e2e_bigger(oracle, tdgp):
    """The hot MLP shape (feat 32, hid 64), 128^2 planes, 96-channel backbone, 48^2 rays x 24 steps."""
    g, img, depth, _ = _e2e(oracle, tdgp, 'e2e_bigger', tdgp.config.config_bigger(), 5)
    assert_image_parity(img, g, 'oracle e2e_bigger img', full_size=True)
    assert_image_parity(depth, g, 'oracle e2e_bigger depth', 'depth', full_size=True)


def test_e2e_tiny_cut_quantile(oracle, tdgp):
    """The non-flatness score's rendering (cut_quantile = 0.5 in both marcher calls, non_flatness_score.py:9)."""
    cfg = tdgp.config.config_tiny()
    g = load_golden('e2e_tiny')
    sd = tdgp.weights.random_state_dict(cfg, seed=21, exercise_all=True)
    cam = {k[4:]: v for k, v in g.items() if k.startswith('cam_')}
    c = cfg.to_dict()
    c['cut_quantile'] = 0.5
    img, depth = oracle.synthesis_forward(sd, c, g['ws'], cam, g['u_coarse'], g['u_fine'], 'const')
    assert np.abs(g['img_cut'] - g['img']).max() > 0.1                 # the option changes the image
    assert_close(img, g['img_cut'], 1e-5, 'img, cut_quantile 0.5', 1.0)
    assert_close(depth, g['depth_cut'], 1e-5, 'depth, cut_quantile 0.5', 1.0)


def test_cut_quantile_above_max_batch_res_is_chunked_by_rays(oracle, tdgp):
    """above max_batch_res the reference renders an eval forward with cut_quantile through run_batchwise over ray chunks of
    2**24 // (B * num_ray_steps * 3) rays (networks_epigraf.py:232-239) -- quantiles per chunk.  4 x 128^2 rays x 96 steps -> chunks of
    14563 rays; golden = the reference's image, inputs regenerated from the seed."""
    cfg = tdgp.config.config_cut_chunked()
    g = load_golden('cut_chunked')
    seed, batch, step = (int(v) for v in g['seed'])
    assert step == 2 ** 24 // (batch * cfg.num_ray_steps * 3) < cfg.img_resolution ** 2
    sd = tdgp.weights.random_state_dict(cfg, seed=seed, exercise_all=True)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=batch, seed=seed)
    c = cfg.to_dict()
    c['cut_quantile'] = 0.5
    img, depth = oracle.synthesis_forward(sd, c, g['ws'], inp['camera'], inp['u_coarse'], inp['u_fine'], 'const')
    assert_close_up_to_threshold_flips(img, g['img_cut'], 'oracle img, cut_quantile 0.5, ray-chunked')
    assert_close_up_to_threshold_flips(depth, g['depth_cut'], 'oracle depth, cut_quantile 0.5, ray-chunked')
    c['max_batch_res'] = 128                   # not above max_batch_res -> one global quantile: a different image
    img1, _ = oracle.synthesis_forward(sd, c, g['ws'], inp['camera'], inp['u_coarse'], inp['u_fine'], 'const')
    assert (np.abs(img1 - g['img_cut']) > 1e-3 * np.abs(g['img_cut']).max()).mean() > 0.2      # ... on most pixels


def test_e2e_tiny_mip(oracle, tdgp):
    cfg = tdgp.config.config_tiny()
    cfg.ray_marcher_type = 'mip'
    cfg.white_back = True
    g, img, depth, _ = _e2e(oracle, tdgp, 'e2e_tiny_mip', cfg, 41)
    assert_image_parity(img, g, 'oracle e2e_tiny_mip img')
    assert_image_parity(depth, g, 'oracle e2e_tiny_mip depth', 'depth')


# ------------------------------------------------------------------------------------------------ adaptors
@pytest.mark.parametrize('idx', [0, 1])
def test_depth_adaptor(oracle, tdgp, idx):
    """DepthAdaptor.forward (eval): normalisation, 5x5 Conv2dLayers, 1x1 heads, 'random'(= last) and 'mean' strategies."""
    from oracle import pipeline as P
    g = load_golden('adaptors')
    tag, cfg = tdgp.config.configs_adaptor_goldens()[idx]
    sd = tdgp.weights.random_state_dict(cfg, seed=51, exercise_all=True)
    res, stack = P.depth_adaptor_forward(sd, cfg.to_dict(), g[f'{tag}_depth'], g[f'{tag}_w'], return_all=True)
    assert_close(stack, g[f'{tag}_outs'], 3e-6, 'per-layer heads', 1.0)
    assert_close(res, g[f'{tag}_depth_adapted'], 3e-6, 'depth_adapted', 1.0)


@pytest.mark.parametrize('idx', [0, 1])
def test_camera_adaptor(oracle, tdgp, idx):
    """CameraAdaptor.forward: normalise -> origin / look-at ParamsAdaptors -> denormalise -> adjust_for_prior (residual on/off)."""
    from oracle import pipeline as P
    g = load_golden('adaptors')
    tag, cfg = tdgp.config.configs_adaptor_goldens()[idx]
    sd = tdgp.weights.random_state_dict(cfg, seed=51, exercise_all=True)
    cam = {k: g[f'{tag}_cam_{k}'] for k in ('angles', 'fov', 'radius', 'look_at')}
    new = P.camera_adaptor_forward(sd, cfg.to_dict(), cam, g[f'{tag}_z'], g[f'{tag}_c'] if cfg.c_dim > 0 else None)
    for k in ('angles', 'fov', 'radius', 'look_at'):
        assert_close(new[k], g[f'{tag}_new_{k}'], 2e-6, k, 1.0)


# ------------------------------------------------------------------------------------------------ bias_act grads
def _bias_act_grad_case(g, act, tag):
    """Arguments of the plugin calls BiasActCudaGrad makes (bias_act.py:165-197) for this activation: which of x / y is saved."""
    ref = dict(linear='', relu='y', lrelu='y', tanh='y', sigmoid='y', elu='y', selu='y', softplus='y', swish='x')[act]
    kw = dict(clamp=0.8, gain=1.3) if tag else {}
    xref = g['x'] if ref == 'x' or tag else None          # bias_act.py:152-153: x is saved when 'x' in ref or has_2nd_grad ... (clamp needs y)
    yref = g[f'y_{act}{tag}'] if ref == 'y' or tag else None
    if act == 'swish':
        xref, yref = g['x'], (g[f'y_{act}{tag}'] if tag else None)
    return xref, yref, kw


@pytest.mark.parametrize('tag', ['', '_clamp'])
def test_bias_act_grad(oracle, tag):
    g = load_golden('bias_act_grad')
    for act in ('linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish'):
        xref, yref, kw = _bias_act_grad_case(g, act, tag)
        dx = oracle.bias_act_grad(g['dy'], g['b'], xref, yref, None, 1, act=act, **kw)
        assert_close(dx, g[f'dx_{act}{tag}'], 2e-5, f'dx {act}{tag}', 1.0)
        ddy = oracle.bias_act_grad(g['d2'], g['b'], xref, yref, None, 1, act=act, **kw)            # d(dx)/d(dy) . d2: the first-derivative form again
        assert_close(ddy, g[f'ddy_{act}{tag}'], 2e-5, f'ddy {act}{tag}', 1.0)
        ddx = oracle.bias_act_grad(g['d2'], g['b'], xref, yref, g['dy'], 2, act=act, **kw)
        assert_close(ddx, g[f'ddx_{act}{tag}'], 5e-5, f'ddx {act}{tag}', 1.0)


@pytest.mark.parametrize('name', ['up2', 'fir', 'down2', 'asym'])
def test_upfirdn2d_backward(oracle, name):
    """The input gradient of upfirdn2d is another upfirdn2d (upfirdn2d.py:251-265): checked against autograd through the reference."""
    from conftest import UPFIRDN_GRAD_CASES, upfirdn2d_backward_args
    g = load_golden('upfirdn2d_grad')
    dy, f = g[f'{name}_dy'], g[f'{name}_f']
    dx = oracle.upfirdn2d(dy, f, **upfirdn2d_backward_args(UPFIRDN_GRAD_CASES[name], f.shape, dy.shape))
    assert_close(dx, g[f'{name}_dx'], 2e-6, f'dx {name}', 1.0)


# ------------------------------------------------------------------------------------------------ training-mode forward
def _train_kwargs(g):
    return dict(resolution=16, patch_scales=g['scales'], patch_offsets=g['offsets'], density_noise=float(g['nerf_noise_std']),
                n_coarse=g['n_coarse'], n_fine=g['n_fine'])


def test_training_mode_forward(oracle, tdgp):
    """SynthesisNetwork.forward in .train() (networks_epigraf.py:220-233): patch rays at train_resolution + density noise from the
    progressive schedule, against the reference run with the same uniform / normal draws."""
    g = load_golden('train_forward')
    cfg = tdgp.config.config_train_golden()
    sd = tdgp.weights.random_state_dict(cfg, seed=91, exercise_all=True)
    cam = {k[4:]: v for k, v in g.items() if k.startswith('cam_')}
    img, depth = oracle.synthesis_forward(sd, cfg.to_dict(), g['ws'], cam, g['u_coarse'], g['u_fine'], 'const', training=_train_kwargs(g))
    assert img.shape == g['img'].shape == (2, 3, 16, 16)
    assert_image_parity(img, g, 'oracle img (training mode)')
    assert_image_parity(depth, g, 'oracle depth (training mode)', 'depth')
    # the noise matters: the same call without it must be visibly different
    img0, _ = oracle.synthesis_forward(sd, cfg.to_dict(), g['ws'], cam, g['u_coarse'], g['u_fine'], 'const',
                                       training=dict(_train_kwargs(g), density_noise=0.0))
    assert np.abs(img0 - g['img']).max() > 1e-3


def test_progressive_schedule_and_w_avg(tdgp):
    """linear_schedule / progressive_update (training_utils.py:8-18, networks_epigraf.py:191-194) and the W moving average
    (layers.py:156-159); the mapping network runs on the CPU here."""
    import torch
    g = load_golden('train_forward')
    cfg = tdgp.config.config_train_golden()
    G = tdgp.generator.Generator(cfg)
    G.load_numpy_state_dict(tdgp.weights.random_state_dict(cfg, seed=91, exercise_all=True))
    G.progressive_update(3000)
    assert G.synthesis.nerf_noise_std == float(g['nerf_noise_std']) == 0.4
    ls = tdgp.adaptors.linear_schedule
    assert ls(0, 1.0, 0.0, 5000) == 1.0 and ls(5000, 1.0, 0.0, 5000) == 0.0 and ls(9999, 1.0, 0.0, 5000) == 0.0 and ls(1250, 0.0, 1.0, 5000) == 0.25
    assert G.synthesis.train_resolution == 16 and G.synthesis.test_resolution == cfg.img_resolution
    with torch.no_grad():
        ws = G.mapping(torch.from_numpy(g['z']), torch.from_numpy(g['c']), update_emas=True)
    assert_close(ws.numpy(), g['ws'], 1e-5, 'ws', 1.0)
    assert_close(G.mapping.w_avg.numpy(), g['w_avg_after'], 1e-6, 'w_avg', 1.0)


# ------------------------------------------------------------------------------------------------ conv2d_gradfix
@pytest.mark.parametrize('name', ['k3', 'k1', 'k5', 'k3s2', 'k3p0'])
def test_conv2d_grad_oracle(oracle, name):
    """Weight gradient (any stride / padding) and, for the 'same' stride-1 forms, forward and input gradient of the oracle against
    autograd through the reference's conv2d_gradfix.conv2d."""
    from conftest import CONV_GRAD_CASES
    g, c = load_golden('conv2d_grad'), CONV_GRAD_CASES[name]
    dw = oracle.conv2d_weight_grad(g[f'{name}_x'], g[f'{name}_dy'], c['k'], c['stride'], c['pad'])
    assert_close(dw, g[f'{name}_dw'], 2e-6, 'dw', 1.0)
    if c['stride'] == 1 and c['pad'] == c['k'] // 2:
        y = oracle.conv2d_same(g[f'{name}_x'], g[f'{name}_w']) + g[f'{name}_b'][None, :, None, None]
        assert_close(y, g[f'{name}_y'], 2e-6, 'y', 1.0)
        assert_close(oracle.conv2d_input_grad(g[f'{name}_dy'], g[f'{name}_w']), g[f'{name}_dx'], 2e-6, 'dx', 1.0)


def test_fma_op(tdgp):
    """ops.fma (fma.py:17-60): value and the un-broadcast gradients, bit for bit (eager tensor arithmetic on both sides)."""
    import torch
    g = load_golden('conv2d_grad')
    a, b, c = (torch.from_numpy(g[f'fma_{k}']).requires_grad_(True) for k in 'abc')
    out = tdgp.ops.fma.fma(a, b, c)
    np.testing.assert_array_equal(out.detach().numpy(), g['fma_out'])
    da, db, dc = torch.autograd.grad(out, [a, b, c], torch.from_numpy(g['fma_dout']))
    for got, key in ((da, 'fma_da'), (db, 'fma_db'), (dc, 'fma_dc')):
        np.testing.assert_array_equal(got.numpy(), g[key])


def test_conv2d_gradfix_cpu_fallback(tdgp):
    """CPU tensors take the reference's own fallback (torch conv2d), conv2d_gradfix.py:36-39."""
    import torch
    g = load_golden('conv2d_grad')
    y = tdgp.ops.conv2d_gradfix.conv2d(torch.from_numpy(g['k3_x']), torch.from_numpy(g['k3_w']), torch.from_numpy(g['k3_b']), padding=1)
    assert_close(y.numpy(), g['k3_y'], 1e-6, 'y', 1.0)


@pytest.mark.parametrize('tag', ['cl_inf', 'cl_noinf_lastback', 'cl_relu', 'mip_inf', 'mip_noinf_white_bias'])
def test_ray_march_grad_oracle(oracle, tag):
    """Gradients of both ray marchers (colours, raw densities) against autograd through the reference, every option."""
    from conftest import MARCH_GRAD_CASES
    g, kw = load_golden('march_grad'), MARCH_GRAD_CASES[tag]
    dc, dd = oracle.ray_march_grad(g[f'{tag}_c'], g['densities'], g['depths'], g[f'{tag}_d_rgb'], g[f'{tag}_d_depth'], g[f'{tag}_d_weights'], **kw)
    assert_close(dc, g[f'{tag}_dc'], 2e-6, 'd_colors', 1.0)
    assert_close(dd, g[f'{tag}_dd'], 1e-5, 'd_densities', 1.0)


@pytest.mark.parametrize('tag', ['small', 'hot'])
@pytest.mark.parametrize('marcher', ['classical', 'mip'])
def test_field_grad_oracle(oracle, tag, marcher):
    """grid_sample backward + MLP backward of the tri-plane field against autograd through the reference."""
    g = load_golden('field_grad')
    k = f'{tag}_{marcher}_'
    dp, dw0, db0, dw1, db1 = oracle.triplane_field_grad(g[f'{tag}_planes'], g[f'{tag}_coords'], g[k + 'w0'], g[k + 'b0'], g[k + 'w1'], g[k + 'b1'],
                                                        g[f'{tag}_d_rgb'], g[f'{tag}_d_sigma'], scale=0.5, mlp_mode=marcher)
    for got, name in ((dp, 'd_planes'), (dw0, 'd_w0'), (db0, 'd_b0'), (dw1, 'd_w1'), (db1, 'd_b1')):
        assert_close(got, g[k + name], 2e-5, name, 1.0)


@pytest.mark.parametrize('marcher', ['classical', 'mip'])
def test_importance_render_grad_oracle(oracle, marcher):
    """Gradient of the whole renderer (coarse pass, constant importance samples, fine pass, sorted merge, marcher) w.r.t. the planes and
    the MLP tensors against autograd through the reference's ImportanceRenderer.forward."""
    g = load_golden('render_grad')
    mlp = tuple(g[f'{marcher}_{n}'] for n in ('w0', 'b0', 'w1', 'b1'))
    opts = dict(box_size=1.0, num_proposal_steps=8, num_fine_steps=8, clamp_mode='softplus', use_inf_depth=True, ray_start=0.75, ray_end=1.25,
                white_back=(marcher == 'mip'), density_bias=0.0, ray_marcher_type=marcher)
    rgb, _, _, _ = oracle.importance_render(g['planes'], mlp, g['ray_o'], g['ray_d'], opts, g['u_coarse'], g['u_fine'])
    assert_close(rgb, g[f'{marcher}_rgb'], 1e-5, 'rgb', 1.0)
    res = oracle.importance_render_grad(g['planes'], mlp, g['ray_o'], g['ray_d'], opts, g['u_coarse'], g['u_fine'], g['d_rgb'], g['d_depth'])
    for got, name in zip(res, ('d_planes', 'd_w0', 'd_b0', 'd_w1', 'd_b1')):
        assert_close(got, g[f'{marcher}_{name}'], 5e-5, name, 1.0)


@pytest.mark.parametrize('tag,demod', [('c3', True), ('rgb', False), ('c3big', True)])
def test_modconv_grad_oracle(oracle, tag, demod):
    """The gradient algebra of the stride-1 modulated convolution, composed from the oracle's pieces in float64 numpy (the same
    decomposition the HIP autograd function uses), against autograd through the reference's unfused modulated_conv2d."""
    g = load_golden('modconv_grad')
    x, w, s, y, dy = (g[f'{tag}_{k}'].astype(np.float64) for k in ('x', 'w', 's', 'y', 'dy'))
    k = w.shape[2]
    w2 = (w ** 2).sum((2, 3))
    d = 1.0 / np.sqrt((s ** 2) @ w2.T + 1e-8) if demod else np.ones((x.shape[0], w.shape[0]))
    dyd = dy * d[:, :, None, None]
    dxm = oracle.conv2d_input_grad(dyd.astype(np.float32), g[f'{tag}_w']).astype(np.float64)
    dx = dxm * s[:, :, None, None]
    ds = (dxm * x).sum((2, 3))
    dw = oracle.conv2d_weight_grad((x * s[:, :, None, None]).astype(np.float32), dyd.astype(np.float32), k, 1, k // 2).astype(np.float64)
    if demod:
        t = (dy * y).sum((2, 3)) / d * d ** 3
        ds = ds - s * (t @ w2)
        dw = dw - w * (t.T @ (s ** 2))[:, :, None, None]
    assert_close(dx, g[f'{tag}_dx'], 1e-5, 'dx', 1.0)
    assert_close(dw, g[f'{tag}_dw'], 1e-5, 'dw', 1.0)
    assert_close(ds, g[f'{tag}_ds'], 1e-5, 'ds', 1.0)


# ------------------------------------------------------------------------------------------------ BASELINE configs[4]: bf16 blocks
def _bf16_vals(a):
    """fixture array -> fp32 values (bf16 tensors are stored as their 16 bits)."""
    return (a.astype(np.uint32) << 16).view(np.float32) if a.dtype == np.uint16 else a


def _ulp_bf16(ref):
    return np.maximum(np.abs(ref), 2.0 ** -126) * 2.0 ** -7        # one bf16 ulp is 2^-7 .. 2^-8 of the value


@pytest.mark.parametrize('tag', ['c3', 'up', 'rgb'])
def test_bf16_modconv(oracle, tag):
    """modulated_conv2d on bf16 activations (the reference's reduced-precision path run with bfloat16): pre-normalisation, bf16
    per-sample weights, fp32 accumulation, bf16 outputs -- the oracle follows the same rounding points, so it lands on the SAME bf16
    value except where an fp32 accumulation-order difference crosses a rounding boundary: <= 1 bf16 ulp, on a small share of elements."""
    g = load_golden('bf16')
    k, up, demod = g[f'{tag}_meta']
    y = oracle.modulated_conv2d(g[f'{tag}_x'], g[f'{tag}_w'], g[f'{tag}_s'], noise=g.get(f'{tag}_noise'), up=int(up), demodulate=bool(demod),
                                resample_filter=g['f'], prec='bf16')
    ref = g[f'{tag}_y']
    assert np.array_equal(y, oracle.round_bf16(y))                                  # bf16 values
    assert (np.abs(y - ref) <= 1.01 * _ulp_bf16(ref)).all()
    assert (y != ref).mean() < 0.02, (y != ref).mean()
    act = oracle.bias_act_bf16(ref, g[f'{tag}_b'], act='lrelu' if demod else 'linear', clamp=256)
    np.testing.assert_array_equal(act, g[f'{tag}_act'])                             # elementwise chain on identical inputs: exact


def test_bf16_full_size_backbone(oracle, tdgp):
    """BASELINE configs[4] at its REAL size: the oracle's bf16 backbone (blocks 64^2 ... 512^2 in bfloat16, 512 channels) against 16384 texels of
    the tri-planes the REFERENCE's own reduced-precision path produced with bfloat16 (tests/golden/bf16_full_c5.npz, tools/gen_goldens.py:
    gen_bf16_full) -- the pin of "oracle == reference" for the bf16 arithmetic at the shapes the GPU test holds the HIP kernels to."""
    g = load_golden('bf16_full_c5')
    cfg = tdgp.config.config_c5()
    seed = int(g['seed'][0])
    sd = tdgp.weights.random_state_dict(cfg, seed=seed, exercise_all=True)
    inp = tdgp.weights.synthetic_inputs(cfg, batch=1, seed=seed + 1)
    from oracle import pipeline as P
    oracle.set_threads(os.cpu_count() or 1)
    ws = oracle.mapping_forward(sd, cfg.to_dict(), inp['z'], inp['c'])
    assert_close(ws, g['ws'], 1e-5, 'ws', 1.0)
    planes = P.synthesis_backbone(sd, cfg.to_dict(), g['ws'], 'const')
    got = planes.reshape(-1)[g['planes_pick']]
    err = np.abs(got - g['planes_vals']) / g['planes_absmax']
    from conftest import report_parity
    report_parity('oracle bf16 C5 full size tri-planes (16384 sampled texels vs the reference)', max_err=float(err.max()), mean_err=float(err.mean()))
    assert err.max() <= 1.5e-2 and err.mean() <= 2e-3, (float(err.max()), float(err.mean()))


def test_bf16_backbone_and_image(oracle, tdgp):
    """The generator with its two highest-resolution blocks in bf16 (config_mid_bf16), block by block against the reference's own run."""
    cfg = tdgp.config.config_mid_bf16()
    g = load_golden('bf16')
    sd = tdgp.weights.random_state_dict(cfg, seed=61, exercise_all=True)
    c = cfg.to_dict()
    from oracle import pipeline as P
    assert P.fp16_resolution(c) == 32
    planes, inter = P.synthesis_backbone(sd, c, g['ws'], 'const', return_intermediates=True)
    assert_close(inter['x16'], g['x16'], 5e-6, 'x16 (fp32 block)', 1.0)
    for r in (32, 64):
        ref = _bf16_vals(g[f'x{r}'])
        got = inter[f'x{r}']
        assert np.array_equal(got, oracle.round_bf16(got))
        # same rounding points; a flipped rounding upstream moves a few downstream values by a bf16 ulp or two
        bad = np.abs(got - ref) > 2.01 * _ulp_bf16(ref) + 1e-3 * np.abs(ref).max()
        assert bad.mean() < 1e-3, (r, bad.mean())
        assert (got != ref).mean() < 0.1, (r, (got != ref).mean())
    # the fp32 skip image accumulates the bf16 ToRGB outputs: a flipped rounding is one bf16 ulp of that output (0.4-0.8 % of it)
    assert_close(planes, g['planes'], 8e-3, 'tri-planes (fp32 skip image fed by bf16 blocks)', 1.0)
    assert (np.abs(planes - g["planes"]) > 1e-3 * np.abs(g["planes"]).max()).mean() < 1e-2        # ... and it is a 0.3 % minority
    cam = {k[4:]: v for k, v in g.items() if k.startswith('cam_')}
    img, depth = P.synthesis_forward(sd, c, g['ws'], cam, g['u_coarse'], g['u_fine'], 'const')
    assert_close(img, g['img'], 5e-3, 'img', 1.0)
    assert_close(depth, g['depth'], 2e-3, 'depth', 1.0)


# ------------------------------------------------------------------------------------------------ round 6: decoders outside the fused form, camera-conditioned mapping
MLP_VARIANTS = dict(n3=dict(F=8, hid=16, n=3, view=False, marcher='classical'), n4mip=dict(F=8, hid=16, n=4, view=False, marcher='mip'),
                    view=dict(F=8, hid=3, n=2, view=True, marcher='classical'), odd=dict(F=12, hid=20, n=2, view=False, marcher='mip'))


@pytest.mark.parametrize('tag', sorted(MLP_VARIANTS))
def test_mlp_variants_oracle(oracle, tag):
    """TriPlaneMLP with n_layers != 2 / has_view_cond / widths outside the fused kernel's table (networks_epigraf.py:35-43): the oracle's lookup + the
    n-layer decode against the reference's simple_tri_plane_renderer (tests/golden/mlp_variants.npz)."""
    from oracle import pipeline
    g, v = load_golden('mlp_variants'), MLP_VARIANTS[tag]
    nl = v['n']
    ws, bs = [g[f'{tag}_w{i}'] for i in range(nl)], [g[f'{tag}_b{i}'] for i in range(nl)]
    # the lookup: orc_triplane_field's feature output (any 2-layer weights of the right width serve the call)
    F = v['F']
    feats = oracle.triplane_field(g[f'{tag}_planes'], g['coords'], np.zeros((16, F), np.float32), np.zeros(16, np.float32), np.zeros((4, 16), np.float32),
                                  np.zeros(4, np.float32), 0.5, return_feats=True)['feats']
    rgb, sigma = pipeline.triplane_decode(feats, ws, bs, v['marcher'])
    assert_close(rgb, g[f'{tag}_rgb'], 2e-6, f'{tag} rgb', max(1.0, float(np.abs(g[f'{tag}_rgb']).max())))
    assert_close(sigma, g[f'{tag}_sigma'], 2e-6, f'{tag} sigma', max(1.0, float(np.abs(g[f'{tag}_sigma']).max())))


@pytest.mark.parametrize('tag', ['four', 'raw'])
def test_mapping_camera_cond_oracle(oracle, tag):
    """MappingNetwork(camera_cond=True) (layers.py:84-93,127-138): explicit angles (yaw beyond +-2 pi), the mean-camera stand-in, truncation."""
    from oracle import pipeline
    g = load_golden('mapping_cam')
    sd = {'mapping.' + k.split('::', 1)[1]: v for k, v in g.items() if k.startswith(tag + '::')}
    c_dim = g[f'{tag}_c'].shape[1]
    cfg = dict(z_dim=16, c_dim=c_dim, w_dim=24, map_depth=2, camera_cond=True, camera_raw_scalars=(tag == 'raw'))
    import oracle.pipeline as P
    nws_was, P.num_ws = P.num_ws, (lambda cfg_: 5)
    try:
        ws = pipeline.mapping_forward(sd, cfg, g[f'{tag}_z'], g[f'{tag}_c'], camera_angles=g[f'{tag}_angles'])
        ws_mean = pipeline.mapping_forward(sd, cfg, g[f'{tag}_z'], g[f'{tag}_c'])
        ws_psi = pipeline.mapping_forward(sd, cfg, g[f'{tag}_z'], g[f'{tag}_c'], truncation_psi=0.6, camera_angles=g[f'{tag}_angles'])
    finally:
        P.num_ws = nws_was
    for got, key in ((ws, 'ws'), (ws_mean, 'ws_mean'), (ws_psi, 'ws_psi06')):
        assert_close(got, g[f'{tag}_{key}'], 1e-5, f'{tag} {key}', max(1.0, float(np.abs(g[f"{tag}_{key}"]).max())))
