"""Host-side FID aggregation and camera priors against vectors captured from the reference."""
import numpy as np
import pytest
import torch

from conftest import load_golden


def test_feature_stats_mean_cov(tdgp):
    """FeatureStats.append / get_mean_cov (metric_utils.py:128-161): fp64 accumulation, max_items truncation -- bit-identical."""
    g = load_golden('metrics')
    st = tdgp.metrics.FeatureStats(capture_all=True, capture_mean_cov=True, max_items=200)
    for f in g['fs_feats']:
        st.append(f)
    mean, cov = st.get_mean_cov()
    assert st.num_items == int(g['fs_num_items']) == 200 and st.is_full()
    np.testing.assert_array_equal(mean, g['fs_mean'])
    np.testing.assert_array_equal(cov, g['fs_cov'])
    np.testing.assert_array_equal(st.get_all(), g['fs_all'])


def test_feature_stats_roundtrip(tdgp, tmp_path):
    g = load_golden('metrics')
    st = tdgp.metrics.FeatureStats(capture_all=True, capture_mean_cov=True)
    st.append(g['fs_feats'][0])
    st.save(tmp_path / 'stats.npz')
    st2 = tdgp.metrics.FeatureStats.load(tmp_path / 'stats.npz')
    for a, b in zip(st.get_mean_cov(), st2.get_mean_cov()):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(st.get_all(), st2.get_all())


def test_frechet_distance(tdgp):
    """frechet_inception_distance.py:35-38 against the eigenvalue form Tr sqrt(S1 S2) = sum sqrt(eig(S1 S2)), and its basic properties."""
    rs = np.random.RandomState(3)
    a, b = rs.randn(400, 12) * 2 + 1, rs.randn(500, 12) * 1.5 - 0.5
    mu1, s1, mu2, s2 = a.mean(0), np.cov(a, rowvar=False), b.mean(0), np.cov(b, rowvar=False)
    fid = tdgp.metrics.frechet_distance(mu1, s1, mu2, s2)
    ev = np.linalg.eigvals(s1 @ s2)
    ref = np.square(mu1 - mu2).sum() + np.trace(s1) + np.trace(s2) - 2 * np.sqrt(np.clip(ev.real, 0, None)).sum()
    assert abs(fid - ref) < 1e-8 * max(1.0, abs(ref))
    assert abs(tdgp.metrics.frechet_distance(mu1, s1, mu1, s1)) < 1e-6
    assert abs(fid - tdgp.metrics.frechet_distance(mu2, s2, mu1, s1)) < 1e-8 * fid


@pytest.mark.parametrize('tag', ['base', 'uniform'])
def test_sample_camera_params(tdgp, tag):
    """rendering_utils.py:146-152 with the reference's draw order: identical torch / numpy seeds -> identical cameras."""
    g = load_golden('metrics')
    cam = tdgp.metrics.camera_base()
    if tag == 'uniform':
        cam['origin'] = dict(radius=cam['origin']['radius'], angles=dict(dist='uniform', yaw=dict(min=-1.57, max=1.57), pitch=dict(min=0.785398163, max=2.35619449)))
        cam['look_at'] = dict(radius=dict(dist='uniform', min=0.0, max=0.2), angles=cam['look_at']['angles'])
    torch.manual_seed(62)
    np.random.seed(62)
    cp = tdgp.metrics.sample_camera_params(cam, 6, 'cpu')
    for k in ('angles', 'fov', 'radius', 'look_at'):
        np.testing.assert_array_equal(cp[k].numpy(), g[f'cam_{tag}_{k}'])


@pytest.mark.parametrize('name', ['point', 'front_circle', 'points', 'line'])
def test_camera_trajectories(tdgp, name):
    """inference.generate_camera_trajectory against the reference's (inference_utils.py:140-186): deterministic, so identical."""
    g = load_golden('trajectories')
    canon = tdgp.generator.TensorGroup(**{k: torch.from_numpy(g[f'canon_{k}']) for k in ('angles', 'fov', 'radius', 'look_at')})
    cp = tdgp.inference.generate_camera_trajectory(tdgp.inference_golden_trajectories()[name], canon)
    for k in ('angles', 'fov', 'radius', 'look_at'):
        np.testing.assert_array_equal(np.asarray(cp[k]), g[f'{name}_{k}'])


def test_tensor_group(tdgp):
    TG = tdgp.generator.TensorGroup
    a = TG(x=torch.arange(6.).reshape(3, 2), y=torch.arange(3.))
    assert len(a) == 3 and a[1:].x.shape == (2, 2) and a.repeat_interleave(2, dim=0).y.tolist() == [0, 0, 1, 1, 2, 2]
    b = (a * 2 + 1).clamp(0, 5)
    assert b.y.tolist() == [1, 3, 5] and TG.cat([a, a]).x.shape == (6, 2) and a.mean(dim=0, keepdim=True).y.shape == (1,)
    assert not hasattr(a, 'no_such_thing')


def _cpu_generator(tdgp):
    cfg = tdgp.config.config_mid()
    G = tdgp.generator.Generator(cfg)
    G.load_numpy_state_dict(tdgp.weights.random_state_dict(cfg, seed=11, exercise_all=True))
    return G.eval()


def test_seed_samplers(tdgp):
    """scripts/inference.py:87-106: z rows and class draws are functions of the seed alone -- bit-identical."""
    g = load_golden('harness')
    cfg = tdgp.config.config_mid()
    seeds = [int(s) for s in g['seeds']]
    np.testing.assert_array_equal(tdgp.inference.sample_z_from_seeds(seeds, cfg.z_dim).numpy(), g['z'])
    np.testing.assert_array_equal(tdgp.inference.sample_c_from_seeds(seeds, cfg.c_dim).numpy(), g['c'])
    assert tdgp.inference.sample_c_from_seeds(seeds, 0).shape == (len(seeds), 0)


@pytest.mark.parametrize('tag,kw', [('psi1', dict(truncation_psi=1.0)), ('psi06', dict(truncation_psi=0.6)),
                                    ('psi06_cls', dict(truncation_psi=0.6, classes=[1, 4])), ('psi1_cls', dict(truncation_psi=1.0, classes=[1, 4])),
                                    ('interp', dict(truncation_psi=0.8, num_interp_steps=5))])
def test_sample_ws_from_seeds(tdgp, tag, kw):
    """scripts/inference.py:110-150: plain / per-class-centre truncation / seeds x classes grid / pairwise interpolation.  The mapping
    network runs on the CPU here (tiny torch GEMMs, as in the reference); same torch seed -> same truncation-centre samples."""
    g = load_golden('harness')
    G = _cpu_generator(tdgp)
    seeds = [int(s) for s in g['seeds']]
    torch.manual_seed(81)
    with torch.no_grad():
        ws, z, c = tdgp.inference.sample_ws_from_seeds(G, seeds, device='cpu', **kw)
    assert ws.shape == g[f'ws_{tag}'].shape
    np.testing.assert_allclose(ws.numpy(), g[f'ws_{tag}'], rtol=0, atol=2e-6)
    if not isinstance(z, tuple):
        np.testing.assert_array_equal(z.numpy(), g[f'z_{tag}'])
        np.testing.assert_array_equal(c.numpy(), g[f'c_{tag}'])


class _Dataset:
    """The same index -> (label, angles) functions as tools/gen_goldens.py:_GoldenDataset."""

    def __init__(self, c_dim, size=11):
        self.c_dim, self.size = c_dim, size

    def __len__(self):
        return self.size

    def get_label(self, i):
        v = np.zeros(self.c_dim, np.float32)
        v[i % self.c_dim] = 1.0
        return v

    def get_camera_angles(self, i):
        return np.array([0.1 * i - 0.5, 1.2 + 0.03 * i, 0.0], np.float32)


@pytest.mark.parametrize('tag,c_dim,custom,frontal', [('cond', 5, False, False), ('custom', 5, True, False), ('frontal', 5, False, True),
                                                      ('uncond', 0, False, False), ('uncond_custom', 0, True, False)])
def test_iterate_random_conditioning(tdgp, tag, c_dim, custom, frontal):
    """metric_utils.py:60-101: numpy index draws then the torch/scipy prior draws, in the reference's order -- bit-identical batches."""
    g = load_golden('harness')
    cam = tdgp.metrics.camera_base()
    if custom:
        cam['origin'] = dict(radius=cam['origin']['radius'], angles=dict(dist='custom'))

    class _G:
        pass
    _G.c_dim = c_dim
    torch.manual_seed(82)
    np.random.seed(82)
    it = tdgp.metrics.iterate_random_conditioning(_G, 4, 'cpu', cam, dataset=_Dataset(max(c_dim, 1)), frontal_camera=frontal)
    for step in range(2):
        c, cp = next(it)
        np.testing.assert_array_equal(c.numpy(), g[f'it_{tag}_{step}_c'])
        for k in ('angles', 'fov', 'radius', 'look_at'):
            np.testing.assert_array_equal(cp[k].numpy(), g[f'it_{tag}_{step}_{k}'], err_msg=f'{tag} step {step} {k}')
    if c_dim or custom:
        with pytest.raises(ValueError):
            next(tdgp.metrics.iterate_random_conditioning(_G, 4, 'cpu', cam))


@pytest.mark.parametrize('tag', ['plain', 'full', 'extra'])
def test_discriminator_module_cpu(tdgp, tag):
    """Discriminator module logic (block wiring, conditioning, hyper-modulation, minibatch stddev, Fourier features) on CPU tensors --
    where every op takes the reference's own torch fallback -- against the reference: logits, d/d img and all parameter gradients.
    The state dict of the module loads into the reference with strict=True (tools/gen_goldens.py)."""
    from conftest import check_discriminator
    assert check_discriminator(tdgp, tag, 'cpu', 2e-5) >= 17


@pytest.mark.parametrize('tag', ['plain', 'full'])
def test_discriminator_r1_cpu(tdgp, tag):
    """R1 penalty of the discriminator module and its second-order parameter gradients (CPU tensors, torch fallbacks) vs the reference."""
    from conftest import check_discriminator_r1
    assert check_discriminator_r1(tdgp, tag, 'cpu', 5e-5) >= 10


def test_training_utils(tdgp):
    """Patch sampling (numpy + torch RNG in the reference's draw order), patch extraction and the blur of loss.py, bit for bit /
    to fp32 rounding against the reference (CPU tensors)."""
    import torch
    g = load_golden('loss')
    TR = tdgp.training
    for dist, extra in (('uniform', {}), ('beta', dict(alpha=1.0, beta=0.4))):
        pc = TR.PatchConfig(distribution=dist, min_scale=0.3, max_scale=0.9, mbstd_group_size=2, **extra)
        np.random.seed(7)
        torch.manual_seed(7)
        pp = TR.sample_patch_params(8, pc, device='cpu')
        np.testing.assert_array_equal(pp['scales'].numpy(), g[f'sp_{dist}_scales'])
        np.testing.assert_array_equal(pp['offsets'].numpy(), g[f'sp_{dist}_offsets'])
    pp0 = dict(scales=torch.from_numpy(g['pp0_scales']), offsets=torch.from_numpy(g['pp0_offsets']))
    np.testing.assert_allclose(TR.extract_patches(torch.from_numpy(g['real']), pp0, 16).numpy(), g['patches'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(TR.maybe_blur(torch.from_numpy(g['real']), 1.3).numpy(), g['blurred'], rtol=0, atol=2e-6)
    # schedule of the patch scale (loss.py:53-62)
    pc = TR.PatchConfig(distribution='uniform', min_scale_trg=0.25, max_scale=1.0, anneal_kimg=100)
    loss = TR.StyleGAN2Loss(None, None, 'cpu', patch_cfg=pc)
    assert pc.min_scale == 1.0
    loss.progressive_update(50)
    assert abs(pc.min_scale - 0.625) < 1e-12
    loss.progressive_update(500)
    assert pc.min_scale == 0.25


def test_training_driver_pieces(tdgp):
    """setup_phases (lazy-regularisation rescaling, training_loop.py:180-205), update_ema (:357-367) and train_iteration (:319-347: phase
    intervals, sub-batch accumulation, one optimiser step per phase) on small CPU modules with a recording loss."""
    import copy
    import torch
    TR = tdgp.training
    torch.manual_seed(0)
    G, D = torch.nn.Linear(4, 3), torch.nn.Linear(3, 1)
    phases = TR.setup_phases(G, D, dict(lr=0.0025, betas=[0.0, 0.99], eps=1e-8), dict(lr=0.002, betas=[0.0, 0.99], eps=1e-8), G_reg_interval=None,
                             D_reg_interval=16)
    assert [p['name'] for p in phases] == ['Gall', 'Dmain', 'Dreg'] and [p['interval'] for p in phases] == [1, 1, 16]
    assert phases[1]['opt'] is phases[2]['opt']
    g = phases[1]['opt'].param_groups[0]
    assert abs(g['lr'] - 0.002 * 16 / 17) < 1e-12 and abs(g['betas'][1] - 0.99 ** (16 / 17)) < 1e-12 and g['betas'][0] == 0.0
    assert phases[0]['opt'].param_groups[0]['lr'] == 0.0025
    # EMA: beta = 0.5 ** (batch / min(ema_kimg * 1000, cur_nimg * rampup)); buffers copied
    G_ema = copy.deepcopy(G)
    with torch.no_grad():
        for p in G.parameters():
            p.add_(1.0)
    before = [p.clone() for p in G_ema.parameters()]
    beta = TR.update_ema(G_ema, G, cur_nimg=20000, batch_size=64, ema_kimg=10.0, ema_rampup=0.05)
    assert abs(beta - 0.5 ** (64 / 1000.0)) < 1e-12
    for pe, p, b in zip(G_ema.parameters(), G.parameters(), before):
        assert torch.allclose(pe, p + (b - p) * beta, atol=1e-6)
    assert TR.update_ema(G_ema, G, cur_nimg=500, batch_size=64, ema_start_kimg=1.0) == 0.0
    assert all(torch.equal(pe, p) for pe, p in zip(G_ema.parameters(), G.parameters()))

    class _Loss:
        def __init__(self):
            self.calls = []

        def accumulate_gradients(self, phase, real_data, gen_data, gain, cur_nimg):
            self.calls.append((phase, tuple(real_data.img.shape), tuple(gen_data.z.shape), gain))
            module = G if phase.startswith('G') else D
            assert all(p.requires_grad for p in module.parameters())
            (sum(p.sum() for p in module.parameters()) * gain).backward()

    TG = tdgp.generator.TensorGroup
    loss = _Loss()
    real = TG(img=torch.zeros(8, 3, 4, 4), c=torch.zeros(8, 0))
    gen = TG(z=torch.arange(24.).view(24, 1), c=torch.zeros(24, 0), camera_params=TG(angles=torch.zeros(24, 3), fov=torch.zeros(24)))
    w0 = D.weight.detach().clone()
    assert TR.train_iteration(loss, phases, real, gen, batch_idx=3, cur_nimg=0, batch_size=8, batch_gpu=4, world=1) == ['Gall', 'Dmain']
    assert [c[0] for c in loss.calls] == ['Gall', 'Gall', 'Dmain', 'Dmain'] and loss.calls[0][1] == (4, 3, 4, 4) and loss.calls[2][2] == (4, 1)
    assert not torch.equal(D.weight, w0) and not any(p.requires_grad for p in D.parameters())
    loss.calls.clear()
    assert TR.train_iteration(loss, phases, real, gen, batch_idx=16, cur_nimg=0, batch_size=8, batch_gpu=8, world=1) == ['Gall', 'Dmain', 'Dreg']
    assert loss.calls[-1][0] == 'Dreg' and loss.calls[-1][3] == 16


def test_bench_flop_model_matches_survey(tdgp):
    """bench.py's algorithmic FLOP per image (what `roofline.achieved` is computed from) against SURVEY.md 8(d): conv1 83.7 G,
    conv0 (x2 layers) 35.4 G, ToRGB 6.2 G, renderer 45.1 G for BASELINE configs[2]; 488.9 G backbone for configs[3]."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    f3 = bench.algorithmic_flops(tdgp.config.config_c3())
    g = {k: v[0] / 1e9 for k, v in f3.items()}
    assert abs(g['conv_mfma_kernel'] - 83.7) < 0.1 and f3['conv_mfma_kernel'][1] == 8
    assert abs(g['upconv_mfma_kernel'] - 35.4) < 0.1 and f3['upconv_mfma_kernel'][1] == 7
    assert abs(g['torgb_mfma_kernel'] - 6.2) < 0.05 and f3['torgb_mfma_kernel'][1] == 8
    assert abs(g['triplane_field_kernel'] - 45.1) < 0.1 and f3['triplane_field_kernel'][1] == 2
    assert abs(sum(g.values()) - 170.4) < 0.3                  # 171.7 G in the SURVEY includes the 1.3 G of the FIR passes (no matrix work)
    # at batch 16 the 32^2 ... 512^2 stride-1 layers run as Winograd F(4x4) (conv_wino4_kernel, round 4) and the x2 layers from 32^2 inputs up as
    # four folded parity kernels on the same GEMM (upconv_wino4_kernel); the algorithmic FLOP do not change
    f16 = bench.algorithmic_flops(tdgp.config.config_c3(), batch=16)
    # (round 6: the two layers with <= 128 channels -- 256^2 x 128, 512^2 x 64 -- run the input transform inside the GEMM kernel: conv_wino4f_kernel)
    assert f16['conv_wino4_kernel'][1] == 3 and f16['conv_wino4f_kernel'][1] == 2 and f16['conv_mfma_kernel'][1] == 3 and 'conv_wino_kernel' not in f16
    assert abs((f16['conv_wino4_kernel'][0] + f16['conv_wino4f_kernel'][0] + f16['conv_mfma_kernel'][0]) / 1e9 - 83.7) < 0.1
    assert f16['upconv_wino4_kernel'][1] == 4 and f16['upconv_mfma_kernel'][1] == 3
    assert abs((f16['upconv_wino4_kernel'][0] + f16['upconv_mfma_kernel'][0]) / 1e9 - 35.4) < 0.1
    assert abs(sum(v[0] for v in f16.values()) / 1e9 - 170.4) < 0.3
    f4b = bench.algorithmic_flops(tdgp.config.config_c3(), batch=4)
    assert f4b['conv_wino4_kernel'][1] + f4b['conv_wino4f_kernel'][1] == 5 and f4b['conv_mfma_kernel'][1] == 3        # 32^2 at B = 4: too few items for the persistent grid -> its input channels split 4 ways
    assert bench.winograd4_fused_takes(16, 64, 64, 512) and bench.winograd4_fused_takes(4, 128, 128, 256) and not bench.winograd4_fused_takes(16, 256, 256, 128) and not bench.winograd4_fused_takes(16, 72, 64, 512)
    assert bench.winograd4_takes(8, 512, 512, 32) and not bench.winograd4_takes(2, 512, 512, 32) and not bench.winograd4_takes(4, 512, 512, 16)
    assert bench.EXECUTED_FRACTION['conv_wino4_kernel'] == 0.25 and bench.EXECUTED_FRACTION['conv_wino4f_kernel'] == 0.25 and 'upconv_wino4_kernel' not in bench.EXECUTED_FRACTION    # 36/16 of 9; 4 x 36/16 = the transposed conv's 9
    f4 = bench.algorithmic_flops(tdgp.config.config_c4())
    back4 = sum(f4[k][0] for k in ('conv_mfma_kernel', 'upconv_mfma_kernel', 'torgb_mfma_kernel')) / 1e9
    assert abs(back4 - 488.9) < 0.5
    assert bench.PEAK_FP32_MFMA_TFLOPS == 157.3
