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
