"""FID-side aggregation and camera priors of the generator harness (host side).

Reference: `src/metrics/metric_utils.py:104-169` (FeatureStats), `:288-320` (compute_feature_stats_for_generator),
`src/metrics/frechet_inception_distance.py:20-39` (compute_fid), `src/training/rendering_utils.py:72-156` (camera priors).

These are host-side pieces: fp64 mean / covariance accumulation in numpy, `scipy.linalg.sqrtm` for the Frechet distance, the
camera prior samplers (torch RNG in the reference's draw order, scipy for the truncated normal).  The device work they drive is
the generator forward (HIP) and ONE all-gather per feature block (`distributed.FeatureGatherer`, RCCL) instead of the
reference's `world` sequential broadcasts.  The Inception detector itself is a URL-fetched TorchScript pickle
(`frechet_inception_distance.py:22`) and is not reproduced: `detector` is any callable `uint8 images [N,3,H,W] -> features [N,F]`.
"""
import numpy as np
import torch

from .generator import TensorGroup


# ----------------------------------------------------------------------------------------------------------------------
# camera priors (configs/camera/base.yaml), rendering_utils.py:72-156
# ----------------------------------------------------------------------------------------------------------------------
def camera_base():
    """configs/camera/base.yaml as a nested dict."""
    return dict(
        ray=dict(start=0.75, end=1.25),
        fov=dict(dist='uniform', min=10.0, max=45.0),
        origin=dict(radius=dict(dist='normal', mean=1.0, std=0.0),
                    angles=dict(dist='truncnorm', yaw=dict(min=-1.57079633, max=1.57079633, mean=0.0, std=0.4),
                                pitch=dict(min=0.392699082, max=2.74889357, mean=1.57, std=0.2))),
        look_at=dict(radius=dict(dist='uniform', min=0.0, max=0.0),
                     angles=dict(dist='spherical_uniform', yaw=dict(min=-3.14159265, max=3.14159265), pitch=dict(min=0.0, max=3.14159265))),
        cube_scale=0.5)


def _g(cfg, path):
    for k in path.split('.'):
        cfg = cfg[k] if isinstance(cfg, dict) else getattr(cfg, k)
    return cfg


def sample_truncnorm(mean, std, lo, hi, batch_size, device):
    """rendering_utils.py:136-142 (scipy's sampler on numpy's global RNG, like the reference)."""
    from scipy.stats import truncnorm
    x = truncnorm.rvs(a=(lo - mean) / std, b=(hi - mean) / std, loc=mean, scale=std, size=(batch_size,))
    return torch.from_numpy(x).float().to(device)


def sample_camera_angles(cfg, batch_size, device):
    """rendering_utils.py:72-109: yaw / pitch / roll = 0 from the configured distribution; pitch clamped to (1e-5, pi - 1e-5)."""
    dist = _g(cfg, 'dist')
    rand = lambda: torch.rand((batch_size, 1), device=device)         # noqa: E731
    randn = lambda: torch.randn((batch_size, 1), device=device)       # noqa: E731
    if dist == 'uniform':
        yaw = rand() * (_g(cfg, 'yaw.max') - _g(cfg, 'yaw.min')) + _g(cfg, 'yaw.min')
        pitch = rand() * (_g(cfg, 'pitch.max') - _g(cfg, 'pitch.min')) + _g(cfg, 'pitch.min')
    elif dist == 'normal':
        yaw = randn() * _g(cfg, 'yaw.std') + _g(cfg, 'yaw.mean')
        pitch = randn() * _g(cfg, 'pitch.std') + _g(cfg, 'pitch.mean')
    elif dist == 'truncnorm':
        yaw = sample_truncnorm((_g(cfg, 'yaw.max') + _g(cfg, 'yaw.min')) * 0.5, _g(cfg, 'yaw.std'), _g(cfg, 'yaw.min'), _g(cfg, 'yaw.max'), batch_size, device).unsqueeze(1)
        pitch = sample_truncnorm((_g(cfg, 'pitch.max') + _g(cfg, 'pitch.min')) * 0.5, _g(cfg, 'pitch.std'), _g(cfg, 'pitch.min'), _g(cfg, 'pitch.max'), batch_size,
                                 device).unsqueeze(1)
    elif dist == 'spherical_uniform':
        yaw_range, yaw_center = _g(cfg, 'yaw.max') - _g(cfg, 'yaw.min'), 0.5 * (_g(cfg, 'yaw.max') + _g(cfg, 'yaw.min'))
        pitch_range, pitch_center = _g(cfg, 'pitch.max') - _g(cfg, 'pitch.min'), 0.5 * (_g(cfg, 'pitch.max') + _g(cfg, 'pitch.min'))
        yaw = (rand() - 0.5) * yaw_range + yaw_center
        v = (rand() - 0.5) * pitch_range + pitch_center
        v = torch.clamp(v / np.pi, 1e-5, 1 - 1e-5)
        pitch = torch.arccos(1 - 2 * v)
    else:
        raise NotImplementedError(f'Unknown distribution: {dist}')
    pitch = torch.clamp(pitch, 1e-5, np.pi - 1e-5)
    return torch.cat([yaw, pitch, torch.zeros_like(yaw)], dim=1)


def sample_bounded_scalar(cfg, batch_size, device):
    """rendering_utils.py:122-132."""
    dist = _g(cfg, 'dist')
    if dist == 'normal':
        assert _g(cfg, 'std') == 0.0, 'Scalar must be bounded'
        return torch.empty(batch_size, device=device, dtype=torch.float32).fill_(_g(cfg, 'mean'))
    if dist == 'truncnorm':
        return sample_truncnorm(_g(cfg, 'mean'), _g(cfg, 'std'), _g(cfg, 'min'), _g(cfg, 'max'), batch_size, device)
    if dist == 'uniform':
        return torch.rand(batch_size, device=device) * (_g(cfg, 'max') - _g(cfg, 'min')) + _g(cfg, 'min')
    raise NotImplementedError(f'Unknown distribution: {dist}')


def sample_camera_params(cfg, batch_size, device='cpu', origin_angles=None):
    """rendering_utils.py:146-152; draw order: origin angles, fov, radius, look-at (angles, radius)."""
    origin_angles = sample_camera_angles(_g(cfg, 'origin.angles'), batch_size, device) if origin_angles is None else origin_angles
    fov = sample_bounded_scalar(_g(cfg, 'fov'), batch_size, device)
    radius = sample_bounded_scalar(_g(cfg, 'origin.radius'), batch_size, device)
    la_angles = sample_camera_angles(_g(cfg, 'look_at.angles'), batch_size, device)
    la_radius = sample_bounded_scalar(_g(cfg, 'look_at.radius'), batch_size, device)
    look_at = torch.cat([la_angles[:, [0, 1]], la_radius.unsqueeze(1)], dim=1)
    return TensorGroup(angles=origin_angles, fov=fov, radius=radius, look_at=look_at)


# ----------------------------------------------------------------------------------------------------------------------
# feature statistics + Frechet distance
# ----------------------------------------------------------------------------------------------------------------------
class _RawMoments:
    """fp64 first and second raw moments (sum of rows, sum of row outer products) of fp32 feature rows.  One block = one
    `rows.sum(0)` and one Gram matrix `rows^T rows` in fp64, added to the running totals: the accumulation order the golden
    `metrics.npz` pins bit for bit (what `metric_utils.py:128-161` computes)."""
    __slots__ = ('s1', 's2')

    def __init__(self, width):
        self.s1 = np.zeros(width, np.float64)
        self.s2 = np.zeros((width, width), np.float64)

    def add_block(self, rows32):
        r = rows32.astype(np.float64)
        self.s1 += r.sum(axis=0)
        self.s2 += r.T @ r

    def central(self, count):
        mu = self.s1 / count
        return mu, self.s2 / count - np.outer(mu, mu)


class FeatureStats:
    """Feature accumulator of the FID loop; public surface of `metric_utils.py:104-169` (`append`, `append_torch`, `is_full`,
    `get_all`, `get_mean_cov`, `num_items`, `num_features`, `max_items`), written from its behaviour:

    * rows arrive in blocks `[n, F]` and are taken as fp32; F is fixed by the first block;
    * with `max_items` set, a block that would overshoot is cut to the remaining room and later blocks are dropped whole;
    * `capture_all` keeps the accepted rows (in arrival order), `capture_mean_cov` keeps fp64 raw moments (`_RawMoments`);
    * multi-rank: one all-gather per block, rows interleaved rank-major (`distributed.FeatureGatherer`), so that every rank
      accumulates the same sequence and row i of the gathered block came from rank i % world (`metric_utils.py:145-155`).
    """

    def __init__(self, capture_all=False, capture_mean_cov=False, max_items=None):
        self.capture_all = bool(capture_all)
        self.capture_mean_cov = bool(capture_mean_cov)
        self.max_items = None if max_items is None else int(max_items)
        self.num_items = 0
        self.num_features = None
        self._kept = []                  # accepted fp32 blocks (capture_all)
        self._moments = None             # _RawMoments (capture_mean_cov), made when F is known

    # -- bookkeeping --------------------------------------------------------------------------------------------------
    def set_num_features(self, num_features):
        """Fix the feature width F (idempotent; a different F later is an error)."""
        num_features = int(num_features)
        if self.num_features is None:
            self.num_features = num_features
            self._moments = _RawMoments(num_features)
        elif self.num_features != num_features:
            raise AssertionError(f'feature width changed: {self.num_features} -> {num_features}')

    def _room(self):
        return None if self.max_items is None else max(self.max_items - self.num_items, 0)

    def is_full(self):
        return self._room() == 0

    raw_mean = property(lambda self: None if self._moments is None else self._moments.s1)
    raw_cov = property(lambda self: None if self._moments is None else self._moments.s2)

    # -- accumulation -------------------------------------------------------------------------------------------------
    def append(self, x):
        rows = np.asarray(x, dtype=np.float32)
        if rows.ndim != 2:
            raise AssertionError(f'expected [n, F] feature rows, got shape {rows.shape}')
        room = self._room()
        if room is not None and rows.shape[0] > room:
            if room == 0:
                return                    # already full: the block is dropped before it can fix F
            rows = rows[:room]
        self.set_num_features(rows.shape[1])
        self.num_items += rows.shape[0]
        if self.capture_all:
            self._kept.append(rows)
        if self.capture_mean_cov:
            self._moments.add_block(rows)

    def append_torch(self, x, num_gpus=1, rank=0, gatherer=None):
        """Rows of every rank, interleaved (item i of the gathered block came from rank i % world, :154).  `gatherer`: a
        `distributed.FeatureGatherer` (one all-gather); created on demand when num_gpus > 1."""
        assert isinstance(x, torch.Tensor) and x.ndim == 2
        assert 0 <= rank < num_gpus
        if num_gpus > 1:
            if gatherer is None:
                from .distributed import FeatureGatherer
                gatherer = FeatureGatherer(side_stream=False)
            x = gatherer.gather(x)
        rows = x.cpu().numpy()                            # a synchronisation point: the features are on the host
        if x.is_cuda:
            from . import _lib
            _lib.raise_on_device_fault('the generator forward behind this feature block')
        self.append(rows)

    # -- results ------------------------------------------------------------------------------------------------------
    def get_all(self):
        if not self.capture_all:
            raise AssertionError('constructed without capture_all')
        return np.concatenate(self._kept, axis=0)

    def get_mean_cov(self):
        if not self.capture_mean_cov:
            raise AssertionError('constructed without capture_mean_cov')
        return self._moments.central(self.num_items)

    # -- persistence: a neutral container (npz) instead of the reference's pickle of __dict__ ---------------------------
    def save(self, path):
        width = self.num_features or 0
        np.savez(path, capture_all=self.capture_all, capture_mean_cov=self.capture_mean_cov, max_items=-1 if self.max_items is None else self.max_items,
                 num_items=self.num_items, raw_mean=self._moments.s1 if self._moments else np.zeros(width), raw_cov=self._moments.s2 if self._moments else np.zeros((width, width)),
                 all_features=self.get_all() if self.capture_all and self._kept else np.zeros([0, width], np.float32))

    @staticmethod
    def load(path):
        d = np.load(path)
        cap = int(d['max_items'])
        st = FeatureStats(capture_all=bool(d['capture_all']), capture_mean_cov=bool(d['capture_mean_cov']), max_items=None if cap < 0 else cap)
        st.set_num_features(d['raw_mean'].shape[0])
        st.num_items = int(d['num_items'])
        st._moments.s1[...] = d['raw_mean']
        st._moments.s2[...] = d['raw_cov']
        if st.capture_all and d['all_features'].size:
            st._kept = [d['all_features']]
        return st


def frechet_distance(mu_gen, sigma_gen, mu_real, sigma_real):
    """frechet_inception_distance.py:35-38."""
    import scipy.linalg
    m = np.square(mu_gen - mu_real).sum()
    s, _ = scipy.linalg.sqrtm(np.dot(sigma_gen, sigma_real), disp=False)
    return float(np.real(m + np.trace(sigma_gen + sigma_real - s * 2)))


def iterate_random_conditioning(G, batch_size, device='cpu', camera_cfg=None, dataset=None, frontal_camera=False):
    """metric_utils.py:60-101: endless (c, camera_params) batches.

    Unconditional generator with a parametric camera prior: c is the empty [batch, 0] tensor and cameras come from the prior.
    Otherwise every batch draws `batch_size` dataset indices with `np.random.randint` (one call per item, the reference's draw
    order) and takes labels (`dataset.get_label`) and -- for the 'custom' angle distribution -- camera angles
    (`dataset.get_camera_angles`) from those items.  `dataset` is any object with `__len__`, `get_label`, `get_camera_angles`
    (the reference constructs its ImageFolder dataset here; datasets are out of scope).  `frontal_camera` pins the origin
    angles to (yaw 0, pitch pi/2, roll 0)."""
    camera_cfg = camera_base() if camera_cfg is None else camera_cfg
    custom = _g(camera_cfg, 'origin.angles')['dist'] == 'custom'
    if (G.c_dim != 0 or custom) and dataset is None:
        raise ValueError('a conditional generator / a custom camera distribution draws labels and angles from a dataset')
    frontal = None
    if frontal_camera:
        frontal = torch.stack([torch.zeros(batch_size, device=device), np.pi / 2 + torch.zeros(batch_size, device=device),
                               torch.zeros(batch_size, device=device)], dim=1)
    c0 = torch.zeros([batch_size, 0], device=device) if G.c_dim == 0 else None
    if G.c_dim == 0 and not custom:
        while True:
            yield c0, sample_camera_params(camera_cfg, batch_size, device, origin_angles=frontal)
    while True:
        idx = [np.random.randint(len(dataset)) for _ in range(batch_size)]
        c = c0 if G.c_dim == 0 else torch.from_numpy(np.stack([dataset.get_label(i) for i in idx])).to(device)
        if frontal_camera:
            angles = frontal
        elif custom:
            angles = torch.from_numpy(np.stack([dataset.get_camera_angles(i) for i in idx])).to(device)
        else:
            angles = None
        yield c, sample_camera_params(camera_cfg, len(idx), device, origin_angles=angles)


def _generator_batches(G, batch_gen, camera_cfg, c_sampler, dataset, device, frontal_camera=False):
    """(z, c, camera) batches for the two feature loops below: z first, then the conditioning draw (metric_utils.py:303-307)."""
    if c_sampler is not None:                 # caller-supplied label sampler instead of a dataset
        def cond():
            while True:
                yield c_sampler(batch_gen).to(device), sample_camera_params(camera_base() if camera_cfg is None else camera_cfg, batch_gen, device)
        it = cond()
    else:
        it = iterate_random_conditioning(G, batch_gen, device, camera_cfg, dataset, frontal_camera)
    while True:
        z = torch.randn([batch_gen, G.z_dim], device=device)
        c, camera_params = next(it)
        if getattr(G.synthesis, 'camera_adaptor', None) is not None:
            camera_params = G.synthesis.camera_adaptor(camera_params, z, c)
        yield z, c, camera_params


def resolve_batch_gen(batch_size, batch_gen=None):
    """metric_utils.py:289-290 / :324-325: `min(batch_size, 4) if opts.batch_gen is None else opts.batch_gen`, which must divide batch_size."""
    batch_gen = min(batch_size, 4) if batch_gen is None else int(batch_gen)
    assert batch_gen >= 1 and batch_size % batch_gen == 0
    return batch_gen


def compute_feature_stats_for_generator(G, detector, max_items, batch_size=64, batch_gen=None, camera_cfg=None, c_sampler=None, num_gpus=1, rank=0,
                                        device='cuda', gatherer=None, G_kwargs=None, dataset=None, **stats_kwargs):
    """metric_utils.py:288-320: generate `batch_size` images per iteration in chunks of `batch_gen`, run the detector, gather the
    feature block across ranks, accumulate.  Conditioning comes from `iterate_random_conditioning` (labels / custom angles from
    `dataset`) or, when given, from `c_sampler(batch) -> c [batch, c_dim]`; cameras come from the prior (`camera_cfg`, default
    camera/base.yaml) and pass through G's camera adaptor when it has one.
    `batch_gen` is the caller's option it is in the reference (MetricOptions.batch_gen, metric_utils.py:26,36): None = the reference's default
    min(batch_size, 4) (:289); 16 generates the same images 8-9 % faster on one MI355X (bench.py --fid-loop prints both)."""
    batch_gen = resolve_batch_gen(batch_size, batch_gen)
    G_kwargs = {} if G_kwargs is None else G_kwargs
    stats = FeatureStats(max_items=max_items, **stats_kwargs)
    batches = _generator_batches(G, batch_gen, camera_cfg, c_sampler, dataset, device)
    while not stats.is_full():
        images = []
        for _ in range(batch_size // batch_gen):
            z, c, camera_params = next(batches)
            img = G(z, c, camera_params, **G_kwargs)
            images.append((img * 127.5 + 128).clamp(0, 255).to(torch.uint8))
        images = torch.cat(images)
        if images.shape[1] == 1:
            images = images.repeat([1, 3, 1, 1])
        stats.append_torch(detector(images), num_gpus=num_gpus, rank=rank, gatherer=gatherer)
    return stats


def compute_flattened_depth_maps(G, max_items, batch_size=64, batch_gen=None, camera_cfg=None, c_sampler=None, num_gpus=1, rank=0, device='cuda',
                                 gatherer=None, G_kwargs=None, dataset=None, cut_quantile=0.0):
    """metric_utils.py:324-349: frontal-camera depth maps of `max_items` generated samples, flattened to [max_items, h*w]
    (input of the reference's non-flatness score)."""
    batch_gen = resolve_batch_gen(batch_size, batch_gen)
    G_kwargs = {} if G_kwargs is None else G_kwargs
    stats = FeatureStats(max_items=max_items, capture_all=True)
    batches = _generator_batches(G, batch_gen, camera_cfg, c_sampler, dataset, device, frontal_camera=True)
    while not stats.is_full():
        depths = []
        for _ in range(batch_size // batch_gen):
            z, c, camera_params = next(batches)
            out = G(z, c, camera_params, render_opts=dict(return_depth=True, cut_quantile=cut_quantile), **G_kwargs)
            depths.append(out.depth)
        stats.append_torch(torch.cat(depths).flatten(start_dim=1), num_gpus=num_gpus, rank=rank, gatherer=gatherer)
    return torch.from_numpy(stats.get_all())
