"""Inference harness around the generator forward.

Reference: `src/training/inference_utils.py:88-215` -- `generate`, `generate_trajectory`, `generate_camera_trajectory`,
`approximate_mean_camera_params`, `sample_posterior_camera_params`; `scripts/inference.py:87-150` -- `sample_z_from_seeds`,
`sample_c_from_seeds`, `c_idx_to_c`, `sample_ws_from_seeds`.  Host-side orchestration only: every frame is one
`G.synthesis` call on the HIP path; trajectories are a few hundred floats of tensor arithmetic kept on the CPU like the reference.
"""
import numpy as np
import torch

from . import _lib
from .generator import TensorGroup
from .metrics import camera_base, sample_camera_params


def _tg(cfg, name, default=None):
    return cfg.get(name, default) if isinstance(cfg, dict) else getattr(cfg, name, default)


def generate(G, ws, camera_params, batch_size=8, **synthesis_kwargs):
    """inference_utils.py:107-126: frames for (ws[i], camera[i]) in chunks of `batch_size`, `noise_mode='const'`, mapped to
    [0, 1] on the CPU; depth (when requested) normalised to [-1, 1] by the ray range first."""
    frames = []
    for b0 in range(0, len(ws), batch_size):
        sl = slice(b0, b0 + batch_size)
        frame = G.synthesis(ws[sl], camera_params=camera_params[sl], noise_mode='const', **synthesis_kwargs)
        if isinstance(frame, TensorGroup) and 'depth' in frame:
            depth_range = G.cfg.ray_end - G.cfg.ray_start
            depth_mid = (G.cfg.ray_start + G.cfg.ray_end) * 0.5
            frame.depth = (frame.depth - depth_mid) / depth_range * 2.0
        frames.append(frame.clamp(-1, 1).cpu() * 0.5 + 0.5)             # a synchronisation point
        if ws.is_cuda:
            _lib.raise_on_device_fault('generate()')
    return TensorGroup.cat(frames, dim=0) if isinstance(frames[0], TensorGroup) else torch.cat(frames, dim=0)


def generate_trajectory(G, ws, camera_params, **generate_kwargs):
    """inference_utils.py:88-103: every `ws` under every camera of its trajectory -> [num_cameras, num_samples, c, h, w]."""
    num_cameras = len(camera_params) // len(ws)
    num_samples = len(camera_params) // num_cameras
    camera_params = camera_params.to(dtype=torch.float32, device=ws.device)
    ws = ws.repeat_interleave(num_cameras, dim=0)
    images = generate(G, ws=ws, camera_params=camera_params, **generate_kwargs)
    if isinstance(images, TensorGroup):
        images = images.reshape_each(lambda x: [num_samples, num_cameras, *x.shape[1:]])
    else:
        images = images.reshape(num_samples, num_cameras, *images.shape[1:])
    return images.permute(1, 0, 2, 3, 4)


def generate_camera_trajectory(trajectory, canonical_camera_params):
    """inference_utils.py:140-186: per canonical camera, the frames of a 'point' / 'front_circle' / 'points' / 'wiggle' / 'line'
    trajectory (all on the CPU); 'wiggle' raises, as it does in the reference."""
    name = _tg(trajectory, 'name')
    num_samples = len(canonical_camera_params)
    num_frames = len(_tg(trajectory, 'yaw_offsets')) if name == 'points' else _tg(trajectory, 'num_frames')
    cp = canonical_camera_params.repeat_interleave(num_frames, dim=0)
    if name == 'point':
        assert num_frames == 1
        angles = cp.angles.cpu() + torch.tensor([_tg(trajectory, 'yaw_offset'), _tg(trajectory, 'pitch_offset'), 0.0]).unsqueeze(0)
        fov = cp.fov.cpu() + _tg(trajectory, 'fov_offset')
    elif name == 'front_circle':
        steps = torch.linspace(0, 1, num_frames).repeat(num_samples)
        yaw = cp.angles[:, 0].cpu() + _tg(trajectory, 'yaw_diff') * torch.sin(steps * 2 * np.pi)
        pitch = cp.angles[:, 1].cpu() + _tg(trajectory, 'pitch_diff') * torch.cos(steps * 2 * np.pi)
        angles = torch.stack([yaw, pitch, cp.angles[:, 2].cpu()], dim=1)
        fov = cp.fov.cpu() + _tg(trajectory, 'fov_diff') * torch.sin(steps * 2 * np.pi)
    elif name == 'points':
        yaw = cp.angles[:, 0].cpu() + torch.tensor(_tg(trajectory, 'yaw_offsets')).repeat(num_samples)
        pitch = cp.angles[:, 1].cpu() + _tg(trajectory, 'pitch_offset')
        angles = torch.stack([yaw, pitch, cp.angles[:, 2].cpu()], dim=1)
        fov = cp.fov.cpu()
    elif name == 'wiggle':
        # inference_utils.py:167-170 builds numpy angles of length num_frames and fails TensorGroup's own type assertion
        raise NotImplementedError("the reference's 'wiggle' trajectory does not run (numpy angles in a TensorGroup, util.py:81)")
    elif name == 'line':
        yaws = torch.linspace(_tg(trajectory, 'yaw_start'), _tg(trajectory, 'yaw_end'), num_frames).repeat(num_samples)
        pitches = torch.linspace(_tg(trajectory, 'pitch_start'), _tg(trajectory, 'pitch_end'), num_frames).repeat(num_samples)
        angles = torch.stack([yaws, pitches, torch.zeros_like(yaws)], axis=1)
        fov = cp.fov.cpu() if _tg(trajectory, 'fov') is None else torch.ones_like(cp.fov.cpu()) * _tg(trajectory, 'fov')
    else:
        raise NotImplementedError(f'Unknown trajectory: {name}')
    return TensorGroup(angles=angles, fov=fov + _tg(trajectory, 'fov_offset', 0.0), radius=cp.radius.cpu(), look_at=cp.look_at.cpu())


def sample_posterior_camera_params(G, z, c, camera_cfg=None):
    """inference_utils.py:208-214: prior sample, passed through the camera adaptor when the generator has one."""
    prior = sample_camera_params(camera_base() if camera_cfg is None else camera_cfg, len(z), device=z.device)
    ca = getattr(G.synthesis, 'camera_adaptor', None)
    return prior if ca is None else ca(prior, z, c)


def approximate_mean_camera_params(G, num_samples=1024, device='cpu', camera_cfg=None, c_sampler=None):
    """inference_utils.py:196-204: Monte-Carlo mean of the (posterior) camera distribution, [1, ...]."""
    z = torch.randn(num_samples, G.z_dim, device=device)
    c = c_sampler(num_samples).to(device) if c_sampler is not None else torch.zeros(num_samples, G.c_dim, device=device)
    return sample_posterior_camera_params(G, z, c, camera_cfg).mean(dim=0, keepdim=True)


# ----------------------------------------------------------------------------------------------------------------------
# seeds -> (z, c) -> ws, scripts/inference.py:87-150
# ----------------------------------------------------------------------------------------------------------------------
def sample_z_from_seeds(seeds, z_dim):
    """scripts/inference.py:87-89: one `RandomState(seed).randn(1, z_dim)` row per seed (fp64 draw, cast to fp32)."""
    rows = [np.random.RandomState(s).randn(1, z_dim) for s in seeds]
    return torch.from_numpy(np.concatenate(rows, axis=0)).float()


def c_idx_to_c(c_idx, c_dim, device='cpu'):
    """scripts/inference.py:101-106: class indices -> one-hot rows [n, c_dim]."""
    idx = np.asarray(c_idx)
    c = np.zeros((len(idx), c_dim))
    c[np.arange(len(idx)), idx] = 1.0
    return torch.from_numpy(c).float().to(device)


def sample_c_from_seeds(seeds, c_dim, device='cpu'):
    """scripts/inference.py:93-97: the class of a seed is the first `choice` draw of its own RandomState."""
    if c_dim == 0:
        return torch.empty(len(seeds), 0)
    return c_idx_to_c([np.random.RandomState(s).choice(np.arange(c_dim), size=1).item() for s in seeds], c_dim, device)


def sample_ws_from_seeds(G, seeds, truncation_psi=1.0, device='cpu', num_interp_steps=0, classes=None, num_samples_to_avg=256):
    """scripts/inference.py:110-150 (`cfg.truncation_psi` passed as a number).

    num_interp_steps == 0: ws for every seed (x every class of `classes` when given).  With truncation_psi < 1 on a conditional
    generator the truncation centre is the PER-CLASS mean of `num_samples_to_avg` mapped samples (torch RNG on `device`), not
    `w_avg`.  Otherwise seeds are consumed pairwise and ws is `num_interp_steps` linear blends between the two ends."""
    if num_interp_steps == 0:
        z = sample_z_from_seeds(seeds, G.z_dim).to(device)
        c = sample_c_from_seeds(seeds, G.c_dim, device=device) if classes is None else c_idx_to_c(classes, G.c_dim, device)
        per_class_centre = truncation_psi < 1.0 and G.c_dim > 0
        if per_class_centre:
            z_avg = torch.randn(len(c) * num_samples_to_avg, G.z_dim, device=z.device)
            ws_avg = G.mapping(z_avg, c.repeat_interleave(num_samples_to_avg, dim=0))
            ws_avg = ws_avg.view(len(c), num_samples_to_avg, G.num_ws, ws_avg.shape[-1]).mean(dim=1)
            if classes is not None:
                ws_avg = ws_avg.repeat_interleave(len(seeds), dim=0)
        if classes is not None:                                            # class-major: every seed under class 0, then class 1, ...
            z = z.repeat(len(c), 1)
            c = c.repeat_interleave(len(seeds), dim=0)
        if per_class_centre:
            ws = G.mapping(z, c) * truncation_psi + ws_avg * (1 - truncation_psi)
        else:
            ws = G.mapping(z, c, truncation_psi=truncation_psi)
        return ws, z, c
    assert classes is None
    z_from, z_to = sample_z_from_seeds(seeds[0::2], G.z_dim).to(device), sample_z_from_seeds(seeds[1::2], G.z_dim).to(device)
    c_from, c_to = sample_c_from_seeds(seeds[0::2], G.c_dim, device=device), sample_c_from_seeds(seeds[1::2], G.c_dim, device=device)
    ws_from = G.mapping(z_from, c_from, truncation_psi=truncation_psi)
    ws_to = G.mapping(z_to, c_to, truncation_psi=truncation_psi)
    alpha = torch.linspace(0, 1, num_interp_steps, device=device).view(num_interp_steps, 1, 1, 1)
    return ws_from.unsqueeze(0) * (1 - alpha) + ws_to.unsqueeze(0) * alpha, (z_from, z_to), (c_from, c_to)
