"""Inference harness around the generator forward.

Reference: `src/training/inference_utils.py:88-215` -- `generate`, `generate_trajectory`, `generate_camera_trajectory`,
`approximate_mean_camera_params`, `sample_posterior_camera_params`.  Host-side orchestration only: every frame is one
`G.synthesis` call on the HIP path; trajectories are a few hundred floats of tensor arithmetic kept on the CPU like the reference.
"""
import numpy as np
import torch

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
        frames.append(frame.clamp(-1, 1).cpu() * 0.5 + 0.5)
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
