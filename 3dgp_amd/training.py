"""Training-side glue around the differentiable generator and the discriminator (last rows).

Reference: `src/training/loss.py:33-330` (StyleGAN2Loss: run_G, run_D, accumulate_gradients for the phases Gmain / Gall / Dmain /
Dreg / Dall, maybe_blur), `src/training/training_utils.py:22-167` (patch sampling and extraction), `training_loop.py:325-347`
(the optimiser step around the gradient exchange).  What is here is the adversarial core every 3dgp run executes: non-saturating
(or hinge) losses, R1 on real patches, patch-wise training with patch-conditioned discriminator, RGB-D discriminator input through
the depth adaptor, image / depth blur schedules.  Not here: the camera-adaptor regularisers (Lipschitz, EMD -- the latter needs the
POT solver), knowledge distillation, path-length regularisation (`pl_weight: 0` in every 3dgp config), ADA.

All device work runs on the library's kernels through the autograd ops (`G.forward_autograd`, `discriminator.Discriminator`);
the arithmetic in this file is the reference's eager tensor arithmetic.
"""
from dataclasses import dataclass

import numpy as np
import torch

from .adaptors import linear_schedule
from .generator import TensorGroup
from .ops import upfirdn2d as _upfirdn2d


@dataclass
class PatchConfig:
    """configs/training/base.yaml:33-43 + patch_{beta,uniform}.yaml."""
    enabled: bool = True
    distribution: str = 'beta'
    resolution: int = 64
    min_scale_trg: float = 0.25            # patch resolution / dataset resolution
    max_scale: float = 1.0
    anneal_kimg: float = 10000
    alpha: float = 1.0
    beta_val_start: float = 0.001
    beta_val_end: float = 0.8
    mbstd_group_size: int = 4
    min_scale: float = 1.0                 # set by progressive_update
    beta: float = 0.001


# ----------------------------------------------------------------------------------------------------------------------
# patches, training_utils.py:22-167
# ----------------------------------------------------------------------------------------------------------------------
def create_patch_params_from_x_scales(patch_scales_x, group_size=1, device='cpu'):
    """training_utils.py:128-143: square patches, offsets uniform in [0, 1 - scale], one draw per minibatch-stddev group."""
    sx = torch.from_numpy(np.asarray(patch_scales_x)).float().to(device)
    scales = torch.stack([sx, sx], dim=1)
    offsets = torch.rand(scales.shape, device=device) * (1.0 - scales)
    return {'scales': scales.repeat_interleave(group_size, dim=0), 'offsets': offsets.repeat_interleave(group_size, dim=0)}


def sample_patch_params(batch_size, patch_cfg, device='cpu'):
    """training_utils.py:57-124 ('uniform' and 'beta' distributions; numpy RNG for the scales, torch RNG for the offsets)."""
    assert patch_cfg.max_scale <= 1.0 and patch_cfg.min_scale <= patch_cfg.max_scale
    groups = batch_size // patch_cfg.mbstd_group_size
    span = patch_cfg.max_scale - patch_cfg.min_scale
    if patch_cfg.distribution == 'uniform':
        sx = np.random.rand(groups) * span + patch_cfg.min_scale
    elif patch_cfg.distribution == 'beta':
        sx = np.random.beta(a=patch_cfg.alpha, b=patch_cfg.beta, size=groups) * span + patch_cfg.min_scale
    else:
        raise NotImplementedError(f'Unkown patch sampling distrubtion: {patch_cfg.distribution}')
    return create_patch_params_from_x_scales(sx, patch_cfg.mbstd_group_size, device=device)


def generate_coords(batch_size, img_size, device='cpu', align_corners=False):
    """training_utils.py:147-167: [-1,1] pixel coordinates, y flipped to the image memory layout."""
    row = torch.linspace(-1, 1, img_size, device=device).float() if align_corners else (torch.arange(0, img_size, device=device).float() / img_size) * 2 - 1
    x = row.view(1, -1).repeat(img_size, 1)
    coords = torch.stack([x, -x.t()], dim=2).view(-1, 2)
    return coords.t().view(1, 2, img_size, img_size).repeat(batch_size, 1, 1, 1).permute(0, 2, 3, 1)


def compute_patch_coords(patch_params, resolution, align_corners=True, for_grid_sample=True):
    """training_utils.py:35-54."""
    scales, offsets = patch_params['scales'], patch_params['offsets']
    B = scales.shape[0]
    coords = generate_coords(B, resolution, device=scales.device, align_corners=align_corners)
    coords = (coords + 1.0) * scales.view(B, 1, 1, 2) - 1.0 + offsets.view(B, 1, 1, 2) * 2.0
    if for_grid_sample:
        coords[:, :, :, 1] = -coords[:, :, :, 1]
    return coords


def extract_patches(x, patch_params, resolution):
    """training_utils.py:22-31: bilinear crop of the real images (data side; eager grid_sample as in the reference)."""
    assert x.shape[2] == x.shape[3], 'Can only work on square images (for now)'
    return torch.nn.functional.grid_sample(x, compute_patch_coords(patch_params, resolution), mode='bilinear', align_corners=True)


def maybe_blur(img, blur_sigma):
    """loss.py:332-338: Gaussian blur through upfirdn2d.filter2d (the HIP kernel on the GPU)."""
    blur_size = np.floor(blur_sigma * 3)
    if blur_size > 0:
        f = torch.arange(-blur_size, blur_size + 1, device=img.device).div(blur_sigma).square().neg().exp2()
        img = _upfirdn2d.filter2d(img, f / f.sum())
    return img


# ----------------------------------------------------------------------------------------------------------------------
# loss.py:33-330
# ----------------------------------------------------------------------------------------------------------------------
class StyleGAN2Loss:
    def __init__(self, G, D, device, r1_gamma=10.0, patch_cfg=None, use_depth=False, adv_loss_type='non_saturating', blur_init_sigma=0, blur_fade_kimg=0,
                 blur_real_depth_sigma=0.0, logits_clamp_val=1e7, learn_camera_dist=False, synthesis_kwargs=None):
        if learn_camera_dist and getattr(G.synthesis, 'camera_adaptor', None) is None:
            raise RuntimeError('learn_camera_dist=True needs a generator built with cfg.camera_adaptor')
        self.G, self.D, self.device = G, D, device
        self.r1_gamma, self.use_depth, self.adv_loss_type = r1_gamma, use_depth, adv_loss_type
        self.blur_init_sigma, self.blur_fade_kimg, self.blur_real_depth_sigma = blur_init_sigma, blur_fade_kimg, blur_real_depth_sigma
        self.logits_clamp_val, self.learn_camera_dist = logits_clamp_val, learn_camera_dist
        self.patch_cfg = patch_cfg if patch_cfg is not None else PatchConfig(enabled=False)
        self.synthesis_kwargs = dict(synthesis_kwargs or {})          # e.g. explicit renderer draws for the parity tests
        self.stats = {}
        self.progressive_update(0)

    def progressive_update(self, cur_kimg):
        """loss.py:53-69 (patch-scale schedule)."""
        p = self.patch_cfg
        if p.enabled:
            if p.distribution == 'uniform':
                p.min_scale = linear_schedule(cur_kimg, p.max_scale, p.min_scale_trg, p.anneal_kimg)
            elif p.distribution == 'beta':
                p.beta = linear_schedule(cur_kimg, p.beta_val_start, p.beta_val_end, p.anneal_kimg)
                p.min_scale = p.min_scale_trg
            else:
                raise NotImplementedError(f'Uknown patch distribution: {p.distribution}')

    def run_G(self, z, c, camera_params, update_emas=False):
        """loss.py:71-86 (style mixing is off in every 3dgp config)."""
        ws = self.G.mapping(z, c, update_emas=update_emas)
        patch_params = sample_patch_params(len(z), self.patch_cfg, device=z.device) if self.patch_cfg.enabled else {}
        patch_kwargs = dict(patch_params=patch_params) if self.patch_cfg.enabled else {}
        if self.learn_camera_dist:                 # loss.py:76-77: the adaptor is trained through ray generation and the sample positions
            camera_params = self.G.synthesis.camera_adaptor(camera_params, z, c)
        out = self.G.synthesis.forward_autograd(ws, camera_params, render_opts=dict(concat_depth=self.use_depth, return_depth=True), **patch_kwargs,
                                                **self.synthesis_kwargs)
        out.ws = ws
        return out, patch_params, camera_params

    def run_D(self, img, c, blur_sigma=0, update_emas=False, **kwargs):
        """loss.py:88-104 (no ADA pipe)."""
        img = maybe_blur(img, blur_sigma)
        assert img.shape[1] == 4 or not self.use_depth, f'Wrong shape: {img.shape}'
        if self.use_depth:
            blur_size = np.floor(blur_sigma * 3)
            f = torch.arange(-blur_size, blur_size + 1, device=img.device).div(30.0).square().neg().exp2()
            img = torch.cat([img[:, :3], _upfirdn2d.filter2d(img[:, [3]], f / f.sum()), img[:, 4:]], dim=1)
        kwargs.pop('camera_angles', None)                               # camera_cond is off in every 3dgp config
        return self.D(img, c, update_emas=update_emas, **kwargs)

    def extract_patches(self, img):
        patch_params = sample_patch_params(len(img), self.patch_cfg, device=img.device)
        return extract_patches(img, patch_params, resolution=self.patch_cfg.resolution), patch_params

    def _g_loss(self, logits):
        if self.adv_loss_type == 'non_saturating':
            return torch.nn.functional.softplus(-logits)
        if self.adv_loss_type == 'hinge':
            return -logits
        raise NotImplementedError(f'Unknown loss: {self.adv_loss_type}')

    def accumulate_gradients(self, phase, real_data, gen_data, gain, cur_nimg):
        """loss.py:117-330.  real_data: TensorGroup(img, c, depth?), gen_data: TensorGroup(z, c, camera_params).  Gradients are
        accumulated into `.grad` of whichever module has requires_grad (the caller toggles that per phase, training_loop.py:328-331)."""
        assert phase in ['Gmain', 'Greg_pl', 'Gall', 'Dmain', 'Dreg', 'Dall']
        if self.r1_gamma == 0:
            phase = {'Dreg': 'none', 'Dall': 'Dmain'}.get(phase, phase)
        blur_sigma = max(1 - cur_nimg / (self.blur_fade_kimg * 1e3), 0) * self.blur_init_sigma if self.blur_fade_kimg > 0 else 0
        real_img = real_data.img
        if self.use_depth:
            real_img = torch.cat([real_img, maybe_blur(real_data.depth, self.blur_real_depth_sigma)], dim=1)

        if phase in ['Gmain', 'Gall']:                                    # maximise logits of generated images
            gen_out, patch_params, cam = self.run_G(gen_data.z, gen_data.c, gen_data.camera_params)
            gen_logits, _ = self.run_D(gen_out.img, gen_data.c, blur_sigma=blur_sigma, patch_params=patch_params)
            loss_Gmain = self._g_loss(gen_logits)
            self.stats['Loss/G/loss'] = loss_Gmain.detach()
            self.stats['Loss/scores/fake'] = gen_logits.detach()
            loss_Gmain.mean().mul(gain).backward()

        loss_Dgen = 0
        if phase in ['Dmain', 'Dall']:                                    # minimise logits of generated images
            with torch.no_grad():
                gen_out, patch_params, cam = self.run_G(gen_data.z, gen_data.c, gen_data.camera_params, update_emas=True)
            gen_logits, _ = self.run_D(gen_out.img, gen_data.c, blur_sigma=blur_sigma, update_emas=True, patch_params=patch_params)
            if self.adv_loss_type == 'non_saturating':
                loss_Dgen = torch.nn.functional.softplus(gen_logits.clamp(min=-self.logits_clamp_val, max=None))
                loss_Dgen = loss_Dgen + 0.0 * gen_logits.max()
            elif self.adv_loss_type == 'hinge':
                loss_Dgen = torch.nn.functional.relu(1.0 + gen_logits)
            else:
                raise NotImplementedError(f'Unknown loss: {self.adv_loss_type}')
            self.stats['Loss/scores/fake'] = gen_logits.detach()
            loss_Dgen.mean().mul(gain).backward()

        if phase in ['Dmain', 'Dreg', 'Dall']:                            # maximise logits of real images, R1
            real_img, patch_params = self.extract_patches(real_img) if self.patch_cfg.enabled else (real_img, None)
            real_img_tmp = real_img.detach().requires_grad_(phase in ['Dreg', 'Dall'])
            real_logits, _ = self.run_D(real_img_tmp, real_data.c, blur_sigma=blur_sigma, patch_params=patch_params)
            self.stats['Loss/scores/real'] = real_logits.detach()
            loss_Dreal = 0.0
            if phase in ['Dmain', 'Dall']:
                if self.adv_loss_type == 'non_saturating':
                    loss_Dreal = torch.nn.functional.softplus(-real_logits.clamp(min=None, max=self.logits_clamp_val))
                    loss_Dreal = loss_Dreal + 0.0 * real_logits.max()
                elif self.adv_loss_type == 'hinge':
                    loss_Dreal = torch.nn.functional.relu(1.0 - real_logits)
                else:
                    raise NotImplementedError(f'Unknown loss: {self.adv_loss_type}')
                self.stats['Loss/D/loss'] = (loss_Dgen + loss_Dreal).detach()
            loss_Dr1 = 0.0
            if phase in ['Dreg', 'Dall']:
                from .ops import conv2d_gradfix as _cg
                with _cg.no_weight_gradients():
                    r1_grads, = torch.autograd.grad(outputs=[real_logits.sum()], inputs=[real_img_tmp], create_graph=True, only_inputs=True)
                r1_penalty = r1_grads.square().sum([1, 2, 3])
                loss_Dr1 = r1_penalty * (self.r1_gamma / 2)
                self.stats['Loss/D/r1_penalty'] = r1_penalty.detach()
            (loss_Dreal + loss_Dr1).mean().mul(gain).backward()


def optimizer_step(module, opt, world=None, grad_clip=None):
    """training_loop.py:334-347: exchange (one flat all-reduce), sanitise, optional clipping, optimiser step."""
    from . import distributed as _dist
    params = [p for p in module.parameters() if p.grad is not None]
    _dist.allreduce_gradients(params, world=world)
    if grad_clip is not None:
        torch.nn.utils.clip_grad_norm_(params, grad_clip)
    opt.step()


# ----------------------------------------------------------------------------------------------------------------------
# training_loop.py:170-210, 319-372: the pieces of the driver that touch the device
# ----------------------------------------------------------------------------------------------------------------------
def broadcast_module(module, src=0):
    """training_loop.py:173-177: every rank starts from rank 0's parameters and buffers (RCCL broadcast, one tensor at a time)."""
    import torch.distributed as dist
    if module is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


def setup_phases(G, D, G_opt_kwargs, D_opt_kwargs, G_reg_interval=None, D_reg_interval=16, pl_weight=0.0, opt_class=torch.optim.Adam):
    """training_loop.py:180-205: one optimiser per network; with lazy regularisation (reg_interval > 0) the learning rate and the Adam
    betas are rescaled by mb_ratio = interval / (interval + 1) and the regulariser gets its own phase every `interval` batches.
    -> list of dict(name, module, opt, interval)."""
    phases = []
    for name, module, opt_kwargs, reg_interval in (('G', G, G_opt_kwargs, G_reg_interval), ('D', D, D_opt_kwargs, D_reg_interval)):
        if reg_interval in (None, 0):
            phases.append(dict(name=name + 'all', module=module, opt=opt_class(module.parameters(), **opt_kwargs), interval=1))
            continue
        mb_ratio = reg_interval / (reg_interval + 1)
        kw = dict(opt_kwargs)
        kw['lr'] = kw['lr'] * mb_ratio
        kw['betas'] = [beta ** mb_ratio for beta in kw['betas']]
        opt = opt_class(module.parameters(), **kw)
        phases.append(dict(name=name + 'main', module=module, opt=opt, interval=1))
        if name == 'G':
            assert pl_weight > 0, 'a G regularisation phase needs the path-length regulariser'
            phases.append(dict(name='Greg_pl', module=module, opt=opt, interval=reg_interval))
        else:
            phases.append(dict(name='Dreg', module=module, opt=opt, interval=reg_interval))
    return phases


@torch.no_grad()
def update_ema(G_ema, G, cur_nimg, batch_size, ema_kimg=10.0, ema_rampup=0.05, ema_start_kimg=0.0):
    """training_loop.py:357-367: half-life of `ema_kimg` thousand images, ramped up with the number of images seen; buffers copied."""
    ema_nimg = ema_kimg * 1000
    if ema_rampup is not None:
        ema_nimg = min(ema_nimg, cur_nimg * ema_rampup)
    ema_beta = 0.5 ** (batch_size / max(ema_nimg, 1e-8))
    if ema_start_kimg > cur_nimg / 1000:
        ema_beta = 0.0
    for p_ema, p in zip(G_ema.parameters(), G.parameters()):
        p_ema.copy_(p.lerp(p_ema, ema_beta))
    for b_ema, b in zip(G_ema.buffers(), G.buffers()):
        b_ema.copy_(b)
    return ema_beta


def _split(group, n):
    """TensorGroup.split(n) of the reference: per-key torch.split, regrouped."""
    keys = list(group.keys())
    parts = {k: (group[k].split(n) if isinstance(group[k], TensorGroup) else torch.split(group[k], n)) for k in keys}
    count = len(next(iter(parts.values())))
    return [TensorGroup(**{k: parts[k][i] for k in keys}) for i in range(count)]


def train_iteration(loss, phases, real_data, all_gen_data, batch_idx, cur_nimg, batch_size, batch_gpu, world=None, grad_clip=None):
    """training_loop.py:319-347: every phase whose interval divides `batch_idx` zeroes its gradients, accumulates them over the
    rank's sub-batches of `batch_gpu`, exchanges them (flat all-reduce) and steps.  `all_gen_data` holds `len(phases) * batch_size`
    latent / camera samples, one `batch_size` slice per phase.  Returns the names of the phases that ran."""
    ran = []
    for phase, gen_data in zip(phases, _split(all_gen_data, batch_size)):
        if batch_idx % phase['interval'] != 0:
            continue
        phase['opt'].zero_grad(set_to_none=True)
        phase['module'].requires_grad_(True)
        for r, g in zip(_split(real_data, batch_gpu), _split(gen_data, batch_gpu)):
            loss.accumulate_gradients(phase=phase['name'], real_data=r, gen_data=g, gain=phase['interval'], cur_nimg=cur_nimg)
        phase['module'].requires_grad_(False)
        clip = grad_clip if phase['name'] in ('Gmain', 'Gall', 'Greg_pl') else None
        optimizer_step(phase['module'], phase['opt'], world=world, grad_clip=clip)
        ran.append(phase['name'])
    return ran
