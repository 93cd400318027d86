"""Training-side glue around the differentiable generator and the discriminator (last rows).

Reference: `src/training/loss.py:33-330` (StyleGAN2Loss: run_G, run_D, accumulate_gradients for the phases Gmain / Gall / Dmain /
Dreg / Dall, maybe_blur), `src/training/training_utils.py:22-167` (patch sampling and extraction), `training_loop.py:325-347`
(the optimiser step around the gradient exchange).  What is here is the adversarial core every 3dgp run executes: non-saturating
(or hinge) losses, R1 on real patches, patch-wise training with patch-conditioned discriminator, RGB-D discriminator input through
the depth adaptor, image / depth blur schedules, and the three camera-adaptor regularisers of `learn_camera_dist` (Lipschitz, earth
mover's distance to the prior, force-mean; loss.py:142-222), and the discriminator's knowledge-distillation term (features predicted
from real images pulled to the dataset's precomputed embeddings, loss.py:279-314).  Not here: path-length regularisation (`pl_weight: 0`
in every 3dgp config), ADA.

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


@dataclass
class CameraRegConfig:
    """configs/model/3dgp.yaml:55-63,75 (`camera_adaptor.lipschitz_weights`, `.emd`, `.force_mean_weight`) with the yaml's defaults.
    `prior` is the `camera:` config node the prior is drawn from (dict in the reference's layout, see metrics.sample_camera_params),
    or a callable (num_samples, device) -> camera parameters for tests."""
    prior: object = None
    lipschitz_enabled: bool = False
    lipschitz_angles: float = 1.0
    lipschitz_radius: float = 1.0
    lipschitz_fov: float = 1.0
    lipschitz_look_at: float = 1.0
    lipschitz_num_samples: int = 256       # loss.py:146 (hard-coded there)
    emd_enabled: bool = True
    emd_anneal_kimg: float = 10000
    emd_num_samples: int = 64
    emd_origin: float = 2.0
    emd_radius: float = 0.0
    emd_fov: float = 0.0001
    emd_look_at: float = 0.0001
    force_mean_weight: float = 10.0
    force_mean_num_samples: int = 256      # loss.py:228
    mean_angles: object = None             # [yaw, pitch, roll] the force-mean term pulls to; None = analytic mean of `prior`

    @classmethod
    def disabled(cls):
        """'No regularisers' said out loud: the adversarial loss alone reaches the camera adaptor.  The reference has no such mode by
        default (3dgp.yaml: emd.enabled, force_mean_weight 10) -- `StyleGAN2Loss(learn_camera_dist=True)` therefore refuses
        `camera_reg=None` and takes this instead when a caller really wants it."""
        return cls(prior=None, lipschitz_enabled=False, emd_enabled=False, force_mean_weight=0.0)

    @property
    def any_enabled(self):
        return bool(self.lipschitz_enabled or self.emd_enabled or self.force_mean_weight > 0)


# ----------------------------------------------------------------------------------------------------------------------
# camera-adaptor regularisers, loss.py:142-238
# ----------------------------------------------------------------------------------------------------------------------
def sample_random_c(batch_size, c_dim, device):
    """training_utils.py:207-214: uniformly random one-hot labels."""
    c = torch.zeros(batch_size, c_dim, device=device)
    if c_dim > 0:
        c[torch.arange(batch_size), torch.randint(low=0, high=c_dim, device=device, size=(batch_size,))] = 1.0
    return c


def emd2_1d(a, b):
    """Squared-Euclidean earth mover's distance between two equally sized, uniformly weighted 1-D samples.

    The reference calls POT (`ot.dist` -> squared distances, `ot.emd2` with weights 1/n; loss.py:195-197; POT is not vendored with
    it).  For a convex cost on the line the optimal plan of that linear programme is the monotone matching, so the value is
    mean((sort(a) - sort(b))^2) and -- `emd2` back-propagates through the cost matrix with the plan held fixed -- so are the
    gradients.  tests/test_training.py checks both against the assignment-problem solution of the same cost matrix."""
    assert a.ndim == 1 and a.shape == b.shape
    return (a.sort().values - b.sort().values).square().mean()


def _weigh_camera_regs(adaptor, regs, origin, radius, fov, look_at):
    """loss.py:170-175 / :208-214: per-component weights; the roll angle does not count."""
    r = adaptor.roll_camera_params(regs)
    return (r.angles[:, :2] * origin).sum() + (r.radius * radius).sum() + (r.fov * fov).sum() + (r.look_at * look_at).sum()


def _prior_and_posterior(adaptor, prior, num_samples, z_dim, c_dim, device, z=None, c=None):
    """loss.py:146-156: draw z, c and prior cameras, make the raw [N, 8] prior a leaf and push it through the adaptor."""
    from .metrics import sample_camera_params
    z = torch.randn(num_samples, z_dim, device=device) if z is None else z
    c = sample_random_c(num_samples, c_dim, device) if c is None else c
    cp = prior(num_samples, device) if callable(prior) else sample_camera_params(prior, num_samples, device=device)
    prior_raw = adaptor.unroll_camera_params(cp).detach().requires_grad_(True)
    posterior_raw = adaptor.unroll_camera_params(adaptor(adaptor.roll_camera_params(prior_raw), z, c))
    return prior_raw, posterior_raw


def camera_lipschitz_regs(adaptor, prior_raw, posterior_raw):
    """loss.py:157-159: g_i = |d posterior_i / d prior_i| per sample (diagonal of the adaptor's Jacobian, one autograd pass per
    component, graph kept: the regulariser is trained through the second derivative), reg_i = mean(g_i + 1 / (g_i + 1e-4)).  [1, 8]"""
    grads = [torch.autograd.grad(outputs=[posterior_raw[:, i].sum()], inputs=[prior_raw], create_graph=True, only_inputs=True)[0][:, i]
             for i in range(posterior_raw.shape[1])]
    g = torch.stack(grads, dim=1).abs()
    return (g + 1.0 / (g + 1e-4)).mean(dim=0, keepdim=True)


def camera_emd_regs(prior_raw, posterior_raw):
    """loss.py:195-198: one 1-D transport problem per camera component.  [1, 8]"""
    return torch.stack([emd2_1d(posterior_raw[:, i], prior_raw[:, i]) for i in range(posterior_raw.shape[1])]).unsqueeze(0)


def mean_angles_of_prior(angles_cfg):
    """rendering_utils.py:180-190."""
    from .metrics import _g
    dist = _g(angles_cfg, 'dist')
    if dist in ('spherical_uniform', 'truncnorm', 'uniform'):
        return [(_g(angles_cfg, 'yaw.max') + _g(angles_cfg, 'yaw.min')) * 0.5, (_g(angles_cfg, 'pitch.max') + _g(angles_cfg, 'pitch.min')) * 0.5, 0.0]
    if dist == 'normal':
        return [_g(angles_cfg, 'yaw.mean'), _g(angles_cfg, 'pitch.mean'), 0.0]
    if dist == 'custom':
        raise ValueError('Cannot compute the mean value analytically for a custom angles distribution.')
    raise NotImplementedError(f'Uknown distribution: `{dist}`')


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
                 blur_real_depth_sigma=0.0, logits_clamp_val=1e7, learn_camera_dist=False, camera_reg=None, kd_weight=0.0, kd_anneal_kimg=100000,
                 kd_loss_type='l2', synthesis_kwargs=None):
        if learn_camera_dist and getattr(G.synthesis, 'camera_adaptor', None) is None:
            raise RuntimeError('learn_camera_dist=True needs a generator built with cfg.camera_adaptor')
        self.G, self.D, self.device = G, D, device
        self.r1_gamma, self.use_depth, self.adv_loss_type = r1_gamma, use_depth, adv_loss_type
        self.blur_init_sigma, self.blur_fade_kimg, self.blur_real_depth_sigma = blur_init_sigma, blur_fade_kimg, blur_real_depth_sigma
        self.logits_clamp_val, self.learn_camera_dist = logits_clamp_val, learn_camera_dist
        self.patch_cfg = patch_cfg if patch_cfg is not None else PatchConfig(enabled=False)
        self.synthesis_kwargs = dict(synthesis_kwargs or {})          # e.g. explicit renderer draws for the parity tests
        self.kd_weight, self.kd_anneal_kimg, self.kd_loss_type = kd_weight, kd_anneal_kimg, kd_loss_type       # configs/model/base.yaml:82 (3dgp.yaml:91: weight 1)
        # The reference ALWAYS adds the EMD and force-mean terms when the camera distribution is learned (loss.py:186-238 under the 3dgp.yaml
        # defaults): silently training the adaptor through the adversarial loss alone would be a different model, so the
        # regularisers are required -- or switched off explicitly with CameraRegConfig.disabled().
        if learn_camera_dist and camera_reg is None:
            raise RuntimeError('learn_camera_dist=True needs camera_reg: CameraRegConfig(prior=<camera config node>) for the reference\'s regularisers '
                               '(loss.py:142-238), or CameraRegConfig.disabled() to train the adaptor through the adversarial loss alone')
        self.camera_reg = camera_reg if learn_camera_dist else None
        if self.camera_reg is not None and self.camera_reg.any_enabled and self.camera_reg.prior is None:
            raise RuntimeError('camera_reg needs the camera prior (CameraRegConfig.prior)')
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
        self.D_kd_weight = linear_schedule(cur_kimg, self.kd_weight, 0.0, self.kd_anneal_kimg)     # loss.py:63: fades OUT
        r = self.camera_reg                        # loss.py:64-67: the EMD term fades IN over emd.anneal_kimg
        self.emd_multiplier = linear_schedule(cur_kimg, 0.0, 1.0, r.emd_anneal_kimg) if r is not None else 0.0

    def camera_regularisers(self):
        """loss.py:142-238: the three terms added to the Gmain loss when the camera distribution is learned.  Returns a scalar
        tensor (0.0 when nothing is enabled); each term draws its own z / c / prior cameras, in the reference's order."""
        r, G = self.camera_reg, self.G
        total = 0.0
        if r is None:
            return total
        A = G.synthesis.camera_adaptor
        if r.lipschitz_enabled:
            prior_raw, post_raw = _prior_and_posterior(A, r.prior, r.lipschitz_num_samples, G.z_dim, G.c_dim, self.device)
            regs = camera_lipschitz_regs(A, prior_raw, post_raw)
            self._report_camera_regs('Dist_lipschitz_reg', A.roll_camera_params(regs))
            total = total + _weigh_camera_regs(A, regs + regs.max() * 0.0, r.lipschitz_angles, r.lipschitz_radius, r.lipschitz_fov, r.lipschitz_look_at)
        if r.emd_enabled and self.emd_multiplier > 0.0:
            prior_raw, post_raw = _prior_and_posterior(A, r.prior, r.emd_num_samples, G.z_dim, G.c_dim, self.device)
            regs = camera_emd_regs(prior_raw, post_raw)
            self._report_camera_regs('Dist_emd_reg', A.roll_camera_params(regs))
            emd = self.emd_multiplier * _weigh_camera_regs(A, regs + regs.max() * 0.0, r.emd_origin, r.emd_radius, r.emd_fov, r.emd_look_at)
            self.stats['Loss/camera_dist/emd_loss'] = emd.detach()
            total = total + emd
        if A.cfg.adjust_angles and r.force_mean_weight > 0:
            from .metrics import _g
            mean_angles = torch.tensor(r.mean_angles if r.mean_angles is not None else mean_angles_of_prior(_g(r.prior, 'origin.angles'))).to(self.device)
            _, post_raw = _prior_and_posterior(A, r.prior, r.force_mean_num_samples, G.z_dim, G.c_dim, self.device)
            raw = (post_raw[:, :3].mean(dim=0) - mean_angles + 1e-8).square().sum().sqrt()
            self.stats['Loss/camera_dist/force_mean'] = (r.force_mean_weight * raw).detach()
            total = total + r.force_mean_weight * raw + 0.0 * post_raw.max()
        return total

    def _report_camera_regs(self, prefix, regs):
        for name, v in (('yaw', regs.angles[:, 0]), ('pitch', regs.angles[:, 1]), ('fov', regs.fov), ('radius', regs.radius),
                        ('look_at_yaw', regs.look_at[:, 0]), ('look_at_pitch', regs.look_at[:, 1]), ('look_at_radius', regs.look_at[:, 2])):
            self.stats[f'{prefix}/{name}'] = v.detach()

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

    def compute_sample_weights(self, patch_params, scale_pow=1):
        """loss.py:107-114: larger patches weigh more in the distillation distance."""
        if not self.patch_cfg.enabled:
            return 1.0
        raw = patch_params['scales'].mean(dim=1) ** scale_pow
        return raw / (raw.mean(dim=0) + 1e-8)

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
            (loss_Gmain + self.camera_regularisers()).mean().mul(gain).backward()     # loss.py:240-241

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
            do_Dkd = self.D_kd_weight > 0 and phase in ['Dmain', 'Dall']
            real_logits, real_feats = self.run_D(real_img_tmp, real_data.c, blur_sigma=blur_sigma, patch_params=patch_params, predict_feat=do_Dkd)
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
            loss_Dkd = 0.0
            if do_Dkd:                                                     # loss.py:301-311: distance of the predicted features to the embeddings
                if self.kd_loss_type == 'l2':
                    distances = (real_feats - real_data.embs).norm(dim=1)
                elif self.kd_loss_type == 'kl':
                    distances = torch.nn.functional.kl_div(real_feats.log_softmax(dim=1), real_data.embs.softmax(dim=1), reduction='none').sum(dim=1)
                else:
                    raise NotImplementedError(f'Unknown loss type: {self.kd_loss_type}')
                distances = distances * self.compute_sample_weights(patch_params)
                loss_Dkd = distances * self.D_kd_weight
                self.stats['Loss/kd/D_dist'] = distances.detach()
                self.stats['Loss/kd/D_loss'] = loss_Dkd.detach()
            else:
                assert real_feats is None, 'There is no sense in predicting features from D'
            loss_Dr1 = 0.0
            if phase in ['Dreg', 'Dall']:
                from .ops import conv2d_gradfix as _cg
                with _cg.no_weight_gradients():
                    r1_grads, = torch.autograd.grad(outputs=[real_logits.sum()], inputs=[real_img_tmp], create_graph=True, only_inputs=True)
                r1_penalty = r1_grads.square().sum([1, 2, 3])
                loss_Dr1 = r1_penalty * (self.r1_gamma / 2)
                self.stats['Loss/D/r1_penalty'] = r1_penalty.detach()
            (loss_Dreal + loss_Dr1 + loss_Dkd).mean().mul(gain).backward()


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
