"""Configuration of the generator-forward hot path.

Field names follow the keys the reference's hot path actually reads:
`configs/model/{base,3dgp}.yaml`, `configs/camera/base.yaml` and the constructor kwargs injected by
`src/train.py:156-273`.  Only what the generator forward consumes is kept.
"""
import math
from dataclasses import dataclass, field, asdict
from typing import Optional


@dataclass
class DepthAdaptorConfig:
    """configs/model/3dgp.yaml:40-50 (the keys DepthAdaptor.forward reads)."""
    kernel_size: int = 5
    hid_dim: int = 64
    num_hid_layers: int = 3
    out_strategy: str = 'random'
    near_plane_offset_max_fraction: float = 0.25
    near_plane_offset_bias: float = -3.0
    selection_start_p: float = 0.1      # training-time head selection (networks_depth_adaptor.py:64-66,86-92)
    anneal_kimg: float = 10000


@dataclass
class CameraRanges:
    """The entries of configs/camera/base.yaml the camera adaptor reads: (min, max) of each prior."""
    yaw: tuple = (-1.57079633, 1.57079633)
    pitch: tuple = (0.392699082, 2.74889357)
    fov: tuple = (10.0, 45.0)
    look_at_yaw: tuple = (-3.14159265, 3.14159265)
    look_at_pitch: tuple = (0.0, 3.14159265)
    look_at_radius: tuple = (0.0, 0.0)


@dataclass
class CameraAdaptorConfig:
    """configs/model/3dgp.yaml:52-75; z_dim / c_dim are filled in from the generator."""
    hid_dim: int = 256
    embed_dim: int = 16
    lr_multiplier: float = 0.1
    residual: bool = False
    adjust_angles: bool = True
    adjust_radius: bool = False
    adjust_fov: bool = True
    adjust_look_at: bool = True
    camera: CameraRanges = field(default_factory=CameraRanges)


@dataclass
class GeneratorConfig:
    z_dim: int = 512
    w_dim: int = 512
    c_dim: int = 0
    map_depth: int = 2
    cbase: int = 32768
    cmax: int = 512
    fmaps: float = 1.0
    use_noise: bool = True
    tri_plane_res: int = 512
    feat_dim: int = 32
    mlp_hid: int = 64
    mlp_n_layers: int = 2           # tri_plane.mlp.n_layers (networks_epigraf.py:35-43): 0 = the planes carry rgb + sigma (feat_dim 4); 2 = the fused kernel's form
    # mapping_kwargs injected by the launcher (train.py:170-172; layers.py:84-93,122-138): camera-conditioned mapping network
    camera_cond: bool = False
    camera_raw_scalars: bool = False
    camera_cond_drop_p: float = 0.0
    mean_camera_params: Optional[tuple] = None     # (yaw, pitch, roll, ...) used at eval time when no angles are passed
    has_view_cond: bool = False     # networks_epigraf.py:39: widens the decoder's output to 1 + hid_dim (the reference's forward then only accepts hid_dim == 3)
    ray_marcher_type: str = 'classical'
    num_ray_steps: int = 32
    ray_start: float = 0.75
    ray_end: float = 1.25
    cube_scale: float = 0.5
    use_inf_depth: bool = True
    last_back: bool = False
    white_back: bool = False
    density_bias: float = 0.0
    img_resolution: int = 256
    # Reduced-precision blocks (networks_epigraf.py:82,99-108, networks_stylegan2.py:237): the `num_fp16_res` highest-resolution blocks
    # of the tri-plane backbone keep their activations and weights in 16 bits -- bfloat16 here (BASELINE configs[4]: "bf16 weights on
    # CDNA4"), fp32 accumulation -- and every conv output is clamped to +-conv_clamp.  0 / None = `fp32_only` (train.py:271-273), the
    # setting of every 3dgp config and of the headline metric.
    num_fp16_res: int = 0
    conv_clamp: Optional[float] = None
    max_batch_res: int = 128        # above it an eval forward with cut_quantile > 0 is rendered in the reference's ray chunks (per-chunk quantiles)
    # training-mode forward: patch-wise rendering resolution (configs/training/patch_beta.yaml `resolution`;
    # None = patch.enabled off -> img_resolution) and the density-noise schedule (configs/model/3dgp.yaml:22-23)
    patch_resolution: Optional[int] = None
    nerf_noise_std_init: float = 1.0
    nerf_noise_kimg_growth: float = 5000
    # enabled in every 3dgp training config (model/base.yaml:32-35); None = module absent, as when
    # `training.use_depth` / `training.learn_camera_dist` are off.  The section-8a hot path (and bench.py's metric) is the
    # generator forward without them.
    depth_adaptor: Optional[DepthAdaptorConfig] = None
    camera_adaptor: Optional[CameraAdaptorConfig] = None

    def to_dict(self):
        return asdict(self)

    @property
    def block_resolutions(self):
        """networks_epigraf.py:94-96 (in_resolution = 0)."""
        return [2 ** i for i in range(2, int(math.log2(self.tri_plane_res)) + 1)]

    @property
    def channels(self):
        """networks_epigraf.py:98."""
        return {r: min(int(self.cbase * self.fmaps) // r, self.cmax) for r in self.block_resolutions}

    @property
    def fp16_resolution(self):
        """networks_epigraf.py:99: blocks at or above this resolution run in reduced precision; None = fp32 only."""
        if self.num_fp16_res <= 0:
            return None
        return max(2 ** (int(math.log2(self.tri_plane_res)) + 1 - self.num_fp16_res), 8)

    @property
    def num_ws(self):
        """networks_epigraf.py:101-112."""
        return 2 * len(self.block_resolutions)

    @property
    def plane_channels(self):
        return 3 * self.feat_dim


# The BASELINE.json configurations.
def config_c1():
    """SDFood-like 64x64, 32 ray steps, single class."""
    return GeneratorConfig(c_dim=0, img_resolution=64, num_ray_steps=32)


def config_c2():
    """Dogs 128x128, 48 ray steps (configs/model/epigraf.yaml:5)."""
    return GeneratorConfig(c_dim=0, img_resolution=128, num_ray_steps=48)


def config_c3():
    """ImageNet 256x256, 64 ray steps (32 x ray_step_multiplier 2, scripts/inference.py:45), cmax 512.
    Ray limits stay at camera/base.yaml's [0.75, 1.25]; inference.yaml's far_plane_offset is a
    visualisation-only setting and is not part of the metric configuration."""
    return GeneratorConfig(c_dim=1000, img_resolution=256, num_ray_steps=64)


def config_c4():
    """As C3 with cmax 1024 / cbase 65536 (README.md:57)."""
    return GeneratorConfig(c_dim=1000, img_resolution=256, num_ray_steps=64, cmax=1024, cbase=65536)


def config_c5(cmax=512, cbase=32768):
    """`config_c5(cmax=1024, cbase=65536)` is the sizing of BASELINE.md section 3, row 3 (bf16 at the 1024-channel backbone of configs[3],
    ~559 GFLOP per image); the default keeps configs[2]'s 512-channel backbone (BASELINE.json's configs[4] names no cmax).
    BASELINE configs[4]: ImageNet 256x256, 96(+96) ray steps, the four highest-resolution backbone blocks (64^2 ... 512^2) in bfloat16
    with fp32 accumulation and conv_clamp 256 -- the reference's `num_fp16_res = 4, conv_clamp = 256` defaults (networks_epigraf.py:82,
    networks_stylegan2.py:220) with bf16 in the place of fp16; skip image / tri-planes and the renderer stay fp32, as there."""
    return GeneratorConfig(c_dim=1000, img_resolution=256, num_ray_steps=96, num_fp16_res=4, conv_clamp=256.0, cmax=cmax, cbase=cbase)


def config_mid_bf16():
    """The reduced-precision golden (tools/gen_goldens.py:gen_bf16): 64^2 tri-planes of 3 x 8 features, 32 channels, the two
    highest-resolution blocks (32^2, 64^2: row widths the MFMA fast paths take) in bf16."""
    return GeneratorConfig(z_dim=32, w_dim=32, c_dim=0, cbase=2048, cmax=32, tri_plane_res=64, feat_dim=8, mlp_hid=16, num_ray_steps=8,
                           img_resolution=16, num_fp16_res=2, conv_clamp=256.0)


def config_tiny():
    """Fixture-sized configuration used by the golden end-to-end vectors."""
    return GeneratorConfig(z_dim=32, w_dim=32, c_dim=0, cbase=256, cmax=16, tri_plane_res=32, feat_dim=8,
                           mlp_hid=16, num_ray_steps=8, img_resolution=16)


def config_cut_chunked():
    """Golden of `cut_quantile` above `max_batch_res` (tools/gen_goldens.py:gen_cut_chunked): the tiny backbone in front of 128^2 rays x
    96 steps, so that a batch of 4 exceeds the reference's 2**24-element quantile chunk (networks_epigraf.py:235-236)."""
    return GeneratorConfig(z_dim=32, w_dim=32, c_dim=0, cbase=256, cmax=16, tri_plane_res=32, feat_dim=8,
                           mlp_hid=16, num_ray_steps=96, img_resolution=128, max_batch_res=64)


def config_mid():
    """Mid-sized golden configuration: exercises MFMA tile edges (channels 64/32, 64^2 planes)."""
    return GeneratorConfig(z_dim=64, w_dim=64, c_dim=10, cbase=2048, cmax=64, tri_plane_res=64, feat_dim=32,
                           mlp_hid=64, num_ray_steps=16, img_resolution=32)


def config_bigger():
    """The hot MLP shape (feat 32, hid 64) on 128^2 planes, 96-channel backbone, 48^2 rays x 24 steps (golden e2e_bigger)."""
    return GeneratorConfig(z_dim=64, w_dim=64, c_dim=0, cbase=4096, cmax=96, tri_plane_res=128, feat_dim=32, mlp_hid=64, num_ray_steps=24,
                           img_resolution=48)


def config_train_golden():
    """The training-mode golden (tools/gen_goldens.py:gen_train_forward): config_mid rendered patch-wise at 16^2."""
    cfg = config_mid()
    cfg.patch_resolution = 16
    return cfg


def configs_adaptor_goldens():
    """The two adaptor configurations behind tests/golden/adaptors.npz (tools/gen_goldens.py:gen_adaptors)."""
    a = config_tiny()
    a.depth_adaptor = DepthAdaptorConfig(hid_dim=16)
    a.camera_adaptor = CameraAdaptorConfig(hid_dim=32, embed_dim=8)
    b = config_mid()            # c_dim 10, 32^2 images -> the W % 32 == 0 fast path of the 5x5 kernel
    b.depth_adaptor = DepthAdaptorConfig(hid_dim=64, num_hid_layers=2, out_strategy='mean', near_plane_offset_max_fraction=0.4)
    b.camera_adaptor = CameraAdaptorConfig(hid_dim=64, embed_dim=16, residual=True, adjust_angles=True, adjust_radius=True, adjust_fov=False,
                                           adjust_look_at=True, camera=CameraRanges(look_at_radius=(0.0, 0.2)))
    return [('a', a), ('b', b)]
