"""Configuration of the generator-forward hot path.

Field names follow the keys the reference's hot path actually reads:
`configs/model/{base,3dgp}.yaml`, `configs/camera/base.yaml` and the constructor kwargs injected by
`src/train.py:156-273`.  Only what the generator forward consumes is kept.
"""
import math
from dataclasses import dataclass, asdict


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
    max_batch_res: int = 128        # kept for API parity (run_batchwise chunking is a no-op for results)

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


def config_tiny():
    """Fixture-sized configuration used by the golden end-to-end vectors."""
    return GeneratorConfig(z_dim=32, w_dim=32, c_dim=0, cbase=256, cmax=16, tri_plane_res=32, feat_dim=8,
                           mlp_hid=16, num_ray_steps=8, img_resolution=16)


def config_mid():
    """Mid-sized golden configuration: exercises MFMA tile edges (channels 64/32, 64^2 planes)."""
    return GeneratorConfig(z_dim=64, w_dim=64, c_dim=10, cbase=2048, cmax=64, tri_plane_res=64, feat_dim=32,
                           mlp_hid=64, num_ray_steps=16, img_resolution=32)
