"""3dgp_amd -- MI355X-native generator-forward hot path of 3DGP (gfx950 HIP kernels behind a C ABI).

The directory name starts with a digit, so import it with ``importlib.import_module('3dgp_amd')``
(tests/conftest.py and the repo-root scripts do).

  config, weights      configuration + reference state-dict layout + deterministic synthetic weights (numpy only)
  _lib                 ctypes binding of csrc/libtdgp_hip.so (include/tdgp.h)
  ops                  bias_act / upfirdn2d / conv2d_resample / modulated_conv2d with the reference's signatures
  renderer             camera, rays, tri-plane field, importance sampling, ray marchers
  generator            Generator / SynthesisNetwork / MappingNetwork with the reference's state-dict names
  adaptors             DepthAdaptor / CameraAdaptor / Conv2dLayer
  metrics              FeatureStats, Frechet distance, camera priors, generator feature loop (host side)
  inference            generate / generate_trajectory / camera trajectories
  compat               `src.*` module aliases so reference-style call sites resolve to this package
  distributed          batch-sharded multi-GPU generation (one process per GPU, RCCL all-gather of features)
  graphs               the whole generator forward as one captured HIP graph per (batch, options)
"""
from . import config, weights  # noqa: F401  (numpy only)
from .config import GeneratorConfig  # noqa: F401


def __getattr__(name):
    # torch-dependent submodules are imported on first use
    if name in ('_lib', 'ops', 'renderer', 'generator', 'adaptors', 'metrics', 'inference', 'discriminator', 'training', 'compat', 'distributed', 'build', 'graphs'):
        import importlib
        return importlib.import_module(f'{__name__}.{name}')
    raise AttributeError(name)


def inference_golden_trajectories():
    """The trajectory configurations behind tests/golden/trajectories.npz (configs/scripts/inference.yaml style entries)."""
    return dict(point=dict(name='point', num_frames=1, yaw_offset=0.3, pitch_offset=-0.1, fov_offset=2.0),
                front_circle=dict(name='front_circle', num_frames=8, yaw_diff=0.4, pitch_diff=0.2, fov_diff=1.0),
                points=dict(name='points', yaw_offsets=[-0.5, 0.0, 0.5], pitch_offset=0.1),
                line=dict(name='line', num_frames=5, yaw_start=-0.6, yaw_end=0.6, pitch_start=1.2, pitch_end=1.8, fov=None, fov_offset=1.5))
