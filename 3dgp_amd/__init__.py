"""3dgp_amd -- MI355X-native generator-forward hot path of 3DGP (gfx950 HIP kernels behind a C ABI).

The directory name starts with a digit, so import it with
``importlib.import_module('3dgp_amd')`` (tests/conftest.py and the repo-root scripts do).
"""
from . import config, weights  # noqa: F401
from .config import GeneratorConfig  # noqa: F401
