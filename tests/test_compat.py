"""`src.*` aliases (3dgp_amd/compat.py).  The second test runs only where the reference tree exists (the build
container): the REFERENCE's own model code is imported with this package's op modules shadowing
`src.torch_utils.ops.*`, and its generator forward must reproduce the golden image -- that is the drop-in proof for the
op API (signatures, defaults, call order).  Nothing here is needed on the GPU box."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('TDGP_REFERENCE', '/root/reference')


def test_aliases_without_reference():
    code = '''
import importlib, sys
sys.path.insert(0, %r)
t = importlib.import_module("3dgp_amd")
bound = t.compat.install_src_aliases()
from src.torch_utils.ops import bias_act, upfirdn2d, conv2d_resample, conv2d_gradfix, fma
assert conv2d_gradfix is t.ops.conv2d_gradfix and fma is t.ops.fma and conv2d_gradfix.enabled and callable(conv2d_gradfix.no_weight_gradients)
from src.torch_utils import custom_ops
from src.dnnlib import EasyDict, TensorGroup
assert bias_act is t.ops.bias_act and upfirdn2d is t.ops.upfirdn2d and conv2d_resample is t.ops.conv2d_resample
p = custom_ops.get_plugin("bias_act_plugin", sources=["bias_act.cpp", "bias_act.cu"], headers=["bias_act.h"], source_dir=".")
assert hasattr(p, "bias_act") and hasattr(custom_ops.get_plugin("upfirdn2d_plugin"), "upfirdn2d")
assert EasyDict(a=1).a == 1
print("ok", len(bound))
''' % REPO
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith('ok'), out.stderr


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'src')), reason='reference tree only exists in the build container')
def test_reference_model_code_runs_on_our_ops():
    code = '''
import importlib, sys, types
import numpy as np
sys.path.insert(0, %r)
om = types.ModuleType("omegaconf"); om.DictConfig = dict; om.OmegaConf = object; sys.modules["omegaconf"] = om
sys.modules["torchvision"] = types.ModuleType("torchvision")
sys.path.insert(0, %r)
t = importlib.import_module("3dgp_amd")
import src.torch_utils.ops                                       # the reference package ...
t.compat.install_src_aliases()                                   # ... with our op modules shadowing its own
import torch
sys.path.insert(0, %r)
import gen_goldens as GG                                         # build_ref_generator / PatchedRNG helpers (reference-side harness)
from src.training import networks_stylegan2
assert networks_stylegan2.bias_act is t.ops.bias_act and networks_stylegan2.upfirdn2d is t.ops.upfirdn2d
cfg = t.config.config_tiny()
sd = t.weights.random_state_dict(cfg, seed=21, exercise_all=True)
G = GG.build_ref_generator(cfg, sd)
g = dict(np.load(%r))
cam = GG.TensorGroup(**{k[4:]: torch.as_tensor(v) for k, v in g.items() if k.startswith("cam_")})
R, S = cfg.img_resolution ** 2, cfg.num_ray_steps
with torch.no_grad(), GG.PatchedRNG(rand_like=[torch.as_tensor(g["u_coarse"]).reshape(2, R, S, 1)], rand=[torch.as_tensor(g["u_fine"])]):
    img = G.synthesis(torch.as_tensor(g["ws"]), camera_params=cam, noise_mode="const").numpy()
err = np.abs(img - g["img"]).max() / np.abs(g["img"]).max()
assert err < 1e-5, err
print("ok", err)
''' % (REPO, REF, os.path.join(REPO, 'tools'), os.path.join(REPO, 'tests', 'golden', 'e2e_tiny.npz'))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr[-3000:]
