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


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'src')), reason='reference tree only exists in the build container')
def test_reference_call_sites_reach_the_c_abi():
    """the REFERENCE's own call sites -> `custom_ops.get_plugin(...)` plugin objects -> ctypes -> `tdgp_bias_act` /
    `tdgp_upfirdn2d`.  There is no GPU here, so a host build of those two entry points (oracle/tdgp_host_abi.c, same prototypes, test
    infrastructure) is loaded in the place of libtdgp_hip.so; the test switches the op modules' device check off (a test-only patch:
    the product has no CPU path) and runs the reference generator: the golden image must come out, and every call must have gone
    through the C ABI -- counted."""
    so = os.path.join(REPO, 'oracle', '_build', 'libtdgp_host_abi.so')
    subprocess.check_call(['make', '-C', os.path.join(REPO, 'oracle'), '_build/libtdgp_host_abi.so'], stdout=subprocess.DEVNULL)
    code = '''
import contextlib, importlib, sys, types
import numpy as np
sys.path.insert(0, %r)
om = types.ModuleType("omegaconf"); om.DictConfig = dict; om.OmegaConf = object; sys.modules["omegaconf"] = om
sys.modules["torchvision"] = types.ModuleType("torchvision")
sys.path.insert(0, %r)
import torch
t = importlib.import_module("3dgp_amd")
L = t._lib
L.LIB_PATH = %r            # test-only: the host build of the same prototypes in the place of libtdgp_hip.so (no GPU here)
# ---- test-only patches: tensors live on the host, the library is the host build of the same ABI ------------------------------
L.require_cuda = lambda x, what: None
L.stream_of = lambda x: None
torch.cuda.device = lambda dev: contextlib.nullcontext()
calls = {}
real_call = L.call
def counting_call(name, *a):
    calls[name] = calls.get(name, 0) + 1
    return real_call(name, *a)
L.call = counting_call
import src.torch_utils.ops
t.compat.install_src_aliases()
from src.torch_utils import custom_ops
ba, uf = t.ops.bias_act, t.ops.upfirdn2d
plug_b = custom_ops.get_plugin("bias_act_plugin", sources=["bias_act.cpp", "bias_act.cu"], headers=["bias_act.h"], source_dir=".", extra_cuda_cflags=["--use_fast_math"])
plug_u = custom_ops.get_plugin("upfirdn2d_plugin", sources=["upfirdn2d.cpp", "upfirdn2d.cu"], headers=["upfirdn2d.h"], source_dir=".")
def bias_act(x, b=None, dim=1, act="linear", alpha=None, gain=None, clamp=None, impl="cuda"):
    # the plugin call of bias_act.py:145-150 (BiasActCuda.forward): absent tensors are empty, clamp < 0 = off
    spec = ba.activation_funcs[act]
    null = torch.empty([0], dtype=x.dtype)
    x = x.contiguous()
    if act == "linear" and (gain is None or gain == 1) and clamp is None and b is None:
        return x
    return plug_b.bias_act(x, b.contiguous() if b is not None else null, null, null, null, 0, dim, spec.cuda_idx, float(spec.def_alpha if alpha is None else alpha),
                           float(spec.def_gain if gain is None else gain), float(-1 if clamp is None else clamp))
def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl="cuda"):
    # the plugin call(s) of upfirdn2d.py:234-245 (Upfirdn2dCuda.forward)
    upx, upy = uf._parse_scaling(up); downx, downy = uf._parse_scaling(down); px0, px1, py0, py1 = uf._parse_padding(padding)
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32)
    if f.ndim == 1 and f.shape[0] == 1:
        f = f.square().unsqueeze(0)
    if f.ndim == 2:
        return plug_u.upfirdn2d(x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip_filter, gain)
    y = plug_u.upfirdn2d(x, f.unsqueeze(0), upx, 1, downx, 1, px0, px1, 0, 0, flip_filter, 1.0)
    return plug_u.upfirdn2d(y, f.unsqueeze(1), 1, upy, 1, downy, 0, 0, py0, py1, flip_filter, gain)
ba.bias_act = bias_act
uf.upfirdn2d = upfirdn2d
uf.upsample2d.__globals__["upfirdn2d"] = upfirdn2d
sys.path.insert(0, %r)
import gen_goldens as GG
from src.training import networks_stylegan2
assert networks_stylegan2.bias_act is ba and networks_stylegan2.upfirdn2d is uf
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
assert calls.get("tdgp_bias_act", 0) == 13 and calls.get("tdgp_upfirdn2d", 0) == 6, calls      # 7 conv + 4 ToRGB + 2 MLP bias_acts; 3 FIR + 3 skip upsamples
# error convention across the boundary: a stub entry point fails with TDGP_EUNSUPPORTED and a message, nothing throws in C
try:
    real_call("tdgp_planes_to_hwc", None, None, 1, 8, 4, 4, None)
    raise SystemExit("stub did not fail")
except RuntimeError as e:
    assert "not part of the host build" in str(e), e
print("ok", err, calls)
''' % (REPO, REF, so, os.path.join(REPO, 'tools'), os.path.join(REPO, 'tests', 'golden', 'e2e_tiny.npz'))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr[-3000:]
