"""`src.*` aliases: make reference-style call sites resolve to this package.

The reference's model code imports its ops as `from src.torch_utils.ops import bias_act, upfirdn2d, conv2d_resample`
(networks_stylegan2.py:18-26, layers.py:7-11) and obtains the native plugins through
`custom_ops.get_plugin('bias_act_plugin' | 'upfirdn2d_plugin', ...)` (bias_act.py:38-48, upfirdn2d.py:23-33).
`install_src_aliases()` registers HIP-backed modules under those dotted names, and plugin objects with the pybind
signatures (bias_act.cpp:32, upfirdn2d.cpp:16) in the plugin cache, so nothing is JIT-compiled or hipified.
If a real `src` package is importable (the reference tree on sys.path) only the op modules are overridden.
"""
import ctypes
import importlib
import sys
import types

import torch

from . import _lib
from .ops import bias_act as _bias_act
from .ops import conv2d_resample as _conv2d_resample
from .ops import upfirdn2d as _upfirdn2d

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.float64: 3}       # include/tdgp.h TDGP_F32 / F16 / BF16 / F64


class BiasActPlugin:
    """Object with the pybind signature of the reference's bias_act plugin (bias_act.cpp:32): forward (grad 0) and the two
    derivative forms the reference's autograd wrappers call (grad 1, 2; bias_act.py:172-197)."""

    @staticmethod
    def bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp):
        _lib.require_cuda(x, 'x')
        if grad not in (0, 1, 2):
            raise RuntimeError('bias_act plugin: grad must be 0, 1 or 2')
        if not x.is_contiguous() and not (x.ndim == 4 and x.is_contiguous(memory_format=torch.channels_last)):
            raise RuntimeError('x must be non-overlapping and dense')
        has_b = b is not None and b.numel() > 0
        if has_b and (b.dtype != x.dtype or b.device != x.device):
            raise RuntimeError('b must have the same dtype and device as x')
        if has_b and b.numel() != x.shape[dim]:
            raise RuntimeError('b has wrong number of elements')
        opt = {}
        for name, t in (('xref', xref), ('yref', yref), ('dy', dy)):        # empty tensor == absent (bias_act.cpp:43-50)
            if t is not None and t.numel() > 0:
                if t.shape != x.shape or t.dtype != x.dtype or t.stride() != x.stride():
                    raise RuntimeError(f'{name} must have the same shape, dtype and layout as x')
                opt[name] = t
        y = torch.empty_like(x)
        if x.numel():
            bp, nb, sb = (b.data_ptr(), b.numel(), x.stride(dim)) if has_b else (None, 1, 1)
            with torch.cuda.device(x.device):
                if grad == 0:
                    _lib.call('tdgp_bias_act', x.data_ptr(), bp, y.data_ptr(), x.numel(), nb, sb, int(act), float(alpha), float(gain), float(clamp), _DT[x.dtype],
                              _lib.stream_of(x))
                else:
                    _lib.call('tdgp_bias_act_grad', x.data_ptr(), bp, _lib.ptr(opt.get('xref')), _lib.ptr(opt.get('yref')), _lib.ptr(opt.get('dy')), y.data_ptr(),
                              x.numel(), nb, sb, int(grad), int(act), float(alpha), float(gain), float(clamp), _DT[x.dtype], _lib.stream_of(x))
        return y


class Upfirdn2dPlugin:
    """Object with the pybind signature of the reference's upfirdn2d plugin (upfirdn2d.cpp:16)."""

    @staticmethod
    def upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain):
        _lib.require_cuda(x, 'x')
        return _upfirdn2d._launch(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain)


_PLUGINS = {'bias_act_plugin': BiasActPlugin, 'upfirdn2d_plugin': Upfirdn2dPlugin}


def get_plugin(module_name, sources=None, headers=None, source_dir=None, **build_kwargs):
    """Signature of custom_ops.get_plugin (custom_ops.py:59); returns the prebuilt binding, never compiles."""
    if module_name not in _PLUGINS:
        raise RuntimeError(f'plugin "{module_name}" has no HIP implementation (bias_act_plugin, upfirdn2d_plugin)')
    _lib.load()
    return _PLUGINS[module_name]


class EasyDict(dict):
    """dnnlib.EasyDict (util.py:42-54)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]


def install_src_aliases(override=True):
    """Register `src.torch_utils.ops.{bias_act,upfirdn2d,conv2d_resample,conv2d_gradfix,fma}`, `src.torch_utils.custom_ops` and
    `src.dnnlib` (when no real one exists).  Returns the list of module names it (re)bound."""
    bound = []

    def ensure_pkg(name):
        try:
            return importlib.import_module(name)
        except ImportError:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
            parent, _, leaf = name.rpartition('.')
            if parent:
                setattr(sys.modules[parent], leaf, m)
            bound.append(name)
            return m

    for pkg in ('src', 'src.torch_utils', 'src.torch_utils.ops'):
        ensure_pkg(pkg)
    from .ops import conv2d_gradfix as _conv2d_gradfix, fma as _fma
    for leaf, mod in (('bias_act', _bias_act), ('upfirdn2d', _upfirdn2d), ('conv2d_resample', _conv2d_resample), ('conv2d_gradfix', _conv2d_gradfix),
                      ('fma', _fma)):
        name = f'src.torch_utils.ops.{leaf}'
        if override or name not in sys.modules:
            sys.modules[name] = mod
            setattr(sys.modules['src.torch_utils.ops'], leaf, mod)
            bound.append(name)
    name = 'src.torch_utils.custom_ops'
    try:
        real = importlib.import_module(name)
        real._cached_plugins.update(_PLUGINS)          # the reference's own loader now returns the prebuilt bindings
    except ImportError:
        m = types.ModuleType(name)
        m.get_plugin, m._cached_plugins, m.verbosity = get_plugin, dict(_PLUGINS), 'none'
        sys.modules[name] = m
        setattr(sys.modules['src.torch_utils'], 'custom_ops', m)
        bound.append(name)
    try:
        importlib.import_module('src.dnnlib')
    except ImportError:
        from .generator import TensorGroup
        m = types.ModuleType('src.dnnlib')
        m.EasyDict, m.TensorGroup = EasyDict, TensorGroup
        sys.modules['src.dnnlib'] = m
        setattr(sys.modules['src'], 'dnnlib', m)
        bound.append('src.dnnlib')
    return bound
