"""Fused bias + activation (forward).

API mirror of the reference's `src/torch_utils/ops/bias_act.py`: `activation_funcs` (:21-31) and
`bias_act(x, b, dim, act, alpha, gain, clamp, impl)` (:52-86) with the same dispatch rule
(`impl == 'cuda' and x.device.type == 'cuda'` -> native kernel, else the `_ref` PyTorch path, :84-86).
The native path calls `tdgp_bias_act` (include/tdgp.h) instead of the JIT-compiled CUDA plugin
(bias_act.cpp:32); it is forward-only (generator inference; gradients are SURVEY.md 8f rank 4).
"""
import numpy as np
import torch

from .. import _lib


class _Spec(dict):
    """Attribute-style record (the reference uses dnnlib.EasyDict)."""
    __getattr__ = dict.__getitem__


def _table():
    F = torch.nn.functional
    s2 = float(np.sqrt(2))
    rows = [
        # name        id  fn                                      alpha gain ref  2nd-grad
        ('linear',    1,  lambda x, **_: x,                       0,    1,   '',  False),
        ('relu',      2,  lambda x, **_: F.relu(x),               0,    s2,  'y', False),
        ('lrelu',     3,  lambda x, alpha, **_: F.leaky_relu(x, alpha), 0.2, s2, 'y', False),
        ('tanh',      4,  lambda x, **_: torch.tanh(x),           0,    1,   'y', True),
        ('sigmoid',   5,  lambda x, **_: torch.sigmoid(x),        0,    1,   'y', True),
        ('elu',       6,  lambda x, **_: F.elu(x),                0,    1,   'y', True),
        ('selu',      7,  lambda x, **_: F.selu(x),               0,    1,   'y', True),
        ('softplus',  8,  lambda x, **_: F.softplus(x),           0,    1,   'y', True),
        ('swish',     9,  lambda x, **_: torch.sigmoid(x) * x,    0,    s2,  'x', True),
    ]
    return {n: _Spec(func=fn, def_alpha=a, def_gain=g, cuda_idx=i, ref=r, has_2nd_grad=h) for n, i, fn, a, g, r, h in rows}


# same keys / fields as the reference's table (bias_act.py:21-31); cuda_idx is the kernel's activation id
activation_funcs = _table()

_DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.float64: 3}      # include/tdgp.h TDGP_F32 / F16 / BF16 / F64


def _init():
    """Load the prebuilt HIP library (raises if absent; never compiles)."""
    _lib.load()
    return True


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    r"""y = clamp(gain * act(x + b)) -- arguments as in the reference (bias_act.py:52-86)."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        if torch.is_grad_enabled() and (x.requires_grad or (b is not None and b.requires_grad)):
            return _bias_act_autograd(x, b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)
        return _bias_act_hip(x, b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)
    return _bias_act_ref(x=x, b=b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)


def _bias_act_autograd(x, b, dim, act, alpha, gain, clamp):
    """The reference's autograd construction (bias_act.py:126-203) over the HIP kernel: forward = plugin call with grad 0; backward =
    the same call with grad 1 (and grad 2 behind it), which reads either the input or the output of the forward pass, whichever the
    activation's derivative is written in (`ref` of the activation table)."""
    from ..compat import BiasActPlugin as plugin
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    null = torch.empty([0], dtype=x.dtype, device=x.device)
    args = (dim, spec.cuda_idx, alpha, gain, clamp)

    class BiasAct(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, b):
            ctx.memory_format = torch.channels_last if x.ndim > 2 and x.stride(1) == 1 else torch.contiguous_format
            x = x.contiguous(memory_format=ctx.memory_format)
            b = b.contiguous() if b is not None else null
            y = x
            if act != 'linear' or gain != 1 or clamp >= 0 or b is not null:
                y = plugin.bias_act(x, b, null, null, null, 0, *args)
            ctx.save_for_backward(x if 'x' in spec.ref or spec.has_2nd_grad else null, b if 'x' in spec.ref or spec.has_2nd_grad else null,
                                  y if 'y' in spec.ref else null)
            return y

        @staticmethod
        def backward(ctx, dy):
            dy = dy.contiguous(memory_format=ctx.memory_format)
            x, b, y = ctx.saved_tensors
            dx = db = None
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
                dx = dy
                if act != 'linear' or gain != 1 or clamp >= 0:
                    dx = BiasActGrad.apply(dy, x, b, y)
            if ctx.needs_input_grad[1]:
                db = dx.sum([i for i in range(dx.ndim) if i != dim])
            return dx, db

    class BiasActGrad(torch.autograd.Function):
        @staticmethod
        def forward(ctx, dy, x, b, y):
            ctx.memory_format = torch.channels_last if dy.ndim > 2 and dy.stride(1) == 1 else torch.contiguous_format
            dx = plugin.bias_act(dy, b, x, y, null, 1, *args)
            ctx.save_for_backward(dy if spec.has_2nd_grad else null, x, b, y)
            return dx

        @staticmethod
        def backward(ctx, d_dx):
            d_dx = d_dx.contiguous(memory_format=ctx.memory_format)
            dy, x, b, y = ctx.saved_tensors
            d_dy = d_x = d_b = None
            if ctx.needs_input_grad[0]:
                d_dy = BiasActGrad.apply(d_dx, x, b, y)
            if spec.has_2nd_grad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
                d_x = plugin.bias_act(d_dx, b, x, y, dy, 2, *args)
            if spec.has_2nd_grad and ctx.needs_input_grad[2]:
                d_b = d_x.sum([i for i in range(d_x.ndim) if i != dim])
            return d_dy, d_x, d_b, None

    return BiasAct.apply(x, b)


def _resolve(act, alpha, gain, clamp):
    if clamp is not None and clamp < 0:
        raise AssertionError('clamp must be None or >= 0')
    spec = activation_funcs[act]
    return (spec, float(spec.def_alpha if alpha is None else alpha), float(spec.def_gain if gain is None else gain),
            float(-1 if clamp is None else clamp))


def _bias_act_ref(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """`impl='ref'`: stock PyTorch ops in the order bias -> activation -> gain -> clamp (bias_act.py:91-120)."""
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    y = x
    if b is not None:
        if not (isinstance(b, torch.Tensor) and b.ndim == 1 and 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]):
            raise AssertionError('b must be a vector matching x.shape[dim]')
        shape = [1] * x.ndim
        shape[dim] = -1
        y = y + b.reshape(shape)
    y = spec.func(y, alpha=alpha)
    if gain != 1:
        y = y * gain
    return y.clamp(-clamp, clamp) if clamp >= 0 else y


def _bias_act_hip(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    if x.dtype not in _DTYPES:
        raise RuntimeError(f'bias_act: dtype {x.dtype} has no HIP kernel (float32 / float64 / float16 / bfloat16)')
    # layout handling of BiasActCuda.forward (bias_act.py:144-150)
    memory_format = torch.channels_last if x.ndim > 2 and x.stride(1) == 1 else torch.contiguous_format
    x = x.contiguous(memory_format=memory_format)
    if b is not None:
        if not (isinstance(b, torch.Tensor) and b.ndim == 1):
            raise RuntimeError('b must have rank 1')
        if b.dtype != x.dtype or b.device != x.device:
            raise RuntimeError('b must have the same dtype and device as x')
        if not (0 <= dim < x.ndim):
            raise RuntimeError('dim is out of bounds')
        if b.shape[0] != x.shape[dim]:
            raise RuntimeError('b has wrong number of elements')
        b = b.contiguous()
    if act == 'linear' and gain == 1 and clamp < 0 and b is None:
        return x
    if x.numel() > 2 ** 31 - 1:
        raise RuntimeError('x is too large')
    y = torch.empty_like(x)
    if x.numel() == 0:
        return y
    step = x.stride(dim) if b is not None else 1
    with torch.cuda.device(x.device):
        _lib.call('tdgp_bias_act', x.data_ptr(), _lib.ptr(b), y.data_ptr(), x.numel(), b.shape[0] if b is not None else 1, step,
                  spec.cuda_idx, alpha, gain, clamp, _DTYPES[x.dtype], _lib.stream_of(x))
    return y
