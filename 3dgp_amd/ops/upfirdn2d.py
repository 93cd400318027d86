"""upfirdn2d: pad, upsample, FIR-filter and downsample a batch of 2-D images (forward).

API mirror of the reference's `src/torch_utils/ops/upfirdn2d.py`: `setup_filter` (:70-114),
`upfirdn2d` (:118-162), `filter2d` (:277-309), `upsample2d` (:313-348), `downsample2d` (:352-387) and the
private helpers other modules import (`_parse_padding`, `_get_filter_size`, `_parse_scaling`).
The native path calls `tdgp_upfirdn2d` (include/tdgp.h) instead of the CUDA plugin (upfirdn2d.cpp:16);
separable filters are issued as two calls exactly like upfirdn2d.py:241-245.
"""
import ctypes

import numpy as np
import torch

from .. import _lib

_DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.float64: 3}      # include/tdgp.h TDGP_F32 / F16 / BF16 / F64


def _init():
    _lib.load()
    return True


def _pair(v, what):
    """int | (x, y) -> (x, y), both >= 1."""
    vx, vy = (v, v) if isinstance(v, int) else tuple(v)
    if not (isinstance(vx, int) and isinstance(vy, int) and vx >= 1 and vy >= 1):
        raise AssertionError(f'{what} must be a positive int or a pair of positive ints, got {v!r}')
    return vx, vy


def _quad(padding):
    """int | (x, y) | (x0, x1, y0, y1) -> (x0, x1, y0, y1); negative values crop."""
    if isinstance(padding, int):
        return padding, padding, padding, padding
    vals = tuple(padding)
    if not all(isinstance(v, int) for v in vals) or len(vals) not in (2, 4):
        raise AssertionError(f'padding must be an int, a pair or a 4-tuple of ints, got {padding!r}')
    if len(vals) == 2:
        return vals[0], vals[0], vals[1], vals[1]
    return vals


def _filter_wh(f):
    """(width, height) of a filter tensor; None is the 1x1 identity."""
    if f is None:
        return 1, 1
    if not (isinstance(f, torch.Tensor) and f.ndim in (1, 2)):
        raise AssertionError('filter must be a 1-D or 2-D tensor')
    return int(f.shape[-1]), int(f.shape[0])


# names other modules of the reference tree import from here (conv2d_resample.py:16-17)
_parse_scaling = lambda scaling: _pair(scaling, 'scaling')   # noqa: E731
_parse_padding = _quad
_get_filter_size = _filter_wh


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """Prepare an FIR filter for upfirdn2d (behaviour of upfirdn2d.py:70-114).

    A 1-D tap list with fewer than 8 taps becomes its 2-D outer product ([1,3,3,1] -> a 4x4 filter summing to 1);
    8 or more taps stay separable.  `gain` is split evenly across the axes.
    """
    taps = torch.as_tensor(1 if f is None else f, dtype=torch.float32)
    if taps.ndim == 0:
        taps = taps.reshape(1)
    if taps.ndim not in (1, 2) or taps.numel() == 0:
        raise AssertionError('filter must be a scalar, a tap list or a 2-D array')
    if separable is None:
        separable = taps.ndim == 1 and taps.numel() >= 8
    if taps.ndim == 1 and not separable:
        taps = torch.outer(taps, taps)
    if taps.ndim != (1 if separable else 2):
        raise AssertionError('a 2-D filter cannot be separable')
    if normalize:
        taps = taps / taps.sum()
    if flip_filter:
        taps = torch.flip(taps, dims=tuple(range(taps.ndim)))
    return (taps * gain ** (taps.ndim / 2)).to(device=device)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Arguments and dispatch rule as in the reference (upfirdn2d.py:118-162)."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        if torch.is_grad_enabled() and x.requires_grad:
            return _Upfirdn2dFunction.apply(x, f, up, down, padding, flip_filter, gain)
        return _upfirdn2d_hip(x, f, up=up, down=down, padding=padding, flip_filter=flip_filter, gain=gain)
    return _upfirdn2d_ref(x, f, up=up, down=down, padding=padding, flip_filter=flip_filter, gain=gain)


def _upfirdn2d_ref(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """`impl='ref'`: the same result from stock PyTorch ops (zero-stuffing, pad/crop, depthwise correlation, stride)."""
    if not (isinstance(x, torch.Tensor) and x.ndim == 4):
        raise AssertionError('x must be a rank-4 tensor')
    ux, uy = _pair(up, 'up')
    dx, dy = _pair(down, 'down')
    px0, px1, py0, py1 = _quad(padding)
    taps = torch.ones([1, 1], dtype=torch.float32, device=x.device) if f is None else f
    if taps.dtype != torch.float32 or taps.ndim not in (1, 2):
        raise AssertionError('filter must be a float32 1-D or 2-D tensor')
    n, c, h, w = x.shape
    if w * ux + px0 + px1 < taps.shape[-1] or h * uy + py0 + py1 < taps.shape[0]:
        raise AssertionError('the padded, upsampled image is smaller than the filter')
    # zero-stuffing
    z = x.new_zeros([n, c, h * uy, w * ux])
    z[:, :, ::uy, ::ux] = x
    # positive padding pads, negative padding crops
    z = torch.nn.functional.pad(z, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    z = z[:, :, max(-py0, 0): z.shape[2] - max(-py1, 0), max(-px0, 0): z.shape[3] - max(-px1, 0)]
    k = (taps * gain ** (taps.ndim / 2)).to(z.dtype)
    if not flip_filter:                                  # conv2d correlates; a convolution needs the flipped taps
        k = torch.flip(k, dims=tuple(range(k.ndim)))
    if k.ndim == 2:
        z = torch.nn.functional.conv2d(z, k[None, None].expand(c, 1, -1, -1), groups=c)
    else:
        z = torch.nn.functional.conv2d(z, k[None, None, None, :].expand(c, 1, 1, -1), groups=c)
        z = torch.nn.functional.conv2d(z, k[None, None, :, None].expand(c, 1, -1, 1), groups=c)
    return z[:, :, ::dy, ::dx]


def _launch(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain):
    """One plugin call: argument set and checks of upfirdn2d.cpp:16-40."""
    if f.device != x.device:
        raise RuntimeError('f must reside on the same device as x')
    if f.dtype != torch.float32:
        raise RuntimeError('f must be float32')
    if x.ndim != 4:
        raise RuntimeError('x must be rank 4')
    if f.ndim != 2:
        raise RuntimeError('f must be rank 2')
    if x.numel() == 0:
        raise RuntimeError('x has zero size')
    if x.dtype not in _DTYPES:
        raise RuntimeError(f'upfirdn2d: dtype {x.dtype} has no HIP kernel (float32 / float64 / float16 / bfloat16)')
    f = f.contiguous()
    n, c, h, w = x.shape
    fh, fw = f.shape
    ow = (w * upx + padx0 + padx1 - fw + downx) // downx
    oh = (h * upy + pady0 + pady1 - fh + downy) // downy
    if ow < 1 or oh < 1:
        raise RuntimeError('output must be at least 1x1')
    mf = torch.channels_last if (x.stride(1) == 1 and c > 1) else torch.contiguous_format   # x.suggest_memory_format()
    y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device, memory_format=mf)
    xs = (ctypes.c_int64 * 4)(*x.stride())
    ys = (ctypes.c_int64 * 4)(*y.stride())
    with torch.cuda.device(x.device):
        _lib.call('tdgp_upfirdn2d', x.data_ptr(), f.data_ptr(), y.data_ptr(), n, c, h, w, xs, oh, ow, ys, fh, fw, upx, upy, downx, downy,
                  padx0, padx1, pady0, pady1, int(bool(flip)), float(gain), _DTYPES[x.dtype], _lib.stream_of(x))
    return y


class _Upfirdn2dFunction(torch.autograd.Function):
    """upfirdn2d under autograd (upfirdn2d.py:197-269): the gradient w.r.t. x is upfirdn2d again with up / down swapped, the filter
    flipped and the padding of :252-257 -- issued through the differentiable entry point, so higher orders work too."""

    @staticmethod
    def forward(ctx, x, f, up, down, padding, flip_filter, gain):
        ctx.args = (up, down, padding, flip_filter, gain)
        ctx.save_for_backward(f)
        ctx.x_shape = x.shape
        y = _upfirdn2d_hip(x, f, up=up, down=down, padding=padding, flip_filter=flip_filter, gain=gain)
        ctx.y_shape = y.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        f, = ctx.saved_tensors
        up, down, padding, flip_filter, gain = ctx.args
        upx, upy = _pair(up, 'up')
        downx, downy = _pair(down, 'down')
        padx0, padx1, pady0, pady1 = _quad(padding)
        fw, fh = _filter_wh(f)
        _, _, ih, iw = ctx.x_shape
        _, _, oh, ow = ctx.y_shape
        p = [fw - padx0 - 1, iw * upx - ow * downx + padx0 - upx + 1, fh - pady0 - 1, ih * upy - oh * downy + pady0 - upy + 1]
        dx = upfirdn2d(dy, f, up=[downx, downy], down=[upx, upy], padding=p, flip_filter=(not flip_filter), gain=gain) if ctx.needs_input_grad[0] else None
        return dx, None, None, None, None, None, None


def _upfirdn2d_hip(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    upx, upy = _pair(up, 'up')
    downx, downy = _pair(down, 'down')
    padx0, padx1, pady0, pady1 = _quad(padding)
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    if f.ndim == 1 and f.shape[0] == 1:
        f = f.square().unsqueeze(0)                     # separable-1 -> full 1x1 (upfirdn2d.py:236-237)
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    if f.ndim == 2:
        return _launch(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip_filter, gain)
    y = _launch(x, f.unsqueeze(0), upx, 1, downx, 1, padx0, padx1, 0, 0, flip_filter, 1.0)
    return _launch(y, f.unsqueeze(1), 1, upy, 1, downy, 0, 0, pady0, pady1, flip_filter, gain)


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Same-size filtering (upfirdn2d.py:277-309): pads by half the filter so the output matches the input."""
    x0, x1, y0, y1 = _quad(padding)
    fw, fh = _filter_wh(f)
    pads = [x0 + fw // 2, x1 + (fw - 1) // 2, y0 + fh // 2, y1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=pads, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Upsample by `up` (upfirdn2d.py:313-348): output = input * up; the gain compensates the inserted zeros."""
    ux, uy = _pair(up, 'up')
    x0, x1, y0, y1 = _quad(padding)
    fw, fh = _filter_wh(f)
    pads = [x0 + (fw + ux - 1) // 2, x1 + (fw - ux) // 2, y0 + (fh + uy - 1) // 2, y1 + (fh - uy) // 2]
    return upfirdn2d(x, f, up=up, padding=pads, flip_filter=flip_filter, gain=gain * ux * uy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Downsample by `down` (upfirdn2d.py:352-387): output = input / down."""
    dx, dy = _pair(down, 'down')
    x0, x1, y0, y1 = _quad(padding)
    fw, fh = _filter_wh(f)
    pads = [x0 + (fw - dx + 1) // 2, x1 + (fw - dx) // 2, y0 + (fh - dy + 1) // 2, y1 + (fh - dy) // 2]
    return upfirdn2d(x, f, down=down, padding=pads, flip_filter=flip_filter, gain=gain, impl=impl)
