"""conv2d_gradfix: convolution with explicit first-order gradients.

Signature of the reference's `src/torch_utils/ops/conv2d_gradfix.py`: `conv2d(input, weight, bias, stride, padding, dilation,
groups)`, `conv_transpose2d(...)`, the `no_weight_gradients()` context manager.  The reference routes CUDA tensors through a
custom autograd function whose backward issues the input gradient as the opposite (transposed) convolution and the weight
gradient as `aten::convolution_backward` (conv2d_gradfix.py:106-150); everything else falls back to `torch.nn.functional`
(conv2d_gradfix.py:36-44).

Here the form the generator / adaptor layers use -- stride 1, dilation 1, groups 1, odd square kernel k <= 5, padding k // 2,
fp32 on the GPU -- runs on the gfx950 kernels:
    forward        tdgp_modconv2d (unmodulated: the same MFMA implicit-GEMM kernel as the synthesis layers)
    input grad     tdgp_modconv2d on dy with the flipped, transposed weights (a 'same' stride-1 convolution is its own adjoint form)
    weight grad    tdgp_conv2d_weight_grad (conv_grad.hip; also stride 2 / any padding through `conv2d_weight_grad`)
    bias grad      dy.sum([0, 2, 3])
and the strided form of the discriminator's down-sampling layers (k in {1,3}, stride in {1,2}, padding <= k - 1): forward
tdgp_conv2d, input gradient = tdgp_conv2d on the zero-stuffed, padded dy with the flipped, transposed weights.
Other forms take the reference's own fallback, `torch.nn.functional.conv2d / conv_transpose2d` (MIOpen on ROCm, cuDNN there).
"""
import contextlib

import torch

from .. import _lib
from . import modconv as _modconv

enabled = True                              # conv2d_gradfix.py:23 (the reference's switch between its custom op and torch; always on here)
weight_gradients_disabled = False           # conv2d_gradfix.py:24


@contextlib.contextmanager
def no_weight_gradients(disable=True):
    """conv2d_gradfix.py:26-32."""
    global weight_gradients_disabled
    old = weight_gradients_disabled
    if disable:
        weight_gradients_disabled = True
    yield
    weight_gradients_disabled = old


_fallback_seen = set()


def _note_fallback(op, input, weight, stride, padding, groups):
    """A GPU tensor taking the torch.nn.functional path is never silent: one warning per (op, form)."""
    if isinstance(input, torch.Tensor) and input.is_cuda:
        key = (op, tuple(weight.shape[2:]), str(stride), str(padding), groups, str(input.dtype))
        if key not in _fallback_seen:
            _fallback_seen.add(key)
            import warnings
            warnings.warn(f'conv2d_gradfix.{op}: kernel {tuple(weight.shape[2:])}, stride {stride}, padding {padding}, groups {groups}, {input.dtype} has no '
                          f'native gfx950 form -- running torch.nn.functional (MIOpen), as the reference falls back to cuDNN', RuntimeWarning, stacklevel=3)


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else tuple(int(t) for t in v)


def _native_form(input, weight, stride, padding, dilation, groups):
    if not (isinstance(input, torch.Tensor) and input.is_cuda and input.dtype == torch.float32 and weight.dtype == torch.float32):
        return False
    kh, kw = int(weight.shape[2]), int(weight.shape[3])
    return (_pair(stride) == (1, 1) and _pair(dilation) == (1, 1) and groups == 1 and kh == kw and kh in (1, 3, 5)
            and _pair(padding) == (kh // 2, kh // 2))


def conv2d_weight_grad(x, dy, weight_shape, stride=1, padding=0):
    """dw [Cout,Cin,k,k] of y = conv2d(x, w, stride, padding) given dy (conv2d_gradfix.py:141-150), on the matrix cores."""
    _lib.require_cuda(x, 'x')
    x, dy = _lib.f32c(x), _lib.f32c(dy)
    cout, cin, k, k2 = (int(v) for v in weight_shape)
    B, _, H, W = x.shape
    _, _, OH, OW = dy.shape
    if k != k2 or x.shape[1] != cin or dy.shape[1] != cout or dy.shape[0] != B:
        raise RuntimeError(f'conv2d_weight_grad: x {tuple(x.shape)}, dy {tuple(dy.shape)} do not fit a weight of shape {tuple(weight_shape)}')
    dw = torch.empty([cout, cin, k, k], dtype=torch.float32, device=x.device)
    nbytes = int(_lib.load().tdgp_conv2d_weight_grad_workspace_bytes(B, cin, cout, OH, k))
    ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call('tdgp_conv2d_weight_grad', x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), nbytes, B, cin, cout, H, W, OH, OW, k,
                  int(stride), int(padding), _lib.stream_of(x))
    return dw


def conv2d_strided(x, weight, bias=None, stride=1, padding=0):
    """Plain convolution (correlation) on tdgp_conv2d: k in {1,3}, stride in {1,2}, any padding -- the forms outside the fused
    generator kernels (adjoint of the x2 transposed convolution, down-sampling convolutions)."""
    _lib.require_cuda(x, 'x')
    x, w = _lib.f32c(x), _lib.f32c(weight.detach())
    cout, cin, k, _ = (int(v) for v in w.shape)
    B, _, H, W = x.shape
    OH, OW = (H + 2 * padding - k) // stride + 1, (W + 2 * padding - k) // stride + 1
    y = torch.empty([B, cout, OH, OW], dtype=torch.float32, device=x.device)
    b = None if bias is None else _lib.f32c(bias.detach())
    with torch.cuda.device(x.device):
        _lib.call('tdgp_conv2d', x.data_ptr(), w.data_ptr(), _lib.ptr(b), y.data_ptr(), B, cin, cout, H, W, OH, OW, k, int(stride), int(padding),
                  _lib.stream_of(x))
    return y


class _Conv2dGradWeight(torch.autograd.Function):
    """conv2d_gradfix.py:141-166: the weight gradient as a function of (grad_output, input) with its own backward, so that
    second-order terms (R1 regularisation differentiates d logits / d img w.r.t. the weights) flow through the same kernels:
        d / d grad_output = conv2d(input, gg_weight)            d / d input = the transposed convolution of grad_output with gg_weight"""

    @staticmethod
    def forward(ctx, grad_output, input, weight_shape, stride, padding):
        ctx.save_for_backward(grad_output, input)
        ctx.weight_shape, ctx.stride, ctx.padding = tuple(weight_shape), stride, padding
        return conv2d_weight_grad(input, grad_output, weight_shape, stride=stride, padding=padding)

    @staticmethod
    def backward(ctx, gg_weight):
        grad_output, input = ctx.saved_tensors
        gg_grad_output = gg_input = None
        if ctx.needs_input_grad[0]:
            gg_grad_output = conv2d(input, gg_weight, None, stride=ctx.stride, padding=ctx.padding)
        if ctx.needs_input_grad[1]:
            gg_input = _input_grad(grad_output, gg_weight, input.shape[2:], ctx.stride, ctx.padding)
        return gg_grad_output, gg_input, None, None, None


def conv_transpose2d_x2(x, weight):
    """y [B,I,2h+1,2w+1] = conv_transpose2d(x [B,O,h,w], weight [O,I,3,3], stride 2): tdgp_conv_transpose2d_x2 (the polyphase MFMA kernel
    of the up-sampling synthesis layers without its FIR pass)."""
    _lib.require_cuda(x, 'x')
    x = _lib.f32c(x)
    B, O, h, w = x.shape
    I = int(weight.shape[1])
    packed = _modconv.PackedConv(weight.detach().transpose(0, 1).contiguous())           # [I,O,3,3]: "Cout" of the transposed op = I
    y = torch.empty([B, I, 2 * h + 1, 2 * w + 1], dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nbytes = int(lib.tdgp_modconv2d_workspace_bytes(B, O, I, h, w, 3, 2))
    ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call('tdgp_conv_transpose2d_x2', x.data_ptr(), packed.buf.data_ptr(), None, y.data_ptr(), B, O, I, h, w, ws.data_ptr(), nbytes, _lib.stream_of(x))
    return y


class _ConvTranspose2dX2(torch.autograd.Function):
    """The transposed 3x3 stride-2 convolution as a differentiable function: its adjoint pair is conv2d(., w, stride 2) and the weight
    gradient with the roles of input and output gradient swapped."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return conv_transpose2d_x2(x, weight)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = conv2d(g, weight, None, stride=2, padding=0)
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            gw = _Conv2dGradWeight.apply(x, g, weight.shape, 2, 0)
        return gx, gw


def _input_grad(grad_output, weight, input_hw, stride, padding):
    """Input gradient of conv2d(x, weight, stride, padding), itself differentiable w.r.t. grad_output and weight: expressed through
    `conv2d` again (flip / transpose / zero-stuffing are eager tensor ops autograd already knows)."""
    k = int(weight.shape[2])
    wt = weight.flip([2, 3]).transpose(0, 1)
    if stride == 1 and padding == k // 2:
        return conv2d(grad_output, wt, None, stride=1, padding=padding)
    H, W = input_hw
    B, cout, OH, OW = grad_output.shape
    if stride == 2 and k == 3 and padding == 0 and H == 2 * OH + 1 and W == 2 * OW + 1 and grad_output.is_cuda:
        return _ConvTranspose2dX2.apply(grad_output, weight)              # no zero-stuffing: a quarter of the multiplies
    LH, LW = stride * (OH - 1) + 1, stride * (OW - 1) + 1
    lo = k - 1 - padding
    d = grad_output.new_zeros([B, cout, lo + LH + (H - LH + padding), lo + LW + (W - LW + padding)])
    d[:, :, lo:lo + LH:stride, lo:lo + LW:stride] = grad_output
    return conv2d(d, wt, None, stride=1, padding=0)


def conv2d_input_grad(dy, weight):
    """dx of a stride-1 'same' convolution: the correlation of dy with the spatially flipped, in/out-transposed weights."""
    wt = weight.detach().flip([2, 3]).transpose(0, 1).contiguous()
    return _modconv.modconv_forward(dy, _modconv.PackedConv(wt), None, demodulate=False, act='linear', gain=1.0)


class _Conv2dSame(torch.autograd.Function):
    """conv2d_gradfix.py:106-139 for the native form."""

    @staticmethod
    def forward(ctx, input, weight, bias):
        ctx.save_for_backward(input, weight)
        ctx.has_bias = bias is not None
        packed = _modconv._packed(weight) if isinstance(weight, torch.nn.Parameter) else _modconv.PackedConv(weight.contiguous())
        return _modconv.modconv_forward(input, packed, None, bias=bias, demodulate=False, act='linear', gain=1.0)

    @staticmethod
    def backward(ctx, grad_output):
        input, weight = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        grad_input = grad_weight = grad_bias = None
        if ctx.needs_input_grad[0]:
            grad_input = _input_grad(grad_output, weight, input.shape[2:], 1, weight.shape[2] // 2)
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            grad_weight = _Conv2dGradWeight.apply(grad_output, input, weight.shape, 1, weight.shape[2] // 2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_bias = grad_output.sum([0, 2, 3])
        return grad_input, grad_weight, grad_bias


def _strided_form(input, weight, stride, padding, dilation, groups):
    """The second native form: k in {1,3}, stride in {1,2}, padding <= k - 1 (the discriminator's down-sampling convolutions)."""
    if not (isinstance(input, torch.Tensor) and input.is_cuda and input.dtype == torch.float32 and weight.dtype == torch.float32):
        return False
    kh, kw = int(weight.shape[2]), int(weight.shape[3])
    st, pd = _pair(stride), _pair(padding)
    return (st[0] == st[1] and st[0] in (1, 2) and _pair(dilation) == (1, 1) and groups == 1 and kh == kw and kh in (1, 3) and pd[0] == pd[1]
            and pd[0] <= kh - 1)


def conv2d_strided_input_grad(dy, weight, input_hw, stride, padding):
    """dx of y = conv2d(x, w, stride, padding) (conv2d_gradfix.py:126-129: the transposed convolution of dy): dy is zero-stuffed by
    the stride, padded by k - 1 - padding on the left / top (and whatever completes the input size on the right / bottom), and
    correlated with the flipped, in/out-transposed weights -- one tdgp_conv2d call."""
    k = int(weight.shape[2])
    H, W = input_hw
    B, cout, OH, OW = dy.shape
    LH, LW = stride * (OH - 1) + 1, stride * (OW - 1) + 1
    lo = k - 1 - padding
    d = torch.zeros([B, cout, lo + LH + (H - LH + padding), lo + LW + (W - LW + padding)], dtype=dy.dtype, device=dy.device)
    d[:, :, lo:lo + LH:stride, lo:lo + LW:stride] = dy
    wt = weight.detach().flip([2, 3]).transpose(0, 1).contiguous()
    return conv2d_strided(d, wt, stride=1, padding=0)


class _Conv2dStrided(torch.autograd.Function):
    """conv2d_gradfix.py:106-139 for the strided form."""

    @staticmethod
    def forward(ctx, input, weight, bias, stride, padding):
        ctx.save_for_backward(input, weight)
        ctx.has_bias, ctx.stride, ctx.padding = bias is not None, stride, padding
        return conv2d_strided(input, weight, bias, stride=stride, padding=padding)

    @staticmethod
    def backward(ctx, grad_output):
        input, weight = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        grad_input = grad_weight = grad_bias = None
        if ctx.needs_input_grad[0]:
            grad_input = _input_grad(grad_output, weight, input.shape[2:], ctx.stride, ctx.padding)
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            grad_weight = _Conv2dGradWeight.apply(grad_output, input, weight.shape, ctx.stride, ctx.padding)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_bias = grad_output.sum([0, 2, 3])
        return grad_input, grad_weight, grad_bias, None, None


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    if _native_form(input, weight, stride, padding, dilation, groups):
        return _Conv2dSame.apply(input, weight, bias)
    if _strided_form(input, weight, stride, padding, dilation, groups):
        return _Conv2dStrided.apply(input, weight, bias, _pair(stride)[0], _pair(padding)[0])
    _note_fallback('conv2d', input, weight, stride, padding, groups)
    return torch.nn.functional.conv2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding, dilation=dilation, groups=groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    """The reference's fallback form (conv2d_gradfix.py:41-44); the generator's x2 transposed convolution runs fused inside tdgp_modconv2d."""
    _note_fallback('conv_transpose2d', input, weight, stride, padding, groups)
    return torch.nn.functional.conv_transpose2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                                                output_padding=output_padding, groups=groups, dilation=dilation)
