"""fma(a, b, c) = a * b + c with the reference's hand-written gradients (src/torch_utils/ops/fma.py:17-60): used by the unfused
modulated convolution for `x * dcoefs + noise` (networks_stylegan2.py:84-85).  Eager tensor arithmetic in the reference too."""
import torch


def fma(a, b, c):
    return _FusedMultiplyAdd.apply(a, b, c)


class _FusedMultiplyAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        out = torch.addcmul(c, a, b)
        ctx.save_for_backward(a, b)
        ctx.c_shape = c.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        da = _unbroadcast(dout * b, a.shape) if ctx.needs_input_grad[0] else None
        db = _unbroadcast(dout * a, b.shape) if ctx.needs_input_grad[1] else None
        dc = _unbroadcast(dout, ctx.c_shape) if ctx.needs_input_grad[2] else None
        return da, db, dc


def _unbroadcast(x, shape):
    """Sum `x` over the dimensions that were broadcast to reach its shape from `shape` (fma.py:50-60)."""
    extra = x.ndim - len(shape)
    assert extra >= 0
    dims = [i for i in range(x.ndim) if x.shape[i] > 1 and (i < extra or shape[i - extra] == 1)]
    if dims:
        x = x.sum(dim=dims, keepdim=True)
    if extra:
        x = x.reshape(-1, *x.shape[extra + 1:])
    assert x.shape == shape
    return x
