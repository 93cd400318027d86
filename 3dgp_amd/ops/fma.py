"""fma(a, b, c): the broadcasting multiply-add `a * b + c` of the unfused modulated convolution (`x * dcoefs + noise`,
networks_stylegan2.py:84-85; behaviour of src/torch_utils/ops/fma.py): one `addcmul` forward; the backward hands every operand the
product-rule term reduced back to that operand's own shape, so broadcast operands ([B,C,1,1] coefficients, [1,1,H,W] noise) receive
gradients of their shape and a second differentiation goes through plain tensor ops."""
import torch


class _MulAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        ctx.save_for_backward(a, b)
        ctx.shapes = (a.shape, b.shape, c.shape)
        return torch.addcmul(c, a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        terms = (lambda: g * b, lambda: g * a, lambda: g)
        # Tensor.sum_to_size folds exactly the dimensions broadcasting expanded (leading ones and size-1 axes)
        return tuple(t().sum_to_size(s) if need else None for t, s, need in zip(terms, ctx.shapes, ctx.needs_input_grad))


def fma(a, b, c):
    return _MulAdd.apply(a, b, c)
