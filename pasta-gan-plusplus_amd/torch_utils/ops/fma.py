"""``fma(a, b, c) = a * b + c`` as one fused multiply-add with gradients that respect broadcasting (API of the reference's
torch_utils/ops/fma.py:15).  It exists for ``x * dcoefs + noise`` after a modulated convolution (networks.py:77); on the
inference route that step lives in the convolution's epilogue (csrc/conv2d_kernel.h, `out_scale` / `noise`), so this
standalone form only runs on the differentiable route."""

import torch


class _MulAdd(torch.autograd.Function):
    """out = addcmul(c, a, b).  Each gradient is the product rule's term reduced back to the operand's own shape with
    ``Tensor.sum_to_size`` (the inverse of broadcasting); the backward is built from differentiable ops, so higher-order
    gradients (R1 differentiates through the generator-side fma) come from autograd itself."""

    @staticmethod
    def forward(ctx, a, b, c):
        ctx.save_for_backward(a, b)
        ctx.shapes = (a.shape, b.shape, c.shape)
        return torch.addcmul(c, a, b)

    @staticmethod
    def backward(ctx, grad):
        a, b = ctx.saved_tensors
        sa, sb, sc = ctx.shapes
        need_a, need_b, need_c = ctx.needs_input_grad
        return ((grad * b).sum_to_size(sa) if need_a else None,
                (grad * a).sum_to_size(sb) if need_b else None,
                grad.sum_to_size(sc) if need_c else None)


def fma(a, b, c):
    return _MulAdd.apply(a, b, c)
