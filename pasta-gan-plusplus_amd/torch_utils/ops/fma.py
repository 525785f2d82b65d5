"""a * b + c with broadcast-aware gradients (reference torch_utils/ops/fma.py:15-58).
Stays a PyTorch-ROCm op here; on the synthesis path it is folded into the conv epilogue
(csrc/conv2d_kernel.h, `out_scale` / `noise`) and this standalone form is not called."""

import torch


def fma(a, b, c):  # => a * b + c
    return _Fma.apply(a, b, c)


def _sum_to_shape(t, shape):
    """Reduce a broadcast result back to `shape`."""
    lead = t.ndim - len(shape)
    assert lead >= 0
    dims = [i for i in range(t.ndim) if t.shape[i] > 1 and (i < lead or shape[i - lead] == 1)]
    if dims:
        t = t.sum(dim=dims, keepdim=True)
    if lead:
        t = t.reshape(-1, *t.shape[lead + 1:])
    assert t.shape == shape
    return t


class _Fma(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        ctx.save_for_backward(a, b)
        ctx.c_shape = c.shape
        return torch.addcmul(c, a, b)

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        da = _sum_to_shape(dout * b, a.shape) if ctx.needs_input_grad[0] else None
        db = _sum_to_shape(dout * a, b.shape) if ctx.needs_input_grad[1] else None
        dc = _sum_to_shape(dout, ctx.c_shape) if ctx.needs_input_grad[2] else None
        return da, db, dc
