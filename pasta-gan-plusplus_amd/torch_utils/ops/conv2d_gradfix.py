"""conv2d / conv_transpose2d entry points of the synthesis path on MI355X.

API of the reference's ``torch_utils/ops/conv2d_gradfix.py``: ``conv2d`` (:35-38),
``conv_transpose2d`` (:40-43), ``no_weight_gradients()`` (:25-31) and the module globals
``enabled`` / ``weight_gradients_disabled`` (:22-23).  In the reference this is where all
conv arithmetic is handed to cuDNN; here float32 NCHW convolutions of the geometries the
generator uses run the hand-written MFMA implicit-GEMM kernel (``conv2d_mfma`` ->
``csrc/conv2d_kernel.h``).  A stride-2 transposed conv is issued as one gather-form
sub-convolution per output phase, so no multiply is spent on stuffed zeros.

Gradients (first and higher order, honouring ``no_weight_gradients``) come from
``aten::convolution_backward`` on the same GPU -- backward kernels are a later row of
SURVEY.md section 8.  Configurations the MFMA kernel does not cover (groups > 1, dilation,
fp16, exotic kernel sizes) go to PyTorch-ROCm's convolution on the GPU.  CPU tensors raise:
the product has no CPU path.
"""

import contextlib

import torch

from . import _native as nat
from . import conv2d_mfma

enabled = False                     # kept for API compatibility (training_loop_fullbody.py:386); the native path is always on
weight_gradients_disabled = False   # forcefully disable weight gradients (R1, loss_fullbody.py:266)


@contextlib.contextmanager
def no_weight_gradients():
    global weight_gradients_disabled
    old = weight_gradients_disabled
    weight_gradients_disabled = True
    try:
        yield
    finally:
        weight_gradients_disabled = old


def _pair(v):
    if isinstance(v, (tuple, list)):
        assert len(v) == 2
        return int(v[0]), int(v[1])
    return int(v), int(v)


def _native_ok(x, w, stride, dilation, groups):
    return (x.dtype == torch.float32 and w.dtype == torch.float32 and x.ndim == 4 and groups == 1
            and dilation == (1, 1) and stride[0] == stride[1])


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    assert isinstance(input, torch.Tensor)
    nat.require_gpu(input, 'conv2d_gradfix.conv2d')
    stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
    kh, kw = int(weight.shape[2]), int(weight.shape[3])
    if _native_ok(input, weight, stride, dilation, groups) and conv2d_mfma.supported(kh, kw, stride[0]) and min(padding) >= 0:
        return _Conv2dMfma.apply(input, weight, bias, stride[0], padding, False, (0, 0))
    return torch.nn.functional.conv2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding, dilation=dilation, groups=groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    assert isinstance(input, torch.Tensor)
    nat.require_gpu(input, 'conv2d_gradfix.conv_transpose2d')
    stride, padding, dilation, output_padding = _pair(stride), _pair(padding), _pair(dilation), _pair(output_padding)
    kh, kw = int(weight.shape[2]), int(weight.shape[3])
    if (_native_ok(input, weight, stride, dilation, groups) and stride[0] > 1 and kh >= stride[0] and kw >= stride[0]
            and all(conv2d_mfma.supported(jy, jx, 1) for jy in {-(-(kh - a) // stride[0]) for a in range(stride[0])}
                    for jx in {-(-(kw - a) // stride[0]) for a in range(stride[0])})):
        return _Conv2dMfma.apply(input, weight, bias, stride[0], padding, True, output_padding)
    return torch.nn.functional.conv_transpose2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                                                output_padding=output_padding, groups=groups, dilation=dilation)


class _Conv2dMfma(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, transposed, output_padding):
        x = x.contiguous()
        n, cin, h, w = x.shape
        kh, kw = int(weight.shape[2]), int(weight.shape[3])
        if not transposed:
            cout = int(weight.shape[0])
            packed = conv2d_mfma.pack_weight(weight)
            y = conv2d_mfma.conv2d_forward(x, packed, cout, kh, kw, stride=stride, pad=padding, bias=bias)
        else:
            cout = int(weight.shape[1])
            out_hw = ((h - 1) * stride - 2 * padding[0] + kh + output_padding[0], (w - 1) * stride - 2 * padding[1] + kw + output_padding[1])
            phases = conv2d_mfma.pack_transposed(weight, stride, padding, (h, w), out_hw)
            y = conv2d_mfma.conv_transpose2d_forward(x, phases, cout, out_hw, stride=stride, bias=bias)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, padding, transposed, output_padding, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, padding, transposed, output_padding, has_bias = ctx.cfg
        want_w = ctx.needs_input_grad[1] and not weight_gradients_disabled
        mask = [ctx.needs_input_grad[0], want_w, has_bias and ctx.needs_input_grad[2]]
        dx, dw, db = torch.ops.aten.convolution_backward(
            dy, x, weight, [weight.shape[1 if transposed else 0]] if has_bias else None,
            [stride, stride], list(padding), [1, 1], transposed, list(output_padding), 1, mask)
        return dx, dw, db, None, None, None, None
