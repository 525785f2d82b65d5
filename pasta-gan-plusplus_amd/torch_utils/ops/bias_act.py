"""Fused bias + activation + gain + clamp on MI355X.

Same public surface as the reference's ``torch_utils/ops/bias_act.py``
(``bias_act(x, b, dim, act, alpha, gain, clamp, impl)`` :55-89 and the
``activation_funcs`` table :23-33), differentiable to second order like
``BiasActCuda``/``BiasActCudaGrad`` (:129-210).  Dispatch is the reference's rule (:86-89): GPU tensors with
``impl='cuda'`` run the hand-written HIP kernel ``csrc/bias_act.hip`` through the C ABI ``pg_bias_act``
(include/pasta_gan_ops.h) -- and fail loudly if that library is missing; ``impl='ref'`` or a CPU tensor takes the
plain-torch composition ``_bias_act_torch`` below (autograd differentiates it).
"""

import ctypes

import numpy as np
import torch

import dnnlib

from .. import custom_ops
from . import _native as nat

# ----------------------------------------------------------------------------
# name -> func (plain torch expression of the activation, informational), default alpha/gain,
# plugin index, which of x / y the backward needs, and whether a 2nd derivative exists.

activation_funcs = {
    'linear':   dnnlib.EasyDict(func=lambda x, **_: x,                                      def_alpha=0,   def_gain=1,          cuda_idx=1, ref='',  has_2nd_grad=False),
    'relu':     dnnlib.EasyDict(func=lambda x, **_: torch.relu(x),                          def_alpha=0,   def_gain=np.sqrt(2), cuda_idx=2, ref='y', has_2nd_grad=False),
    'lrelu':    dnnlib.EasyDict(func=lambda x, alpha, **_: torch.where(x > 0, x, x * alpha), def_alpha=0.2, def_gain=np.sqrt(2), cuda_idx=3, ref='y', has_2nd_grad=False),
    'tanh':     dnnlib.EasyDict(func=lambda x, **_: torch.tanh(x),                          def_alpha=0,   def_gain=1,          cuda_idx=4, ref='y', has_2nd_grad=True),
    'sigmoid':  dnnlib.EasyDict(func=lambda x, **_: torch.sigmoid(x),                       def_alpha=0,   def_gain=1,          cuda_idx=5, ref='y', has_2nd_grad=True),
    'elu':      dnnlib.EasyDict(func=lambda x, **_: torch.where(x >= 0, x, torch.expm1(x)), def_alpha=0,   def_gain=1,          cuda_idx=6, ref='y', has_2nd_grad=True),
    'selu':     dnnlib.EasyDict(func=lambda x, **_: torch.selu(x),                          def_alpha=0,   def_gain=1,          cuda_idx=7, ref='y', has_2nd_grad=True),
    'softplus': dnnlib.EasyDict(func=lambda x, **_: torch.log1p(torch.exp(-x.abs())) + x.clamp(min=0), def_alpha=0, def_gain=1, cuda_idx=8, ref='y', has_2nd_grad=True),
    'swish':    dnnlib.EasyDict(func=lambda x, **_: x * torch.sigmoid(x),                   def_alpha=0,   def_gain=np.sqrt(2), cuda_idx=9, ref='x', has_2nd_grad=True),
}

# ----------------------------------------------------------------------------

_plugin = None


def _init():
    """Load (building if stale) ``bias_act_plugin``; raises if that is impossible."""
    global _plugin
    if _plugin is None:
        plugin = custom_ops.get_plugin('bias_act_plugin')
        fn = plugin.lib.pg_bias_act
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
        _plugin = plugin
    return True


def _same_layout(a, b):
    return a.shape == b.shape and all(sa == sb for n, sa, sb in zip(a.shape, a.stride(), b.stride()) if n >= 2)


def _native_bias_act(x, b, xref, yref, dy, grad, dim, act_idx, alpha, gain, clamp):
    """Tensor-level twin of the reference plugin entry ``bias_act(...)`` (bias_act.cpp:32-90):
    `None` stands where the reference passes an empty tensor; returns a new tensor laid out like x."""
    _init()
    if x.dtype not in nat.PG_DTYPE:
        raise nat.NativeOpError(f'bias_act: unsupported dtype {x.dtype}')
    for name, t in (('xref', xref), ('yref', yref), ('dy', dy)):
        if t is not None and (t.dtype != x.dtype or t.device != x.device or not _same_layout(t, x)):
            raise nat.NativeOpError(f'bias_act: {name} must have the same shape, dtype, device and layout as x')
    if not nat.is_dense(x):
        raise nat.NativeOpError('bias_act: x must be non-overlapping and dense')
    if b is not None:
        if b.ndim != 1 or b.dtype != x.dtype or b.device != x.device or not b.is_contiguous():
            raise nat.NativeOpError('bias_act: b must be a contiguous rank-1 tensor with the dtype and device of x')
        if not (0 <= dim < x.ndim) or b.numel() != x.shape[dim]:
            raise nat.NativeOpError('bias_act: b has the wrong number of elements for dimension `dim`')
    y = torch.empty_like(x)     # preserves x's (dense) strides
    if x.numel() == 0:
        return y
    size_b = b.numel() if b is not None else 0
    step_b = x.stride(dim) if b is not None else 1
    with torch.cuda.device(x.device):
        st = _plugin.lib.pg_bias_act(nat.ptr(x), nat.ptr(b), nat.ptr(xref), nat.ptr(yref), nat.ptr(dy), nat.ptr(y),
                                     nat.PG_DTYPE[x.dtype], x.numel(), size_b, max(int(step_b), 1),
                                     int(grad), int(act_idx), float(alpha), float(gain), float(clamp), nat.stream_of(x))
    nat.check(st, 'pg_bias_act')
    return y


# ----------------------------------------------------------------------------

def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    r"""y = clamp(act(x + b) * gain); arguments as in the reference (bias_act.py:55-84).

    ``impl='cuda'`` on a GPU tensor = the HIP kernel; ``impl='ref'`` or a CPU tensor = the plain-torch composition.
    """
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    assert clamp is None or clamp >= 0
    spec = activation_funcs[act]
    alpha = float(alpha if alpha is not None else spec.def_alpha)
    gain = float(gain if gain is not None else spec.def_gain)
    clamp = float(clamp if clamp is not None else -1)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.ndim == 1
        assert 0 <= dim < x.ndim
        assert b.shape[0] == x.shape[dim]
    if impl == 'cuda' and x.device.type == 'cuda':
        return _BiasAct.apply(x, b, dim, act, alpha, gain, clamp)
    return _bias_act_torch(x, b, dim, act, alpha, gain, clamp)


def _bias_act_torch(x, b, dim, act, alpha, gain, clamp):
    """The op as four elementwise torch steps, in the order the kernel applies them (bias_act.hip; reference
    `_bias_act_ref`, bias_act.py:93-123): add the bias along `dim`, activate, scale, clamp (clamp < 0 = off)."""
    if b is not None:
        shape = [1] * x.ndim
        shape[dim] = -1
        x = x + b.reshape(shape)
    y = activation_funcs[act].func(x, alpha=alpha)
    if gain != 1:
        y = y * gain
    return y.clamp(-clamp, clamp) if clamp >= 0 else y


def _dense(t):
    """Contiguous in its own memory format (channels_last kept, like bias_act.py:148)."""
    if nat.is_dense(t):
        return t
    fmt = torch.channels_last if t.ndim == 4 and t.stride(1) == 1 else torch.contiguous_format
    return t.contiguous(memory_format=fmt)


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b, dim, act, alpha, gain, clamp):
        spec = activation_funcs[act]
        x = _dense(x)
        b = b.contiguous() if b is not None else None
        trivial = act == 'linear' and gain == 1 and clamp < 0 and b is None
        y = x if trivial else _native_bias_act(x, b, None, None, None, 0, dim, spec.cuda_idx, alpha, gain, clamp)
        need_x = 'x' in spec.ref or spec.has_2nd_grad
        # 'linear' saves nothing in the reference (bias_act.py:154-157), which makes its backward ignore an
        # active clamp; keeping y in that one case gives the true (clamp-masked) gradient instead.
        need_y = 'y' in spec.ref or (act == 'linear' and clamp >= 0)
        ctx.save_for_backward(x if need_x else None, b if need_x else None, y if need_y else None)
        ctx.cfg = (dim, act, alpha, gain, clamp)
        return y

    @staticmethod
    def backward(ctx, dy):
        dim, act, alpha, gain, clamp = ctx.cfg
        x, b, y = ctx.saved_tensors
        dx = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dx = dy
            if act != 'linear' or gain != 1 or clamp >= 0:
                dx = _BiasActGrad.apply(dy, x, b, y, dim, act, alpha, gain, clamp)
        if ctx.needs_input_grad[1]:
            db = dx.sum([i for i in range(dx.ndim) if i != dim])
        return dx, db, None, None, None, None, None


class _BiasActGrad(torch.autograd.Function):
    """dx = dy * act'(.) * gain (clamp-masked); itself differentiable (R1 needs d/d(dy), bias_act.py:197-198)."""

    @staticmethod
    def forward(ctx, dy, x, b, y, dim, act, alpha, gain, clamp):
        spec = activation_funcs[act]
        ref = y if y is not None else x
        dy = _dense(dy)
        if ref is not None and not _same_layout(dy, ref):    # match the saved tensors' layout (bias_act.py:160-162)
            dy = torch.empty_like(ref).copy_(dy)
        dx = _native_bias_act(dy, b, x, y, None, 1, dim, spec.cuda_idx, alpha, gain, clamp)
        ctx.save_for_backward(dy if spec.has_2nd_grad else None, x, b, y)
        ctx.cfg = (dim, act, alpha, gain, clamp)
        return dx

    @staticmethod
    def backward(ctx, d_dx):
        dim, act, alpha, gain, clamp = ctx.cfg
        spec = activation_funcs[act]
        dy, x, b, y = ctx.saved_tensors
        d_dy = d_x = d_b = None
        if ctx.needs_input_grad[0]:
            d_dy = _BiasActGrad.apply(d_dx, x, b, y, dim, act, alpha, gain, clamp)
        if spec.has_2nd_grad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            ref = y if y is not None else x
            d_dx_c = _dense(d_dx)
            if not _same_layout(d_dx_c, ref):
                d_dx_c = torch.empty_like(ref).copy_(d_dx_c)
            d_x = _native_bias_act(d_dx_c, b, x, y, dy, 2, dim, spec.cuda_idx, alpha, gain, clamp)
        if spec.has_2nd_grad and ctx.needs_input_grad[2]:
            d_b = d_x.sum([i for i in range(d_x.ndim) if i != dim])
        return d_dy, d_x, d_b, None, None, None, None, None, None
