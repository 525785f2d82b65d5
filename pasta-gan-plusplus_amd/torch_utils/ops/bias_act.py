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
import os

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
        ws = plugin.lib.pg_bias_act_grad_bias_workspace
        ws.restype = ctypes.c_int64
        ws.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int64]
        gb = plugin.lib.pg_bias_act_grad_bias
        gb.restype = ctypes.c_int
        gb.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                                               ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
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


_FUSED_DB_ACTS = ('linear', 'relu', 'lrelu')
fused_bias_gradient = os.environ.get('PG_FUSED_DB', '1') != '0'     # dx and db in one pass over dy (PG_FUSED_DB=0: the reference's dx.sum())


def _native_grad_bias(dy, y, dim, act, alpha, gain, clamp, write=True):
    """(dx, db) of the first-derivative form in ONE pass over dy (csrc/bias_act.hip, pg_bias_act_grad_bias): dx as grad == 1 of
    `_native_bias_act`, db = dx summed over every axis but `dim` (fixed-order, deterministic).  `write=False`: db only (dx = None).
    Returns None where the layout / activation is not covered -- the caller then composes the two steps like the reference
    (bias_act.py:176-186)."""
    if not fused_bias_gradient or act not in _FUSED_DB_ACTS or dy.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        return None
    if not dy.is_cuda or dy.numel() == 0 or not nat.is_dense(dy) or (y is not None and (y.dtype != dy.dtype or not _same_layout(y, dy) or not nat.is_dense(y))):
        return None
    _init()
    c = int(dy.shape[dim])
    if dy.is_contiguous():                           # the kernel's channel of element i is (i / step) % c
        step = int(np.prod(dy.shape[dim + 1:], dtype=np.int64))
    elif dy.ndim == 4 and dim == 1 and dy.is_contiguous(memory_format=torch.channels_last):
        step = 1
    else:
        return None
    nbytes = _plugin.lib.pg_bias_act_grad_bias_workspace(nat.PG_DTYPE[dy.dtype], dy.numel(), c, step)
    if nbytes <= 0:
        return None
    work = torch.empty([nbytes // 4], dtype=torch.float32, device=dy.device)
    dx = torch.empty_like(dy) if write else None
    db = torch.empty([c], dtype=dy.dtype, device=dy.device)
    with torch.cuda.device(dy.device):
        st = _plugin.lib.pg_bias_act_grad_bias(nat.ptr(dy), nat.ptr(y), nat.ptr(dx), nat.ptr(db), nat.ptr(work), nbytes,
                                               nat.PG_DTYPE[dy.dtype], dy.numel(), c, step, activation_funcs[act].cuda_idx,
                                               float(alpha), float(gain), float(clamp), nat.stream_of(dy))
    if st == -2:                                      # PG_ERR_UNSUPPORTED (alignment): compose
        return None
    nat.check(st, 'pg_bias_act_grad_bias')
    return dx, db


def channel_sum(t, dim=1):
    """t summed over every axis but `dim` -- the bias gradient of a convolution / bias_act (`dy.sum([0, 2, 3])`): one deterministic native pass
    where covered and nothing asks for the sum's own gradient, torch otherwise."""
    if t.is_cuda and not (torch.is_grad_enabled() and t.requires_grad):
        out = _native_grad_bias(_dense(t), None, dim, 'linear', 0.0, 1.0, -1.0, write=False)
        if out is not None:
            return out[1]
    return t.sum([i for i in range(t.ndim) if i != dim])


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
                if ctx.needs_input_grad[1]:          # dx and its channel sums from one pass over dy
                    dx, db = _BiasActGrad.apply(dy, x, b, y, dim, act, alpha, gain, clamp, True)
                else:
                    dx = _BiasActGrad.apply(dy, x, b, y, dim, act, alpha, gain, clamp)
        if ctx.needs_input_grad[1] and db is None:
            db = channel_sum(dx, dim)
        return dx, db, None, None, None, None, None


class _BiasActGrad(torch.autograd.Function):
    """dx = dy * act'(.) * gain (clamp-masked); itself differentiable (R1 needs d/d(dy), bias_act.py:197-198)."""

    @staticmethod
    def forward(ctx, dy, x, b, y, dim, act, alpha, gain, clamp, with_db=False):
        """`with_db` (private): also return db = dx summed over every axis but `dim`, gathered in the same pass where the kernel covers it."""
        spec = activation_funcs[act]
        ref = y if y is not None else x
        dy = _dense(dy)
        if ref is not None and not _same_layout(dy, ref):    # match the saved tensors' layout (bias_act.py:160-162)
            dy = torch.empty_like(ref).copy_(dy)
        fused = _native_grad_bias(dy, y, dim, act, alpha, gain, clamp) if (with_db and x is None) else None
        if fused is not None:
            dx, db = fused
        else:
            dx = _native_bias_act(dy, b, x, y, None, 1, dim, spec.cuda_idx, alpha, gain, clamp)
            db = channel_sum(dx.detach(), dim) if with_db else None
        ctx.save_for_backward(dy if spec.has_2nd_grad else None, x, b, y)
        ctx.cfg = (dim, act, alpha, gain, clamp)
        ctx.dx_shape = tuple(dx.shape)              # (backward may see d_db alone while neither x nor y was saved)
        #  no zero-filled gradient for an output nobody consumed: with the default materialisation every R1 double backward handed `backward`
        # a zero d_db and paid a full-size `d_dx + d_db.reshape(...)` pass per layer for it
        ctx.set_materialize_grads(False)
        return (dx, db) if with_db else dx

    @staticmethod
    def backward(ctx, d_dx, d_db=None):
        dim, act, alpha, gain, clamp = ctx.cfg
        spec = activation_funcs[act]
        dy, x, b, y = ctx.saved_tensors
        if d_dx is None and d_db is None:
            return (None,) * 10
        if d_db is not None:                                 # db = sum(dx): its gradient is spread back over dx
            shape = [1] * len(ctx.dx_shape)
            shape[dim] = -1
            d_dx = d_db.reshape(shape).expand(ctx.dx_shape) if d_dx is None else d_dx + d_db.reshape(shape)
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
        return d_dy, d_x, d_b, None, None, None, None, None, None, None
