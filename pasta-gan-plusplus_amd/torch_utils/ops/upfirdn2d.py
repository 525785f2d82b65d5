"""Pad / zero-stuff / FIR / decimate for batches of 2-D images on MI355X.

Public surface of the reference's ``torch_utils/ops/upfirdn2d.py``: ``setup_filter``
(:72-116), ``upfirdn2d`` (:120-164), ``filter2d`` (:272-304), ``upsample2d``
(:308-343), ``downsample2d`` (:347-382) and the private helpers its siblings import
(``_parse_scaling``, ``_parse_padding``, ``_get_filter_size``, :37-68).  Every
GPU evaluation runs ``csrc/upfirdn2d.hip`` through the C ABI ``pg_upfirdn2d``; the
gradient is another upfirdn2d with the factors swapped (:245-264), so arbitrary
order derivatives work.  Dispatch is the reference's rule (:161-164): ``impl='cuda'`` on a
GPU tensor = the HIP kernels (failing loudly if the library is missing); ``impl='ref'`` or
a CPU tensor = the plain-torch composition ``_upfirdn2d_torch``.
"""

import ctypes

import numpy as np
import torch

from .. import custom_ops
from .. import misc
from . import _native as nat

_plugin = None


def _init():
    global _plugin
    if _plugin is None:
        plugin = custom_ops.get_plugin('upfirdn2d_plugin')
        fn = plugin.lib.pg_upfirdn2d
        fn.restype = ctypes.c_int
        i, p64 = ctypes.c_int, ctypes.POINTER(ctypes.c_int64)
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, i, i, i, i, i, p64, i, i, p64, i, i, p64,
                       i, i, i, i, i, i, i, ctypes.c_float, ctypes.c_void_p]
        fn2 = plugin.lib.pg_upfirdn2d_bias_act
        fn2.restype = ctypes.c_int
        fn2.argtypes = fn.argtypes[:-1] + [ctypes.POINTER(_FirEpilogue), ctypes.c_void_p]
        fn3 = plugin.lib.pg_upfirdn2d_with_odd_samples
        fn3.restype = ctypes.c_int
        fn3.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, i, i, i, i, i, p64, i, i, p64, i, i, p64, i, i, i, ctypes.c_float, ctypes.c_void_p]
        _plugin = plugin
    return True


class _FirEpilogue(ctypes.Structure):
    """Mirror of ``pg_fir_epilogue`` (include/pasta_gan_ops.h)."""
    _fields_ = [('noise', ctypes.c_void_p), ('noise_batch_stride', ctypes.c_int64), ('noise_gain', ctypes.c_float),
                ('bias', ctypes.c_void_p), ('act', ctypes.c_int), ('alpha', ctypes.c_float), ('act_gain', ctypes.c_float), ('clamp', ctypes.c_float)]


_FUSED_ACTS = {'linear': 1, 'relu': 2, 'lrelu': 3}


def upfirdn2d_bias_act(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, noise=None, b=None, act='linear', alpha=0.2,
                       act_gain=1.0, clamp=None):
    """`bias_act(upfirdn2d(x, f, ...) + noise, b, act, gain=act_gain, clamp)` in ONE pass (the tail of an up-sampling
    SynthesisLayer: FIR -> +noise -> bias_act).  Inference only (no autograd); float32 dense NCHW and a 2-D filter the
    tiled kernel covers, or a channels-last tensor (float32 / fp16 / bf16, no up-sampling, filter up to 4 x 4: the
    channels-last kernel); otherwise returns None and the caller composes the separate ops."""
    nat.require_gpu(x, 'upfirdn2d_bias_act')
    _init()
    if x.ndim != 4 or f is None or f.ndim != 2 or act not in _FUSED_ACTS:
        return None
    channels_last = x.shape[1] > 1 and x.stride(1) == 1 and (x.is_contiguous(memory_format=torch.channels_last) or
                                                              (x.stride(3) == x.shape[1] and x.stride(2) >= x.shape[3] * x.shape[1] and x.stride(0) >= x.shape[2] * x.stride(2)))   # dense NHWC or a [:, :, :H, :W] view of one
    pitched = (x.dtype == torch.float32 and x.stride(3) == 1 and x.stride(2) >= x.shape[3] and x.stride(1) == x.shape[2] * x.stride(2)
               and x.stride(0) == x.shape[1] * x.stride(1))        # dense NCHW, or NCHW with padded rows (conv2d_mfma.conv_up2_forward)
    if not (pitched or (channels_last and x.dtype in (torch.float32, torch.float16, torch.bfloat16))):
        return None
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    n, c, ih, iw = x.shape
    fh, fw = f.shape
    ow = (iw * upx + padx0 + padx1 - fw + downx) // downx
    oh = (ih * upy + pady0 + pady1 - fh + downy) // downy
    assert ow >= 1 and oh >= 1
    y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device,
                    memory_format=torch.channels_last if (channels_last and not x.is_contiguous()) else torch.contiguous_format)
    ep = _FirEpilogue()
    keep = []
    if noise is not None:
        noise = noise.to(torch.float32).contiguous()
        assert noise.numel() in (oh * ow, n * oh * ow)
        ep.noise, ep.noise_batch_stride, ep.noise_gain = noise.data_ptr(), (0 if noise.numel() == oh * ow else oh * ow), 1.0
        keep.append(noise)
    if b is not None:
        b = b.to(torch.float32).contiguous()
        assert b.numel() == c
        ep.bias = b.data_ptr()
        keep.append(b)
    ep.act, ep.alpha, ep.act_gain, ep.clamp = _FUSED_ACTS[act], float(alpha), float(act_gain), (-1.0 if clamp is None else float(clamp))
    with torch.cuda.device(x.device):
        st = _plugin.lib.pg_upfirdn2d_bias_act(nat.ptr(x), nat.ptr(f), nat.ptr(y), nat.PG_DTYPE[x.dtype], n, c, ih, iw, nat.i64arr(x.stride()),
                                               fh, fw, nat.i64arr(f.stride()), oh, ow, nat.i64arr(y.stride()),
                                               upx, upy, downx, downy, padx0, pady0, int(bool(flip_filter)), float(gain), ctypes.byref(ep), nat.stream_of(x))
    if st == -2:
        return None
    nat.check(st, 'pg_upfirdn2d_bias_act')
    return y


# ---------------------------------------------------------------------------- argument algebra

def _parse_scaling(scaling):
    if isinstance(scaling, int):
        scaling = [scaling, scaling]
    assert isinstance(scaling, (list, tuple)) and len(scaling) == 2
    assert all(isinstance(v, int) for v in scaling)
    sx, sy = scaling
    assert sx >= 1 and sy >= 1
    return sx, sy


def _parse_padding(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    assert isinstance(padding, (list, tuple))
    assert all(isinstance(v, int) for v in padding)
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    px0, px1, py0, py1 = padding
    return px0, px1, py0, py1


def _get_filter_size(f):
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    fw, fh = int(f.shape[-1]), int(f.shape[0])
    misc.assert_shape(f, [fh, fw][:f.ndim])
    assert fw >= 1 and fh >= 1
    return fw, fh


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    r"""FIR taps for `upfirdn2d()`: `[fh, fw]`, `[taps]` (separable), `[]` (impulse) or None (identity).
    1-D tap lists shorter than 8 become their outer product unless `separable` says otherwise."""
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32)
    assert f.ndim in [0, 1, 2] and f.numel() > 0
    if f.ndim == 0:
        f = f[np.newaxis]
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)


# ---------------------------------------------------------------------------- native call

def filter_with_odd_samples(x, f, padding, flip_filter=False, gain=1):
    """(y, y_odd): y = upfirdn2d(x, f, padding=padding) and y_odd = y[:, :, 1::2, 1::2] as a dense tensor, from ONE pass over x (round 6,
    pg_upfirdn2d_with_odd_samples).  The two FIR calls of a `down = 2` ResBlock: `upfirdn2d(x, f, padding=p + 1)` in front of the strided 3x3 convolution
    (conv2d_resample.py:119-122) and `upfirdn2d(x, f, down=2, padding=p)` in front of the 1x1 skip convolution (:107-110) -- the second is the odd samples of the
    first.  Inference only (no autograd); float32 GPU tensors; raises NativeNotCovered when the kernel declines."""
    _init()
    px0, px1, py0, py1 = _parse_padding(padding)
    if x.dtype != torch.float32 or not x.is_cuda or f is None or f.ndim != 2 or f.device != x.device or not x.is_contiguous():
        raise nat.NativeNotCovered('upfirdn2d.filter_with_odd_samples: a contiguous float32 GPU tensor and a 2-D filter')
    n, c, ih, iw = x.shape
    fh, fw = f.shape
    ow, oh = iw + px0 + px1 - fw + 1, ih + py0 + py1 - fh + 1
    if ow < 3 or oh < 3 or n == 0 or c == 0:
        raise nat.NativeNotCovered('upfirdn2d.filter_with_odd_samples: empty output')
    y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device)
    y_odd = torch.empty([n, c, (oh - 1) // 2, (ow - 1) // 2], dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        st = _plugin.lib.pg_upfirdn2d_with_odd_samples(nat.ptr(x), nat.ptr(f), nat.ptr(y), nat.ptr(y_odd), nat.PG_DTYPE[x.dtype], n, c, ih, iw, nat.i64arr(x.stride()),
                                                       fh, fw, nat.i64arr(f.stride()), oh, ow, nat.i64arr(y.stride()), px0, py0, int(bool(flip_filter)), float(gain),
                                                       nat.stream_of(x))
    if st == -2:
        raise nat.NativeNotCovered('pg_upfirdn2d_with_odd_samples declined the call')
    nat.check(st, 'pg_upfirdn2d_with_odd_samples')
    return y, y_odd


def _native_upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain):
    """Tensor-level twin of the reference plugin entry ``upfirdn2d(...)`` (upfirdn2d.cpp:16-94)."""
    _init()
    if x.dtype not in nat.PG_DTYPE:
        raise nat.NativeOpError(f'upfirdn2d: unsupported dtype {x.dtype}')
    if f.device != x.device:
        raise nat.NativeOpError('upfirdn2d: f must reside on the same device as x')
    if f.dtype != torch.float32:
        raise nat.NativeOpError('upfirdn2d: f must be float32')
    if x.ndim != 4 or f.ndim != 2:
        raise nat.NativeOpError('upfirdn2d: x must be rank 4 and f rank 2')
    if upx < 1 or upy < 1 or downx < 1 or downy < 1:
        raise nat.NativeOpError('upfirdn2d: up/down factors must be at least 1')
    n, c, ih, iw = x.shape
    fh, fw = f.shape
    ow = (iw * upx + padx0 + padx1 - fw + downx) // downx
    oh = (ih * upy + pady0 + pady1 - fh + downy) // downy
    if ow < 1 or oh < 1:
        raise nat.NativeOpError('upfirdn2d: output must be at least 1x1')
    channels_last = x.stride(1) == 1 and c > 1 and x.is_contiguous(memory_format=torch.channels_last)
    y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device,
                    memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    if n == 0 or c == 0:
        return y
    with torch.cuda.device(x.device):
        st = _plugin.lib.pg_upfirdn2d(nat.ptr(x), nat.ptr(f), nat.ptr(y), nat.PG_DTYPE[x.dtype], n, c, ih, iw, nat.i64arr(x.stride()),
                                      fh, fw, nat.i64arr(f.stride()), oh, ow, nat.i64arr(y.stride()),
                                      upx, upy, downx, downy, padx0, pady0, int(bool(flip)), float(gain), nat.stream_of(x))
    nat.check(st, 'pg_upfirdn2d')
    return y


class _Upfirdn2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f, up, down, padding, flip_filter, gain):
        assert isinstance(x, torch.Tensor) and x.ndim == 4
        upx, upy = up
        downx, downy = down
        padx0, padx1, pady0, pady1 = padding
        if f is None:
            f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
        assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
        if f.ndim == 2:
            y = _native_upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip_filter, gain)
        else:   # separable: one pass per axis, sqrt(gain) each (upfirdn2d.py:239-240)
            y = _native_upfirdn2d(x, f.unsqueeze(0), upx, 1, downx, 1, padx0, padx1, 0, 0, flip_filter, np.sqrt(gain))
            y = _native_upfirdn2d(y, f.unsqueeze(1), 1, upy, 1, downy, 0, 0, pady0, pady1, flip_filter, np.sqrt(gain))
        ctx.save_for_backward(f)
        ctx.cfg = (x.shape, up, down, padding, flip_filter, gain)
        return y

    @staticmethod
    def backward(ctx, dy):
        f, = ctx.saved_tensors
        x_shape, (upx, upy), (downx, downy), (padx0, _, pady0, _), flip_filter, gain = ctx.cfg
        _, _, ih, iw = x_shape
        _, _, oh, ow = dy.shape
        fw, fh = _get_filter_size(f)
        p = (fw - padx0 - 1, iw * upx - ow * downx + padx0 - upx + 1,
             fh - pady0 - 1, ih * upy - oh * downy + pady0 - upy + 1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _Upfirdn2d.apply(dy, f, (downx, downy), (upx, upy), p, not flip_filter, gain)
        assert not ctx.needs_input_grad[1]
        return dx, None, None, None, None, None, None


# ---------------------------------------------------------------------------- public ops

def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda'):
    r"""Upsample by zero insertion (`up`), pad/crop (`padding`, negative = crop), convolve with `f`
    (true convolution unless `flip_filter`), keep every `down`-th sample, scale by `gain`.
    Arguments as in the reference (upfirdn2d.py:120-159)."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    assert f is None or (isinstance(f, torch.Tensor) and f.dtype == torch.float32 and not f.requires_grad)
    if impl == 'cuda' and x.device.type == 'cuda':
        return _Upfirdn2d.apply(x, f, _parse_scaling(up), _parse_scaling(down), _parse_padding(padding), bool(flip_filter), gain)
    return _upfirdn2d_torch(x, f, _parse_scaling(up), _parse_scaling(down), _parse_padding(padding), bool(flip_filter), gain)


def _upfirdn2d_torch(x, f, up, down, padding, flip_filter, gain):
    """upfirdn2d from stock torch ops (any device, differentiable to any order): scatter the samples onto the up-sampled
    grid, pad / crop, depth-wise correlate with the (flipped) taps, keep every down-th sample.  Same arithmetic as the
    reference's `_upfirdn2d_ref` (upfirdn2d.py:168-208): taps are cast to x's dtype, the gain is split over the passes."""
    assert x.ndim == 4
    (upx, upy), (downx, downy), (px0, px1, py0, py1) = up, down, padding
    n, c, h, w = x.shape
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    grid = x.new_zeros([n, c, h * upy, w * upx])
    grid[:, :, ::upy, ::upx] = x                                     # zero stuffing: sample first, then up-1 zeros
    grid = torch.nn.functional.pad(grid, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    grid = grid[:, :, max(-py0, 0):grid.shape[2] - max(-py1, 0), max(-px0, 0):grid.shape[3] - max(-px1, 0)]
    taps = (f * (gain ** (f.ndim / 2))).to(x.dtype)
    if not flip_filter:                                              # true convolution = correlation with the flipped taps
        taps = taps.flip(list(range(taps.ndim)))
    depthwise = lambda t, k: torch.nn.functional.conv2d(t, k[None, None].repeat(c, 1, 1, 1), groups=c)
    if taps.ndim == 2:
        grid = depthwise(grid, taps)
    else:                                                            # separable: rows, then columns
        grid = depthwise(depthwise(grid, taps[None, :]), taps[:, None])
    return grid[:, :, ::downy, ::downx]


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    r"""FIR filtering that keeps the image size (plus user padding); upfirdn2d.py:272-304."""
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + fw // 2, padx1 + (fw - 1) // 2, pady0 + fh // 2, pady1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    r"""FIR upsampling to `up` times the size (plus user padding); upfirdn2d.py:308-343."""
    upx, upy = _parse_scaling(up)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + (fw + upx - 1) // 2, padx1 + (fw - upx) // 2, pady0 + (fh + upy - 1) // 2, pady1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    r"""FIR downsampling to 1/`down` of the size (plus user padding); upfirdn2d.py:347-382."""
    downx, downy = _parse_scaling(down)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + (fw - downx + 1) // 2, padx1 + (fw - downx) // 2, pady0 + (fh - downy + 1) // 2, pady1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
