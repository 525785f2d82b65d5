"""2-D convolution with optional FIR up/down-sampling on MI355X.

``conv2d_resample(x, w, f, up, down, padding, groups, flip_weight, flip_filter)`` with the
semantics of the reference op (torch_utils/ops/conv2d_resample.py:59-154): padding is applied
once, up front, relative to the up-sampled image; which composition of FIR and convolution is
used depends only on (kernel size, up, down, padding) and fixes where rounding happens, so the
routes below are the reference's:

    pointwise_down   1x1 kernel, down > 1       FIR-decimate, then mix channels        (:107-110)
    pointwise_up     1x1 kernel, up > 1         mix channels, then FIR-interpolate     (:113-116)
    strided          down > 1                   blur, then stride-`down` conv          (:119-122)
    transposed       up > 1                     stride-`up` transposed conv, then blur (:125-142)
    plain            symmetric non-negative pad one padded conv                        (:145-147)
    generic          anything else              pad/crop by upfirdn2d, conv, decimate  (:150-154)

The route is chosen by ``_plan`` (pure integer algebra, testable on its own); the pieces are this
package's HIP ops (``upfirdn2d`` and the MFMA convolution behind ``conv2d_gradfix``).
"""

import collections

import torch

from .. import misc
from . import bias_act
from . import conv2d_gradfix
from . import upfirdn2d
from .upfirdn2d import _get_filter_size
from .upfirdn2d import _parse_padding

_Plan = collections.namedtuple('_Plan', 'route fir_pad conv_pad')


def _get_weight_shape(w):
    shape = [int(sz) for sz in w.shape]
    misc.assert_shape(w, shape)
    return shape


def _plan(kh, kw, fw, fh, up, down, padding):
    """Route and paddings for one call: `fir_pad` = [x0, x1, y0, y1] handed to upfirdn2d, `conv_pad` = (y, x) of the conv."""
    pad = list(_parse_padding(padding))                      # x0, x1, y0, y1
    ext = (fw, fw, fh, fh)
    for i in range(4):                                       # footprint of the resampling filters, split lo/hi
        lo = (i % 2 == 0)
        if up > 1:
            pad[i] += (ext[i] + up - 1) // 2 if lo else (ext[i] - up) // 2
        if down > 1:
            pad[i] += (ext[i] - down + 1) // 2 if lo else (ext[i] - down) // 2
    pointwise = (kh == 1 and kw == 1)
    if pointwise and up == 1 and down > 1:
        return _Plan('pointwise_down', pad, (0, 0))
    if pointwise and down == 1 and up > 1:
        return _Plan('pointwise_up', pad, (0, 0))
    if up == 1 and down > 1:
        return _Plan('strided', pad, (0, 0))
    if up > 1:
        ker = (kw, kw, kh, kh)
        for i in range(4):                                   # what the transposed conv itself adds on each side
            pad[i] -= (ker[i] - 1) if i % 2 == 0 else (ker[i] - up)
        tx = max(min(-pad[0], -pad[1]), 0)                   # the part of a negative pad the conv can absorb
        ty = max(min(-pad[2], -pad[3]), 0)
        return _Plan('transposed', [pad[0] + tx, pad[1] + tx, pad[2] + ty, pad[3] + ty], (ty, tx))
    if pad[0] == pad[1] and pad[2] == pad[3] and pad[0] >= 0 and pad[2] >= 0:
        return _Plan('plain', pad, (pad[2], pad[0]))
    return _Plan('generic', pad, (0, 0))


def _conv2d_wrapper(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True, bias=None, epilogue=None):
    """Cross-correlation (flip_weight=True, what conv2d computes) or true convolution of x with w."""
    if not flip_weight:
        w = w.flip([2, 3])
    if transpose:
        assert bias is None and epilogue is None
        return conv2d_gradfix.conv_transpose2d(x, w, stride=stride, padding=padding, groups=groups)
    return conv2d_gradfix.conv2d(x, w, bias=bias, stride=stride, padding=padding, groups=groups, _epilogue=epilogue)


@misc.profiled_function
def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False, _epilogue=None):
    """Signature of the reference (conv2d_resample.py:69).  `_epilogue` (private): dict(bias, act, alpha, gain, clamp) -- the `bias_act`
    the calling layer applies to the result (networks.py:176-178); it is ALWAYS applied on return: inside the convolution launch where the
    convolution is the last step of the route (plain, strided, point-wise down), as the separate op after the FIR otherwise."""
    if _epilogue is not None:
        ep = dict(_epilogue)
        b = ep.pop('bias', None)
        _, _, kh_, kw_ = _get_weight_shape(w)
        route = _plan(kh_, kw_, *_get_filter_size(f), up, down, padding).route
        if route in ('plain', 'strided', 'pointwise_down'):
            return _conv2d_resample(x, w, f, up, down, padding, groups, flip_weight, flip_filter, bias=b, epilogue=ep)
        y = _conv2d_resample(x, w, f, up, down, padding, groups, flip_weight, flip_filter)
        return bias_act.bias_act(y, b, act=ep.get('act', 'linear'), alpha=ep.get('alpha'), gain=ep.get('gain'), clamp=ep.get('clamp'))
    return _conv2d_resample(x, w, f, up, down, padding, groups, flip_weight, flip_filter)


def _conv2d_resample(x, w, f, up, down, padding, groups, flip_weight, flip_filter, bias=None, epilogue=None):
    assert isinstance(x, torch.Tensor) and (x.ndim == 4)
    assert isinstance(w, torch.Tensor) and (w.ndim == 4) and (w.dtype == x.dtype or (w.dtype == torch.float32 and x.is_cuda))     # (float32 weights beside 16-bit GPU activations: conv2d_gradfix packs from float32)
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in [1, 2] and f.dtype == torch.float32)
    assert isinstance(up, int) and (up >= 1)
    assert isinstance(down, int) and (down >= 1)
    assert isinstance(groups, int) and (groups >= 1)
    cout, cin_per_group, kh, kw = _get_weight_shape(w)
    fw, fh = _get_filter_size(f)
    plan = _plan(kh, kw, fw, fh, up, down, padding)
    fir = lambda t, **kw_: upfirdn2d.upfirdn2d(x=t, f=f, flip_filter=flip_filter, **kw_)
    conv = lambda t, **kw_: _conv2d_wrapper(x=t, w=w, groups=groups, flip_weight=flip_weight, bias=bias, epilogue=epilogue, **kw_)

    if plan.route == 'pointwise_down':
        return conv(fir(x, down=down, padding=plan.fir_pad))
    assert (bias is None and epilogue is None) or plan.route in ('plain', 'strided', 'pointwise_down')
    if plan.route == 'pointwise_up':
        return fir(conv(x), up=up, padding=plan.fir_pad, gain=up ** 2)
    if plan.route == 'strided':
        return conv(fir(x, padding=plan.fir_pad), stride=down)
    if plan.route == 'transposed':
        if groups == 1:
            wt = w.transpose(0, 1)
        else:                                                # per-group O<->I swap
            wt = w.reshape(groups, cout // groups, cin_per_group, kh, kw).transpose(1, 2)
            wt = wt.reshape(groups * cin_per_group, cout // groups, kh, kw)
        y = _conv2d_wrapper(x=x, w=wt, stride=up, padding=list(plan.conv_pad), groups=groups, transpose=True, flip_weight=(not flip_weight))
        y = fir(y, padding=plan.fir_pad, gain=up ** 2)
        return fir(y, down=down) if down > 1 else y
    if plan.route == 'plain':
        return conv(x, padding=list(plan.conv_pad))
    # generic: explicit pad/crop (and zero-stuffing) through upfirdn2d, unpadded conv, optional decimation
    y = upfirdn2d.upfirdn2d(x=x, f=(f if up > 1 else None), up=up, padding=plan.fir_pad, gain=up ** 2, flip_filter=flip_filter)
    y = conv(y)
    return fir(y, down=down) if down > 1 else y
