"""CPU restatement of the reference's operator semantics (TEST INFRASTRUCTURE ONLY).

Every function states the arithmetic directly from its definition (no call
into the product, no call into ``/root/reference``) and cites the reference
lines it follows.  All paths run on CPU tensors; float64 inputs are supported
so tests can use a high-precision version of the same formula.

The dense contraction itself (``F.conv2d`` / ``F.conv_transpose2d``) is the
third-party arithmetic the reference also delegates to
(torch_utils/ops/conv2d_gradfix.py:35-43): torch, here 2.10.0 CPU.
"""

import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# Parameter algebra (torch_utils/ops/upfirdn2d.py:37-68)


def _pair(v):
    if isinstance(v, int):
        return v, v
    a, b = v
    return int(a), int(b)


def _pad4(p):
    """-> (x0, x1, y0, y1); upfirdn2d.py:46-55."""
    if isinstance(p, int):
        return p, p, p, p
    p = [int(v) for v in p]
    if len(p) == 2:
        return p[0], p[0], p[1], p[1]
    assert len(p) == 4
    return tuple(p)


def filter_size(f):
    """-> (fw, fh); upfirdn2d.py:57-68."""
    if f is None:
        return 1, 1
    return int(f.shape[-1]), int(f.shape[0])


def setup_filter(f, normalize=True, flip_filter=False, gain=1, separable=None):
    """FIR taps as the reference prepares them (upfirdn2d.py:72-116).

    1-D tap lists shorter than 8 become their outer product; >= 8 taps stay 1-D
    (separable).  Normalised to unit DC, optionally flipped, scaled by
    gain**(ndim/2).
    """
    f = torch.as_tensor(1 if f is None else f, dtype=torch.float32)
    if f.ndim == 0:
        f = f.reshape(1)
    if separable is None:
        separable = f.ndim == 1 and f.numel() >= 8
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(tuple(range(f.ndim)))
    return f * (gain ** (f.ndim / 2))


# --------------------------------------------------------------------------
# upfirdn2d (upfirdn2d.py:168-208; kernel semantics upfirdn2d.cu:29-92)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """Zero-stuff by `up`, pad/crop, FIR, keep every `down`-th sample.

    Written as an explicit tap sum over shifted views of the padded signal:
        y[oy,ox] = gain * sum_{ky,kx} g[ky,kx] * xp[oy*dy+ky, ox*dx+kx]
    with g = f flipped unless flip_filter (true convolution by default,
    upfirdn2d.py:195-196).  Separable 1-D taps are applied as the outer
    product, which is what the two-pass reference computes up to rounding.
    """
    assert x.ndim == 4
    upx, upy = _pair(up)
    dnx, dny = _pair(down)
    px0, px1, py0, py1 = _pad4(padding)
    n, c, h, w = x.shape
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32)
    f = f.to(torch.float64 if x.dtype == torch.float64 else torch.float32)
    if f.ndim == 1:
        f2 = torch.outer(f, f)
    else:
        f2 = f
    fh, fw = f2.shape
    g = f2 if flip_filter else f2.flip((0, 1))

    acc_dtype = torch.float64 if x.dtype == torch.float64 else torch.float32
    # zero-stuffed + padded canvas, built by scatter instead of reshape/pad
    ch, cw = h * upy + py0 + py1, w * upx + px0 + px1
    oh = (ch - fh + dny) // dny
    ow = (cw - fw + dnx) // dnx
    assert oh >= 1 and ow >= 1
    big = torch.zeros([n, c, h * upy + max(py0, 0) + max(py1, 0), w * upx + max(px0, 0) + max(px1, 0)], dtype=acc_dtype)
    big[:, :, max(py0, 0):max(py0, 0) + h * upy:upy, max(px0, 0):max(px0, 0) + w * upx:upx] = x.to(acc_dtype)
    # negative padding == crop
    big = big[:, :, max(-py0, 0):big.shape[2] - max(-py1, 0), max(-px0, 0):big.shape[3] - max(-px1, 0)]
    assert big.shape[2] == ch and big.shape[3] == cw
    y = torch.zeros([n, c, oh, ow], dtype=acc_dtype)
    for ky in range(fh):
        for kx in range(fw):
            y += g[ky, kx].to(acc_dtype) * big[:, :, ky:ky + (oh - 1) * dny + 1:dny, kx:kx + (ow - 1) * dnx + 1:dnx]
    return (y * gain).to(x.dtype)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1):
    """upfirdn2d.py:308-343."""
    upx, upy = _pair(up)
    px0, px1, py0, py1 = _pad4(padding)
    fw, fh = filter_size(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1):
    """upfirdn2d.py:347-382."""
    dnx, dny = _pair(down)
    px0, px1, py0, py1 = _pad4(padding)
    fw, fh = filter_size(f)
    p = [px0 + (fw - dnx + 1) // 2, px1 + (fw - dnx) // 2, py0 + (fh - dny + 1) // 2, py1 + (fh - dny) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain)


def filter2d(x, f, padding=0, flip_filter=False, gain=1):
    """upfirdn2d.py:272-304."""
    px0, px1, py0, py1 = _pad4(padding)
    fw, fh = filter_size(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain)


def upfirdn2d_backward_params(x_shape, dy_shape, f, up, down, padding, flip_filter):
    """Arguments of the upfirdn2d call that computes dx (upfirdn2d.py:245-264)."""
    upx, upy = _pair(up)
    dnx, dny = _pair(down)
    px0, _, py0, _ = _pad4(padding)
    _, _, ih, iw = x_shape
    _, _, oh, ow = dy_shape
    fw, fh = filter_size(f)
    p = [fw - px0 - 1, iw * upx - ow * dnx + px0 - upx + 1, fh - py0 - 1, ih * upy - oh * dny + py0 - upy + 1]
    return dict(up=(dnx, dny), down=(upx, upy), padding=p, flip_filter=not flip_filter)


# --------------------------------------------------------------------------
# bias_act (bias_act.py:23-33, 93-123; kernel bias_act.cu:38-146)

_SELU_SCALE = 1.0507009873554804934193349852946
_SELU_ALPHA = 1.6732632423543772848170429916717

# name -> (default alpha, default gain, plugin index, which tensor backward needs, has 2nd grad)
ACTIVATIONS = {
    'linear':   (0.0, 1.0,          1, '',  False),
    'relu':     (0.0, math.sqrt(2), 2, 'y', False),
    'lrelu':    (0.2, math.sqrt(2), 3, 'y', False),
    'tanh':     (0.0, 1.0,          4, 'y', True),
    'sigmoid':  (0.0, 1.0,          5, 'y', True),
    'elu':      (0.0, 1.0,          6, 'y', True),
    'selu':     (0.0, 1.0,          7, 'y', True),
    'softplus': (0.0, 1.0,          8, 'y', True),
    'swish':    (0.0, math.sqrt(2), 9, 'x', True),
}


def _act(name, x, alpha):
    if name == 'linear':
        return x
    if name == 'relu':
        return torch.where(x > 0, x, torch.zeros_like(x))
    if name == 'lrelu':
        return torch.where(x > 0, x, x * alpha)
    if name == 'tanh':
        return torch.tanh(x)
    if name == 'sigmoid':
        return torch.sigmoid(x)
    if name == 'elu':
        return torch.where(x >= 0, x, torch.expm1(x))
    if name == 'selu':
        return torch.where(x >= 0, _SELU_SCALE * x, (_SELU_SCALE * _SELU_ALPHA) * torch.expm1(x))
    if name == 'softplus':
        return F.softplus(x)
    if name == 'swish':
        return x * torch.sigmoid(x)
    raise KeyError(name)


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """y = clamp(act(x + b) * gain); bias_act.py:93-123."""
    d_alpha, d_gain, _, _, _ = ACTIVATIONS[act]
    alpha = float(d_alpha if alpha is None else alpha)
    gain = float(d_gain if gain is None else gain)
    if b is not None:
        assert b.ndim == 1 and b.shape[0] == x.shape[dim]
        shape = [1] * x.ndim
        shape[dim] = -1
        x = x + b.reshape(shape)
    y = _act(act, x, alpha)
    if gain != 1:
        y = y * gain
    if clamp is not None and clamp >= 0:
        y = y.clamp(-clamp, clamp)
    return y


def bias_act_grad(dy, x, b, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """dx of ``bias_act`` for upstream gradient dy, by autograd on the forward
    restatement (what BiasActCuda.backward must reproduce, bias_act.py:160-176)."""
    xx = x.detach().clone().requires_grad_(True)
    bb = b.detach().clone().requires_grad_(True) if b is not None else None
    y = bias_act(xx, bb, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)
    grads = torch.autograd.grad(y, [xx] + ([bb] if bb is not None else []), dy)
    return grads[0], (grads[1] if bb is not None else None)


# --------------------------------------------------------------------------
# fma (fma.py:15-58)


def fma(a, b, c):
    return a * b + c


# --------------------------------------------------------------------------
# conv2d_resample (conv2d_resample.py:29-154)


def _corr(x, w, stride=1, padding=(0, 0), groups=1, transpose=False, flip_weight=True):
    """conv2d_resample.py:29-54: cross-correlation unless flip_weight=False."""
    if not flip_weight:
        w = w.flip((2, 3))
    if transpose:
        return F.conv_transpose2d(x, w, stride=stride, padding=padding, groups=groups)
    return F.conv2d(x, w, stride=stride, padding=padding, groups=groups)


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    """Convolution with optional FIR up/down-sampling; conv2d_resample.py:59-154.

    The branch structure is the reference's (it fixes where rounding happens):
    1x1 kernels resample on the cheap side; down>1 blurs then strides; up>1 runs
    a stride-`up` transposed conv then the FIR with gain up**2; everything else
    is a plain padded conv; odd paddings go through the generic route.
    """
    assert x.ndim == 4 and w.ndim == 4
    cout, cin_g, kh, kw = (int(s) for s in w.shape)
    fw, fh = filter_size(f)
    px0, px1, py0, py1 = _pad4(padding)
    if up > 1:
        px0 += (fw + up - 1) // 2
        px1 += (fw - up) // 2
        py0 += (fh + up - 1) // 2
        py1 += (fh - up) // 2
    if down > 1:
        px0 += (fw - down + 1) // 2
        px1 += (fw - down) // 2
        py0 += (fh - down + 1) // 2
        py1 += (fh - down) // 2

    one_by_one = kh == 1 and kw == 1
    if one_by_one and down > 1 and up == 1:
        x = upfirdn2d(x, f, down=down, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _corr(x, w, groups=groups, flip_weight=flip_weight)
    if one_by_one and up > 1 and down == 1:
        x = _corr(x, w, groups=groups, flip_weight=flip_weight)
        return upfirdn2d(x, f, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    if down > 1 and up == 1:
        x = upfirdn2d(x, f, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _corr(x, w, stride=down, groups=groups, flip_weight=flip_weight)
    if up > 1:
        if groups == 1:
            wt = w.transpose(0, 1)
        else:
            wt = w.reshape(groups, cout // groups, cin_g, kh, kw).transpose(1, 2)
            wt = wt.reshape(groups * cin_g, cout // groups, kh, kw)
        px0 -= kw - 1
        px1 -= kw - up
        py0 -= kh - 1
        py1 -= kh - up
        pxt = max(min(-px0, -px1), 0)
        pyt = max(min(-py0, -py1), 0)
        x = _corr(x, wt, stride=up, padding=(pyt, pxt), groups=groups, transpose=True, flip_weight=not flip_weight)
        x = upfirdn2d(x, f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2, flip_filter=flip_filter)
        if down > 1:
            x = upfirdn2d(x, f, down=down, flip_filter=flip_filter)
        return x
    if px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:
        return _corr(x, w, padding=(py0, px0), groups=groups, flip_weight=flip_weight)
    x = upfirdn2d(x, f if up > 1 else None, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    x = _corr(x, w, groups=groups, flip_weight=flip_weight)
    if down > 1:
        x = upfirdn2d(x, f, down=down, flip_filter=flip_filter)
    return x


# --------------------------------------------------------------------------
# modulated_conv2d (training/networks.py:37-94)


def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None,
                     demodulate=True, flip_weight=True, fused_modconv=True):
    """StyleGAN2 (de)modulated convolution; networks.py:37-94.

    fused:     per-sample weights w[n] = W * s[n] (* d[n]) through a grouped conv
    non-fused: conv(x * s, W) * d + noise
    d[n,o] = rsqrt(sum_{i,k} (W[o,i,k] s[n,i])^2 + 1e-8).
    fp16 inputs get the reference's inf-norm pre-normalisation (:57-59).
    """
    n = x.shape[0]
    cout, cin, kh, kw = weight.shape
    assert styles.shape == (n, cin)
    if x.dtype == torch.float16 and demodulate:
        weight = weight * (1 / math.sqrt(cin * kh * kw) / weight.abs().amax(dim=(1, 2, 3), keepdim=True))
        styles = styles / styles.abs().amax(dim=1, keepdim=True)
    wmod = None
    dcoefs = None
    if demodulate or fused_modconv:
        wmod = weight[None] * styles[:, None, :, None, None]
    if demodulate:
        dcoefs = (wmod.square().sum(dim=(2, 3, 4)) + 1e-8).rsqrt()
    if not fused_modconv:
        y = x * styles.to(x.dtype)[:, :, None, None]
        y = conv2d_resample(y, weight.to(x.dtype), f=resample_filter, up=up, down=down, padding=padding, flip_weight=flip_weight)
        if demodulate:
            y = y * dcoefs.to(x.dtype)[:, :, None, None]
        if noise is not None:
            y = y + noise.to(x.dtype)
        return y
    if demodulate:
        wmod = wmod * dcoefs[:, :, None, None, None]
    y = conv2d_resample(x.reshape(1, n * cin, *x.shape[2:]), wmod.reshape(n * cout, cin, kh, kw).to(x.dtype),
                        f=resample_filter, up=up, down=down, padding=padding, groups=n, flip_weight=flip_weight)
    y = y.reshape(n, cout, *y.shape[2:])
    if noise is not None:
        y = y + noise
    return y


# --------------------------------------------------------------------------
# Algorithmic work model -- used by bench.py's roofline.


def conv_flops(n, cout, oh, ow, cin_g, kh, kw):
    return 2.0 * n * cout * oh * ow * cin_g * kh * kw


def to_numpy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
