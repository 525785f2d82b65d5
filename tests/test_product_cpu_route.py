"""The product's own plain-torch route (impl='ref' / CPU tensors: the reference's dispatch rule, upfirdn2d.py:161-164,
bias_act.py:86-89, conv2d_gradfix.py:53-56) against the golden vectors the REFERENCE produced (G1-G4) -- the same
fixtures that pin the oracle; the product code imports nothing from oracle/.  CPU only."""

import math

import numpy as np
import pytest
import torch

import cases as C
from detgen import det_tensor


def close(a, b, rtol, atol):
    a = a.detach().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    np.testing.assert_allclose(a, np.asarray(b, dtype=np.float64), rtol=rtol, atol=atol)


def _filter(spec, name):
    from torch_utils.ops import upfirdn2d
    if spec is None:
        return None
    kind, v = spec
    return upfirdn2d.setup_filter(v) if kind == 'taps' else det_tensor(name + '.f', v, 'uniform')


@pytest.mark.parametrize('case', C.UPFIRDN2D_CASES, ids=[c[0] for c in C.UPFIRDN2D_CASES])
@pytest.mark.parametrize('impl', ['ref', 'cuda'])          # 'cuda' on a CPU tensor dispatches to the same route, as in the reference
def test_upfirdn2d_cpu_route(golden, case, impl):
    from torch_utils.ops import upfirdn2d
    g = golden('g1_upfirdn2d.npz')
    name, xs, fspec, up, down, pad, flip, gain = case
    f = _filter(fspec, name)
    x = det_tensor(name + '.x', xs).requires_grad_(True)
    y = upfirdn2d.upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain, impl=impl)
    assert tuple(y.shape) == g[f'{name}/y'].shape
    scale = max(1.0, float(np.abs(g[f'{name}/y']).max()))
    close(y, g[f'{name}/y'], rtol=2e-5, atol=2e-6 * scale)
    dx, = torch.autograd.grad(y, [x], det_tensor(name + '.dy', y.shape))
    close(dx, g[f'{name}/dx'], rtol=2e-5, atol=2e-6 * scale)


@pytest.mark.parametrize('act', C.ACTS)
def test_bias_act_cpu_route(golden, act):
    from torch_utils.ops import bias_act
    g = golden('g2_bias_act.npz')
    for vname, has_b, gain, clamp, dim, xs in C.BIAS_ACT_VARIANTS:
        name = f'{act}.{vname}'
        x = det_tensor(name + '.x', xs, scale=2.0).requires_grad_(True)
        b = det_tensor(name + '.b', [xs[dim]]).requires_grad_(True) if has_b else None
        y = bias_act.bias_act(x, b, dim=dim, act=act, gain=gain, clamp=clamp, impl='ref')
        close(y, g[f'{name}/y'], rtol=2e-5, atol=2e-6)
        grads = torch.autograd.grad(y, [x] + ([b] if has_b else []), det_tensor(name + '.dy', xs))
        close(grads[0], g[f'{name}/dx'], rtol=2e-5, atol=2e-6)
        if has_b:
            close(grads[1], g[f'{name}/db'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('case', C.CONV2D_RESAMPLE_CASES, ids=[c[0] for c in C.CONV2D_RESAMPLE_CASES])
def test_conv2d_resample_cpu_route(golden, case):
    from torch_utils.ops import conv2d_resample, upfirdn2d
    g = golden('g3_conv2d_resample.npz')
    name, xs, wsh, taps, up, down, pad, groups, flipw = case
    f = upfirdn2d.setup_filter(taps)
    x = det_tensor(name + '.x', xs).requires_grad_(True)
    w = det_tensor(name + '.w', wsh, scale=1 / math.sqrt(wsh[1] * wsh[2] * wsh[3])).requires_grad_(True)
    y = conv2d_resample.conv2d_resample(x, w, f=f, up=up, down=down, padding=pad, groups=groups, flip_weight=flipw)
    close(y, g[f'{name}/y'], rtol=1e-4, atol=1e-5)
    dx, dw = torch.autograd.grad(y, [x, w], det_tensor(name + '.dy', y.shape))
    close(dx, g[f'{name}/dx'], rtol=1e-4, atol=1e-5)
    close(dw, g[f'{name}/dw'], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('case', C.MODCONV_CASES, ids=[c[0] for c in C.MODCONV_CASES])
def test_modulated_conv2d_cpu_route(golden, case):
    from torch_utils.ops import upfirdn2d
    from training import networks
    g = golden('g4_modconv.npz')
    name, n, cin, cout, k, h, up, demod, fused, noise_kind = case
    f = upfirdn2d.setup_filter(C.FIR_1331)
    x = det_tensor(name + '.x', [n, cin, h, h])
    w = det_tensor(name + '.w', [cout, cin, k, k])
    s = det_tensor(name + '.s', [n, cin]) + 1.0
    hh = h * up
    noise = {'none': None, 'const': det_tensor(name + '.noise', [hh, hh]) * 0.1,
             'per_sample': det_tensor(name + '.noise', [n, 1, hh, hh]) * 0.1}[noise_kind]
    y = networks.modulated_conv2d(x, w, s, noise=noise, up=up, padding=k // 2, resample_filter=f, demodulate=demod,
                                  flip_weight=(up == 1), fused_modconv=fused)
    close(y, g[f'{name}/y'], rtol=1e-4, atol=1e-5)
