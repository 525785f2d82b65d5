"""The oracle's scalar C restatement of the plugin kernels agrees with the golden vectors of the
reference's Python ops (and so with oracle/ops_ref.py).  CPU only."""

import ctypes
import math

import numpy as np
import pytest
import torch

import cases as C
from detgen import det_tensor
from oracle import build as obuild
from oracle import ops_ref as R


@pytest.fixture(scope='module')
def lib():
    return obuild.load()


def _p(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else ctypes.c_void_p(0)


@pytest.mark.parametrize('case', [c for c in C.UPFIRDN2D_CASES if c[0] != 'sym_up2_sep' and c[0] != 'sym_down2_sep'], ids=lambda c: c[0])
def test_c_upfirdn2d(lib, golden, case):
    g = golden('g1_upfirdn2d.npz')
    name, xs, fspec, up, down, pad, flip, gain = case
    x = det_tensor(name + '.x', xs).numpy()
    f = np.ones((1, 1), np.float32) if fspec is None else np.ascontiguousarray(g[f'{name}/f'])
    upx, upy = R._pair(up)
    dnx, dny = R._pair(down)
    px0, _, py0, _ = R._pad4(pad)
    want = g[f'{name}/y']
    y = np.empty(want.shape, np.float32)
    n, c, h, w = xs
    st = lib.oracle_upfirdn2d(_p(x), _p(f), _p(y), n, c, h, w, f.shape[0], f.shape[1], want.shape[2], want.shape[3],
                              upx, upy, dnx, dny, px0, py0, int(flip), float(gain))
    assert st == 0
    np.testing.assert_allclose(y, want, rtol=2e-5, atol=3e-6 * max(1.0, float(np.abs(want).max())))


@pytest.mark.parametrize('act', C.ACTS)
def test_c_bias_act(lib, golden, act):
    g = golden('g2_bias_act.npz')
    for vname, has_b, gain, clamp, dim, xs in C.BIAS_ACT_VARIANTS:
        name = f'{act}.{vname}'
        x = det_tensor(name + '.x', xs, scale=2.0).numpy()
        b = det_tensor(name + '.b', [xs[dim]]).numpy() if has_b else None
        step = int(np.prod(xs[dim + 1:])) if has_b else 1
        spec = R.ACTIVATIONS[act]
        y = np.empty(xs, np.float32)
        st = lib.oracle_bias_act(_p(x), _p(b), _p(y), x.size, xs[dim] if has_b else 1, step, spec[2], float(spec[0]),
                                 float(spec[1] if gain is None else gain), float(-1 if clamp is None else clamp))
        assert st == 0
        np.testing.assert_allclose(y, g[f'{name}/y'], rtol=2e-5, atol=2e-6)


def test_c_conv2d(lib, golden):
    g = golden('g3_conv2d_resample.npz')
    for name, xs, wsh, taps, up, down, pad, groups, flipw in C.CONV2D_RESAMPLE_CASES:
        if up != 1 or down != 1 or groups != 1 or not flipw or not isinstance(pad, int):
            continue
        x = det_tensor(name + '.x', xs).numpy()
        w = det_tensor(name + '.w', wsh, scale=1 / math.sqrt(wsh[1] * wsh[2] * wsh[3])).numpy()
        want = g[f'{name}/y']
        y = np.empty(want.shape, np.float32)
        assert lib.oracle_conv2d(_p(x), _p(w), _p(None), _p(y), xs[0], xs[1], xs[2], xs[3], wsh[0], wsh[2], wsh[3], 1, pad, pad) == 0
        np.testing.assert_allclose(y, want, rtol=1e-4, atol=1e-5)
