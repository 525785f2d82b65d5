"""CPU-only checks of the product's host side: the C-ABI libraries load and export every
symbol include/pasta_gan_ops.h declares; argument algebra; the transposed-conv phase
decomposition (emulated with torch CPU ops standing in for one pg_conv2d_forward launch);
and the reference's dispatch contract (CPU tensors / impl='ref' take the plain-torch route, GPU tensors the native op)."""

import ctypes
import os
import re

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'pasta_gan_ops.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pg_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for s in ('pg_bias_act', 'pg_upfirdn2d', 'pg_conv2d_forward', 'pg_conv2d_pack_weight', 'pg_conv2d_packed_size',
              'pg_modconv_dcoefs', 'pg_modconv_w2', 'pg_modconv_prep', 'pg_instance_norm_stats', 'pg_spade_norm'):
        assert s in syms


def test_c_abi_libraries_export_every_declared_symbol():
    from torch_utils import custom_ops
    custom_ops.verbosity = 'none'
    libs = [ctypes.CDLL(custom_ops.get_plugin(n, build_only=True)) for n in custom_ops.PLUGIN_SOURCES]
    for sym in declared_symbols():
        assert any(hasattr(lib, sym) for lib in libs), f'{sym} declared in include/pasta_gan_ops.h but exported by no plugin'
    for lib, name in zip(libs, custom_ops.PLUGIN_SOURCES):
        fn = getattr(lib, 'pg_' + name.replace('_plugin', '') + '_abi_version')
        fn.restype = ctypes.c_int
        assert fn() == custom_ops.ABI_VERSION
    # host-only entry point: packed-weight size needs no GPU
    conv = libs[list(custom_ops.PLUGIN_SOURCES).index('conv2d_plugin')]
    conv.pg_conv2d_packed_size.restype = ctypes.c_int64
    assert conv.pg_conv2d_packed_size(64, 3, 7, 7) == 16 * 49 * 64
    assert conv.pg_conv2d_packed_size(3, 64, 1, 1) == 64 * 1 * 32
    assert conv.pg_conv2d_packed_size(0, 64, 1, 1) == 0
    conv.pg_conv2d_winograd_packed_size.restype = ctypes.c_int64
    assert conv.pg_conv2d_winograd_packed_size(70, 20) == 16 * 32 * 128            # [16 positions][CinP = 32][CoutP64 = 128]
    # host-only: the split-K planner (no GPU touched)
    plan = conv.pg_conv2d_splitk_plan
    plan.restype = ctypes.c_int
    assert plan(8, 512, 8, 8, 512, 1, 1, 1) > 1 and 32 % plan(8, 512, 8, 8, 512, 1, 1, 1) == 0     # 64 tiles for 256 CUs: split; divides the 32 chunks
    assert plan(8, 512, 9, 9, 512, 2, 2, 1) > 1
    assert plan(8, 64, 512, 512, 64, 3, 3, 1) == 1 and plan(8, 512, 8, 8, 512, 5, 5, 1) == 1 and plan(8, 24, 8, 8, 512, 1, 1, 1) == 1
    # argument validation returns before any launch: sizes a 32-bit buffer descriptor cannot address are refused, not wrapped
    i64x4 = (ctypes.c_int64 * 4)(1, 1, 1, 1)
    dummy = ctypes.c_void_p(64)                                                      # non-null, never dereferenced on these paths
    fwd = conv.pg_conv2d_forward
    fwd.restype = ctypes.c_int
    argt = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 12 + [ctypes.POINTER(ctypes.c_int64)] + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
    fwd.argtypes = argt
    assert fwd(dummy, dummy, dummy, 1, 64, 4096, 4096, 8, 3, 3, 1, 1, 1, 4096, 4096, i64x4, 1, 1, 0, 0, None, None) == -3      # PG_ERR_TOO_LARGE: 4 GiB image
    assert fwd(None, dummy, dummy, 1, 64, 8, 8, 8, 3, 3, 1, 1, 1, 8, 8, i64x4, 1, 1, 0, 0, None, None) == -1                    # PG_ERR_INVALID_ARG
    assert fwd(dummy, dummy, dummy, 1, 64, 8, 8, 8, 3, 3, 1, 1, 1, 8, 8, i64x4, 0, 1, 0, 0, None, None) == -1


def test_dispatch_follows_the_reference():
    """upfirdn2d.py:161-164 / bias_act.py:86-89: `impl='cuda'` on a GPU tensor takes the native op, everything else the
    plain-torch route (product-own, pinned against G1-G3 in test_product_cpu_route.py).  A GPU tensor never lands on the
    torch route silently: the native wrapper raises when its plugin is missing (checked here without a GPU by making the
    loader fail)."""
    from torch_utils import custom_ops
    from torch_utils.ops import bias_act, upfirdn2d, conv2d_gradfix, _native
    x = torch.randn(1, 2, 4, 4)
    assert torch.equal(bias_act.bias_act(x, act='relu', gain=1), torch.relu(x))
    assert torch.equal(bias_act.bias_act(x, act='relu', gain=1, impl='ref'), torch.relu(x))
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    assert upfirdn2d.upfirdn2d(x, f, padding=2).shape == (1, 2, 5, 5)
    assert torch.equal(upfirdn2d.upfirdn2d(x, None, impl='ref'), x)
    w = torch.randn(3, 2, 3, 3)
    assert torch.equal(conv2d_gradfix.conv2d(x, w, padding=1), F.conv2d(x, w, padding=1))
    with pytest.raises(AssertionError):
        bias_act.bias_act(x, impl='bogus')
    # a missing / unbuildable plugin is an error, never a detour
    real, was = custom_ops.get_plugin, bias_act._plugin if hasattr(bias_act, '_plugin') else None
    def broken(*a, **k):
        raise RuntimeError('no hipcc')
    custom_ops.get_plugin = broken
    try:
        if hasattr(bias_act, '_plugin'):
            bias_act._plugin = None
        with pytest.raises(RuntimeError):
            bias_act._init()
    finally:
        custom_ops.get_plugin = real
        if hasattr(bias_act, '_plugin'):
            bias_act._plugin = was


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'pasta-gan-plusplus_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', text, flags=re.M), f'{fn} imports the oracle'


def test_setup_filter_and_padding_algebra(golden):
    import numpy as np
    import cases as C
    from torch_utils.ops import upfirdn2d
    g = golden('g1_upfirdn2d.npz')
    np.testing.assert_allclose(upfirdn2d.setup_filter(C.FIR_1331).numpy(), g['setup/1331'], rtol=1e-7)
    np.testing.assert_allclose(upfirdn2d.setup_filter(C.FIR_12).numpy(), g['setup/12'], rtol=1e-7)
    np.testing.assert_allclose(upfirdn2d.setup_filter([1, 2, 3, 4], flip_filter=True, gain=3).numpy(), g['setup/1331_flip_gain'], rtol=1e-6)
    np.testing.assert_allclose(upfirdn2d.setup_filter(C.FIR_1331, separable=True, gain=2).numpy(), g['setup/sep_forced'], rtol=1e-6)
    assert upfirdn2d._parse_padding(3) == (3, 3, 3, 3)
    assert upfirdn2d._parse_padding([1, 2]) == (1, 1, 2, 2)
    assert upfirdn2d._parse_padding([1, 2, 3, 4]) == (1, 2, 3, 4)
    assert upfirdn2d._parse_scaling(2) == (2, 2)
    assert upfirdn2d._get_filter_size(None) == (1, 1)
    assert upfirdn2d._get_filter_size(torch.zeros(3, 5)) == (5, 3)


@pytest.mark.parametrize('k,stride,pad,h,w,opad', [(3, 2, 0, 8, 8, 0), (3, 2, 1, 7, 9, 0), (3, 2, 1, 6, 5, 1), (4, 2, 1, 5, 6, 0), (2, 2, 0, 4, 4, 0), (3, 3, 1, 5, 4, 0)])
def test_transposed_phase_decomposition(k, stride, pad, h, w, opad):
    """conv_transpose2d == union over output phases of small gather-form correlations."""
    from torch_utils.ops import conv2d_mfma
    torch.manual_seed(0)
    cin, cout = 3, 4
    x = torch.randn(2, cin, h, w, dtype=torch.float64)
    wt = torch.randn(cin, cout, k, k, dtype=torch.float64)
    ref = F.conv_transpose2d(x, wt, stride=stride, padding=pad, output_padding=opad)
    out_hw = ((h - 1) * stride - 2 * pad + k + opad, (w - 1) * stride - 2 * pad + k + opad)
    assert tuple(ref.shape[2:]) == out_hw
    phases = conv2d_mfma.transposed_phases(k, k, stride, pad, pad, (h, w), out_hw)
    y = torch.full_like(ref, float('nan'))
    for ph in phases:
        wsel = wt[:, :, ph['ky'], :][:, :, :, ph['kx']].transpose(0, 1)          # OIHW gather-form weights
        jy, jx = len(ph['ky']), len(ph['kx'])
        oh, ow = ph['out_hw']
        py, px = ph['pad']
        # emulate one pg_conv2d_forward launch: y[oy] = sum_t w[t] * x[oy + t - pad], zero outside
        big = F.pad(x, (max(px, 0) + jx, jx + ow, max(py, 0) + jy, jy + oh))
        oy0, ox0 = max(py, 0) + jy - py, max(px, 0) + jx - px
        part = F.conv2d(big, wsel)[:, :, oy0:oy0 + oh, ox0:ox0 + ow]
        y[:, :, ph['off'][0]::stride, ph['off'][1]::stride][:, :, :oh, :ow] = part
    assert not torch.isnan(y).any(), 'some output position is covered by no phase'
    torch.testing.assert_close(y, ref, rtol=1e-10, atol=1e-10)


def test_transposed_phases_rejects_kernel_smaller_than_stride():
    from torch_utils.ops import conv2d_mfma
    assert conv2d_mfma.transposed_phases(1, 1, 2, 0, 0, (4, 4), (7, 7)) is None


def test_infinite_sampler_partitions_ranks():
    from torch_utils import misc
    data = list(range(10))
    streams = []
    for rank in range(2):
        it = iter(misc.InfiniteSampler(data, rank=rank, num_replicas=2, shuffle=True, seed=3, window_size=0))
        streams.append([int(next(it)) for _ in range(10)])
    merged = [v for pair in zip(*streams) for v in pair]
    assert sorted(merged[:10]) == data      # one epoch split across ranks covers every index once


def test_conv2d_resample_plan():
    """Route / padding algebra of conv2d_resample (pure integers), against hand-derived values of the reference's
    formulas (conv2d_resample.py:92-147)."""
    from torch_utils.ops.conv2d_resample import _plan
    # SynthesisLayer up=2: 3x3 kernel, 4x4 FIR, padding 1 -> transposed conv without padding, FIR pad 1 (2H+1 -> 2H)
    p = _plan(3, 3, 4, 4, 2, 1, 1)
    assert p.route == 'transposed' and p.conv_pad == (0, 0) and p.fir_pad == [1, 1, 1, 1]
    # Conv2dLayer down=2, 3x3, padding 1 -> blur with pad 2 then stride-2 conv
    p = _plan(3, 3, 4, 4, 1, 2, 1)
    assert p.route == 'strided' and p.fir_pad == [2, 2, 2, 2]
    # 1x1 skip with down=2 / up=2
    assert _plan(1, 1, 4, 4, 1, 2, 0) == ('pointwise_down', [1, 1, 1, 1], (0, 0))
    assert _plan(1, 1, 4, 4, 2, 1, 0) == ('pointwise_up', [2, 1, 2, 1], (0, 0))
    # plain convs
    assert _plan(3, 3, 4, 4, 1, 1, 1) == ('plain', [1, 1, 1, 1], (1, 1))
    assert _plan(7, 7, 4, 4, 1, 1, 3).conv_pad == (3, 3)
    # asymmetric padding falls to the generic route
    assert _plan(3, 3, 4, 4, 1, 1, [1, 0, 2, 1]).route == 'generic'
    # a transposed conv with a larger kernel absorbs part of the (negative) FIR padding itself
    p = _plan(5, 5, 4, 4, 2, 1, 0)
    assert p.route == 'transposed' and p.conv_pad == (2, 2) and p.fir_pad == [0, 0, 0, 0]


def test_stack_batched_affine_matches_per_layer_affine():
    """SynthesisStack.all_styles (inference: one GEMM + one gather for every affine layer) == affine(w) (* ToRGB weight gain) layer by layer."""
    import torch
    from training import networks
    net = networks.SynthesisStack(w_dim=512, img_resolution=64, channel_base=2048, channel_max=64).eval()
    gen = torch.Generator().manual_seed(3)
    for p in net.parameters():
        p.data.copy_(torch.randn(p.shape, generator=gen))
    ws = torch.randn([3, net.num_ws, 512], generator=gen)
    with torch.no_grad():
        styles = net.all_styles(ws)
        start = 0
        for res in net.block_resolutions:
            block = getattr(net, f'b{res}')
            assert len(styles[res]) == block.num_conv + block.num_torgb
            for (layer, i, g), got in zip(block.affine_layers(), styles[res]):
                ref = layer.affine(ws[:, start + i]) * g
                assert got.is_contiguous() and got.shape == ref.shape
                assert float((ref - got).abs().max()) <= 2e-6 * float(ref.abs().max())
            start += block.num_conv
        # a parameter update invalidates the concatenated weights
        net.b8.conv1.affine.weight.add_(1.0)
        again = net.all_styles(ws)
        ref = net.b8.conv1.affine(ws[:, 0])
        assert float((ref - again[8][0]).abs().max()) <= 2e-6 * float(ref.abs().max())


def test_winograd_launch_policy(monkeypatch):
    """conv2d_mfma.use_winograd: 0 = direct, 1 = F(2x2,3x3), 2 = F(4x4,3x3) (round 3) -- pure host logic, no plugin needed."""
    import importlib
    conv2d_mfma = importlib.import_module('torch_utils.ops.conv2d_mfma')      # (conftest.py puts the package on sys.path)
    monkeypatch.delenv('PG_CONV_ALGO', raising=False)
    uw = conv2d_mfma.use_winograd
    from torch_utils.ops.conv2d_mfma import F4_FORM as FN      # images narrower than 64 pixels: 3 = conv2d_wino4b.h (two workgroups per CU); PG_WINO4B=0: 2
    from torch_utils.ops.conv2d_mfma import F4_WIDE as FW      # images at least 64 pixels wide: 4 = conv2d_wino4.h with the GEMM on the bf16 pipe (three-term splits); PG_WINO4_X3=0: 2
    assert FW in (2, 4) and FN in (2, 3)
    assert uw(3, 3, 1, 128, 128, pad=(1, 1), hw=(256, 256)) == FW and uw(3, 3, 1, 64, 64, pad=(1, 1), hw=(512, 512)) == FW
    assert uw(3, 3, 1, 512, 512, pad=(1, 1), hw=(32, 32)) == FN and uw(3, 3, 1, 512, 512, pad=(1, 1), hw=(16, 16)) == 1 and uw(3, 3, 1, 512, 512, pad=(1, 1), hw=(8, 8)) == 1
    assert uw(3, 3, 1, 128, 128, pad=(1, 1)) == 1                                   # no image size given: F(2x2)
    assert uw(3, 3, 1, 128, 128, pad=(1, 1), hw=(256, 254)) == 1                   # width no multiple of 4
    assert uw(3, 3, 1, 128, 128, pad=(2, 2), hw=(256, 256)) == 1                   # output width 258
    assert uw(3, 3, 1, 128, 128, pad=(1, 1), hw=(256, 256), xf=True) == 1          # input pre-activation stage: F(2x2) has the prologue
    assert uw(3, 3, 1, 96, 128, pad=(1, 1), hw=(256, 256)) == 1 and uw(3, 3, 1, 128, 32, pad=(1, 1), hw=(256, 256)) == 1
    assert uw(3, 3, 1, 16, 16, hw=(256, 256)) == 0 and uw(3, 3, 2, 128, 128) == 0 and uw(1, 1, 1, 128, 128) == 0 and uw(3, 3, 1, 64, 16, pad=(1, 5)) == 0
    monkeypatch.setenv('PG_CONV_ALGO', 'winograd2')
    assert uw(3, 3, 1, 128, 128, pad=(1, 1), hw=(256, 256)) == 1
    monkeypatch.setenv('PG_CONV_ALGO', 'direct')
    assert uw(3, 3, 1, 128, 128, pad=(1, 1), hw=(256, 256)) == 0


def test_modconv16_policy_is_pure_host_logic():
    """The form a 16-bit modulated convolution takes (training/networks._modconv16_policy: composite up = 2 kernels, one shared weight pack for the
    batch where weights outweigh activations) is decided from shapes alone -- the layers and the stack's batched style preparation must agree on it."""
    import torch
    from training import networks as PN
    from torch_utils.ops import upfirdn2d
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    comp, merged, shared, tpad, fir_pad, fused_x = PN._modconv16_policy((1024, 1024, 3, 3), (8, 8), 2, 1, f)
    assert comp and not merged and not fused_x and shared and tuple(tpad) == (0, 0) and list(fir_pad) == [1, 1, 1, 1]      # 8^2: below the fused-x form's smallest image
    comp, merged, shared, tpad, fir_pad, fused_x = PN._modconv16_policy((32, 64, 3, 3), (512, 512), 2, 1, f)
    assert fused_x and not comp and not shared                 # round 5: the y half of the FIR in the weights, the x half in the epilogue; activations dominate: per-sample weights
    comp, merged, shared, tpad, fir_pad, fused_x = PN._modconv16_policy((512, 1024, 3, 3), (32, 32), 2, 1, f)
    assert fused_x and shared == (512 * 24 > 32 * 32)          # 24 tap slots per weight (18 non-zero); shared where the packed weights outweigh one sample's pixels (ratio 1 since round 5: +2 % on config 5)
    assert PN._modconv16_policy((32, 64, 3, 3), (512, 512), 2, 1, upfirdn2d.setup_filter([1, 3, 3, 1]) + torch.eye(4) * 0.01)[5] is False      # not separable: composite
    comp, merged, shared, tpad, fir_pad, fused_x = PN._modconv16_policy((512, 512, 3, 3), (64, 64), 1, 1, f)
    assert not comp and not merged and not fused_x and shared == (512 * 9 > 64 * 64) and tpad is None
    fy, fx = PN._separable_taps(f)
    assert torch.allclose(torch.outer(fy, fx), f) and PN._separable_taps(f) is PN._separable_taps(f)      # decided once per filter tensor
    assert PN._is_1331(f) and not PN._is_1331(upfirdn2d.setup_filter([1, 2, 1])) and not PN._is_1331(upfirdn2d.setup_filter([1, 3, 3, 1]) * 2)


def test_bf16x3_weight_gradient_rule_is_pure_host_logic(monkeypatch):
    """Which float32 weight gradients take the three-term bf16 operand split (torch_utils/ops/conv2d_mfma._bf16x3_wanted, round 5): by default the 3x3 layers,
    stride 1 or 2, with at least 64 channels on both sides -- where the training step measured it faster; never a geometry the 16-bit kernel does not cover."""
    from torch_utils.ops import conv2d_mfma
    want = conv2d_mfma._bf16x3_wanted
    monkeypatch.delenv('PG_WGRAD_BF16X3', raising=False)
    monkeypatch.delenv('PG_WGRAD_BF16X3_MIN_C', raising=False)
    monkeypatch.delenv('PG_WGRAD_BF16X3_K1_MIN_C', raising=False)
    assert want(4, 128, 128, 256, 256, 3, 3, 1) and want(4, 64, 64, 512, 512, 3, 3, 1) and want(4, 64, 128, 513, 513, 3, 3, 2)
    assert not want(4, 32, 64, 512, 512, 3, 3, 1) and not want(4, 64, 32, 512, 512, 3, 3, 1)          # narrower than 64 channels: the splitting passes outweigh the multiplies
    assert not want(4, 128, 128, 256, 256, 1, 1, 1)                                                  # 1x1: measured no gain
    assert not want(4, 3, 64, 512, 512, 7, 7, 1) and not want(4, 68, 128, 64, 64, 3, 3, 1)           # 7x7 stem; channel counts that are no multiple of 8
    monkeypatch.setenv('PG_WGRAD_BF16X3', '0')
    assert not want(4, 128, 128, 256, 256, 3, 3, 1)
    monkeypatch.setenv('PG_WGRAD_BF16X3', '1')
    assert want(4, 16, 24, 64, 64, 3, 3, 1) and want(4, 128, 128, 64, 64, 1, 1, 1) and not want(4, 128, 128, 64, 64, 1, 1, 2)
    monkeypatch.setenv('PG_WGRAD_BF16X3', 'auto')
    monkeypatch.setenv('PG_WGRAD_BF16X3_K1_MIN_C', '256')
    assert want(4, 256, 512, 64, 64, 1, 1, 1) and not want(4, 128, 512, 64, 64, 1, 1, 1)


def test_custom_ops_clean_is_per_plugin_and_spares_live_links(tmp_path, monkeypatch):
    """`clean()` sweeps compiler temporaries plugin by plugin under each plugin's build lock, matches exact plugin names (not a prefix glob) and never
    deletes the `<so>.tmp<pid>` of a link that is still running."""
    import subprocess
    import sys
    from torch_utils import custom_ops
    csrc, dev = tmp_path / 'csrc', tmp_path / 'csrc' / 'dev'
    dev.mkdir(parents=True)
    monkeypatch.setattr(custom_ops, 'CSRC_DIR', str(csrc))
    monkeypatch.setattr(custom_ops, 'DEV_DIR', str(dev))
    sleeper = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(60)'])
    try:
        dead = subprocess.Popen([sys.executable, '-c', 'pass'])
        dead.wait()
        live_tmp = csrc / f'conv2d_plugin.so.tmp{sleeper.pid}'
        dead_tmp = csrc / f'conv2d_plugin.so.tmp{dead.pid}'
        host_tmp = csrc / 'conv2d_plugin.so.3.host-x86_64-unknown-linux-gnu.o'
        other = dev / 'wino4b_exp10.so.tmp999999999'
        keep = [csrc / 'conv2d_plugin.so', csrc / 'conv2d_plugin.so.digest', dev / 'wino4b_exp10.so']
        for f in [live_tmp, dead_tmp, host_tmp, other] + keep:
            f.write_bytes(b'x')
        gone = custom_ops.clean(only='wino4b_exp1')                 # a sibling name: must not match wino4b_exp10's files
        assert gone == [] and other.exists()
        gone = custom_ops.clean()
        assert live_tmp.exists() and not dead_tmp.exists() and not host_tmp.exists() and not other.exists()
        assert all(f.exists() for f in keep) and len(gone) == 3
    finally:
        sleeper.kill()
        sleeper.wait()
