"""Pins the CPU oracle (oracle/ops_ref.py, oracle/network_ref.py) against the golden
vectors produced by the REFERENCE ITSELF (tests/golden/make_golden.py).  CPU only."""

import math

import numpy as np
import pytest
import torch

import cases as C
from detgen import det_tensor, fill_module_, synthesis_inputs
from oracle import ops_ref as R
from oracle import network_ref as NR

F32_TOL = dict(rtol=2e-5, atol=2e-6)


def _filter(spec, name):
    if spec is None:
        return None
    kind, v = spec
    return R.setup_filter(v) if kind == 'taps' else det_tensor(name + '.f', v, 'uniform')


def close(a, b, rtol, atol):
    a = a.detach().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    np.testing.assert_allclose(a, np.asarray(b, dtype=np.float64), rtol=rtol, atol=atol)


def test_setup_filter(golden):
    g = golden('g1_upfirdn2d.npz')
    close(R.setup_filter(C.FIR_1331), g['setup/1331'], 1e-7, 0)
    close(R.setup_filter(C.FIR_12), g['setup/12'], 1e-7, 0)
    close(R.setup_filter([1, 2, 3, 4], flip_filter=True, gain=3), g['setup/1331_flip_gain'], 1e-6, 0)
    close(R.setup_filter(C.FIR_1331, separable=True, gain=2), g['setup/sep_forced'], 1e-6, 0)


@pytest.mark.parametrize('case', C.UPFIRDN2D_CASES, ids=[c[0] for c in C.UPFIRDN2D_CASES])
def test_upfirdn2d(golden, case):
    g = golden('g1_upfirdn2d.npz')
    name, xs, fspec, up, down, pad, flip, gain = case
    f = _filter(fspec, name)
    x = det_tensor(name + '.x', xs)
    y = R.upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
    assert tuple(y.shape) == g[f'{name}/y'].shape
    scale = max(1.0, float(np.abs(g[f'{name}/y']).max()))
    close(y, g[f'{name}/y'], rtol=2e-5, atol=2e-6 * scale)
    # backward == upfirdn2d with swapped factors (upfirdn2d.py:245-264)
    dy = det_tensor(name + '.dy', y.shape)
    kw = R.upfirdn2d_backward_params(x.shape, y.shape, f, up, down, pad, flip)
    close(R.upfirdn2d(dy, f, gain=gain, **kw), g[f'{name}/dx'], rtol=2e-5, atol=2e-6 * scale)
    if name in C.UPFIRDN2D_DTYPE_CASES:
        close(R.upfirdn2d(x.double(), f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain), g[f'{name}/y_f64'], 1e-6, 1e-7)
        close(R.upfirdn2d(x.half(), f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain), g[f'{name}/y_f16'], 4e-3, 4e-3)
        close(R.upfirdn2d(x.bfloat16(), f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain), g[f'{name}/y_bf16'], 3e-2, 3e-2)


@pytest.mark.parametrize('act', C.ACTS)
def test_bias_act(golden, act):
    g = golden('g2_bias_act.npz')
    for vname, has_b, gain, clamp, dim, xs in C.BIAS_ACT_VARIANTS:
        name = f'{act}.{vname}'
        x = det_tensor(name + '.x', xs, scale=2.0)
        b = det_tensor(name + '.b', [xs[dim]]) if has_b else None
        y = R.bias_act(x, b, dim=dim, act=act, gain=gain, clamp=clamp)
        close(y, g[f'{name}/y'], **F32_TOL)
        dy = det_tensor(name + '.dy', xs)
        dx, db = R.bias_act_grad(dy, x, b, dim=dim, act=act, gain=gain, clamp=clamp)
        close(dx, g[f'{name}/dx'], **F32_TOL)
        if has_b:
            close(db, g[f'{name}/db'], rtol=1e-4, atol=1e-5)
        if vname in ('bias', 'bias_clamp'):
            close(R.bias_act(x.double(), b.double(), dim=dim, act=act, gain=gain, clamp=clamp), g[f'{name}/y_f64'], 1e-6, 1e-7)


@pytest.mark.parametrize('case', C.CONV2D_RESAMPLE_CASES, ids=[c[0] for c in C.CONV2D_RESAMPLE_CASES])
def test_conv2d_resample(golden, case):
    g = golden('g3_conv2d_resample.npz')
    name, xs, wsh, taps, up, down, pad, groups, flipw = case
    f = R.setup_filter(taps)
    x = det_tensor(name + '.x', xs).requires_grad_(True)
    w = det_tensor(name + '.w', wsh, scale=1 / math.sqrt(wsh[1] * wsh[2] * wsh[3])).requires_grad_(True)
    y = R.conv2d_resample(x, w, f=f, up=up, down=down, padding=pad, groups=groups, flip_weight=flipw)
    close(y, g[f'{name}/y'], rtol=1e-4, atol=1e-5)
    dx, dw = torch.autograd.grad(y, [x, w], det_tensor(name + '.dy', y.shape))
    close(dx, g[f'{name}/dx'], rtol=1e-4, atol=1e-5)
    close(dw, g[f'{name}/dw'], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('case', C.MODCONV_CASES, ids=[c[0] for c in C.MODCONV_CASES])
def test_modulated_conv2d(golden, case):
    g = golden('g4_modconv.npz')
    name, n, cin, cout, k, h, up, demod, fused, noise_kind = case
    f = R.setup_filter(C.FIR_1331)
    x = det_tensor(name + '.x', [n, cin, h, h])
    w = det_tensor(name + '.w', [cout, cin, k, k])
    s = det_tensor(name + '.s', [n, cin]) + 1.0
    hh = h * up
    noise = {'none': None, 'const': det_tensor(name + '.noise', [hh, hh]) * 0.1,
             'per_sample': det_tensor(name + '.noise', [n, 1, hh, hh]) * 0.1}[noise_kind]
    y = R.modulated_conv2d(x, w, s, noise=noise, up=up, padding=k // 2, resample_filter=f, demodulate=demod,
                           flip_weight=(up == 1), fused_modconv=fused)
    close(y, g[f'{name}/y'], rtol=1e-4, atol=1e-5)


def test_modulated_conv2d_fp16_prenorm(golden):
    g = golden('g4_modconv.npz')
    if 'fp16_prenorm/y' not in g:
        pytest.skip('reference could not run fp16 conv on CPU when the fixture was made')
    name = 'fp16_prenorm'
    x = det_tensor(name + '.x', [2, 5, 9, 9]).half()
    y = R.modulated_conv2d(x, det_tensor(name + '.w', [6, 5, 3, 3]), det_tensor(name + '.s', [2, 5]) + 1.0, padding=1,
                           resample_filter=R.setup_filter(C.FIR_1331), fused_modconv=False)
    close(y.float(), g[f'{name}/y'], rtol=1e-2, atol=1e-2)


def test_blocks(golden):
    g = golden('g5_blocks.npz')
    tol = dict(rtol=2e-4, atol=2e-5)
    with torch.no_grad():
        blk = fill_module_(NR.Spade_ResBlockV4_512(8, 8, spade_channels=5), 'g5.spade.')
        close(blk(det_tensor('g5.spade.x', [2, 8, 24, 24]), det_tensor('g5.spade.feat', [2, 5, 24, 24])), g['spade/y'], **tol)
        rb = fill_module_(NR.ResBlock(6, 10, activation='relu', down=2), 'g5.resdown.')
        close(rb(det_tensor('g5.resdown.x', [2, 6, 32, 32])), g['resdown/y'], **tol)
        rb1 = fill_module_(NR.ResBlock(6, 6, activation='relu'), 'g5.res.')
        close(rb1(det_tensor('g5.res.x', [2, 6, 20, 20])), g['res/y'], **tol)
        c7 = fill_module_(NR.Conv2dLayer(3, 8, kernel_size=7, activation='relu'), 'g5.conv7.')
        close(c7(det_tensor('g5.conv7.x', [2, 3, 20, 20])), g['conv7/y'], **tol)
        cup = fill_module_(NR.Conv2dLayer(4, 6, kernel_size=3, activation='lrelu', up=2, conv_clamp=0.5), 'g5.convup.')
        close(cup(det_tensor('g5.convup.x', [2, 4, 8, 8]), gain=math.sqrt(0.5)), g['convup/y'], **tol)
        fc = fill_module_(NR.FullyConnectedLayer(12, 7, bias_init=1), 'g5.fc.')
        close(fc(det_tensor('g5.fc.x', [3, 12])), g['fc/y'], **tol)
        fca = fill_module_(NR.FullyConnectedLayer(12, 7, activation='lrelu', lr_multiplier=0.01), 'g5.fca.')
        close(fca(det_tensor('g5.fca.x', [3, 12])), g['fca/y'], **tol)
        sl = fill_module_(NR.SynthesisLayer(5, 6, w_dim=12, resolution=16, up=2, conv_clamp=256), 'g5.synup.').eval()
        xw = det_tensor('g5.synup.x', [2, 5, 8, 8]), det_tensor('g5.synup.w', [2, 12])
        close(sl(*xw, noise_mode='const', fused_modconv=True), g['synup_fused/y'], **tol)
        close(sl(*xw, noise_mode='const', fused_modconv=False, gain=math.sqrt(0.5)), g['synup_nonfused/y'], **tol)
        tr = fill_module_(NR.ToRGBLayerFull(6, 3, w_dim=12, conv_clamp=256, is_last=True, is_style=True), 'g5.torgb.')
        yi, yp = tr(det_tensor('g5.torgb.x', [2, 6, 16, 16]), det_tensor('g5.torgb.w', [2, 12]))
        close(yi, g['torgb/img'], **tol)
        close(yp, g['torgb/parsing'], **tol)


@pytest.mark.parametrize('variant,labels,fused', [('labels', True, None), ('argmax', False, None), ('labels_nonfused', True, False)])
def test_synthesis_network_reduced(golden, variant, labels, fused):
    """Full-resolution (512^2) reduced-width SynthesisNetworkFull_v18 vs the reference classes."""
    g = golden('g6_synthesis.npz')
    torch.manual_seed(0)
    net = NR.SynthesisNetworkFull_v18(**C.G6_KW)
    assert sorted(n for n, _ in net.named_parameters()) == list(g['param_names'])
    assert net.num_ws == int(g['num_ws']) == 14
    fill_module_(net, 'g6.')
    net.eval()
    inp = synthesis_inputs(1, w_dim=C.G6_KW['w_dim'], num_ws=net.num_ws, feat_ch=C.G6_FEAT_CH, seed_tag='g6', labels=labels)
    kw = dict(noise_mode='const')
    if fused is not None:
        kw['fused_modconv'] = fused
    with torch.no_grad():
        img, fimg, pp = net(inp['ws'], inp['pose_feat'], inp['cat_feat'], inp['denorm_upper_input'], inp['denorm_lower_input'],
                            inp['denorm_upper_mask'], inp['denorm_lower_mask'], inp['gt_parsing'], **kw)
    y0, y1, x0, x1 = C.G6_CROP
    for nm, t in (('img', img), ('finetune_img', fimg), ('pred_parsing', pp)):
        scale = float(np.abs(g[f'{variant}/{nm}_sub']).max())
        close(t[..., ::C.G6_SUB, ::C.G6_SUB], g[f'{variant}/{nm}_sub'], rtol=1e-3, atol=2e-4 * scale)
        close(t[..., y0:y1, x0:x1], g[f'{variant}/{nm}_crop'], rtol=1e-3, atol=2e-4 * scale)
        np.testing.assert_allclose(float(t.double().abs().sum()), float(g[f'{variant}/{nm}_abssum']), rtol=1e-4)


def test_encoders_and_mapping(golden):
    """Rows of SURVEY section 8f (f1): encoders + mapping of the reference, reduced widths."""
    g = golden('g7_encoders.npz')
    tol = dict(rtol=3e-4, atol=3e-5)
    with torch.no_grad():
        ce = fill_module_(NR.ConstEncoderNetwork(input_nc=5, output_nc=64, ngf=8, n_downsampling=6), 'g7.const.')
        close(ce(det_tensor('g7.const.x', [2, 5, 128, 128], 'uniform')), g['const/y'], **tol)
        se = fill_module_(NR.StyleEncoderNetworkV18(input_nc=45, output_nc=64, ngf=8, n_downsampling=6), 'g7.style.')
        code, feats = se(det_tensor('g7.style.parts', [2, 45, 32, 32], 'uniform'), det_tensor('g7.style.retain', [2, 6, 64, 64], 'uniform'))
        close(code, g['style/code'], rtol=1e-3, atol=1e-4)
        for i, f in enumerate(feats):
            close(f, g[f'style/feat{i}'], **tol)
        mp = fill_module_(NR.MappingNetwork(z_dim=0, c_dim=64, w_dim=32, num_ws=14, num_layers=1), 'g7.map.').eval()
        close(mp(torch.zeros([2, 0]), det_tensor('g7.map.c', [2, 64])), g['map/ws'], **tol)
        mp2 = fill_module_(NR.MappingNetwork(z_dim=16, c_dim=8, w_dim=32, num_ws=5, num_layers=3), 'g7.map2.').eval()
        mp2.w_avg.copy_(det_tensor('g7.map2.w_avg', [32]))
        close(mp2(det_tensor('g7.map2.z', [3, 16]), det_tensor('g7.map2.c', [3, 8]), truncation_psi=0.7, truncation_cutoff=3), g['map2/ws'], **tol)
        dn = fill_module_(NR.Dense(6, 10), 'g7.dense.')
        close(dn(det_tensor('g7.dense.x', [2, 6, 9, 11])), g['dense/y'], **tol)


G8_KW = dict(c_dim=16, img_resolution=32, img_channels=6, channel_base=512, channel_max=32, conv_clamp=256,
             mapping_kwargs=dict(num_layers=2), epilogue_kwargs=dict(mbstd_group_size=2))


def test_discriminator_with_r1_double_backward(golden):
    """Row f2: Discriminator logits and the R1 penalty's parameter gradients (double backward, loss_fullbody.py:262-274)."""
    g = golden('g8_discriminator.npz')
    d = fill_module_(NR.Discriminator(**G8_KW), 'g8.d.')
    assert [n for n, _ in d.named_parameters()] == list(g['r1_grad_names'])
    img = det_tensor('g8.img', [4, 6, 32, 32], 'uniform').requires_grad_(True)
    logits = d(img, det_tensor('g8.c', [4, 16]))
    close(logits, g['logits'], rtol=1e-3, atol=1e-4)
    gi, = torch.autograd.grad(logits.sum(), img, create_graph=True)
    pen = gi.square().sum([1, 2, 3])
    close(pen, g['r1_penalty'], rtol=2e-3, atol=1e-6)
    grads = torch.autograd.grad(pen.sum(), list(d.parameters()), allow_unused=True)
    got = np.array([float(x.abs().sum()) if x is not None else 0.0 for x in grads])
    np.testing.assert_allclose(got, g['r1_grad_abssum'], rtol=5e-3, atol=1e-6)
