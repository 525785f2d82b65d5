"""ctypes binding of ``conv2d_plugin`` (csrc/conv2d*.hip): the fp32 MFMA implicit-GEMM
convolution with fused prologue/epilogue and its support kernels.

This module has no counterpart in the reference -- there the contraction is cuDNN's,
reached through ``conv2d_gradfix`` (torch_utils/ops/conv2d_gradfix.py:35-43).  It is the
native layer under this package's ``conv2d_gradfix`` / ``conv2d_resample`` /
``training.networks``; forward only (gradients are attached in ``conv2d_gradfix``).
"""

import ctypes
import os

import torch

from .. import custom_ops
from . import _native as nat

ACT_INDEX = {'linear': 1, 'relu': 2, 'lrelu': 3, 'tanh': 4, 'sigmoid': 5, 'elu': 6, 'selu': 7, 'softplus': 8, 'swish': 9}

FUSED_ACTS = ('linear', 'relu', 'lrelu')     # activations the conv prologue/epilogue can apply (csrc/conv2d_kernel.h)

# (KH, KW, stride) geometries instantiated in csrc/conv2d_inst_*.hip
SUPPORTED = {(3, 3, 1), (1, 1, 1), (2, 2, 1), (2, 1, 1), (1, 2, 1), (7, 7, 1), (3, 3, 2), (1, 1, 2)}


class Fusion(ctypes.Structure):
    """Mirror of ``pg_conv2d_fusion`` (include/pasta_gan_ops.h)."""
    _fields_ = [
        ('in_scale', ctypes.c_void_p), ('in_bias', ctypes.c_void_p), ('in_act', ctypes.c_int), ('in_alpha', ctypes.c_float),
        ('in_gain', ctypes.c_float), ('in_clamp', ctypes.c_float),
        ('out_scale', ctypes.c_void_p), ('noise', ctypes.c_void_p), ('noise_batch_stride', ctypes.c_int64), ('noise_gain', ctypes.c_float),
        ('bias', ctypes.c_void_p), ('act', ctypes.c_int), ('alpha', ctypes.c_float), ('gain', ctypes.c_float), ('clamp', ctypes.c_float),
        ('residual', ctypes.c_void_p),
        ('spade_x', ctypes.c_void_p), ('spade_mean', ctypes.c_void_p), ('spade_rstd', ctypes.c_void_p),
        ('x2', ctypes.c_void_p), ('cin_split', ctypes.c_int),
        ('stats_partial', ctypes.c_void_p),
    ]


_plugin = None

# Optional launch timeline for bench.py's roofline: when a list, every convolution launch appends
# (geometry, algorithmic FLOPs, start event, end event, algorithmic bytes) recorded on the launch stream.
_timeline = None
_wgrad_timeline = None      # the same for the weight-gradient launches (bench.py --mode train): its own list, so a training run does not pay two events on each of its forward launches


def reserve_cus(n):
    """Keep `n` CUs free of the plugin's persistent grids (pg_conv2d_reserve_cus): room for RCCL's channel workgroups beside the backward pass.  Returns the CU
    count the grids use from now on."""
    lib = _init().lib
    lib.pg_conv2d_reserve_cus.restype = ctypes.c_int
    lib.pg_conv2d_reserve_cus.argtypes = [ctypes.c_int]
    return int(lib.pg_conv2d_reserve_cus(int(n)))


def start_timeline():
    global _timeline
    _timeline = []
    return _timeline


def start_wgrad_timeline():
    global _wgrad_timeline
    _wgrad_timeline = []
    return _wgrad_timeline


def stop_wgrad_timeline():
    global _wgrad_timeline
    tl, _wgrad_timeline = _wgrad_timeline, None
    return tl


def stop_timeline():
    global _timeline
    tl, _timeline = _timeline, None
    return tl


def _init(plugin_name='conv2d_plugin'):
    global _plugin
    if _plugin is None:
        plugin = custom_ops.get_plugin(plugin_name)
        lib = plugin.lib
        i, f, vp, i64 = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_int64
        lib.pg_conv2d_packed_size.restype = i64
        lib.pg_conv2d_packed_size.argtypes = [i, i, i, i]
        lib.pg_conv2d_pack_weight.restype = i
        lib.pg_conv2d_pack_weight.argtypes = [vp, vp, i, i, i, i, f, i, i, vp]
        lib.pg_conv2d_forward.restype = i
        lib.pg_conv2d_forward.argtypes = [vp, vp, vp, i, i, i, i, i, i, i, i, i, i, i, i, ctypes.POINTER(i64), i, i, i, i, ctypes.POINTER(Fusion), vp]
        lib.pg_conv2d_splitk_plan.restype = i
        lib.pg_conv2d_splitk_plan.argtypes = [i] * 8
        lib.pg_conv2d_forward_splitk.restype = i
        lib.pg_conv2d_forward_splitk.argtypes = [vp, vp, vp, i, i, i, i, i, i, i, i, i, i, i, i, ctypes.POINTER(i64), i, i, i, i, ctypes.POINTER(Fusion), vp, i, vp]
        lib.pg_conv2d_winograd_packed_size.restype = i64
        lib.pg_conv2d_winograd_packed_size.argtypes = [i, i]
        lib.pg_conv2d_winograd_pack_weight.restype = i
        lib.pg_conv2d_winograd_pack_weight.argtypes = [vp, vp, i, i, f, i, i, vp]
        lib.pg_conv2d_winograd_forward.restype = i
        lib.pg_conv2d_winograd_forward.argtypes = [vp, vp, vp, i, i, i, i, i, i, i, i, i, ctypes.POINTER(i64), ctypes.POINTER(Fusion), vp]
        lib.pg_conv2d_winograd4_packed_size.restype = i64
        lib.pg_conv2d_winograd4_packed_size.argtypes = [i, i]
        lib.pg_conv2d_winograd4_pack_weight.restype = i
        lib.pg_conv2d_winograd4_pack_weight.argtypes = [vp, vp, i, i, f, i, i, vp]
        lib.pg_conv2d_winograd4_forward.restype = i
        lib.pg_conv2d_winograd4_forward.argtypes = [vp, vp, vp, i, i, i, i, i, i, i, i, i, ctypes.POINTER(i64), ctypes.POINTER(Fusion), vp]
        lib.pg_conv2d_winograd4x3_packed_size.restype = i64
        lib.pg_conv2d_winograd4x3_packed_size.argtypes = [i, i]
        lib.pg_conv2d_winograd4x3_pack_weight.restype = i
        lib.pg_conv2d_winograd4x3_pack_weight.argtypes = [vp, vp, i, i, f, i, i, vp]
        lib.pg_conv2d_winograd4x3_forward.restype = i
        lib.pg_conv2d_winograd4x3_forward.argtypes = [vp, vp, vp, i, i, i, i, i, i, i, i, i, ctypes.POINTER(i64), ctypes.POINTER(Fusion), vp]
        lib.pg_conv2d_winograd4_stats_tiles.restype = i
        lib.pg_conv2d_winograd4_stats_tiles.argtypes = [i, i]
        lib.pg_instance_norm_finish.restype = i
        lib.pg_instance_norm_finish.argtypes = [vp, vp, vp, i, i, i, i, f, vp]
        lib.pg_conv2d_winograd4b_pack_weight.restype = i
        lib.pg_conv2d_winograd4b_pack_weight.argtypes = [vp, vp, i, i, f, i, i, vp]
        lib.pg_conv2d_winograd4b_forward.restype = i
        lib.pg_conv2d_winograd4b_forward.argtypes = [vp, vp, vp, i, i, i, i, i, i, i, i, i, ctypes.POINTER(i64), ctypes.POINTER(Fusion), vp]
        lib.pg_spade_masked_sums.restype = i
        lib.pg_spade_masked_sums.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, vp]
        lib.pg_spade_feat_assemble.restype = i
        lib.pg_spade_feat_assemble.argtypes = [vp] * 11 + [i, i, i, i, vp]
        lib.pg_modconv_dcoefs.restype = i
        lib.pg_modconv_dcoefs.argtypes = [vp, vp, vp, i, i, i, i, f, vp]
        lib.pg_modconv_w2.restype = i
        lib.pg_modconv_w2.argtypes = [vp, vp, i, i, i, f, vp]
        lib.pg_modconv_prep.restype = i
        lib.pg_modconv_prep.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, i, vp]
        lib.pg_modconv_prep_batched.restype = i
        lib.pg_modconv_prep_batched.argtypes = [ctypes.POINTER(PrepJobs), i, vp]
        lib.pg_instance_norm_stats.restype = i
        lib.pg_instance_norm_stats.argtypes = [vp, vp, vp, i, i64, f, vp]
        lib.pg_spade_norm.restype = i
        lib.pg_spade_norm.argtypes = [vp, vp, vp, vp, vp, vp, i, i64, vp]
        lib.pg_spade_train_forward.restype = i
        lib.pg_spade_train_forward.argtypes = [vp] * 6 + [i, i, i64, i64, i64, vp]
        lib.pg_spade_train_backward.restype = i
        lib.pg_spade_train_backward.argtypes = [vp] * 9 + [i, i, i64, i64, i64, i64, vp]
        lib.pg_conv2d_wgrad_plan.restype = i
        lib.pg_conv2d_wgrad_plan.argtypes = [i] * 8
        lib.pg_conv2d_wgrad.restype = i
        lib.pg_conv2d_wgrad.argtypes = [vp, vp, vp, vp] + [i] * 13 + [vp]
        lib.pg_conv2d_up2_forward.restype = i
        lib.pg_conv2d_up2_forward.argtypes = [vp, vp, vp, i, i, i, i, i, ctypes.POINTER(ctypes.c_int64), vp, vp, vp]
        lib.pg_split3_bf16_cl.restype = i
        lib.pg_split3_bf16_cl.argtypes = [vp, vp, i, i, i64, vp]
        lib.pg_conv2d16_wgrad_x3.restype = i
        lib.pg_conv2d16_wgrad_x3.argtypes = [vp, vp, vp, vp] + [i] * 13 + [vp]
        lib.pg_conv2d16_wgrad_plan.restype = i
        lib.pg_conv2d16_wgrad_plan.argtypes = [i] * 8
        lib.pg_conv2d_up2_splitk_plan.restype = i
        lib.pg_conv2d_up2_splitk_plan.argtypes = [i, i, i, i, i]
        lib.pg_conv2d_up2x3_packed_size.restype = i64
        lib.pg_conv2d_up2x3_packed_size.argtypes = [i, i]
        lib.pg_conv2d_up2x3_pack_weight.restype = i
        lib.pg_conv2d_up2x3_pack_weight.argtypes = [vp, vp, i, i, vp]
        lib.pg_conv2d_up2x3_forward.restype = i
        lib.pg_conv2d_up2x3_forward.argtypes = [vp, vp, vp, vp, i, i, i, i, i, ctypes.POINTER(ctypes.c_int64), vp, vp, vp, vp]
        lib.pg_conv2d_up2_forward_splitk.restype = i
        lib.pg_conv2d_up2_forward_splitk.argtypes = [vp, vp, vp, i, i, i, i, i, ctypes.POINTER(ctypes.c_int64), vp, vp, vp, i, vp]
        lib.pg_conv2d_stem7x3_packed_size.restype = i64
        lib.pg_conv2d_stem7x3_packed_size.argtypes = [i]
        lib.pg_conv2d_stem7x3_pack_weight.restype = i
        lib.pg_conv2d_stem7x3_pack_weight.argtypes = [vp, vp, i, ctypes.c_float, i, vp]
        lib.pg_conv2d_stem7x3_forward.restype = i
        lib.pg_conv2d_stem7x3_forward.argtypes = [vp, vp, vp, i, i, i, i, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(Fusion), vp]
        lib.pg_conv3x3_cin1.restype = i
        lib.pg_conv3x3_cin1.argtypes = [vp, vp, vp, i, i, i, i, f, i, vp]
        lib.pg_conv1x1_small.restype = i
        lib.pg_conv1x1_small.argtypes = [vp] * 6 + [i, i, i64, i, f, f, vp]
        _plugin = plugin
    return _plugin


# Which F(4x4,3x3) kernel the policy hands out: 2 = csrc/conv2d_wino4.h (one 12-wave workgroup per CU, 8 x 64-pixel tiles, round 3), 3 = csrc/conv2d_wino4b.h
# (two 8-wave workgroups per CU on the 16x16x4 MFMA, 8 x 32-pixel tiles, round 4).  Measured on the config-2 step (same box, profiles/r04_wino4_forms.txt): the
# two-workgroup form is 1.5x faster on 32-pixel-wide images (no half-empty tiles, twice the workgroups) and 1-11 % slower everywhere else (twice the weight
# stream from L2), so it serves images narrower than 64 pixels.  PG_WINO4B=0: never, PG_WINO4B=2: wherever F(4x4) runs (A/B runs).
_WINO4B = os.environ.get('PG_WINO4B', '1')
F4_FORM = 2 if _WINO4B == '0' else 3      # the form for narrow images (tests import this)
# Round 6: 4 = the one-workgroup kernel with its transform-domain GEMM on the bf16 pipe (six products of exact three-term splits, fp32 accumulation; csrc/conv2d_wino4.h "X3"):
# same results class (error against float64 at or below the fp32 form's, tools/wino4x3_probe.py), 1.03-1.26x faster per launch.  PG_WINO4_X3=0: the fp32-MFMA form (A/B runs).
F4_WIDE = 2 if os.environ.get('PG_WINO4_X3', '1') == '0' else 4      # the form for images at least 64 pixels wide (tests import this)


def f4_form(hw):
    if _WINO4B == '2':
        return 3
    return F4_FORM if int(hw[1]) < 64 else F4_WIDE


def use_winograd(kh, kw, stride, cout, cin=None, x2=None, pad=None, hw=None, xf=False, ep=None):
    """Launch policy for 3x3 stride-1 convolutions.  Returns 0 (direct implicit GEMM), 1 (Winograd F(2x2,3x3), csrc/conv2d_wino.h) or
    2 (Winograd F(4x4,3x3), csrc/conv2d_wino4.h) -- truthy = some Winograd kernel, and the value is what `pack_weight(winograd=...)` /
    `conv2d_forward(winograd=...)` take.  F(4x4) needs the image size (`hw`): its 8 x 64-pixel tiles and 16-channel chunks pay on layers
    with Cin >= 64, Cout a multiple of 64 and images of at least 32 x 32 (the two-workgroup form serves those narrower than 64 pixels: `f4_form`) whose width -- and output width -- is a multiple of 4; `xf` (an input pre-activation
    stage) stays on F(2x2).  PG_CONV_ALGO=direct|winograd|winograd2|winograd4 overrides (A/B measurements): 'winograd2' = never F(4x4),
    'winograd4' = F(4x4) wherever the kernel accepts the launch.  `ep` = the fused epilogue's keyword arguments when the caller has them: the
    F(4x4) tail evaluates the activation as max(v * gain, v * gain * slope), exact for gain > 0 and 0 <= alpha <= 1 only (the kernel declines
    anything else with PG_ERR_UNSUPPORTED); such launches stay on F(2x2), whose tail uses the select form."""
    if (int(kh), int(kw), int(stride)) != (3, 3, 1) or x2 is not None:
        return 0
    if pad is not None and not 0 <= int(pad[1]) <= 4:         # the kernels' LDS halo row starts 4 columns left of the tile
        return 0
    mode = os.environ.get('PG_CONV_ALGO', 'auto')
    if mode == 'direct':
        return 0
    f4_possible = hw is not None and not xf and int(hw[1]) % 4 == 0 and (pad is None or (int(hw[1]) + 2 * int(pad[1]) - 2) % 4 == 0)
    if ep:
        gain = ep.get('gain', 1.0)
        alpha = ep.get('alpha', 0.2)
        if (gain is not None and not float(gain) > 0) or (ep.get('act', 'linear') == 'lrelu' and alpha is not None and not 0 <= float(alpha) <= 1):
            f4_possible = False
    if mode == 'winograd4' and f4_possible:
        return f4_form(hw)
    if mode in ('winograd', 'winograd2', 'winograd4'):
        return 1
    if not (int(cout) > 32 and (cin is None or int(cin) >= 16)):
        return 0
    if f4_possible and cin is not None and int(cin) >= 64 and int(cin) % 16 == 0 and int(cout) % 64 == 0 and int(hw[0]) >= 32 and int(hw[1]) >= 32:
        return f4_form(hw)
    # (16 x 16 images: the two-workgroup F(4x4) form beats F(2x2) there too -- 112 vs 177 us at N = 8, tools/wino_small_probe.py -- but F(4x4)'s ~15x larger
    # rounding error in one of the FIRST layers of the style branch is amplified by everything behind it: the generator-gradient parity test went from 9e-4
    # to 2.7e-3 worst signature mismatch (bar 2e-3).  65 us per step is not worth that margin: F(2x2) keeps the 8 x 8 and 16 x 16 layers.)
    return 1


def supported(kh, kw, stride):
    return (int(kh), int(kw), int(stride)) in SUPPORTED


def _f32c(t, name):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_cuda:
        raise nat.NativeOpError(f'conv2d_mfma: {name} must be a float32 GPU tensor')
    return t.contiguous()


def pack_weight(w, scale=1.0, flip=False, transpose_oi=False, winograd=False):
    """OIHW (or IOHW when `transpose_oi`) float32 weights -> the kernel's [CinP][taps][CoutP] layout, or, with
    `winograd` = 1 / True, the pre-transformed [16][CinP][CoutP64] layout of the F(2x2,3x3) kernel, with `winograd` = 2 the
    36 * CinP * CoutP64 operand stream of the F(4x4,3x3) kernel (3x3 weights only)."""
    lib = _init().lib
    w = _f32c(w.detach(), 'weight')
    if transpose_oi:
        cin, cout, kh, kw = w.shape
    else:
        cout, cin, kh, kw = w.shape
    if winograd:
        if (kh, kw) != (3, 3):
            raise nat.NativeOpError('conv2d_mfma: the Winograd layout is for 3x3 weights')
        size, pack = ((lib.pg_conv2d_winograd4x3_packed_size, lib.pg_conv2d_winograd4x3_pack_weight) if int(winograd) == 4 else
                      (lib.pg_conv2d_winograd4_packed_size, lib.pg_conv2d_winograd4b_pack_weight) if int(winograd) == 3 else
                      (lib.pg_conv2d_winograd4_packed_size, lib.pg_conv2d_winograd4_pack_weight) if int(winograd) == 2 else
                      (lib.pg_conv2d_winograd_packed_size, lib.pg_conv2d_winograd_pack_weight))
        packed = torch.empty([size(cout, cin)], dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            st = pack(nat.ptr(w), nat.ptr(packed), cout, cin, float(scale), int(bool(flip)), int(bool(transpose_oi)), nat.stream_of(w))
        nat.check(st, 'pg_conv2d_winograd_pack_weight')
        return packed
    packed = torch.empty([lib.pg_conv2d_packed_size(cout, cin, kh, kw)], dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        st = lib.pg_conv2d_pack_weight(nat.ptr(w), nat.ptr(packed), cout, cin, kh, kw, float(scale), int(bool(flip)), int(bool(transpose_oi)), nat.stream_of(w))
    nat.check(st, 'pg_conv2d_pack_weight')
    return packed


def conv2d_forward(x, packed, cout, kh, kw, stride=1, pad=(0, 0), out_hw=None, y=None, out_step=(1, 1), out_off=(0, 0),
                   in_scale=None, in_bias=None, in_act='linear', in_alpha=0.0, in_gain=1.0, in_clamp=None,
                   out_scale=None, noise=None, noise_gain=1.0, bias=None, act='linear', alpha=0.0, gain=1.0, clamp=None,
                   residual=None, spade=None, x2=None, winograd=False, stats_eps=None):
    """One launch of the MFMA convolution (`winograd`: of its F(2x2,3x3) variant; `packed` must then come from
    `pack_weight(..., winograd=True)`; 3x3 stride 1, no spade / x2).  `x` [N,Cin,H,W] float32 contiguous; `packed` from
    `pack_weight`.  Writes y[n, co, oy*step+off, ox*step+off] for oy < out_hw[0], ox < out_hw[1]
    (allocating a dense [N,Cout,OH,OW] `y` when none is given) and returns `y`.
    `spade=(x_norm, mean, rstd)` selects the SPADE combine epilogue: `packed` holds interleaved gamma/beta rows
    (`pack_spade_gamma_beta`), `cout` = 2*C, and the result is the [N, C, OH, OW] tensor
    (x_norm - mean) * rstd * (1 + gamma) + beta.
    `stats_eps` (F(4x4) one-workgroup launches with the plain tail only: winograd=2, no in_scale / noise / residual / spade; anything else raises
    NativeNotCovered): also gather the instance-norm statistics of the OUTPUT where it is produced -- returns (y, (mean, rstd)) with
    rstd = 1 / sqrt(var + stats_eps), what `instance_norm_stats(y, stats_eps)` would compute in a second pass over y."""
    lib = _init().lib
    x = _f32c(x, 'x')
    n, cin, h, w = x.shape
    cin_split = 0
    if x2 is not None:          # conv(torch.cat([x, x2], 1)) without materialising the concatenation
        x2 = _f32c(x2, 'x2')
        if x2.shape[0] != n or x2.shape[2:] != x.shape[2:] or cin % 16 != 0:
            raise nat.NativeOpError('conv2d_mfma: x2 must match x in N, H, W and x must have a multiple of 16 channels')
        cin_split, cin = cin, cin + x2.shape[1]
    pad_y, pad_x = pad
    if out_hw is None:
        out_hw = ((h + 2 * pad_y - kh) // stride + 1, (w + 2 * pad_x - kw) // stride + 1)
    oh, ow = out_hw
    ychan = cout // 2 if spade is not None else cout
    if y is None:
        assert tuple(out_step) == (1, 1) and tuple(out_off) == (0, 0)
        y = torch.empty([n, ychan, oh, ow], dtype=torch.float32, device=x.device)
    else:
        assert y.dtype == torch.float32 and y.device == x.device and y.shape[0] == n and y.shape[1] == ychan
    fz = Fusion()
    keep = []

    def dev(t, name, numel=None):
        t = _f32c(t, name)
        if t is None:
            return None
        if numel is not None and t.numel() != numel:
            raise nat.NativeOpError(f'conv2d_mfma: {name} has {t.numel()} elements, expected {numel}')
        keep.append(t)
        return t.data_ptr()

    fz.in_scale = dev(in_scale, 'in_scale', n * cin)
    fz.in_bias = dev(in_bias, 'in_bias', cin)
    fz.in_act, fz.in_alpha, fz.in_gain = ACT_INDEX[in_act], float(in_alpha), float(in_gain)
    fz.in_clamp = -1.0 if in_clamp is None else float(in_clamp)
    fz.out_scale = dev(out_scale, 'out_scale', n * cout)
    if noise is not None:
        noise = _f32c(noise, 'noise')
        if noise.numel() == oh * ow:
            fz.noise_batch_stride = 0
        elif noise.numel() == n * oh * ow:
            fz.noise_batch_stride = oh * ow
        else:
            raise nat.NativeOpError('conv2d_mfma: noise must have OH*OW or N*OH*OW elements')
        keep.append(noise)
        fz.noise = noise.data_ptr()
    fz.noise_gain = float(noise_gain)
    fz.bias = dev(bias, 'bias', cout)
    fz.act, fz.alpha, fz.gain = ACT_INDEX[act], float(alpha), float(gain)
    fz.clamp = -1.0 if clamp is None else float(clamp)
    if residual is not None:
        if residual.dtype != torch.float32 or residual.shape != y.shape or residual.stride() != y.stride():
            raise nat.NativeOpError('conv2d_mfma: residual must match y in dtype, shape and strides')
        keep.append(residual)
        fz.residual = residual.data_ptr()
    if spade is not None:
        sx, smean, srstd = spade
        if sx.dtype != torch.float32 or sx.shape != y.shape or sx.stride() != y.stride() or smean.numel() != n * ychan or srstd.numel() != n * ychan:
            raise nat.NativeOpError('conv2d_mfma: spade tensors must match the [N, C, OH, OW] output')
        keep += [sx, smean, srstd]
        fz.spade_x, fz.spade_mean, fz.spade_rstd = sx.data_ptr(), smean.data_ptr(), srstd.data_ptr()
    if x2 is not None:
        keep.append(x2)
        fz.x2, fz.cin_split = x2.data_ptr(), cin_split
    stats_part = None
    if stats_eps is not None:
        if int(winograd) not in (2, 4) or spade is not None or in_scale is not None or noise is not None or residual is not None or x2 is not None or in_act != 'linear':
            raise nat.NativeNotCovered('conv2d_mfma: output statistics are gathered by the F(4x4) kernel\'s plain tail only')
        stats_T = lib.pg_conv2d_winograd4_stats_tiles(int(oh), int(ow))
        stats_part = torch.empty([n * cout * stats_T * 2], dtype=torch.float32, device=x.device)
        fz.stats_partial = stats_part.data_ptr()
    with torch.cuda.device(x.device):
        if _timeline is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        if winograd:
            if (kh, kw, int(stride)) != (3, 3, 1) or tuple(out_step) != (1, 1) or tuple(out_off) != (0, 0):
                raise nat.NativeOpError('conv2d_mfma: winograd=True needs a 3x3 stride-1 dense-output launch')
            fwd = {1: lib.pg_conv2d_winograd_forward, 2: lib.pg_conv2d_winograd4_forward, 3: lib.pg_conv2d_winograd4b_forward, 4: lib.pg_conv2d_winograd4x3_forward}[int(winograd)]
            st = fwd(nat.ptr(x), nat.ptr(packed), nat.ptr(y), n, cin, h, w, cout, int(pad_y), int(pad_x),
                     int(oh), int(ow), nat.i64arr(y.stride()), ctypes.byref(fz), nat.stream_of(x))
        else:
            ksplit = 1
            if spade is None and x2 is None and os.environ.get('PG_CONV_SPLITK', '1') != '0':
                ksplit = lib.pg_conv2d_splitk_plan(n, cin, int(oh), int(ow), cout, kh, kw, int(stride))
            if ksplit > 1:      # few output tiles (low-resolution layers): share each tile's K loop among `ksplit` workgroups
                ws = torch.empty([ksplit * n * cout * int(oh) * int(ow)], dtype=torch.float32, device=x.device)
                st = lib.pg_conv2d_forward_splitk(nat.ptr(x), nat.ptr(packed), nat.ptr(y), n, cin, h, w, cout, kh, kw, int(stride), int(pad_y), int(pad_x),
                                                  int(oh), int(ow), nat.i64arr(y.stride()), int(out_step[0]), int(out_step[1]), int(out_off[0]), int(out_off[1]),
                                                  ctypes.byref(fz), nat.ptr(ws), ksplit, nat.stream_of(x))
            else:
                st = lib.pg_conv2d_forward(nat.ptr(x), nat.ptr(packed), nat.ptr(y), n, cin, h, w, cout, kh, kw, int(stride), int(pad_y), int(pad_x),
                                           int(oh), int(ow), nat.i64arr(y.stride()), int(out_step[0]), int(out_step[1]), int(out_off[0]), int(out_off[1]),
                                           ctypes.byref(fz), nat.stream_of(x))
        if _timeline is not None:
            ev1.record()
            _timeline.append(((kh, kw, int(stride), ('winograd4x3' if int(winograd) == 4 else 'winograd4' if int(winograd) >= 2 else 'winograd') if winograd else 'direct', f'N{n} {cin}->{cout} {h}x{w}' + (' spade' if spade is not None else '') + (' xf' if in_act != 'linear' else '') + (' mod' if in_scale is not None else '') + (' res' if residual is not None else '')), 2.0 * n * cout * oh * ow * cin * kh * kw, ev0, ev1,
                              4 * (x.numel() + (x2.numel() if x2 is not None else 0) + n * ychan * oh * ow)))
    nat.check(st, 'pg_conv2d_forward')
    if stats_part is not None:
        mean = torch.empty([n * cout], dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        with torch.cuda.device(x.device):
            st = lib.pg_instance_norm_finish(nat.ptr(stats_part), nat.ptr(mean), nat.ptr(rstd), n * cout, stats_T, int(oh), int(ow), float(stats_eps), nat.stream_of(x))
        nat.check(st, 'pg_instance_norm_finish')
        return y, (mean, rstd)
    return y


def pack_spade_gamma_beta(w_gamma, w_beta, scale_gamma=1.0, scale_beta=1.0, winograd=False):
    """Packed weights of the fused gamma/beta convolution of a Spade_Norm_Block: rows [64j, 64j+32) are gamma channels
    [32j, 32j+32), rows [64j+32, 64j+64) the beta rows of the same channels, so that the two M-tiles of one 64-row
    workgroup tile hold gamma and beta of the same (channel, pixel) in the same lane (the Winograd kernel exchanges
    8 rows of each M-tile per round, so the same order serves it)."""
    c = int(w_gamma.shape[0])
    grp = 32
    assert w_gamma.shape == w_beta.shape and c % 32 == 0
    if int(winograd) == 3:      # the two-workgroup F(4x4) kernel finishes rows (2c, 2c + 1) in one lane: gamma / beta of a channel as ADJACENT rows
        gw, bw = w_gamma.detach() * scale_gamma, w_beta.detach() * scale_beta
        return pack_weight(torch.stack([gw, bw], dim=1).reshape(2 * c, *w_gamma.shape[1:]).contiguous(), winograd=3)
    g = (w_gamma.detach() * scale_gamma).reshape(c // grp, grp, *w_gamma.shape[1:])
    b = (w_beta.detach() * scale_beta).reshape(c // grp, grp, *w_beta.shape[1:])
    return pack_weight(torch.cat([g, b], dim=1).reshape(2 * c, *w_gamma.shape[1:]).contiguous(), winograd=winograd)


def transposed_phases(kh, kw, stride, pad_y, pad_x, in_hw, out_hw):
    """Decompose a stride-`s` transposed convolution into gather-form sub-convolutions, one per
    output phase.  For output row o, phase a = (o + pad) mod s, m = (o + pad - a) / s:
        out[o] = sum_j x[m - j] * w[a + s*j],  j < J_a = ceil((K - a) / s)
    which as a correlation over taps t = J_a-1-j reads x[m - (J_a-1) + t] * w[a + s*(J_a-1-t)].
    Returns a list of dicts: tap index lists (ky, kx), conv pads, output offsets and counts."""
    s = stride

    def axis(k, pad, n_out):
        res = []
        for a in range(s):
            j = -(-(k - a) // s)            # ceil
            if j <= 0:
                return None                 # a phase with no tap (K < s): outputs there are zero
            o_first = (a - pad) % s
            if o_first >= n_out:
                res.append(None)
                continue
            m_first = (o_first + pad - a) // s
            count = (n_out - 1 - o_first) // s + 1
            taps = [a + s * (j - 1 - t) for t in range(j)]
            res.append(dict(taps=taps, conv_pad=(j - 1) - m_first, off=o_first, count=count))
        return res

    ys, xs = axis(kh, pad_y, out_hw[0]), axis(kw, pad_x, out_hw[1])
    if ys is None or xs is None:
        return None
    return [dict(ky=py['taps'], kx=px['taps'], pad=(py['conv_pad'], px['conv_pad']), off=(py['off'], px['off']), out_hw=(py['count'], px['count']))
            for py in ys if py is not None for px in xs if px is not None]


_tap_indices = {}


def _tap_index(taps, device):
    """Device index tensor of a tap list, made once per (taps, device): indexing with a Python list builds the index on the host and copies
    it over at every call -- a host-to-device copy, which a hipGraph capture refuses (training.training_step)."""
    key = (tuple(int(t) for t in taps), str(device))
    if key not in _tap_indices:
        _tap_indices[key] = torch.tensor(key[0], dtype=torch.int64, device=device)
    return _tap_indices[key]


def _desc_taps_slice(taps):
    """The ascending slice whose reversal is the tap list `taps` (equal descending steps, or one tap); None otherwise."""
    taps = [int(t) for t in taps]
    if len(taps) == 1:
        return slice(taps[0], taps[0] + 1)
    step = taps[0] - taps[1]
    if step > 0 and all(taps[i] - taps[i + 1] == step for i in range(len(taps) - 1)):
        return slice(taps[-1], taps[0] + 1, step)
    return None


def pack_transposed(w_iohw, stride, pad, in_hw, out_hw, scale=1.0):
    """Packed per-phase weights of conv_transpose2d(x, w_iohw, stride, padding=pad) -> list of (phase, packed)."""
    kh, kw = int(w_iohw.shape[2]), int(w_iohw.shape[3])
    phases = transposed_phases(kh, kw, stride, pad[0], pad[1], in_hw, out_hw)
    if phases is None:
        return None
    out = []
    for ph in phases:
        if not supported(len(ph['ky']), len(ph['kx']), 1):
            return None
        sy, sx = _desc_taps_slice(ph['ky']), _desc_taps_slice(ph['kx'])
        if sy is not None and sx is not None:
            # the tap lists of a phase descend in equal steps ([2, 0], [1]: transposed_phases): a strided view read back to front -- one copy + the pack's
            # own flip instead of two index_select launches (440 of them per training iteration, 2.2 ms)
            out.append((ph, pack_weight(w_iohw.detach()[:, :, sy, sx], scale=scale, transpose_oi=True, flip=True)))
            continue
        sel = w_iohw.detach().index_select(2, _tap_index(ph['ky'], w_iohw.device)).index_select(3, _tap_index(ph['kx'], w_iohw.device))
        out.append((ph, pack_weight(sel, scale=scale, transpose_oi=True)))
    return out


def conv_transpose2d_forward(x, packed_phases, cout, out_hw, **fusion):
    """Run the phases of `pack_transposed` into one dense [N, Cout, OH, OW] output."""
    n = x.shape[0]
    y = torch.empty([n, cout, out_hw[0], out_hw[1]], dtype=torch.float32, device=x.device)
    stride = fusion.pop('stride', 2)
    for ph, packed in packed_phases:
        conv2d_forward(x, packed, cout, len(ph['ky']), len(ph['kx']), stride=1, pad=ph['pad'], out_hw=ph['out_hw'], y=y,
                       out_step=(stride, stride), out_off=ph['off'], **fusion)
    return y


def weight_gradient_supported(n, cin, oh, ow, cout, kh, kw, stride=1):
    """True when `weight_gradient` covers this geometry (pg_conv2d_wgrad_plan > 0)."""
    return _init().lib.pg_conv2d_wgrad_plan(int(n), int(cin), int(oh), int(ow), int(cout), int(kh), int(kw), int(stride)) > 0


def _bf16x3_wanted(n, cin, cout, h, w, kh, kw, stride):
    """Which float32 weight gradients run on the bf16 matrix pipe by three-term operand splitting (round 5; `_weight_gradient_bf16x3`).
    PG_WGRAD_BF16X3: auto (default) = the 3x3 layers (stride 1 | 2) with at least PG_WGRAD_BF16X3_MIN_C (64) channels on both sides -- where the probe and the
    training step measured it faster (tools/wgrad_bf16x3_probe.py: 0.52 ... 0.82 of the fp32 kernel's time at N = 4; config 4 258.6 -> 247-248 ms per
    iteration; the 1x1 layers and narrower 3x3 layers lose: the splitting passes cost 10 bytes per operand element); 0 = the fp32 MFMA kernel everywhere;
    1 = wherever the 16-bit kernel covers the geometry (3x3 stride 1 | 2, 1x1; channel counts multiples of 8)."""
    mode = os.environ.get('PG_WGRAD_BF16X3', 'auto')
    if mode == '0' or kh != kw or (kh, stride) not in ((3, 1), (3, 2), (1, 1)) or cin % 8 or cout % 8:
        return False
    if mode == 'auto':
        if kh == 1:       # (1x1 layers: off unless PG_WGRAD_BF16X3_K1_MIN_C names a width)
            k1 = int(os.environ.get('PG_WGRAD_BF16X3_K1_MIN_C', '0'))
            return k1 > 0 and min(cin, cout) >= k1
        return min(cin, cout) >= int(os.environ.get('PG_WGRAD_BF16X3_MIN_C', '64'))
    return True


def _weight_gradient_bf16x3(lib, x, dy, weight_shape, pad, stride, out_hw):
    """float32 dw on the bf16 matrix pipe: x = x1 + x2 + x3, dy = d1 + d2 + d3 (bf16 terms, exact), the six largest products in one launch of the 16-bit
    weight-gradient kernel over 6 N plane-mapped images, float32 accumulation (csrc/conv2d_wgrad.hip, pg_split3_bf16_cl + pg_conv2d16_wgrad_x3)."""
    cout, cin, kh, kw = weight_shape
    n, _, h, w = x.shape
    oh, ow = out_hw
    splits = lib.pg_conv2d16_wgrad_plan(6 * n, cin, oh, ow, cout, kh, kw, stride)
    if splits <= 0:
        return None
    x, dy = x.contiguous(), dy.contiguous()
    x3 = torch.empty([3, n, h, w, cin], dtype=torch.bfloat16, device=x.device)
    d3 = torch.empty([3, n, oh, ow, cout], dtype=torch.bfloat16, device=x.device)
    dw = torch.empty([cout, cin, kh, kw], dtype=torch.float32, device=x.device)
    ws = torch.empty([splits * kh * kw * cout * cin], dtype=torch.float32, device=x.device)
    tl = _wgrad_timeline
    with torch.cuda.device(x.device):
        if tl is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        st = lib.pg_split3_bf16_cl(nat.ptr(x), nat.ptr(x3), n, cin, h * w, nat.stream_of(x))
        if st == 0:
            st = lib.pg_split3_bf16_cl(nat.ptr(dy), nat.ptr(d3), n, cout, oh * ow, nat.stream_of(x))
        if st == 0:
            st = lib.pg_conv2d16_wgrad_x3(nat.ptr(x3), nat.ptr(d3), nat.ptr(dw), nat.ptr(ws), n, cin, h, w, cout, kh, kw, stride, int(pad[0]), int(pad[1]), oh, ow, splits,
                                          nat.stream_of(x))
        if tl is not None:
            ev1.record()
            tl.append(((kh, kw, stride, 'wgrad_bf16x3', f'N{n} {cin}->{cout} {h}x{w}'), 2.0 * n * cout * oh * ow * cin * kh * kw, ev0, ev1, 4 * (x.numel() + dy.numel() + dw.numel())))
    if st == -2:
        return None
    nat.check(st, 'pg_conv2d16_wgrad_x3')
    return dw


def weight_gradient(x, dy, weight_shape, pad, stride=1):
    """d(loss)/d(weight) of y = conv2d(x, w, stride, padding=pad) for 3x3 (stride 1 | 2) / 1x1 (stride 1) kernels: a GEMM over pixels
    on the fp32 MFMA (csrc/conv2d_wgrad.hip).  Returns None when the geometry is not covered (callers then ask aten)."""
    lib = _init().lib
    cout, cin, kh, kw = (int(v) for v in weight_shape)
    n, _, h, w = x.shape
    oh, ow = int(dy.shape[2]), int(dy.shape[3])
    stride = int(stride)
    if x.dtype != torch.float32 or dy.dtype != torch.float32 or min(pad) < 0 or (oh, ow) != ((h + 2 * pad[0] - kh) // stride + 1, (w + 2 * pad[1] - kw) // stride + 1):
        return None
    if _bf16x3_wanted(n, cin, cout, h, w, kh, kw, stride):
        dw = _weight_gradient_bf16x3(lib, x, dy, (cout, cin, kh, kw), pad, stride, (oh, ow))
        if dw is not None:
            return dw
    splits = lib.pg_conv2d_wgrad_plan(n, cin, oh, ow, cout, kh, kw, stride)
    if splits <= 0:
        return None
    x, dy = x.contiguous(), dy.contiguous()
    dw = torch.empty([cout, cin, kh, kw], dtype=torch.float32, device=x.device)
    ws = torch.empty([splits * kh * kw * cout * cin], dtype=torch.float32, device=x.device)
    tl = _wgrad_timeline
    with torch.cuda.device(x.device):
        if tl is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        st = lib.pg_conv2d_wgrad(nat.ptr(x), nat.ptr(dy), nat.ptr(dw), nat.ptr(ws), n, cin, h, w, cout, kh, kw, stride, int(pad[0]), int(pad[1]), oh, ow, splits,
                                 nat.stream_of(x))
        if tl is not None:
            ev1.record()
            tl.append(((kh, kw, stride, 'wgrad', f'N{n} {cin}->{cout} {h}x{w}'), 2.0 * n * cout * oh * ow * cin * kh * kw, ev0, ev1, 4 * (x.numel() + dy.numel() + dw.numel())))
    nat.check(st, 'pg_conv2d_wgrad')
    return dw


def pack_up2(weight, flip=False, x3=True):
    """Packed weights of `conv_up2_forward` for the OIHW 3x3 kernel w = `weight` (flipped spatially iff `flip`): the plain 3x3 pack."""
    w = weight.detach().float()
    if flip:
        w = w.flip([2, 3])
    packs = dict(main=pack_weight(w.contiguous()))
    cout, cin = int(w.shape[0]), int(w.shape[1])
    if x3 and UP2_X3 and cin % 16 == 0 and cin >= 32:     # round 6: the main tiles on the bf16 pipe (three-term operand splits, csrc/conv2d_up2x3.h): the split planes of the same pack, once per weight version
        lib = _init().lib
        x3 = torch.empty([lib.pg_conv2d_up2x3_packed_size(cout, cin)], dtype=torch.uint8, device=w.device)
        with torch.cuda.device(w.device):
            nat.check(lib.pg_conv2d_up2x3_pack_weight(nat.ptr(packs['main']), nat.ptr(x3), cout, cin, nat.stream_of(w)), 'pg_conv2d_up2x3_pack_weight')
        packs['x3'] = x3
    return packs


def pack_s2x3_planes(packed32, cout, cin):
    """The bf16 plane slabs of `conv_up2_forward`'s bf16-pipe form from the float32 pack of a 3x3 kernel (`pack_weight`): what `pack_up2` stores under 'x3'."""
    lib = _init().lib
    x3 = torch.empty([lib.pg_conv2d_up2x3_packed_size(cout, cin)], dtype=torch.uint8, device=packed32.device)
    with torch.cuda.device(packed32.device):
        nat.check(lib.pg_conv2d_up2x3_pack_weight(nat.ptr(packed32), nat.ptr(x3), cout, cin, nat.stream_of(packed32)), 'pg_conv2d_up2x3_pack_weight')
    return x3


# PG_UP2_X3=0: the fp32-MFMA kernel everywhere (A/B runs).  Default: layers whose input is wider than 16 pixels (a multiple of 4) with Cin % 16 == 0 multiply on the bf16 pipe.
UP2_X3 = os.environ.get('PG_UP2_X3', '1') != '0'


def conv_up2_forward(x, packs, cout, in_scale=None, out_scale=None, x3=None, edge_column=True):
    """conv_transpose2d(x * in_scale, w, stride=2, padding=0) * out_scale for a 3x3 kernel: all four output parities and the last
    output column in one launch (csrc/conv2d_up2.h).  `packs` = pack_up2(w).  Returns a [N, Cout, 2H+1, 2W+1] view whose rows are
    padded to a multiple of 4 floats (aligned pair stores here, aligned rows for the FIR pass that follows).
    `edge_column` = False (tests): the bf16-pipe form's edge kernel gathers input column W - 1 from x instead of reading the dense copy the main kernel leaves."""
    lib = _init().lib
    x = _f32c(x, 'x')
    n, cin, h, w = x.shape
    oh, ow = 2 * h + 1, 2 * w + 1
    pitch = (ow + 3) // 4 * 4
    buf = torch.empty([n, cout, oh, pitch], dtype=torch.float32, device=x.device)
    y = buf[:, :, :, :ow]
    s_in = _f32c(in_scale, 'in_scale') if in_scale is not None else None
    s_out = _f32c(out_scale, 'out_scale') if out_scale is not None else None
    tl = _timeline
    with torch.cuda.device(x.device):
        if tl is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        ksplit = lib.pg_conv2d_up2_splitk_plan(n, cin, h, w, cout)        # the 8^2 / 16^2 layers: shares of the input channels side by side, one summing pass
        # `x3` = False: the fp32-MFMA kernel whatever the pack holds (the training route: see training/networks.py _ModConvUp2Train)
        x3 = packs.get('x3') if (x3 is not False and UP2_X3 and ksplit == 1 and w > 16 and w % 4 == 0 and cin % 16 == 0 and cin >= 32 and x.data_ptr() % 16 == 0) else None
        if x3 is not None:
            xcol = torch.empty([n * cin * h], dtype=torch.float32, device=x.device) if edge_column else None      # scratch: input column W - 1, dense, for the edge kernel
            st = lib.pg_conv2d_up2x3_forward(nat.ptr(x), nat.ptr(packs['main']), nat.ptr(x3), nat.ptr(y), n, cin, h, w, cout, nat.i64arr(y.stride()), nat.ptr(s_in), nat.ptr(s_out),
                                             nat.ptr(xcol), nat.stream_of(x))
        elif ksplit > 1:
            ws = torch.empty([ksplit * buf.numel()], dtype=torch.float32, device=x.device)
            st = lib.pg_conv2d_up2_forward_splitk(nat.ptr(x), nat.ptr(packs['main']), nat.ptr(y), n, cin, h, w, cout, nat.i64arr(y.stride()), nat.ptr(s_in), nat.ptr(s_out),
                                                  nat.ptr(ws), ksplit, nat.stream_of(x))
        else:
            st = lib.pg_conv2d_up2_forward(nat.ptr(x), nat.ptr(packs['main']), nat.ptr(y), n, cin, h, w, cout, nat.i64arr(y.stride()), nat.ptr(s_in), nat.ptr(s_out),
                                           nat.stream_of(x))
        if tl is not None:
            ev1.record()
            tl.append(((3, 3, 2, 'direct', f'N{n} {cin}->{cout} {h}x{w} up2' + (' x3' if x3 is not None else '') + (' mod' if in_scale is not None else '')), 2.0 * n * cout * cin * 9 * h * w, ev0, ev1,
                       4 * (x.numel() + n * cout * oh * ow)))
    nat.check(st, 'pg_conv2d_up2_forward')
    return y


STEM7_X3 = os.environ.get('PG_STEM7_X3', '1') != '0'


def pack_stem7(weight, scale=1.0, flip=False):
    """Packed weights of `conv_stem7_forward` for the OIHW [Cout, 3, 7, 7] float32 kernel `weight` * `scale` (flipped spatially iff `flip`): three bf16 planes
    per weight, in the MFMA A-fragment order of csrc/conv2d_stem7x3.h."""
    lib = _init().lib
    w = _f32c(weight.detach(), 'weight')
    cout, cin, kh, kw = w.shape
    if (cin, kh, kw) != (3, 7, 7):
        raise nat.NativeOpError('conv2d_mfma.pack_stem7: a [Cout, 3, 7, 7] kernel')
    packed = torch.empty([lib.pg_conv2d_stem7x3_packed_size(cout)], dtype=torch.uint8, device=w.device)
    with torch.cuda.device(w.device):
        nat.check(lib.pg_conv2d_stem7x3_pack_weight(nat.ptr(w), nat.ptr(packed), cout, float(scale), int(bool(flip)), nat.stream_of(w)), 'pg_conv2d_stem7x3_pack_weight')
    return packed


def conv_stem7_forward(x, packed, cout, bias=None, act='linear', alpha=0.0, gain=1.0, clamp=None):
    """clamp(act(conv2d(x [N, 3, H, W], w, padding=3) + bias) * gain) for a 7x7 kernel on the bf16 matrix pipe (round 6, csrc/conv2d_stem7x3.h: float32
    operands as exact sums of three bf16 values, six plane products per float32 product, float32 accumulation -- float32-class).  `packed` = pack_stem7(w)."""
    lib = _init().lib
    x = _f32c(x, 'x')
    n, cin, h, w = x.shape
    if cin != 3 or act not in FUSED_ACTS:
        raise nat.NativeNotCovered('conv2d_mfma.conv_stem7_forward: three input channels, linear / relu / lrelu')
    y = torch.empty([n, cout, h, w], dtype=torch.float32, device=x.device)
    fz = Fusion()
    b = _f32c(bias, 'bias') if bias is not None else None
    fz.bias = b.data_ptr() if b is not None else None
    fz.act, fz.alpha, fz.gain = ACT_INDEX[act], float(alpha), float(gain)
    fz.clamp = -1.0 if clamp is None else float(clamp)
    fz.in_clamp, fz.in_gain = -1.0, 1.0
    tl = _timeline
    with torch.cuda.device(x.device):
        if tl is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        st = lib.pg_conv2d_stem7x3_forward(nat.ptr(x), nat.ptr(packed), nat.ptr(y), n, h, w, cout, nat.i64arr(y.stride()), ctypes.byref(fz), nat.stream_of(x))
        if tl is not None:
            ev1.record()
            tl.append(((7, 7, 1, 'direct', f'N{n} 3->{cout} {h}x{w} x3'), 2.0 * n * cout * h * w * 3 * 49, ev0, ev1, 4 * (x.numel() + y.numel())))
    if st == -2:
        raise nat.NativeNotCovered('pg_conv2d_stem7x3_forward declined the launch')
    nat.check(st, 'pg_conv2d_stem7x3_forward')
    return y


def conv3x3_cin1_ok(x, weight, padding=1):
    return (x.dtype == torch.float32 and x.is_cuda and x.is_contiguous() and int(x.shape[1]) == 1 and tuple(weight.shape[1:]) == (1, 3, 3)
            and int(padding) == 1 and x.shape[3] % 4 == 0)


def conv3x3_cin1(x, weight, scale=1.0, act='linear'):
    """act(conv2d(x, weight * scale, padding=1)) for a one-channel x as a 9-tap stencil per output channel (act: linear | relu, gain 1)."""
    lib = _init().lib
    n, _, h, w = x.shape
    cout = int(weight.shape[0])
    wf = _f32c(weight.detach(), 'weight')
    y = torch.empty([n, cout, h, w], dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.pg_conv3x3_cin1(nat.ptr(x), nat.ptr(wf), nat.ptr(y), n, h, w, cout, float(scale), ACT_INDEX[act], nat.stream_of(x))
    nat.check(rc, 'pg_conv3x3_cin1')
    return y


def conv1x1_small_ok(x, weight, skip=None):
    """Shapes the streaming 1x1 head kernel takes: float32 dense NCHW, Cout <= 8, H*W a multiple of 4."""
    return (x.dtype == torch.float32 and x.is_cuda and x.is_contiguous() and tuple(weight.shape[2:]) == (1, 1) and int(weight.shape[0]) <= 8
            and (x.shape[2] * x.shape[3]) % 4 == 0 and int(x.shape[1]) <= 2048 and (skip is None or (skip.dtype == torch.float32 and skip.is_contiguous())))


def conv1x1_small(x, weight, styles=None, bias=None, skip=None, scale=1.0, clamp=None):
    """ToRGB-style head in one streaming pass: clamp(conv1x1(x * styles[:, :, None, None], weight * scale) + bias) + skip
    (networks.py:306-316 with modulated_conv2d(demodulate=False), networks.py:37-94)."""
    lib = _init().lib
    n, cin, h, w = x.shape
    cout = int(weight.shape[0])
    wf = _f32c(weight.detach().reshape(cout, cin), 'weight')
    st = _f32c(styles.detach(), 'styles') if styles is not None else None
    b = _f32c(bias.detach(), 'bias') if bias is not None else None
    y = torch.empty([n, cout, h, w], dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.pg_conv1x1_small(nat.ptr(x), nat.ptr(wf), nat.ptr(st), nat.ptr(b), nat.ptr(skip), nat.ptr(y), n, cin, h * w, cout,
                                  float(scale), float(clamp) if clamp is not None else -1.0, nat.stream_of(x))
    nat.check(rc, 'pg_conv1x1_small')
    return y


def modconv_dcoefs(weight, styles, scale=1.0):
    """rsqrt(sum_{i,k} (w[o,i,k] * scale * s[n,i])^2 + 1e-8) -> [N, Cout] (networks.py:64-68)."""
    lib = _init().lib
    weight, styles = _f32c(weight.detach(), 'weight'), _f32c(styles.detach(), 'styles')
    cout, cin, kh, kw = weight.shape
    n = styles.shape[0]
    assert styles.shape[1] == cin
    d = torch.empty([n, cout], dtype=torch.float32, device=weight.device)
    with torch.cuda.device(weight.device):
        st = lib.pg_modconv_dcoefs(nat.ptr(weight), nat.ptr(styles), nat.ptr(d), n, cout, cin, kh * kw, float(scale), nat.stream_of(weight))
    nat.check(st, 'pg_modconv_dcoefs')
    return d


def modconv_w2(weight, scale=1.0):
    """Tap energy scale^2 * sum_k w[o,i,k]^2 -> [Cout, Cin]: the weight-only half of the demodulation coefficients
    (networks.py:64-68), computed once per weight version by callers that cache it."""
    lib = _init().lib
    weight = _f32c(weight.detach(), 'weight')
    cout, cin, kh, kw = weight.shape
    w2 = torch.empty([cout, cin], dtype=torch.float32, device=weight.device)
    with torch.cuda.device(weight.device):
        st = lib.pg_modconv_w2(nat.ptr(weight), nat.ptr(w2), cout, cin, kh * kw, float(scale), nat.stream_of(weight))
    nat.check(st, 'pg_modconv_w2')
    return w2


PREP_MAX_JOBS = 32


class PrepJobs(ctypes.Structure):
    """pg_modconv_prep_jobs of include/pasta_gan_ops.h."""
    _fields_ = [('w2', ctypes.c_void_p * PREP_MAX_JOBS), ('styles', ctypes.c_void_p * PREP_MAX_JOBS), ('out', ctypes.c_void_p * PREP_MAX_JOBS),
                ('s_norm', ctypes.c_void_p * PREP_MAX_JOBS), ('s16', ctypes.c_void_p * PREP_MAX_JOBS), ('cout', ctypes.c_int * PREP_MAX_JOBS),
                ('cin', ctypes.c_int * PREP_MAX_JOBS), ('flags', ctypes.c_int * PREP_MAX_JOBS), ('njobs', ctypes.c_int), ('half_dtype', ctypes.c_int)]


_prep_registry = {}     # styles.data_ptr() -> (normalize, demodulate, half_dtype, (out, s_norm, s16)): results of modconv_prep_batched awaiting their layer


def modconv_prep_batched(jobs, half_dtype=None):
    """`modconv_prep` for a whole network in ONE launch.  jobs: list of (w2 | None, styles [N, Cin] float32 contiguous, cout, normalize, demodulate).
    The results are parked in a registry keyed by the styles' address; the layer's own `modconv_prep(...)` call with the same arguments picks
    its entry up (and removes it) instead of launching.  Call `modconv_prep_clear()` when the forward pass is over."""
    lib = _init().lib
    assert 0 < len(jobs) <= PREP_MAX_JOBS
    table = PrepJobs()
    table.njobs, table.half_dtype = len(jobs), (nat.PG_DTYPE[half_dtype] if half_dtype is not None else 0)
    n = int(jobs[0][1].shape[0])
    keep = []
    for j, (w2, styles, cout, normalize, demodulate) in enumerate(jobs):
        assert styles.dtype == torch.float32 and styles.is_contiguous() and styles.shape[0] == n
        cin = int(styles.shape[1])
        out = torch.empty([n, cout], dtype=torch.float32, device=styles.device)
        s_norm = torch.empty_like(styles) if normalize else None
        s16 = torch.empty([n, cin], dtype=half_dtype, device=styles.device) if (normalize and half_dtype is not None) else None
        if demodulate:
            w2 = _f32c(w2, 'w2')
            assert tuple(w2.shape) == (cout, cin)
        table.w2[j] = w2.data_ptr() if demodulate else None
        table.styles[j], table.out[j] = styles.data_ptr(), out.data_ptr()
        table.s_norm[j] = s_norm.data_ptr() if s_norm is not None else None
        table.s16[j] = s16.data_ptr() if s16 is not None else None
        table.cout[j], table.cin[j], table.flags[j] = int(cout), cin, int(bool(normalize)) | (int(bool(demodulate)) << 1)
        keep.append((styles, w2))
        _prep_registry[styles.data_ptr()] = (bool(normalize), bool(demodulate), half_dtype if normalize else None, (out, s_norm, s16), styles)
    with torch.cuda.device(jobs[0][1].device):
        st = lib.pg_modconv_prep_batched(ctypes.byref(table), n, nat.stream_of(jobs[0][1]))
    nat.check(st, 'pg_modconv_prep_batched')


def modconv_prep_peek(styles):
    """The demodulation coefficients [N, Cout] a `modconv_prep_batched` call is producing for `styles` (plain form: not normalised, demodulated), or None;
    the entry stays for the layer."""
    hit = _prep_registry.get(styles.data_ptr())
    if hit is None or hit[0] or not hit[1]:
        return None
    return hit[3][0]


def modconv_prep_clear():
    _prep_registry.clear()


def modconv_prep(w2, styles, cout, normalize=False, demodulate=True, half_dtype=None):
    """One launch per modulated convolution: returns (out [N, Cout], s_norm, s16).  out = demodulation coefficients of the
    (normalised, if `normalize`) styles, or the per-sample style maximum when not demodulating; s_norm / s16 = the
    normalised styles in float32 / `half_dtype` (None unless `normalize`)."""
    hit = _prep_registry.pop(styles.data_ptr(), None) if _prep_registry else None
    if hit is not None and hit[0] == bool(normalize) and hit[1] == bool(demodulate) and hit[2] == (half_dtype if normalize else None) and hit[3][0].shape[1] == cout:
        return hit[3]
    lib = _init().lib
    styles = _f32c(styles.detach(), 'styles')
    n, cin = styles.shape
    if demodulate:
        w2 = _f32c(w2, 'w2')
        assert tuple(w2.shape) == (cout, cin)
    out = torch.empty([n, cout], dtype=torch.float32, device=styles.device)
    s_norm = torch.empty_like(styles) if normalize else None
    s16 = torch.empty([n, cin], dtype=half_dtype, device=styles.device) if (normalize and half_dtype is not None) else None
    code = 0 if s16 is None else nat.PG_DTYPE[half_dtype]
    with torch.cuda.device(styles.device):
        st = lib.pg_modconv_prep(nat.ptr(w2) if demodulate else None, nat.ptr(styles), nat.ptr(out), nat.ptr(s_norm) if s_norm is not None else None,
                                 nat.ptr(s16) if s16 is not None else None, code, n, cout, cin, int(bool(normalize)), int(bool(demodulate)), nat.stream_of(styles))
    nat.check(st, 'pg_modconv_prep')
    return out, s_norm, s16


def instance_norm_stats(x, eps=1e-5):
    """Per-(n,c) mean and 1/sqrt(var+eps) (biased variance) of a contiguous float32 [N,C,H,W]."""
    lib = _init().lib
    x = _f32c(x, 'x')
    n, c, h, w = x.shape
    mean = torch.empty([n * c], dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    with torch.cuda.device(x.device):
        st = lib.pg_instance_norm_stats(nat.ptr(x), nat.ptr(mean), nat.ptr(rstd), n * c, h * w, float(eps), nat.stream_of(x))
    nat.check(st, 'pg_instance_norm_stats')
    return mean, rstd


def spade_norm(x, mean, rstd, gamma, beta):
    """(x - mean) * rstd * (1 + gamma) + beta (networks.py:1715-1722)."""
    lib = _init().lib
    x, gamma, beta = _f32c(x, 'x'), _f32c(gamma, 'gamma'), _f32c(beta, 'beta')
    assert gamma.shape == x.shape and beta.shape == x.shape
    n, c, h, w = x.shape
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        st = lib.pg_spade_norm(nat.ptr(x), nat.ptr(mean), nat.ptr(rstd), nat.ptr(gamma), nat.ptr(beta), nat.ptr(y), n * c, h * w, nat.stream_of(x))
    nat.check(st, 'pg_spade_norm')
    return y


def _gamma_beta_planes(gb, c):
    """[N, 2C, H, W] contiguous float32 -> (tensor, per-sample stride in floats, offset of the beta half in floats)."""
    gb = _f32c(gb, 'gamma_beta')
    assert gb.shape[1] == 2 * c
    return gb, int(gb.stride(0)), int(gb.stride(1)) * c


def spade_train_forward(x, mean, rstd, gamma_beta):
    """Training route of the SPADE combine: y = (x - mean) rstd (1 + gamma) + beta with gamma = gamma_beta[:, :C], beta = gamma_beta[:, C:] read in
    place (csrc/conv2d.hip, pg_spade_train_forward)."""
    lib = _init().lib
    x = _f32c(x, 'x')
    n, c, h, w = x.shape
    gb, gs, off = _gamma_beta_planes(gamma_beta, c)
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        st = lib.pg_spade_train_forward(nat.ptr(x), nat.ptr(mean), nat.ptr(rstd), gb.data_ptr(), gb.data_ptr() + 4 * off, nat.ptr(y), n, c, h * w, gs, gs, nat.stream_of(x))
    nat.check(st, 'pg_spade_train_forward')
    return y


def spade_train_backward(dy, x, mean, rstd, gamma_beta, need_dx=True, need_dgb=True):
    """Gradient of `spade_train_forward` (and of the instance norm inside it) for upstream dy: (dx, d_gamma_beta), either None when not needed.
    Two launches: the plane means of g = dy (1 + gamma) and g x_hat, then one elementwise pass writing dx, dgamma = dy x_hat and dbeta = dy."""
    lib = _init().lib
    x, dy = _f32c(x, 'x'), _f32c(dy, 'dy')
    n, c, h, w = x.shape
    gb, gs, off = _gamma_beta_planes(gamma_beta, c)
    sums = torch.empty([2 * n * c], dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x) if need_dx else None
    dgb = torch.empty_like(gb) if need_dgb else None
    with torch.cuda.device(x.device):
        st = lib.pg_spade_train_backward(nat.ptr(dy), nat.ptr(x), nat.ptr(mean), nat.ptr(rstd), gb.data_ptr(), nat.ptr(sums), nat.ptr(dx),
                                         dgb.data_ptr() if need_dgb else None, dgb.data_ptr() + 4 * off if need_dgb else None,
                                         n, c, h * w, gs, gs, gs, nat.stream_of(x))
    nat.check(st, 'pg_spade_train_backward')
    return dx, dgb


def spade_feat_assemble(feat_upper, feat_lower, mask_upper, mask_lower, denorm_mask_upper, denorm_mask_lower):
    """The masked-mean inpainting of both garment feature maps and their merge (networks.py:2253-2276, 2311-2316) as three
    launches.  feat_*: [N,C,H,W]; the four masks: [N,1,2H,2W] (thresholded at 0.9 and sampled at the even pixels here)."""
    lib = _init().lib
    fu, fl = _f32c(feat_upper, 'feat_upper'), _f32c(feat_lower, 'feat_lower')
    masks = [_f32c(m, 'mask') for m in (mask_upper, mask_lower, denorm_mask_upper, denorm_mask_lower)]
    n, c, h, w = fu.shape
    if fl.shape != fu.shape or any(tuple(m.shape) != (n, 1, 2 * h, 2 * w) for m in masks):
        raise nat.NativeOpError('spade_feat_assemble: feat [N,C,H,W] x2 and masks [N,1,2H,2W] x4 expected')
    mu, ml, du, dl = masks
    sums = torch.empty([2, n * c], dtype=torch.float32, device=fu.device)
    counts = torch.empty([2, n], dtype=torch.float32, device=fu.device)
    out = torch.empty_like(fu)
    with torch.cuda.device(fu.device):
        st = nat.stream_of(fu)
        nat.check(lib.pg_spade_masked_sums(nat.ptr(fu), nat.ptr(mu), nat.ptr(du), nat.ptr(sums[0]), nat.ptr(counts[0]), n, c, h, w, st), 'pg_spade_masked_sums')
        nat.check(lib.pg_spade_masked_sums(nat.ptr(fl), nat.ptr(ml), nat.ptr(dl), nat.ptr(sums[1]), nat.ptr(counts[1]), n, c, h, w, st), 'pg_spade_masked_sums')
        nat.check(lib.pg_spade_feat_assemble(nat.ptr(fu), nat.ptr(fl), nat.ptr(mu), nat.ptr(ml), nat.ptr(du), nat.ptr(dl), nat.ptr(sums[0]), nat.ptr(sums[1]),
                                             nat.ptr(counts[0]), nat.ptr(counts[1]), nat.ptr(out), n, c, h, w, st), 'pg_spade_feat_assemble')
    return out
