"""ctypes binding of the 16-bit (bf16 / fp16) MFMA convolution of ``conv2d_plugin`` (csrc/conv2d16.hip,
csrc/conv2d_kernel16.h): channels-last activations, fp32 accumulation, the StyleGAN2 tail in the epilogue.

No counterpart in the reference -- there cuDNN runs the half-precision convolutions behind ``conv2d_gradfix``
(torch_utils/ops/conv2d_gradfix.py:35-43) for the discriminator's fp16 blocks (training/networks.py:444-523) and the
fp16 synthesis blocks (networks.py:2147-2194).  Forward only; gradients are attached in ``conv2d_gradfix``.

Tensors keep PyTorch's logical NCHW shape; the kernels want ``torch.channels_last`` storage (NHWC) and produce it.
"""

import ctypes
import os

import torch

from . import _native as nat
from . import conv2d_mfma

ACT_INDEX = conv2d_mfma.ACT_INDEX
FUSED_ACTS = conv2d_mfma.FUSED_ACTS

# (KH, KW, stride) geometries instantiated in csrc/conv2d16_inst_*.hip
SUPPORTED = {(3, 3, 1), (1, 1, 1), (2, 2, 1), (2, 1, 1), (1, 2, 1), (3, 3, 2)}
DTYPES = (torch.bfloat16, torch.float16)


class Fusion16(ctypes.Structure):
    """Mirror of ``pg_conv2d16_fusion`` (include/pasta_gan_ops.h)."""
    _fields_ = [('out_scale', ctypes.c_void_p), ('noise', ctypes.c_void_p), ('noise_batch_stride', ctypes.c_int64), ('noise_gain', ctypes.c_float),
                ('bias', ctypes.c_void_p), ('act', ctypes.c_int), ('alpha', ctypes.c_float), ('gain', ctypes.c_float), ('clamp', ctypes.c_float),
                ('residual', ctypes.c_void_p), ('phase_cout', ctypes.c_int), ('noise_phase_stride', ctypes.c_int64)]


_lib = None


def _init():
    global _lib
    if _lib is None:
        lib = conv2d_mfma._init().lib
        i, f, vp, i64 = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_int64
        pi, p64 = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64)
        lib.pg_conv2d16_packed_size.restype = i64
        lib.pg_conv2d16_packed_size.argtypes = [i, i, i, i]
        lib.pg_conv2d16_pack_weight.restype = i
        lib.pg_conv2d16_pack_weight.argtypes = [vp, vp, i, i, i, i, i, pi, i, pi, i, f, i, i, vp, vp, i, vp]
        lib.pg_conv2d16_pack_weight_grouped.restype = i
        lib.pg_conv2d16_pack_weight_grouped.argtypes = [vp, vp, i, i, i64, i, i, i, i, f, i, i, vp, vp, i, vp]
        fwd = [vp, vp, vp, i, i, i, i, i, i, i, i, i, i, i, i, i, i, i64, p64, i, i, i, i, ctypes.POINTER(Fusion16)]
        lib.pg_conv2d16_forward.restype = i
        lib.pg_conv2d16_forward.argtypes = fwd + [vp]
        lib.pg_conv2d16_splitk_plan.restype = i
        lib.pg_conv2d16_splitk_plan.argtypes = [i] * 8
        lib.pg_conv2d16_forward_splitk.restype = i
        lib.pg_conv2d16_forward_splitk.argtypes = fwd + [vp, i, vp]
        lib.pg_conv1x1_small16.restype = i
        lib.pg_conv1x1_small16.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i64, i, f, i, vp]
        lib.pg_conv2d16_pack_weight_batched.restype = i
        lib.pg_conv2d16_pack_weight_batched.argtypes = [ctypes.POINTER(PackJobs), vp]
        lib.pg_conv2d16_up2_fused.restype = i
        lib.pg_conv2d16_up2_fused.argtypes = [vp, vp, vp, i, i, i, i, i, i, i64, p64, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(Fusion16), vp]
        lib.pg_conv2d16_wgrad_plan.restype = i
        lib.pg_conv2d16_wgrad_plan.argtypes = [i] * 8
        lib.pg_conv2d16_wgrad.restype = i
        lib.pg_conv2d16_wgrad.argtypes = [vp, vp, vp, vp, i] + [i] * 13 + [vp]
        _lib = lib
    return _lib


def supported(kh, kw, stride):
    return (int(kh), int(kw), int(stride)) in SUPPORTED


def weight_gradient(x, dy, weight_shape, pad, stride=1):
    """d(loss)/d(weight) (float32, [Cout, Cin, kh, kw]) of y = conv2d(x, w, stride, padding=pad) for 16-bit tensors: a GEMM over pixels on the
    16-bit MFMA with channels-last operands read through the transposing LDS load (csrc/conv2d_wgrad.hip, conv2d16_wgrad).  Image channels that
    are not a multiple of 8 (fromrgb: 6 / 10) are zero-padded here.  None = geometry not covered (callers then ask aten)."""
    lib = _init()
    cout, cin, kh, kw = (int(v) for v in weight_shape)
    n, _, h, w = x.shape
    oh, ow = int(dy.shape[2]), int(dy.shape[3])
    stride = int(stride)
    if x.dtype not in DTYPES or dy.dtype != x.dtype or min(pad) < 0 or (oh, ow) != ((h + 2 * pad[0] - kh) // stride + 1, (w + 2 * pad[1] - kw) // stride + 1):
        return None
    cin_p, cout_p = -(-cin // 8) * 8, -(-cout // 8) * 8
    splits = lib.pg_conv2d16_wgrad_plan(n, cin_p, oh, ow, cout_p, kh, kw, stride)
    if splits <= 0:
        return None
    if cin_p != cin:
        x = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, cin_p - cin))
    if cout_p != cout:
        dy = torch.nn.functional.pad(dy, (0, 0, 0, 0, 0, cout_p - cout))
    x, dy = to_channels_last(x), to_channels_last(dy)
    dw = torch.empty([cout_p, cin_p, kh, kw], dtype=torch.float32, device=x.device)
    ws = torch.empty([splits * kh * kw * cout_p * cin_p], dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = lib.pg_conv2d16_wgrad(nat.ptr(x), nat.ptr(dy), nat.ptr(dw), nat.ptr(ws), nat.PG_DTYPE[x.dtype], n, cin_p, h, w, cout_p, kh, kw, stride,
                                   int(pad[0]), int(pad[1]), oh, ow, splits, nat.stream_of(x))
    if st == -2:
        return None
    nat.check(st, 'pg_conv2d16_wgrad')
    return dw[:cout, :cin] if (cin_p != cin or cout_p != cout) else dw


def to_channels_last(x):
    """NHWC storage of a logical [N, C, H, W] tensor (a no-op when it already is channels-last)."""
    return x.contiguous(memory_format=torch.channels_last)


def _f32(t, name, numel=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise nat.NativeOpError(f'conv2d_mfma16: {name} must be a GPU tensor')
    t = t.detach().to(torch.float32).contiguous()
    if numel is not None and t.numel() != numel:
        raise nat.NativeOpError(f'conv2d_mfma16: {name} has {t.numel()} elements, expected {numel}')
    return t


PACK_MAX_JOBS = 16


class PackJobs(ctypes.Structure):
    """pg_conv2d16_pack_jobs of include/pasta_gan_ops.h."""
    _fields_ = [('w', ctypes.c_void_p * PACK_MAX_JOBS), ('packed', ctypes.c_void_p * PACK_MAX_JOBS), ('styles', ctypes.c_void_p * PACK_MAX_JOBS),
                ('dcoefs', ctypes.c_void_p * PACK_MAX_JOBS), ('cout', ctypes.c_int * PACK_MAX_JOBS), ('cin', ctypes.c_int * PACK_MAX_JOBS),
                ('flags', ctypes.c_int * PACK_MAX_JOBS), ('dcoefs_mod', ctypes.c_int * PACK_MAX_JOBS), ('scale', ctypes.c_float * PACK_MAX_JOBS),
                ('njobs', ctypes.c_int), ('nsamples', ctypes.c_int), ('dtype', ctypes.c_int)]


_pack_registry = {}     # (weight address, styles address) -> (dtype, flip, transpose_oi, packed, per, keep-alive): results of pack_weight_batched awaiting their layer


def pack_weight_batched(jobs, dtype):
    """The per-sample 3x3 packs of several modulated convolutions in ONE launch (pg_conv2d16_pack_weight_batched).  jobs: list of
    (w float32 [O, I, 3, 3] -- [I, O, 3, 3] when transpose_oi; or the [I, 4 O, 3, 2] stack of `conv_up2_fused` --, flip, transpose_oi, styles [N, Cin], dcoefs [N, D] | None) with D dividing the
    pack's Cout (the four stacked phases of an up = 2 layer share one coefficient row).  Results wait in a registry keyed by (w, styles) addresses for
    the layer's own `pack_lookup`; `pack_clear()` empties it."""
    lib = _init()
    assert dtype in DTYPES and 0 < len(jobs) <= PACK_MAX_JOBS
    table = PackJobs()
    n = int(jobs[0][3].shape[0])
    table.njobs, table.nsamples, table.dtype = len(jobs), n, nat.PG_DTYPE[dtype]
    for j, (w, flip, transpose_oi, styles, dcoefs) in enumerate(jobs):
        assert w.dtype == torch.float32 and w.is_contiguous() and w.is_cuda and tuple(w.shape[2:]) in ((3, 3), (3, 2))
        k32 = int(w.shape[3]) == 2          # the 3 x 2 stack of `conv_up2_fused`
        cin, cout = (int(w.shape[0]), int(w.shape[1])) if transpose_oi else (int(w.shape[1]), int(w.shape[0]))
        assert styles.dtype == torch.float32 and styles.is_contiguous() and tuple(styles.shape) == (n, cin)
        per = lib.pg_conv2d16_packed_size(cout, cin, 3, 2 if k32 else 3)
        packed = torch.empty([n * per], dtype=dtype, device=w.device)
        table.w[j], table.packed[j], table.styles[j] = w.data_ptr(), packed.data_ptr(), styles.data_ptr()
        if dcoefs is not None:
            assert dcoefs.dtype == torch.float32 and dcoefs.is_contiguous() and dcoefs.shape[0] == n and cout % int(dcoefs.shape[1]) == 0
            table.dcoefs[j], table.dcoefs_mod[j] = dcoefs.data_ptr(), int(dcoefs.shape[1])
        else:
            table.dcoefs[j], table.dcoefs_mod[j] = None, cout
        table.cout[j], table.cin[j], table.flags[j], table.scale[j] = cout, cin, int(bool(flip)) | (int(bool(transpose_oi)) << 1) | (int(k32) << 2), 1.0
        _pack_registry[(w.data_ptr(), styles.data_ptr())] = (dtype, bool(flip), bool(transpose_oi), packed, per, (w, styles, dcoefs))
    with torch.cuda.device(jobs[0][0].device):
        st = lib.pg_conv2d16_pack_weight_batched(ctypes.byref(table), nat.stream_of(jobs[0][0]))
    nat.check(st, 'pg_conv2d16_pack_weight_batched')


def pack_lookup(w, dtype, flip, transpose_oi, styles):
    """(packed, per_sample_stride) a `pack_weight_batched` call left for this (weight, styles) pair, or None."""
    if not _pack_registry or styles is None:
        return None
    hit = _pack_registry.pop((w.data_ptr(), styles.data_ptr()), None)
    if hit is None or hit[0] != dtype or hit[1] != bool(flip) or hit[2] != bool(transpose_oi):
        return None
    return hit[3], hit[4]


def pack_clear():
    _pack_registry.clear()


def pack_weight(w, dtype, scale=1.0, flip=False, transpose_oi=False, taps=None, styles=None, dcoefs=None):
    """float32 OIHW (IOHW when `transpose_oi`) weights -> the 16-bit kernel layout, one copy per style row when `styles`
    ([N, Cin]) and / or `dcoefs` ([N, Cout]) are given: T(w * scale * styles[n, ci] * dcoefs[n, co]) -- the per-sample
    weights of the reference's fused modulated convolution (networks.py:85-94).  `taps` = (rows, cols) index lists
    selects a sub-kernel (the phases of a transposed convolution).  Returns (packed, per_sample_stride or 0, (kh, kw))."""
    lib = _init()
    assert dtype in DTYPES
    w = _f32(w, 'weight')
    if transpose_oi:
        cin, cout, kh, kw = w.shape
    else:
        cout, cin, kh, kw = w.shape
    ty, tx = (list(taps[0]), list(taps[1])) if taps is not None else (list(range(kh)), list(range(kw)))
    n = 1
    if styles is not None:
        styles = _f32(styles, 'styles')
        n = styles.shape[0]
        assert styles.shape[1] == cin
    if dcoefs is not None:
        dcoefs = _f32(dcoefs, 'dcoefs', None)
        n = dcoefs.shape[0]
        assert dcoefs.shape[1] == cout and (styles is None or styles.shape[0] == n)
    per = lib.pg_conv2d16_packed_size(cout, cin, len(ty), len(tx))
    packed = torch.empty([n * per], dtype=dtype, device=w.device)
    ay, ax = (ctypes.c_int * len(ty))(*ty), (ctypes.c_int * len(tx))(*tx)
    with torch.cuda.device(w.device):
        st = lib.pg_conv2d16_pack_weight(nat.ptr(w), nat.ptr(packed), nat.PG_DTYPE[dtype], cout, cin, kh, kw, ay, len(ty), ax, len(tx),
                                         float(scale), int(bool(flip)), int(bool(transpose_oi)), nat.ptr(styles), nat.ptr(dcoefs), n, nat.stream_of(w))
    nat.check(st, 'pg_conv2d16_pack_weight')
    return packed, (per if (styles is not None or dcoefs is not None) else 0), (len(ty), len(tx))


def pack_weight_grouped(ws, dtype, scale=1.0, flip=False, transpose_oi=False, styles=None, dcoefs=None):
    """`pack_weight` of G same-shape weight tensors stacked as [G, ...] in one launch (shared styles / dcoefs): returns
    (packed [G, N * per], per_sample_stride or 0)."""
    lib = _init()
    assert dtype in DTYPES
    ws = _f32(ws, 'weights')
    g = ws.shape[0]
    if transpose_oi:
        cin, cout, kh, kw = ws.shape[1:]
    else:
        cout, cin, kh, kw = ws.shape[1:]
    n = 1
    if styles is not None:
        styles = _f32(styles, 'styles')
        n = styles.shape[0]
        assert styles.shape[1] == cin
    if dcoefs is not None:
        dcoefs = _f32(dcoefs, 'dcoefs', None)
        n = dcoefs.shape[0]
        assert dcoefs.shape[1] == cout and (styles is None or styles.shape[0] == n)
    per = lib.pg_conv2d16_packed_size(cout, cin, kh, kw)
    packed = torch.empty([g, n * per], dtype=dtype, device=ws.device)
    with torch.cuda.device(ws.device):
        st = lib.pg_conv2d16_pack_weight_grouped(nat.ptr(ws), nat.ptr(packed), nat.PG_DTYPE[dtype], g, ws[0].numel(), cout, cin, kh, kw,
                                                 float(scale), int(bool(flip)), int(bool(transpose_oi)), nat.ptr(styles), nat.ptr(dcoefs), n, nat.stream_of(ws))
    nat.check(st, 'pg_conv2d16_pack_weight_grouped')
    return packed, (per if (styles is not None or dcoefs is not None) else 0)


def phases_supported(cout):
    """Four-phase launches need the per-phase channel count to be a whole number of cout blocks (pg_conv2d16_fusion)."""
    return cout == 32 or cout % 64 == 0


def conv2d_forward(x, packed, cout, kh, kw, stride=1, pad=(0, 0), out_hw=None, y=None, out_step=(1, 1), out_off=(0, 0), sample_stride=0,
                   out_dtype=None, out_scale=None, noise=None, noise_gain=1.0, bias=None, act='linear', alpha=0.0, gain=1.0, clamp=None, residual=None,
                   phases=False):
    """One launch of the 16-bit MFMA convolution.  `x`: logical [N, Cin, H, W] bf16 / fp16, Cin % 16 == 0 (converted to
    channels-last storage if it is not); `packed` from `pack_weight` (`sample_stride` = its per-sample stride for
    modulated weights).  Writes y[n, co, oy*step+off, ox*step+off]; allocates a channels-last `y` of x's dtype (or a
    contiguous float32 one for out_dtype=torch.float32) when none is given.
    `phases=True`: `packed` holds 4 * cout output channels -- the four phase kernels (a, b) = (0,0), (0,1), (1,0), (1,1) of an
    up-by-2 layer stacked along Cout -- and block 2a + b is written to y[n, :, 2*oy + a, 2*ox + b] (y given, [N, cout, 2*OH, 2*OW]);
    out_scale / bias have cout entries per row, noise is [N or 1, 4, OH, OW] (phase-major)."""
    lib = _init()
    if x.dtype not in DTYPES or not x.is_cuda or x.ndim != 4:
        raise nat.NativeOpError('conv2d_mfma16: x must be a 4-D bf16 / fp16 GPU tensor')
    x = to_channels_last(x)
    n, cin, h, w = x.shape
    if cin % 16 != 0:
        raise nat.NativeOpError('conv2d_mfma16: Cin must be a multiple of 16 (pad the channels)')
    pad_y, pad_x = pad
    if out_hw is None:
        out_hw = ((h + 2 * pad_y - kh) // stride + 1, (w + 2 * pad_x - kw) // stride + 1)
    oh, ow = int(out_hw[0]), int(out_hw[1])
    out_dtype = out_dtype or x.dtype
    if y is None:
        assert tuple(out_step) == (1, 1) and tuple(out_off) == (0, 0)
        y = torch.empty([n, cout, oh, ow], dtype=out_dtype, device=x.device,
                        memory_format=torch.contiguous_format if out_dtype == torch.float32 else torch.channels_last)
    else:
        assert y.dtype == out_dtype and y.device == x.device and y.shape[0] == n and y.shape[1] == cout
    if phases:
        assert residual is None and tuple(y.shape) == (n, cout, 2 * oh, 2 * ow) and phases_supported(cout)
        out_step, out_off = (2, 2), (0, 0)
    fz = Fusion16()
    keep = []

    def dev(t, name, numel):
        t = _f32(t, name, numel)
        if t is None:
            return None
        keep.append(t)
        return t.data_ptr()

    fz.out_scale = dev(out_scale, 'out_scale', n * cout)
    if noise is not None:
        noise = _f32(noise, 'noise')
        per_image = (4 if phases else 1) * oh * ow
        if noise.numel() == per_image:
            fz.noise_batch_stride = 0
        elif noise.numel() == n * per_image:
            fz.noise_batch_stride = per_image
        else:
            raise nat.NativeOpError('conv2d_mfma16: noise must have OH*OW or N*OH*OW elements (x 4, phase-major, for a four-phase launch)')
        keep.append(noise)
        fz.noise = noise.data_ptr()
    fz.noise_gain = float(noise_gain)
    fz.bias = dev(bias, 'bias', cout)
    fz.act, fz.alpha, fz.gain = ACT_INDEX[act], float(alpha), float(gain)
    fz.clamp = -1.0 if clamp is None else float(clamp)
    if residual is not None:
        if residual.dtype != y.dtype or residual.shape != y.shape or residual.stride() != y.stride():
            raise nat.NativeOpError('conv2d_mfma16: residual must match y in dtype, shape and strides')
        keep.append(residual)
        fz.residual = residual.data_ptr()
    cout_total = cout
    if phases:
        fz.phase_cout, fz.noise_phase_stride, cout_total = cout, oh * ow, 4 * cout
    args = [nat.ptr(x), nat.ptr(packed), nat.ptr(y), nat.PG_DTYPE[x.dtype], nat.PG_DTYPE[out_dtype], n, cin, h, w, cout_total, kh, kw, int(stride),
            int(pad_y), int(pad_x), oh, ow, int(sample_stride), nat.i64arr(y.stride()), int(out_step[0]), int(out_step[1]), int(out_off[0]), int(out_off[1]),
            ctypes.byref(fz)]
    tl = conv2d_mfma._timeline
    with torch.cuda.device(x.device):
        if tl is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        ksplit = 1
        if os.environ.get('PG_CONV_SPLITK', '1') != '0':
            ksplit = lib.pg_conv2d16_splitk_plan(n, cin, oh, ow, cout_total, kh, kw, int(stride))
        if ksplit > 1:
            ws = torch.empty([ksplit * n * cout_total * oh * ow], dtype=torch.float32, device=x.device)
            st = lib.pg_conv2d16_forward_splitk(*args, nat.ptr(ws), ksplit, nat.stream_of(x))
        else:
            st = lib.pg_conv2d16_forward(*args, nat.stream_of(x))
        if tl is not None:
            ev1.record()
            tl.append(((kh, kw, int(stride), 'mfma16', f'N{n} {cin}->{cout_total}{" (4 phases)" if phases else ""} {h}x{w} {str(x.dtype)[6:]}'), 2.0 * n * cout_total * oh * ow * cin * kh * kw, ev0, ev1,
                       x.element_size() * (x.numel() + n * cout_total * oh * ow) + (y.element_size() - x.element_size()) * n * cout_total * oh * ow
                       + packed.numel() * packed.element_size()))       # algorithmic bytes: x + y + the packed weights this launch reads
    nat.check(st, 'pg_conv2d16_forward')
    return y


def conv_up2_fused(x, packed, cout, fir_x, sample_stride=0, out_scale=None, noise=None, noise_gain=1.0, bias=None, act='linear', alpha=0.0, gain=1.0, clamp=None):
    """The up = 2 modulated 3x3 layer of conv2d_resample.py:125-142 (transposed convolution, then the separable 4-tap filter with padding 1 and gain 4)
    + the StyleGAN2 tail in ONE launch (pg_conv2d16_up2_fused, csrc/conv2d_up2f16.h): `packed` = `pack_weight(transpose_oi=True)` of the [Cin, 4 * cout, 3, 2]
    stack that carries the y half of the filter, `fir_x` = 2 * fx (four floats), `noise` = [N or 1, 4, H, W] phase-major.  Returns channels-last [N, cout, 2H, 2W]."""
    lib = _init()
    if x.dtype not in DTYPES or not x.is_cuda or x.ndim != 4:
        raise nat.NativeOpError('conv2d_mfma16: x must be a 4-D bf16 / fp16 GPU tensor')
    x = to_channels_last(x)
    n, cin, h, w = x.shape
    y = torch.empty([n, cout, 2 * h, 2 * w], dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    fz = Fusion16()
    keep = []

    def dev(t, name, numel):
        t = _f32(t, name, numel)
        if t is None:
            return None
        keep.append(t)
        return t.data_ptr()

    fz.out_scale = dev(out_scale, 'out_scale', n * cout)
    if noise is not None:
        noise = _f32(noise, 'noise')
        if noise.numel() == 4 * h * w:
            fz.noise_batch_stride = 0
        elif noise.numel() == n * 4 * h * w:
            fz.noise_batch_stride = 4 * h * w
        else:
            raise nat.NativeOpError('conv2d_mfma16: the fused up-by-2 layer wants noise [N or 1, 4, H, W], phase-major')
        keep.append(noise)
        fz.noise = noise.data_ptr()
    fz.noise_gain = float(noise_gain)
    fz.noise_phase_stride = h * w
    fz.bias = dev(bias, 'bias', cout)
    fz.act, fz.alpha, fz.gain = ACT_INDEX[act], float(alpha), float(gain)
    fz.clamp = -1.0 if clamp is None else float(clamp)
    fz.phase_cout = cout
    taps = (ctypes.c_float * 4)(*[float(v) for v in fir_x])
    tl = conv2d_mfma._timeline
    with torch.cuda.device(x.device):
        if tl is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        st = lib.pg_conv2d16_up2_fused(nat.ptr(x), nat.ptr(packed), nat.ptr(y), nat.PG_DTYPE[x.dtype], n, cin, h, w, cout, int(sample_stride),
                                       nat.i64arr(y.stride()), taps, ctypes.byref(fz), nat.stream_of(x))
        if tl is not None:
            ev1.record()
            # executed flops: 18 tap-products per position and (cin, cout) pair (the launch's tiles overlap by 2 of 32 columns on top of that);
            # SURVEY 8d counts the transposed convolution as 2 * N * Cin * Hin * Win * Cout * 9: `honest_flops`
            tl.append(((3, 3, 1, 'mfma16', f'N{n} {cin}->{cout} (up2 fused-x) {h}x{w} {str(x.dtype)[6:]}'), 2.0 * n * cout * h * w * cin * 18, ev0, ev1,
                       x.element_size() * (x.numel() + y.numel()) + packed.numel() * packed.element_size()))
    nat.check(st, 'pg_conv2d16_up2_fused')
    return y


def pack_transposed(w_iohw, dtype, stride, pad, in_hw, out_hw, scale=1.0, styles=None, dcoefs=None):
    """Per-phase packed weights of conv_transpose2d(x, w_iohw, stride, padding=pad): list of (phase, packed, sample_stride)."""
    kh, kw = int(w_iohw.shape[2]), int(w_iohw.shape[3])
    phases = conv2d_mfma.transposed_phases(kh, kw, stride, pad[0], pad[1], in_hw, out_hw)
    if phases is None or not all(supported(len(ph['ky']), len(ph['kx']), 1) for ph in phases):
        return None
    out = []
    for ph in phases:
        packed, per, _ = pack_weight(w_iohw, dtype, scale=scale, transpose_oi=True, taps=(ph['ky'], ph['kx']), styles=styles, dcoefs=dcoefs)
        out.append((ph, packed, per))
    return out


def conv_transpose2d_forward(x, packed_phases, cout, out_hw, stride=2, **fusion):
    """Run the phases of `pack_transposed` into one dense channels-last [N, Cout, OH, OW] output."""
    n = x.shape[0]
    y = torch.empty([n, cout, out_hw[0], out_hw[1]], dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    x = to_channels_last(x)
    for ph, packed, per in packed_phases:
        conv2d_forward(x, packed, cout, len(ph['ky']), len(ph['kx']), stride=1, pad=ph['pad'], out_hw=ph['out_hw'], y=y,
                       out_step=(stride, stride), out_off=ph['off'], sample_stride=per, **fusion)
    return y


def conv1x1_small(x, w, styles=None, bias=None, skip=None, clamp=None, skip_up2=False):
    """The ToRGB / parsing head (networks.py:1957-1967) as one streaming pass: float32 NCHW
    clamp(sum_c x[n,c,p] * w[o,c] * styles[n,c] + bias[o]) + skip, x 16-bit channels-last, Cout <= 8.  `skip_up2`: `skip` is the half-resolution
    image [N, Cout, H/2, W/2]; it is up-sampled in the same pass as upfirdn2d.upsample2d(skip, [1, 3, 3, 1]) would."""
    lib = _init()
    x = to_channels_last(x)
    n, cin, h, wd = x.shape
    w = _f32(w, 'weight').reshape(w.shape[0], -1)
    cout = w.shape[0]
    assert w.shape[1] == cin
    styles, bias = _f32(styles, 'styles', n * cin), _f32(bias, 'bias', cout)
    y = torch.empty([n, cout, h, wd], dtype=torch.float32, device=x.device)
    if skip is not None:
        want = (n, cout, h // 2, wd // 2) if skip_up2 else tuple(y.shape)
        if skip.dtype != torch.float32 or tuple(skip.shape) != want or (skip_up2 and (h % 2 or wd % 2)):
            raise nat.NativeOpError('conv1x1_small: skip must be float32 [N, Cout, H, W] (or [N, Cout, H/2, W/2] with skip_up2)')
        skip = skip.contiguous()
    with torch.cuda.device(x.device):
        st = lib.pg_conv1x1_small16(nat.ptr(x), nat.ptr(w), nat.ptr(styles), nat.ptr(bias), nat.ptr(skip), nat.ptr(y), nat.PG_DTYPE[x.dtype],
                                    n, cin, h * wd, cout, -1.0 if clamp is None else float(clamp), wd if (skip_up2 and skip is not None) else 0, nat.stream_of(x))
    nat.check(st, 'pg_conv1x1_small16')
    return y
