"""Patch routing on the GPU: the 10-part perspective normalise / de-normalise of the reference's data loader
(training/dataset.py:2373-2542 ``get_crop``, :2555-2700 ``normalize``).

The reference does this per sample with ~44 ``cv2.warpPerspective`` calls, 15 ``cv2.erode`` calls and NumPy compositing on the
DataLoader's main thread.  Here the keypoint geometry stays on the host (18 joints, 20 tiny 8x8 solves), and the pixel work runs
as three batched launches of ``patch_routing_plugin`` (csrc/patch_routing.hip) plus the paste kernels:

1. every image -> patch warp of the sample (up to 30 jobs, 128x128 patches) in one launch,
2. every patch -> canvas warp (up to 30 jobs, 512x512) in one launch,
3. the erode-and-paste of each part, in the reference's order (later parts overwrite earlier ones).

Images are uint8 HxWx3 tensors on the GPU (NumPy arrays are uploaded); the five results have the reference's shapes and dtype.
GPU tensors run the HIP kernels; CPU tensors (device='cpu': BASELINE config 1, the loader without a GPU) the same arithmetic in NumPy.
"""

import ctypes

import numpy as np
import torch

from torch_utils import custom_ops
from torch_utils.ops import _native as nat

ORDER = ['cnose', 'cneck', 'rshoulder', 'relbow', 'rwrist', 'lshoulder', 'lelbow', 'lwrist', 'rhip', 'rknee', 'rankle', 'lhip', 'lknee',
         'lankle', 'reye', 'leye', 'rear', 'lear']
BPARTS = [["rshoulder", "rhip", "lhip", "lshoulder"], ["lshoulder", "rshoulder", "cnose"], ["lshoulder", "lelbow"], ["lelbow", "lwrist"],
          ["rshoulder", "relbow"], ["relbow", "rwrist"], ["lhip", "lknee"], ["lknee", "lankle"], ["rhip", "rknee"], ["rknee", "rankle"]]
SLEEVE_PARTS = (2, 3, 4, 5)
_J = {name: i for i, name in enumerate(ORDER)}


class WarpJob(ctypes.Structure):
    """Mirror of ``pg_warp_job`` (include/pasta_gan_ops.h)."""
    _fields_ = [('src', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('src_h', ctypes.c_int), ('src_w', ctypes.c_int), ('dst_h', ctypes.c_int),
                ('dst_w', ctypes.c_int), ('channels', ctypes.c_int), ('block_w', ctypes.c_int), ('minv', ctypes.c_double * 9)]


class ComposeJob(ctypes.Structure):
    """Mirror of ``pg_compose_job`` (include/pasta_gan_ops.h)."""
    _fields_ = [('canvas', ctypes.c_void_p), ('canvas2', ctypes.c_void_p), ('patch', ctypes.c_void_p * 10), ('mask', ctypes.c_void_p * 10),
                ('nparts', ctypes.c_int), ('to_canvas2', ctypes.c_int * 10), ('pad_', ctypes.c_int)]


_WARP_DT = np.dtype([('src', 'u8'), ('dst', 'u8'), ('src_h', 'i4'), ('src_w', 'i4'), ('dst_h', 'i4'), ('dst_w', 'i4'), ('channels', 'i4'), ('block_w', 'i4'),
                     ('minv', 'f8', (9,))])
_COMPOSE_DT = np.dtype([('canvas', 'u8'), ('canvas2', 'u8'), ('patch', 'u8', (10,)), ('mask', 'u8', (10,)), ('nparts', 'i4'), ('to_canvas2', 'i4', (10,)), ('pad_', 'i4')])
assert _WARP_DT.itemsize == ctypes.sizeof(WarpJob) and _COMPOSE_DT.itemsize == ctypes.sizeof(ComposeJob)

_plugin = None


def _init():
    global _plugin
    if _plugin is None:
        plugin = custom_ops.get_plugin('patch_routing_plugin')
        lib = plugin.lib
        lib.pg_warp_perspective_u8.restype = ctypes.c_int
        lib.pg_warp_perspective_u8.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        lib.pg_patch_compose_u8.restype = ctypes.c_int
        lib.pg_patch_compose_u8.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
        lib.pg_patch_compose_ordered_u8.restype = ctypes.c_int
        lib.pg_patch_compose_ordered_u8.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        _plugin = plugin
    return _plugin


# ------------------------------------------------------------------------------------------- host geometry

def get_perspective_transform(src, dst):
    """cv2.getPerspectiveTransform: the homography through four point pairs (8 unknowns, m22 = 1)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    x, y, X, Y = src[:, 0], src[:, 1], dst[:, 0], dst[:, 1]
    zero, one = np.zeros(4), np.ones(4)
    a = np.concatenate([np.stack([x, y, one, zero, zero, zero, -x * X, -y * X], 1), np.stack([zero, zero, zero, x, y, one, -x * Y, -y * Y], 1)])
    return np.append(np.linalg.solve(a, np.concatenate([X, Y])), 1.0).reshape(3, 3)


def invert3x3(m):
    """The adjugate formula cv::invert uses for 3x3 matrices (so the kernel sees the matrix OpenCV's warp would see)."""
    m = np.asarray(m, np.float64)
    c = lambda r0, c0, r1, c1: m[r0, c0] * m[r1, c1]
    det = m[0, 0] * (c(1, 1, 2, 2) - c(1, 2, 2, 1)) - m[0, 1] * (c(1, 0, 2, 2) - c(1, 2, 2, 0)) + m[0, 2] * (c(1, 0, 2, 1) - c(1, 1, 2, 0))
    if det == 0:
        return np.zeros((3, 3))
    d = 1.0 / det
    return np.array([[(c(1, 1, 2, 2) - c(1, 2, 2, 1)) * d, (c(0, 2, 2, 1) - c(0, 1, 2, 2)) * d, (c(0, 1, 1, 2) - c(0, 2, 1, 1)) * d],
                     [(c(1, 2, 2, 0) - c(1, 0, 2, 2)) * d, (c(0, 0, 2, 2) - c(0, 2, 2, 0)) * d, (c(0, 2, 1, 0) - c(0, 0, 1, 2)) * d],
                     [(c(1, 0, 2, 1) - c(1, 1, 2, 0)) * d, (c(0, 1, 2, 0) - c(0, 0, 2, 1)) * d, (c(0, 0, 1, 1) - c(0, 1, 1, 0)) * d]])


def _valid(joints, names):
    return bool((joints[[_J[n] for n in names], 2] >= 0.1).all())


_LEG_FALLBACK = {('lhip', 'lknee'): ('lhip', 'lknee', 0.85), ('rhip', 'rknee'): ('rhip', 'rknee', 0.85),
                 ('lknee', 'lankle'): ('lknee', 'lankle', 0.80), ('rknee', 'rankle'): ('rknee', 'rankle', 0.80)}

def get_crop(keypoints, bpart, wh, o_w, o_h, ar=1.0, _quad_only=False):
    """Homographies (image -> patch, patch -> image) of one body part, or (None, None) when its joints are missing.
    Same decisions as dataset.py:2373-2542: joint-confidence fall-backs, leg completion from the torso length, widened torso
    and neck quadrilaterals, limb strips of aspect ratio `ar` with side-dependent widening."""
    kp = np.asarray(keypoints)
    names = list(bpart)
    pts = lambda ns: np.float32(kp[[_J[n] for n in ns], :2])
    quad = None
    if not _valid(kp, names):
        if tuple(names) in _LEG_FALLBACK:
            root, missing, factor = _LEG_FALLBACK[tuple(names)]
            if not _valid(kp, [root]) or not _valid(kp, ['lhip', 'rhip', 'cneck']):
                return None, None
            a = pts([root])[0]
            torso = pts(['lhip', 'rhip', 'cneck'])
            length = (np.linalg.norm(torso[2] - torso[1]) + np.linalg.norm(torso[2] - torso[0])) / 2
            far = kp[_J[missing]]
            if far[2] > 0:                                   # a low-confidence joint still gives the direction
                b = a + length * ((far[0:2] - a) / np.linalg.norm(a - far[0:2])) * factor
            else:
                b = np.float32([a[0], a[1] + length * factor])
            names, src = [root], np.float32([a, b])
        elif names == ['lshoulder', 'rshoulder', 'cnose']:
            names = ['lshoulder', 'rshoulder', 'rshoulder']
            if not _valid(kp, names):
                return None, None
            src = pts(names)
        else:
            return None, None
    else:
        src = pts(names)

    inside = lambda q: q[0] > 0 and q[1] > 0 and q[0] < o_w and q[1] < o_h

    def widen(i_left, i_right, frac):                        # push two corners apart by `frac` of their distance, if that stays inside
        seg = (src[i_right] - src[i_left]) / frac
        lo, hi = src[i_left] - seg, src[i_right] + seg
        if inside(lo):
            src[i_left] = lo
        if inside(hi):
            src[i_right] = hi

    if src.shape[0] == 4:
        widen(1, 2, 4)
        widen(0, 3, 5)
        quad = src
    elif src.shape[0] == 3:
        widen(1, 0, 5)
        seg = src[1] - src[0]
        normal = np.array([-seg[1], seg[0]])
        if normal[1] > 0.0:
            normal = -normal
        a, b, c, d = src[0] + normal, src[0], src[1], src[1] + normal
        lift = ((c[1] + b[1]) / 2 - (a[1] + d[1]) / 2) / 2
        a[1] += lift
        d[1] += lift
        quad = np.float32([d, c, b, a])
    else:
        seg = src[1] - src[0]
        normal = np.array([-seg[1], seg[0]])
        a, b, c, d = _limb_corners(src, normal, ar / 2.0, names)
        quad = np.float32([a, d, c, b])

    if _quad_only:
        return quad
    part_dst = np.float32(wh * np.float32([[0.0, 0.0], [0.0, 1.0], [1.0, 1.0], [1.0, 0.0]]))
    return get_perspective_transform(quad, part_dst), get_perspective_transform(part_dst, quad)


def perspective_transforms(src, dst):
    """`get_perspective_transform` of K point-quadruple pairs at once: src, dst [K, 4, 2] -> [K, 3, 3].  One batched LAPACK call (each system is factored
    on its own, exactly as the single solves are: the results are bit-identical, tests/test_patch_routing.py)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    k = src.shape[0]
    x, y, X, Y = src[:, :, 0], src[:, :, 1], dst[:, :, 0], dst[:, :, 1]
    zero, one = np.zeros((k, 4)), np.ones((k, 4))
    a = np.concatenate([np.stack([x, y, one, zero, zero, zero, -x * X, -y * X], 2), np.stack([zero, zero, zero, x, y, one, -x * Y, -y * Y], 2)], axis=1)
    sol = np.linalg.solve(a, np.concatenate([X, Y], axis=1)[:, :, None])[:, :, 0]
    return np.concatenate([sol, np.ones((k, 1))], axis=1).reshape(k, 3, 3)


def invert3x3_batch(m):
    """`invert3x3` of [K, 3, 3] matrices, the same products and sums in the same order (element-wise float64: bit-identical to the scalar form)."""
    m = np.asarray(m, np.float64)
    c = lambda r0, c0, r1, c1: m[:, r0, c0] * m[:, r1, c1]
    det = m[:, 0, 0] * (c(1, 1, 2, 2) - c(1, 2, 2, 1)) - m[:, 0, 1] * (c(1, 0, 2, 2) - c(1, 2, 2, 0)) + m[:, 0, 2] * (c(1, 0, 2, 1) - c(1, 1, 2, 0))
    with np.errstate(divide='ignore', invalid='ignore'):
        d = np.where(det == 0, 0.0, 1.0 / np.where(det == 0, 1.0, det))
    out = np.stack([(c(1, 1, 2, 2) - c(1, 2, 2, 1)) * d, (c(0, 2, 2, 1) - c(0, 1, 2, 2)) * d, (c(0, 1, 1, 2) - c(0, 2, 1, 1)) * d,
                    (c(1, 2, 2, 0) - c(1, 0, 2, 2)) * d, (c(0, 0, 2, 2) - c(0, 2, 2, 0)) * d, (c(0, 2, 1, 0) - c(0, 0, 1, 2)) * d,
                    (c(1, 0, 2, 1) - c(1, 1, 2, 0)) * d, (c(0, 1, 2, 0) - c(0, 0, 2, 1)) * d, (c(0, 0, 1, 1) - c(0, 1, 1, 0)) * d], axis=1)
    return np.where((det == 0)[:, None], 0.0, out)


def _limb_corners(src, normal, alpha, names):
    """Corners of a limb strip: +-alpha * normal around the bone, then side-dependent widening in the reference's order (:2513-2535)."""
    a, b = src[0] + alpha * normal, src[0] - alpha * normal
    c, d = src[1] - alpha * normal, src[1] + alpha * normal
    for group, (up, down) in ((('rhip', 'rknee'), (1.0, None)), (('lhip', 'lknee'), (None, 1.0)),
                              (('relbow', 'rwrist'), (0.45, 0.1)), (('lelbow', 'lwrist'), (0.1, 0.45))):
        if any(g in names for g in group):
            if up is not None:
                a, d = a + alpha * normal * up, d + alpha * normal * up
            if down is not None:
                b, c = b - alpha * normal * down, c - alpha * normal * down
    return a, b, c, d


# ------------------------------------------------------------------------------------------- device work

def _block_width(dst_h, dst_w):
    bh0 = min(16, dst_h)
    return min(1024 // bh0, dst_w)


def _gpu_u8(img, device):
    """uint8 image on `device` (name kept from the GPU-only version; CPU tensors take the NumPy route below)."""
    if isinstance(img, torch.Tensor):
        t = img
    else:
        t = torch.from_numpy(np.ascontiguousarray(img))
    if t.dtype != torch.uint8:
        raise nat.NativeOpError('patch_routing: images must be uint8')
    return t.to(device).contiguous()


def _warp_perspective_cpu(src, m, wh):
    """The arithmetic of `warp_perspective_u8_kernel` (csrc/patch_routing.hip) in vectorised NumPy, for CPU tensors (config 1
    runs the loader without a GPU): fp64 coordinates evaluated block origin + offset with separately rounded products and sums,
    5 fractional bits, 15-bit bilinear weights, zero border."""
    w, h = wh
    a = src.numpy()
    squeeze = a.ndim == 2
    if squeeze:
        a = a[:, :, None]
    sh, sw, c = a.shape
    mi = invert3x3(m).reshape(9)
    bw = _block_width(h, w)
    ys, xs = np.mgrid[0:h, 0:w]
    xb = (xs // bw) * bw
    x1 = (xs - xb).astype(np.float64)
    xb, yf = xb.astype(np.float64), ys.astype(np.float64)
    X0 = (mi[0] * xb + mi[1] * yf) + mi[2]
    Y0 = (mi[3] * xb + mi[4] * yf) + mi[5]
    W0 = (mi[6] * xb + mi[7] * yf) + mi[8]
    W = W0 + mi[6] * x1
    with np.errstate(divide='ignore', invalid='ignore'):
        W = np.where(W != 0.0, 32.0 / W, 0.0)
    fX = np.clip((X0 + mi[0] * x1) * W, -2147483648.0, 2147483647.0)
    fY = np.clip((Y0 + mi[3] * x1) * W, -2147483648.0, 2147483647.0)
    X, Y = np.rint(fX).astype(np.int64), np.rint(fY).astype(np.int64)            # round half to even, as cvRound
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    w00, w01, w10, w11 = (32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32
    exact = (fx | fy) == 0
    w00, w11 = np.where(exact, 32767, w00), np.where(exact, 1, w11)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < sh) & (xx >= 0) & (xx < sw)
        return np.where(ok[..., None], a[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)], 0).astype(np.int64)
    v = (tap(sy, sx) * w00[..., None] + tap(sy, sx + 1) * w01[..., None] + tap(sy + 1, sx) * w10[..., None] + tap(sy + 1, sx + 1) * w11[..., None] + (1 << 14)) >> 15
    out = np.clip(v, 0, 255).astype(np.uint8)
    return torch.from_numpy(out[:, :, 0] if squeeze else out)


def _patch_compose_cpu_(canvas, patch, mask, canvas2):
    """`patch_compose_u8_kernel` in NumPy: paste where every in-range tap of the 8x8 window anchored at (4, 4) is 255."""
    m = mask.numpy()
    m = (m[:, :, 0] if m.ndim == 3 else m) == 255
    h, w = m.shape
    pad = np.ones((h + 8, w + 8), dtype=bool)                                    # out-of-image taps are ignored = count as white
    pad[4:4 + h, 4:4 + w] = m
    keep = np.ones((h, w), dtype=bool)
    for ky in range(8):
        for kx in range(8):
            keep &= pad[ky:ky + h, kx:kx + w]
    sel = torch.from_numpy(keep)[:, :, None]
    canvas.copy_(torch.where(sel, patch, canvas))
    if canvas2 is not None:
        canvas2.copy_(torch.where(sel, patch, canvas2))
    return canvas


traffic_counter = None      # bench.py sets this to a dict(bytes=0, launches=0): algorithmic bytes (source image + destination, each once per job) of the launches


def warp_perspective_batch(jobs):
    """jobs: list of (src uint8 [H,W,C] GPU tensor, forward 3x3 matrix as cv2.warpPerspective takes it, (w, h)) -> list of outputs."""
    if not jobs:
        return []
    if traffic_counter is not None:
        traffic_counter['bytes'] += sum(int(src.numel()) + int(w) * int(h) * (int(src.shape[2]) if src.ndim == 3 else 1) for src, _, (w, h) in jobs)
        traffic_counter['launches'] += 1
    dev = jobs[0][0].device
    if dev.type != 'cuda':
        return [_warp_perspective_cpu(src, m, wh) for src, m, wh in jobs]
    lib = _init().lib
    table = (WarpJob * len(jobs))()
    outs, keep = [], []
    maxpix = 0
    for k, (src, m, (w, h)) in enumerate(jobs):
        src = src.contiguous()
        c = src.shape[2] if src.ndim == 3 else 1
        dst = torch.empty([h, w, c] if src.ndim == 3 else [h, w], dtype=torch.uint8, device=dev)
        minv = invert3x3(m).reshape(9)
        j = table[k]
        j.src, j.dst = src.data_ptr(), dst.data_ptr()
        j.src_h, j.src_w, j.dst_h, j.dst_w, j.channels, j.block_w = int(src.shape[0]), int(src.shape[1]), int(h), int(w), int(c), _block_width(h, w)
        for i in range(9):
            j.minv[i] = float(minv[i])
        outs.append(dst)
        keep.append(src)
        maxpix = max(maxpix, h * w)
    # the job table in device memory: through a pinned staging buffer, asynchronously -- a pageable upload is a host sync per launch, and the routing of a
    # batch is ~45 launches per sample that should run behind the previous batch's generator pass, not in lock step with the host
    raw = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).pin_memory().to(dev, non_blocking=True)
    with torch.cuda.device(dev):
        st = lib.pg_warp_perspective_u8(raw.data_ptr(), len(jobs), int(maxpix), nat.stream_of(outs[0]))
    nat.check(st, 'pg_warp_perspective_u8')
    return outs


def patch_compose_(canvas, patch, mask, canvas2=None):
    """canvas[p] = patch[p] where erode8x8(mask[..., 0]) == 255 (also into canvas2), in place."""
    if canvas.device.type != 'cuda':
        return _patch_compose_cpu_(canvas, patch, mask, canvas2)
    if traffic_counter is not None:     # patch + mask read, the canvas (and its copy) written where the eroded mask is set: counted whole
        traffic_counter['bytes'] += int(patch.numel()) + int(mask.numel()) + int(canvas.numel()) * (2 if canvas2 is not None else 1)
        traffic_counter['launches'] += 1
    lib = _init().lib
    h, w = canvas.shape[:2]
    mc = mask.shape[2] if mask.ndim == 3 else 1
    with torch.cuda.device(canvas.device):
        st = lib.pg_patch_compose_u8(patch.data_ptr(), mask.data_ptr(), canvas.data_ptr(), canvas2.data_ptr() if canvas2 is not None else None,
                                     int(h), int(w), int(mc), nat.stream_of(canvas))
    nat.check(st, 'pg_patch_compose_u8')
    return canvas


def normalize(upper_img, lower_img, upper_clothes_mask, lower_clothes_mask, sleeve_mask, clothes_keypoints, person_keypoints, box_factor, device='cuda'):
    """The reference's ``normalize`` (dataset.py:2555-2700): returns (img [h,w,30], img_lower [h,w,15], denorm_upper_img,
    denorm_upper_img_wo_sleeve, denorm_lower_img) as uint8 GPU tensors."""
    dev = torch.device(device)
    up, lo = _gpu_u8(upper_img, dev), _gpu_u8(lower_img, dev)
    um, lm = _gpu_u8(upper_clothes_mask, dev), _gpu_u8(lower_clothes_mask, dev)
    o_h, o_w = int(up.shape[0]), int(up.shape[1])
    h, w = o_h // 2 ** box_factor, o_w // 2 ** box_factor
    wh = np.expand_dims(np.array([w, h]), 0)
    if sleeve_mask is not None:
        sl = _gpu_u8(sleeve_mask, dev)
        src_sleeve, src_body = (up * sl, um * sl), (up * (1 - sl), um * (1 - sl))
    else:
        src_sleeve = src_body = (up, um)

    crops = []
    for ii, bpart in enumerate(BPARTS):
        ar = 0.5 if ii < 6 else 0.4
        crops.append((get_crop(clothes_keypoints, bpart, wh, o_w, o_h, ar), get_crop(person_keypoints, bpart, wh, o_w, o_h, ar)))

    # stage 1: image -> patch
    jobs, slot = [], {}
    for ii, ((c_m, _), (p_m, _)) in enumerate(crops):
        if c_m is not None:
            img_s, mask_s = src_sleeve if ii in SLEEVE_PARTS else src_body
            slot[('img', ii)], slot[('mask', ii)] = len(jobs), len(jobs) + 1
            jobs += [(img_s, c_m, (w, h)), (mask_s, c_m, (w, h))]
        if (ii == 0 or ii >= 6) and p_m is not None:
            slot[('img_lower', ii)], slot[('mask_lower', ii)] = len(jobs), len(jobs) + 1
            jobs += [(lo, p_m, (w, h)), (lm, p_m, (w, h))]
    out1 = warp_perspective_batch(jobs)
    zeros = lambda: torch.zeros([h, w, 3], dtype=torch.uint8, device=dev)
    get = lambda kind, ii: out1[slot[(kind, ii)]] if (kind, ii) in slot else zeros()
    part_imgs = [get('img', ii) for ii in range(10)]
    part_masks = [get('mask', ii) for ii in range(10)]
    lower_ids = [0, 6, 7, 8, 9]
    part_imgs_lower = [get('img_lower', ii) for ii in lower_ids]
    part_masks_lower = [get('mask_lower', ii) for ii in lower_ids]

    # stage 2: patch -> canvas
    jobs, slot2 = [], {}
    for ii, ((c_m, _), (p_m, p_inv)) in enumerate(crops):
        if c_m is not None and p_inv is not None:
            slot2[('up', ii)] = len(jobs)
            jobs += [(part_imgs[ii], p_inv, (o_w, o_h)), (part_masks[ii], p_inv, (o_w, o_h))]
        if (ii == 0 or ii >= 6) and p_m is not None and p_inv is not None:
            k = lower_ids.index(ii)
            slot2[('lo', ii)] = len(jobs)
            jobs += [(part_imgs_lower[k], p_inv, (o_w, o_h)), (part_masks_lower[k], p_inv, (o_w, o_h))]
    out2 = warp_perspective_batch(jobs)

    # stage 3: erode + paste, in part order
    denorm_upper = torch.zeros_like(up)
    denorm_upper_wo_sleeve = torch.zeros_like(up)
    denorm_lower = torch.zeros_like(up)
    for ii in range(10):
        if ('up', ii) in slot2:
            k = slot2[('up', ii)]
            patch_compose_(denorm_upper, out2[k], out2[k + 1], None if ii in SLEEVE_PARTS else denorm_upper_wo_sleeve)
        if ('lo', ii) in slot2:
            k = slot2[('lo', ii)]
            patch_compose_(denorm_lower, out2[k], out2[k + 1])

    # lower-garment parts give way to the upper garment; a missing sleeve is mirrored from the other side (:2655-2693)
    for lower_i, upper_i in ((0, 0), (1, 6), (3, 8)):
        keep = 1 - (part_masks[upper_i].to(torch.int32).sum(dim=2, keepdim=True) > 0).to(torch.uint8)
        part_imgs_lower[lower_i] = part_imgs_lower[lower_i] * keep
        part_masks_lower[lower_i] = part_masks_lower[lower_i] * keep
    # (the decisions stay on the device -- `bool(mask.any())` was ten host syncs per sample: torch.where on 0-dim conditions picks the same tensors)
    has = [(m != 0).any() for m in part_masks]
    flip = lambda t: torch.flip(t, dims=[1])
    c24, c42 = (~has[2]) & has[4], (~has[4]) & has[2]        # `if not has[2] and has[4] ... elif not has[4] and has[2]`: the two cannot both hold
    i2, m2, i4, m4 = part_imgs[2], part_masks[2], part_imgs[4], part_masks[4]
    part_imgs[2], part_masks[2] = torch.where(c24, flip(i4), i2), torch.where(c24, flip(m4), m2)
    part_imgs[4], part_masks[4] = torch.where(c42, flip(i2), i4), torch.where(c42, flip(m2), m4)
    c35, c53 = (~has[3]) & has[5], (~has[5]) & has[3]
    i3, m3, i5, m5 = part_imgs[3], part_masks[3], part_imgs[5], part_masks[5]
    part_imgs[3], part_masks[3] = torch.where(c35, flip(i3), i3), torch.where(c35, flip(m5), m3)     # as written in the reference: the image mirrored is part 3's own
    part_imgs[5], part_masks[5] = torch.where(c53, flip(i5), i5), torch.where(c53, flip(m3), m5)

    return torch.cat(part_imgs, dim=2), torch.cat(part_imgs_lower, dim=2), denorm_upper, denorm_upper_wo_sleeve, denorm_lower


def _upload_table(arr, dev):
    """A job table (NumPy structured array) in device memory: through pinned memory, asynchronously."""
    return torch.from_numpy(arr.view(np.uint8).reshape(-1)).pin_memory().to(dev, non_blocking=True)


def normalize_batch(samples, box_factor, device='cuda'):
    """`normalize` of a whole batch with THREE native launches instead of ~17 per sample: one image -> patch warp launch and one
    patch -> canvas warp launch over every job of every sample, one ordered erode-and-paste launch over every canvas (pg_patch_compose_ordered_u8), the
    homographies of all samples from one batched solve and the bookkeeping around them as a handful of batched tensor operations.

    samples: list of (upper_img, lower_img, upper_clothes_mask, lower_clothes_mask, sleeve_mask | None, clothes_keypoints, person_keypoints), images uint8
    [H, W, 3] (all the same size).  Returns the five results of `normalize`, stacked: img [N, h, w, 30], img_lower [N, h, w, 15], denorm_upper_img,
    denorm_upper_img_wo_sleeve, denorm_lower_img [N, H, W, 3] -- bit-identical to the per-sample calls (tests/test_patch_routing.py)."""
    dev = torch.device(device)
    n = len(samples)
    if dev.type != 'cuda':
        outs = [normalize(*s, box_factor, device=device) for s in samples]
        return tuple(torch.stack([o[i] for o in outs]) for i in range(5))
    lib = _init().lib
    ups = torch.stack([_gpu_u8(s[0], dev) for s in samples])
    los = torch.stack([_gpu_u8(s[1], dev) for s in samples])
    ums = torch.stack([_gpu_u8(s[2], dev) for s in samples])
    lms = torch.stack([_gpu_u8(s[3], dev) for s in samples])
    o_h, o_w = int(ups.shape[1]), int(ups.shape[2])
    h, w = o_h // 2 ** box_factor, o_w // 2 ** box_factor
    wh = np.expand_dims(np.array([w, h]), 0)
    with_sleeve = [i for i, s in enumerate(samples) if s[4] is not None]
    if with_sleeve:
        sl = torch.stack([_gpu_u8(samples[i][4], dev) for i in with_sleeve])
        u_s, m_s = ups[with_sleeve], ums[with_sleeve]
        sleeve_img, sleeve_msk, body_img, body_msk = u_s * sl, m_s * sl, u_s * (1 - sl), m_s * (1 - sl)
    pos = {i: k for k, i in enumerate(with_sleeve)}

    def sources(i, ii):                                      # (image, mask) the clothes-side warps of part ii read
        if i in pos:
            return (sleeve_img[pos[i]], sleeve_msk[pos[i]]) if ii in SLEEVE_PARTS else (body_img[pos[i]], body_msk[pos[i]])
        return ups[i], ums[i]

    # ---- host geometry: the quadrilaterals per sample and part, then every homography in one batched solve
    part_dst = np.float32(wh * np.float32([[0.0, 0.0], [0.0, 1.0], [1.0, 1.0], [1.0, 0.0]]))
    cq, pq = {}, {}
    for i, s in enumerate(samples):
        for ii, bpart in enumerate(BPARTS):
            ar = 0.5 if ii < 6 else 0.4
            for store, kp in ((cq, s[5]), (pq, s[6])):
                q = get_crop(kp, bpart, wh, o_w, o_h, ar, _quad_only=True)
                if not isinstance(q, tuple):
                    store[(i, ii)] = q
    ckeys, pkeys = list(cq), list(pq)
    src = [cq[k] for k in ckeys] + [pq[k] for k in pkeys] + [part_dst] * len(pkeys)
    dst = [part_dst] * (len(ckeys) + len(pkeys)) + [pq[k] for k in pkeys]
    mats = perspective_transforms(np.stack(src), np.stack(dst)) if src else np.zeros((0, 3, 3))
    invs = invert3x3_batch(mats).reshape(-1, 9)              # what the warp kernel takes: the inverse of the matrix cv2.warpPerspective is given
    c_m = {k: invs[j] for j, k in enumerate(ckeys)}
    p_m = {k: invs[len(ckeys) + j] for j, k in enumerate(pkeys)}
    p_inv = {k: invs[len(ckeys) + len(pkeys) + j] for j, k in enumerate(pkeys)}

    # ---- stage 1: image -> patch.  Slots of P1: part images 0..9, part masks 10..19, lower images 20..24, lower masks 25..29 (missing parts stay zero)
    lower_ids = [0, 6, 7, 8, 9]
    P1 = torch.zeros([n, 30, h, w, 3], dtype=torch.uint8, device=dev)
    P2 = torch.empty([n, 30, o_h, o_w, 3], dtype=torch.uint8, device=dev)
    p1, p2 = P1.data_ptr(), P2.data_ptr()
    s1, s2 = h * w * 3, o_h * o_w * 3
    j1, j2 = [], []
    for i in range(n):
        for ii in range(10):
            if (i, ii) in c_m:
                img_s, msk_s = sources(i, ii)
                j1.append((img_s.data_ptr(), p1 + (i * 30 + ii) * s1, c_m[(i, ii)]))
                j1.append((msk_s.data_ptr(), p1 + (i * 30 + 10 + ii) * s1, c_m[(i, ii)]))
                if (i, ii) in p_inv:
                    j2.append((p1 + (i * 30 + ii) * s1, p2 + (i * 30 + ii) * s2, p_inv[(i, ii)]))
                    j2.append((p1 + (i * 30 + 10 + ii) * s1, p2 + (i * 30 + 10 + ii) * s2, p_inv[(i, ii)]))
            if (ii == 0 or ii >= 6) and (i, ii) in p_m:
                k = lower_ids.index(ii)
                j1.append((los[i].data_ptr(), p1 + (i * 30 + 20 + k) * s1, p_m[(i, ii)]))
                j1.append((lms[i].data_ptr(), p1 + (i * 30 + 25 + k) * s1, p_m[(i, ii)]))
                j2.append((p1 + (i * 30 + 20 + k) * s1, p2 + (i * 30 + 20 + k) * s2, p_inv[(i, ii)]))
                j2.append((p1 + (i * 30 + 25 + k) * s1, p2 + (i * 30 + 25 + k) * s2, p_inv[(i, ii)]))

    def warp_table(jobs, sh, sw, dh, dw):
        t = np.zeros(len(jobs), dtype=_WARP_DT)
        t['src'], t['dst'] = [j[0] for j in jobs], [j[1] for j in jobs]
        t['src_h'], t['src_w'], t['dst_h'], t['dst_w'], t['channels'], t['block_w'] = sh, sw, dh, dw, 3, _block_width(dh, dw)
        t['minv'] = np.stack([j[2] for j in jobs])
        return t
    stream = nat.stream_of(P1)
    keep = []
    with torch.cuda.device(dev):
        for jobs, geo in ((j1, (o_h, o_w, h, w)), (j2, (h, w, o_h, o_w))):
            if not jobs:
                continue
            tab = _upload_table(warp_table(jobs, *geo), dev)
            keep.append(tab)
            nat.check(lib.pg_warp_perspective_u8(tab.data_ptr(), len(jobs), geo[2] * geo[3], stream), 'pg_warp_perspective_u8')
            if traffic_counter is not None:
                traffic_counter['bytes'] += len(jobs) * 3 * (geo[0] * geo[1] + geo[2] * geo[3])
                traffic_counter['launches'] += 1

        # ---- stage 3: every canvas of the batch in one ordered erode-and-paste launch
        D = torch.empty([3, n, o_h, o_w, 3], dtype=torch.uint8, device=dev)       # denorm_upper, denorm_upper_wo_sleeve, denorm_lower
        d0 = D.data_ptr()
        ct = np.zeros(2 * n, dtype=_COMPOSE_DT)
        for i in range(n):
            up_parts = [ii for ii in range(10) if (i, ii) in c_m and (i, ii) in p_inv]
            lo_parts = [ii for ii in lower_ids if (i, ii) in p_m]
            a, b = ct[2 * i], ct[2 * i + 1]
            a['canvas'], a['canvas2'], a['nparts'] = d0 + i * s2, d0 + (n + i) * s2, len(up_parts)
            for k, ii in enumerate(up_parts):
                a['patch'][k], a['mask'][k], a['to_canvas2'][k] = p2 + (i * 30 + ii) * s2, p2 + (i * 30 + 10 + ii) * s2, int(ii not in SLEEVE_PARTS)
            b['canvas'], b['canvas2'], b['nparts'] = d0 + (2 * n + i) * s2, 0, len(lo_parts)
            for k, ii in enumerate(lo_parts):
                kk = lower_ids.index(ii)
                b['patch'][k], b['mask'][k] = p2 + (i * 30 + 20 + kk) * s2, p2 + (i * 30 + 25 + kk) * s2
        tab = _upload_table(ct, dev)
        keep.append(tab)
        nat.check(lib.pg_patch_compose_ordered_u8(tab.data_ptr(), 2 * n, o_h, o_w, 3, stream), 'pg_patch_compose_ordered_u8')
        if traffic_counter is not None:
            traffic_counter['bytes'] += int(sum(int(c_['nparts']) for c_ in ct)) * 2 * s2 + 3 * n * s2
            traffic_counter['launches'] += 1

    # ---- the rest of `normalize` on the batched buffers: lower-garment parts give way to the upper garment; a missing sleeve is mirrored from the other side
    imgs, masks, imgs_lo, masks_lo = P1[:, 0:10], P1[:, 10:20], P1[:, 20:25], P1[:, 25:30]
    gone = (masks[:, [0, 6, 8]].to(torch.int32).sum(dim=4, keepdim=True) > 0)            # [N, 3, h, w, 1]
    keep_lo = (~gone).to(torch.uint8)
    idx = torch.tensor([0, 1, 3], device=dev)
    imgs_lo = imgs_lo.index_copy(1, idx, imgs_lo[:, [0, 1, 3]] * keep_lo)
    masks_lo = masks_lo.index_copy(1, idx, masks_lo[:, [0, 1, 3]] * keep_lo)
    has = (masks != 0).flatten(2).any(dim=2)                                              # [N, 10]
    col = lambda t: t[:, None, None, None]
    flip = lambda t: torch.flip(t, dims=[2])                                              # [N, h, w, 3]: the width axis
    c24, c42 = col(~has[:, 2] & has[:, 4]), col(~has[:, 4] & has[:, 2])
    c35, c53 = col(~has[:, 3] & has[:, 5]), col(~has[:, 5] & has[:, 3])
    i2, m2, i3, m3, i4, m4, i5, m5 = imgs[:, 2], masks[:, 2], imgs[:, 3], masks[:, 3], imgs[:, 4], masks[:, 4], imgs[:, 5], masks[:, 5]
    new_imgs = torch.stack([imgs[:, 0], imgs[:, 1], torch.where(c24, flip(i4), i2), torch.where(c35, flip(i3), i3), torch.where(c42, flip(i2), i4),
                            torch.where(c53, flip(i5), i5), imgs[:, 6], imgs[:, 7], imgs[:, 8], imgs[:, 9]], dim=1)      # (parts 3 / 5: the image mirrored is the part's own, as in the reference)
    img = new_imgs.permute(0, 2, 3, 1, 4).reshape(n, h, w, 30)
    img_lower = imgs_lo.permute(0, 2, 3, 1, 4).reshape(n, h, w, 15)
    return img, img_lower, D[0], D[1], D[2]
