"""CPU restatement of the patch-routing warp of the reference's data loader.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference does this with OpenCV (``cv2.getPerspectiveTransform``, ``cv2.warpPerspective``, ``cv2.erode``,
``cv2.flip``; training/dataset.py:2373-2700).  ``cv2`` is not installed in the build container and is no part of the reference
tree (an un-vendored third-party dependency: opencv-python, version not pinned by the reference), so neither the reference's
``normalize`` nor OpenCV itself can be run here, and the reference holds no fixtures for this step.  What follows restates
OpenCV's published algorithms (modules/imgproc/src/imgwarp.cpp: ``warpPerspective`` -> ``WarpPerspectiveInvoker`` ->
``remap`` / ``remapBilinear`` with ``FixedPtCast<int, uchar, INTER_REMAP_COEF_BITS>``; ``initInterTab2D``; morph.cpp) for
8-bit images, INTER_LINEAR, BORDER_CONSTANT(0):

* the forward matrix is inverted in double (3x3 cofactor formula of ``cv::invert``);
* destination pixels are walked in blocks (bw x bh from BLOCK_SZ = 32); per row of a block
  ``X0 = M0*x + M1*y + M2`` (block origin x), then per pixel ``W = 32 / (W0 + M6*x1)``, ``fX = (X0 + M0*x1) * W`` clipped to the
  int range, ``X = cvRound(fX)`` (round half to even); integer part ``X >> 5``, fraction ``X & 31`` (same for Y);
* bilinear weights are 15-bit integers ``(32-fx)(32-fy)*32`` ... (they sum to 2^15 exactly; the only table entry that
  needs OpenCV's correction is (0,0): [32767, 0, 0, 1]); result ``(sum w*p + 2^14) >> 15``; taps outside the image read 0;
* erode: minimum over the 8x8 window anchored at (4, 4), pixels outside the image ignored (morphologyDefaultBorderValue).

``get_crop`` / ``normalize`` follow training/dataset.py:2373-2542 and :2555-2700 line by line.
"""

import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
BLOCK_SZ = 32

ORDER = ['cnose', 'cneck', 'rshoulder', 'relbow', 'rwrist', 'lshoulder', 'lelbow', 'lwrist', 'rhip', 'rknee', 'rankle', 'lhip', 'lknee',
         'lankle', 'reye', 'leye', 'rear', 'lear']                                                   # dataset.py:2576-2578
BPARTS = [["rshoulder", "rhip", "lhip", "lshoulder"], ["lshoulder", "rshoulder", "cnose"], ["lshoulder", "lelbow"], ["lelbow", "lwrist"],
          ["rshoulder", "relbow"], ["relbow", "rwrist"], ["lhip", "lknee"], ["lknee", "lankle"], ["rhip", "rknee"], ["rknee", "rankle"]]   # :2564-2574


def get_perspective_transform(src, dst):
    """3x3 float64 M with M @ [x, y, 1] ~ [X, Y, 1] for the four point pairs (cv2.getPerspectiveTransform: the 8x8 system of
    imgwarp.cpp solved by LU with partial pivoting)."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    a = np.zeros((8, 8), np.float64)
    b = np.zeros(8, np.float64)
    for i in range(4):
        x, y = src[i]
        X, Y = dst[i]
        a[i] = [x, y, 1, 0, 0, 0, -x * X, -y * X]
        a[i + 4] = [0, 0, 0, x, y, 1, -x * Y, -y * Y]
        b[i], b[i + 4] = X, Y
    m = np.linalg.solve(a, b)
    return np.append(m, 1.0).reshape(3, 3)


def invert3x3(m):
    """cv::invert for a 3x3 double matrix: cofactors times 1/det."""
    m = np.asarray(m, np.float64)
    d = (m[0, 0] * (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) - m[0, 1] * (m[1, 0] * m[2, 2] - m[1, 2] * m[2, 0]) +
         m[0, 2] * (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]))
    if d == 0:
        return np.zeros((3, 3))
    d = 1.0 / d
    t = np.empty((3, 3), np.float64)
    t[0, 0] = (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) * d
    t[0, 1] = (m[0, 2] * m[2, 1] - m[0, 1] * m[2, 2]) * d
    t[0, 2] = (m[0, 1] * m[1, 2] - m[0, 2] * m[1, 1]) * d
    t[1, 0] = (m[1, 2] * m[2, 0] - m[1, 0] * m[2, 2]) * d
    t[1, 1] = (m[0, 0] * m[2, 2] - m[0, 2] * m[2, 0]) * d
    t[1, 2] = (m[0, 2] * m[1, 0] - m[0, 0] * m[1, 2]) * d
    t[2, 0] = (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]) * d
    t[2, 1] = (m[0, 1] * m[2, 0] - m[0, 0] * m[2, 1]) * d
    t[2, 2] = (m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]) * d
    return t


def block_width(dst_h, dst_w):
    bh0 = min(BLOCK_SZ // 2, dst_h)
    bw0 = min(BLOCK_SZ * BLOCK_SZ // bh0, dst_w)
    return bw0


def warp_coords(minv, dst_h, dst_w):
    """Fixed-point source coordinates (X, Y as int32 with 5 fractional bits) of every destination pixel."""
    m = np.asarray(minv, np.float64).reshape(9)
    bw = block_width(dst_h, dst_w)
    ys = np.arange(dst_h, dtype=np.float64)[:, None]
    xs = np.arange(dst_w)[None, :]
    xb = ((xs // bw) * bw).astype(np.float64)                # block origin
    x1 = (xs - (xs // bw) * bw).astype(np.float64)           # offset in the block
    X0 = m[0] * xb + m[1] * ys + m[2]
    Y0 = m[3] * xb + m[4] * ys + m[5]
    W0 = m[6] * xb + m[7] * ys + m[8]
    W = W0 + m[6] * x1
    with np.errstate(divide='ignore', invalid='ignore'):
        W = np.where(W != 0, INTER_TAB_SIZE / W, 0.0)
    lo, hi = float(np.iinfo(np.int32).min), float(np.iinfo(np.int32).max)
    fX = np.maximum(lo, np.minimum(hi, (X0 + m[0] * x1) * W))
    fY = np.maximum(lo, np.minimum(hi, (Y0 + m[3] * x1) * W))
    return np.rint(fX).astype(np.int64), np.rint(fY).astype(np.int64)


def warp_perspective_u8(src, m, dsize, inverse_map=False):
    """cv2.warpPerspective(src, M, (w, h), flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0) for uint8 HxWxC."""
    src = np.asarray(src)
    assert src.dtype == np.uint8
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    dst_w, dst_h = int(dsize[0]), int(dsize[1])
    minv = np.asarray(m, np.float64) if inverse_map else invert3x3(m)
    X, Y = warp_coords(minv, dst_h, dst_w)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)             # saturate_cast<short>
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    fx, fy = (X & (INTER_TAB_SIZE - 1)), (Y & (INTER_TAB_SIZE - 1))
    w00 = (32 - fx) * (32 - fy) * 32
    w01 = fx * (32 - fy) * 32
    w10 = (32 - fx) * fy * 32
    w11 = fx * fy * 32
    zero = (fx == 0) & (fy == 0)                            # initInterTab2D: 1.0 saturates to 32767, the correction lands on w11
    w00 = np.where(zero, 32767, w00)
    w11 = np.where(zero, 1, w11)
    H, Wd = src.shape[:2]

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < Wd)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, Wd - 1)].astype(np.int64)
        return np.where(ok[..., None], v, 0)
    acc = (tap(sy, sx) * w00[..., None] + tap(sy, sx + 1) * w01[..., None] + tap(sy + 1, sx) * w10[..., None] + tap(sy + 1, sx + 1) * w11[..., None])
    out = np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def erode_u8(img, ksize=8):
    """cv2.erode(img, np.ones((k, k), np.uint8), iterations=1): anchor (k//2, k//2), outside pixels ignored."""
    img = np.asarray(img)
    a = ksize // 2
    h, w = img.shape[:2]
    big = np.full((h + ksize, w + ksize) + img.shape[2:], 255, np.uint8)
    big[a:a + h, a:a + w] = img
    out = np.full_like(img, 255)
    for ky in range(ksize):
        for kx in range(ksize):
            out = np.minimum(out, big[ky:ky + h, kx:kx + w])
    return out


def valid_joints(joint):
    return bool((joint >= 0.1).all())                        # dataset.py:825-826


def get_crop(keypoints, bpart, wh, o_w, o_h, ar=1.0, order=ORDER):
    """dataset.py:2373-2542: the source quadrilateral of one body part and its two homographies (M: image -> patch, M_inv)."""
    joints = keypoints
    bpart = list(bpart)
    idx = [order.index(b) for b in bpart]
    part_src = np.float32(joints[idx][:, :2])
    if not valid_joints(joints[idx][:, 2]):                  # fall backs (:2378-2399)
        if bpart[0] == "lhip" and bpart[1] == "lknee":
            bpart = ["lhip"]
        elif bpart[0] == "rhip" and bpart[1] == "rknee":
            bpart = ["rhip"]
        elif bpart[0] == "lknee" and bpart[1] == 'lankle':
            bpart = ["lknee"]
        elif bpart[0] == "rknee" and bpart[1] == 'rankle':
            bpart = ["rknee"]
        elif bpart[0] == "lshoulder" and bpart[1] == "rshoulder" and bpart[2] == "cnose":
            bpart = ["lshoulder", "rshoulder", "rshoulder"]
        idx = [order.index(b) for b in bpart]
        part_src = np.float32(joints[idx][:, :2])
    if not valid_joints(joints[idx][:, 2]):
        return None, None

    if part_src.shape[0] == 1:                               # leg fallback from the torso length (:2404-2457)
        torso_idx = [order.index(b) for b in ["lhip", "rhip", "cneck"]]
        if not valid_joints(joints[torso_idx][:, 2]):
            return None, None
        a = part_src[0]
        invalid_label = 'lknee' if 'lhip' in bpart else 'rknee' if 'rhip' in bpart else 'lankle' if 'lknee' in bpart else 'rankle'
        invalid_joint = joints[order.index(invalid_label)]
        part_torso = np.float32(joints[torso_idx][:, :2])
        torso_length = (np.linalg.norm(part_torso[2] - part_torso[1]) + np.linalg.norm(part_torso[2] - part_torso[0])) / 2
        factor = 0.85 if 'hip' in bpart[0] else 0.80
        if invalid_joint[2] > 0:
            direction = (invalid_joint[0:2] - a) / np.linalg.norm(a - invalid_joint[0:2])
            b = a + torso_length * direction * factor
        else:
            b = np.float32([a[0], a[1] + torso_length * factor])
        part_src = np.float32([a, b])

    inside = lambda q: q[0] > 0 and q[1] > 0 and q[0] < o_w and q[1] < o_h
    if part_src.shape[0] == 4:                               # torso: widen hips by 1/4, shoulders by 1/5 (:2459-2481)
        hip_seg = (part_src[2] - part_src[1]) / 4
        hip_l_new, hip_r_new = part_src[1] - hip_seg, part_src[2] + hip_seg
        if inside(hip_l_new):
            part_src[1] = hip_l_new
        if inside(hip_r_new):
            part_src[2] = hip_r_new
        shoulder_seg = (part_src[3] - part_src[0]) / 5
        shoulder_l_new, shoulder_r_new = part_src[0] - shoulder_seg, part_src[3] + shoulder_seg
        if inside(shoulder_l_new):
            part_src[0] = shoulder_l_new
        if inside(shoulder_r_new):
            part_src[3] = shoulder_r_new
    elif part_src.shape[0] == 3:                             # neck patch above the shoulder line (:2482-2508)
        shoulder_seg = (part_src[0] - part_src[1]) / 5
        shoulder_l_new, shoulder_r_new = part_src[1] - shoulder_seg, part_src[0] + shoulder_seg
        if inside(shoulder_l_new):
            part_src[1] = shoulder_l_new
        if inside(shoulder_r_new):
            part_src[0] = shoulder_r_new
        segment = part_src[1] - part_src[0]
        normal = np.array([-segment[1], segment[0]])
        if normal[1] > 0.0:
            normal = -normal
        a, b, c, d = part_src[0] + normal, part_src[0], part_src[1], part_src[1] + normal
        part_height = (c[1] + b[1]) / 2 - (a[1] + d[1]) / 2
        a[1] += part_height / 2
        d[1] += part_height / 2
        part_src = np.float32([d, c, b, a])
    else:                                                    # limbs: a strip of aspect ratio `ar` around the bone (:2509-2537)
        assert part_src.shape[0] == 2
        segment = part_src[1] - part_src[0]
        normal = np.array([-segment[1], segment[0]])
        alpha = ar / 2.0
        a, b = part_src[0] + alpha * normal, part_src[0] - alpha * normal
        c, d = part_src[1] - alpha * normal, part_src[1] + alpha * normal
        if 'rhip' in bpart or 'rknee' in bpart:
            a, d = a + alpha * normal * 1.0, d + alpha * normal * 1.0
        if 'lhip' in bpart or 'lknee' in bpart:
            b, c = b - alpha * normal * 1.0, c - alpha * normal * 1.0
        if 'relbow' in bpart or 'rwrist' in bpart:
            a, d = a + alpha * normal * 0.45, d + alpha * normal * 0.45
            b, c = b - alpha * normal * 0.1, c - alpha * normal * 0.1
        if 'lelbow' in bpart or 'lwrist' in bpart:
            a, d = a + alpha * normal * 0.1, d + alpha * normal * 0.1
            b, c = b - alpha * normal * 0.45, c - alpha * normal * 0.45
        part_src = np.float32([a, d, c, b])

    dst = np.float32([[0.0, 0.0], [0.0, 1.0], [1.0, 1.0], [1.0, 0.0]])
    part_dst = np.float32(wh * dst)
    return get_perspective_transform(part_src, part_dst), get_perspective_transform(part_dst, part_src)


def normalize(upper_img, lower_img, upper_clothes_mask, lower_clothes_mask, sleeve_mask, clothes_keypoints, person_keypoints, box_factor):
    """dataset.py:2555-2700.  Images HxWx3 uint8 (masks: 0/255 or 0/1 as the caller prepares them); keypoints [18, 3]."""
    h, w = upper_img.shape[:2]
    o_h, o_w = h, w
    h, w = h // 2 ** box_factor, w // 2 ** box_factor
    wh = np.expand_dims(np.array([w, h]), 0)
    part_imgs, part_imgs_lower, part_clothes_masks, part_clothes_masks_lower = [], [], [], []
    denorm_upper_img = np.zeros_like(upper_img)
    denorm_upper_img_wo_sleeve = np.zeros_like(upper_img)
    denorm_lower_img = np.zeros_like(upper_img)
    warp = lambda img, m, size: warp_perspective_u8(img, m, size)

    for ii, bpart in enumerate(BPARTS):
        ar = 0.5 if ii < 6 else 0.4
        part_img = np.zeros((h, w, 3), np.uint8)
        part_img_lower = np.zeros((h, w, 3), np.uint8)
        part_clothes_mask = np.zeros((h, w, 3), np.uint8)
        part_clothes_mask_lower = np.zeros((h, w, 3), np.uint8)
        clothes_M, clothes_M_inv = get_crop(clothes_keypoints, bpart, wh, o_w, o_h, ar)
        person_M, person_M_inv = get_crop(person_keypoints, bpart, wh, o_w, o_h, ar)

        if clothes_M is not None:
            if sleeve_mask is not None:
                sel = sleeve_mask if ii in (2, 3, 4, 5) else (1 - sleeve_mask)
                part_img = warp(upper_img * sel, clothes_M, (w, h))
                part_clothes_mask = warp(upper_clothes_mask * sel, clothes_M, (w, h))
            else:
                part_img = warp(upper_img, clothes_M, (w, h))
                part_clothes_mask = warp(upper_clothes_mask, clothes_M, (w, h))
            if person_M_inv is not None:
                denorm_patch = warp(part_img, person_M_inv, (o_w, o_h))
                m = warp(part_clothes_mask, person_M_inv, (o_w, o_h))[..., 0:1]
                m = erode_u8(m[..., 0], 8)[..., np.newaxis]
                m = (m == 255).astype(np.uint8)
                denorm_upper_img = denorm_patch * m + denorm_upper_img * (1 - m)
                if ii not in (2, 3, 4, 5):
                    denorm_upper_img_wo_sleeve = denorm_patch * m + denorm_upper_img_wo_sleeve * (1 - m)

        if ii == 0 or ii >= 6:
            if person_M is not None:
                part_img_lower = warp(lower_img, person_M, (w, h))
                part_clothes_mask_lower = warp(lower_clothes_mask, person_M, (w, h))
                if person_M_inv is not None:
                    denorm_patch_lower = warp(part_img_lower, person_M_inv, (o_w, o_h))
                    m = warp(part_clothes_mask_lower, person_M_inv, (o_w, o_h))[..., 0:1]
                    m = erode_u8(m[..., 0], 8)[..., np.newaxis]
                    m = (m == 255).astype(np.uint8)
                    denorm_lower_img = denorm_patch_lower * m + denorm_lower_img * (1 - m)

        part_imgs.append(part_img)
        part_clothes_masks.append(part_clothes_mask)
        if ii == 0 or ii >= 6:
            part_imgs_lower.append(part_img_lower)
            part_clothes_masks_lower.append(part_clothes_mask_lower)

    any_mask = lambda m: (np.sum(m, axis=2, keepdims=True) > 0).astype(np.uint8)
    for lower_i, upper_i in ((0, 0), (1, 6), (3, 8)):          # lower-garment parts give way to the upper garment (:2655-2664)
        keep = 1 - any_mask(part_clothes_masks[upper_i])
        part_imgs_lower[lower_i] = part_imgs_lower[lower_i] * keep
        part_clothes_masks_lower[lower_i] = part_clothes_masks_lower[lower_i] * keep

    # a missing sleeve is mirrored from the other side (:2666-2693; the bottom-sleeve branches read part_imgs[3] / [5] exactly as written there)
    if np.sum(part_clothes_masks[2]) == 0 and np.sum(part_clothes_masks[4]) > 0:
        part_imgs[2], part_clothes_masks[2] = part_imgs[4][:, ::-1], part_clothes_masks[4][:, ::-1]
    elif np.sum(part_clothes_masks[4]) == 0 and np.sum(part_clothes_masks[2]) > 0:
        part_imgs[4], part_clothes_masks[4] = part_imgs[2][:, ::-1], part_clothes_masks[2][:, ::-1]
    if np.sum(part_clothes_masks[3]) == 0 and np.sum(part_clothes_masks[5]) > 0:
        part_imgs[3], part_clothes_masks[3] = part_imgs[3][:, ::-1], part_clothes_masks[5][:, ::-1]
    elif np.sum(part_clothes_masks[5]) == 0 and np.sum(part_clothes_masks[3]) > 0:
        part_imgs[5], part_clothes_masks[5] = part_imgs[5][:, ::-1], part_clothes_masks[3][:, ::-1]

    img = np.concatenate(part_imgs, axis=2)
    img_lower = np.concatenate(part_imgs_lower, axis=2)
    return img, img_lower, denorm_upper_img, denorm_upper_img_wo_sleeve, denorm_lower_img
