"""Test-time loader of try-on pairs (BASELINE config 1; reference training/dataset.py:1952-2726,
``UvitonDatasetFull_512_test_upper``): reads the reference's file formats and produces the 16-tuple its ``__getitem__``
returns (:2702-2726) -- uint8 CHW arrays

    image[3,512,512] clothes[3,512,512] pose[3,512,512] clothes_pose[3,512,512] norm_img[30,128,128] norm_img_lower[15,128,128]
    denorm_upper_img[3,512,512] denorm_lower_img[3,512,512] denorm_upper_mask[1,512,512] denorm_lower_mask[1,512,512]
    retain_mask[1,512,512] skin_average[3,512,512] lower_label_map[1,512,512] lower_clothes_upper_bound[1,512,512]
    person_name clothes_name

Directory layout (test.py:109-116): ``image/<name>.jpg`` (RGB JPEG, 320x512), ``parsing/<name>.png`` (L mode, LIP labels
0-19), ``garment_parsing/<name>.png`` (labels in channel 0), ``keypoints/<name>_keypoints.json`` (OpenPose-18:
``people[0].pose_keypoints_2d``, 54 floats) and a pairs file with ``<clothes_name> <person_name>`` per line.

What is restated and what is not: the label algebra (garment-class resolution, retain / skin / bound maps) follows the
reference statement by statement; the 10-part patch routing runs on this package's kernels (training.patch_routing: HIP on a
GPU, the same arithmetic in NumPy on the CPU).  The reference rasterises the pose map and the palm masks with OpenCV, scikit-image
and pycocotools, none of which exist in this image; `_Raster` below draws the same primitives (thick segments, discs, convex
quadrilaterals, square dilation) with its own pixel-coverage rules, so those two maps are NOT pinned bit for bit against the
reference's libraries (shapes, dtypes, value sets and topology are; DESIGN.md says so).
"""

import json
import os

import numpy as np
import torch

from . import patch_routing

try:
    import PIL.Image
except ImportError:                                   # pragma: no cover
    PIL = None

KPT_COLORS = [[255, 0, 0], [255, 85, 0], [255, 170, 0], [255, 255, 0], [170, 255, 0], [85, 255, 0], [0, 255, 0], [0, 255, 85], [0, 255, 170],
              [0, 255, 255], [0, 170, 255], [0, 85, 255], [0, 0, 255], [85, 0, 255], [170, 0, 255], [255, 0, 255], [255, 0, 170], [255, 0, 85], [255, 0, 0]]
LIMBS = [[2, 3], [2, 6], [3, 4], [4, 5], [6, 7], [7, 8], [2, 9], [9, 10], [10, 11], [2, 12], [12, 13], [13, 14], [2, 1], [1, 15], [15, 17],
         [1, 16], [16, 18], [3, 17], [6, 18]]
SIDE = 512


class _Raster:
    """Minimal rasteriser for the loader's drawings (own coverage rules; see the module docstring)."""

    @staticmethod
    def grid(h, w):
        return np.mgrid[0:h, 0:w]

    @staticmethod
    def segment(canvas, p0, p1, colour, thickness):
        """Pixels within thickness/2 of the segment p0-p1 (x, y)."""
        ys, xs = _Raster.grid(*canvas.shape[:2])
        d = np.array([p1[0] - p0[0], p1[1] - p0[1]], dtype=np.float64)
        ln = float(d @ d)
        t = np.zeros(xs.shape) if ln == 0 else np.clip(((xs - p0[0]) * d[0] + (ys - p0[1]) * d[1]) / ln, 0.0, 1.0)
        dist2 = (xs - (p0[0] + t * d[0])) ** 2 + (ys - (p0[1] + t * d[1])) ** 2
        canvas[dist2 <= (thickness / 2.0) ** 2] = colour

    @staticmethod
    def disc(canvas, centre, radius, colour):
        ys, xs = _Raster.grid(*canvas.shape[:2])
        canvas[(xs - centre[0]) ** 2 + (ys - centre[1]) ** 2 < radius ** 2] = colour

    @staticmethod
    def quad(h, w, pts):
        """Filled convex quadrilateral (4 x (x, y), in order): pixels on the same side of all four edges."""
        ys, xs = _Raster.grid(h, w)
        pts = np.asarray(pts, dtype=np.float64)
        sign = None
        inside = np.ones((h, w), dtype=bool)
        for i in range(4):
            a, b = pts[i], pts[(i + 1) % 4]
            cross = (b[0] - a[0]) * (ys - a[1]) - (b[1] - a[1]) * (xs - a[0])
            if sign is None:
                c = pts[(i + 2) % 4]
                sign = np.sign((b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])) or 1.0
            inside &= cross * sign >= 0
        return inside

    @staticmethod
    def dilate(mask, k):
        """Square k x k dilation (anchor at the centre, as cv2.dilate with a ones kernel)."""
        h, w = mask.shape
        lo, hi = k // 2, k - 1 - k // 2
        pad = np.zeros((h + k - 1, w + k - 1), dtype=bool)
        pad[lo:lo + h, lo:lo + w] = mask
        out = np.zeros((h, w), dtype=bool)
        for dy in range(k):
            rows = pad[dy:dy + h]
            for dx in range(k):
                out |= rows[:, dx:dx + w]
        return out


def _erode_white(mask_u8, k=8):
    """(cv2.erode(mask, ones(k, k)) == 255)[..., 0:1] with the conventions of this package's paste kernel: window anchored at
    k/2, out-of-image taps ignored."""
    m = mask_u8[:, :, 0] == 255
    h, w = m.shape
    pad = np.ones((h + k, w + k), dtype=bool)
    pad[k // 2:k // 2 + h, k // 2:k // 2 + w] = m
    out = np.ones((h, w), dtype=bool)
    for dy in range(k):
        for dx in range(k):
            out &= pad[dy:dy + h, dx:dx + w]
    return out[:, :, None].astype(np.uint8)


def _bbox(mask):
    """[xmin, ymin, xmax, ymax] of the non-zero pixels, or None (dataset.py:999-1008)."""
    ys, xs = np.nonzero(mask[..., 0] >= 0.5) if mask.ndim == 3 else np.nonzero(mask >= 0.5)
    if ys.size == 0:
        return None
    return [int(xs.min()), int(ys.min()), int(xs.max()), int(ys.max())]


def _pad_square(a, fill):
    """Centre a HxW[xC] array horizontally in a HxH frame (dataset.py:2038-2040: 320 -> 512 columns)."""
    h, w = a.shape[:2]
    left = (h - w) // 2
    widths = ((0, 0), (left, h - w - left)) + ((0, 0),) * (a.ndim - 2)
    return np.pad(a, widths, 'constant', constant_values=fill), left


def _garment_classes(parsing):
    """tops / dresses / pants / skirt masks with the reference's tie-breaking (dataset.py:2083-2111): pants and skirt are merged
    into whichever is larger; a dress is attributed to tops, to the lower garment, or swallows both, by area."""
    is_ = lambda *labels: np.isin(parsing, labels).astype(np.uint8)
    tops, dresses, pants, skirt = is_(5, 7), is_(6), is_(9), is_(12)
    if pants.sum() > skirt.sum():
        pants, skirt = pants + skirt, skirt * 0
    else:
        skirt, pants = skirt + pants, pants * 0
    if dresses.sum() > 0:
        if pants.sum() > 0:
            tops, dresses = tops + dresses, dresses * 0
        elif dresses.sum() > tops.sum() + skirt.sum():
            dresses, tops, skirt = dresses + tops + skirt, tops * 0, skirt * 0
        else:
            if tops.sum() > skirt.sum():
                skirt = skirt + dresses
            else:
                tops = tops + dresses
            dresses = dresses * 0
    return tops, dresses, pants, skirt


class TryOnTestSet(torch.utils.data.Dataset):
    """``UvitonDatasetFull_512_test_upper`` of the reference: transfer the UPPER garment of `clothes_name` onto `person_name`."""

    def __init__(self, path, test_txt='test_pairs.txt', use_sleeve_mask=False, device='cpu'):
        if PIL is None:
            raise ImportError('TryOnTestSet needs Pillow')
        self.path, self.use_sleeve_mask, self.device = path, use_sleeve_mask, device
        self.pairs = []
        with open(os.path.join(path, test_txt)) as f:
            for line in f:
                if line.strip():
                    clothes_name, person_name = line.split()
                    self.pairs.append((clothes_name, person_name))
        if not self.pairs:
            raise IOError('no pairs listed in ' + test_txt)

    def __len__(self):
        return len(self.pairs)

    # ------------------------------------------------------------------ file readers
    def _image(self, name):
        return np.array(PIL.Image.open(os.path.join(self.path, 'image', name)).convert('RGB'))

    def _labels(self, folder, name):
        a = np.array(PIL.Image.open(os.path.join(self.path, folder, os.path.splitext(name)[0] + '.png')))
        return (a if a.ndim == 2 else a[..., 0])[..., None]           # channel 0, as cv2.imread(...)[..., 0:1] of a grey / label image

    def _keypoints(self, name):
        with open(os.path.join(self.path, 'keypoints', os.path.splitext(name)[0] + '_keypoints.json')) as f:
            people = json.load(f)['people']
        if not people:
            return np.zeros((18, 3))
        return np.array(people[0]['pose_keypoints_2d'], dtype=np.float64).reshape(-1, 3)

    # ------------------------------------------------------------------ drawings
    @staticmethod
    def pose_map(kp, size):
        """Coloured skeleton (dataset.py:779-813): limbs as 5-pixel segments, joints as radius-5 discs; leg joints too close to
        the frame are demoted to confidence 0.01 (the side effect the reference's drawing has on the keypoints)."""
        h, w = size
        canvas = np.zeros((h, w, 3), dtype=np.uint8)
        for i, (a, b) in enumerate(LIMBS):
            pa, pb = kp[a - 1], kp[b - 1]
            if pa[2] < 0.05 or pb[2] < 0.05:
                continue
            _Raster.segment(canvas, (int(pa[0]), int(pa[1])), (int(pb[0]), int(pb[1])), KPT_COLORS[i], 5)
        for i in range(len(kp)):
            if kp[i][2] < 0.05:
                continue
            if i in (9, 10, 12, 13) and (kp[i][0] <= 0 or kp[i][1] <= 0 or kp[i][0] >= w - 50 or kp[i][1] >= h - 50):
                kp[i][2] = 0.01
                continue
            _Raster.disc(canvas, (int(kp[i][0]), int(kp[i][1])), 5, KPT_COLORS[i])
        return canvas, kp

    @staticmethod
    def _limb_band(a, b, c, d):
        """Quadrilateral around the limb (a,b)-(c,d), a quarter of its length wide on each side (dataset.py:2250-2275)."""
        ox, oy = (b - d) / 4.0, (c - a) / 4.0
        return [(a + ox, b + oy), (a - ox, b - oy), (c - ox, d - oy), (c + ox, d + oy)]

    def _arm_masks(self, joints):
        (sx, sy, sc), (ex, ey, ec), (wx, wy, wc) = joints
        upper = np.ones((SIDE, SIDE), dtype=bool)
        lower = np.ones((SIDE, SIDE), dtype=bool)
        if sc > 0.1 and ec > 0.1:
            upper = _Raster.dilate(_Raster.quad(SIDE, SIDE, self._limb_band(sx, sy, ex, ey)), 35)
        if ec > 0.1 and wc > 0.1:
            lower = _Raster.dilate(_Raster.quad(SIDE, SIDE, self._limb_band(ex, ey, wx, wy)), 28)
        return upper, lower

    def palm_mask(self, kp, parsing):
        """Hand label minus the upper-arm and fore-arm bands = the palms (dataset.py:753-777)."""
        out = np.zeros((SIDE, SIDE), dtype=bool)
        for label, idx in ((14, [5, 6, 7]), (15, [2, 3, 4])):
            upper, lower = self._arm_masks(kp[idx])
            out |= (parsing[..., 0] == label) & ~upper & ~lower
        return out[..., None].astype(np.uint8)

    # ------------------------------------------------------------------ one pair
    def __getitem__(self, idx):
        clothes_name, person_name = self.pairs[idx]
        raw = self._image(person_name)
        assert raw.shape[0] == SIDE, 'images are 512 pixels high (320 x 512 in the reference data)'
        image, left = _pad_square(raw, 255)
        pose, kp = self.pose_map(self._keypoints(person_name), raw.shape[:2])          # drawn in the unpadded frame, like the reference
        pose, _ = _pad_square(pose, 0)
        kp[:, 0] += left
        parsing, _ = _pad_square(self._labels('parsing', person_name), 0)

        is_ = lambda *labels: np.isin(parsing, labels).astype(np.uint8)
        retain_mask = is_(18, 19) + self.palm_mask(kp, parsing) + is_(1, 2, 4, 13)             # shoes + palms + head
        skin = is_(10, 13) * image                                                             # neck + face
        medians = []
        for ch in range(3):
            vals = skin[..., ch].reshape(-1)
            vals = vals[vals > 0]
            medians.append(np.median(vals) if vals.size else np.nan)
        skin_average = np.stack([np.full((SIDE, SIDE), m) for m in medians], axis=2)

        tops, dresses, pants, skirt = _garment_classes(parsing)
        lower_mask = skirt + pants
        lower_image = lower_mask * image
        lower_bbox = _bbox(lower_mask.copy())
        bound = np.zeros((SIDE, SIDE, 1), dtype=np.uint8)
        lhip, rhip = kp[11], kp[8]
        if lhip[2] > 0.05 and rhip[2] > 0.05:                # start of the lower garment: the hips, or the parsing if that is higher
            via_kps = int((lhip[1] + rhip[1]) / 2 - 3 * np.linalg.norm(lhip[0:2] - rhip[0:2]) / 4)
            top = via_kps if lower_bbox is None else min(lower_bbox[1], via_kps)
            bound[top:] += 255                               # (NumPy slice semantics, negative values included, as in the reference)
        elif lower_bbox is not None:
            bound[lower_bbox[1]:] += 255

        craw = self._image(clothes_name)
        clothes, _ = _pad_square(craw, 255)
        clothes_pose, ckp = self.pose_map(self._keypoints(clothes_name), craw.shape[:2])
        clothes_pose, _ = _pad_square(clothes_pose, 0)
        ckp[:, 0] += left
        cparsing, _ = _pad_square(self._labels('parsing', clothes_name), 0)
        ctops, cdresses, _, _ = _garment_classes(cparsing)
        upper_mask = ctops + cdresses
        upper_image = upper_mask * clothes
        if cdresses.sum() > 0:                               # a dress replaces the person's lower garment entirely
            lower_mask, pants, skirt, lower_image, bound = lower_mask * 0, pants * 0, skirt * 0, lower_image * 0, bound * 0
        upper_rgb, lower_rgb = np.repeat(upper_mask, 3, axis=2) * 255, np.repeat(lower_mask, 3, axis=2) * 255
        sleeve = None
        if self.use_sleeve_mask:
            gp, _ = _pad_square(self._labels('garment_parsing', clothes_name), 0)
            sleeve = np.isin(gp, (10, 11)).astype(np.uint8)

        routed = patch_routing.normalize(upper_image.astype(np.uint8), lower_image.astype(np.uint8), upper_rgb.astype(np.uint8), lower_rgb.astype(np.uint8),
                                         sleeve, ckp, kp, 2, device=self.device)
        norm_img, norm_img_lower, denorm_upper, denorm_upper_wo_sleeve, _ = (t.cpu().numpy() for t in routed)
        denorm_lower = lower_image * _erode_white(lower_rgb.astype(np.uint8))                  # the person's own lower garment, edge eroded

        upper_bbox = _bbox((denorm_upper_wo_sleeve.sum(axis=2, keepdims=True) > 0).astype(np.uint8))
        if upper_bbox is not None:
            bound[0:upper_bbox[3]] *= 0
        label = 0.0 if pants.sum() > 0 else (1.0 if skirt.sum() > 0 else (2.0 if cdresses.sum() > 0 else 1.0))
        lower_label_map = np.full((SIDE, SIDE, 1), label / 2.0 * 255)

        chw = lambda a: np.ascontiguousarray(np.transpose(a, (2, 0, 1)))
        denorm_upper, denorm_lower = chw(denorm_upper), chw(denorm_lower.astype(np.uint8))
        return (chw(image), chw(clothes), chw(pose), chw(clothes_pose), chw(norm_img), chw(norm_img_lower), denorm_upper, denorm_lower,
                (denorm_upper.sum(axis=0, keepdims=True) > 0).astype(np.uint8), (denorm_lower.sum(axis=0, keepdims=True) > 0).astype(np.uint8),
                chw(retain_mask), chw(skin_average), chw(lower_label_map), chw(bound), person_name, clothes_name)


def to_generator_inputs(batch, device):
    """The tensor preparation of test.py:126-147: uint8 arrays of `TryOnTestSet` (stacked by a DataLoader) -> the keyword
    arguments of ``GeneratorFull_v20.forward``."""
    (image, clothes, pose, _, norm_img, norm_img_lower, den_up, den_lo, den_up_mask, den_lo_mask, retain_mask, skin_average, lower_label_map,
     lower_bound) = [torch.as_tensor(t).to(device) for t in batch[:14]]
    unit = lambda t: t.to(torch.float32) / 127.5 - 1
    image_t, retain = unit(image), retain_mask.to(torch.float32)
    retain_t = torch.cat([image_t * retain - (1 - retain), unit(skin_average)], dim=1)
    return dict(z=torch.zeros([image.shape[0], 0], device=device), c=torch.cat([unit(norm_img), unit(norm_img_lower)], dim=1), retain=retain_t,
                pose=torch.cat([unit(pose), unit(lower_label_map), unit(lower_bound)], dim=1),
                denorm_upper_input=unit(den_up), denorm_lower_input=unit(den_lo),
                denorm_upper_mask=den_up_mask.to(torch.float32), denorm_lower_mask=den_lo_mask.to(torch.float32))
