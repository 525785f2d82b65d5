"""Row f3 of SURVEY.md section 8: the data loader's patch-routing warps (training/dataset.py:2373-2700).

PARITY UNPINNED: the reference does this with OpenCV, which is neither installed here nor part of the reference tree, and the
reference holds no fixtures for this step.  The oracle (oracle/patch_routing_ref.py) restates OpenCV's published algorithms; these
tests check (CPU) the oracle's own invariants and that the product's host geometry is bit-identical to it, and (GPU) that the HIP
kernels reproduce the oracle bit for bit -- through the C ABI."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JOINTS = dict(cnose=(256, 60), cneck=(256, 110), rshoulder=(200, 120), relbow=(180, 200), rwrist=(170, 270), lshoulder=(312, 120), lelbow=(335, 200),
              lwrist=(345, 270), rhip=(220, 290), rknee=(215, 390), rankle=(212, 480), lhip=(292, 290), lknee=(297, 390), lankle=(300, 480),
              reye=(246, 50), leye=(266, 50), rear=(236, 55), lear=(276, 55))


def keypoints(rng, jitter=10.0, drop=()):
    from oracle import patch_routing_ref as R
    kp = np.zeros((18, 3))
    for k, (x, y) in JOINTS.items():
        kp[R.ORDER.index(k)] = (x + rng.normal(0, jitter), y + rng.normal(0, jitter), 0.0 if k in drop else 1.0)
    return kp


def test_oracle_warp_invariants():
    from oracle import patch_routing_ref as R
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    assert np.array_equal(R.warp_perspective_u8(img, np.eye(3), (50, 40)), img)                      # identity
    t = np.array([[1, 0, 3], [0, 1, -2], [0, 0, 1.0]])
    want = np.zeros_like(img)
    want[:38, 3:] = img[2:, :47]
    assert np.array_equal(R.warp_perspective_u8(img, t, (50, 40)), want)                             # integer shift, zero border
    half = R.warp_perspective_u8(img, np.array([[1, 0, 0.5], [0, 1, 0], [0, 0, 1.0]]), (50, 40))
    a, b = img[:, :-1].astype(int), img[:, 1:].astype(int)
    assert np.array_equal(half[:, 1:], ((a * 16384 + b * 16384 + 16384) >> 15).astype(np.uint8))     # half-pixel: rounded mean of neighbours
    up2 = R.warp_perspective_u8(img[:, :, 0], np.diag([2.0, 2.0, 1.0]), (100, 80))                   # 2x zoom hits the samples at even pixels
    assert np.array_equal(up2[::2, ::2], img[:, :, 0])
    src = np.float32([[10, 10], [10, 30], [40, 35], [42, 8]])
    dst = np.float32([[0, 0], [0, 32], [32, 32], [32, 0]])
    m = R.get_perspective_transform(src, dst)
    for s, d in zip(src, dst):
        p = m @ np.array([s[0], s[1], 1.0])
        assert np.allclose(p[:2] / p[2], d, atol=1e-9)
    assert np.allclose(R.invert3x3(m) @ m / (R.invert3x3(m) @ m)[2, 2], np.eye(3), atol=1e-12)
    mask = np.full((20, 20), 255, np.uint8)
    mask[10, 10] = 0
    er = R.erode_u8(mask, 8)
    assert (er[7:15, 7:15] == 0).all() and er[6, 10] == 255 and er[15, 10] == 255 and er[0, 0] == 255   # window [-4, +3] around the hole


def test_host_geometry_is_bit_identical_to_the_oracle():
    from oracle import patch_routing_ref as R
    from training import patch_routing as P
    rng = np.random.default_rng(1)
    wh = np.array([[128, 128]])
    checked = skipped = 0
    for trial in range(200):
        kp = keypoints(rng, 12.0)
        kp[:, 2] = rng.choice([1.0, 1.0, 1.0, 0.05, 0.0], size=18)
        for ii, bp in enumerate(R.BPARTS):
            ar = 0.5 if ii < 6 else 0.4
            a, b = R.get_crop(kp, bp, wh, 512, 512, ar), P.get_crop(kp, bp, wh, 512, 512, ar)
            assert (a[0] is None) == (b[0] is None)
            if a[0] is None:
                skipped += 1
                continue
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
            checked += 1
    assert checked > 500 and skipped > 100                     # both the regular and the fall-back / missing-joint paths were exercised
    m = rng.normal(size=(3, 3))
    assert np.array_equal(R.invert3x3(m), P.invert3x3(m))
    assert P._block_width(512, 512) == R.block_width(512, 512) == 64 and P._block_width(8, 40) == R.block_width(8, 40) == 40


@pytest.mark.gpu
@pytest.mark.parametrize('sh,sw,dh,dw,c', [(64, 80, 64, 80, 3), (512, 512, 128, 128, 3), (128, 128, 512, 512, 3), (37, 53, 45, 29, 1), (512, 512, 512, 512, 3)])
def test_warp_perspective_matches_the_oracle_bit_for_bit(sh, sw, dh, dw, c):
    from oracle import patch_routing_ref as R
    from training import patch_routing as P
    rng = np.random.default_rng(sh * 1000 + dw)
    jobs, want = [], []
    for k in range(6):
        img = rng.integers(0, 256, (sh, sw, c) if c > 1 else (sh, sw), dtype=np.uint8)
        src = np.float32([[rng.uniform(-0.2, 0.4) * sw, rng.uniform(-0.2, 0.4) * sh], [rng.uniform(-0.2, 0.4) * sw, rng.uniform(0.6, 1.2) * sh],
                          [rng.uniform(0.6, 1.2) * sw, rng.uniform(0.6, 1.2) * sh], [rng.uniform(0.6, 1.2) * sw, rng.uniform(-0.2, 0.4) * sh]])
        dst = np.float32([[0, 0], [0, dh], [dw, dh], [dw, 0]])
        m = R.get_perspective_transform(src, dst) if k else np.array([[1.0, 0, 0.5], [0, 1.0, -3.0], [0, 0, 1.0]])      # job 0: a pure shift
        jobs.append((torch.from_numpy(img).cuda(), m, (dw, dh)))
        want.append(R.warp_perspective_u8(img, m, (dw, dh)))
    got = P.warp_perspective_batch(jobs)
    for g, w_ in zip(got, want):
        assert g.dtype == torch.uint8 and tuple(g.shape) == w_.shape
        assert np.array_equal(g.cpu().numpy(), w_)


@pytest.mark.gpu
def test_patch_compose_matches_erode_and_paste():
    from oracle import patch_routing_ref as R
    from training import patch_routing as P
    rng = np.random.default_rng(7)
    h, w = 96, 120
    patch = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    canvas = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    mask = (rng.random((h, w, 3)) > 0.02).astype(np.uint8) * 255
    mask[:, :, 0][rng.random((h, w)) > 0.995] = 254                 # near-white is not white
    m = (R.erode_u8(mask[..., 0], 8)[..., None] == 255).astype(np.uint8)
    want = patch * m + canvas * (1 - m)
    c1, c2 = torch.from_numpy(canvas).cuda(), torch.zeros(h, w, 3, dtype=torch.uint8, device='cuda')
    P.patch_compose_(c1, torch.from_numpy(patch).cuda(), torch.from_numpy(mask).cuda(), c2)
    assert np.array_equal(c1.cpu().numpy(), want)
    assert np.array_equal(c2.cpu().numpy(), patch * m)
    assert 0.05 < m.mean() < 0.9


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['all_joints', 'missing_knees_and_nose', 'no_left_arm_with_sleeve_mask', 'nothing_valid'])
def test_normalize_matches_the_oracle_bit_for_bit(case):
    from oracle import patch_routing_ref as R
    from training import patch_routing as P
    rng = np.random.default_rng(len(case))
    drop = dict(all_joints=(), missing_knees_and_nose=('lknee', 'rknee', 'cnose'), no_left_arm_with_sleeve_mask=('lelbow', 'lwrist'),
                nothing_valid=tuple(JOINTS))[case]
    ckp, pkp = keypoints(rng, 8.0, drop), keypoints(rng, 8.0, drop)
    up, lo = (rng.integers(0, 256, (512, 512, 3), dtype=np.uint8) for _ in range(2))
    um = np.zeros((512, 512, 3), np.uint8)
    um[90:310, 150:370] = 255
    lm = np.zeros((512, 512, 3), np.uint8)
    lm[270:505, 190:330] = 255
    sleeve = None
    if 'sleeve' in case:
        sleeve = np.zeros((512, 512, 1), np.uint8)
        sleeve[100:300, :215] = 1
        sleeve[100:300, 300:] = 1
    want = R.normalize(up, lo, um, lm, sleeve, ckp, pkp, 2)
    got = P.normalize(up, lo, um, lm, sleeve, ckp, pkp, 2)
    names = ('norm_img', 'norm_img_lower', 'denorm_upper_img', 'denorm_upper_img_wo_sleeve', 'denorm_lower_img')
    for nm, g, w_ in zip(names, got, want):
        assert tuple(g.shape) == w_.shape and g.dtype == torch.uint8, nm
        assert np.array_equal(g.cpu().numpy(), w_), nm
    if case == 'all_joints':
        assert all(int(w_.astype(np.int64).sum()) > 0 for w_ in want)
    if case == 'nothing_valid':
        assert all(int(w_.astype(np.int64).sum()) == 0 for w_ in want)


@pytest.mark.gpu
def test_normalize_batch_equals_the_oracle_per_sample_bit_for_bit():
    """`normalize_batch` (round 6: three native launches for the whole batch -- two warp launches over every job of every sample, one ordered erode-and-paste
    launch over every canvas --, one batched solve for every homography) against the ORACLE's per-sample `normalize`, bit for bit, on a batch that mixes the
    four cases of the per-sample test: all joints, missing knees and nose (leg fall-backs), a missing left arm with a sleeve mask (mirrored sleeve), nothing valid
    (empty job lists, canvases written as zeros by the compose launch); and against the product's own per-sample route."""
    from oracle import patch_routing_ref as R
    from training import patch_routing as P
    cases = ['all_joints', 'missing_knees_and_nose', 'no_left_arm_with_sleeve_mask', 'nothing_valid', 'all_joints_b', 'no_right_arm_with_sleeve_mask']
    samples = []
    for case in cases:
        rng = np.random.default_rng(len(case) + 100)
        drop = dict(all_joints=(), all_joints_b=(), missing_knees_and_nose=('lknee', 'rknee', 'cnose'), no_left_arm_with_sleeve_mask=('lelbow', 'lwrist'),
                    no_right_arm_with_sleeve_mask=('relbow', 'rwrist'), nothing_valid=tuple(JOINTS))[case]
        ckp, pkp = keypoints(rng, 8.0, drop), keypoints(rng, 8.0, drop)
        up, lo = (rng.integers(0, 256, (512, 512, 3), dtype=np.uint8) for _ in range(2))
        um = np.zeros((512, 512, 3), np.uint8)
        um[90:310, 150:370] = 255
        lm = np.zeros((512, 512, 3), np.uint8)
        lm[270:505, 190:330] = 255
        sleeve = None
        if 'sleeve' in case:
            sleeve = np.zeros((512, 512, 1), np.uint8)
            sleeve[100:300, :215] = 1
            sleeve[100:300, 300:] = 1
        samples.append((up, lo, um, lm, sleeve, ckp, pkp))
    P.traffic_counter = dict(bytes=0, launches=0)
    try:
        got = P.normalize_batch(samples, 2)
        launches = P.traffic_counter['launches']
    finally:
        P.traffic_counter = None
    assert launches == 3
    names = ('norm_img', 'norm_img_lower', 'denorm_upper_img', 'denorm_upper_img_wo_sleeve', 'denorm_lower_img')
    for i, s in enumerate(samples):
        want = R.normalize(*s, 2)
        own = P.normalize(*s, 2)
        for nm, g, w_, o in zip(names, got, want, own):
            assert tuple(g[i].shape) == w_.shape and g.dtype == torch.uint8, (cases[i], nm)
            assert np.array_equal(g[i].cpu().numpy(), w_), (cases[i], nm)
            assert torch.equal(g[i], o), (cases[i], nm)


def test_batched_host_geometry_is_bit_identical_to_the_single_solves():
    from training import patch_routing as P
    rng = np.random.default_rng(5)
    src, dst = rng.normal(size=(64, 4, 2)) * 60 + 200, rng.normal(size=(64, 4, 2)) * 60 + 200
    mats = P.perspective_transforms(src, dst)
    assert np.array_equal(mats, np.stack([P.get_perspective_transform(s, d) for s, d in zip(src, dst)]))
    mats[3] = 0.0                                                    # a singular matrix: cv::invert's zero result
    assert np.array_equal(P.invert3x3_batch(mats), np.stack([P.invert3x3(m).reshape(9) for m in mats]))
