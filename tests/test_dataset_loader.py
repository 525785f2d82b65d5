"""BASELINE config 1 plumbing on the CPU: the product's own loader (training/dataset.py) on SYNTHETIC pairs written in the
reference's file formats (RGB JPEG 320x512, L-mode parsing PNG with LIP labels, garment-parsing PNG, OpenPose-18 JSON), the
16-tuple contract of dataset.py:2702-2726, the patch routing's CPU route against the oracle (bit for bit), and one batch-1
generator forward on the CPU through the ops' plain-torch route."""

import json
import os

import numpy as np
import pytest
import torch

PIL = pytest.importorskip('PIL.Image')

JOINTS = dict(cnose=(160, 60), cneck=(160, 110), rshoulder=(104, 120), relbow=(84, 200), rwrist=(74, 270), lshoulder=(216, 120), lelbow=(239, 200),
              lwrist=(249, 270), rhip=(124, 290), rknee=(119, 390), rankle=(116, 480), lhip=(196, 290), lknee=(201, 390), lankle=(204, 480),
              reye=(150, 50), leye=(170, 50), rear=(140, 55), lear=(180, 55))
ORDER = ['cnose', 'cneck', 'rshoulder', 'relbow', 'rwrist', 'lshoulder', 'lelbow', 'lwrist', 'rhip', 'rknee', 'rankle', 'lhip', 'lknee', 'lankle',
         'reye', 'leye', 'rear', 'lear']


def _write_person(root, name, rng, dress=False):
    """One synthetic 'photo': noise image, a blocky person-shaped label map, its garment parsing and jittered keypoints."""
    for d in ('image', 'parsing', 'garment_parsing', 'keypoints'):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    img = rng.integers(30, 226, (512, 320, 3), dtype=np.uint8)
    PIL.fromarray(img, 'RGB').save(os.path.join(root, 'image', name + '.jpg'), quality=95)
    lab = np.zeros((512, 320), np.uint8)
    lab[30:90, 130:190] = 13          # face
    lab[15:30, 130:190] = 2           # hair
    lab[90:112, 145:175] = 10         # neck
    lab[112:300, 100:220] = 6 if dress else 5        # dress / top
    lab[112:280, 70:100] = 14         # left arm + hand
    lab[112:280, 220:250] = 15
    if dress:
        lab[300:420, 100:220] = 6
    else:
        lab[290:470, 110:210] = 9     # pants
    lab[470:500, 105:150] = 18
    lab[470:500, 170:215] = 19
    PIL.fromarray(lab, 'L').save(os.path.join(root, 'parsing', name + '.png'))
    gp = np.zeros((512, 320, 3), np.uint8)
    gp[112:200, 70:100, 0] = 10
    gp[112:200, 220:250, 0] = 11
    PIL.fromarray(gp, 'RGB').save(os.path.join(root, 'garment_parsing', name + '.png'))
    kp = []
    for k in ORDER:
        x, y = JOINTS[k]
        kp += [float(x + rng.normal(0, 4)), float(y + rng.normal(0, 4)), 0.9]
    with open(os.path.join(root, 'keypoints', name + '_keypoints.json'), 'w') as f:
        json.dump(dict(version=1.3, people=[dict(pose_keypoints_2d=kp)]), f)


@pytest.fixture(scope='module')
def synthetic_root(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('pairs'))
    rng = np.random.default_rng(5)
    _write_person(root, 'person_a', rng)
    _write_person(root, 'person_b', rng)
    _write_person(root, 'dress_c', rng, dress=True)
    with open(os.path.join(root, 'test_pairs.txt'), 'w') as f:
        f.write('person_b.jpg person_a.jpg\ndress_c.jpg person_b.jpg\n')
    return root


SHAPES = [(3, 512, 512), (3, 512, 512), (3, 512, 512), (3, 512, 512), (30, 128, 128), (15, 128, 128), (3, 512, 512), (3, 512, 512),
          (1, 512, 512), (1, 512, 512), (1, 512, 512), (3, 512, 512), (1, 512, 512), (1, 512, 512)]


def test_loader_contract(synthetic_root):
    from training.dataset import TryOnTestSet
    ds = TryOnTestSet(synthetic_root, use_sleeve_mask=True, device='cpu')
    assert len(ds) == 2
    for idx in range(2):
        item = ds[idx]
        assert len(item) == 16 and item[14].endswith('.jpg') and item[15].endswith('.jpg')
        for a, shape in zip(item[:14], SHAPES):
            assert tuple(a.shape) == shape, (a.shape, shape)
        image, clothes, pose, cpose, norm_img, norm_lower, dup, dlo, mup, mlo, retain, skin, label, bound = item[:14]
        for a in (image, clothes, pose, cpose, norm_img, norm_lower, dup, dlo, mup, mlo, retain, bound):
            assert a.dtype == np.uint8
        assert (image[:, :, :96] == 255).all() and (image[:, :, 416:] == 255).all()          # 320 -> 512 white side bars
        assert set(np.unique(mup)) <= {0, 1} and set(np.unique(mlo)) <= {0, 1} and set(np.unique(retain)) <= {0, 1}
        assert mup.sum() > 1000                                                              # the garment was routed onto the person
        assert np.array_equal(mup[0], (dup.sum(axis=0) > 0).astype(np.uint8))
        assert retain[0, 40:80, 96 + 140:96 + 180].all() and retain[0, 475:495, 96 + 110:96 + 145].all()   # face and a shoe are retained
        assert pose.any() and set(np.unique(bound)) <= {0, 255} and np.isfinite(skin).all()
        assert float(label.max()) in (0.0, 127.5, 255.0)
    # a dress as the garment removes the person's lower garment (dataset.py:2173-2178)
    assert ds[1][7].sum() == 0 and ds[1][9].sum() == 0 and float(ds[1][12].max()) == 255.0


def test_loader_routing_matches_the_oracle_on_cpu(synthetic_root):
    """The loader's patch routing on the CPU (NumPy route of training.patch_routing) == the oracle's normalize, bit for bit."""
    from training import patch_routing as P
    from oracle import patch_routing_ref as R
    rng = np.random.default_rng(11)
    kp = np.array([[JOINTS[k][0] + 96 + rng.normal(0, 6), JOINTS[k][1] + rng.normal(0, 6), 1.0] for k in ORDER])
    kp2 = kp + rng.normal(0, 5, kp.shape) * [1, 1, 0]
    up, lo = (rng.integers(0, 256, (512, 512, 3), dtype=np.uint8) for _ in range(2))
    um = np.zeros((512, 512, 3), np.uint8); um[100:300, 180:330] = 255
    lm = np.zeros((512, 512, 3), np.uint8); lm[280:480, 200:320] = 255
    sleeve = np.zeros((512, 512, 1), np.uint8); sleeve[100:300, :215] = 1
    want = R.normalize(up, lo, um, lm, sleeve, kp, kp2, 2)
    got = P.normalize(up, lo, um, lm, sleeve, kp, kp2, 2, device='cpu')
    for g, w in zip(got, want):
        assert g.device.type == 'cpu' and np.array_equal(g.numpy(), w)


def test_config1_generator_forward_on_cpu(synthetic_root):
    """test.py's flow at batch 1 without a GPU: loader -> tensors (test.py:126-147) -> GeneratorFull_v20 through the ops'
    plain-torch route.  Reduced synthesis width (channel_base 4096) keeps it to seconds; shapes are the real ones."""
    from training.dataset import TryOnTestSet, to_generator_inputs
    from training import networks as PN
    from detgen import fill_module_
    ds = TryOnTestSet(synthetic_root, device='cpu')
    loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, num_workers=0)
    batch = next(iter(loader))
    inp = to_generator_inputs(batch, 'cpu')
    assert inp['c'].shape == (1, 45, 128, 128) and inp['retain'].shape == (1, 6, 512, 512) and inp['pose'].shape == (1, 5, 512, 512)
    torch.manual_seed(0)
    G = fill_module_(PN.GeneratorFull_v20(z_dim=0, c_dim=512, w_dim=64, img_resolution=512, img_channels=3, mapping_kwargs=dict(num_layers=1),
                                          synthesis_kwargs=dict(channel_base=4096, channel_max=512, conv_clamp=256)), 'cfg1.').eval()
    with torch.no_grad():
        img, finetune_img, pred_parsing = G(**inp, noise_mode='const')
    assert img.shape == finetune_img.shape == (1, 3, 512, 512) and pred_parsing.shape == (1, 7, 512, 512)
    assert all(torch.isfinite(t).all() for t in (img, finetune_img, pred_parsing))
