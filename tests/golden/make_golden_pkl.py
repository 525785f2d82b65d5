#!/usr/bin/env python3
"""Generate tests/golden/g10_snapshot.pkl: a network snapshot in THE REFERENCE'S OWN pickle format.

Runs only in the build container (needs /root/reference).  The reference's training loop writes snapshots with
``pickle.dump(dict(G=..., D=..., G_ema=..., training_set_kwargs=...), f)`` (training_loop_fullbody.py:723-736); every
``@persistence.persistent_class`` module pickles as ``_reconstruct_persistent_obj(meta)`` with
``meta = dict(type, version, module_src, class_name, state)`` (torch_utils/persistence.py:118-126).  This script builds small
instances of the reference's real classes with name-keyed deterministic weights and pickles them through the reference's
own ``persistence`` module, so the wire format is the reference's byte for byte -- EXCEPT ``module_src``: the reference
embeds the source text of training/networks.py there; the fixture carries a one-line placeholder instead (patched into
``persistence._module_to_src`` before the classes are decorated), because reference source must not enter this repository.
Readers that follow the reference (``legacy.load_network_pkl``) exec that field; this package's reader never does.

Also writes g10_snapshot_expected.npz: the tensors the reference's own ``state_dict()`` reports for the same modules
(the expected output of ``training.checkpoint.read_state_dicts``).
"""

import os
import pickle
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from detgen import fill_module_  # noqa: E402

REF = '/root/reference'
if not os.path.isdir(REF):
    sys.exit('make_golden_pkl.py needs /root/reference (build container only)')
torch.version.cuda = '10.0'
sys.path.insert(0, REF)
os.chdir(REF)

from torch_utils import persistence  # noqa: E402

PLACEHOLDER = '# module source elided by tests/golden/make_golden_pkl.py (the reference embeds training/networks.py here)\n'
persistence._module_to_src = lambda module: PLACEHOLDER      # must happen before training.networks is imported (decoration time)

import dnnlib  # noqa: E402
import training.networks as RN  # noqa: E402


def build():
    kw = dict(z_dim=0, c_dim=16, w_dim=16, num_ws=4)
    mapping = fill_module_(RN.MappingNetwork(num_layers=2, **kw), 'g10.map.')
    torgb = fill_module_(RN.ToRGBLayerFull_v1_v5(8, 3, w_dim=16, conv_clamp=256, is_last=True, is_style=True), 'g10.torgb.')
    resblock = fill_module_(RN.ResBlock(6, 8, 3, down=2), 'g10.res.')
    spade = fill_module_(RN.Spade_ResBlockV4_512(8, 8, spade_channels=5), 'g10.spade.')
    disc = fill_module_(RN.Discriminator(c_dim=0, img_resolution=16, img_channels=3, channel_base=128, channel_max=16, num_fp16_res=1,
                                         conv_clamp=256, block_kwargs={}, mapping_kwargs={}, epilogue_kwargs=dict(mbstd_group_size=2)), 'g10.d.')
    holder = torch.nn.ModuleDict(dict(mapping=mapping, torgb=torgb, resblock=resblock, spade=spade))
    return holder, disc


def main():
    g, d = build()
    g_ema = build()[0]
    for m in (g, d, g_ema):
        m.eval().requires_grad_(False).cpu()
    snapshot = dict(training_set_kwargs=dict(dnnlib.EasyDict(class_name='training.dataset.UvizFullBodyDataset', path='/data', resolution=512)),
                    G=g, D=d, G_ema=g_ema, augment_pipe=None)
    path = os.path.join(HERE, 'g10_snapshot.pkl')
    with open(path, 'wb') as f:
        pickle.dump(snapshot, f)
    raw = open(path, 'rb').read()
    assert b'def modulated_conv2d' not in raw and b'NVIDIA' not in raw, 'reference source leaked into the fixture'
    expected = {}
    for key in ('G', 'D', 'G_ema'):
        for name, t in snapshot[key].state_dict().items():
            expected[f'{key}/{name}'] = t.numpy()
    expected['__init_kwargs_D__'] = np.array(repr(dict(d.init_kwargs)))
    np.savez_compressed(os.path.join(HERE, 'g10_snapshot_expected.npz'), **expected)
    print(f'wrote {path}: {len(raw) / 1024:.0f} KiB, {len(expected) - 1} tensors')


if __name__ == '__main__':
    main()
