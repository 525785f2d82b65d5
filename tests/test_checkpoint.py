"""Row f4 of SURVEY.md section 8: reading the reference's snapshot pickles by parameter name, without executing them.

The fixture ``tests/golden/g10_snapshot.pkl`` was written by the reference's own ``torch_utils.persistence`` from the reference's
real network classes (``tests/golden/make_golden_pkl.py``; the embedded module source is replaced by a placeholder -- reference
source must not enter this repository); ``g10_snapshot_expected.npz`` holds what the reference's ``state_dict()`` reports.
"""
import io
import os
import pickle
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
PKL = os.path.join(HERE, 'golden', 'g10_snapshot.pkl')
EXPECTED = os.path.join(HERE, 'golden', 'g10_snapshot_expected.npz')


def test_state_dicts_match_the_reference_by_name_order_and_value():
    from training import checkpoint as ck
    exp = np.load(EXPECTED)
    sds = ck.read_state_dicts(PKL)
    assert list(sds) == ['G', 'D', 'G_ema']
    total = 0
    for key, sd in sds.items():
        names = [k.split('/', 1)[1] for k in exp.files if k.startswith(key + '/')]
        assert list(sd.keys()) == names                      # torch.nn.Module.state_dict() order
        for name, t in sd.items():
            want = exp[f'{key}/{name}']
            assert tuple(t.shape) == want.shape and str(t.dtype).replace('torch.', '') == str(want.dtype)
            assert np.array_equal(t.numpy(), want)
            total += 1
    assert total == 126


def test_snapshot_metadata_is_plain_data():
    from training import checkpoint as ck
    snap = ck.read_snapshot(PKL)
    assert snap['training_set_kwargs'] == dict(class_name='training.dataset.UvizFullBodyDataset', path='/data', resolution=512)
    assert snap['augment_pipe'] is None
    d = snap['D']
    assert d.class_name == 'Discriminator' and d.persistent and not d.training
    assert d.init_kwargs.img_resolution == 16 and d.init_kwargs.epilogue_kwargs == dict(mbstd_group_size=2)
    assert d.src_info[0] > 0 and len(d.src_info[1]) == 40          # the embedded source was measured, not run
    g = snap['G']
    assert g.class_name == 'ModuleDict' and g.class_module == 'torch.nn.modules.container' and not g.persistent
    assert list(g.children()) == ['mapping', 'torgb', 'resblock', 'spade']
    assert g.children()['spade'].children()['spade0'].children()['param_free_norm'].class_name == 'InstanceNorm2d'
    with open(PKL, 'rb') as f:                                     # file objects work as well as paths
        assert list(ck.read_state_dicts(f, ['D'])) == ['D']
    with pytest.raises(KeyError):
        ck.read_state_dicts(PKL, ['G_missing'])


def test_loads_into_this_packages_networks_strictly():
    from training import checkpoint as ck
    from training import networks as PN
    kw = dict(z_dim=0, c_dim=16, w_dim=16, num_ws=4)
    g = torch.nn.ModuleDict(dict(
        mapping=PN.MappingNetwork(num_layers=2, **kw),
        torgb=PN.ToRGBLayerFull_v1_v5(8, 3, w_dim=16, conv_clamp=256, is_last=True, is_style=True),
        resblock=PN.ResBlock(6, 8, 3, down=2),
        spade=PN.Spade_ResBlockV4_512(8, 8, spade_channels=5)))
    d = PN.Discriminator(c_dim=0, img_resolution=16, img_channels=3, channel_base=128, channel_max=16, num_fp16_res=1, conv_clamp=256,
                         block_kwargs={}, mapping_kwargs={}, epilogue_kwargs=dict(mbstd_group_size=2))
    res = ck.load_into(g, PKL, key='G_ema', strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    ck.load_into(d, ck.read_snapshot(PKL)['D'])
    exp = np.load(EXPECTED)
    for name, t in d.state_dict().items():
        assert np.array_equal(t.numpy(), exp[f'D/{name}'])
    for name, t in g.state_dict().items():
        assert np.array_equal(t.numpy(), exp[f'G_ema/{name}'])


class _Boom:
    def __reduce__(self):
        return (os.system, ('echo pwned > /tmp/pg_checkpoint_pwned',))


def test_nothing_in_a_snapshot_is_executed(tmp_path, monkeypatch):
    from training import checkpoint as ck
    # 1. an object that would run a shell command under pickle.load is refused at the global lookup
    blob = pickle.dumps(dict(G=_Boom()))
    with pytest.raises(pickle.UnpicklingError, match='allow-list'):
        ck.read_snapshot(io.BytesIO(blob))
    assert not os.path.exists('/tmp/pg_checkpoint_pwned')
    # 2. a persistent object whose embedded module source has side effects: the reference's loader would exec it
    #    (persistence.py:189); here it is only hashed
    marker = tmp_path / 'executed'
    src = f"open({str(marker)!r}, 'w').write('x')\nclass Net: pass\n"
    meta = dict(type='class', version=6, module_src=src, class_name='Net',
                state=dict(_parameters={'w': torch.nn.Parameter(torch.ones(2))}, _buffers={}, _modules={}, training=False))

    # pickle it exactly as a persistent_class instance does: reduce to torch_utils.persistence._reconstruct_persistent_obj(meta)
    import types
    fake = types.ModuleType('torch_utils.persistence')

    def _reconstruct_persistent_obj(meta):
        raise AssertionError('the writer side is never called')
    _reconstruct_persistent_obj.__module__ = 'torch_utils.persistence'
    _reconstruct_persistent_obj.__qualname__ = '_reconstruct_persistent_obj'
    fake._reconstruct_persistent_obj = _reconstruct_persistent_obj
    monkeypatch.setitem(sys.modules, 'torch_utils.persistence', fake)

    class _P:
        def __reduce__(self):
            return (_reconstruct_persistent_obj, (meta,))
    blob = pickle.dumps(dict(G_ema=_P()))
    assert b'torch_utils.persistence' in blob and b'_reconstruct_persistent_obj' in blob
    snap = ck.read_snapshot(io.BytesIO(blob))
    assert not marker.exists()
    assert snap['G_ema'].class_name == 'Net' and torch.equal(snap['G_ema'].state_dict()['w'], torch.ones(2))
    # 3. other callables, even harmless builtins, stay off the list
    for mod, name in (('builtins', 'eval'), ('torch', 'load'), ('torch.nn.modules.module', 'os.system'), ('subprocess', 'Popen')):
        with pytest.raises(pickle.UnpicklingError):
            ck.SnapshotUnpickler(io.BytesIO(b'')).find_class(mod, name)

