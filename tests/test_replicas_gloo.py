"""N > 1 path on CPU: two gloo ranks shard a batch of images (rank-strided), each runs the synthesis forward on
its share, and rank 0 reassembles the outputs -- which must equal the single-process result bit for bit.  The
forward engine here is the CPU oracle (the product network needs the GPU); the code under test is the product's
sharding / gathering / timing-reduction host logic (training/replicas.py)."""

import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases as C
from detgen import fill_module_, synthesis_inputs


def _make_net():
    from oracle import network_ref as NR
    torch.manual_seed(0)
    return fill_module_(NR.SynthesisNetworkFull_v18(**C.G6_KW), 'g6.').eval()


def _batch(n):
    inp = synthesis_inputs(n, w_dim=C.G6_KW['w_dim'], num_ws=14, feat_ch=C.G6_FEAT_CH, seed_tag='rep', labels=True)
    return inp


def _forward(net, b):
    with torch.no_grad():
        return net(b['ws'], b['pose_feat'], b['cat_feat'], b['denorm_upper_input'], b['denorm_lower_input'],
                   b['denorm_upper_mask'], b['denorm_lower_mask'], b['gt_parsing'], noise_mode='const')


def _worker(rank, world, init_file, n_items, out_file):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pasta-gan-plusplus_amd'))
    from training import replicas
    torch.set_num_threads(3)
    dist.init_process_group('gloo', init_method=f'file://{init_file}', rank=rank, world_size=world)
    net = _make_net()
    outs, idx = replicas.run_sharded(lambda b: _forward(net, b), _batch(n_items), n_items)
    assert idx == list(range(rank, n_items, world))
    full = replicas.gather_outputs(outs, idx, n_items, dst=0)
    slowest = replicas.max_over_ranks(1.0 + rank)
    assert slowest == float(world)                      # MAX over ranks of (1 + rank)
    if rank == 0:
        np.savez(out_file, img=full[0].numpy(), fimg=full[1].numpy(), pp=full[2].numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_items', [2, 3])
def test_two_rank_sharding_matches_single_process(n_items):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pasta-gan-plusplus_amd'))
    from training import replicas
    assert replicas.shard_indices(5, 1, 2) == [1, 3] and replicas.shard_indices(1, 1, 2) == []
    with tempfile.TemporaryDirectory() as tmp:
        init_file, out_file = os.path.join(tmp, 'rdzv'), os.path.join(tmp, 'out.npz')
        mp.spawn(_worker, args=(2, init_file, n_items, out_file), nprocs=2, join=True)
        got = np.load(out_file)
        torch.set_num_threads(6)
        ref = _forward(_make_net(), _batch(n_items))
        for key, t in zip(('img', 'fimg', 'pp'), ref):
            assert got[key].shape == tuple(t.shape)
            np.testing.assert_allclose(got[key], t.numpy(), rtol=1e-5, atol=1e-5 * float(t.abs().max()))
