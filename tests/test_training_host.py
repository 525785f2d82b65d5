"""CPU tests of the training step's host logic (loss / phase orchestration, flat-bucket gradient exchange), with tiny
stub networks: the product's loss is pinned against the REFERENCE's StyleGAN2Loss (golden G9), and the 2-rank gloo
run must reproduce the single-process averaged gradients."""

import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import stubs

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pasta-gan-plusplus_amd')


def _loss(nets):
    from training.loss import StyleGAN2Loss
    return StyleGAN2Loss(device=torch.device('cpu'), **nets, augment_pipe=None, style_mixing_prob=0, r1_gamma=10, pl_weight=0,
                         l1_weight=50, vgg_weight=0, contextual_weight=0, mask_weight=1.0)


@pytest.mark.parametrize('phase', stubs.PHASES)
def test_loss_phases_match_reference(golden, phase):
    g = golden('g9_loss.npz')
    nets = stubs.build()
    loss = _loss(nets)
    stubs.zero_grads(nets)
    stubs.set_phase_trainable(nets, phase)
    loss.accumulate_gradients(phase=phase, sync=True, gain=(16 if phase.endswith('reg') else 1), **stubs.batch())
    sig = stubs.grad_signature(nets)
    assert list(sig.keys()) == list(g[f'{phase}/names'])
    np.testing.assert_allclose(np.array(list(sig.values())), g[f'{phase}/abssum'], rtol=2e-4, atol=1e-7)


def test_vgg_terms_are_refused():
    from training.loss import StyleGAN2Loss
    with pytest.raises(NotImplementedError):
        StyleGAN2Loss(device=torch.device('cpu'), **stubs.build(), vgg_weight=50)


def test_training_step_schedule_and_ema():
    from training.training_step import TrainingStep
    nets = stubs.build()
    G_parts = {k: v for k, v in nets.items() if k.startswith('G_')}
    step = TrainingStep(G_parts, nets['D'], nets['D_parsing'], _loss(nets), batch_size=4)
    assert [p.name for p in step.phases] == ['Gmain', 'Greg', 'Dmain', 'Dreg', 'D_parsingmain', 'D_parsingreg', 'D_parsingmain', 'D_parsingreg']
    assert [p.interval for p in step.phases] == [1, 4, 1, 16, 1, 16, 1, 16]
    g_opt = step.phases[0].opt.param_groups[0]
    assert abs(g_opt['lr'] - 0.0005 * 4 / 5) < 1e-12 and abs(g_opt['betas'][1] - 0.99 ** (4 / 5)) < 1e-12
    d_opt = step.phases[2].opt.param_groups[0]
    assert abs(d_opt['lr'] - 0.0005 * 16 / 17) < 1e-12
    before = {k: [p.clone() for p in m.parameters()] for k, m in nets.items()}
    ema_before = [p.clone() for p in step.G_ema_parts['G_synthesis'].parameters()]
    b = stubs.batch()
    step.run([b])                                   # batch_idx 0: every phase is due
    assert step.batch_idx == 1 and [p.name for p in step.due_phases()] == ['Gmain', 'Dmain', 'D_parsingmain', 'D_parsingmain']
    for k, m in nets.items():
        assert any(not torch.equal(a, p) for a, p in zip(before[k], m.parameters())), f'{k} was not updated'
        assert all(not p.requires_grad for p in m.parameters())
    assert any(not torch.equal(a, p) for a, p in zip(ema_before, step.G_ema_parts['G_synthesis'].parameters()))
    step.run([b])
    assert step.batch_idx == 2 and all(torch.isfinite(p).all() for m in nets.values() for p in m.parameters())


def _ddp_worker(rank, world, init_file, out_file, gather):
    sys.path.insert(0, PKG)
    from training import ddp
    torch.set_num_threads(2)
    dist.init_process_group('gloo', init_method=f'file://{init_file}', rank=rank, world_size=world)
    nets = stubs.build()
    loss = _loss(nets)
    full = stubs.batch(4)
    mine = {k: v[rank::world] for k, v in full.items()}            # rank-strided shard of the batch
    stubs.set_phase_trainable(nets, 'Dboth')
    params = list(nets['D'].parameters())
    extra = torch.nn.Parameter(torch.zeros(3))                      # a parameter that never receives a gradient
    bucket = ddp.GradBucket([extra] + params, segments=3, gather=gather)           # first registered = last segment (like synthesis.b8.const): earlier segments still launch from hooks
    assert len(bucket.seg_range) >= 2                               # several segments: some launch from the hooks, the rest in finish()
    bucket.begin()
    if gather:
        assert all(p.grad is None for p in bucket.params)           # autograd keeps the produced tensors; a segment is gathered into the bucket in front of its exchange
    else:
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))      # gradients ARE the bucket
    loss.on_last_backward = bucket.last_round                       # Dboth = two backward calls; only the second may exchange
    loss.accumulate_gradients(phase='Dboth', sync=True, **mine)
    launched_by_hooks = sum(1 for _, who in bucket.launch_log if who == 'hook')
    assert bucket.finish() is True
    assert extra.grad is None                                       # untouched on every rank: stays None (no zero gradient for Adam)
    assert all(p.grad is not None and p.grad.data_ptr() == bucket.views[1 + i].data_ptr() for i, p in enumerate(params))      # (either mode: .grad ends as the bucket view)
    if rank == 0:
        np.savez(out_file, launched_by_hooks=launched_by_hooks, **{f'p{i}': p.grad.numpy() for i, p in enumerate(params)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('gather', [False, True], ids=['grad_views', 'gathered'])
def test_flat_bucket_overlapped_exchange_two_ranks(gather):
    """(`gathered`: the GPU default of round 5 -- .grad starts as None, a complete segment is copied into the bucket in one multi-tensor copy.)
    mean over 2 ranks of per-shard gradients == gradient of the mean loss over the whole batch (the StubD has no
    cross-sample coupling), for a phase with two backward calls and a double-backward term (R1); at least one segment
    of the bucket was exchanged from the autograd hooks, i.e. while the last backward was still running."""
    with tempfile.TemporaryDirectory() as tmp:
        init_file, out_file = os.path.join(tmp, 'rdzv'), os.path.join(tmp, 'g.npz')
        mp.spawn(_ddp_worker, args=(2, init_file, out_file, gather), nprocs=2, join=True)
        got = np.load(out_file)
    assert int(got['launched_by_hooks']) >= 1
    nets = stubs.build()
    loss = _loss(nets)
    stubs.set_phase_trainable(nets, 'Dboth')
    loss.accumulate_gradients(phase='Dboth', **stubs.batch(4))
    for i, p in enumerate(nets['D'].parameters()):
        np.testing.assert_allclose(got[f'p{i}'], p.grad.numpy(), rtol=2e-4, atol=1e-6)


def _diverging_worker(rank, world, init_file, out_file, gather=False):
    sys.path.insert(0, PKG)
    from training import ddp
    torch.set_num_threads(2)
    dist.init_process_group('gloo', init_method=f'file://{init_file}', rank=rank, world_size=world)
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(n)) for n in (5, 7, 3, 6, 4, 8)]
    bucket = ddp.GradBucket(ps, segments=3, gather=gather)
    assert len(bucket.seg_range) == 3
    logs = []
    for it in range(2):
        bucket.begin()
        bucket.last_round()
        # rank 1 never uses ps[3] (a middle segment), rank 0 never uses ps[0] (the LAST segment); nobody uses ps[5] in iteration 1
        used = [i for i in range(6) if not (rank == 1 and i == 3) and not (rank == 0 and i == 0) and not (it == 1 and i == 5)]
        sum(((rank + 1.0) * (i + 1.0) * ps[i]).sum() for i in used).backward()
        assert bucket.finish() is True
        logs.append(list(bucket.launch_log))
        assert [k for k, _ in bucket.launch_log] == [0, 1, 2]            # every rank: the same collectives in the same order
        for i, p in enumerate(ps):
            users = [r for r in range(world) if not (r == 1 and i == 3) and not (r == 0 and i == 0) and not (it == 1 and i == 5)]
            if not users:
                assert p.grad is None
            else:
                want = sum((r + 1.0) * (i + 1.0) for r in users) / world
                assert torch.allclose(p.grad, torch.full_like(p.grad, want)), (it, i, p.grad, want)
    if rank == 0:
        np.savez(out_file, hook0=np.array([sum(1 for _, w in l if w == 'hook') for l in logs]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('gather', [False, True], ids=['grad_views', 'gathered'])
def test_collective_order_does_not_depend_on_which_parameters_got_gradients(gather):
    """ranks whose graphs differ (a parameter unused on one rank only) must still issue the same
    all-reduces in the same order -- segment 0, 1, 2 and nothing else -- and end with the right averaged gradients; a parameter
    nobody used keeps grad None.  (With the round-2 protocol this test pairs a segment all-reduce with the flags all-reduce.)"""
    with tempfile.TemporaryDirectory() as tmp:
        init_file, out_file = os.path.join(tmp, 'rdzv'), os.path.join(tmp, 'o.npz')
        mp.spawn(_diverging_worker, args=(2, init_file, out_file, gather), nprocs=2, join=True)
        assert np.load(out_file)['hook0'].tolist() == [2, 0]            # rank 0: segments 0, 1 from hooks in iteration 0 (ps[0] unused -> segment 2 in finish); iteration 1: ps[5] unused -> none


def _step_worker(rank, world, init_file, out_file):
    sys.path.insert(0, PKG)
    from training.training_step import TrainingStep
    torch.set_num_threads(2)
    if world > 1:
        dist.init_process_group('gloo', init_method=f'file://{init_file}', rank=rank, world_size=world)
    nets = stubs.build()
    G_parts = {k: v for k, v in nets.items() if k.startswith('G_')}
    step = TrainingStep(G_parts, nets['D'], nets['D_parsing'], _loss(nets), batch_size=4)
    full = stubs.batch(4)
    mine = {k: v[rank::world] for k, v in full.items()}
    for _ in range(5):                                              # iterations 0 and 4 contain a Greg phase
        step.run([mine])
    if rank == 0:
        g_opt = step.phases[0].opt
        st = [g_opt.state[p] for p in g_opt.param_groups[0]['params'] if p in g_opt.state]
        np.savez(out_file, steps=np.array([float(s_['step']) for s_ in st]), v=np.array([float(s_['exp_avg_sq'].sum()) for s_ in st]),
                 w=np.array([float(p.double().sum()) for p in nets['G_synthesis'].parameters()]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_training_step_is_world_size_invariant_through_greg():
    """a Greg phase produces no gradient; it must not step Adam on zero gradients when world > 1.  Five
    iterations on 1 rank and on 2 ranks (same global batch) must leave identical Adam step counts and (up to the
    reduction order) identical second-moment statistics and weights."""
    res = {}
    for world in (1, 2):
        with tempfile.TemporaryDirectory() as tmp:
            init_file, out_file = os.path.join(tmp, 'rdzv'), os.path.join(tmp, 'o.npz')
            if world == 1:
                _step_worker(0, 1, init_file, out_file)
            else:
                mp.spawn(_step_worker, args=(world, init_file, out_file), nprocs=world, join=True)
            res[world] = dict(np.load(out_file))
    assert np.array_equal(res[1]['steps'], res[2]['steps']) and res[1]['steps'].max() == 5     # Gmain stepped 5 times, Greg never
    np.testing.assert_allclose(res[2]['v'], res[1]['v'], rtol=1e-2)         # (stepping on zero gradients would show as ~20 %)
    np.testing.assert_allclose(res[2]['w'], res[1]['w'], rtol=2e-3, atol=1e-4)        # two half-batch means vs one full-batch mean: reduction order only
