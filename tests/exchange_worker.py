"""Child process of tests/test_exchange_gpu.py (not a test): one rank of a reduced-width training run on cuda:0 -- product discriminators (HIP operators, equalised-LR
convolutions, R1 double backward), stub generator, the 8-phase TrainingStep of training/training_step.py -- that writes rank 0's weights and Adam statistics.

    python exchange_worker.py OUT.npz --backend nccl|gloo|none --rank R --world W --port P --iters N [--sync-debug] [--graphs]

`--backend none`: no process group (the plain single-process step).  `--backend nccl --world 1` with PG_FORCE_EXCHANGE=1 in the environment: the multi-rank protocol of
training/ddp.py on a one-rank RCCL group.  `--backend gloo --world 2`: two ranks sharing GPU 0, CUDA tensors through gloo, rank r takes samples r::2 of the batch.
`--sync-debug`: torch.cuda.set_sync_debug_mode('error') around every eager phase from iteration 1 on -- any host synchronisation inside a phase raises."""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, 'pasta-gan-plusplus_amd'), os.path.join(HERE, 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('out')
    ap.add_argument('--backend', default='none')
    ap.add_argument('--rank', type=int, default=0)
    ap.add_argument('--world', type=int, default=1)
    ap.add_argument('--port', type=int, default=29533)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--sync-debug', action='store_true')
    ap.add_argument('--graphs', action='store_true')
    a = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    from torch_utils import custom_ops
    custom_ops.verbosity = 'none'
    import stubs
    from detgen import fill_module_
    from training import networks as PN
    from training.loss import StyleGAN2Loss
    from training.training_step import TrainingStep
    from training import ddp
    from oracle import network_ref as NR            # (test infrastructure: the deterministic initial weights come from the oracle's modules)
    dev = 'cuda:0'
    torch.cuda.set_device(0)
    if a.backend != 'none':
        dist.init_process_group(a.backend, init_method=f'tcp://127.0.0.1:{a.port}', rank=a.rank, world_size=a.world)
    torch.manual_seed(0)
    nets = stubs.build(dev)
    dkw = lambda ch: dict(c_dim=6, img_resolution=16, img_channels=ch, channel_base=256, channel_max=32, conv_clamp=256, mapping_kwargs=dict(num_layers=1),
                          epilogue_kwargs=dict(mbstd_group_size=2))
    for name, ch in (('D', 6), ('D_parsing', 10)):
        ref = fill_module_(NR.Discriminator(**dkw(ch)), f'xw.{name}.')
        d = PN.Discriminator(**dkw(ch))
        d.load_state_dict(ref.state_dict(), strict=False)
        nets[name] = d.to(dev).train()
    loss = StyleGAN2Loss(device=torch.device(dev), **nets, style_mixing_prob=0, r1_gamma=10, l1_weight=50, mask_weight=1.0)
    G_parts = {k: v for k, v in nets.items() if k.startswith('G_')}
    step = TrainingStep(G_parts, nets['D'], nets['D_parsing'], loss, batch_size=a.batch, graphs=a.graphs)
    full = stubs.batch(a.batch, dev)
    mine = {k: v[a.rank::a.world].contiguous() for k, v in full.items()}
    buckets = {id(ph.bucket): ph.bucket for ph in step.phases}.values()
    hooks, finishes, phases_run = 0, 0, 0
    if a.sync_debug:
        inner = step._phase

        def guarded(ph, rounds):
            if step.batch_idx == 0:                 # first iteration: plugin loading, kernel attributes, pinned allocations
                return inner(ph, rounds)
            torch.cuda.set_sync_debug_mode('error')
            try:
                return inner(ph, rounds)
            finally:
                torch.cuda.set_sync_debug_mode('default')
        step._phase = guarded
    for _ in range(a.iters):
        step.run([mine])
        for b in buckets:
            hooks += sum(1 for _, who in b.launch_log if who == 'hook')
            finishes += sum(1 for _, who in b.launch_log if who == 'finish')
            b.launch_log = []
    torch.cuda.synchronize()
    if a.rank == 0:
        out = {f'{k}.{n}': p.detach().float().cpu().numpy() for k, m in nets.items() for n, p in m.named_parameters()}
        for i, ph in enumerate(step.phases):
            opt = ph.opt
            if hasattr(opt, 'exp_avg_sq'):
                out[f'adam{i}.v'] = opt.exp_avg_sq.detach().cpu().numpy()
                out[f'adam{i}.steps'] = opt.steps[0].detach().cpu().numpy()
        b0 = step.phases[0].bucket
        np.savez(a.out, __hooks=np.array(hooks), __finishes=np.array(finishes), __exchange=np.array(int(b0.exchange)), __device_flags=np.array(int(b0.device_flags)),
                 __comm_cus=np.array(-1 if ddp._reserved[0] is None else ddp._reserved[0]), __graphed=np.array(len(step.graphed_phases())), **out)
    if a.backend != 'none':
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
