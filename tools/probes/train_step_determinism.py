"""Dev probe (GPU box): which phase / parameter of a training iteration is not bit-reproducible between two identical runs?  Stub generator + product
discriminators with one half-precision block (tests' set-up); gradients captured by the step's observer when they are final."""
import os, sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/pasta-gan-plusplus_amd', '/root/repo/tests', '/root/repo/tests/golden']
import importlib
t = importlib.import_module('test_hip_parity')
import stubs
from training import networks as PN
from training.loss import StyleGAN2Loss
from training.training_step import TrainingStep
from oracle import network_ref as NR
DEV = 'cuda'


def run(fp16res, iters=2):
    torch.manual_seed(0)
    nets = stubs.build(DEV)
    for name, ch in (('D', 6), ('D_parsing', 10)):
        ref = t.fill_module_(NR.Discriminator(**t._d_kw(ch)), f'gf.{name}.')
        d = PN.Discriminator(**t._d_kw(ch), num_fp16_res=fp16res)
        d.load_state_dict(ref.state_dict(), strict=False)
        nets[name] = d.to(DEV).train()
    loss = StyleGAN2Loss(device=torch.device(DEV), **nets, style_mixing_prob=0, r1_gamma=10, l1_weight=50, mask_weight=1.0)
    G_parts = {k: v for k, v in nets.items() if k.startswith('G_')}
    step = TrainingStep(G_parts, nets['D'], nets['D_parsing'], loss, batch_size=4, graphs=False)
    names = {id(p): f'{k}.{n}' for k, m in nets.items() for n, p in m.named_parameters()}
    log = []

    def obs(event, ph):
        if event == 'gradients':
            log.append((step.batch_idx, ph.name, {names[id(p)]: p.grad.detach().clone() for p in ph.bucket.params if p.grad is not None}))
    step.observer = obs
    b = stubs.batch(4, DEV)
    for _ in range(iters):
        step.run([b])
    torch.cuda.synchronize()
    return log


for fp16res in (1,):
    a, b = run(fp16res), run(fp16res)
    for (i, ph, ga), (_, _, gb) in zip(a, b):
        bad = [(k, float((ga[k] - gb[k]).abs().max() / (ga[k].abs().max() + 1e-30))) for k in ga if not torch.equal(ga[k], gb[k])]
        print(f'iteration {i} phase {ph}: {len(bad)} of {len(ga)} gradient tensors differ', bad[:6])
