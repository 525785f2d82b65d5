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
def run(fold, fp16res, edit=True):
    os.environ['PG_GAIN_FOLD'] = '1' if fold else '0'
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
    if edit:
        with torch.no_grad():
            nets['D'].b8.conv0.weight.mul_(1.25)
    b = stubs.batch(4, DEV)
    for _ in range(5):
        step.run([b])
    torch.cuda.synchronize()
    return {f'{k}.{n}': p.detach().clone() for k, m in nets.items() for n, p in m.named_parameters()}
def worst(a, b):
    w = max(((float((a[k] - b[k]).abs().max()), k) for k in a))
    return w
for fp16res in (0, 1):
    p1, p2, f1 = run(False, fp16res), run(False, fp16res), run(True, fp16res)
    print('fp16res', fp16res, 'plain vs plain', worst(p1, p2), '| fold vs plain', worst(f1, p1))
