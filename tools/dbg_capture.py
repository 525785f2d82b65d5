import sys, os, traceback
ROOT='/root/repo'
for p in (ROOT, ROOT+'/pasta-gan-plusplus_amd', ROOT+'/tests/golden', ROOT+'/tests'):
    sys.path.insert(0, p)
import torch
from torch_utils import custom_ops
custom_ops.verbosity='none'
import stubs
from detgen import fill_module_
from training import networks as PN
from training.loss import StyleGAN2Loss
from training.training_step import TrainingStep
from oracle import network_ref as NR
DEV='cuda'
def _d_kw(img_channels):
    return dict(c_dim=6, img_resolution=16, img_channels=img_channels, channel_base=256, channel_max=32, conv_clamp=256,
                mapping_kwargs=dict(num_layers=1), epilogue_kwargs=dict(mbstd_group_size=2))
nets = stubs.build(DEV)
for name, ch in (('D', 6), ('D_parsing', 10)):
    d = PN.Discriminator(**_d_kw(ch)); nets[name] = d.to(DEV).train()
loss = StyleGAN2Loss(device=torch.device(DEV), **nets, style_mixing_prob=0, r1_gamma=10, l1_weight=50, mask_weight=1.0)
G_parts = {k: v for k, v in nets.items() if k.startswith('G_')}
step = TrainingStep(G_parts, nets['D'], nets['D_parsing'], loss, batch_size=4, graphs=True)
b = stubs.batch(4, DEV)
step.run([b])          # eager
torch.cuda.synchronize()
import warnings
for idx, ph in enumerate(step.phases):
    if ph.interval != 1: continue
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            step._phase(ph, [b])
        print(ph.name, 'captured OK')
    except Exception:
        print(ph.name, 'FAILED'); traceback.print_exc(limit=25)
        torch.cuda.synchronize()
        break
