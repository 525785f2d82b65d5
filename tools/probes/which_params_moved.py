"""Dev probe (GPU box): one full iteration of the 8-phase step on the full-width networks, then the names of the parameters that did NOT move."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden')); sys.path.insert(0, ROOT)
import torch
from training import networks as PN
from training.loss import StyleGAN2Loss
from training.training_step import TrainingStep
from training.synthetic import fill_module_, det_tensor
DEV = 'cuda'
n = int(os.environ.get('N', '2'))
g_kw = dict(z_dim=0, c_dim=512, w_dim=512, img_resolution=512, img_channels=3, mapping_kwargs=dict(num_layers=1), synthesis_kwargs=dict(channel_base=32768, channel_max=512, conv_clamp=256))
d_kw = lambda ch: dict(c_dim=512, img_resolution=512, img_channels=ch, channel_base=32768, channel_max=512, conv_clamp=256, num_fp16_res=int(os.environ.get('FP16', '0')), epilogue_kwargs=dict(mbstd_group_size=min(n, 4)))
net = dict(G=fill_module_(PN.GeneratorFull_v20(**g_kw), 'c4w.G.', noise_strength=0.0), D=fill_module_(PN.Discriminator(**d_kw(6)), 'c4w.D.'), D_parsing=fill_module_(PN.Discriminator(**d_kw(10)), 'c4w.DP.'))
sc = float(os.environ.get('OUT_SCALE', '1'))
with torch.no_grad():
    for k in ('D', 'D_parsing'):
        net[k].b4.out.weight.mul_(sc); net[k].b4.out.bias.mul_(sc)
net = {k: m.to(DEV).train() for k, m in net.items()}
with torch.no_grad():
    img = det_tensor('c4w.real', [n, 3, 512, 512], 'uniform').to(DEV)
    print('D logits on [real, real]:', net['D'](torch.cat([img, img], 1), torch.zeros([n, 512], device=DEV)).flatten().tolist())
parts = lambda g: dict(G_mapping=g.mapping, G_synthesis=g.synthesis, G_const_encoding=g.const_encoding, G_style_encoding=g.style_encoding)
loss = StyleGAN2Loss(device=torch.device(DEV), **parts(net['G']), D=net['D'], D_parsing=net['D_parsing'], style_mixing_prob=0, r1_gamma=10, l1_weight=50, mask_weight=1.0)
u = lambda name, *shape: det_tensor('c4w.' + name, shape, 'uniform')
batch = dict(real_img=u('real', n, 3, 512, 512), gen_z=torch.zeros([n, 0]), style_input=u('style', n, 45, 128, 128), retain=u('retain', n, 6, 512, 512),
             pose=u('pose', n, 5, 512, 512), denorm_upper_input=u('du', n, 3, 512, 512), denorm_lower_input=u('dl', n, 3, 512, 512),
             denorm_upper_mask=det_tensor('c4w.mu', [n, 1, 512, 512], 'blockmask'), denorm_lower_mask=det_tensor('c4w.ml', [n, 1, 512, 512], 'blockmask'),
             gt_parsing=det_tensor('c4w.gt', [n, 1, 512, 512], 'labels7'))
step = TrainingStep(parts(net['G']), net['D'], net['D_parsing'], loss, batch_size=n)
seen = {}
def observer(event, ph):
    if event == 'gradients':
        own = {'G': 'G', 'D': 'D', 'D_': 'D_parsing'}['D_' if ph.name.startswith('D_parsing') else ph.name[0]]
        seen[ph.name] = {pn: (None if p.grad is None else float(p.grad.abs().max())) for pn, p in net[own].named_parameters()}
step.observer = observer
before = {k: {pn: p.detach().clone() for pn, p in m.named_parameters()} for k, m in net.items()}
step.run([{k: v.to(DEV) for k, v in batch.items()}])
torch.cuda.synchronize()
for k, m in net.items():
    still = [pn for pn, p in m.named_parameters() if torch.equal(p, before[k][pn])]
    print(k, len(still), 'of', len(before[k]), 'unchanged:', still[:8])
for ph, d in seen.items():
    zero = [pn for pn, v in d.items() if v is None or v == 0.0]
    print(ph, 'zero / None gradients:', len(zero), zero[:6])
