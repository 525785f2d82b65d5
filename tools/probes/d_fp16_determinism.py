import os, sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/pasta-gan-plusplus_amd', '/root/repo/tests', '/root/repo/tests/golden']
import importlib
t = importlib.import_module('test_hip_parity')
from training import networks as PN
from oracle import network_ref as NR
torch.manual_seed(0)
for res, kw in ((16, t._d_kw(6)), (64, dict(c_dim=6, img_resolution=64, img_channels=6, channel_base=1024, channel_max=64, conv_clamp=256, mapping_kwargs=dict(num_layers=1), epilogue_kwargs=dict(mbstd_group_size=2)))):
    d = PN.Discriminator(**kw, num_fp16_res=1).cuda().train()
    x = torch.randn(4, 6, res, res, device='cuda'); c = torch.randn(4, 6, device='cuda')
    outs, grads = [], []
    for it in range(3):
        for p in d.parameters(): p.grad = None
        y = d(x, c)
        y.sum().backward()
        outs.append(y.detach().clone()); grads.append({n: p.grad.detach().clone() for n, p in d.named_parameters() if p.grad is not None})
    print('res', res, 'forward identical:', torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]))
    bad = [n for n in grads[0] if not (torch.equal(grads[0][n], grads[1][n]) and torch.equal(grads[0][n], grads[2][n]))]
    print('  parameters whose gradient differs between identical runs:', bad[:12])
    # R1-style double backward (loss_fullbody.py:247-256)
    g2 = []
    for it in range(3):
        for p in d.parameters(): p.grad = None
        xr = x.detach().requires_grad_(True)
        y = d(xr, c)
        gx, = torch.autograd.grad(y.sum(), xr, create_graph=True)
        (gx.square().sum([1, 2, 3]).mean() * 5).backward()
        g2.append({n: p.grad.detach().clone() for n, p in d.named_parameters() if p.grad is not None})
    bad = [n for n in g2[0] if not (torch.equal(g2[0][n], g2[1][n]) and torch.equal(g2[0][n], g2[2][n]))]
    print('  R1 double backward: parameters whose gradient differs between identical runs:', bad[:16])
