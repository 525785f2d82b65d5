"""Tiny stand-in networks with the call signatures StyleGAN2Loss uses (loss_fullbody.py:75-114), so the loss / phase
orchestration can be pinned against the REFERENCE's own StyleGAN2Loss on CPU in seconds.  These are the build's own
test doubles (plain torch.nn), not part of the product and not derived from the reference networks."""

import torch
import torch.nn as nn

from detgen import det_tensor, fill_module_

RES, CDIM, WDIM, NUM_WS = 16, 6, 5, 3


class StubStyleEncoding(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(4, 3, 3, padding=1)
        self.fc = nn.Linear(3, CDIM)
        self.feat = nn.Conv2d(2, 3, 1)

    def forward(self, style_input, retain):
        c = self.fc(torch.tanh(self.conv(style_input)).mean(dim=(2, 3)))
        return c, [self.feat(retain)]


class StubConstEncoding(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(5, 4, 3, padding=1)

    def forward(self, pose):
        return torch.tanh(self.conv(pose))


class StubMapping(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc = nn.Linear(CDIM, WDIM)

    def forward(self, z, c, **_):
        return self.fc(c).unsqueeze(1).repeat(1, NUM_WS, 1)


class StubSynthesis(nn.Module):
    def __init__(self):
        super().__init__()
        self.mix = nn.Conv2d(4 + 3 + 3 + 3 + 1 + 1, 8, 3, padding=1)
        self.style = nn.Linear(WDIM, 8)
        self.rgb, self.rgb2, self.parse = nn.Conv2d(8, 3, 1), nn.Conv2d(8, 3, 1), nn.Conv2d(8, 7, 1)

    def forward(self, ws, pose_feat, cat_feats, du, dl, mu, ml, gt_parsing, **_):
        x = torch.cat([pose_feat, cat_feats[str(du.shape[-1])], du, dl, mu, ml], dim=1)
        h = torch.tanh(self.mix(x)) * (1 + self.style(ws[:, 0])[:, :, None, None])
        return torch.tanh(self.rgb(h)), torch.tanh(self.rgb2(h)), self.parse(h)


class StubD(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.c1 = nn.Conv2d(cin, 6, 3, padding=1, stride=2)
        self.c2 = nn.Conv2d(6, 6, 3, padding=1, stride=2)
        self.fc = nn.Linear(6, CDIM)

    def forward(self, img, c, **_):
        h = torch.nn.functional.softplus(self.c2(torch.nn.functional.softplus(self.c1(img)))).mean(dim=(2, 3))   # smooth => non-zero 2nd derivative for R1
        return (self.fc(h) * c).sum(dim=1, keepdim=True)


def build(device='cpu'):
    torch.manual_seed(0)
    nets = dict(G_mapping=StubMapping(), G_synthesis=StubSynthesis(), G_const_encoding=StubConstEncoding(),
                G_style_encoding=StubStyleEncoding(), D=StubD(6), D_parsing=StubD(10))
    for name, m in nets.items():
        fill_module_(m, f'stub.{name}.', bias_scale=0.3)
        m.to(device)
    return nets


def batch(n=4, device='cpu', res=None):
    RES = res or globals()['RES']                      # the goldens (g9) use the default 16; the full-width config-4 test passes 512
    b = dict(real_img=det_tensor('stub.real', [n, 3, RES, RES], 'uniform'), gen_z=torch.zeros([n, 0]),
             style_input=det_tensor('stub.style', [n, 4, RES, RES], 'uniform'), retain=det_tensor('stub.retain', [n, 2, RES, RES], 'uniform'),
             pose=det_tensor('stub.pose', [n, 5, RES, RES], 'uniform'), denorm_upper_input=det_tensor('stub.du', [n, 3, RES, RES], 'uniform'),
             denorm_lower_input=det_tensor('stub.dl', [n, 3, RES, RES], 'uniform'), denorm_upper_mask=det_tensor('stub.mu', [n, 1, RES, RES], 'mask'),
             denorm_lower_mask=det_tensor('stub.ml', [n, 1, RES, RES], 'mask'),
             gt_parsing=torch.from_numpy((det_tensor('stub.gt', [n, 1, RES, RES], 'uniform').numpy() * 3.49 + 3.5).round().clip(0, 6)).float())
    return {k: v.to(device) for k, v in b.items()}


PHASES = ['Gmain', 'Dmain', 'Dreg', 'Dboth', 'D_parsingmain', 'D_parsingreg', 'Gboth']


def grad_signature(nets):
    """{module.param: sum|grad|} over every stub parameter (0 where no gradient arrived)."""
    sig = {}
    for mname, m in nets.items():
        for pname, p in m.named_parameters():
            sig[f'{mname}.{pname}'] = 0.0 if p.grad is None else float(p.grad.double().abs().sum())
    return sig


def zero_grads(nets):
    for m in nets.values():
        for p in m.parameters():
            p.grad = None


def set_phase_trainable(nets, phase):
    """As the reference loop does (training_loop_fullbody.py:612-613, 632): only the phase's module requires grad."""
    owner = 'G' if phase.startswith('G') else ('D_parsing' if phase.startswith('D_parsing') else 'D')
    for name, m in nets.items():
        m.requires_grad_(name.startswith('G_') if owner == 'G' else name == owner)
