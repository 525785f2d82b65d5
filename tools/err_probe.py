import sys, os
sys.path.insert(0, '/root/repo/pasta-gan-plusplus_amd'); sys.path.insert(0, '/root/repo/tests/golden'); sys.path.insert(0, '/root/repo')
import torch
from detgen import fill_module_, synthesis_inputs
from training import networks as PN
from oracle import network_ref as NR
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
kw = dict(w_dim=512, img_resolution=512, img_channels=3, channel_base=32768, channel_max=512, conv_clamp=256)
ref_net = fill_module_(NR.SynthesisNetworkFull_v18(**kw), 'cfg2.').eval()
net = PN.SynthesisNetworkFull_v18(**kw); net.load_state_dict(ref_net.state_dict()); net = net.cuda().eval()
inp = synthesis_inputs(1, labels=True)
args = lambda f: (f(inp['ws']), f(inp['pose_feat']), {k: f(v) for k, v in inp['cat_feat'].items()}, f(inp['denorm_upper_input']),
                  f(inp['denorm_lower_input']), f(inp['denorm_upper_mask']), f(inp['denorm_lower_mask']), f(inp['gt_parsing']))
with torch.no_grad():
    ref = ref_net(*args(lambda t: t), noise_mode='const')
    for algo in ('winograd(auto)', 'direct'):
        os.environ['PG_CONV_ALGO'] = 'auto' if algo.startswith('w') else 'direct'
        for m in net.modules():
            if hasattr(m, '_cache'): m._cache._store.clear()
        out = net(*args(lambda t: t.cuda()), noise_mode='const')
        for nm, a, b in zip(('img', 'finetune_img', 'pred_parsing'), out, ref):
            s = float(b.abs().max()); d = float((a.cpu().double() - b.double()).abs().max())
            print(f'{algo:15s} {nm:13s} max-abs delta {d:.3e}  output range {s:.3e}  relative {d/s:.2e}', flush=True)
