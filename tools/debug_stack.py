"""Layer-by-layer comparison of the half-precision SynthesisStack with the fp32 oracle (dev tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pasta-gan-plusplus_amd'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import torch
from detgen import det_tensor, fill_module_
from training import networks as PN
from oracle import network_ref as NR
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
dt = dict(bf16=torch.bfloat16, fp16=torch.float16, f32=torch.float32)[sys.argv[1] if len(sys.argv) > 1 else 'bf16']
kw = dict(w_dim=64, img_resolution=128, img_channels=3, channel_base=4096, channel_max=256, conv_clamp=256)
ref = fill_module_(NR.SynthesisStack(**kw), 'stack.').eval()
net = PN.SynthesisStack(num_fp16_res=5, half_dtype=dt if dt != torch.float32 else torch.float16, **kw)
net.load_state_dict(ref.state_dict(), strict=False)
net = net.cuda().eval()
ws = det_tensor('stack.ws', [2, net.num_ws, 64])
force = dt == torch.float32
def rel(a, b):
    return float((a.float().cpu() - b).abs().max()) / float(b.abs().max())
with torch.no_grad():
    x = img = None; xr = imgr = None; start = 0
    for res in net.block_resolutions:
        blk, rb = getattr(net, f'b{res}'), getattr(ref, f'b{res}')
        w = ws[:, start:start + blk.num_conv + 1]
        wd = w.cuda()
        half = not force
        fmt = dict(dtype=dt if half else torch.float32, memory_format=torch.channels_last if half else torch.contiguous_format)
        if res == 8:
            xr = rb.const[None].expand(2, -1, -1, -1)
            x = blk.const.to(fmt['dtype'])[None].expand(2, -1, -1, -1).contiguous(memory_format=fmt['memory_format'])
            x = blk.conv1(x, wd[:, 0], noise_mode='const'); xr = rb.conv1(xr, w[:, 0], noise_mode='const')
            print(res, 'conv1', rel(x, xr))
        else:
            x0 = blk.conv0(x.to(**fmt), wd[:, 0], noise_mode='const'); xr0 = rb.conv0(xr, w[:, 0], noise_mode='const')
            print(res, 'conv0', rel(x0, xr0), 'given exact input:', rel(blk.conv0(xr.cuda().to(**fmt), wd[:, 0], noise_mode='const'), xr0))
            x = blk.conv1(x0, wd[:, 1], noise_mode='const'); xr = rb.conv1(xr0, w[:, 1], noise_mode='const')
            print(res, 'conv1', rel(x, xr), 'given exact input:', rel(blk.conv1(xr0.cuda().to(**fmt), wd[:, 1], noise_mode='const'), xr))
        y, _ = blk.torgb(x, wd[:, blk.num_conv]); yr, _ = rb.torgb(xr, w[:, blk.num_conv])
        print(res, 'torgb', rel(y, yr), 'given exact input:', rel(blk.torgb(xr.cuda().to(**fmt), wd[:, blk.num_conv])[0], yr))
        start += blk.num_conv
