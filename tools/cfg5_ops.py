"""Dev tool (GPU box): which Python call sites launch the small torch kernels of the config-5 forward (torch.profiler, with stacks)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'pasta-gan-plusplus_amd'), os.path.join(ROOT, 'tests'), ROOT]
import torch
import bench
from training import networks
from detgen import fill_module_
dev = torch.device('cuda', 0)
net = fill_module_(networks.SynthesisStack(num_fp16_res=8, half_dtype=torch.bfloat16, channel_max=1024, **bench.CFG5), 'cfg5.').to(dev).eval()
ws = torch.randn([4, net.num_ws, 512], device=dev)
with torch.no_grad():
    for _ in range(3):
        net(ws, noise_mode='const')
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True) as prof:
        net(ws, noise_mode='const')
    torch.cuda.synchronize()
names = collections.Counter(ev.name for ev in prof.events())
for name, c in names.most_common(45):
    print(f'{c:4d}  {name}')
