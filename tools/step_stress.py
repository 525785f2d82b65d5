"""Dev tool (GPU box): the config-2 step and the config-5 bf16 stack repeated under load; every kernel on the path is deterministic,
so any output that differs from the first pass is a race (that is how the F(4x4) tail race of round 3 showed up: 20 % of full-size
launches wrong while every small test passed).
    python tools/step_stress.py [passes]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import torch
import bench
bench.torch = torch
from training import networks

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device('cuda', 0)
net = bench.init_weights(networks.SynthesisNetworkFull_v18(**bench.CFG2)).to(dev).eval()
for n in (8, 3):
    inp = bench.make_inputs(n, dev, seed=100 + n)
    other = torch.randn(64, 64, 256, 256, device=dev)
    with torch.no_grad():
        ref = [t.clone() for t in bench.run_net(net, inp) if torch.is_tensor(t)]
        bad = 0
        for it in range(passes):
            if it % 4 == 1:
                other.mul_(1.0001)                       # different cache / clock state between passes
            out = [t for t in bench.run_net(net, inp) if torch.is_tensor(t)]
            same = [bool(torch.equal(a, b)) for a, b in zip(ref, out)]
            if not all(same):
                bad += 1
                if bad <= 3:
                    d = [float((a - b).abs().max()) for a, b in zip(ref, out)]
                    print(f'   pass {it}: outputs differ from the first pass: max |d| per output {d}', flush=True)
    print(f'config 2, N={n}: {bad} of {passes} passes differ from the first ({len(ref)} outputs compared bit for bit)', flush=True)

# config 5: the bf16 1024^2 stack (two-role 16-bit kernel, split-K, four-phase launches), fp16 as well
from detgen import fill_module_
for half in (torch.bfloat16, torch.float16):
    stack = fill_module_(networks.SynthesisStack(num_fp16_res=8, half_dtype=half, channel_max=1024, **bench.CFG5), 'cfg5.').to(dev).eval()
    ws = torch.randn([4, stack.num_ws, 512], generator=torch.Generator().manual_seed(3)).to(dev)
    other = torch.randn(64, 64, 256, 256, device=dev)
    with torch.no_grad():
        ref = stack(ws, noise_mode='const').clone()
        bad = 0
        for it in range(passes):
            if it % 4 == 1:
                other.mul_(1.0001)
            out = stack(ws, noise_mode='const')
            if not torch.equal(ref, out):
                bad += 1
                if bad <= 3:
                    print(f'   pass {it}: max |d| {float((ref.float() - out.float()).abs().max()):.3e}', flush=True)
    print(f'config 5 ({half}), N=4: {bad} of {passes} passes differ from the first', flush=True)
