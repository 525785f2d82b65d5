"""Dev tool (GPU box): the config-2 step (and the config-3 generator) repeated under load; every kernel on the path is deterministic,
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
