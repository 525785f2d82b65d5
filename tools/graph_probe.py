"""Dev tool (GPU box): config-2 step (fp32 512^2 synthesis forward, N=8) as eager launches vs ONE hipGraph replay.
    python tools/graph_probe.py [steps]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import torch
import bench
bench.torch = torch
from training import networks
from training.graphed import GraphedForward

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda', 0)
net = bench.init_weights(networks.SynthesisNetworkFull_v18(**bench.CFG2)).to(dev).eval()
inp = bench.make_inputs(8, dev, seed=100)
names = ['ws', 'pose_feat', 'du', 'dl', 'mu', 'ml']
cat_keys = sorted(inp['cat_feat'])
flat = [inp[k] for k in names] + [inp['cat_feat'][k] for k in cat_keys]


def fn(*a):
    d = dict(zip(names, a[:len(names)]))
    d['cat_feat'] = dict(zip(cat_keys, a[len(names):]))
    return bench.run_net(net, d)


def timed(f):
    for _ in range(3):
        out = f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = f()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps, out


with torch.no_grad():
    t_eager, ref = timed(lambda: fn(*flat))
    ref = [t.clone() for t in ref if torch.is_tensor(t)]
    g = GraphedForward(fn, flat, warmup=2)
    t_graph, out = timed(lambda: g(*flat))
    out = [t for t in out if torch.is_tensor(t)]
print(f'eager {t_eager:.3f} ms/step | one hipGraph replay {t_graph:.3f} ms/step | outputs identical: {all(torch.equal(a, b) for a, b in zip(ref, out))}')
