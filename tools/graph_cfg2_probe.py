"""GPU box: config 2 (the headline step) eager vs one hipGraph replay per step (training/graphed.py).  python tools/graph_cfg2_probe.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
import bench
bench.torch = torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from training import networks
from training.graphed import GraphedForward

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda', 0)
net = bench.init_weights(networks.SynthesisNetworkFull_v18(**bench.CFG2)).to(dev).eval()
inp = bench.make_inputs(8, dev, seed=0)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


with torch.no_grad():
    t_eager, out_e = timed(lambda: bench.run_net(net, inp))
    out_e = [o.clone() for o in out_e]
    flat, spec = [], []

    def walk(v, path):
        if torch.is_tensor(v):
            flat.append(v); spec.append(path)
        elif isinstance(v, dict):
            for k in v:
                walk(v[k], path + (k,))
    walk(inp, ())

    def fn(*ts):
        d = {}
        for t, path in zip(ts, spec):
            cur = d
            for k in path[:-1]:
                cur = cur.setdefault(k, {})
            cur[path[-1]] = t
        return bench.run_net(net, d)
    fwd = GraphedForward(fn, flat, warmup=2)
    t_graph, out_g = timed(lambda: fwd(*flat))
    t_eager2, _ = timed(lambda: bench.run_net(net, inp))
print(f'eager {t_eager:.3f} ms  graph {t_graph:.3f} ms  eager again {t_eager2:.3f} ms   ({8e3 / t_eager:.1f} / {8e3 / t_graph:.1f} images/s)')
print('identical outputs:', all(torch.equal(a, b) for a, b in zip(out_e, out_g)))
