"""Dev tool (GPU box): config-2 forward as ONE stream at N=8 vs two streams at N=4 each (latency-bound low-resolution kernels of
one half overlapping the matrix-bound kernels of the other)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import torch
import bench
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from training import networks
dev = torch.device('cuda', 0)
net = bench.init_weights(networks.SynthesisNetworkFull_v18(**bench.CFG2)).to(dev).eval()
inp = bench.make_inputs(8, dev, seed=0)

def sl(v, a, b):
    return {k: sl(x, a, b) for k, x in v.items()} if isinstance(v, dict) else v[a:b].contiguous()
halves = [sl(inp, 0, 4), sl(inp, 4, 8)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def one():
    return bench.run_net(net, inp)

def two():
    cur = torch.cuda.current_stream()
    outs = []
    for s, h in zip(streams, halves):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(bench.run_net(net, h))
    for s in streams:
        cur.wait_stream(s)
    return outs

def timeit(fn, n=10):
    with torch.no_grad():
        for _ in range(3): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for rep in range(2):
    print(f'one stream N=8: {timeit(one):.2f} ms   two streams 2 x N=4: {timeit(two):.2f} ms', flush=True)
