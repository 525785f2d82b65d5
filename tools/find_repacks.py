"""Dev tool (GPU box): report which call sites re-pack conv weights after warm-up (they should all be cached)."""
import sys, os, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden')); sys.path.insert(0, ROOT)
import torch
import bench
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma
from training import networks
net = bench.init_weights(networks.SynthesisNetworkFull_v18(**bench.CFG2)).cuda().eval()
inp = bench.make_inputs(2, 'cuda', 0)
with torch.no_grad():
    bench.run_net(net, inp); bench.run_net(net, inp)
    sites = collections.Counter()
    orig = conv2d_mfma.pack_weight
    def spy(*a, **k):
        st = traceback.extract_stack()[:-1]
        sites[' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}:{f.name}' for f in st[-5:])] += 1
        return orig(*a, **k)
    conv2d_mfma.pack_weight = spy
    bench.run_net(net, inp)
for k, v in sites.most_common():
    print(v, k)
print('total repacks in one warm forward:', sum(sites.values()))
