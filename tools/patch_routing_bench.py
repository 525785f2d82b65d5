"""Dev tool (GPU box): throughput of training.patch_routing.normalize (row f3) on synthetic 512x512 samples, inputs resident on the GPU."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np, torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from training import patch_routing as P
from test_patch_routing import keypoints
rng = np.random.default_rng(0)
up, lo = (torch.from_numpy(rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)).cuda() for _ in range(2))
um = torch.zeros(512, 512, 3, dtype=torch.uint8, device='cuda'); um[90:310, 150:370] = 255
lm = torch.zeros(512, 512, 3, dtype=torch.uint8, device='cuda'); lm[270:505, 190:330] = 255
kps = [(keypoints(rng, 8.0), keypoints(rng, 8.0)) for _ in range(16)]
for ckp, pkp in kps[:3]:
    P.normalize(up, lo, um, lm, None, ckp, pkp, 2)
torch.cuda.synchronize()
t0 = time.perf_counter()
for rep in range(4):
    for ckp, pkp in kps:
        out = P.normalize(up, lo, um, lm, None, ckp, pkp, 2)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 64
t1 = time.perf_counter()
for ckp, pkp in kps:
    for ii, bp in enumerate(P.BPARTS):
        P.get_crop(ckp, bp, np.array([[128, 128]]), 512, 512, 0.5); P.get_crop(pkp, bp, np.array([[128, 128]]), 512, 512, 0.5)
host = (time.perf_counter() - t1) / 16
print(f'normalize: {dt * 1e3:.2f} ms per sample ({1 / dt:.0f} samples/s), of which host keypoint geometry {host * 1e3:.2f} ms')
