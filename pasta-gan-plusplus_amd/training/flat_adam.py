"""nan_to_num + Adam of a training phase as ONE native launch over flat buffers (csrc/optim.hip, pg_adam_flat_step).

The reference cleans every gradient with torch.nan_to_num and calls torch.optim.Adam.step() per phase (training/training_loop_fullbody.py:632-639):
on the GPU that is ~120 multi-tensor launches and ~11 passes over the phase's parameters.  `training.ddp.GradBucket` already keeps the phase's gradients
in one flat fp32 bucket whose slices are the parameters' .grad; `FlatAdam` gives the parameters and both moments the SAME flat layout (the parameters'
.data become views of `flat_p`), so one kernel reads p, g, m, v once and writes g (cleaned), m, v, p.  Which parameters received a gradient on any
rank is read from `bucket.alive` on the device -- no host sync (`GradBucket.device_flags`); untouched parameters keep moments and step count, as
torch's Adam does for `grad is None`.  GPU tensors only; the CPU path of the training step keeps torch.optim.Adam.
"""

import ctypes

import torch

from torch_utils.ops import _native as nat
from torch_utils.ops import conv2d_mfma


class FlatAdam:
    def __init__(self, bucket, lr, betas, eps, nan=0.0, posinf=1e5, neginf=-1e5, share_params_with=None):
        """`share_params_with`: another FlatAdam over the SAME bucket whose flat parameter buffer this one steps too, with moments and step counts of its
        own -- the reference lists D_parsing twice in its phase table (training_loop_fullbody.py:470-471), i.e. two optimizers over one module."""
        assert bucket.flat.is_cuda, 'FlatAdam is the GPU path; CPU tensors keep torch.optim.Adam'
        self.bucket, self.lr, self.betas, self.eps = bucket, float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.nan, self.posinf, self.neginf = float(nan), float(posinf), float(neginf)
        lib = conv2d_mfma._init().lib
        vp, i, f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        lib.pg_adam_flat_chunk.restype = i
        lib.pg_adam_flat_step.restype = i
        lib.pg_adam_flat_step.argtypes = [vp, vp, vp, vp, vp, i, vp, vp, vp, f, f, f, f, f, f, f, vp]
        self._lib = lib
        dev = bucket.flat.device
        assert share_params_with is None or share_params_with.bucket is bucket
        self.flat_p = torch.zeros_like(bucket.flat) if share_params_with is None else share_params_with.flat_p
        self.exp_avg = torch.zeros_like(bucket.flat)
        self.exp_avg_sq = torch.zeros_like(bucket.flat)
        chunk = int(lib.pg_adam_flat_chunk())
        table = []
        with torch.no_grad():
            for idx, p in enumerate(bucket.params):
                off, n = bucket.offset[idx], p.numel()
                if share_params_with is None:
                    view = self.flat_p[off:off + n].view_as(p)
                    view.copy_(p.detach())
                    p.data = view                                    # the parameter now lives in the flat buffer (same Parameter object, same module)
                for c0 in range(0, n, chunk):
                    table.append((off + c0, min(chunk, n - c0), idx, 1 if c0 == 0 else 0))
        self.chunks = torch.tensor(table, dtype=torch.int32).to(dev)
        self.steps = [torch.zeros([len(bucket.params)], dtype=torch.float32, device=dev) for _ in range(2)]
        bucket.device_flags = True
        self.param_groups = [dict(params=list(bucket.params), lr=self.lr, betas=self.betas, eps=self.eps)]      # torch.optim's attribute: `step` reads lr / betas / eps from it, so schedulers that edit it act

    def step(self):
        b = self.bucket
        g0 = self.param_groups[0]                      # (inside a replayed hipGraph the values of the capture are frozen, like every other scalar kernel argument)
        self.lr, self.betas, self.eps = float(g0['lr']), (float(g0['betas'][0]), float(g0['betas'][1])), float(g0['eps'])
        src, dst = self.steps[0], self.steps[1]       # (fixed roles + a device-side copy back, not a host-side swap: inside a replayed hipGraph the pointers are frozen)
        with torch.cuda.device(b.flat.device):
            st = self._lib.pg_adam_flat_step(nat.ptr(self.flat_p), nat.ptr(b.flat), nat.ptr(self.exp_avg), nat.ptr(self.exp_avg_sq), nat.ptr(self.chunks),
                                             int(self.chunks.shape[0]), nat.ptr(b.alive), nat.ptr(src), nat.ptr(dst), self.lr, self.betas[0], self.betas[1], self.eps,
                                             self.nan, self.posinf, self.neginf, nat.stream_of(b.flat))
        nat.check(st, 'pg_adam_flat_step')
        src.copy_(dst)
        # the kernel wrote through raw pointers: move the version counters, so that everything keyed on them (the packed-weight caches) sees new weights
        torch.autograd.graph.increment_version(b.params)

    def state_for(self, p):
        """(step, exp_avg, exp_avg_sq) of one parameter, like torch.optim.Adam's `state[p]` (tests)."""
        idx = next(i for i, q in enumerate(self.bucket.params) if q is p)
        off, n = self.bucket.offset[idx], p.numel()
        return dict(step=self.steps[0][idx], exp_avg=self.exp_avg[off:off + n].view_as(p), exp_avg_sq=self.exp_avg_sq[off:off + n].view_as(p))
