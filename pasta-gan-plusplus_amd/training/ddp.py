"""Data-parallel gradient exchange for the training step, MI355X-first.

The reference wraps six sub-modules in ``torch.nn.parallel.DistributedDataParallel`` with
``find_unused_parameters=True`` (training_loop_fullbody.py:451-460) and lets NCCL all-reduce many per-module
buckets from autograd hooks.  Here every phase owns ONE flat fp32 bucket covering all parameters its optimizer
steps (Gmain: 43 M floats = 172 MB; each discriminator: 31.6 M = 126 MB).  After the local backward of the last
accumulation round the phase's gradients are packed into the bucket (parameters that received no gradient --
e.g. ``synthesis.b8.const`` -- contribute zeros, which replaces DDP's unused-parameter graph walk), summed
across ranks by a single RCCL collective over xGMI, scaled by 1/world, and unpacked in place.  One large
collective per phase suits the point-to-point xGMI mesh (7 links per GPU): RCCL can split 126-172 MB over all
links, instead of serialising ~6 smaller rings.  ``reduce_scatter`` + ``all_gather`` is used when the bucket
divides evenly (each rank reduces 1/world of the bucket), otherwise a plain ``all_reduce``.
"""

import torch
import torch.distributed as dist


class GradBucket:
    def __init__(self, params, group=None):
        self.params = [p for p in params]
        assert self.params, 'empty bucket'
        self.group = group
        self.sizes = [p.numel() for p in self.params]
        self.total = sum(self.sizes)
        dev = self.params[0].device
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.world = world
        self.padded = (self.total + world - 1) // world * world            # even shards for reduce_scatter
        self.flat = torch.zeros([self.padded], dtype=torch.float32, device=dev)

    def pack(self):
        off = 0
        for p, n in zip(self.params, self.sizes):
            dst = self.flat[off:off + n]
            if p.grad is None:
                dst.zero_()
            else:
                dst.copy_(p.grad.reshape(-1))
            off += n
        return self.flat

    def unpack(self):
        off = 0
        for p, n in zip(self.params, self.sizes):
            g = self.flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n

    def all_reduce_mean(self):
        """Average the gradients of this bucket's parameters over all ranks (no-op for a single rank)."""
        if self.world == 1:
            return
        self.pack()
        if self.flat.is_cuda:
            shard = torch.empty([self.padded // self.world], dtype=torch.float32, device=self.flat.device)
            dist.reduce_scatter_tensor(shard, self.flat, op=dist.ReduceOp.SUM, group=self.group)
            shard.mul_(1.0 / self.world)
            dist.all_gather_into_tensor(self.flat, shard, group=self.group)
        else:                                                               # gloo (CPU tests): plain all-reduce
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.mul_(1.0 / self.world)
        self.unpack()


def broadcast_parameters(modules, src=0, group=None):
    """Make every rank start from rank `src`'s weights and buffers (the reference relies on equal seeds and checks with
    check_ddp_consistency, misc.py:184-196)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src=src, group=group)
