"""Data-parallel gradient exchange for the training step, MI355X-first.

The reference wraps six sub-modules in ``torch.nn.parallel.DistributedDataParallel`` with
``find_unused_parameters=True`` (training_loop_fullbody.py:451-460) and lets NCCL all-reduce many per-module
buckets from autograd hooks.  Here every phase owns ONE flat fp32 bucket covering all parameters its optimizer
steps (Gmain: 43 M floats = 172 MB; each discriminator: 31.6 M = 126 MB):

* the parameters' ``.grad`` tensors ARE views into the flat bucket (``begin()``), so autograd accumulates straight
  into it -- no pack / unpack copies and no second copy of any gradient;
* the bucket is cut into a few large segments in reverse parameter order (gradients arrive roughly last layer
  first).  A post-accumulate hook on every parameter counts its segment down; the moment a segment's last gradient
  has been enqueued, its sum-reduction is launched on a side stream behind an event, i.e. it runs over xGMI while
  the backward kernels of the earlier layers are still executing (what DDP's bucket hooks give the reference).
  Large segments suit the point-to-point xGMI mesh (7 links per GPU): RCCL splits 30-60 MB over all links;
* parameters that received no gradient on ANY rank keep ``grad = None`` -- exactly what the reference's optimizers
  see (DDP leaves unused parameters untouched): a tiny MAX all-reduce of the per-parameter "touched" flags decides,
  and ``any_touched`` tells the caller whether the phase produced a gradient at all (the Greg phase does not: no
  exchange, no optimizer step, so Adam's statistics do not depend on the world size).
"""

import torch
import torch.distributed as dist


class GradBucket:
    def __init__(self, params, group=None, segments=4):
        self.params = [p for p in params]
        assert self.params, 'empty bucket'
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        dev = self.params[0].device
        # reverse registration order: the last layers' gradients are produced first
        order = list(range(len(self.params)))[::-1]
        sizes = [self.params[i].numel() for i in order]
        self.total = sum(sizes)
        self.flat = torch.zeros([self.total], dtype=torch.float32, device=dev)
        self.views, off = [None] * len(self.params), 0
        self.offset = [0] * len(self.params)
        for i, n in zip(order, sizes):
            self.views[i] = self.flat[off:off + n].view_as(self.params[i])
            self.offset[i] = off
            off += n
        # segment boundaries on parameter boundaries, ~equal sizes
        nseg = max(1, min(int(segments), len(self.params)))
        target, self.seg_of, self.seg_range = self.total / nseg, [0] * len(self.params), []
        start, seg = 0, 0
        for k, (i, n) in enumerate(zip(order, sizes)):
            self.seg_of[i] = seg
            end = self.offset[i] + n
            if (end - start >= target and seg < nseg - 1) or k == len(order) - 1:
                self.seg_range.append((start, end))
                start, seg = end, seg + 1
        self.seg_members = [sum(1 for s in self.seg_of if s == k) for k in range(len(self.seg_range))]
        self.touched = torch.zeros([len(self.params)], dtype=torch.float32, device=dev)
        self._touched_host = [False] * len(self.params)
        self._pending, self._launched, self._works = [], [], []
        self._active = False
        self._comm_stream = torch.cuda.Stream(device=dev) if dev.type == 'cuda' else None
        self.any_touched = False
        for i, p in enumerate(self.params):
            was = p.requires_grad                            # the step freezes every module between phases; hooks need a leaf that requires grad
            p.requires_grad_(True)
            p.register_post_accumulate_grad_hook(self._make_hook(i))
            p.requires_grad_(was)

    # ------------------------------------------------------------------ per-phase protocol
    def begin(self):
        """Before the phase's backward passes: zero the bucket, point every .grad into it, arm the hooks."""
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v
        self._touched_host = [False] * len(self.params)
        self._pending = list(self.seg_members)
        self._launched = [False] * len(self.seg_range)
        self._works = []
        self._active = True
        self._sync_round = False

    def last_round(self):
        """Call before the LAST accumulation round's backward: from now on a finished segment may be exchanged."""
        self._sync_round = True
        self._pending = list(self.seg_members)

    def _make_hook(self, i):
        def hook(p):
            if not self._active:
                return
            self._touched_host[i] = True
            if self._sync_round and self.world > 1:
                k = self.seg_of[i]
                self._pending[k] -= 1
                if self._pending[k] == 0:
                    self._launch(k)
        return hook

    def _launch(self, k):
        if self._launched[k]:
            return
        self._launched[k] = True
        a, b = self.seg_range[k]
        seg = self.flat[a:b]
        if self._comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record()                                      # everything enqueued so far: this segment's gradients are among it
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ev)
                dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group)     # RCCL picks the algorithm for the 7-link mesh
                seg.mul_(1.0 / self.world)
        else:
            self._works.append((dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True), seg))

    def finish(self):
        """After the last backward: exchange what the hooks could not (segments holding unused parameters), decide which
        gradients exist anywhere, hand `None` back to the rest.  Returns True if the phase produced any gradient."""
        self._active = False
        flags = torch.tensor([1.0 if t else 0.0 for t in self._touched_host], dtype=torch.float32, device=self.flat.device)
        if self.world > 1:
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)
        alive = flags.cpu().tolist() if self.world > 1 else [1.0 if t else 0.0 for t in self._touched_host]
        self.any_touched = any(a > 0 for a in alive)
        if self.world > 1 and self.any_touched:
            for k in range(len(self.seg_range)):
                self._launch(k)
            if self._comm_stream is not None:
                torch.cuda.current_stream(self.flat.device).wait_stream(self._comm_stream)
            for work, seg in self._works:
                work.wait()
                seg.mul_(1.0 / self.world)
        for p, a in zip(self.params, alive):
            if a == 0:
                p.grad = None                                # no rank produced it: the optimizer must not see a zero gradient
        return self.any_touched

    # ------------------------------------------------------------------ one-shot form (kept for callers that do not use the hooks)
    def all_reduce_mean(self):
        """Average whatever .grad tensors exist over all ranks (no-op for a single rank)."""
        if self.world == 1:
            return
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
            p.grad = v
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.mul_(1.0 / self.world)


def broadcast_parameters(modules, src=0, group=None):
    """Make every rank start from rank `src`'s weights and buffers (the reference relies on equal seeds and checks with
    check_ddp_consistency, misc.py:184-196).  In-place under no_grad, so every tensor's version counter moves and
    caches keyed on it (the packed-weight caches of training.networks) are invalidated."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for m in modules:
            for t in list(m.parameters()) + list(m.buffers()):
                dist.broadcast(t, src=src, group=group)
