"""Image-sharded ("replica") execution of the synthesis forward across the GPUs of one node.

The forward has no cross-image coupling (InstanceNorm, demodulation and the SPADE statistics are per
sample; there is no BatchNorm on the path -- SURVEY.md section 8e), so data parallelism needs no
collective on the data path: rank r owns images r, r + world, r + 2*world, ... exactly as the
reference's ``InfiniteSampler(rank, num_replicas)`` deals indices to ranks (torch_utils/misc.py:138-146),
every rank holds a full copy of the weights, and the only communication is optional result gathering
and the timing reduction of ``bench.py``.  One process per GPU; ``torch.distributed`` backend "nccl"
(= RCCL over xGMI) on the GPUs, "gloo" in the CPU tests.
"""

import torch
import torch.distributed as dist


def shard_indices(num_items, rank, world):
    """Indices of the items rank `rank` owns out of `num_items` (rank-strided, like InfiniteSampler)."""
    assert 0 <= rank < world
    return list(range(rank, num_items, world))


def take(batch, idx):
    """Select rows `idx` of every tensor in a (nested) dict/list/tuple batch; non-tensors pass through."""
    if isinstance(batch, torch.Tensor):
        return batch[idx]
    if isinstance(batch, dict):
        return {k: take(v, idx) for k, v in batch.items()}
    if isinstance(batch, (list, tuple)):
        return type(batch)(take(v, idx) for v in batch)
    return batch


def run_sharded(forward, batch, num_items, rank=None, world=None):
    """Run `forward` on this rank's share of `batch`; returns (outputs, owned indices)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    idx = shard_indices(num_items, rank, world)
    if not idx:
        return None, idx
    return forward(take(batch, torch.as_tensor(idx))), idx


def gather_outputs(outputs, idx, num_items, dst=0):
    """Reassemble per-rank outputs (tuple of tensors with a leading image dimension) on rank `dst`
    in original image order.  Uses all_gather_object-free tensor collectives so it runs on RCCL and gloo."""
    world, rank = dist.get_world_size(), dist.get_rank()
    outputs = tuple(outputs) if outputs is not None else ()
    counts = [len(shard_indices(num_items, r, world)) for r in range(world)]
    full = []
    for k in range(max(len(outputs), int(_bcast_int(len(outputs), src=dst if counts[dst] else _first_nonempty(counts))))):
        mine = outputs[k] if outputs else None
        shape_tail, dtype, device = _describe(mine, counts, k)
        chunks = [torch.empty([c, *shape_tail], dtype=dtype, device=device) for c in counts]
        mine = mine.contiguous() if mine is not None else torch.empty([0, *shape_tail], dtype=dtype, device=device)
        dist.all_gather(chunks, mine) if len(set(counts)) == 1 else _all_gather_ragged(chunks, mine, counts)
        if rank == dst:
            out = torch.empty([num_items, *shape_tail], dtype=dtype, device=device)
            for r, chunk in enumerate(chunks):
                out[shard_indices(num_items, r, world)] = chunk
            full.append(out)
    return tuple(full) if rank == dst else None


def _first_nonempty(counts):
    return next(r for r, c in enumerate(counts) if c)


def _bcast_int(v, src):
    t = torch.tensor([int(v)], dtype=torch.int64, device=_coll_device())
    dist.broadcast(t, src=src)
    return int(t.item())


def _coll_device():
    return torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')


def _describe(t, counts, k):
    """Agree on (trailing shape, dtype, device) of output k across ranks (a rank may own nothing)."""
    dev = _coll_device()
    info = torch.zeros([8], dtype=torch.int64, device=dev)
    if t is not None:
        info[0] = t.ndim - 1
        info[1:t.ndim] = torch.tensor(t.shape[1:], dtype=torch.int64)
        info[7] = {torch.float32: 0, torch.float64: 1, torch.float16: 2, torch.bfloat16: 3, torch.int64: 4}[t.dtype]
    dist.all_reduce(info, op=dist.ReduceOp.MAX)
    nd = int(info[0])
    dtype = [torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.int64][int(info[7])]
    return [int(v) for v in info[1:1 + nd]], dtype, dev


def _all_gather_ragged(chunks, mine, counts):
    """all_gather with unequal leading sizes: pad to the maximum, gather, trim."""
    m = max(counts)
    pad = torch.zeros([m, *mine.shape[1:]], dtype=mine.dtype, device=mine.device)
    pad[:mine.shape[0]] = mine
    bufs = [torch.empty_like(pad) for _ in counts]
    dist.all_gather(bufs, pad)
    for c, buf, out in zip(counts, bufs, chunks):
        out.copy_(buf[:c])


def max_over_ranks(seconds, device=None):
    """MAX-reduce a scalar (the timed region) over all ranks; identity when not distributed."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else _coll_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
