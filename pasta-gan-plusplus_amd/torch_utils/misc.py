"""Helpers the operator modules import (reference torch_utils/misc.py:86-109, 127-146)."""

import contextlib
import warnings

import torch


def assert_shape(tensor, ref_shape):
    """Check `tensor.shape` against `ref_shape`; None entries are wildcards (misc.py:86-99)."""
    if tensor.ndim != len(ref_shape):
        raise AssertionError(f'Wrong number of dimensions: got {tensor.ndim}, expected {len(ref_shape)}')
    for idx, (size, ref) in enumerate(zip(tensor.shape, ref_shape)):
        if ref is None:
            continue
        if isinstance(ref, torch.Tensor):
            ref = int(ref)
        if int(size) != int(ref):
            raise AssertionError(f'Wrong size for dimension {idx}: got {int(size)}, expected {int(ref)}')


@contextlib.contextmanager
def suppress_tracer_warnings():
    """Silence TracerWarnings inside the block (misc.py:75-80)."""
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=torch.jit.TracerWarning)
        yield


def profiled_function(fn):
    """Wrap `fn` in a profiler range named after it (misc.py:104-109)."""
    def decorator(*args, **kwargs):
        with torch.autograd.profiler.record_function(fn.__name__):
            return fn(*args, **kwargs)
    decorator.__name__ = fn.__name__
    return decorator


class InfiniteSampler(torch.utils.data.Sampler):
    """Rank-strided endless index stream (misc.py:114-146): every rank draws the same
    permutation stream and keeps indices with position % num_replicas == rank."""

    def __init__(self, dataset, rank=0, num_replicas=1, shuffle=True, seed=0, window_size=0.5):
        assert len(dataset) > 0 and num_replicas > 0 and 0 <= rank < num_replicas and 0 <= window_size <= 1
        super().__init__()
        self.dataset, self.rank, self.num_replicas = dataset, rank, num_replicas
        self.shuffle, self.seed, self.window_size = shuffle, seed, window_size

    def __iter__(self):
        import numpy as np
        order = np.arange(len(self.dataset))
        rnd, window = None, 0
        if self.shuffle:
            rnd = np.random.RandomState(self.seed)
            rnd.shuffle(order)
            window = int(np.rint(order.size * self.window_size))
        idx = 0
        while True:
            i = idx % order.size
            if idx % self.num_replicas == self.rank:
                yield order[i]
            if window >= 2:
                j = (i - rnd.randint(window)) % order.size
                order[i], order[j] = order[j], order[i]
            idx += 1
