"""Replay an inference forward as ONE hipGraph launch.

The low-resolution end of the synthesis stack is launch-bound on the host: ~250 small kernels per step, each a few
microseconds of GPU time behind 10-20 us of Python / ctypes dispatch (bf16 1024^2 stack, N=4: 4.3 ms of kernels in a 5.2 ms
step).  Every kernel of this package is enqueued on torch's current stream with plain device pointers and allocates through
torch's caching allocator, so a whole forward can be captured by `torch.cuda.CUDAGraph` (hipGraph on ROCm) and replayed with a
single launch; per-sample weight packing, demodulation coefficients etc. are kernels of the graph like any other, so the replay
does exactly the work of the eager step for whatever the static input buffers hold.

    fwd = GraphedForward(lambda ws: net(ws, noise_mode='const'), [ws])
    img = fwd(ws_new)            # copies ws_new into the static input, replays, returns the static output

Restrictions (those of graph capture): fixed shapes, no host synchronisation or value-dependent Python branching inside the
forward, eval mode / no autograd.  Weight caches keyed on parameter versions are filled by the warm-up passes that run before
the capture; after an in-place weight update build a new GraphedForward.
"""

import torch


class GraphedForward:
    def __init__(self, fn, example_inputs, warmup=3):
        assert all(t.is_cuda for t in example_inputs), 'graph capture needs GPU tensors'
        self.fn = fn
        self.static_inputs = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream(device=self.static_inputs[0].device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):              # first-call work (plugin loading, kernel attributes, weight caches) stays out of the graph
                fn(*self.static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_outputs = fn(*self.static_inputs)

    def __call__(self, *inputs):
        for dst, src in zip(self.static_inputs, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self.graph.replay()
        return self.static_outputs
