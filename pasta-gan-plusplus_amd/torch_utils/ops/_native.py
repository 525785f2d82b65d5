"""Small shared helpers for the ctypes bindings of the native plugins."""

import ctypes

import torch

PG_DTYPE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.float64: 3}

PG_ERRORS = {-1: 'invalid argument', -2: 'unsupported configuration', -3: 'tensor too large for 32-bit kernel indexing'}


# Every packed-weight cache of this package keys its entries on (parameter address, parameter version, ..., cache_epoch[0]).  A hipGraph
# replay of a training phase updates parameters WITHOUT moving their Python-side version counters and a capture must not rely on packs made
# outside it, so training.training_step bumps the epoch before every capture and after every replay (stale entries then simply miss).
cache_epoch = [0]


def invalidate_packed_weights():
    cache_epoch[0] += 1


class NativeOpError(RuntimeError):
    pass


class NativeNotCovered(NativeOpError):
    """The kernels declined a VALID request (PG_ERR_UNSUPPORTED / PG_ERR_TOO_LARGE): callers holding a composed or aten
    route for the same maths catch this and take it."""


def check(status, what):
    """Turn a C-ABI status into a RuntimeError (TORCH_CHECK analogue of the reference plugins)."""
    if status != 0:
        msg = PG_ERRORS.get(status, f'hipError_t {status}' if status > 0 else f'error {status}')
        raise (NativeNotCovered if status in (-2, -3) else NativeOpError)(f'{what}: {msg}')


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else ctypes.c_void_p(0)


def stream_of(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def i64arr(vals):
    return (ctypes.c_int64 * len(vals))(*[int(v) for v in vals])


def require_gpu(x, opname):
    if x.device.type != 'cuda':
        raise NativeOpError(
            f'{opname}: this entry point is a hand-written HIP kernel and needs a GPU tensor; got {x.device.type}.')


def is_dense(t):
    """True if `t` occupies one gap-free, non-overlapping block of memory in some dimension order
    (what the reference plugin checks with is_non_overlapping_and_dense(), bias_act.cpp:48)."""
    if t.numel() == 0:
        return True
    expected = 1
    for st, sz in sorted((st, sz) for sz, st in zip(t.shape, t.stride()) if sz != 1):
        if st != expected:
            return False
        expected *= sz
    return True
