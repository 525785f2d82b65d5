"""The data-parallel exchange of the training step ON THE GPU.  The reference's step is DDP over 8 ranks
(training/training_loop_fullbody.py:451-460, 604-639); there is one GPU here, so the branch an 8-GPU run takes -- `GradBucket._launch` on the side stream from
the autograd hooks, flags travelling inside the segments, `finish()`'s device-side flags, `FlatAdam` reading `alive` after the all-reduce, grids sized for
CUs - PG_COMM_CUS -- is executed (a) on a one-rank RCCL group with PG_FORCE_EXCHANGE=1 and (b) by two gloo ranks that share GPU 0.  Every run is a fresh child
process (tests/exchange_worker.py): a process group inside the pytest process would change what every later test sees."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, 'exchange_worker.py')
pytestmark = pytest.mark.gpu


def _port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run(out, env=None, **kw):
    args = [sys.executable, WORKER, out]
    for k, v in kw.items():
        if v is True:
            args.append('--' + k.replace('_', '-'))
        elif v is not False:
            args += ['--' + k.replace('_', '-'), str(v)]
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    e.update(env or {})
    return subprocess.Popen(args, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def _wait(procs, timeout=900):
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-4000:]
    return outs


def test_forced_exchange_on_one_rank_equals_the_plain_step():
    """PG_FORCE_EXCHANGE=1 on a one-rank `nccl` (RCCL) group: every phase zeroes its bucket, exchanges its segments from the hooks / from finish() on the side stream
    (identity all-reduces), gathers the summed flags on the device, and FlatAdam skips by them; with the same CU reservation (PG_COMM_CUS=8 on both sides: it sizes
    every persistent grid and split-K plan, i.e. the summation order) the weights and second moments after 5 iterations must equal the plain single-process step.
    "Equal" = within the run-to-run noise of the plain step itself, measured here by a second plain run (the stub generator's aten convolution gradients use atomics:
    two plain runs differ in the last bits), with the same Adam step counts exactly.  The forced run executes under torch.cuda.set_sync_debug_mode('error') from the
    second iteration on: no host synchronisation inside a phase."""
    with tempfile.TemporaryDirectory() as tmp:
        plain, again, forced = (os.path.join(tmp, n) for n in ('plain.npz', 'again.npz', 'forced.npz'))
        _wait([_run(plain, env=dict(PG_COMM_CUS='8'), backend='none', iters=5)])
        _wait([_run(again, env=dict(PG_COMM_CUS='8'), backend='none', iters=5)])
        _wait([_run(forced, env=dict(PG_COMM_CUS='8', PG_FORCE_EXCHANGE='1'), backend='nccl', world=1, port=_port(), iters=5, sync_debug=True)])
        a, a2, b = dict(np.load(plain)), dict(np.load(again)), dict(np.load(forced))
    assert int(b['__exchange']) == 1 and int(a['__exchange']) == 0 and int(b['__device_flags']) == 1 and int(b['__comm_cus']) == 8
    assert int(b['__hooks']) >= 5 and int(b['__hooks']) + int(b['__finishes']) >= 20, (b['__hooks'], b['__finishes'])      # segments went out from the autograd hooks (and the rest from finish())
    assert int(a['__hooks']) == 0
    keys = [k for k in a if not k.startswith('__')]
    assert len(keys) > 40 and set(keys) == {k for k in b if not k.startswith('__')}
    for k in keys:
        if k.endswith('.steps'):
            assert np.array_equal(a[k], b[k]), k
            continue
        noise = float(np.abs(a[k] - a2[k]).max())
        sc = max(1e-6, float(np.abs(a[k]).max()))
        assert float(np.abs(a[k] - b[k]).max()) <= max(4 * noise, 2e-6 * sc), (k, float(np.abs(a[k] - b[k]).max()), noise, sc)


def test_two_gloo_ranks_sharing_the_gpu_match_the_single_process_step():
    """Two fresh ranks on GPU 0 (backend gloo, CUDA tensors), each one TrainingStep.run per iteration on its half of the batch: gradients are bucket views on the device,
    segments leave from the hooks of the last backward on the side stream, the flags travel in the segments, FlatAdam steps from the device flags.  Rank 0's weights
    after 3 iterations against the single-process step on the whole batch: two half-batch means vs one full-batch mean -- reduction order only (the GPU twin of
    tests/test_training_host.py::test_flat_bucket_overlapped_exchange_two_ranks / test_training_step_is_world_size_invariant_through_greg).  The minibatch-stddev
    group is 2 and rank r holds samples r::2, so the groups differ between the two runs; the epilogue's statistics term is why the tolerance is not tighter."""
    with tempfile.TemporaryDirectory() as tmp:
        one, two = os.path.join(tmp, 'one.npz'), os.path.join(tmp, 'two.npz')
        _wait([_run(one, env=dict(PG_COMM_CUS='8'), backend='none', iters=3)])
        port = _port()
        _wait([_run(two if r == 0 else os.path.join(tmp, 'r1.npz'), env=dict(PG_COMM_CUS='8'), backend='gloo', rank=r, world=2, port=port, iters=3) for r in range(2)])
        a, b = dict(np.load(one)), dict(np.load(two))
    assert int(b['__exchange']) == 1 and int(b['__device_flags']) == 1 and int(b['__hooks']) >= 3
    for k in a:
        if k.startswith('__') or k.startswith('adam'):
            continue
        sc = max(1e-3, float(np.abs(a[k]).max()))
        assert float(np.abs(a[k] - b[k]).max()) <= 2e-2 * sc, (k, float(np.abs(a[k] - b[k]).max()), sc)
    for k in a:
        if k.endswith('.steps'):
            assert np.array_equal(a[k], b[k]), k                    # the same parameters were stepped the same number of times
