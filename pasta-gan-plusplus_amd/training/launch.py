"""Start one process per GPU from a single command -- what the reference's entry points do themselves
(train.py:563-568: ``torch.multiprocessing.spawn(fn=subprocess_fn, nprocs=num_gpus)``, each rank then calls
``init_process_group`` in ``subprocess_fn``, train.py:390-410).

Differences, on purpose:
* the ranks are fresh ``python`` child processes of a parent that has NOT touched the GPU (on this pool a process that
  has initialised HIP must never exec / fork into another GPU user); the parent imports neither torch nor this
  package's plugins, it only waits;
* rendezvous is the ``env://`` form over 127.0.0.1 with a port the parent found free (the reference uses a file in a
  temp dir; the driver's own launcher -- torch.distributed.run -- uses the same variables, so a rank cannot tell who
  started it): RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT;
* a rank that dies takes the others down (the parent kills the process group it started) and its exit code is returned;
* a parent that is told to stop (SIGTERM / SIGHUP / SIGINT: a driver's timeout, a cancelled job) takes its ranks with it: the
  ranks run in sessions of their own, so the signal is turned into an exception that runs the same clean-up (the reference's
  ``torch.multiprocessing.spawn`` children are daemonic and die with their parent).
"""

import os
import signal
import socket
import subprocess
import sys
import time

RANK_VARS = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')


def launched_as_rank(env=None):
    """True inside a process that some launcher (this one or torch.distributed.run) started as one rank of a job."""
    env = os.environ if env is None else env
    return 'RANK' in env and 'WORLD_SIZE' in env


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: what RCCL needs on this host driver
    return env


def spawn_ranks(argv, nprocs, timeout=None, poll=0.05):
    """Run ``python *argv`` as `nprocs` ranks (children inherit stdout / stderr: rank 0 prints the result line).
    Returns the first non-zero exit code, or 0.  `timeout` (seconds) bounds the whole job."""
    assert nprocs >= 1
    port = free_port()

    def _stop(signum, _frame):
        raise SystemExit(128 + signum)                        # unwinds through the `finally` below: the ranks are killed first

    handled = (signal.SIGTERM, signal.SIGHUP, signal.SIGINT)
    previous = {}
    try:
        for sig in handled:
            previous[sig] = signal.signal(sig, _stop)
    except ValueError:                                        # not the main thread: the caller owns signal handling
        pass
    procs = []
    deadline = None if timeout is None else time.monotonic() + timeout
    code = 0
    try:
        for r in range(nprocs):
            procs.append(subprocess.Popen([sys.executable, *argv], env=rank_env(r, nprocs, port), start_new_session=True))
        alive = list(procs)
        while alive:
            for p in list(alive):
                rc = p.poll()
                if rc is None:
                    continue
                alive.remove(p)
                if rc != 0 and code == 0:
                    code = rc
            if code != 0 or (deadline is not None and time.monotonic() > deadline):
                if code == 0:
                    code = 124
                break
            time.sleep(poll)
    finally:
        # a second signal (drivers send TERM twice, or TERM then INT) must not abort the clean-up below and leave ranks alive in their own sessions,
        # holding GPUs: the handled signals are ignored until every rank has been killed and reaped
        for sig in previous:
            signal.signal(sig, signal.SIG_IGN)
        for p in procs:                                       # only the exact process groups started above
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except ProcessLookupError:
                    pass
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                p.wait()
        for sig, old in previous.items():
            signal.signal(sig, old)
    return code
