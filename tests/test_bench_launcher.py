"""`python bench.py --gpus N` must start its own N ranks (reference train.py:563-568, 390-410): the launcher
parent (training/launch.py) never imports torch, every rank checks WORLD_SIZE == --gpus, rank 0 prints ONE JSON line with
n_gpus = N.  Runs on CPU: `--mode selftest` is the same protocol over gloo with a stub forward."""

import json
import os
import subprocess

import pytest
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}


def _json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith('{')]


def test_bench_spawns_its_own_ranks():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--mode', 'selftest', '--steps', '3', '--warmup', '1', '--batch', '2'],
                       capture_output=True, text=True, timeout=300, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                      # ONE line, from rank 0
    j = lines[0]
    assert j['n_gpus'] == 2 and j['config']['ranks_seen'] == 2 and j['config']['global_batch'] == 4
    assert j['steps'] == 3 and j['warmup'] == 1 and j['scaling'] == 'weak'


def test_bench_as_a_rank_of_an_external_launcher():
    # what the driver does for N > 1: torch.distributed.run sets RANK / WORLD_SIZE, bench.py must NOT spawn again
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29653', BENCH, '--gpus', '2', '--mode', 'selftest', '--steps', '2', '--warmup', '1'],
                       capture_output=True, text=True, timeout=300, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]['n_gpus'] == 2 and lines[0]['config']['ranks_seen'] == 2


def test_world_size_mismatch_is_refused():
    env = dict(_clean_env(), RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29654')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--mode', 'selftest'], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout)


def test_launcher_parent_does_not_import_torch_and_propagates_failure(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
    probe = tmp_path / 'probe.py'
    probe.write_text('import os, sys\n'
                     'assert os.environ["WORLD_SIZE"] == "3" and os.environ["MASTER_ADDR"] == "127.0.0.1"\n'
                     'open(sys.argv[1] + os.environ["RANK"], "w").write(os.environ["LOCAL_RANK"])\n'
                     'sys.exit(7 if os.environ["RANK"] == "1" and len(sys.argv) > 2 else 0)\n')
    code = ('import sys; sys.path.insert(0, %r); from training import launch; rc = launch.spawn_ranks([%r, %r] + sys.argv[1:], 3); '
            'assert "torch" not in sys.modules; sys.exit(rc)' % (os.path.join(ROOT, 'pasta-gan-plusplus_amd'), str(probe), str(tmp_path / 'r')))
    assert subprocess.run([sys.executable, '-c', code], env=_clean_env(), timeout=60).returncode == 0
    assert sorted(p.name for p in tmp_path.glob('r?')) == ['r0', 'r1', 'r2'] and (tmp_path / 'r2').read_text() == '2'
    assert subprocess.run([sys.executable, '-c', code, 'fail'], env=_clean_env(), timeout=60).returncode == 7


@pytest.mark.parametrize('signals', ['TERM', 'TERM,TERM,INT'], ids=['one', 'three'])
def test_sigterm_to_the_launcher_takes_the_ranks_down(tmp_path, signals):
    # the ranks run in sessions of their own -- a launcher that dies of SIGTERM without cleaning up would leave them holding the GPUs.
    # a SECOND signal arriving while the launcher is already killing / reaping its ranks must not abort that clean-up -- the ranks here ignore
    # SIGTERM for a moment, so the launcher is still inside its clean-up when signals two and three land
    import signal
    import time
    probe = tmp_path / 'sleeper.py'
    probe.write_text('import os, signal, sys, time\nsignal.signal(signal.SIGTERM, lambda *a: (time.sleep(1.0), sys.exit(0)))\n'
                     'open(sys.argv[1] + os.environ["RANK"], "w").write(str(os.getpid()))\ntime.sleep(120)\n')
    code = ('import sys; sys.path.insert(0, %r); from training import launch; sys.exit(launch.spawn_ranks([%r, %r], 2))'
            % (os.path.join(ROOT, 'pasta-gan-plusplus_amd'), str(probe), str(tmp_path / 'pid')))
    parent = subprocess.Popen([sys.executable, '-c', code], env=_clean_env())
    try:
        deadline = time.monotonic() + 30
        while time.monotonic() < deadline and not all((tmp_path / f'pid{r}').exists() and (tmp_path / f'pid{r}').read_text() for r in (0, 1)):
            time.sleep(0.05)
        pids = [int((tmp_path / f'pid{r}').read_text()) for r in (0, 1)]
        for i, name in enumerate(signals.split(',')):
            if i:
                time.sleep(0.2)
            if parent.poll() is None:
                parent.send_signal(getattr(signal, 'SIG' + name))
        assert parent.wait(timeout=30) == 128 + signal.SIGTERM
        deadline = time.monotonic() + 10
        def alive(pid):
            try:
                os.kill(pid, 0)
                with open(f'/proc/{pid}/stat') as f:
                    return f.read().split(')')[-1].split()[0] != 'Z'
            except (ProcessLookupError, FileNotFoundError):
                return False
        while time.monotonic() < deadline and any(alive(p) for p in pids):
            time.sleep(0.05)
        assert not any(alive(p) for p in pids), 'rank processes survived the launcher'
    finally:
        if parent.poll() is None:
            parent.kill()
