"""Loader for the gfx950 native plugins (C-ABI shared libraries built from ``csrc/*.hip``).

Same role and entry point as the reference's ``torch_utils/custom_ops.py`` --
``get_plugin(module_name, sources, **build_kwargs)`` (reference custom_ops.py:46-124),
module global ``verbosity`` (:23), one cached instance per process (:44-51), a
file lock so concurrent ranks build once (:95-105) -- but MI355X-native: the sources
are hand-written HIP compiled by ``hipcc --offload-arch=gfx950`` into an in-tree
``csrc/<module_name>.so`` that is bound with ``ctypes`` (no pybind, no torch
extension, no hipify).  A content digest next to the .so makes the build
incremental.  The .so files are build products: git-ignored (a fresh checkout builds them on
first use, or ahead of time with ``build_all()`` / ``tools/ensure_built.py`` -- do that in a plain process, never under
a profiler) but not gpurun-ignored, so a tree built here carries them to the GPU box, where an up-to-date .so is just loaded.

There is no fallback: if the library can neither be found nor built this raises,
and the ops built on it raise too.  A library older than its sources is only loaded when
``PG_ALLOW_STALE_PLUGIN=1`` says so, and every loaded library must report the ABI version the
ctypes bindings were written against (``pg_<plugin>_abi_version()`` == ``ABI_VERSION``).
"""

import concurrent.futures
import ctypes
import fcntl
import glob
import hashlib
import os
import re
import shutil
import subprocess
import sys

verbosity = 'brief'  # Verbosity level: 'none', 'brief', 'full' (reference custom_ops.py:23)

CSRC_DIR = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'csrc'))
INCLUDE_DIR = os.path.normpath(os.path.join(CSRC_DIR, '..', '..', 'include'))
DEV_DIR = os.path.join(CSRC_DIR, 'dev')
OFFLOAD_ARCH = 'gfx950'
HIPCC_FLAGS = ['-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', f'--offload-arch={OFFLOAD_ARCH}']

# Default source lists of the three plugins (callers may pass their own `sources`).
PLUGIN_SOURCES = {
    'bias_act_plugin': ['bias_act.hip'],
    'upfirdn2d_plugin': ['upfirdn2d.hip'],
    'patch_routing_plugin': ['patch_routing.hip'],
    'conv2d_plugin': ['conv2d.hip', 'conv2d_inst_k3s1.hip', 'conv2d_inst_k1s1.hip', 'conv2d_inst_k2x2.hip', 'conv2d_inst_k2x1.hip',
                      'conv2d_inst_k1x2.hip', 'conv2d_inst_k7s1.hip', 'conv2d_inst_k3s2.hip', 'conv2d_inst_k1s2.hip', 'conv2d_inst_wino.hip', 'conv2d_inst_wino4.hip', 'conv2d_inst_wino4s.hip', 'conv2d_inst_wino4t.hip', 'conv2d_inst_wino4x.hip', 'conv2d_inst_wino4xs.hip', 'conv2d_inst_wino4xt.hip', 'conv2d_inst_wino4b.hip', 'conv2d_inst_wino4bs.hip', 'conv2d_inst_up2.hip', 'conv2d_inst_stem7.hip', 'conv2d_inst_s1x1.hip',
                      'conv2d_wgrad.hip', 'conv2d16.hip', 'conv2d16_inst_k3s1.hip', 'conv2d16_inst_k3s1b.hip', 'conv2d16_inst_k1s1.hip', 'conv2d16_inst_k2x2.hip', 'conv2d16_inst_k2x1.hip',
                      'conv2d16_inst_k1x2.hip', 'conv2d16_inst_k3s2.hip', 'conv2d16_inst_up2f.hip', 'conv1x1_head16.hip', 'optim.hip'],
}

ABI_VERSION = 13     # == PG_ABI_VERSION of include/pasta_gan_ops.h; bumped with every struct / signature change

_cached_plugins = dict()


class NativePlugin:
    """A loaded C-ABI library: ``plugin.lib`` is the ``ctypes.CDLL``; ``plugin.path`` its file."""

    def __init__(self, name, path, abi_name=None):
        self.name = name
        self.path = path
        self.lib = ctypes.CDLL(path)
        probe = getattr(self.lib, 'pg_' + (abi_name or name).replace('_plugin', '') + '_abi_version', None)
        if probe is None:
            raise RuntimeError(f'{path} does not export its ABI version; rebuild it from csrc/')
        probe.restype = ctypes.c_int
        if probe() != ABI_VERSION:
            raise RuntimeError(f'{path} implements ABI version {probe()}, the bindings expect {ABI_VERSION}; rebuild it from csrc/')

    def __repr__(self):
        return f'<NativePlugin {self.name} at {self.path}>'


def _find_hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.isfile(cand):
            return cand
    return None


def _digest(sources, extra_flags):
    h = hashlib.md5()
    deps = sorted(set(list(sources) + glob.glob(os.path.join(CSRC_DIR, '*.h')) + glob.glob(os.path.join(INCLUDE_DIR, '*.h'))))
    for path in deps:
        h.update(os.path.basename(path).encode())
        with open(path, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(HIPCC_FLAGS + list(extra_flags)).encode())
    return h.hexdigest()


def _log(msg, end='\n'):
    if verbosity != 'none':
        print(msg, end=end, flush=True)


def source_flags(src):
    """Per-source compiler flags: a `// hipcc-flags: ...` comment in the first lines of a translation unit (part of the
    source text, hence of the build digest)."""
    flags = []
    with open(src) as f:
        for _ in range(8):
            m = re.match(r'\s*//\s*hipcc-flags:\s*(.*)', f.readline())
            if m:
                flags += m.group(1).split()
    return flags


def _compile_one(hipcc, src, obj, extra_flags):
    cmd = [hipcc] + HIPCC_FLAGS + list(extra_flags) + source_flags(src) + ['-I', INCLUDE_DIR, '-I', CSRC_DIR, '-c', src, '-o', obj]
    if verbosity == 'full':
        print(' '.join(cmd), flush=True)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f'hipcc failed on {src}:\n{proc.stdout}')
    return obj


def _build(module_name, sources, so_path, digest, extra_flags):
    hipcc = _find_hipcc()
    if hipcc is None:
        raise RuntimeError(f'cannot build "{module_name}": hipcc not found and no up-to-date {so_path}')
    obj_dir = os.path.join(CSRC_DIR, 'build', module_name)
    os.makedirs(obj_dir, exist_ok=True)
    jobs = max(1, min(len(sources), int(os.environ.get('PG_BUILD_JOBS', os.cpu_count() or 1))))
    objs = [os.path.join(obj_dir, os.path.splitext(os.path.basename(s))[0] + '.o') for s in sources]
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
        list(pool.map(lambda so: _compile_one(hipcc, so[0], so[1], extra_flags), zip(sources, objs)))
    tmp = so_path + f'.tmp{os.getpid()}'
    cmd = [hipcc, '-shared', '-fPIC', f'--offload-arch={OFFLOAD_ARCH}', '-o', tmp] + objs
    try:
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f'link of {module_name} failed:\n{proc.stdout}')
        os.replace(tmp, so_path)
    finally:
        clean(only=module_name)                   # hipcc's `<out>.N.host-*` temporaries and a half-written tmp of a failed / interrupted link
    with open(so_path + '.digest', 'w') as f:
        f.write(digest)


def plugin_path(module_name):
    """Product plugins live in csrc/; every other name (development variants built by tools/ with extra -D flags) in csrc/dev/,
    which `clean(dev=True)` empties -- dev builds never sit beside the product libraries."""
    if module_name in PLUGIN_SOURCES and module_name.endswith('_plugin'):
        return os.path.join(CSRC_DIR, module_name + '.so')
    os.makedirs(DEV_DIR, exist_ok=True)
    return os.path.join(DEV_DIR, module_name + '.so')


def _pid_alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except OSError:
        return True             # exists, not ours
    return True


def _temporaries_of(name):
    """Compiler / link temporaries of exactly the plugin `name` in csrc/ and csrc/dev/: hipcc's `<name>.so.tmp<pid>.N.host-*` and `<name>.so.N*` files
    and the `<name>.so.tmp<pid>` a link writes before its os.replace.  Exact names, no prefix glob (`wino4b_exp1` must not match `wino4b_exp10`)."""
    out = []
    for d in (CSRC_DIR, DEV_DIR):
        for pat in (name + '.so.tmp*', name + '.so.[0-9]*', name + '.so.*.host-*', name + '.so-*host-*'):
            out += glob.glob(os.path.join(d, pat))
    return sorted(set(out))


def clean(dev=False, only=''):
    """Delete compiler temporaries an interrupted hipcc left behind and, with dev=True, the development builds of csrc/dev/.  `only` = one plugin's files
    (a build cleans up after itself, inside that plugin's build lock).  Without `only`, every known plugin -- and every library found in csrc/ or
    csrc/dev/ -- is swept one by one while holding ITS build lock (a plugin being linked by another process right now is skipped after its lock wait), and a
    `<so>.tmp<pid>` whose pid is alive is never touched (a global sweep used to delete the half-written link output of a concurrent build).
    Returns the removed paths."""
    gone = []

    def sweep(name):
        for path in _temporaries_of(name):
            m = re.search(r'\.so\.tmp(\d+)', os.path.basename(path))
            if m and int(m.group(1)) != os.getpid() and _pid_alive(int(m.group(1))):
                continue
            try:
                os.remove(path)
                gone.append(path)
            except OSError:
                pass
    if only:
        sweep(only)             # (called from _build, under get_plugin's lock on this plugin)
    else:
        names = set(PLUGIN_SOURCES)
        for d in (CSRC_DIR, DEV_DIR):
            names |= {os.path.basename(f).split('.so')[0] for f in glob.glob(os.path.join(d, '*.so*'))}
        os.makedirs(os.path.join(CSRC_DIR, 'build'), exist_ok=True)
        for name in sorted(names):
            with open(os.path.join(CSRC_DIR, 'build', name + '.lock'), 'w') as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                try:
                    sweep(name)
                finally:
                    fcntl.flock(lock, fcntl.LOCK_UN)
    if dev and os.path.isdir(DEV_DIR):
        gone += glob.glob(os.path.join(DEV_DIR, '*'))
        shutil.rmtree(DEV_DIR, ignore_errors=True)
    return gone


def is_up_to_date(module_name, sources=None, extra_flags=()):
    sources = _resolve_sources(module_name, sources)
    so_path = plugin_path(module_name)
    try:
        with open(so_path + '.digest') as f:
            return os.path.isfile(so_path) and f.read().strip() == _digest(sources, extra_flags)
    except OSError:
        return False


def _resolve_sources(module_name, sources):
    if sources is None:
        sources = PLUGIN_SOURCES[module_name]
    return [s if os.path.isabs(s) else os.path.join(CSRC_DIR, s) for s in sources]


def get_plugin(module_name, sources=None, extra_hipcc_flags=(), build_only=False, abi_name=None, **_ignored_build_kwargs):
    """Build (if stale) and load the native plugin `module_name`; returns a `NativePlugin`.

    Extra keyword arguments of the reference signature (``extra_cuda_cflags=...``) are accepted
    and ignored -- there is no nvcc here.
    """
    assert verbosity in ['none', 'brief', 'full']
    if module_name in _cached_plugins and not build_only:
        return _cached_plugins[module_name]

    sources = _resolve_sources(module_name, sources)
    so_path = plugin_path(module_name)
    digest = _digest(sources, extra_hipcc_flags)
    os.makedirs(os.path.join(CSRC_DIR, 'build'), exist_ok=True)
    lock_path = os.path.join(CSRC_DIR, 'build', module_name + '.lock')
    with open(lock_path, 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)          # concurrent ranks: first one builds, the rest wait
        try:
            if not is_up_to_date(module_name, sources, extra_hipcc_flags):
                if os.path.isfile(so_path) and _find_hipcc() is None:
                    if os.environ.get('PG_ALLOW_STALE_PLUGIN', '0') != '1':
                        raise RuntimeError(f'"{module_name}" is older than its sources and hipcc is missing; '
                                           'set PG_ALLOW_STALE_PLUGIN=1 to load the stale library anyway')
                    _log(f'Warning: "{module_name}" is older than its sources and hipcc is missing; loading the stale library (PG_ALLOW_STALE_PLUGIN=1).')
                else:
                    _log(f'Building native plugin "{module_name}" for {OFFLOAD_ARCH}... ', end='' if verbosity == 'brief' else '\n')
                    try:
                        _build(module_name, sources, so_path, digest, extra_hipcc_flags)
                    except Exception:
                        _log('Failed!')
                        raise
                    _log('Done.')
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    if build_only:
        return so_path
    plugin = NativePlugin(module_name, so_path, abi_name)      # abi_name: dev variants of a plugin built under another name
    _cached_plugins[module_name] = plugin
    return plugin


def build_all(names=None):
    """Compile every plugin (used by __graft_entry__.build()); leaves no compiler temporaries behind."""
    clean()
    return [get_plugin(n, build_only=True) for n in (names or PLUGIN_SOURCES)]


if __name__ == '__main__':
    verbosity = 'full' if '-v' in sys.argv else 'brief'
    for p in build_all([a for a in sys.argv[1:] if not a.startswith('-')] or None):
        print(p)
