"""Read network snapshots written by the reference, by parameter name, without executing anything they carry.

Wire format.  The reference's training loop writes
``pickle.dump(dict(G=, D=, G_ema=, D_parsing=..., training_set_kwargs=, augment_pipe=), f)``
(training/training_loop_fullbody.py:723-736).  Every ``@persistence.persistent_class`` module in it reduces to
``torch_utils.persistence._reconstruct_persistent_obj(meta)`` with
``meta = dict(type='class', version, module_src, class_name, state)`` (torch_utils/persistence.py:118-126), where
``module_src`` is the *source text* of the defining module and ``state`` the module's ``__dict__``
(``_parameters`` / ``_buffers`` / ``_modules`` OrderedDicts, ``_init_args`` / ``_init_kwargs`` ...).  The reference's
loader (``legacy.load_network_pkl`` -> ``_reconstruct_persistent_obj``, persistence.py:179-202) ``exec``s that source to
rebuild the classes -- i.e. loading a checkpoint runs whatever code the file contains.

This reader never does.  It is a ``pickle.Unpickler`` with a closed allow-list:

* ``_reconstruct_persistent_obj`` and every ``torch.nn.modules.*`` class become ``ModuleState`` records (class name +
  state dict); ``module_src`` is dropped (only its length and SHA-1 are kept for diagnostics);
* tensors are rebuilt by torch's own ``_rebuild_tensor_v2`` / ``_rebuild_parameter``; storages (``torch.storage.
  _load_from_bytes``) are decoded with ``torch.load(..., weights_only=True)``;
* ``collections.OrderedDict``, ``dnnlib.util.EasyDict`` (as a plain attribute dict), numpy scalars / dtypes / arrays;
* any other global raises ``pickle.UnpicklingError``.

``ModuleState.state_dict()`` reproduces ``torch.nn.Module.state_dict()`` naming (parameters, persistent buffers, then
sub-modules in registration order), so the result loads into this package's networks, whose parameter names are the
reference's (``training/networks.py``): ``load_into(G, 'network-snapshot.pkl', key='G_ema')``.
"""

import collections
import hashlib
import io
import pickle

import numpy as np
import torch


class EasyDict(dict):
    """Attribute-style dict standing in for ``dnnlib.util.EasyDict`` (dnnlib/util.py:40-53)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


class ModuleState:
    """What a snapshot holds for one ``torch.nn.Module``: where the class lived, and its ``__dict__``."""

    class_module = None         # python module path for plain torch classes; None for persistent (source-carrying) classes
    class_name = None

    def __init__(self, class_name=None, state=None, class_module=None, src_info=None):
        if class_name is not None:
            self.class_name = class_name
        if class_module is not None:
            self.class_module = class_module
        self.state = dict(state or {})
        self.src_info = src_info            # (length, sha1) of the embedded module source that was NOT executed

    def __setstate__(self, state):          # BUILD opcode of plain (non-persistent) modules
        self.state = dict(state)
        self.src_info = None

    @property
    def persistent(self):
        return self.class_module is None

    @property
    def init_args(self):
        return tuple(self.state.get('_init_args', ()))

    @property
    def init_kwargs(self):
        return EasyDict(self.state.get('_init_kwargs', {}))

    @property
    def training(self):
        return bool(self.state.get('training', False))

    def children(self):
        return collections.OrderedDict((k, v) for k, v in (self.state.get('_modules') or {}).items() if v is not None)

    def state_dict(self, prefix=''):
        """name -> tensor, in ``torch.nn.Module.state_dict()`` order (module.py: parameters, persistent buffers, children)."""
        out = collections.OrderedDict()
        for name, p in (self.state.get('_parameters') or {}).items():
            if p is not None:
                out[prefix + name] = p.detach()
        skip = self.state.get('_non_persistent_buffers_set') or ()
        for name, b in (self.state.get('_buffers') or {}).items():
            if b is not None and name not in skip:
                out[prefix + name] = b
        for name, child in self.children().items():
            if not isinstance(child, ModuleState):
                raise pickle.UnpicklingError(f'sub-module {prefix + name} is a {type(child).__name__}, not a module record')
            out.update(child.state_dict(prefix + name + '.'))
        return out

    def __repr__(self):
        return f'<ModuleState {self.class_name} ({len(self.state_dict())} tensors)>'


def _reconstruct_persistent_obj(meta):
    """Stand-in for torch_utils/persistence.py:179-202: keep name and state, never touch ``module_src``."""
    if meta.get('type') != 'class':
        raise pickle.UnpicklingError(f"persistent object of type {meta.get('type')!r}")
    src = meta.get('module_src') or ''
    info = (len(src), hashlib.sha1(src.encode('utf-8', 'replace')).hexdigest())
    return ModuleState(class_name=str(meta['class_name']), state=meta.get('state') or {}, src_info=info)


def _load_storage_from_bytes(b):
    """torch.storage._load_from_bytes without the arbitrary-pickle inner load."""
    return torch.load(io.BytesIO(b), weights_only=True)


_stub_cache = {}


def _module_stub(module, name):
    key = (module, name)
    if key not in _stub_cache:
        _stub_cache[key] = type(name, (ModuleState,), dict(class_module=module, class_name=name))
    return _stub_cache[key]


_NUMPY_CORE = ('numpy.core.multiarray', 'numpy._core.multiarray')
_ALLOWED = {
    ('collections', 'OrderedDict'): collections.OrderedDict,
    ('dnnlib.util', 'EasyDict'): EasyDict,
    ('torch_utils.persistence', '_reconstruct_persistent_obj'): _reconstruct_persistent_obj,
    ('torch.storage', '_load_from_bytes'): _load_storage_from_bytes,
    ('torch', 'Size'): torch.Size,
    ('numpy', 'dtype'): np.dtype,
    ('numpy', 'ndarray'): np.ndarray,
}
for _m in _NUMPY_CORE:                               # numpy internals: resolved lazily from whichever numpy is installed
    _ALLOWED[(_m, 'scalar')] = None
    _ALLOWED[(_m, '_reconstruct')] = None
_TORCH_REBUILD = ('_rebuild_tensor_v2', '_rebuild_tensor', '_rebuild_parameter', '_rebuild_parameter_with_state')
_TORCH_STORAGES = ('FloatStorage', 'HalfStorage', 'DoubleStorage', 'BFloat16Storage', 'LongStorage', 'IntStorage', 'ShortStorage',
                   'CharStorage', 'ByteStorage', 'BoolStorage', 'UntypedStorage')
_TORCH_DTYPES = ('float32', 'float16', 'bfloat16', 'float64', 'int64', 'int32', 'int16', 'int8', 'uint8', 'bool')


class SnapshotUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            obj = _ALLOWED[(module, name)]
            if obj is None:
                try:
                    import numpy._core.multiarray as ncm
                except ImportError:                              # numpy < 2
                    import numpy.core.multiarray as ncm
                obj = getattr(ncm, name)
            return obj
        if module == 'torch._utils' and name in _TORCH_REBUILD:
            return getattr(torch._utils, name)
        if module == 'torch' and (name in _TORCH_STORAGES or name in _TORCH_DTYPES):
            return getattr(torch, name)
        if module.startswith('torch.nn.modules.') and name.isidentifier():
            return _module_stub(module, name)                    # plain torch modules: state only, no torch class is instantiated
        raise pickle.UnpicklingError(f'refusing to resolve {module}.{name}: not on the snapshot allow-list')


def read_snapshot(file):
    """Decode a reference snapshot (path or binary file object) into plain data: ``ModuleState`` for every network."""
    if isinstance(file, (str, bytes)) or hasattr(file, '__fspath__'):
        with open(file, 'rb') as f:
            return SnapshotUnpickler(f).load()
    return SnapshotUnpickler(file).load()


def read_state_dicts(file, keys=None):
    """``{key: OrderedDict(name -> tensor)}`` for the networks of a snapshot (default: every network in it)."""
    data = read_snapshot(file)
    if not isinstance(data, dict):
        raise pickle.UnpicklingError(f'snapshot root is a {type(data).__name__}; the TensorFlow-era tuple format is not supported')
    if keys is None:
        keys = [k for k, v in data.items() if isinstance(v, ModuleState)]
    out = collections.OrderedDict()
    for k in keys:
        if not isinstance(data.get(k), ModuleState):
            raise KeyError(f'snapshot has no network {k!r} (has: {[kk for kk, v in data.items() if isinstance(v, ModuleState)]})')
        out[k] = data[k].state_dict()
    return out


def load_into(module, file_or_state, key='G_ema', strict=True):
    """Copy the tensors of network `key` of a reference snapshot into `module` (a network of this package) by name."""
    if isinstance(file_or_state, ModuleState):
        sd = file_or_state.state_dict()
    elif isinstance(file_or_state, collections.abc.Mapping) and all(isinstance(v, torch.Tensor) for v in file_or_state.values()):
        sd = file_or_state
    else:
        sd = read_state_dicts(file_or_state, [key])[key]
    return module.load_state_dict(sd, strict=strict)
