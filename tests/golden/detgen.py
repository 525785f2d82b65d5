"""The deterministic, name-keyed tensor generator now lives in the product tree
(``pasta-gan-plusplus_amd/training/synthetic.py``: ``bench.py`` builds its synthetic weights with it and must not depend on
``tests/``).  This shim loads that file BY PATH -- not as ``training.synthetic`` -- because ``make_golden.py`` runs with the
reference's own ``training`` package on ``sys.path``."""

import importlib.util
import os

_path = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'pasta-gan-plusplus_amd', 'training', 'synthetic.py'))
_spec = importlib.util.spec_from_file_location('_pg_synthetic', _path)
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)

det_array = _mod.det_array
det_tensor = _mod.det_tensor
fill_module_ = _mod.fill_module_
synthesis_inputs = _mod.synthesis_inputs
