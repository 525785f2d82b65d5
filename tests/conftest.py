import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'pasta-gan-plusplus_amd')
for p in (ROOT, PKG, os.path.join(ROOT, 'tests', 'golden'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    class Golden:
        def __init__(self):
            self._cache = {}

        def __call__(self, fname):
            if fname not in self._cache:
                self._cache[fname] = np.load(os.path.join(ROOT, 'tests', 'golden', fname), allow_pickle=False)
            return self._cache[fname]
    return Golden()
