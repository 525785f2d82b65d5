"""Build (or verify) every plugin in a plain, unprofiled process.  The profiling scripts call this first: under rocprofv3 the loader's
hipcc children would inherit the profiler's preload from a process whose GPU the profiler has already initialised."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pasta-gan-plusplus_amd'))
from torch_utils import custom_ops  # noqa: E402

custom_ops.verbosity = 'none'
for path in custom_ops.build_all():
    assert os.path.isfile(path), path
