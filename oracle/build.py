"""Build the oracle's C restatement (oracle/c/pg_oracle.c -> oracle/_build/libpg_oracle.so) with gcc.
TEST INFRASTRUCTURE ONLY.  The reference's own native sources are CUDA (.cu) and cannot be compiled in
this image (no nvcc; hipify is out of bounds), so there is no oracle/_ref build: see DESIGN.md."""

import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'c', 'pg_oracle.c')
OUT_DIR = os.path.join(HERE, '_build')
OUT = os.path.join(OUT_DIR, 'libpg_oracle.so')


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.isfile(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        subprocess.run(['gcc', '-O2', '-fPIC', '-shared', '-std=c99', SRC, '-o', OUT, '-lm'], check=True)
    return OUT


def load():
    lib = ctypes.CDLL(build())
    f32p, i, i64, f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    lib.oracle_upfirdn2d.restype = i
    lib.oracle_upfirdn2d.argtypes = [f32p, f32p, f32p] + [i] * 15 + [f]
    lib.oracle_bias_act.restype = i
    lib.oracle_bias_act.argtypes = [f32p, f32p, f32p, i64, i, i64, i, f, f, f]
    lib.oracle_conv2d.restype = i
    lib.oracle_conv2d.argtypes = [f32p, f32p, f32p, f32p] + [i] * 10
    return lib


if __name__ == '__main__':
    print(build(force=True))
