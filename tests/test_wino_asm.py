"""The Winograd kernel requests its U-operand ring with inline-asm loads the compiler cannot see and waits for them with
hand-counted s_waitcnt (csrc/conv2d_wino.h).  That is only sound if the compiler never touches a ring register while its word
is in flight -- inside the K loop the dataflow guarantees it, across the loop back edge and through the tile tail it is a
property of the generated code.  This test compiles the translation unit for gfx950 and checks that property on the assembly
(tools/check_wino_asm.py): loop-carried ring slots keep their registers at the back edge, and nothing reads or writes them
between the last refill and the drain."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.isfile('/opt/rocm/bin/hipcc'), reason='needs hipcc')
def test_ring_registers_untouched_while_in_flight(tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    src = os.path.join(ROOT, 'pasta-gan-plusplus_amd', 'csrc', 'conv2d_inst_wino.hip')
    sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
    from torch_utils import custom_ops
    cmd = [hipcc] + custom_ops.HIPCC_FLAGS + custom_ops.source_flags(src) + ['-I', os.path.join(ROOT, 'include'),
           '-I', os.path.dirname(src), '-c', src, '-o', str(tmp_path / 'w.o'), '-save-temps=obj']
    subprocess.run(cmd, check=True, cwd=tmp_path, capture_output=True, timeout=600)
    asm = glob.glob(str(tmp_path / '*gfx950.s'))
    assert len(asm) == 1
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import check_wino_asm
    import check_kloop_waits
    assert check_wino_asm.check(asm[0])
    assert check_kloop_waits.check(asm[0], verbose=False)


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.isfile('/opt/rocm/bin/hipcc'), reason='needs hipcc')
@pytest.mark.parametrize('unit', ['conv2d_inst_k3s1', 'conv2d16_inst_k3s1'])
def test_no_compiler_vmcnt_wait_in_k_loop(tmp_path, unit):
    """The chunk loops double-buffer their halo DMA behind hand-counted waits; a compiler-made vmcnt wait in front of a chunk's
    first LDS read would wait for the DMA just requested (tools/check_kloop_waits.py)."""
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    src = os.path.join(ROOT, 'pasta-gan-plusplus_amd', 'csrc', unit + '.hip')
    sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
    from torch_utils import custom_ops
    cmd = [hipcc] + custom_ops.HIPCC_FLAGS + custom_ops.source_flags(src) + ['-I', os.path.join(ROOT, 'include'),
           '-I', os.path.dirname(src), '-c', src, '-o', str(tmp_path / 'k.o'), '-save-temps=obj']
    subprocess.run(cmd, check=True, cwd=tmp_path, capture_output=True, timeout=900)
    asm = glob.glob(str(tmp_path / '*gfx950.s'))
    assert len(asm) == 1
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import check_kloop_waits
    assert check_kloop_waits.check(asm[0], verbose=False)
