"""Dev tool (build container): verify on the generated gfx950 assembly that the compiler put no `s_waitcnt vmcnt(..)` of its own
into (or just in front of) the K loop of the MFMA convolution kernels.  Their global traffic inside that loop -- halo DMA into
LDS, the Winograd U ring -- is inline asm with hand-counted waits; a compiler-made `vmcnt(0)` there also waits for the DMA
requested a moment earlier, i.e. it serialises the double buffering (this happened in the round-1 build: the once-per-tile
scale / bias loads left a pending-VMEM mark on registers the loop reuses).  Usage:
    hipcc -O3 -std=c++17 -I include --offload-arch=gfx950 -c pasta-gan-plusplus_amd/csrc/conv2d_inst_wino.hip -o /tmp/w.o -save-temps=obj
    python tools/check_kloop_waits.py /tmp/conv2d_inst_wino-hip-amdgcn-amd-amdhsa-gfx950.s
"""
import re
import sys


def check(path, name_filter='', verbose=True):
    txt = open(path).read()
    ok, seen = True, 0
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if name_filter and name_filter not in name:
            continue
        lines = body.split('\n')
        mf = [i for i, l in enumerate(lines) if 'v_mfma' in l]
        if not mf:
            continue
        seen += 1
        bad, inasm = [], False
        for i, l in enumerate(lines):
            if 'ASMSTART' in l:
                inasm = True
            elif 'ASMEND' in l:
                inasm = False
            elif re.search(r's_waitcnt.*vmcnt\(\d+\)', l) and not inasm and mf[0] - 60 <= i <= mf[-1]:
                bad.append((i, l.strip(), lines[i + 1].strip()))
        if verbose:
            print(f'{name[:90]}: {len(mf)} MFMAs, compiler-made vmcnt waits in / just before the K loop: {len(bad)}')
            for b in bad[:6]:
                print('    line', b[0], b[1], '|', b[2])
        ok &= not bad
    return ok and seen > 0


if __name__ == '__main__':
    sys.exit(0 if check(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '') else 1)
