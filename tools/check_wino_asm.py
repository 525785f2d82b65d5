"""Dev tool (build container): verify on the generated gfx950 assembly that the Winograd kernel's in-flight U-ring registers are
never touched outside the K loop.  The ring words are requested by inline-asm loads the compiler cannot see; if it ever copied
or spilled such a register between the request and the hand-written s_waitcnt it would move stale data.  Usage:
    hipcc -O3 -std=c++17 -I include --offload-arch=gfx950 -c pasta-gan-plusplus_amd/csrc/conv2d_inst_wino.hip -o /tmp/w.o -save-temps=obj
    python tools/check_wino_asm.py /tmp/conv2d_inst_wino-hip-amdgcn-amd-amdhsa-gfx950.s
"""
import re
import sys


def regs(line):
    out = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]', line):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', line):
        out.add(int(m.group(1)))
    return out


def check(path):
    txt = open(path).read()
    ok = True
    for name, body in re.findall(r'^(_ZN6pgconv11conv2d_wino\w+):[^\n]*\n(.*?)s_endpgm', txt, re.S | re.M):
        lines = body.split('\n')
        loads = [i for i, l in enumerate(lines) if re.match(r'\s*global_load_dwordx4 v\[', l)]
        dst = lambda i: frozenset(regs(lines[i].split(',')[0]))
        mf = [i for i, l in enumerate(lines) if 'v_mfma' in l]
        hi = mf[-1]
        pro = [i for i in loads if i < mf[0]][:4]                         # the kernel prologue's four ring words
        last = [i for i in loads if i <= hi + 16][-4:]                    # the last four refills of the K loop body
        carried = set().union(*[dst(i) for i in pro])
        same = {dst(i) for i in pro} == {dst(i) for i in last}            # loop-carried slots live in the same registers
        # window 1: prologue loads .. first hand-written vmcnt(0); window 2: last refill .. the drain in the tail
        def window(start):
            for j in range(start, len(lines)):
                if 's_waitcnt vmcnt(0)' in lines[j] and 'ASMSTART' in lines[j - 1]:
                    return j
            return len(lines)
        bad = []
        for lo_, hi_ in ((pro[-1] + 1, window(pro[-1])), (last[-1] + 1, window(hi))):
            for i in range(lo_, hi_):
                l = lines[i]
                if l.startswith('\t') and not l.strip().startswith(';') and 'v_mfma' not in l and regs(l) & carried:
                    bad.append((i, l.strip()))
        print(f'{name}: loop-carried ring VGPRs {sorted(carried)[0]}..{sorted(carried)[-1]} ({len(carried)}), same registers at the back edge: {same}, '
              f'touched while in flight outside the K loop: {len(bad)}')
        for b_ in bad[:10]:
            print('    line', b_[0], b_[1])
        ok &= same and not bad
    return ok


if __name__ == '__main__':
    sys.exit(0 if check(sys.argv[1]) else 1)
