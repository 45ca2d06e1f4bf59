#!/usr/bin/env python
"""List compiler-inserted s_waitcnt vmcnt(...) (i.e. outside ;;#ASMSTART .. ;;#ASMEND) that sit inside a loop containing MFMAs, per
kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only). In a software-pipelined loop whose loads are issued by inline asm
such a wait usually is the compiler protecting a register it believes a load from BEFORE the loop still owns (match_dft.hip,
round 5: a vmcnt(0) behind the first staging DMA of every step). Test infrastructure.

    python tools/waitcnt_scan.py file.s
"""
import re
import sys


def main():
    kernel, lines = None, []
    out = {}
    for raw in open(sys.argv[1]):
        s = raw.split(';')[0].strip() if not raw.strip().startswith(';;#') else raw.strip()
        if s.endswith(':') and s.startswith('_Z'):
            kernel, lines = s[:-1], []
            out[kernel] = lines
            continue
        if kernel is not None and s:
            lines.append(s)
    for k, ls in out.items():
        labels = {l[:-1]: i for i, l in enumerate(ls) if l.endswith(':')}
        loops = []
        for i, l in enumerate(ls):
            m = re.match(r's_cbranch_\w+\s+(\S+)', l) or re.match(r's_branch\s+(\S+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        hits = []
        for a, b in loops:
            body = ls[a:b + 1]
            n_mfma = sum(1 for l in body if l.startswith('v_mfma'))
            if n_mfma < 8:
                continue
            in_asm, found = False, []
            for l in body:
                if l.startswith(';;#ASMSTART'):
                    in_asm = True
                elif l.startswith(';;#ASMEND'):
                    in_asm = False
                elif not in_asm and l.startswith('s_waitcnt') and 'vmcnt' in l:
                    found.append(l)
            if found:
                hits.append((b - a, n_mfma, found))
        for size, n_mfma, found in hits:
            print('%s\n    loop of %d lines, %d MFMAs: %s' % (k[:120], size, n_mfma, ', '.join(found[:8])))


if __name__ == '__main__':
    main()
