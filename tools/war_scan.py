#!/usr/bin/env python
"""Scan gfx950 assembly (hipcc -S --cuda-device-only) for the pattern that cost the spectral match 7 % (docs/experiments.md,
round 5): a VALU instruction that WRITES a register which one of the last MFMAs still names as its A / B operand. On this chip
such a write waits until that MFMA has left the matrix pipe. Test infrastructure.

    python tools/war_scan.py file.s [window]      # window = how many MFMAs back to look (default 1)
"""
import re
import sys
from collections import Counter

REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def main():
    path = sys.argv[1]
    window = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    kernel, recent, hits, mfmas = None, [], Counter(), Counter()
    examples = {}
    for line in open(path):
        s = line.split(';')[0].strip()
        if s.endswith(':') and s.startswith('_Z'):
            kernel, recent = s[:-1], []
            continue
        if not s or s[0] in ';.' or kernel is None:
            continue
        op = s.split()[0]
        args = s[len(op):].split(';')[0]
        parts = [a.strip() for a in args.split(',')]
        if op.startswith('v_mfma') or op.startswith('v_smfma'):
            mfmas[kernel] += 1
            recent.append(regs(parts[1]) | regs(parts[2]))
            recent = recent[-window:]
            continue
        if op.startswith('s_barrier') or op.startswith('s_cbranch') or op.startswith('s_branch'):
            recent = []
            continue
        if op.startswith('v_') and not op.startswith('v_cmp') and recent:
            dst = regs(parts[0])
            if any(dst & r for r in recent):
                hits[kernel] += 1
                examples.setdefault(kernel, s)
    for k, n in hits.most_common():
        print('%6d writes / %6d MFMAs  %s\n        e.g. %s' % (n, mfmas[k], k[:110], examples[k]))


if __name__ == '__main__':
    main()
