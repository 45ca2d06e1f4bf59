"""Code-generation properties the spectral match's speed rests on (csrc/match_dft.hip), checked on the assembly hipcc
writes for gfx950 -- no GPU needed. Each of them was lost at least once during round 5 without a single result changing:
  * no register of the plain / value-only instantiations is spilled (the GAP one may spill a few),
  * in the step loop no VALU instruction writes a register that one of the MFMAs just issued names as its A / B operand
    (tools/war_scan.py; 7 % of the kernel when the compiler formed the read addresses in the registers the reads overwrite),
  * the step loop holds exactly ONE wait on vmcnt, the hand-written one in front of the barrier (a compiler-inserted vmcnt(0)
    behind the first staging DMA cost 5 % in the value-only instantiation)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


@pytest.fixture(scope='module')
def listing(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip('no hipcc')
    out = str(tmp_path_factory.mktemp('isa') / 'match_dft.s')
    src = os.path.join(ROOT, 'witw_amd', 'csrc', 'match_dft.hip')
    sys.path.insert(0, ROOT)
    from witw_amd import build
    flags = [f for f in build.FLAGS if f not in ('-fPIC',)]
    subprocess.check_call([HIPCC] + flags + ['--cuda-device-only', '-S', '-o', out, src], stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernels(text):
    """mangled name -> (body lines, metadata text) of every match_dft_kernel instantiation"""
    bodies = {}
    for m in re.finditer(r'^(_ZN\S*match_dft_kernel\S*):.*?s_endpgm', text, re.S | re.M):
        bodies[m.group(1)] = m.group(0).splitlines()
    return bodies


def _step_loop(lines):
    """the innermost loop that holds the 80 MFMAs of a step: from its header label to the backward branch"""
    labels = {l.split(':')[0]: i for i, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)}
    best = None
    for i, l in enumerate(lines):
        m = re.match(r'\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = lines[labels[m.group(1)]:i + 1]
            n = sum(1 for b in body if b.strip().startswith('v_mfma'))
            if n >= 80 and (best is None or len(body) < len(best)):
                best = body
    assert best is not None, 'no step loop found'
    return best


def test_plain_and_value_only_instantiations_do_not_spill(listing):
    spills = dict(re.findall(r'\.name:\s+(\S*match_dft_kernel\S*)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)', listing))
    assert len(spills) == 4, sorted(spills)
    for name, n in spills.items():
        rec, gap = 'ILb1E' in name, 'ILb0ELb1E' in name
        if not rec and not gap:
            assert int(n) == 0, (name, n)
        if gap:
            assert int(n) <= 16, (name, n)


def test_step_loop_has_one_vmcnt_wait_and_no_operand_overwrites(listing):
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    bodies = _kernels(listing)
    product = [k for k in bodies if 'ILb0E' in k.split('match_dft_kernel')[1][:5]]      # REC = false
    assert len(product) == 3, sorted(bodies)
    reg = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')

    def regs(tok):
        out = set()
        for m in reg.finditer(tok):
            out.update([int(m.group(1))] if m.group(1) is not None else range(int(m.group(2)), int(m.group(3)) + 1))
        return out

    for k in product:
        loop = [l.split(';')[0].strip() for l in _step_loop(bodies[k])]
        loop = [l for l in loop if l]
        assert sum(1 for l in loop if l.startswith('v_mfma')) == 80, k
        waits = [l for l in loop if l.startswith('s_waitcnt') and 'vmcnt' in l]
        gap = 'ILb0ELb1E' in k      # the GAP instantiation reloads one spilled address per tile; its first use (group 14) carries a wait
        assert waits == ['s_waitcnt vmcnt(0)'] * (2 if gap and len(waits) == 2 else 1), (k, waits)
        assert not any(l.startswith('scratch_') or l.startswith('v_accvgpr') for l in loop), k
        hits, last = 0, set()
        for l in loop:
            op, args = l.split()[0], [a.strip() for a in l[len(l.split()[0]):].split(',')]
            if op.startswith('v_mfma'):
                last = regs(args[1]) | regs(args[2])
            elif op.startswith('s_barrier') or op.startswith('s_cbranch'):
                last = set()
            elif op.startswith('v_') and not op.startswith('v_cmp') and regs(args[0]) & last:
                hits += 1
        assert hits <= 2, (k, hits)
