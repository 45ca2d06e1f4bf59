#!/usr/bin/env python
"""Time the spectral match against the direct kernel at a retrieval-shaped problem (test infrastructure).

    python tools/time_match_dft.py [gallery rows] [queries]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from witw_amd import ops  # noqa: E402


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    Q = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1)
    ov = torch.randn((G, 16, 4, 64), generator=gen, device='cuda')
    su = torch.randn((Q, 16, 4, 64), generator=gen, device='cuda')

    def timed(fn, n=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n, out

    t_spec, spec_g = timed(lambda: ops.match_spectrum(ov, overhead=True))
    t_specq, spec_q = timed(lambda: ops.match_spectrum(su, overhead=False))
    t_dft, (ori1, d1) = timed(lambda: ops.match_fwd_dft(ov, su, spec_ov=spec_g, spec_su=spec_q))
    t_dir, (ori0, d0) = timed(lambda: ops.match_fwd(ov, su), n=1)
    pairs = G * Q
    print('gallery %d x queries %d' % (G, Q))
    print('spectrum: gallery %.2f ms, queries %.2f ms' % (t_spec * 1e3, t_specq * 1e3))
    print('dft    match: %.2f ms  %.3e pairs/s  (%.1f TF/s of its own 21,120 FLOP/pair)' % (t_dft * 1e3, pairs / t_dft,
                                                                                          pairs * 21120 / t_dft / 1e12))
    print('direct match: %.2f ms  %.3e pairs/s  (%.1f TF/s of 524,288 FLOP/pair)' % (t_dir * 1e3, pairs / t_dir,
                                                                                    pairs * 524288 / t_dir / 1e12))
    print('speed-up %.1fx; max |distance diff| %.2e; orientations equal %.5f' % (t_dir / t_dft, float((d1 - d0).abs().max()),
                                                                             float((ori1 == ori0).float().mean())))


if __name__ == '__main__':
    main()
