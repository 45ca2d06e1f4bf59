#!/usr/bin/env python
"""Throughput of the fused match kernel at retrieval-like sizes (development aid)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from witw_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bo', type=int, default=8192)
    ap.add_argument('--bs', type=int, default=1024)
    ap.add_argument('--we', type=int, default=64)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    ov = torch.randn((a.bo, 16, 4, 64), device=dev)
    su = torch.randn((a.bs, 16, 4, a.we), device=dev)
    for _ in range(2):
        ops.match_fwd(ov, su)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 3
    for _ in range(n):
        ori, d = ops.match_fwd(ov, su)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * a.bo * a.bs * 64 * 64 * a.we
    print('match Bo=%d Bs=%d We=%d: %.3f ms  %.1f TF/s  (%.1f M pairs/s)' % (a.bo, a.bs, a.we, ms, fl / ms / 1e9,
                                                                        a.bo * a.bs / ms / 1e3))


if __name__ == '__main__':
    main()
