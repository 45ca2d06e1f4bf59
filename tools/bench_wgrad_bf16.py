#!/usr/bin/env python
"""bf16 weight-gradient kernels, layer by layer at the bench batch (development aid): the NHWC-direct kernel (round 5) against the
batch-octet one (round 1, incl. its two re-layout passes), per trainable layer of FOV_DSM (model/cvig_fov.py:263-278) and
cvig_semantic's layer 0 (model/cvig_semantic.py:301-309). HIP events over `--iters` back-to-back calls; FLOP = 2*cin*cout*9*Ho*W*B."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from witw_amd import ops  # noqa: E402

LAYERS = [  # name, H, W, Cin (stored), cin_real, Cout, stride_h
    ('L17', 16, 64, 256, 256, 512, 1), ('L19', 16, 64, 512, 512, 512, 1), ('L21', 16, 64, 512, 512, 512, 1),
    ('L23', 16, 64, 512, 512, 256, 2), ('L25', 8, 64, 256, 256, 64, 2), ('L27', 4, 64, 64, 64, 16, 1),
    ('sem L0', 128, 512, 16, 5, 64, 1), ('sem L2', 128, 512, 64, 64, 64, 1), ('sem L5', 64, 256, 64, 64, 128, 1),
    ('sem L7', 64, 256, 128, 128, 128, 1), ('sem L10', 32, 128, 128, 128, 256, 1), ('sem L12', 32, 128, 256, 256, 256, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--layers', default='')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    tot = {'nhwc': 0.0, 'octet': 0.0}
    for name, H, W, cin, cin_real, cout, sh in LAYERS:
        if a.layers and name not in a.layers.split(','):
            continue
        B = a.batch
        Ho = (H + 2 - 3) // sh + 1
        x = torch.randn((B, H, W, cin), generator=g, device=dev).bfloat16()
        dz = torch.randn((B, Ho, W, cout), generator=g, device=dev).bfloat16()
        fl = 2.0 * cin_real * cout * 9 * Ho * W * B
        res = {}
        for layout in ('nhwc', 'octet'):
            if layout == 'octet' and x.numel() * 2 * 8 >= 2 ** 31:
                continue
            ops.conv3x3_wgrad_bf16(x, dz, cin_real, stride_h=sh, circular=True, layout=layout)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                dw, db = ops.conv3x3_wgrad_bf16(x, dz, cin_real, stride_h=sh, circular=True, layout=layout)
            e1.record()
            e1.synchronize()
            res[layout] = (e0.elapsed_time(e1) / a.iters, dw, db)
            if not name.startswith('sem'):
                tot[layout] += res[layout][0]
        line = '%-8s B=%d %3dx%3d %3d->%3d s%d  %7.1f GF' % (name, B, H, W, cin_real, cout, sh, fl / 1e9)
        for layout, (ms, dw, db) in res.items():
            line += ' | %-5s %7.3f ms %7.1f TF/s' % (layout, ms, fl / ms / 1e9)
        if len(res) == 2:
            d = float((res['nhwc'][1] - res['octet'][1]).abs().max() / res['octet'][1].abs().max())
            line += ' | max rel diff of dW %.1e' % d
        print(line, flush=True)
    print('cvig_fov trainable layers, one encoder: nhwc %.3f ms, octet (incl. re-layout) %.3f ms' % (tot['nhwc'], tot['octet']))


if __name__ == '__main__':
    main()
