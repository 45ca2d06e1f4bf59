#!/usr/bin/env python
"""Per-layer timing of the FOV_DSM conv stack on one GPU (development aid, not the headline bench)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from witw_amd import ops, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--width', type=int, default=512)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--circ', type=int, default=1)
    ap.add_argument('--bf16', action='store_true')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    w = synth.fov_dsm_weights(1)
    x = torch.from_numpy(synth.normalized_images(1, 0, (a.batch, 3, 128, a.width))).to(dev)
    PK = ops.PackedConvBf16 if a.bf16 else ops.PackedConv
    packed = {i: PK(torch.from_numpy(w[i][0]).to(dev), torch.from_numpy(w[i][1]).to(dev)) for i in w}
    h = ops.nchw_to_nhwc_bf16(x) if a.bf16 else ops.nchw_to_nhwc8(x)

    def conv(h, pk, **kw):
        if a.bf16:
            return ops.conv3x3_bf16_fwd(h, pk, stride_h=kw['stride_h'], circular=kw['circular'], relu=kw['relu'], pool=kw['pool'],
                                        out_nchw_f32=(pk.cout == 16))
        return ops.conv3x3_fwd(h, pk, **kw)
    total_ms, total_fl = 0.0, 0.0
    for (idx, cin, cout, sh, relu, pool, drop) in synth.FOV_LAYERS:
        B, H, W, C = h.shape
        Ho = (H - 1) // sh + 1
        fl = 2.0 * cin * cout * 9 * Ho * W * B
        for _ in range(2):
            y = conv(h, packed[idx], stride_h=sh, circular=bool(a.circ), relu=relu, pool=pool)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            y = conv(h, packed[idx], stride_h=sh, circular=bool(a.circ), relu=relu, pool=pool)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        total_ms += ms
        total_fl += fl
        print('L%-2d %4d->%-4d in %3dx%-3d s%d pool%d  %8.3f ms  %7.2f TF/s' % (idx, cin, cout, H, W, sh, pool, ms,
                                                                             fl / ms / 1e9), flush=True)
        h = y
    print('total %.3f ms  %.2f TF/s  -> %.1f images/s per encoder' % (total_ms, total_fl / total_ms / 1e9,
                                                                  a.batch / total_ms * 1e3))


if __name__ == '__main__':
    main()
