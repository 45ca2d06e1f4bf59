#!/usr/bin/env python
"""cvig_baseline (BASELINE config 1 shape: 32 pairs, ground 500x500, overhead 512x512) on one GPU: eval and train step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from witw_amd import cvig_baseline as cb, cvig_fov, synth  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device('cuda:0')
    xs = torch.from_numpy(synth.images_u8(1, 1, (B, 3, 500, 500))).to(dev)
    xo = torch.from_numpy(synth.images_u8(1, 2, (B, 3, 512, 512))).to(dev)
    se, oe = cb.SurfaceEncoder().to(dev), cb.OverheadEncoder().to(dev)
    flops = 2 * (6.64e9 + 7.17e9) * B          # MACs per image
    modes = {'eval': (False,), 'train': (True,)}.get(sys.argv[2] if len(sys.argv) > 2 else '', (False, True))
    for train in modes:
        se.train(train)
        oe.train(train)
        opt = cvig_fov.Adam(list(se.parameters()) + list(oe.parameters()))

        def step():
            with torch.set_grad_enabled(train):
                es, eo = se(xs), oe(xo)
                loss = cb.exhaustive_minibatch_triplet_loss(es, eo)
                if train:
                    opt.zero_grad()
                    loss.backward()
                    opt.step()
            return loss
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print('cvig_baseline %s B=%d: %.2f ms/step  %.1f pairs/s  (forward %.1f TF/s algorithmic%s)' % (
            'train' if train else 'eval ', B, ms, B / ms * 1e3, flops / ms / 1e9, ', x3 with backward' if train else ''))


if __name__ == '__main__':
    main()
