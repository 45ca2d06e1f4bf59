#!/usr/bin/env python
"""Time the full cvig_fov training step (model/cvig_fov.py:444-461) on one GPU (development aid)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from witw_amd import cvig_fov, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--fov', type=int, default=360)
    ap.add_argument('--steps', type=int, default=5)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    ws = int(a.fov / 360 * 512)
    w = synth.fov_dsm_weights(1)
    se = cvig_fov.FOV_DSM(False, weights=w).to(dev).train()
    oe = cvig_fov.FOV_DSM(True, weights=w).to(dev).train()
    opt = cvig_fov.Adam(list(se.parameters()) + list(oe.parameters()), lr=1e-5)
    xs = torch.from_numpy(synth.normalized_images(1, 0, (a.batch, 3, 128, ws))).to(dev)
    xo = torch.from_numpy(synth.normalized_images(1, 1, (a.batch, 3, 128, 512))).to(dev)

    def step():
        s_emb = se(xs)
        o_emb = oe(xo)
        ori, dist = cvig_fov.match(o_emb, s_emb)
        loss = cvig_fov.triplet_loss(dist)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    fl = 124.15e9 * a.batch if a.fov == 360 else float('nan')
    print('train step B=%d fov=%d: %.2f ms  -> %.1f pairs/s  (%.1f TF/s algorithmic)  loss %.5f' % (
        a.batch, a.fov, dt * 1e3, a.batch / dt, fl / dt / 1e12, loss.item()))


if __name__ == '__main__':
    main()
