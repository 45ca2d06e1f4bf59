"""HBM-bound kernels of the step in isolation (B = 128): ground-side resize (3 / 5 channels), overhead polar_from_raw, fp32 first layer.
Run on the GPU box; WITW_FIRST_STAGGER=0 switches the first-layer start offset off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from witw_amd import ops, synth, cvig_fov
from oracle import cvig_fov_oracle as O
dev = torch.device('cuda:0')
B = 128


def timed(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for C in (3, 5):
    g = torch.from_numpy(synth.images_u8(3, C, (B, C, 224, 224))).to(dev)
    o = torch.from_numpy(synth.images_u8(4, C, (B, C, 512, 512))).to(dev)
    mean, std = (list(O.SEM_MEAN), list(O.SEM_STD)) if C == 5 else (list(O.IMG_MEAN), list(O.IMG_STD))
    t = timed(lambda: ops.resize_bilinear(g, (128, 512), mean, std, 3))
    nb = g.numel() * 4 + B * C * 128 * 512 * 4
    print('resize ground C=%d: %.1f us  %.2f TB/s (%.2f of 8)' % (C, t, nb / t / 1e6, nb / t / 1e6 / 8))
    t = timed(lambda: ops.polar_from_raw(o, mean=mean, std=std, n_div255=3))
    nb = o.numel() * 4 + B * C * 128 * 512 * 4
    print('polar_from_raw C=%d: %.1f us  %.2f TB/s (%.2f of 8)' % (C, t, nb / t / 1e6, nb / t / 1e6 / 8))
w = synth.fov_dsm_weights(1)
x = torch.from_numpy(synth.normalized_images(1, 0, (B, 3, 128, 512))).to(dev)
pf = ops.PackedFirstConv(torch.from_numpy(w[0][0]).to(dev), torch.from_numpy(w[0][1]).to(dev), bf16=False)
t = timed(lambda: ops.conv3x3_first_fwd(x, pf, circular=True, relu=True))
nb = x.numel() * 4 + B * 128 * 512 * 64 * 4
print('conv3x3_first_kernel fp32 (stagger %s): %.1f us  %.2f TB/s (%.2f of 8)' % (os.environ.get('WITW_FIRST_STAGGER', '1'), t, nb / t / 1e6, nb / t / 1e6 / 8))
