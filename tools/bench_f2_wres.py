"""Times of the two weight-resident bf16 kernels at the config-4 shapes (B = 128): conv_first2_bf16_kernel (layers 0 + 2, 5-channel input)
and conv3x3_bf16_wres_kernel (layer 5). Run on the GPU box from the repo root; WITW_LIB selects an alternative build (tools/ab_lib.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from witw_amd import ops, synth
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
C = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w = synth.fov_dsm_weights(5, in_channels=C)
x = torch.from_numpy(synth.normalized_images(5, C, (B, C, 128, 512))).to(dev)
pf = ops.PackedFirstConv(torch.from_numpy(w[0][0]).to(dev), torch.from_numpy(w[0][1]).to(dev), bf16=True)
p2 = ops.PackedConvBf16(torch.from_numpy(w[2][0]).to(dev), torch.from_numpy(w[2][1]).to(dev))
p5 = ops.PackedConvBf16(torch.from_numpy(w[5][0]).to(dev), torch.from_numpy(w[5][1]).to(dev))


def timed(fn, n=30):
    for _ in range(3):
        y = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        y = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, y


t2, h = timed(lambda: ops.conv_first2_bf16(x, pf, p2, circular=True))
t5, y = timed(lambda: ops.conv3x3_bf16_fwd(h, p5, circular=True, relu=True))
fl2 = 2.0 * (C + 64) * 64 * 9 * 128 * 512 * B
fl5 = 2.0 * 64 * 128 * 9 * 64 * 256 * B
print('first2 %.4f ms %.0f TF/s | L5 %s %.4f ms %.0f TF/s' % (t2, fl2 / t2 / 1e9, ops.last_kernel_variant(), t5, fl5 / t5 / 1e9))
