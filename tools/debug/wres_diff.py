"""Where does conv3x3_bf16_wres_kernel differ from the tiled 32x32x16 kernel? (debug aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from witw_amd import ops
dev = torch.device('cuda:0')
B, H, W, cout = 64, 64, 256, 128
g = torch.Generator(device='cuda'); g.manual_seed(3)
x = torch.randn((B, H, W, 64), generator=g, device=dev).bfloat16()
w = torch.randn((cout, 64, 3, 3), generator=g, device=dev) * 0.06
b = torch.randn((cout,), generator=g, device=dev) * 0.1
pk = ops.PackedConvBf16(w, b)
for circ in (True, False):
    y = ops.conv3x3_bf16_fwd(x, pk, circular=circ, relu=True)
    print(ops.last_kernel_variant())
    ops.bf16_wres(False); ops.bf16_mfma16(False)
    t = ops.conv3x3_bf16_fwd(x, pk, circular=circ, relu=True)
    ops.bf16_wres(True); ops.bf16_mfma16(True)
    d = (y.view(torch.int16) != t.view(torch.int16))
    print('circ', circ, 'mismatching elements', int(d.sum()), 'of', d.numel(), 'max abs', float((y.float() - t.float()).abs().max()))
    if d.any():
        idx = d.nonzero()
        print('images', idx[:, 0].unique()[:10].tolist(), 'n', idx[:, 0].unique().numel())
        print('rows', idx[:, 1].unique()[:40].tolist())
        print('cols', idx[:, 2].unique()[:40].tolist(), 'n', idx[:, 2].unique().numel())
        print('chans', idx[:, 3].unique()[:40].tolist(), 'n', idx[:, 3].unique().numel())
        print('first', idx[:8].tolist())
        i0 = idx[0].tolist()
        print('y', float(y[tuple(i0)]), 't', float(t[tuple(i0)]))
