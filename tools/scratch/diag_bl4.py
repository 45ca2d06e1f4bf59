import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.nn.functional as F
from witw_amd import synth, cvig_baseline as cb, ops, _lib
dev = torch.device('cuda:0')
prm = synth.baseline_params(4243)
enc = cb.SurfaceEncoder()
with torch.no_grad():
    for i, q in enumerate(prm, 1):
        getattr(enc, 'conv%d' % i).weight.copy_(torch.from_numpy(q['w'])); getattr(enc, 'conv%d' % i).bias.copy_(torch.from_numpy(q['b']))
        bn = getattr(enc, 'bn%d' % i)
        bn.weight.copy_(torch.from_numpy(q['gamma'])); bn.bias.copy_(torch.from_numpy(q['beta']))
        bn.running_mean.copy_(torch.from_numpy(q['mean'])); bn.running_var.copy_(torch.from_numpy(q['var']))
enc = enc.to(dev).eval()
lib = _lib.load()
for B in (3, 4):
    x = torch.from_numpy(synth.images_u8(9, B, (B, 3, 512, 512)))
    with torch.no_grad():
        xr = -1. + 2. * (x / 255.)
        h = ops.space_to_depth2(x.to(dev).contiguous(), in_nchw=True, normalize=True, cpad=enc._layer(1)[3])
        vh = vw = 512
        for i in range(1, 8):
            packed, scale, shift, _cp = enc._layer(i)
            q = prm[i - 1]
            xr = F.conv2d(xr, torch.from_numpy(q['w']), torch.from_numpy(q['b']), stride=2)
            xr = F.leaky_relu(xr, 0.2)
            xr = F.batch_norm(xr, torch.from_numpy(q['mean']), torch.from_numpy(q['var']), torch.from_numpy(q['gamma']), torch.from_numpy(q['beta']), training=False, eps=1e-5)
            vh, vw = (vh - 4) // 2 + 1, (vw - 4) // 2 + 1
            Bh, Hh, Wh, Ch = h.shape
            y = ops.conv3x3_fwd(h, packed, relu=False, lrelu_slope=0.2, post_scale=scale, post_shift=shift)
            got = y[:, :vh, :vw, :].permute(0, 3, 1, 2).cpu()
            d = (got - xr).abs()
            print('B', B, 'layer', i, 'in', tuple(h.shape), 'valid', (vh, vw), 'tile_n', lib.witw_conv3x3_tile_n(packed.cout), 'waves', lib.witw_conv3x3_workgroup_waves(Bh, Hh, Wh, packed.cout, 1),
                  'max diff %.2e' % d.max(), 'rows bad', sorted(set(torch.nonzero(d > 1e-3)[:, 2].tolist()))[:8], 'cols bad', sorted(set(torch.nonzero(d > 1e-3)[:, 3].tolist()))[:8])
            if i < 7:
                h = ops.space_to_depth2(y, valid_hw=(vh, vw), cpad=enc._layer(i + 1)[3])
