import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import cvig_fov_oracle as O
from witw_amd import ops
sys.path.insert(0, 'tests')
def layer(seed, B, H, W, cin, cout):
    g = np.random.Generator(np.random.Philox(key=[seed, cin * 1000 + cout]))
    x = torch.from_numpy(g.standard_normal((B, cin, H, W), dtype=np.float32))
    w = torch.from_numpy(g.standard_normal((cout, cin, 3, 3), dtype=np.float32) * (2.0 / (9 * cin)) ** 0.5)
    b = torch.from_numpy(g.standard_normal((cout,), dtype=np.float32) * 0.1)
    return x, w, b
dev = torch.device('cuda:0')
for seed in (31, 32, 33):
  for (B, H, W, cin, cout, sh, circ) in ((128, 32, 64, 512, 256, 2, True), (128, 32, 64, 512, 256, 2, False), (16, 32, 64, 512, 256, 2, True)):
    x, w, b = layer(seed, B, H, W, cin, cout)
    x = x.bfloat16().float()
    pk = ops.PackedConvBf16(w.to(dev), b.to(dev))
    y = ops.conv3x3_bf16_fwd(x.to(dev).permute(0, 2, 3, 1).contiguous().bfloat16(), pk, stride_h=sh, circular=circ, relu=True)
    var = ops.last_kernel_variant()
    sel = sorted({0, B // 3, B // 2 + 1, B - 1})
    ref = torch.relu(O.conv3x3(x[sel], w.bfloat16().float(), b, sh, circ))
    ref64 = torch.relu(O.conv3x3(x[sel].double(), w.bfloat16().double(), b.double(), sh, circ))
    got = y[sel].float().cpu().permute(0, 3, 1, 2)
    r16 = ref.bfloat16().float()
    bad = (got - r16).abs() > 2.0 ** -7 * torch.maximum(r16.abs(), got.abs()) + 1e-6
    print(seed, (B, H, W, cin, cout, sh, circ), var, 'bad', int(bad.sum()), 'of', bad.numel(), 'max fp32-vs-fp64 ref err', float((ref - ref64).abs().max()))
    for idx in bad.nonzero()[:10].tolist():
        i = tuple(idx)
        print('   at', i, 'got', float(got[i]), 'ref32', float(ref[i]), 'ref64', float(ref64[i]), 'r16', float(r16[i]))
