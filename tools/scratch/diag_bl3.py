import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from witw_amd import synth, cvig_baseline as cb, ops
from oracle import cvig_baseline_oracle as OB
dev = torch.device('cuda:0')
def load(seed):
    prm = synth.baseline_params(seed)
    e = cb.SurfaceEncoder()
    with torch.no_grad():
        for i, q in enumerate(prm, 1):
            getattr(e, 'conv%d' % i).weight.copy_(torch.from_numpy(q['w'])); getattr(e, 'conv%d' % i).bias.copy_(torch.from_numpy(q['b']))
            bn = getattr(e, 'bn%d' % i)
            bn.weight.copy_(torch.from_numpy(q['gamma'])); bn.bias.copy_(torch.from_numpy(q['beta']))
            bn.running_mean.copy_(torch.from_numpy(q['mean'])); bn.running_var.copy_(torch.from_numpy(q['var']))
    return e.to(dev).eval(), [{k: torch.from_numpy(v) for k, v in q.items()} for q in prm]
for pseed in (4243, 20261004):
    enc, prm = load(pseed)
    for hw in (512, 500, 510):
        for B in (1, 2, 3, 4, 8):
            x = torch.from_numpy(synth.images_u8(9, B, (B, 3, hw, hw)))
            with torch.no_grad():
                e = enc(x.to(dev)).cpu()
                ref = OB.encoder_forward(x, prm)
            d = (e - ref).abs()
            print(pseed, hw, B, 'max diff %.2e' % d.max(), 'per-sample', ['%.1e' % v for v in d.max(dim=1).values.tolist()], 'per-block', ['%.1e' % d[:, 512*j:512*(j+1)].max() for j in range(3)])
