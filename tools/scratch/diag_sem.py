import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import cvig_fov_oracle as O
from witw_amd import synth, cvig_semantic
g = np.load('tests/golden/trainstep_semantic.npz')
seed, B = int(g['seed']), int(g['B'])
w5 = synth.fov_dsm_weights(seed, in_channels=5)
xo = torch.from_numpy(synth.normalized_images(seed, 2, (B, 5, 128, 512)))
drops = {i: torch.from_numpy(g['drop_o_%d' % i]) for i in (17, 19, 21)}
wt = {k: (torch.from_numpy(a.copy()), torch.from_numpy(c.copy())) for k, (a, c) in w5.items()}
with torch.no_grad():
    _, acts = O.fov_dsm_forward(xo, wt, True, dropout_scales=drops, return_activations=True)
dev = torch.device('cuda:0')
oe = cvig_semantic.FOV_DSM(True, weights=w5).to(dev).train()
oe.keep_activations = True
out = oe(xo.to(dev), dropout_scales={k: v.to(dev) for k, v in drops.items()})
for idx in (0, 2, 5, 7, 10, 12, 14, 17):
    y = oe._last_kept[idx][1].permute(0, 3, 1, 2).cpu()
    a = acts[idx]
    d = (y - a).abs()
    gm = (y > 0) != (a > 0)
    print(idx, 'max|act| %.1f std %.2f max diff %.2e  gate mismatches %d, values at mismatches (gpu, cpu): %s' % (
        a.max(), a.std(), d.max(), int(gm.sum()), [(float(u), float(v)) for u, v in zip(y[gm][:6], a[gm][:6])]))
idx = 10
y = oe._last_kept[idx][1].permute(0, 3, 1, 2).cpu().contiguous()
gate = y > 0
ref = torch.from_numpy(g['gcount:o:%d' % idx])
fr = torch.from_numpy(g['gfrag:o:%d' % idx]); fv = torch.from_numpy(g['gfragv:o:%d' % idx]).bool()
cnt0 = gate.sum(dim=(2, 3)).to(torch.int32)
print('raw count diffs at (b,c):', torch.nonzero(cnt0 != ref).tolist(), (cnt0 - ref)[cnt0 != ref].tolist())
flat = gate.reshape(-1).clone(); flat[fr] = fv
cnt1 = flat.view_as(gate).sum(dim=(2, 3)).to(torch.int32)
print('after reconcile diffs at (b,c):', torch.nonzero(cnt1 != ref).tolist(), (cnt1 - ref)[cnt1 != ref].tolist())
gm = (y > 0) != (acts[idx] > 0)
pos = torch.nonzero(gm).tolist()
print('mismatch positions vs box oracle', pos)
C, H, W = y.shape[1:]
for (b, c, h, w) in pos:
    fi = ((b * C + c) * H + h) * W + w
    print((b, c, h, w), 'flat', fi, 'in fragile list:', bool((fr == fi).any()), 'ref side', fv[fr == fi].tolist())
