import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from witw_amd import synth, cvig_baseline, cvig_fov
from tests.test_baseline_gpu import _load_encoder
g = np.load('tests/golden/baseline_train.npz')
seed, B = int(g['seed']), int(g['B'])
xs = torch.from_numpy(synth.images_u8(seed, 40, (B, 3, 400, 400))).cuda()
xo = torch.from_numpy(synth.images_u8(seed, 41, (B, 3, 416, 416))).cuda()
se = _load_encoder(cvig_baseline.SurfaceEncoder, seed + 10).train()
oe = _load_encoder(cvig_baseline.OverheadEncoder, seed + 11).train()
es, eo = se(xs), oe(xo)
loss = cvig_baseline.exhaustive_minibatch_triplet_loss(es, eo)
loss.backward()
print('loss', loss.item(), float(g['loss']), 'embed diff', float((es.detach().cpu() - torch.from_numpy(g['embed_surface'])).abs().max()),
      float((eo.detach().cpu() - torch.from_numpy(g['embed_overhead'])).abs().max()))
named = {('surface.' + n): p for n, p in se.named_parameters()}
named.update({('overhead.' + n): p for n, p in oe.named_parameters()})
for name in g['names']:
    p = named[str(name)]
    ref = g['gsamp:' + str(name)]
    got = p.grad.detach().reshape(-1).cpu()
    got_s = got[::max(1, got.numel() // 129)].numpy()
    gn = float(g['gnorm:' + str(name)])
    print('%-26s norm rel %.2e  samp rel %.2e' % (name, abs(got.double().norm().item() - gn) / gn, np.linalg.norm(got_s - ref) / np.linalg.norm(ref)))
