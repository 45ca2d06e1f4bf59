import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from tests.test_trainstep_golden import load_case, sample
from witw_amd import cvig_fov
g, xs, xo, w, drops = load_case('tests/golden')
dev = torch.device('cuda:0')
se = cvig_fov.FOV_DSM(False, weights=w).to(dev).train()
oe = cvig_fov.FOV_DSM(True, weights=w).to(dev).train()
s_emb = se(xs.to(dev), dropout_scales={k: v.to(dev) for k, v in drops['s'].items()})
o_emb = oe(xo.to(dev), dropout_scales={k: v.to(dev) for k, v in drops['o'].items()})
ori, dist = cvig_fov.match(o_emb, s_emb)
loss = cvig_fov.triplet_loss(dist)
loss.backward()
named = {('s.' + n): p for n, p in se.named_parameters()}
named.update({('o.' + n): p for n, p in oe.named_parameters()})
for name in g['names']:
    got = sample(named[str(name)].grad.detach().cpu()).numpy(); ref = g['gsamp:'+str(name)]
    d = np.abs(got-ref)
    print('%-45s n=%4d  relL2 %.2e  maxabs %.2e (max|ref| %.2e)  #>1e-3rel %d' % (name, len(ref), np.linalg.norm(got-ref)/np.linalg.norm(ref), d.max(), np.abs(ref).max(), int((d > 1e-3*np.abs(ref).max()).sum())))
