"""cvig_baseline training losses over the first steps of the bench batch (run on the GPU box from the repo root)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from witw_amd import cvig_baseline as cb, cvig_fov, synth
dev = torch.device('cuda:0')
B = 32
xs = torch.from_numpy(synth.images_u8(4242, 1, (B, 3, 500, 500))).to(dev)
xo = torch.from_numpy(synth.images_u8(4242, 2, (B, 3, 512, 512))).to(dev)
se, oe = cb.SurfaceEncoder().to(dev).train(), cb.OverheadEncoder().to(dev).train()
opt = cvig_fov.Adam(list(se.parameters()) + list(oe.parameters()), lr=float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3)
for i in range(8):
    es, eo = se(xs), oe(xo)
    l = cb.exhaustive_minibatch_triplet_loss(es, eo)
    opt.zero_grad()
    l.backward()
    gn = sum(float(p.grad.norm()) ** 2 for p in se.parameters() if p.grad is not None) ** 0.5
    opt.step()
    D = torch.cdist(es.detach(), eo.detach()) ** 2
    print(i, 'loss', float(l), 'grad norm', gn, 'embed abs mean', float(es.abs().mean()), 'nan', bool(torch.isnan(es).any()),
          'd_pos mean', float(D.diag().mean()), 'd mean', float(D.mean()))
