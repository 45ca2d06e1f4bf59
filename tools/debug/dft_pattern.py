"""Which (overhead, surface) positions of a 32 x 32 tile disagree between the spectral and the direct match (test infrastructure)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from witw_amd import ops
gen = torch.Generator(device='cuda'); gen.manual_seed(1)
ov = torch.randn((64, 16, 4, 64), generator=gen, device='cuda')
su = torch.randn((64, 16, 4, 64), generator=gen, device='cuda')
o1, d1 = ops.match_fwd_dft(ov, su, spec_ov=ops.match_spectrum(ov, overhead=True), spec_su=ops.match_spectrum(su, overhead=False))
o0, d0 = ops.match_fwd(ov, su)
bad = (o1 != o0)
print('bad fraction', float(bad.float().mean()))
print('by surface column (s % 32):', bad.float().mean(0).reshape(2, 32).mean(0).cpu().numpy().round(2))
print('by overhead row (o % 32):', bad.float().mean(1).reshape(2, 32).mean(0).cpu().numpy().round(2))
i = bad.nonzero()[:12]
for o, s in i.tolist():
    print(o, s, 'dft', int(o1[o, s]), 'direct', int(o0[o, s]), float(d1[o, s]), float(d0[o, s]))
