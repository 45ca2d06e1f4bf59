import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from witw_amd import ops
gen = torch.Generator(device='cuda'); gen.manual_seed(1)
ov = torch.randn((16384, 16, 4, 64), generator=gen, device='cuda')
su = torch.randn((4096, 16, 4, 64), generator=gen, device='cuda')
sg, sq = ops.match_spectrum(ov), ops.match_spectrum(su)
for _ in range(2):
    ops.match_fwd_dft(ov, su, spec_ov=sg, spec_su=sq)
torch.cuda.synchronize()
print('done', flush=True)
