"""Time the value-only instantiation of the spectral match (no orientation output, full-width surfaces); with a
-DWITW_DFT_PHASES=1 build and WITW_DFT_STAMPS=2 the kernel prints its phase sums (test infrastructure)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from witw_amd import ops
G, Q = 32768, 4096
gen = torch.Generator(device='cuda'); gen.manual_seed(1)
ov = torch.randn((G, 16, 4, 64), generator=gen, device='cuda')
su = torch.randn((Q, 16, 4, 64), generator=gen, device='cuda')
sg, sq = ops.match_spectrum(ov, overhead=True), ops.match_spectrum(su, overhead=False)
f = lambda: ops.match_fwd_dft(ov, su, spec_ov=sg, spec_su=sq, want_orientation=False)
f(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): f()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print('value-only: %.2f ms  (%.1f TF/s of 21,120 FLOP/pair)' % (dt * 1e3, G * Q * 21120 / dt / 1e12))
