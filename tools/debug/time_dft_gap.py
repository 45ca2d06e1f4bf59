"""Time witw_match_fwd_dft_gap (narrow surfaces) at a retrieval-shaped problem (test infrastructure)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from witw_amd import ops
G, Q, We = 32768, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 12
gen = torch.Generator(device='cuda'); gen.manual_seed(1)
ov = torch.randn((G, 16, 4, 64), generator=gen, device='cuda')
su = torch.randn((Q, 16, 4, We), generator=gen, device='cuda')
sg, sq = ops.match_spectrum(ov, overhead=True), ops.match_spectrum(su, overhead=False)
for gap in (True, False):
    f = lambda: ops.match_fwd_dft(ov, su, spec_ov=sg, spec_su=sq, want_orientation=False, want_gap=gap)
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize()
    print('We=%d gap=%s: %.2f ms' % (We, gap, (time.perf_counter() - t0) / 3 * 1e3))
