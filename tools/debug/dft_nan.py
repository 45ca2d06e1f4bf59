"""Do NaN surfaces disturb their finite neighbours in the spectral / direct match? (test infrastructure)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from witw_amd import ops
rng = np.random.default_rng(11)
ov = torch.from_numpy(rng.standard_normal((40, 16, 4, 64)).astype(np.float32)).cuda()
su = torch.from_numpy(rng.standard_normal((70, 16, 4, 64)).astype(np.float32)).cuda()
mixed = su.clone(); mixed[::2] = float('nan')
for name, fn in (('dft', ops.match_fwd_dft), ('direct', ops.match_fwd)):
    o_ref, d_ref = fn(ov, su)[:2]
    o_mix, d_mix = fn(ov, mixed)[:2]
    same = (o_mix == o_ref)
    print(name, 'finite surfaces unchanged per surface column:', same.float().mean(0)[1::2].cpu().numpy().round(2))
    print(name, 'orientation of NaN surfaces (first 5):', o_mix[0, 0:10:2].cpu().numpy())
sp = ops.match_spectrum(mixed, overhead=False)
t = sp.tensor if hasattr(sp, 'tensor') else sp[0] if isinstance(sp, tuple) else sp
try:
    tt = t if torch.is_tensor(t) else t.spec
    print('spectra finite per surface (first 8):', torch.isfinite(tt).reshape(tt.shape[0], -1).all(1)[:8].cpu().numpy())
except Exception as e:
    print('spectrum object', type(sp), e)
print('--- one NaN surface at a time, 32 x 32')
ov = torch.from_numpy(rng.standard_normal((32, 16, 4, 64)).astype(np.float32)).cuda()
su = torch.from_numpy(rng.standard_normal((32, 16, 4, 64)).astype(np.float32)).cuda()
o_ref, d_ref, s_ref = ops.match_fwd_dft(ov, su, want_score=True)
for bad in (0, 1, 5, 16, 31):
    m = su.clone(); m[bad] = float('nan')
    o, d, s = ops.match_fwd_dft(ov, m, want_score=True)
    changed = [j for j in range(32) if j != bad and not torch.equal(s[:, j], s_ref[:, j])]
    nanrows = [j for j in range(32) if torch.isnan(s[:, j]).any()]
    print('NaN surface', bad, '-> surfaces whose scores changed:', changed, '; with NaN/-inf scores:', nanrows, [float(s[0, j]) for j in changed[:3]])
m = su.clone(); m[3, 0, 0, 0] = float('nan')
sp = ops.match_spectrum(m, overhead=False).data
print('single NaN element in surface 3: non-finite spectra rows:', [j for j in range(32) if not torch.isfinite(sp[j]).all()])
