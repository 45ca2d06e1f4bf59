import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from witw_amd import cvig_fov, ops, synth
B=128
ov = torch.from_numpy(synth.embeddings(1, 1, (B, 16, 4, 64))).cuda()
su = torch.from_numpy(synth.embeddings(1, 2, (B, 16, 4, 64))).cuda()
ori0, d0, s0 = ops.match_fwd(ov, su, want_score=True)
ori1, d1, s1 = ops.match_fwd(torch.roll(ov, shifts=-9, dims=3).contiguous(), su, want_score=True)
bad = ((ori0 - 9) % 64) != ori1
print('bad', int(bad.sum()), 'of', bad.numel(), 'score max diff', float((s0 - s1).abs().max()))
idx = bad.nonzero()[:5]
for o, s in idx.tolist():
    print(o, s, int(ori0[o, s]), int(ori1[o, s]), float(s0[o, s]), float(s1[o, s]))
w = synth.fov_dsm_weights(11)
dev = torch.device('cuda:0')
s_enc = cvig_fov.FOV_DSM(False, weights=w).to(dev).eval(); ov_enc = cvig_fov.FOV_DSM(True, weights=w).to(dev).eval()
batch = torch.from_numpy(synth.normalized_images(12, 0, (B, 3, 128, 512))).cuda()
with torch.no_grad():
    su, ov = s_enc(batch), ov_enc(batch)
ori0, d0, s0 = ops.match_fwd(ov, su, want_score=True)
ori1, d1, s1 = ops.match_fwd(torch.roll(ov, shifts=-9, dims=3).contiguous(), su, want_score=True)
bad = ((ori0 - 9) % 64) != ori1
print('enc: bad', int(bad.sum()), 'of', bad.numel(), 'score max diff', float((s0 - s1).abs().max()), 'emb std over W', float(ov.std(dim=3).mean()), float(ov.abs().mean()))
for o, s in bad.nonzero()[:6].tolist():
    print(o, s, int(ori0[o, s]), int(ori1[o, s]), float(s0[o, s]), float(s1[o, s]))
