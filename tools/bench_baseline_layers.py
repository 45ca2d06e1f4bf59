"""Per-block timing of one cvig_baseline encoder (overhead, 32 x 512 x 512, eval path): run on the GPU box from the repo root."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from witw_amd import cvig_baseline as cb, ops, synth
dev = torch.device('cuda:0')
x = torch.from_numpy(synth.images_u8(1, 1, (32, 3, 512, 512))).to(dev)
enc = cb.OverheadEncoder().to(dev).eval()
for _ in range(3):
    enc(x)
torch.cuda.synchronize()
ops.PROFILE = []
for _ in range(5):
    enc(x)
torch.cuda.synchronize()
prof, ops.PROFILE = ops.PROFILE, None
n = len(prof) // 5
for i in range(n):
    ms = sum(prof[i + k * n][2].elapsed_time(prof[i + k * n][3]) for k in range(5)) / 5
    v, fl = prof[i][0], prof[i][1]
    print('launch %d variant %s  %.3f ms  %.1f GFLOP  %.1f TF/s' % (i, v, ms, fl / 1e9, fl / ms / 1e9))
