"""In-kernel phase stamps of conv_first2_bf16_kernel (WITW_F2_STAMPS=1 selects the recording instantiation): run on the GPU box from the repo root."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from witw_amd import cvig_semantic, synth
w = synth.fov_dsm_weights(5, in_channels=5)
x = torch.from_numpy(synth.normalized_images(5, 5, (128, 5, 128, 512))).cuda()
enc = cvig_semantic.FOV_DSM(circ_padding=True, weights=w).cuda().eval()
for _ in range(3):
    enc.forward_bf16(x)
torch.cuda.synchronize()
