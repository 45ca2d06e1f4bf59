"""In-kernel phase stamps of conv3x3_bf16_wres_kernel (run on the GPU box from the repo root with WITW_WRES_STAMPS=1)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from witw_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((128, 64, 256, 64), generator=g, device=dev).bfloat16()
w = torch.randn((128, 64, 3, 3), generator=g, device=dev) * 0.06
b = torch.randn((128,), generator=g, device=dev) * 0.1
pk = ops.PackedConvBf16(w, b)
for _ in range(3):
    y = ops.conv3x3_bf16_fwd(x, pk, circular=True, relu=True)
torch.cuda.synchronize()
print(ops.last_kernel_variant())
