import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from witw_amd import synth, cvig_baseline as cb
from oracle import cvig_baseline_oracle as OB
dev = torch.device('cuda:0')
seed = 4242
B = 4
xo = torch.from_numpy(synth.images_u8(seed, 2, (B, 3, 512, 512))).to(dev)
prm_o = synth.baseline_params(seed + 1)
for mode in ('to_then_copy', 'copy_then_to'):
    oe = cb.OverheadEncoder()
    if mode == 'to_then_copy':
        oe = oe.to(dev).eval()
    with torch.no_grad():
        for i, q in enumerate(prm_o, 1):
            getattr(oe, 'conv%d' % i).weight.copy_(torch.from_numpy(q['w']))
            getattr(oe, 'conv%d' % i).bias.copy_(torch.from_numpy(q['b']))
            bn = getattr(oe, 'bn%d' % i)
            bn.weight.copy_(torch.from_numpy(q['gamma'])); bn.bias.copy_(torch.from_numpy(q['beta']))
            bn.running_mean.copy_(torch.from_numpy(q['mean'])); bn.running_var.copy_(torch.from_numpy(q['var']))
    if mode != 'to_then_copy':
        oe = oe.to(dev).eval()
    with torch.no_grad():
        eo = oe(xo)
        ref = OB.encoder_forward(xo.cpu(), [{k: torch.from_numpy(v) for k, v in q.items()} for q in prm_o])
    print(mode, float((eo.cpu() - ref).abs().max()), float(ref.abs().max()))
xs = torch.from_numpy(synth.images_u8(seed, 1, (B, 3, 500, 500))).to(dev)
xs2 = (torch.nn.functional.interpolate(xo, size=(500, 500), mode='bilinear', align_corners=False) * 0.7 + xs * 0.3).round().contiguous()
prm_s = synth.baseline_params(seed)
se = cb.SurfaceEncoder()
with torch.no_grad():
    for i, q in enumerate(prm_s, 1):
        getattr(se, 'conv%d' % i).weight.copy_(torch.from_numpy(q['w'])); getattr(se, 'conv%d' % i).bias.copy_(torch.from_numpy(q['b']))
        bn = getattr(se, 'bn%d' % i)
        bn.weight.copy_(torch.from_numpy(q['gamma'])); bn.bias.copy_(torch.from_numpy(q['beta']))
        bn.running_mean.copy_(torch.from_numpy(q['mean'])); bn.running_var.copy_(torch.from_numpy(q['var']))
se = se.to(dev).eval()
for name, x in (('noise', xs), ('planted', xs2)):
    with torch.no_grad():
        es = se(x)
        ref = OB.encoder_forward(x.cpu(), [{k: torch.from_numpy(v) for k, v in q.items()} for q in prm_s])
    print(name, float((es.cpu() - ref).abs().max()), float(ref.abs().max()), x.is_contiguous(), x.stride())
