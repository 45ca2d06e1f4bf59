import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import cvig_fov_oracle as O
from witw_amd import ops
def rand(seed, shape, scale=1.0):
    g = np.random.Generator(np.random.Philox(key=[seed, 7]))
    return (g.standard_normal(shape, dtype=np.float32) * np.float32(scale)).astype(np.float32)
B, H, W, Cin, Cout, sh, circ, relu, pool = [int(v) for v in sys.argv[1:10]]
x = rand(1, (B, Cin, H, W)); w = rand(2, (Cout, Cin, 3, 3), (2.0/(9*Cin))**0.5); b = rand(3, (Cout,), 0.1)
ref = O.conv3x3(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), sh, bool(circ))
if relu: ref = torch.relu(ref)
if pool: ref = torch.nn.functional.max_pool2d(ref, 2, 2)
dev = torch.device('cuda:0')
xd = torch.from_numpy(x).to(dev).permute(0, 2, 3, 1).contiguous()
pk = ops.PackedConv(torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev))
y = ops.conv3x3_fwd(xd, pk, stride_h=sh, circular=bool(circ), relu=bool(relu), pool=bool(pool)).cpu().permute(0, 3, 1, 2)
err = (y - ref).abs()
print('max err', err.max().item())
bad = (err > 1e-3)
print('bad frac', bad.float().mean().item())
print('bad by row   ', bad.float().mean(dim=(0, 1, 3)).numpy().round(2))
print('bad by col/8 ', bad.float().mean(dim=(0, 1, 2)).reshape(-1, 8).mean(1).numpy().round(2))
print('bad by chan/8', bad.float().mean(dim=(0, 2, 3)).reshape(-1, 8).mean(1).numpy().round(2))
# tap decomposition: which taps are present in the GPU result? (least squares over the 9 single-tap convs, Cin small)
xs = torch.from_numpy(x); ws = torch.from_numpy(w)
parts = []
for kh in range(3):
    for kw in range(3):
        wt = torch.zeros_like(ws); wt[:, :, kh, kw] = ws[:, :, kh, kw]
        parts.append(O.conv3x3(xs, wt, None, sh, bool(circ)))
P = torch.stack([pp.reshape(-1) for pp in parts], 1).double()
if not relu and not pool:
    t = (y - torch.from_numpy(b)[None, :, None, None]).reshape(-1).double()
    coef = torch.linalg.lstsq(P, t[:, None]).solution.squeeze()
    print('tap coefficients (1 = present):', coef.numpy().round(3))
    # per-row coefficients
    for row in range(min(H, 8)):
        sel = torch.zeros_like(y, dtype=torch.bool); sel[:, :, row, :] = True
        c2 = torch.linalg.lstsq(P[sel.reshape(-1)], t[sel.reshape(-1)][:, None]).solution.squeeze()
        print(' row', row, c2.numpy().round(2))
