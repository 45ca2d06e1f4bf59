"""CPU oracle for the cvig_baseline hot path — TEST INFRASTRUCTURE ONLY (see cvig_fov_oracle.py header).
Restates model/cvig_baseline.py's encoder (eval mode), loss and ranking in plain CPU torch ops; pinned to the
reference through tests/golden/baseline.npz (tests/test_oracle_golden.py)."""
import numpy as np
import torch
import torch.nn.functional as F


def encoder_forward(x, params, p=3., train=False):
    """SurfaceEncoder.forward, model/cvig_baseline.py:264-279. eval-mode BatchNorm by default; train=True uses batch
    statistics and updates params[i]['mean'/'var'] in place (momentum 0.1), as nn.BatchNorm2d does under .train().
    params: list of 7 dicts {w,b,gamma,beta,mean,var}."""
    x = x / 255.
    x = -1. + 2. * x
    feats = []
    for i, q in enumerate(params):
        x = F.conv2d(x, q['w'], q['b'], stride=2, padding=0)
        x = F.leaky_relu(x, 0.2)
        x = F.batch_norm(x, q['mean'], q['var'], q['gamma'], q['beta'], training=train, momentum=0.1, eps=1e-5)
        if i >= 4:
            feats.append(torch.pow(torch.mean(torch.pow(F.relu(x), p), [2, 3]), 1. / p))
    f = torch.cat(feats, 1)
    return f / torch.unsqueeze(torch.pow(torch.linalg.norm(f, dim=1), 0.5), 1)


def exhaustive_minibatch_triplet_loss(embed1, embed2, soft_margin=False, alpha=10., margin=1.):
    """model/cvig_baseline.py:286-315."""
    loss = torch.tensor(0.)
    bsz = embed1.size(0)
    for (a, p) in [(embed1, embed2), (embed2, embed1)]:
        for shift in range(1, bsz):
            n = torch.roll(p, shift, dims=0)
            ap = torch.sum((p - a) ** 2, dim=1)
            an = torch.sum((n - a) ** 2, dim=1)
            this = torch.log(1. + torch.exp(alpha * (ap - an))) if soft_margin else F.relu(ap - an + margin)
            loss = loss + torch.sum(this)
    return loss / (2 * bsz * (bsz - 1))


def ranks(overhead_embed, surface_embed):
    """model/cvig_baseline.py:454-460."""
    count = surface_embed.size(0)
    out = np.zeros([count], dtype=np.int64)
    for idx in range(count):
        q = torch.unsqueeze(surface_embed[idx, :], 0)
        d = torch.pow(torch.sum(torch.pow(overhead_embed - q, 2), dim=1), 0.5)
        out[idx] = torch.sum(torch.le(d, d[idx])).item()
    return out


def rotate(img, angle):
    """torchvision==0.9.1 transforms.functional.rotate(img, angle) for a float tensor [..,C,H,W] with the defaults
    the reference uses (model/cvig_baseline.py:142: nearest, expand=False, centre, no fill) — restated from
    torchvision 0.9.1's published functional.py / functional_tensor.py (_get_inverse_affine_matrix,
    _gen_affine_grid, _apply_grid_transform); torchvision is not installed here, so this restatement is
    PARITY-UNPINNED (no reference test holds vectors for it)."""
    import math
    squeeze = img.dim() == 3
    x = img.unsqueeze(0) if squeeze else img
    h, w = x.shape[-2], x.shape[-1]
    rot = math.radians(-angle)
    matrix = [math.cos(rot), math.sin(rot), 0.0, -math.sin(rot), math.cos(rot), 0.0]
    theta = torch.tensor(matrix, dtype=torch.float32).reshape(1, 2, 3)
    base = torch.empty(1, h, w, 3, dtype=torch.float32)
    base[..., 0].copy_(torch.linspace(-w * 0.5 + 0.5, w * 0.5 + 0.5 - 1, steps=w))
    base[..., 1].copy_(torch.linspace(-h * 0.5 + 0.5, h * 0.5 + 0.5 - 1, steps=h).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=torch.float32)
    grid = base.view(1, h * w, 3).bmm(rescaled).view(1, h, w, 2).expand(x.shape[0], h, w, 2)
    y = F.grid_sample(x, grid, mode='nearest', padding_mode='zeros', align_corners=False)
    return y.squeeze(0) if squeeze else y
