#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (read-only at /root/reference).

Runs only in the build container: the GPU box has no /root/reference and uses the committed
.npz files. The reference's missing third-party imports (torchvision, skimage, tifffile,
tensorboard) are replaced by empty stub modules; torch.hub.load is replaced by a function that
returns a torchvision-VGG16-shaped nn.Module filled with the seeded synthetic weights of
witw_amd/synth.py, so that the reference's own FOV_DSM.__init__ does the slicing, extra convs,
freezing, circular-padding and dropout wrapping. No reference source is copied: only inputs
(regenerable from seeds) and outputs are stored.

    python tests/golden/gen_golden.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from witw_amd import synth  # noqa: E402

REF = '/root/reference/model'
SEED = 20261003


def import_reference():
    sys.dont_write_bytecode = True
    for n in ["torchvision", "torchvision.transforms", "torchvision.transforms.functional",
              "skimage", "skimage.io", "tifffile", "torch.utils.tensorboard"]:
        sys.modules[n] = types.ModuleType(n)
    tv = sys.modules["torchvision"]
    tv.transforms = sys.modules["torchvision.transforms"]
    tv.transforms.functional = sys.modules["torchvision.transforms.functional"]

    class Normalize:
        def __init__(self, mean, std):
            self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
            self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

        def __call__(self, x):
            return (x - self.mean) / self.std

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    def resize(img, size):  # torchvision 0.9.1 tensor semantics; UNPINNED (torchvision absent)
        return F.interpolate(img.unsqueeze(0), size=list(size), mode='bilinear', align_corners=False).squeeze(0)

    tv.transforms.Normalize = Normalize
    tv.transforms.Compose = Compose
    tv.transforms.functional.resize = resize
    sys.modules["skimage"].io = sys.modules["skimage.io"]
    sys.modules["torch.utils.tensorboard"].SummaryWriter = object
    sys.path.insert(0, REF)
    import cvig_fov
    import cvig_semantic
    import cvig_baseline
    return cvig_fov, cvig_semantic, cvig_baseline


class FakeVGG(nn.Module):
    """torchvision vgg16 attribute layout (cfg D features, avgpool, classifier)."""

    def __init__(self, weights):
        super().__init__()
        cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
        layers, cin = [], 3
        for v in cfg:
            if v == 'M':
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = nn.Sequential(nn.Linear(8, 8), nn.ReLU(True), nn.Dropout(), nn.Linear(8, 8), nn.ReLU(True),
                                        nn.Dropout(), nn.Linear(8, 8))  # unused by the hot path; kept tiny
        with torch.no_grad():
            for i, m in enumerate(self.features):
                if isinstance(m, nn.Conv2d) and i in weights:
                    w, b = weights[i]
                    if w.shape[1] == m.weight.shape[1]:
                        m.weight.copy_(torch.from_numpy(w))
                        m.bias.copy_(torch.from_numpy(b))


def conv_of(layer):
    while not isinstance(layer, nn.Conv2d):
        layer = layer.layer
    return layer


def build_ref_encoder(mod, circ, weights):
    torch.hub.load = lambda *a, **k: FakeVGG(weights)
    enc = mod.FOV_DSM(circ_padding=circ)
    with torch.no_grad():
        for i in (0, 23, 25, 27):
            w, b = weights[i]
            c = conv_of(enc.model.features[i])
            assert tuple(c.weight.shape) == w.shape, (i, c.weight.shape, w.shape)
            c.weight.copy_(torch.from_numpy(w))
            c.bias.copy_(torch.from_numpy(b))
    return enc


def main():
    torch.set_num_threads(8)
    torch.manual_seed(0)
    fov, sem, base = import_reference()
    fov.device = torch.device('cpu')
    out = {}

    # ---- A3 polar transform (model/cvig_fov.py:156-209)
    ov_img = torch.from_numpy(synth.normalized_images(SEED, 1, (3, 256, 256)))
    polar = fov.PolarTransform()({'overhead': ov_img})['polar']
    assert polar.shape == (3, 128, 512) and polar.dtype == torch.float32
    np.savez(os.path.join(HERE, 'polar.npz'), seed=SEED, stream=1,
             rows=np.arange(0, 128, 8), polar_rows=polar[:, ::8, :].numpy(),
             polar_sum=np.float64(polar.double().sum().item()),
             polar_abs_sum=np.float64(polar.double().abs().sum().item()),
             zero_taps=np.array([[0, 0], [0, 384]]),   # (y,x) with all four weights 0
             zero_vals=np.stack([polar[:, 0, 0].numpy(), polar[:, 0, 384].numpy()]))

    # ---- A2 normalisation (model/cvig_fov.py:137-149)
    raw = torch.from_numpy(synth.images_u8(SEED, 2, (3, 16, 24)))
    d = fov.ImageNormalization()({'surface': raw.clone(), 'overhead': raw.clone()})
    raw5 = torch.cat((raw, torch.from_numpy(synth.normalized_images(SEED, 3, (2, 16, 24))).abs() / 3), 0)
    d5 = sem.ImageNormalization()({'surface': raw5.clone(), 'overhead': raw5.clone()})
    np.savez(os.path.join(HERE, 'normalize.npz'), seed=SEED, raw5=raw5.numpy(), norm3=d['surface'].numpy(),
             norm5=d5['surface'].numpy())

    # ---- A4-A6 encoders, eval mode
    weights = synth.fov_dsm_weights(SEED)
    x360 = torch.from_numpy(synth.normalized_images(SEED, 10, (2, 3, 128, 512)))
    x70 = torch.from_numpy(synth.normalized_images(SEED, 11, (2, 3, 128, 99)))
    enc = {}
    res = {'seed': SEED}
    for circ in (False, True):
        e = build_ref_encoder(fov, circ, weights).eval()
        enc[circ] = e
        with torch.no_grad():
            res['embed360_circ%d' % circ] = e(x360).numpy()
            res['embed70_circ%d' % circ] = e(x70).numpy()
    # state-dict key layout (SURVEY §5 checkpoint row)
    res['keys_surface'] = np.array(sorted(enc[False].state_dict().keys()))
    res['keys_overhead'] = np.array(sorted(enc[True].state_dict().keys()))
    res['trainable_surface'] = np.array(sorted(n for n, p in enc[False].named_parameters() if p.requires_grad))

    # train mode with captured Dropout2d masks (model/cvig_fov.py:234-245,287-288)
    e = enc[True].train()
    captured = {}

    def mk_hook(i):
        def hook(m, inp, outp):
            xi, xo = inp[0].detach(), outp.detach()
            b, c = xi.shape[:2]
            flat_i = xi.reshape(b, c, -1)
            pos = flat_i.abs().argmax(-1, keepdim=True)
            captured[i] = (xo.reshape(b, c, -1).gather(2, pos) / flat_i.gather(2, pos)).squeeze(-1).numpy()
        return hook
    hs = [e.model.features[i].postlayer.register_forward_hook(mk_hook(i)) for i in (17, 19, 21)]
    torch.manual_seed(1234)
    with torch.no_grad():
        res['embed360_circ1_train'] = e(x360).numpy()
    for h in hs:
        h.remove()
    for i in (17, 19, 21):
        s = captured[i]
        assert np.all((np.abs(s) < 1e-6) | (np.abs(s - 1.25) < 1e-4)), s
        res['drop_scale_%d' % i] = np.where(np.abs(s) < 1e-6, 0.0, 1.25).astype(np.float32)
    e.eval()
    np.savez(os.path.join(HERE, 'encoder.npz'), **res)

    # semantic variant, 5-channel (model/cvig_semantic.py:275-325)
    w5 = synth.fov_dsm_weights(SEED, in_channels=5)
    x5 = torch.from_numpy(synth.normalized_images(SEED, 12, (1, 5, 128, 512)))
    torch.hub.load = lambda *a, **k: FakeVGG(weights)
    e5 = sem.FOV_DSM(circ_padding=True)
    with torch.no_grad():
        for i in (0, 23, 25, 27):
            w, b = w5[i]
            c = conv_of(e5.model.features[i])
            c.weight.copy_(torch.from_numpy(w))
            c.bias.copy_(torch.from_numpy(b))
        e5.eval()
        emb5 = e5(x5).numpy()
    np.savez(os.path.join(HERE, 'encoder_semantic.npz'), seed=SEED, embed5_circ1=emb5,
             trainable=np.array(sorted(n for n, p in e5.named_parameters() if p.requires_grad)))

    # ---- A7-A10 matching + loss (+ autograd gradients)
    res = {'seed': SEED}
    cases = [('a', 4, 4, 64), ('b', 37, 29, 12), ('c', 5, 3, 33), ('d', 6, 6, 12), ('e', 9, 9, 64)]
    for (tag, bo, bs, we) in cases:
        ov = torch.from_numpy(synth.embeddings(SEED, 100 + ord(tag), (bo, 16, 4, 64))).requires_grad_(True)
        su = torch.from_numpy(synth.embeddings(SEED, 200 + ord(tag), (bs, 16, 4, we))).requires_grad_(True)
        ori = fov.correlation(ov, su)
        crop = fov.crop_overhead(ov, ori, we)
        dist = fov.l2_distance(crop, su)
        res['%s_shape' % tag] = np.array([bo, bs, we])
        res['%s_orientation' % tag] = ori.numpy()
        res['%s_distance' % tag] = dist.detach().numpy()
        # top-2 correlation gap: goldens must not sit on an argmax near-tie
        n = we - 1
        sc = F.conv2d(torch.cat((ov, ov[:, :, :, :n]), 3), su).squeeze(-2).detach()
        top2 = sc.topk(2, -1).values
        gap = (top2[..., 0] - top2[..., 1]).min().item()
        assert gap > 1e-3, (tag, gap)
        res['%s_min_gap' % tag] = np.float32(gap)
        if bo == bs:
            loss = fov.triplet_loss(dist)
            loss.backward()
            res['%s_loss' % tag] = loss.detach().numpy()
            res['%s_grad_ov' % tag] = ov.grad.numpy()
            res['%s_grad_su' % tag] = su.grad.numpy()
        if tag == 'c':
            res['c_crop'] = crop.detach().numpy()
    # triplet loss on a plain matrix, alpha variants (model/cvig_fov.py:366-382)
    dm = torch.from_numpy(synth.embeddings(SEED, 300, (7, 7))).abs()
    res['loss_in'] = dm.numpy()
    res['loss_a10'] = fov.triplet_loss(dm).numpy()
    res['loss_a3'] = fov.triplet_loss(dm, alpha=3.).numpy()
    np.savez(os.path.join(HERE, 'matching.npz'), **res)

    # ---- A12 ranking loop body (model/cvig_fov.py:543-558), planted matches
    res = {'seed': SEED}
    for (tag, n, we, nf) in [('r360', 24, 64, 20.0), ('r70', 24, 12, 8.0)]:
        ov = torch.from_numpy(synth.embeddings(SEED, 400 + we, (n, 16, 4, 64)))
        noise = torch.from_numpy(synth.embeddings(SEED, 500 + we, (n, 16, 4, we)))
        shifts = (np.arange(n) * 7) % 64
        su = torch.stack([torch.roll(ov[i], -int(shifts[i]), dims=2)[:, :, :we] for i in range(n)]) + nf * noise
        count = n
        ranks = np.zeros([count], dtype=int)
        dmat = np.zeros((n, n), dtype=np.float32)
        omat = np.zeros((n, n), dtype=np.int64)
        for idx in range(count):
            q = torch.unsqueeze(su[idx, :], 0)
            ori = fov.correlation(ov, q)
            crop = fov.crop_overhead(ov, ori, q.shape[3])
            distances = torch.squeeze(fov.l2_distance(crop, q))
            ranks[idx] = torch.sum(torch.le(distances, distances[idx])).item()
            dmat[:, idx] = distances.numpy()
            omat[:, idx] = ori.squeeze(1).numpy()
        res['%s_n_we' % tag] = np.array([n, we])
        res['%s_noise' % tag] = np.float32(nf)
        assert 1 < ranks.max() and (ranks == 1).sum() >= 3, ranks
        res['%s_shifts' % tag] = shifts
        res['%s_ranks' % tag] = ranks
        res['%s_dist' % tag] = dmat
        res['%s_ori' % tag] = omat
        res['%s_table' % tag] = np.array([np.sum(ranks <= 1) / count * 100, np.sum(ranks <= 5) / count * 100,
                                          np.sum(ranks <= 10) / count * 100, np.sum(ranks * 100 <= count) / count * 100,
                                          np.mean(ranks), np.median(ranks)])
    np.savez(os.path.join(HERE, 'ranking.npz'), **res)
    print('goldens written to', HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print('  %-28s %8d bytes' % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == '__main__' and not any(a in sys.argv for a in ('--trainstep', '--baseline', '--baseline-train')):
    main()


def gen_trainstep():
    """One iteration of the reference's training loop body (model/cvig_fov.py:444-461) on a tiny batch,
    Dropout2d masks captured, then one torch.optim.Adam step (lr as :418). Stores the loss, per-parameter
    gradient norms and strided samples of gradients and of updated parameters."""
    fov, sem, base = import_reference()
    fov.device = torch.device('cpu')
    weights = synth.fov_dsm_weights(SEED + 1)
    B, ws = 3, 96      # fov 67.5 deg -> int(67.5/360*512) = 96 -> embedding width 12
    xs = torch.from_numpy(synth.normalized_images(SEED, 20, (B, 3, 128, ws)))
    xo = torch.from_numpy(synth.normalized_images(SEED, 21, (B, 3, 128, 512)))
    se = build_ref_encoder(fov, False, weights).train()
    oe = build_ref_encoder(fov, True, weights).train()
    captured = {}

    def mk_hook(tag, i):
        def hook(m, inp, outp):
            xi, xo_ = inp[0].detach(), outp.detach()
            b, c = xi.shape[:2]
            fi = xi.reshape(b, c, -1)
            pos = fi.abs().argmax(-1, keepdim=True)
            sc = (xo_.reshape(b, c, -1).gather(2, pos) / fi.gather(2, pos)).squeeze(-1).numpy()
            captured[(tag, i)] = np.where(np.abs(sc) < 1e-6, 0.0, 1.25).astype(np.float32)
        return hook
    for tag, e in (('s', se), ('o', oe)):
        for i in (17, 19, 21):
            e.model.features[i].postlayer.register_forward_hook(mk_hook(tag, i))
    params = [p for p in list(se.parameters()) + list(oe.parameters())]
    opt = torch.optim.Adam(params, lr=1.E-5)
    torch.manual_seed(77)
    s_emb = se(xs)
    o_emb = oe(xo)
    ori = fov.correlation(o_emb, s_emb)
    crop = fov.crop_overhead(o_emb, ori, s_emb.shape[3])
    dist = fov.l2_distance(crop, s_emb)
    loss = fov.triplet_loss(dist)
    opt.zero_grad()
    loss.backward()
    res = {'seed': SEED, 'B': B, 'ws': ws, 'loss': loss.detach().numpy(), 'orientation': ori.numpy(),
           'distance': dist.detach().numpy()}
    for (tag, i), v in captured.items():
        res['drop_%s_%d' % (tag, i)] = v
    named = [('s.' + n, p) for n, p in se.named_parameters()] + [('o.' + n, p) for n, p in oe.named_parameters()]
    names = []
    for n, p in named:
        if p.grad is None or 'classifier' in n:
            continue
        names.append(n)
        g = p.grad.detach().reshape(-1)
        res['gnorm:' + n] = np.float64(g.double().norm().item())
        res['gsamp:' + n] = g[::max(1, g.numel() // 257)].numpy()
    opt.step()
    for n, p in named:
        if n in names:
            res['psamp:' + n] = p.detach().reshape(-1)[::max(1, p.numel() // 257)].numpy()
    res['names'] = np.array(names)
    np.savez(os.path.join(HERE, 'trainstep.npz'), **res)
    print('trainstep.npz: loss %.6f, %d trainable tensors' % (loss.item(), len(names)))


if __name__ == '__main__' and '--trainstep' in sys.argv:
    gen_trainstep()


def gen_baseline():
    """cvig_baseline goldens: encoder eval forward (500x500 surface / 512x512 overhead, SURVEY C1 shapes),
    exhaustive minibatch triplet loss (hard and soft margin) and the Euclidean ranking loop body."""
    fov, sem, base = import_reference()
    res = {'seed': SEED}
    for tag, cls, hw, stream in (('surface', base.SurfaceEncoder, 500, 30), ('overhead', base.OverheadEncoder, 512, 31)):
        enc = cls()
        prm = synth.baseline_params(SEED + (0 if tag == 'surface' else 1))
        with torch.no_grad():
            for i, q in enumerate(prm, 1):
                getattr(enc, 'conv%d' % i).weight.copy_(torch.from_numpy(q['w']))
                getattr(enc, 'conv%d' % i).bias.copy_(torch.from_numpy(q['b']))
                bn = getattr(enc, 'bn%d' % i)
                bn.weight.copy_(torch.from_numpy(q['gamma']))
                bn.bias.copy_(torch.from_numpy(q['beta']))
                bn.running_mean.copy_(torch.from_numpy(q['mean']))
                bn.running_var.copy_(torch.from_numpy(q['var']))
        enc.eval()
        x = torch.from_numpy(synth.images_u8(SEED, stream, (2, 3, hw, hw)))
        with torch.no_grad():
            res['embed_' + tag] = enc(x).numpy()
        res['keys_' + tag] = np.array(sorted(enc.state_dict().keys()))
    e1 = torch.from_numpy(synth.embeddings(SEED, 600, (5, 1536))) * 0.018
    e2 = e1 + torch.from_numpy(synth.embeddings(SEED, 601, (5, 1536))) * 0.02
    res['loss_hard'] = base.exhaustive_minibatch_triplet_loss(e1, e2).numpy()
    res['loss_soft'] = base.exhaustive_minibatch_triplet_loss(e1, e2, soft_margin=True).numpy()
    res['loss_hard_m03'] = base.exhaustive_minibatch_triplet_loss(e1 * 0.55, e2 * 0.55, margin=0.3).numpy()
    res['loss_soft_a2'] = base.exhaustive_minibatch_triplet_loss(e1, e2, soft_margin=True, alpha=2.).numpy()
    n = 14
    ov = torch.from_numpy(synth.embeddings(SEED, 602, (n, 1536)))
    su = ov + 14.0 * torch.from_numpy(synth.embeddings(SEED, 603, (n, 1536)))
    ranks = np.zeros([n], dtype=int)
    for idx in range(n):
        q = torch.unsqueeze(su[idx, :], 0)
        d = torch.pow(torch.sum(torch.pow(ov - q, 2), dim=1), 0.5)
        ranks[idx] = torch.sum(torch.le(d, d[idx])).item()
    res['ranks'] = ranks
    assert ranks.max() > 1 and float(res['loss_hard']) > 0 and float(res['loss_hard_m03']) > 0
    np.savez(os.path.join(HERE, 'baseline.npz'), **res)
    print('baseline.npz written; ranks', ranks, 'losses', res['loss_hard'], res['loss_soft'])


if __name__ == '__main__' and '--baseline' in sys.argv:
    gen_baseline()


def gen_baseline_train():
    """One iteration of the reference's cvig_baseline training loop body (model/cvig_baseline.py:373-387): train-mode
    encoders (BatchNorm batch statistics), exhaustive_minibatch_triplet_loss, backward, Adam(lr=1e-3) step."""
    fov, sem, base = import_reference()
    B = 3
    encs = {}
    for tag, cls, hw, stream, off in (('surface', base.SurfaceEncoder, 400, 40, 0), ('overhead', base.OverheadEncoder, 416, 41, 1)):
        enc = cls()
        prm = synth.baseline_params(SEED + 10 + off)
        with torch.no_grad():
            for i, q in enumerate(prm, 1):
                getattr(enc, 'conv%d' % i).weight.copy_(torch.from_numpy(q['w']))
                getattr(enc, 'conv%d' % i).bias.copy_(torch.from_numpy(q['b']))
                bn = getattr(enc, 'bn%d' % i)
                bn.weight.copy_(torch.from_numpy(q['gamma']))
                bn.bias.copy_(torch.from_numpy(q['beta']))
                bn.running_mean.copy_(torch.from_numpy(q['mean']))
                bn.running_var.copy_(torch.from_numpy(q['var']))
        encs[tag] = (enc.train(), torch.from_numpy(synth.images_u8(SEED, stream, (B, 3, hw, hw))))
    params = list(encs['surface'][0].parameters()) + list(encs['overhead'][0].parameters())
    opt = torch.optim.Adam(params)                                   # default lr 1e-3, :349
    s_emb = encs['surface'][0](encs['surface'][1])
    o_emb = encs['overhead'][0](encs['overhead'][1])
    loss = base.exhaustive_minibatch_triplet_loss(s_emb, o_emb)
    opt.zero_grad()
    loss.backward()
    res = {'seed': SEED, 'B': B, 'loss': loss.detach().numpy(), 'embed_surface': s_emb.detach().numpy(),
           'embed_overhead': o_emb.detach().numpy()}
    names = []
    for tag in ('surface', 'overhead'):
        for n, p in encs[tag][0].named_parameters():
            key = '%s.%s' % (tag, n)
            names.append(key)
            gflat = p.grad.reshape(-1)
            res['gnorm:' + key] = np.float64(gflat.double().norm().item())
            res['gsamp:' + key] = gflat[::max(1, gflat.numel() // 129)].numpy()
    opt.step()
    for tag in ('surface', 'overhead'):
        for n, p in encs[tag][0].named_parameters():
            res['psamp:%s.%s' % (tag, n)] = p.detach().reshape(-1)[::max(1, p.numel() // 129)].numpy()
        for n, bbuf in encs[tag][0].named_buffers():
            if 'num_batches' not in n:
                res['buf:%s.%s' % (tag, n)] = bbuf.detach().numpy()
    res['names'] = np.array(names)
    assert float(loss) > 0
    np.savez(os.path.join(HERE, 'baseline_train.npz'), **res)
    print('baseline_train.npz written; loss', float(loss))


if __name__ == '__main__' and '--baseline-train' in sys.argv:
    gen_baseline_train()
