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


if __name__ == '__main__' and not any(a in sys.argv for a in ('--trainstep', '--trainstep-semantic', '--baseline', '--baseline-train', '--augment', '--checkpoint-interop')):
    main()


TAU = 1e-4      # |pre-activation| (or max-pool top-2 gap), relative to max(1, std of the layer), below which a ReLU gate (pool route) is listed as fragile


def gate_records(res, tag, z_by_layer, pooled, dropped):
    """For every ReLU of the reference whose gate shapes a gradient: per-(sample,channel) counts of open gates, and the
    FRAGILE positions -- |pre-activation| < TAU -- with the side the reference took. For a ReLU followed by MaxPool2d (the
    build fuses both into the conv epilogue and gates on the pooled value): the same on the window maximum, plus the arg-max
    position (dy*2+dx) summed per (sample,channel) over the open windows and listed where the top-2 gap is < TAU."""
    for idx, z in z_by_layer.items():
        tau = TAU * max(1.0, float(z.std()))          # absolute 1e-4 at unit scale; the early VGG layers run at larger magnitudes
        res['tau:%s:%d' % (tag, idx)] = np.float32(tau)
        if idx in pooled:
            b, c, h, w = z.shape
            win = z[:, :, :h // 2 * 2, :w // 2 * 2].reshape(b, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(b, c, h // 2, w // 2, 4)
            zmax, code = win.max(-1)            # first maximum wins, as MaxPool2d does
            top2 = win.topk(2, -1).values
            gap = top2[..., 0] - top2[..., 1]
            gate = zmax > 0
            res['pcsum:%s:%d' % (tag, idx)] = (code * gate).sum(dim=(2, 3)).numpy().astype(np.int64)
            pf = torch.nonzero(((gap < tau) & gate).reshape(-1)).squeeze(1)
            res['pfrag:%s:%d' % (tag, idx)] = pf.numpy().astype(np.int64)
            res['pfragv:%s:%d' % (tag, idx)] = code.reshape(-1)[pf].numpy().astype(np.uint8)
            z = zmax
        gate = z > 0
        res['gcount:%s:%d' % (tag, idx)] = gate.sum(dim=(2, 3)).numpy().astype(np.int32)
        live = torch.ones_like(z, dtype=torch.bool)
        if idx in dropped:          # channels Dropout2d zeroed are exact zeros with no gradient: not fragile
            live = (torch.from_numpy(dropped[idx]) != 0)[:, :, None, None].expand_as(z)
        fr = torch.nonzero(((z.abs() < tau) & live).reshape(-1)).squeeze(1)
        res['gfrag:%s:%d' % (tag, idx)] = fr.numpy().astype(np.int64)
        res['gfragv:%s:%d' % (tag, idx)] = gate.reshape(-1)[fr].numpy().astype(np.uint8)


def run_reference_step(mod, se, oe, xs, xo, lr, relu_layers, pooled, nsamp):
    """One iteration of the reference's training loop body (model/cvig_fov.py:444-461; cvig_semantic.py:475-492) with the
    Dropout2d masks and the ReLU pre-activations captured, then one torch.optim.Adam step."""
    captured, zs = {}, {'s': {}, 'o': {}}

    def mk_hook(tag, i):
        def hook(m, inp, outp):
            xi, xo_ = inp[0].detach(), outp.detach()
            b, c = xi.shape[:2]
            fi = xi.reshape(b, c, -1)
            pos = fi.abs().argmax(-1, keepdim=True)
            sc = (xo_.reshape(b, c, -1).gather(2, pos) / fi.gather(2, pos)).squeeze(-1).numpy()
            captured[(tag, i)] = np.where(np.abs(sc) < 1e-6, 0.0, 1.25).astype(np.float32)
        return hook

    def mk_pre(tag, i):
        def pre(m, inp):
            zs[tag][i] = inp[0].detach().clone()        # the ReLU is in-place: copy its input first
        return pre
    for tag, e in (('s', se), ('o', oe)):
        for i in (17, 19, 21):
            e.model.features[i].postlayer.register_forward_hook(mk_hook(tag, i))
        for i in relu_layers:
            assert isinstance(e.model.features[i + 1], nn.ReLU)
            e.model.features[i + 1].register_forward_pre_hook(mk_pre(tag, i))
    params = [p for p in list(se.parameters()) + list(oe.parameters())]
    opt = torch.optim.Adam(params, lr=lr)
    torch.manual_seed(77)
    s_emb = se(xs)
    o_emb = oe(xo)
    ori = mod.correlation(o_emb, s_emb)
    crop = mod.crop_overhead(o_emb, ori, s_emb.shape[3])
    dist = mod.l2_distance(crop, s_emb)
    loss = mod.triplet_loss(dist)
    opt.zero_grad()
    loss.backward()
    res = {'loss': loss.detach().numpy(), 'orientation': ori.numpy(), 'distance': dist.detach().numpy(),
           'embed_s': s_emb.detach().numpy(), 'embed_o': o_emb.detach().numpy(), 'tau': np.float32(TAU)}
    sc = F.conv2d(torch.cat((o_emb, o_emb[:, :, :, :s_emb.shape[3] - 1]), 3), s_emb).squeeze(-2).detach()
    top2 = sc.topk(2, -1).values
    res['min_gap'] = np.float32((top2[..., 0] - top2[..., 1]).min().item())
    bound = o_emb.detach().flatten(1).norm(dim=1)[:, None] * s_emb.detach().flatten(1).norm(dim=1)[None, :]
    res['min_gap_rel'] = np.float32(((top2[..., 0] - top2[..., 1]) / bound).min().item())
    assert res['min_gap_rel'] > 2e-5, res['min_gap_rel']        # orientations well clear of fp32 summation-order noise (~2e-6)
    for (tag, i), v in captured.items():
        res['drop_%s_%d' % (tag, i)] = v
    for tag in 'so':
        gate_records(res, tag, zs[tag], pooled, {i: captured[(tag, i)] for i in (17, 19, 21)})
    named = [('s.' + n, p) for n, p in se.named_parameters()] + [('o.' + n, p) for n, p in oe.named_parameters()]
    names = []
    before = {}
    for n, p in named:
        if p.grad is None or 'classifier' in n:
            continue
        names.append(n)
        g = p.grad.detach().reshape(-1)
        res['gnorm:' + n] = np.float64(g.double().norm().item())
        res['gsamp:' + n] = g[::max(1, g.numel() // nsamp)].numpy()
        before[n] = p.detach().reshape(-1)[::max(1, p.numel() // nsamp)].clone()
    opt.step()
    for n, p in named:
        if n in names:
            after = p.detach().reshape(-1)[::max(1, p.numel() // nsamp)]
            res['psamp:' + n] = after.numpy()
            res['dsamp:' + n] = (after - before[n]).numpy()      # what the Adam step moved (fp32 subtraction, as stored)
    res['names'] = np.array(names)
    return res, loss


def baseline_gate_records(res, tag, z_by_layer):
    """LeakyReLU(0.2) gates of cvig_baseline (model/cvig_baseline.py:267-273; torch: derivative 1 where the conv output is
    > 0, else 0.2): per-(sample,channel) counts of positive conv outputs and the fragile positions, as gate_records()."""
    for i, z in z_by_layer.items():
        tau = TAU * max(1.0, float(z.std()))
        res['tau:%s:%d' % (tag, i)] = np.float32(tau)
        gate = z > 0
        res['gcount:%s:%d' % (tag, i)] = gate.sum(dim=(2, 3)).numpy().astype(np.int32)
        fr = torch.nonzero((z.abs() < tau).reshape(-1)).squeeze(1)
        res['gfrag:%s:%d' % (tag, i)] = fr.numpy().astype(np.int64)
        res['gfragv:%s:%d' % (tag, i)] = gate.reshape(-1)[fr].numpy().astype(np.uint8)


def gen_trainstep(out_name='trainstep.npz', B=3, ws=96, wseed=SEED + 1, streams=(20, 21)):
    """The reference's cvig_fov training step on a tiny batch: loss, per-parameter gradient norms and strided samples of
    gradients, of updated parameters and of the update itself, ReLU gate records of layers 17-25 (gate_records).
    trainstep.npz: B=3, fov 67.5 (surface width 96 -> embedding width 12); trainstep360.npz: the config-2 geometry
    (fov 360: surface width 512 -> embedding width 64), B=4."""
    fov, sem, base = import_reference()
    fov.device = torch.device('cpu')
    weights = synth.fov_dsm_weights(wseed)
    xs = torch.from_numpy(synth.normalized_images(SEED, streams[0], (B, 3, 128, ws)))
    xo = torch.from_numpy(synth.normalized_images(SEED, streams[1], (B, 3, 128, 512)))
    se = build_ref_encoder(fov, False, weights).train()
    oe = build_ref_encoder(fov, True, weights).train()
    res, loss = run_reference_step(fov, se, oe, xs, xo, 1.E-5, (17, 19, 21, 23, 25), (), 257)
    res.update({'seed': SEED, 'B': B, 'ws': ws, 'wseed': wseed, 'streams': np.array(streams)})
    np.savez_compressed(os.path.join(HERE, out_name), **res)
    print('%s: loss %.6f, %d trainable tensors, min top-2 score gap %.3g, fragile gates %s' % (
        out_name, loss.item(), len(res['names']), float(res['min_gap']),
        {k: len(v) for k, v in res.items() if k.startswith('gfrag:')}))


def gen_semantic_trainstep():
    """The reference's cvig_semantic training step (model/cvig_semantic.py:475-492: 5-channel encoders, layer 0 trainable,
    so the backward crosses all 13 convs and the 3 max-pools). B=2, surface 5x128x64, overhead 5x128x512."""
    fov, sem, base = import_reference()
    sem.device = torch.device('cpu')
    seed = 321
    w5 = synth.fov_dsm_weights(seed, in_channels=5)
    w3 = synth.fov_dsm_weights(seed, in_channels=3)
    B = 2
    xs = torch.from_numpy(synth.normalized_images(seed, 1, (B, 5, 128, 64)))
    xo = torch.from_numpy(synth.normalized_images(seed, 2, (B, 5, 128, 512)))
    encs = []
    for circ in (False, True):
        torch.hub.load = lambda *a, **k: FakeVGG(w3)
        e = sem.FOV_DSM(circ_padding=circ)
        with torch.no_grad():
            for i, (w, b) in w5.items():            # synth's 5-channel set defines every layer (layer 0: all 5 input channels)
                c = conv_of(e.model.features[i])
                assert tuple(c.weight.shape) == w.shape, (i, c.weight.shape, w.shape)
                c.weight.copy_(torch.from_numpy(w))
                c.bias.copy_(torch.from_numpy(b))
        encs.append(e.train())
    relu_layers = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 23, 25)
    res, loss = run_reference_step(sem, encs[0], encs[1], xs, xo, 1.E-5, relu_layers, (2, 7, 14), 257)
    res.update({'seed': seed, 'B': B, 'ws': 64})
    del res['embed_s'], res['embed_o']
    np.savez_compressed(os.path.join(HERE, 'trainstep_semantic.npz'), **res)
    print('trainstep_semantic.npz: loss %.6f, %d trainable tensors, fragile gates %s, fragile routes %s' % (
        loss.item(), len(res['names']), {k: len(v) for k, v in res.items() if k.startswith('gfrag:')},
        {k: len(v) for k, v in res.items() if k.startswith('pfrag:')}))


if __name__ == '__main__' and '--trainstep' in sys.argv:
    gen_trainstep()
    # streams picked so that no orientation sits on a near-tie: smallest top-2 score gap 4e-5 |ov||su| (fp32 summation
    # orders differ by ~2e-6 |ov||su|; streams 22..27 give 1e-6..1e-5)
    gen_trainstep('trainstep360.npz', B=4, ws=512, wseed=SEED + 2, streams=(28, 29))

if __name__ == '__main__' and '--trainstep-semantic' in sys.argv:
    gen_semantic_trainstep()


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
    zs = {'surface': {}, 'overhead': {}}

    def mk_pre(tag):
        def pre(m, inp):                       # the one LeakyReLU module is called once per block, in order
            zs[tag][len(zs[tag]) + 1] = inp[0].detach().clone()
        return pre
    for tag in zs:
        encs[tag][0].activation.register_forward_pre_hook(mk_pre(tag))
    params = list(encs['surface'][0].parameters()) + list(encs['overhead'][0].parameters())
    opt = torch.optim.Adam(params)                                   # default lr 1e-3, :349
    s_emb = encs['surface'][0](encs['surface'][1])
    o_emb = encs['overhead'][0](encs['overhead'][1])
    loss = base.exhaustive_minibatch_triplet_loss(s_emb, o_emb)
    opt.zero_grad()
    loss.backward()
    res = {'seed': SEED, 'B': B, 'loss': loss.detach().numpy(), 'embed_surface': s_emb.detach().numpy(),
           'embed_overhead': o_emb.detach().numpy()}
    for tag in zs:
        assert sorted(zs[tag]) == list(range(1, 8))
        baseline_gate_records(res, tag, zs[tag])
    # the hinge terms relu(d_ap^2 - d_an^2 + margin) (:304-313) must not sit on their kink either
    D = ((s_emb.detach()[:, None, :] - o_emb.detach()[None, :, :]) ** 2).sum(-1)
    t1 = (D.diagonal()[:, None] - D + 1.0)[~torch.eye(B, dtype=torch.bool)]
    t2 = (D.diagonal()[None, :] - D + 1.0)[~torch.eye(B, dtype=torch.bool)]
    res['min_abs_hinge'] = np.float32(min(t1.abs().min().item(), t2.abs().min().item()))
    assert res['min_abs_hinge'] > 1e-3, res['min_abs_hinge']
    names = []
    before = {}
    for tag in ('surface', 'overhead'):
        for n, p in encs[tag][0].named_parameters():
            key = '%s.%s' % (tag, n)
            names.append(key)
            gflat = p.grad.reshape(-1)
            res['gnorm:' + key] = np.float64(gflat.double().norm().item())
            res['gsamp:' + key] = gflat[::max(1, gflat.numel() // 129)].numpy()
            before[key] = p.detach().reshape(-1)[::max(1, p.numel() // 129)].clone()
    opt.step()
    for tag in ('surface', 'overhead'):
        for n, p in encs[tag][0].named_parameters():
            after = p.detach().reshape(-1)[::max(1, p.numel() // 129)]
            res['psamp:%s.%s' % (tag, n)] = after.numpy()
            res['dsamp:%s.%s' % (tag, n)] = (after - before['%s.%s' % (tag, n)]).numpy()
        for n, bbuf in encs[tag][0].named_buffers():
            if 'num_batches' not in n:
                res['buf:%s.%s' % (tag, n)] = bbuf.detach().numpy()
    res['names'] = np.array(names)
    assert float(loss) > 0
    np.savez_compressed(os.path.join(HERE, 'baseline_train.npz'), **res)
    print('baseline_train.npz written; loss', float(loss), 'min |hinge argument|', float(res['min_abs_hinge']),
          'fragile gates', {k: len(v) for k, v in res.items() if k.startswith('gfrag:')})


if __name__ == '__main__' and '--baseline-train' in sys.argv:
    gen_baseline_train()


AUGMENT_SHIFTS = (('pixels', 3), ('pixel', -2.5), ('p', 7.5), ('P', 40), ('fraction', 0.25), ('f', -0.4), ('Fractions', 1.7),
                  ('degrees', 90), ('degree', 45.3), ('d', 359), ('D', -200.5), ('radians', 1.0), ('radian', -2.2), ('r', 6.5))


def gen_augment():
    """cvig_baseline's host-side augmentation helpers (model/cvig_baseline.py:97-128): horizontal_shift for every unit alias
    and rounding case, quantized_rotation for factors -2..6, as computed by the reference."""
    fov, sem, base = import_reference()
    img = torch.arange(2 * 3 * 5 * 12, dtype=torch.float32).reshape(2, 3, 5, 12)
    res = {'img': img.numpy(), 'units': np.array([u for u, _ in AUGMENT_SHIFTS]), 'shifts': np.array([s for _, s in AUGMENT_SHIFTS], dtype=np.float64)}
    for i, (u, s) in enumerate(AUGMENT_SHIFTS):
        res['shift_%d' % i] = base.horizontal_shift(img, s, unit=u).numpy()
    try:
        base.horizontal_shift(img, 1, unit='turns')
        raise AssertionError('reference accepted an unknown unit')
    except Exception as e:      # the reference raises a bare Exception with this text
        res['bad_unit_message'] = np.array(str(e))
    rect = torch.arange(3 * 4 * 6, dtype=torch.float32).reshape(3, 4, 6)
    res['rect'] = rect.numpy()
    for f in range(-2, 7):
        res['rot_%d' % f] = base.quantized_rotation(rect, f).contiguous().numpy()
    np.savez(os.path.join(HERE, 'augment.npz'), **res)
    print('augment.npz written:', len(AUGMENT_SHIFTS), 'shifts, 9 rotations')


if __name__ == '__main__' and '--augment' in sys.argv:
    gen_augment()


class FullVGG(FakeVGG):
    """FakeVGG with torchvision's real classifier shapes: what a strict load of a reference-format checkpoint needs."""

    def __init__(self, weights):
        super().__init__(weights)
        self.classifier = nn.Sequential(nn.Linear(25088, 4096), nn.ReLU(True), nn.Dropout(), nn.Linear(4096, 4096), nn.ReLU(True),
                                        nn.Dropout(), nn.Linear(4096, 1000))


def tensor_digest(t):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()[:16]


INTEROP_SEED = SEED + 7


def gen_checkpoint_interop():
    """The proof that checkpoints written by witw_amd.cvig_fov.save_reference_state_dict are the reference's format: the file is
    strict-loaded by the REFERENCE's FOV_DSM.load_state_dict (model/cvig_fov.py:511-512; a VGG16-shaped torch.hub stand-in with
    other weights underneath, so every tensor that counts has to come from the file), and the reference then embeds seeded
    inputs. Stored: the reference module's key list, per-tensor digests of its state after the load, its embeddings.
    tests/test_checkpoints.py re-creates the file from the same seeds and compares."""
    import tempfile
    from witw_amd import cvig_fov as our_fov, cvig_semantic as our_sem
    fov, sem, base = import_reference()
    res = {'seed': INTEROP_SEED}
    for tag, ours, ref_mod, circ, c, stream in (('surface', our_fov, fov, False, 3, 70), ('overhead', our_fov, fov, True, 3, 71),
                                                ('semantic_overhead', our_sem, sem, True, 5, 72)):
        w = synth.fov_dsm_weights(INTEROP_SEED, in_channels=c)
        enc = ours.FOV_DSM(circ_padding=circ, weights=w)
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, 'ck.pth')
            our_fov.save_reference_state_dict(enc, path)
            res['file_bytes_' + tag] = os.path.getsize(path)
            other = synth.fov_dsm_weights(INTEROP_SEED + 100, in_channels=3)
            torch.hub.load = lambda *a, **k: FullVGG(other)
            ref = ref_mod.FOV_DSM(circ_padding=circ)
            out = ref.load_state_dict(torch.load(path, map_location='cpu'))          # strict (the default), as the reference calls it
            assert not out.missing_keys and not out.unexpected_keys
        ref.eval()
        sd = ref.state_dict()
        res['keys_' + tag] = np.array(list(sd.keys()))
        res['digests_' + tag] = np.array([tensor_digest(v) for v in sd.values()])
        res['shapes_' + tag] = np.array([','.join(str(n) for n in v.shape) for v in sd.values()])
        x = torch.from_numpy(synth.normalized_images(INTEROP_SEED, stream, (2, c, 128, 512)))
        with torch.no_grad():
            res['embed_' + tag] = ref(x).numpy()
        res['stream_' + tag] = stream
        print(tag, 'strict load ok:', len(sd), 'tensors, file', res['file_bytes_' + tag], 'bytes')
    np.savez_compressed(os.path.join(HERE, 'checkpoint_interop.npz'), **res)


if __name__ == '__main__' and '--checkpoint-interop' in sys.argv:
    gen_checkpoint_interop()
