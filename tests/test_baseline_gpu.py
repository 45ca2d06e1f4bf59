"""GPU parity of the cvig_baseline path (model/cvig_baseline.py) against the reference goldens."""
import os

import numpy as np
import pytest
import torch

from oracle import cvig_baseline_oracle as OB
from witw_amd import synth

pytestmark = pytest.mark.gpu


def _load_encoder(cls, seed):
    enc = cls()
    with torch.no_grad():
        for i, q in enumerate(synth.baseline_params(seed), 1):
            getattr(enc, 'conv%d' % i).weight.copy_(torch.from_numpy(q['w']))
            getattr(enc, 'conv%d' % i).bias.copy_(torch.from_numpy(q['b']))
            bn = getattr(enc, 'bn%d' % i)
            bn.weight.copy_(torch.from_numpy(q['gamma']))
            bn.bias.copy_(torch.from_numpy(q['beta']))
            bn.running_mean.copy_(torch.from_numpy(q['mean']))
            bn.running_var.copy_(torch.from_numpy(q['var']))
    return enc.cuda().eval()


def test_baseline_encoders_match_reference_goldens(golden_dir):
    from witw_amd import cvig_baseline
    g = np.load(os.path.join(golden_dir, 'baseline.npz'))
    seed = int(g['seed'])
    for tag, cls, hw, stream, off in (('surface', cvig_baseline.SurfaceEncoder, 500, 30, 0),
                                      ('overhead', cvig_baseline.OverheadEncoder, 512, 31, 1)):
        enc = _load_encoder(cls, seed + off)
        assert sorted(enc.state_dict().keys()) == list(g['keys_' + tag])
        x = torch.from_numpy(synth.images_u8(seed, stream, (2, 3, hw, hw))).cuda()
        e = enc(x).cpu().numpy()
        assert e.shape == (2, 1536)
        np.testing.assert_allclose(e, g['embed_' + tag], rtol=0, atol=1e-4)   # north_star tolerance; values ~0.03-0.3
    with pytest.raises(Exception):
        enc(torch.zeros(1, 3, 224, 224).cuda())      # the reference fails below 382 px too


def test_baseline_odd_sizes_vs_oracle():
    from witw_amd import cvig_baseline
    enc = _load_encoder(cvig_baseline.SurfaceEncoder, 77)
    prm = [{k: torch.from_numpy(v) for k, v in q.items()} for q in synth.baseline_params(77)]
    x = torch.from_numpy(synth.images_u8(5, 1, (1, 3, 448, 611)))
    with torch.no_grad():
        ref = OB.encoder_forward(x, prm).numpy()
    np.testing.assert_allclose(enc(x.cuda()).cpu().numpy(), ref, rtol=0, atol=1e-4)


@pytest.mark.parametrize('shape', [(4, 512, 512), (8, 512, 512), (5, 500, 500), (32, 512, 512), (3, 382, 382), (7, 384, 640), (2, 768, 1024)])
def test_baseline_encoder_config1_batches_vs_oracle(shape):
    """BASELINE config 1 shapes at batch sizes that pick the 8-wave workgroups (>= 4 images of 512 x 512): round 1 only ran 2
    images here, and the 8-wave 64-channel 4-tap variant wrote two of its epilogue slabs past the end of the LDS buffer. The
    smallest legal input (382: block 7 has a single valid output), non-square maps (different mosaic factors per axis are not
    allowed: the smaller one is taken) and maps too large for a mosaic in block 5 exercise the eval path's block 5-7 layouts."""
    from witw_amd import cvig_baseline
    B, H, W = shape
    enc = _load_encoder(cvig_baseline.OverheadEncoder, 4243)
    prm = [{k: torch.from_numpy(v) for k, v in q.items()} for q in synth.baseline_params(4243)]
    x = torch.from_numpy(synth.images_u8(9, B, (B, 3, H, W)))
    with torch.no_grad():
        ref = OB.encoder_forward(x, prm).numpy()
    got = enc(x.cuda()).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-4 * max(1.0, float(np.abs(ref).max())))


def test_baseline_loss_and_ranks(golden_dir):
    from witw_amd import cvig_baseline
    g = np.load(os.path.join(golden_dir, 'baseline.npz'))
    seed = int(g['seed'])
    e1 = torch.from_numpy(synth.embeddings(seed, 600, (5, 1536))) * 0.018
    e2 = e1 + torch.from_numpy(synth.embeddings(seed, 601, (5, 1536))) * 0.02
    f = cvig_baseline.exhaustive_minibatch_triplet_loss
    np.testing.assert_allclose(f(e1.cuda(), e2.cuda()).item(), float(g['loss_hard']), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(f(e1.cuda(), e2.cuda(), soft_margin=True).item(), float(g['loss_soft']), rtol=1e-3, atol=1e-8)
    np.testing.assert_allclose(f((e1 * 0.55).cuda(), (e2 * 0.55).cuda(), margin=0.3).item(), float(g['loss_hard_m03']),
                               rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(f(e1.cuda(), e2.cuda(), soft_margin=True, alpha=2.).item(), float(g['loss_soft_a2']), rtol=1e-4)
    ov = torch.from_numpy(synth.embeddings(seed, 602, (14, 1536)))
    su = ov + 14.0 * torch.from_numpy(synth.embeddings(seed, 603, (14, 1536)))
    np.testing.assert_array_equal(cvig_baseline.ranks(ov.cuda(), su.cuda()), g['ranks'])       # bit-exact ranks
    big1 = torch.from_numpy(synth.embeddings(1, 2, (300, 1536))) * 0.02
    big2 = big1 + torch.from_numpy(synth.embeddings(1, 3, (300, 1536))) * 0.02
    np.testing.assert_allclose(f(big1.cuda(), big2.cuda()).item(), OB.exhaustive_minibatch_triplet_loss(big1, big2).item(),
                               rtol=1e-4)


def _reconciled_lrelu_acts(g, tag, enc, out):
    """LeakyReLU gates of the GPU forward (sign of the recorded activations) against the reference's: they may differ only at
    positions the golden lists as fragile (tests/test_trainstep_golden.reconcile_gates); at those the reference's side is
    taken by giving the activation the reference's sign (its magnitude is rounding-level there). -> positions flipped."""
    from tests.test_trainstep_golden import reconcile_gates
    flips = 0
    for i in range(1, 8):
        _h, a, (vh, vw), *_rest = enc._last_saved[i - 1]
        gate = (a[:, :vh, :vw, :] > 0).permute(0, 3, 1, 2).cpu()
        fixed, n = reconcile_gates(g, tag, i, gate)
        flips += n
        if n:
            want = torch.zeros(a.shape, dtype=torch.bool, device=a.device)
            want[:, :vh, :vw, :] = fixed.permute(0, 2, 3, 1).to(a.device)
            inside = torch.zeros_like(want)
            inside[:, :vh, :vw, :] = True
            mag = a.abs()
            out[i] = torch.where(inside & (want != (a > 0)), torch.where(want, mag + 1e-30, -mag), a).contiguous()
    return flips


def test_baseline_training_step_matches_reference_golden(golden_dir):
    """model/cvig_baseline.py:373-387 on the GPU against the reference's own run: train-mode encoders (BatchNorm batch
    statistics + running-stat update), exhaustive triplet loss, backward through everything, Adam(lr=1e-3).
    LeakyReLU gates may differ from the reference's only at listed fragile positions (|conv output| < 1e-4 x max(1, std));
    with the reference's side taken there all 56 gradients are within 1e-4 of their norm and the Adam update is the reference's."""
    from witw_amd import cvig_baseline, cvig_fov
    from tests.test_trainstep_golden import check_adam_update
    g = np.load(os.path.join(golden_dir, 'baseline_train.npz'))
    seed, B = int(g['seed']), int(g['B'])
    xs = torch.from_numpy(synth.images_u8(seed, 40, (B, 3, 400, 400))).cuda()
    xo = torch.from_numpy(synth.images_u8(seed, 41, (B, 3, 416, 416))).cuda()
    se = _load_encoder(cvig_baseline.SurfaceEncoder, seed + 10).train()
    oe = _load_encoder(cvig_baseline.OverheadEncoder, seed + 11).train()
    se.keep_activations = oe.keep_activations = True
    opt = cvig_fov.Adam(list(se.parameters()) + list(oe.parameters()), lr=1e-3)
    acts_s, acts_o = {}, {}
    es = se(xs, lrelu_acts=acts_s)
    flips = _reconciled_lrelu_acts(g, 'surface', se, acts_s)
    eo = oe(xo, lrelu_acts=acts_o)
    flips += _reconciled_lrelu_acts(g, 'overhead', oe, acts_o)
    print('baseline: GPU forward differs from the reference at %d fragile LeakyReLU gates' % flips)
    assert flips <= 40, flips
    loss = cvig_baseline.exhaustive_minibatch_triplet_loss(es, eo)
    opt.zero_grad()
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(g['loss']), rtol=0, atol=1e-4)
    np.testing.assert_allclose(es.detach().cpu().numpy(), g['embed_surface'], rtol=0, atol=1e-4)
    np.testing.assert_allclose(eo.detach().cpu().numpy(), g['embed_overhead'], rtol=0, atol=1e-4)
    named = {('surface.' + n): p for n, p in se.named_parameters()}
    named.update({('overhead.' + n): p for n, p in oe.named_parameters()})
    worst = 0.0
    for name in g['names']:
        name = str(name)
        p = named[name]
        ref = g['gsamp:' + name]
        got = p.grad.detach().reshape(-1).cpu()
        got_s = got[::max(1, got.numel() // 129)].numpy()
        gn = float(g['gnorm:' + name])
        assert abs(got.double().norm().item() - gn) <= 1e-4 * gn + 1e-12, name
        rel = np.linalg.norm(got_s - ref) / (np.linalg.norm(ref) + 1e-30)
        worst = max(worst, rel)
        assert rel <= 1e-4, (name, rel)
    print('baseline: worst gradient deviation %.2e of its norm' % worst)
    for tag, enc in (('surface', se), ('overhead', oe)):
        for n, bbuf in enc.named_buffers():
            if 'num_batches' not in n:
                np.testing.assert_allclose(bbuf.cpu().numpy(), g['buf:%s.%s' % (tag, n)], rtol=1e-4, atol=1e-6)
    before = {str(nm): named[str(nm)].detach().clone() for nm in g['names']}
    opt.step()
    for name in g['names']:
        check_adam_update(g, str(name), before[str(name)].cpu(), named[str(name)].detach().cpu(), 1e-3, n=129,
                          own_grad=named[str(name)].grad.detach().cpu(), min_cover=0.5)    # the last blocks' bias gradients are small


@pytest.mark.parametrize('case', [(2, 20, 70, 24, 128, 4), (1, 64, 64, 16, 64, 8), (2, 9, 30, 64, 192, 4), (3, 16, 130, 8, 64, 4),
                                  (4, 256, 256, 16, 64, 8)])      # the last: block 1 of a 512 x 512 batch of 4 (8 waves, 64-channel tile)
def test_taps4_kernels_equal_the_zero_filled_3x3_form(case):
    """The 4-tap kernels (witw_conv3x3_fwd_taps4 / _wgrad_taps4) against the full 3x3 kernels on a filter whose first tap
    row/column are zero: the zero taps only ever add exact zeros, so forward, dgrad and the live weight-gradient taps are
    BIT-identical; the dead taps come back as exact zeros."""
    import torch
    from witw_amd import ops
    B, H, W, Cin, Cout, _nw = case
    g = np.random.Generator(np.random.Philox(key=[77, Cin + Cout]))
    dev = torch.device('cuda:0')
    x = torch.from_numpy(g.standard_normal((B, H, W, Cin), dtype=np.float32)).to(dev)
    k3 = torch.from_numpy(g.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) * 0.05)
    k3[:, :, 0, :] = 0
    k3[:, :, :, 0] = 0
    k3 = k3.to(dev)
    b = torch.from_numpy(g.standard_normal((Cout,), dtype=np.float32) * 0.1).to(dev)
    full, live = ops.PackedConv(k3, b), ops.PackedConv(k3, b, taps4=True)
    y9 = ops.conv3x3_fwd(x, full, relu=False, lrelu_slope=0.2)
    y4 = ops.conv3x3_fwd(x, live, relu=False, lrelu_slope=0.2)
    assert torch.equal(y9, y4)
    dz = torch.from_numpy(g.standard_normal((B, H, W, Cout), dtype=np.float32)).to(dev)
    if Cout % 8 == 0:
        d9 = ops.conv3x3_fwd(dz, ops.PackedConv(k3, None, transpose_flip=True), relu=False)
        d4 = ops.conv3x3_fwd(dz, ops.PackedConv(k3, None, transpose_flip=True, taps4=True), relu=False)
        assert torch.equal(d9, d4)
    w9, b9 = ops.conv3x3_wgrad(x, dz, Cin)
    w4, b4 = ops.conv3x3_wgrad(x, dz, Cin, taps4=True)
    assert torch.equal(w9[:, :, 1:, 1:], w4[:, :, 1:, 1:]) and torch.equal(b9, b4)
    assert float(w4[:, :, 0, :].abs().max()) == 0.0 and float(w4[:, :, :, 0].abs().max()) == 0.0


def _taps4_case(B, H, W, Cin, Cout, seed=5):
    import torch
    from witw_amd import ops
    g = np.random.Generator(np.random.Philox(key=[seed, Cin + Cout + H]))
    dev = torch.device('cuda:0')
    x = torch.from_numpy(g.standard_normal((B, H, W, Cin), dtype=np.float32)).to(dev)
    k3 = torch.from_numpy(g.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) * 0.05)
    k3[:, :, 0, :] = 0
    k3[:, :, :, 0] = 0
    b = torch.from_numpy(g.standard_normal((Cout,), dtype=np.float32) * 0.1).to(dev)
    sc = torch.from_numpy(1 + 0.1 * g.standard_normal((Cout,), dtype=np.float32)).to(dev)
    sh = torch.from_numpy(0.1 * g.standard_normal((Cout,), dtype=np.float32)).to(dev)
    return x, ops.PackedConv(k3.to(dev), b, taps4=True), sc, sh


@pytest.mark.parametrize('case', [(2, 20, 70, 24, 128, (19, 69)), (1, 64, 64, 16, 64, (63, 63)), (3, 16, 16, 64, 192, (15, 15)),
                                  (2, 9, 30, 64, 64, (8, 29)), (4, 256, 256, 16, 64, (255, 255)), (2, 32, 32, 8, 128, (30, 31))])
def test_taps4_s2d_epilogue_equals_conv_then_space_to_depth(case):
    """conv_taps4_s2d (the next block's space-to-depth input written by the conv epilogue) is BIT-identical to the conv followed
    by the separate witw_space_to_depth2 pass, zeros outside the valid region included (wide 4- and 8-wave and narrow tiles)."""
    import torch
    from witw_amd import ops
    B, H, W, Cin, Cout, valid = case
    x, packed, sc, sh = _taps4_case(B, H, W, Cin, Cout)
    y = ops.conv3x3_fwd(x, packed, relu=False, lrelu_slope=0.2, post_scale=sc, post_shift=sh)
    want = ops.space_to_depth2(y, valid_hw=valid, cpad=4 * Cout)
    got = ops.conv_taps4_s2d(x, packed, valid, lrelu_slope=0.2, post_scale=sc, post_shift=sh)
    assert got.shape == want.shape and torch.equal(got, want)


@pytest.mark.parametrize('case', [(5, 1, 16, 256, 128, None), (5, 2, 8, 256, 128, 3), (7, 4, 4, 512, 64, 5), (32, 4, 4, 2048, 512, None),
                                  (3, 1, 12, 64, 64, 8), (16, 2, 8, 2048, 512, 32)])
def test_taps4_splitk_mosaic_equals_plain_conv(case):
    """Split-K over a g x g mosaic (what blocks 5-7 of the eval encoder run) against the plain taps4 conv per image: the K slices
    are summed in a different order, so equality is to fp32 rounding of the K = 4*Cin sum; B not a multiple of g^2, uneven K
    slices and the library's own ksplit choice are covered; the mosaic space-to-depth with g = 1 is the plain one bit for bit."""
    import torch
    from witw_amd import ops, _lib
    B, g, h, Cin, Cout, ksplit = case
    x, packed, sc, sh = _taps4_case(B, h, h, Cin, Cout, seed=9)
    y = ops.conv3x3_fwd(x, packed, relu=False, lrelu_slope=0.2, post_scale=sc, post_shift=sh)[:, :h - 1, :h - 1].contiguous()
    # build the mosaic of the INPUT maps by hand: image b -> cell (b % g^2) of mosaic b // g^2
    Bm = (B + g * g - 1) // (g * g)
    xm = torch.zeros((Bm, g * h, g * h, Cin), device=x.device)
    for b in range(B):
        bm, cell = divmod(b, g * g)
        cy, cx = divmod(cell, g)
        xm[bm, cy * h:(cy + 1) * h, cx * h:(cx + 1) * h] = x[b]
    got = ops.conv_taps4_splitk(xm, packed, B, g, (h - 1, h - 1), lrelu_slope=0.2, post_scale=sc, post_shift=sh, ksplit=ksplit)
    assert got.shape == y.shape
    scale = float(y.abs().max())
    assert float((got - y).abs().max()) <= 4e-6 * scale * max(1.0, (4 * Cin) ** 0.5 / 16)
    if ksplit is None:
        assert _lib.load().witw_conv3x3_taps4_ksplit(Bm, g * h, g * h, Cin, Cout) >= 1
    # mosaic space-to-depth: cell b of the output is the plain space-to-depth image of y[b]
    plain = ops.space_to_depth2(y, cpad=4 * Cout)
    assert torch.equal(ops.space_to_depth2_mosaic(y, 1), plain)
    mos = ops.space_to_depth2_mosaic(y, g)
    h2 = plain.shape[1]
    assert mos.shape == (Bm, g * h2, g * h2, 4 * Cout)
    for b in range(B):
        bm, cell = divmod(b, g * g)
        cy, cx = divmod(cell, g)
        assert torch.equal(mos[bm, cy * h2:(cy + 1) * h2, cx * h2:(cx + 1) * h2], plain[b])
    for cell in range(B % (g * g) or g * g, g * g):       # cells without an image are zero
        cy, cx = divmod(cell, g)
        assert float(mos[Bm - 1, cy * h2:(cy + 1) * h2, cx * h2:(cx + 1) * h2].abs().max()) == 0.0


def test_taps4_ex_rejects_bad_arguments():
    import torch
    from witw_amd import ops, _lib
    x, packed, sc, sh = _taps4_case(2, 16, 16, 64, 64)
    with pytest.raises(_lib.WitwError):
        ops.conv_taps4_s2d(x, packed, (17, 15))                    # valid region outside the map
    with pytest.raises(_lib.WitwError):
        ops.conv_taps4_s2d(x, packed, (15, 16 + 1))
    with pytest.raises(_lib.WitwError):
        ops.conv_taps4_splitk(x, packed, 2, 1, (15, 15), ksplit=9)  # more slices than K-chunks (64 / 8 = 8)
    with pytest.raises(_lib.WitwError):
        ops.conv_taps4_splitk(x, packed, 9, 2, (7, 7))             # 9 images do not fit 2 mosaics of 2x2


@pytest.mark.parametrize('case', [(2, 12, 10, (11, 9), 64), (1, 8, 8, (8, 8), 128), (3, 7, 9, (5, 9), 8)])
def test_space_to_depth_quad_form_equals_the_scalar_form(case):
    """witw_space_to_depth2 takes 16-byte accesses when C % 4 == 0 and Cpad == 4C; the scalar kernel (reached here through a
    wider Cpad) must give the same bits, with and without the train-mode BatchNorm affine, and the plain re-layout must equal
    an index-by-index copy."""
    import torch
    from witw_amd import ops
    B, Hp, Wp, valid, C = case
    g = np.random.Generator(np.random.Philox(key=[78, C + Hp]))
    dev = torch.device('cuda:0')
    x = torch.from_numpy(g.standard_normal((B, Hp, Wp, C), dtype=np.float32)).to(dev)
    sc = torch.from_numpy(1 + 0.1 * g.standard_normal((C,), dtype=np.float32)).to(dev)
    sh = torch.from_numpy(0.1 * g.standard_normal((C,), dtype=np.float32)).to(dev)
    for kw in ({}, {'scale': sc, 'shift': sh}):
        quad = ops.space_to_depth2(x, valid_hw=valid, cpad=4 * C, **kw)
        scalar = ops.space_to_depth2(x, valid_hw=valid, cpad=4 * C + 8, **kw)
        assert torch.equal(quad, scalar[..., :4 * C]) and float(scalar[..., 4 * C:].abs().max()) == 0.0
    H, W = valid
    xv = torch.zeros((B, 2 * ((H + 1) // 2), 2 * ((W + 1) // 2), C), device=dev)
    xv[:, :H, :W] = x[:, :H, :W]
    want = torch.cat([xv[:, dy::2, dx::2] for dy in (0, 1) for dx in (0, 1)], dim=3)
    assert torch.equal(ops.space_to_depth2(x, valid_hw=valid, cpad=4 * C), want)


@pytest.mark.parametrize('case', [(3, 10, 12, (9, 11), 64, False), (2, 8, 8, (7, 7), 128, True), (2, 6, 10, (6, 9), 6, False),
                                  (4, 64, 64, (61, 61), 64, True)])
def test_bn_lrelu_backward_against_autograd(case):
    """witw_bn_lrelu_bwd(_ex) (channel-quad form for C % 4 == 0, scalar otherwise; gradient in place or in the space-to-depth
    layout of the next block's data-gradient conv) against torch autograd through BatchNorm2d(train)(LeakyReLU(z)) in float64
    over the valid region (model/cvig_baseline.py:267-275)."""
    import torch
    from witw_amd import ops
    B, Hp, Wp, (H, W), C, s2d = case
    g = np.random.Generator(np.random.Philox(key=[79, C + Hp]))
    z = torch.from_numpy(g.standard_normal((B, Hp, Wp, C), dtype=np.float32))
    gamma = torch.from_numpy(1 + 0.1 * g.standard_normal((C,), dtype=np.float32))
    beta = torch.from_numpy(0.1 * g.standard_normal((C,), dtype=np.float32))
    dy = torch.from_numpy(g.standard_normal((B, Hp, Wp, C), dtype=np.float32))
    dy[:, H:] = 0
    dy[:, :, W:] = 0
    zr = z[:, :H, :W].double().permute(0, 3, 1, 2).clone().requires_grad_(True)
    gr, br = gamma.double().clone().requires_grad_(True), beta.double().clone().requires_grad_(True)
    y = torch.nn.functional.batch_norm(torch.nn.functional.leaky_relu(zr, 0.2), None, None, gr, br, True, 0.1, 1e-5)
    y.backward(dy[:, :H, :W].double().permute(0, 3, 1, 2))
    dev = torch.device('cuda:0')
    a = torch.nn.functional.leaky_relu(z, 0.2).to(dev)
    mean, invstd, _sc, _sh = ops.bn_train_stats(a, (H, W), gamma.to(dev), beta.to(dev))
    dyd = dy.to(dev)
    if s2d:
        dyd = ops.space_to_depth2(dyd, valid_hw=(H, W), cpad=4 * C)
    dz, dg, db = ops.bn_lrelu_bwd(a, dyd, (H, W), mean, invstd, gamma.to(dev), 0.2, dy_s2d=s2d)
    want = zr.grad.permute(0, 2, 3, 1)
    got = dz.cpu().double()
    assert float(got[:, H:].abs().max() if H < Hp else 0) == 0.0 and float(got[:, :, W:].abs().max() if W < Wp else 0) == 0.0
    scale = float(want.abs().max())
    assert float((got[:, :H, :W] - want).abs().max()) <= 2e-5 * scale
    assert float((dg.cpu().double() - gr.grad).abs().max()) <= 1e-5 * float(gr.grad.abs().max())
    assert float((db.cpu().double() - br.grad).abs().max()) <= 1e-5 * float(br.grad.abs().max())


@pytest.mark.parametrize('case', [(2, 3, 500, 500), (3, 5, 382, 390), (1, 3, 448, 611), (4, 1, 65, 130)])
def test_first_block_in_one_launch_vs_float64_and_two_launch_form(case):
    """witw_conv4x4s2_first_fwd (normalisation + Conv2d(k=4,s=2) + LeakyReLU + eval BatchNorm affine, written as the second block's
    space-to-depth input; model/cvig_baseline.py:236-240, 265-268) against the same arithmetic in float64 (torch conv2d) and against
    the library's two-launch form (space_to_depth2(normalize) + 2x2-tap conv with the s2d epilogue): same zeros outside the valid
    outputs, values within fp32 rounding of each other (different summation order)."""
    from witw_amd import ops
    B, C, H, W = case
    g = np.random.Generator(np.random.Philox(key=[81, C * 1000 + H]))
    dev = torch.device('cuda:0')
    x = torch.from_numpy(g.integers(0, 256, (B, C, H, W)).astype(np.float32))
    w = torch.from_numpy(g.standard_normal((64, C, 4, 4), dtype=np.float32) * 0.1)
    b = torch.from_numpy(g.standard_normal((64,), dtype=np.float32) * 0.1)
    sc = torch.from_numpy(1 + 0.1 * g.standard_normal((64,), dtype=np.float32))
    sh = torch.from_numpy(0.1 * g.standard_normal((64,), dtype=np.float32))
    y = ops.conv4x4s2_first(x.to(dev), w.to(dev), b.to(dev), sc.to(dev), sh.to(dev))
    assert ops.last_kernel_variant() == 'conv4x4s2_first_kernel<%d>' % C
    vh, vw = (H - 4) // 2 + 1, (W - 4) // 2 + 1
    xn = -1.0 + 2.0 * (x.double() / 255.0)
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(xn, w.double(), b.double(), stride=2), 0.2)
    ref = ref * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]          # [B,64,vh,vw]
    full = torch.zeros((B, 64, 2 * ((vh + 1) // 2), 2 * ((vw + 1) // 2)), dtype=torch.float64)
    full[:, :, :vh, :vw] = ref
    want = torch.cat([full[:, :, dy::2, dx::2] for dy in (0, 1) for dx in (0, 1)], dim=1).permute(0, 2, 3, 1)      # channel (dy*2+dx)*64 + n
    got = y.cpu().double()
    assert got.shape == want.shape
    assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max())
    outside = want == 0
    assert bool((got[outside] == 0).all())
    # the two-launch form of the same block
    cpad = (4 * C + 7) // 8 * 8
    k3 = ops.conv4x4_to_k3(w.to(dev), cpad)
    packed = ops.PackedConv(k3, b.to(dev), taps4=True)
    h = ops.space_to_depth2(x.to(dev), in_nchw=True, normalize=True, cpad=cpad)
    two = ops.conv_taps4_s2d(h, packed, (vh, vw), lrelu_slope=0.2, post_scale=sc.to(dev), post_shift=sh.to(dev))
    assert two.shape == y.shape
    assert float((two - y).abs().max()) <= 2e-6 * float(want.abs().max())
    assert bool(((two == 0) == (y == 0)).all()) or float(((two == 0) != (y == 0)).float().mean()) < 1e-6


def test_baseline_eval_forward_with_and_without_the_fused_first_block():
    """SurfaceEncoder's eval forward through witw_conv4x4s2_first_fwd equals the forward through the re-layout pass + 2x2-tap conv to
    fp32 rounding (1e-5 of the embedding scale), for 3-band and 5-input (orientation) encoders."""
    from witw_amd import cvig_baseline
    for orientation, C in ((False, 3), (True, 5)):
        torch.manual_seed(5)
        enc = cvig_baseline.SurfaceEncoder(orientation=orientation).cuda().eval()
        x = torch.from_numpy(synth.images_u8(11, 3, (3, C, 400, 420))).cuda()
        enc.fused_first = True
        a = enc(x)
        enc.fused_first = False
        b = enc(x)
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
