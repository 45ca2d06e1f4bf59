"""fp16x3: fp32-grade inference on the fp16 MFMA (csrc/conv3x3_f16x3.hip). Every value is carried as fp16 hi + fp16 lo and
a product is hi*hi + lo*hi + hi*lo with fp32 accumulation. The path is held to the SAME reference goldens and the SAME
1e-4 tolerance as the fp32 MFMA path (embeddings, orientation, distances, ranks), plus single-layer checks against fp64."""
import os

import numpy as np
import pytest
import torch

from oracle import cvig_fov_oracle as O
from witw_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', [(2, 8, 64, 16, 64, 1, True, True, False), (1, 16, 64, 64, 128, 1, False, True, True),
                                  (2, 16, 24, 32, 256, 2, True, True, False), (1, 12, 99, 8, 64, 1, True, True, True),
                                  (2, 4, 64, 64, 16, 1, True, False, False), (1, 9, 130, 24, 136, 1, False, True, False)])
def test_conv3x3_f16x3_vs_fp64(case):
    from witw_amd import ops
    B, H, W, Cin, Cout, sh, circ, relu, pool = case
    g = np.random.Generator(np.random.Philox(key=[17, Cin + Cout]))
    x = torch.from_numpy(g.standard_normal((B, Cin, H, W), dtype=np.float32))
    w = torch.from_numpy((g.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) * (2.0 / (9 * Cin)) ** 0.5))
    b = torch.from_numpy(g.standard_normal((Cout,), dtype=np.float32) * 0.1)
    ref = O.conv3x3(x.double(), w.double(), b.double(), sh, circ)
    if relu:
        ref = torch.relu(ref)
    if pool:
        ref = torch.nn.functional.max_pool2d(ref, 2, 2)
    dev = torch.device('cuda:0')
    xs = ops.nchw_to_split_f16(x.to(dev), Cin)
    back = ops.split_f16_to_f32(xs).cpu().permute(0, 3, 1, 2)
    assert float((back - x).abs().max()) <= 2.0 ** -21 * float(x.abs().max()) + 1e-7          # the split keeps 22 bits
    pk = ops.PackedConvF16x3(w.to(dev), b.to(dev))
    last = Cout % 8 != 0 or Cout == 16
    y = ops.conv3x3_f16x3_fwd(xs, pk, stride_h=sh, circular=circ, relu=relu, pool=pool, out_nchw_f32=last)
    got = y.cpu() if last else ops.split_f16_to_f32(y).cpu().permute(0, 3, 1, 2)
    np.testing.assert_allclose(got.numpy(), ref.float().numpy(), rtol=0, atol=5e-6 * max(1.0, float(ref.abs().max())))


def test_encoder_f16x3_meets_the_fp32_goldens(golden_dir):
    """The reference's own embeddings (tests/golden/encoder.npz) at the fp32 tolerance of 1e-4, both encoders, fov 360
    and fov 70, and the CPU emulation of the split arithmetic."""
    from witw_amd import cvig_fov
    g = np.load(os.path.join(golden_dir, 'encoder.npz'))
    seed = int(g['seed'])
    w = synth.fov_dsm_weights(seed)
    wt = {k: (torch.from_numpy(a), torch.from_numpy(b)) for k, (a, b) in w.items()}
    x360 = torch.from_numpy(synth.normalized_images(seed, 10, (2, 3, 128, 512)))
    worst = 0.0
    for circ in (False, True):
        enc = cvig_fov.FOV_DSM(circ_padding=circ, weights=w).cuda().eval()
        e = enc.forward_f16x3(x360.cuda()).cpu().numpy()
        assert e.shape == (2, 16, 4, 64) and e.dtype == np.float32
        np.testing.assert_allclose(e, g['embed360_circ%d' % circ], rtol=0, atol=1e-4)
        worst = max(worst, float(np.abs(e - g['embed360_circ%d' % circ]).max()))
        with torch.no_grad():
            emu = O.fov_dsm_forward_f16x3_emulated(x360, wt, circ).numpy()
        np.testing.assert_allclose(e, emu, rtol=0, atol=6e-5)       # same arithmetic, different fp32 summation order
        with torch.no_grad():
            e32 = enc(x360.cuda()).cpu().numpy()
        np.testing.assert_allclose(e, e32, rtol=0, atol=6e-5)
    print('fp16x3 encoder vs the reference goldens: max |diff| %.2e (bound 1e-4)' % worst)
    # fov 70 ground branch (ragged widths 99 / 49 / 24 / 12) and the 5-channel semantic variant against their goldens
    x70 = torch.from_numpy(synth.normalized_images(seed, 11, (2, 3, 128, 99)))
    e70 = cvig_fov.FOV_DSM(circ_padding=False, weights=w).cuda().eval().forward_f16x3(x70.cuda()).cpu().numpy()
    if 'embed70_circ0' in g:
        np.testing.assert_allclose(e70, g['embed70_circ0'], rtol=0, atol=1e-4)
    with torch.no_grad():
        np.testing.assert_allclose(e70, O.fov_dsm_forward(x70, wt, False).numpy(), rtol=0, atol=1e-4)
    from witw_amd import cvig_semantic
    gs = np.load(os.path.join(golden_dir, 'encoder_semantic.npz'))
    w5 = synth.fov_dsm_weights(seed, in_channels=5)
    x5 = torch.from_numpy(synth.normalized_images(seed, 12, (1, 5, 128, 512)))
    e5 = cvig_semantic.FOV_DSM(circ_padding=True, weights=w5).cuda().eval().forward_f16x3(x5.cuda()).cpu().numpy()
    np.testing.assert_allclose(e5, gs['embed5_circ1'], rtol=0, atol=1e-4)
    with pytest.raises(Exception):
        enc.train().forward_f16x3(x360.cuda())


def test_f16x3_step_keeps_orientation_and_ranks_of_the_fp32_step():
    from witw_amd import cvig_fov
    dev = torch.device('cuda:0')
    w = synth.fov_dsm_weights(77)
    se = cvig_fov.FOV_DSM(False, weights=w).to(dev).eval()
    oe = cvig_fov.FOV_DSM(True, weights=w).to(dev).eval()
    xo = torch.from_numpy(synth.normalized_images(78, 0, (24, 3, 128, 512))).to(dev)
    xs = torch.stack([torch.roll(xo[i], -17 * i, dims=2) for i in range(24)]) + \
        0.5 * torch.from_numpy(synth.normalized_images(78, 1, (24, 3, 128, 512))).to(dev)
    with torch.no_grad():
        ov32, su32 = oe(xo), se(xs.contiguous())
        ori32, d32 = cvig_fov.match(ov32, su32)
        ori16, d16 = cvig_fov.match(oe.forward_f16x3(xo), se.forward_f16x3(xs.contiguous()))
        sc = O.correlation_scores(ov32.cpu(), su32.cpu())
    # With random-init weights the correlation of NON-matching pairs is nearly flat over the 64 shifts (relative top-2 margins
    # below 1e-3), so their arg-max is a near tie by construction: orientation is compared where the margin is clear, the
    # distance (a function of the maximum only; full-width windows have shift-independent norms) everywhere.
    top2 = torch.topk(sc, 2, dim=-1).values
    clear = ((top2[..., 0] - top2[..., 1]) > 1e-3 * top2[..., 0].abs()).to(dev)
    assert torch.equal(ori32[clear], ori16[clear])
    np.testing.assert_allclose(d16.cpu().numpy(), d32.cpu().numpy(), rtol=0, atol=1e-4)
    # the true matches (diagonal) are far from any tie: identical orientation there, and the ranks agree
    assert torch.equal(torch.diagonal(ori32), torch.diagonal(ori16))
    from witw_amd import ops
    r32, r16 = ops.rank_count(d32), ops.rank_count(d16)
    assert int((r32 != r16).sum()) <= 1, (r32, r16)


def test_training_step_on_f16x3():
    """precision 'fp16x3' under training: forward (frozen trunk and trainable layers, Dropout2d) and dgrad on the fp16x3
    kernels, weight gradients on the exact-fp32 wgrad kernel. Loss within 1e-4 of the fp32 step, every gradient within
    2e-3 of its norm."""
    from witw_amd import cvig_fov
    dev = torch.device('cuda:0')
    B, seed = 8, 91
    w = synth.fov_dsm_weights(seed)
    xo = torch.from_numpy(synth.normalized_images(seed, 1, (B, 3, 128, 512))).to(dev)
    xs = (xo + 0.3 * torch.from_numpy(synth.normalized_images(seed, 2, (B, 3, 128, 512))).to(dev)).contiguous()
    drops = {t: {i: torch.from_numpy(synth.dropout_scales(seed, 10 * k + i, B, 512)).to(dev) for i in (17, 19, 21)}
             for k, t in enumerate('so')}
    out = {}
    for prec in ('fp32', 'fp16x3'):
        se = cvig_fov.FOV_DSM(False, weights=w).to(dev).train()
        oe = cvig_fov.FOV_DSM(True, weights=w).to(dev).train()
        se.precision = oe.precision = prec
        _, dist = cvig_fov.match(oe(xo, dropout_scales=drops['o']), se(xs, dropout_scales=drops['s']))
        loss = cvig_fov.triplet_loss(dist)
        loss.backward()
        grads = {('s.' + n): p.grad for n, p in se.named_parameters() if p.grad is not None}
        grads.update({('o.' + n): p.grad for n, p in oe.named_parameters() if p.grad is not None})
        out[prec] = (loss.item(), grads)
    (l32, g32), (l3, g3) = out['fp32'], out['fp16x3']
    assert abs(l3 - l32) <= 1e-4, (l3, l32)
    assert set(g3) == set(g32) and len(g3) == 24
    worst = max(float((g3[k] - g32[k]).norm() / (g32[k].norm() + 1e-30)) for k in g32)
    assert worst < 2e-3, worst
    print('fp16x3 training step vs fp32: loss %.7f vs %.7f, worst gradient deviation %.2e of its norm' % (l3, l32, worst))


@pytest.mark.parametrize('case', [(16, 16, 64, 64, 128, 1, True), (10, 6, 12, 64, 16, 1, True), (8, 8, 24, 72, 64, 2, False),
                                  (3, 7, 20, 16, 256, 2, True), (9, 5, 9, 128, 64, 1, False)])
def test_wgrad_f16x3_matches_autograd(case):
    """Weight gradient with split products vs CPU autograd in fp64 (operands random fp32: the split keeps 22 bits, the lo*lo
    term is dropped): 3e-6 of the largest entry; bit-reproducible."""
    from witw_amd import ops
    B, H, W, Cin, Cout, sh, circ = case
    g = np.random.Generator(np.random.Philox(key=[23, Cin + Cout]))
    x = torch.from_numpy(g.standard_normal((B, Cin, H, W), dtype=np.float32))
    w = torch.zeros((Cout, Cin, 3, 3), dtype=torch.float64, requires_grad=True)
    b = torch.zeros((Cout,), dtype=torch.float64, requires_grad=True)
    y = O.conv3x3(x.double(), w, b, sh, circ)
    gy = torch.from_numpy(g.standard_normal(tuple(y.shape)).astype(np.float32))
    y.backward(gy.double())
    dev = torch.device('cuda:0')
    xs, gs = ops.nchw_to_split_f16(x.to(dev), Cin), ops.nchw_to_split_f16(gy.to(dev), Cout)
    oct_ = ops.split_f16_to_octet(xs)
    assert oct_.shape == ((B + 7) // 8, H, W, Cin, 2, 8)
    dw, db = ops.conv3x3_wgrad_f16x3(xs, gs, Cin, stride_h=sh, circular=circ)
    np.testing.assert_allclose(dw.cpu().numpy(), w.grad.float().numpy(), rtol=0, atol=3e-6 * max(1.0, float(w.grad.abs().max())))
    np.testing.assert_allclose(db.cpu().numpy(), b.grad.float().numpy(), rtol=0, atol=3e-6 * max(1.0, float(b.grad.abs().max())))
    dw2, db2 = ops.conv3x3_wgrad_f16x3(xs, gs, Cin, stride_h=sh, circular=circ)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


def test_semantic_f16x3_training_step_vs_fp32_path():
    """cvig_semantic (layer 0 trainable): the fp16x3 backward walks all 13 layers (dgrad through the frozen VGG filters, arg-max
    scatter behind the three fused max-pools on split tensors, layer-0 wgrad with 5 of 8 padded channels)."""
    from witw_amd import cvig_fov, cvig_semantic
    dev = torch.device('cuda:0')
    B, seed = 8, 41
    w = synth.fov_dsm_weights(seed, in_channels=5)
    xo = torch.from_numpy(synth.normalized_images(seed, 1, (B, 5, 128, 512))).to(dev)
    xs = (xo + 0.3 * torch.from_numpy(synth.normalized_images(seed, 2, (B, 5, 128, 512))).to(dev)).contiguous()
    drops = {t: {i: torch.from_numpy(synth.dropout_scales(seed, 10 * k + i, B, 512)).to(dev) for i in (17, 19, 21)}
             for k, t in enumerate('so')}
    out = {}
    for prec in ('fp32', 'fp16x3'):
        se = cvig_semantic.FOV_DSM(False, weights=w).to(dev).train()
        oe = cvig_semantic.FOV_DSM(True, weights=w).to(dev).train()
        se.precision = oe.precision = prec
        _, dist = cvig_fov.match(oe(xo, dropout_scales=drops['o']), se(xs, dropout_scales=drops['s']))
        loss = cvig_fov.triplet_loss(dist)
        loss.backward()
        grads = {('s.' + n): p.grad for n, p in se.named_parameters() if p.grad is not None}
        grads.update({('o.' + n): p.grad for n, p in oe.named_parameters() if p.grad is not None})
        out[prec] = (loss.item(), grads)
    (l32, g32), (l3, g3) = out['fp32'], out['fp16x3']
    assert abs(l3 - l32) <= 1e-4, (l3, l32)
    assert set(g3) == set(g32) and len(g3) == 28
    worst = ('', 0.0)
    for k in g32:
        rel = float((g3[k] - g32[k]).norm() / (g32[k].norm() + 1e-30))
        if rel > worst[1]:
            worst = (k, rel)
        # layer 0 sits behind three arg-max routings: a near-tie that routes differently moves one pixel's gradient
        assert rel < (5e-2 if 'features.0.' in k else 5e-3), (k, rel)
    print('semantic fp16x3 vs fp32 step: loss %.7f vs %.7f, worst gradient deviation %.2e (%s)' % (l3, l32, worst[1], worst[0]))


def test_f16x3_flags_values_beyond_the_fp16_range():
    """|v| > 65504 cannot be carried as fp16 hi + lo: the kernel raises the device flag instead of passing infinities on."""
    from witw_amd import ops
    dev = torch.device('cuda:0')
    ops.f16x3_overflowed(dev)           # clear
    x = torch.ones(1, 8, 8, 64)
    w = torch.zeros(64, 8, 3, 3)
    w[:, :, 1, 1] = 1.0
    pk = ops.PackedConvF16x3(w.to(dev), torch.zeros(64).to(dev))
    ops.conv3x3_f16x3_fwd(ops.nchw_to_split_f16((x * 100.0).to(dev), 8), pk)                     # 8 * 100: fine
    assert not ops.f16x3_overflowed(dev)
    ops.conv3x3_f16x3_fwd(ops.nchw_to_split_f16((x * 2.0e4).to(dev), 8), pk)                     # 8 * 2e4 = 1.6e5 > 65504
    assert ops.f16x3_overflowed(dev) and not ops.f16x3_overflowed(dev)
