"""Mixed-precision (bf16 MFMA) training step of the FOV_DSM encoder: FOV_DSM.precision = 'bf16'.

The reference trains in fp32 only (model/cvig_fov.py:439-465), so the bf16 step has no reference golden; it is pinned
  * kernel by kernel against CPU autograd on bf16-representable operands (products of bf16 values are exact in fp32,
    so the weight gradient must agree to fp32 summation order: 2e-5 of the largest entry; bf16 activation gradients
    to one bf16 ulp),
  * end to end against the fp32 HIP path, which itself is pinned to the reference's training-loop golden
    (tests/test_backward_gpu.py): loss within 2e-2 relative, every trainable gradient within 8e-2 of its norm,
  * and by the loss going down under Adam.
"""
import numpy as np
import pytest
import torch

from oracle import cvig_fov_oracle as O
from witw_amd import synth

pytestmark = pytest.mark.gpu


def _rand_bf16(stream, shape, scale=1.0):
    """fp32 tensor whose values are exactly representable in bf16."""
    g = np.random.Generator(np.random.Philox(key=[4242, stream]))
    return (torch.from_numpy(g.standard_normal(shape, dtype=np.float32)) * scale).bfloat16().float()


def _nhwc_bf16(t, dev):
    return t.permute(0, 2, 3, 1).contiguous().to(dev).bfloat16()


WGRAD_CASES = [  # B, H, W, Cin, Cout, stride_h, circ
    (16, 16, 64, 64, 128, 1, True),
    (10, 6, 12, 64, 16, 1, True),        # ragged image octet, ragged column segment (fov-70 width), Cout 16
    (8, 8, 24, 72, 64, 2, False),        # stride (2,1), second ci tile mostly out of range
    (3, 7, 20, 16, 256, 2, True),        # odd height under stride 2, two co tiles
    (9, 5, 9, 128, 64, 1, False),        # odd width
    (4, 16, 64, 256, 512, 1, True),      # layer 17's shape (16 tiles); two row groups of 8, four column segments, circular wrap
    (2, 16, 64, 512, 256, 2, False),     # layer 23 (stride (2,1)): halo rows 2r .. 2r + 2, row groups of 4
    (5, 19, 37, 40, 24, 1, True),        # nothing divides: 3 row groups (one ragged), 3 column segments (one ragged), partial tiles
    (2, 32, 48, 8, 64, 1, False),        # 8 input channels (one 16-byte chunk per pixel): cvig_semantic's layer 0 is 5 -> padded
    (3, 9, 16, 64, 64, 2, True),         # W == one column segment: both circular wraps inside one stage
    (3, 12, 32, 16, 128, 1, True),       # wave roles 1 x 4 (one ci slab), two row blocks along k
    (6, 10, 20, 24, 32, 1, False),       # wave roles 1 x 1: all 8 waves split the stage's rows
    (4, 9, 24, 64, 16, 2, True),         # stride (2,1) with one co slab: roles 2 x 1, four row blocks
    # the 16x16x32 form (stride 1, W % 32 == 0, Cin > 32, Cout > 64): stages of 4 rows x 32 columns
    (3, 6, 32, 72, 136, 1, True),        # one column segment (both circular wraps in one stage), ragged row group, partial ci and co tiles
    (2, 9, 96, 64, 128, 1, False),       # three column segments, three row groups (one ragged)
    (5, 4, 64, 128, 72, 1, True),        # second co block of the wave mostly out of range
]


@pytest.fixture
def wgrad_mfma16():
    """the 16x16x32 form of the NHWC weight gradient (off by default: it measured slower), switched on for one test"""
    from witw_amd import _lib
    lib = _lib.load()
    prev = lib.witw_conv3x3_wgrad_bf16_mfma16(1)
    yield
    lib.witw_conv3x3_wgrad_bf16_mfma16(prev)


@pytest.mark.parametrize('case', [0])
def test_wgrad_bf16_mfma16_form_matches_autograd(case, wgrad_mfma16):
    """every eligible case of WGRAD_CASES on conv3x3_wgrad_bf16_nhwc16_kernel"""
    n = 0
    for c in WGRAD_CASES:
        B, H, W, Cin, Cout, sh, circ = c
        if sh == 1 and W % 32 == 0 and Cin > 32 and Cout > 64:
            test_wgrad_bf16_matches_autograd(c, 'nhwc', mfma16=True)
            n += 1
    assert n >= 5


@pytest.mark.parametrize('layout', ['nhwc', 'octet'])
@pytest.mark.parametrize('case', WGRAD_CASES)
def test_wgrad_bf16_matches_autograd(case, layout, mfma16=False):
    """Both bf16 weight-gradient kernels -- round 5's NHWC-direct one (pixels as the MFMA's k index, ds_read_b64_tr_b16; the one
    the training step calls) and round 1's batch-octet one -- against CPU autograd through the oracle's conv on bf16-exact operands
    (model/cvig_fov.py:447-460: autograd through torch.nn.Conv2d)."""
    from witw_amd import ops
    B, H, W, Cin, Cout, sh, circ = case
    x = _rand_bf16(1, (B, Cin, H, W))
    w = torch.zeros((Cout, Cin, 3, 3), requires_grad=True)
    b = torch.zeros((Cout,), requires_grad=True)
    y = O.conv3x3(x, w, b, sh, circ)
    gy = _rand_bf16(2, tuple(y.shape))
    y.backward(gy)
    dev = torch.device('cuda:0')
    xd, gyd = _nhwc_bf16(x, dev), _nhwc_bf16(gy, dev)
    oct_ = ops.nhwc_bf16_to_octet(xd)
    assert oct_.shape == ((B + 7) // 8, H, W, Cin, 8)
    back = oct_.permute(0, 4, 1, 2, 3).reshape(-1, H, W, Cin)          # [B8*8,H,W,C]
    assert torch.equal(back[:B], xd) and float(back[B:].float().abs().max() if back.shape[0] > B else 0.) == 0.
    dw, db = ops.conv3x3_wgrad_bf16(xd, gyd, Cin, stride_h=sh, circular=circ, layout=layout)
    if layout == 'nhwc':      # which instantiation ran: stride, rows per stage, wave roles (ci slabs x co slabs; the rest of the 8 waves along k)
        nwm, nwn = (1 if Cin <= 32 and sh == 1 else 2), (1 if Cout <= 32 else 2 if Cout <= 64 else 4)
        want = 'conv3x3_wgrad_bf16_nhwc_kernel<%d,%d,%d,%d>' % (sh, 8 if sh == 1 else 4, nwm, nwn)
        if mfma16 and sh == 1 and W % 32 == 0 and Cin > 32 and Cout > 64:
            want = 'conv3x3_wgrad_bf16_nhwc16_kernel<1,4>'          # the 16x16x32 MFMA form (witw_conv3x3_wgrad_bf16_mfma16(1))
        assert ops.last_kernel_variant() == want, ops.last_kernel_variant()
    assert dw.dtype == torch.float32 and dw.shape == (Cout, Cin, 3, 3)
    np.testing.assert_allclose(dw.cpu().numpy(), w.grad.numpy(), rtol=0, atol=2e-5 * max(1.0, float(w.grad.abs().max())))
    np.testing.assert_allclose(db.cpu().numpy(), b.grad.numpy(), rtol=0, atol=2e-5 * max(1.0, float(b.grad.abs().max())))
    # bitwise reproducible (fixed-order split-K reduction)
    dw2, db2 = ops.conv3x3_wgrad_bf16(xd, gyd, Cin, stride_h=sh, circular=circ, layout=layout)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    # cin_real < Cin (a channel-padded input, as layer 0 of cvig_semantic: 5 of 16): the padded channels are dropped
    if Cin >= 16:
        dwp, _ = ops.conv3x3_wgrad_bf16(xd, gyd, Cin - 3, stride_h=sh, circular=circ, layout=layout, want_bias=False)
        assert dwp.shape == (Cout, Cin - 3, 3, 3) and torch.equal(dwp, dw[:, :Cin - 3])


DGRAD_CASES = [  # B, H (layer input rows), W, Cin, Cout, stride_h, circ
    (2, 16, 64, 64, 128, 1, True),
    (2, 8, 12, 256, 64, 2, True),
    (3, 7, 24, 64, 64, 2, False),
    (2, 4, 64, 64, 16, 1, False),
]


@pytest.mark.parametrize('case', DGRAD_CASES)
def test_dgrad_bf16_gate_dropout_dilation(case):
    """The bf16 forward kernel on the transposed, tap-rotated filter = data gradient; Dropout2d scale and the ReLU gate
    of the previous layer in the epilogue; zero-interleaved rows for the stride-(2,1) layers."""
    from witw_amd import ops
    B, H, W, Cin, Cout, sh, circ = case
    x = _rand_bf16(3, (B, Cin, H, W)).requires_grad_(True)
    w = _rand_bf16(4, (Cout, Cin, 3, 3), 0.05)
    y = O.conv3x3(x, w, torch.zeros(Cout), sh, circ)
    gy = _rand_bf16(5, tuple(y.shape))
    y.backward(gy)
    dev = torch.device('cuda:0')
    gate = _rand_bf16(6, (B, Cin, H, W))
    scale = torch.from_numpy(synth.dropout_scales(8, 0, B, Cin))
    ref = x.grad * scale[:, :, None, None] * (gate > 0).float()
    pt = ops.PackedConvBf16(w.to(dev), None, transpose_flip=True)
    cpad = (Cout + 15) // 16 * 16
    gyd = torch.zeros((B, y.shape[2], W, cpad), dtype=torch.bfloat16, device=dev)
    gyd[..., :Cout] = _nhwc_bf16(gy, dev)
    dx = ops.conv3x3_bf16_fwd(gyd, pt, stride_h=1, circular=circ, relu=False, drop_scale=scale.to(dev),
                              gate=_nhwc_bf16(gate, dev), dilate_h=(sh == 2), out_h=H if sh == 2 else None)
    assert dx.shape == (B, H, W, Cin) and dx.dtype == torch.bfloat16
    got = dx.float().cpu().permute(0, 3, 1, 2).numpy()
    np.testing.assert_allclose(got, ref.bfloat16().float().numpy(), rtol=2 ** -7, atol=2e-5 * float(ref.abs().max()))


def _encoders(precision, w, dev):
    from witw_amd import cvig_fov
    se = cvig_fov.FOV_DSM(False, weights=w).to(dev).train()
    oe = cvig_fov.FOV_DSM(True, weights=w).to(dev).train()
    se.precision = oe.precision = precision
    return se, oe


@pytest.mark.parametrize('ws', [512, 96])
def test_bf16_training_step_vs_fp32_path(ws):
    from witw_amd import cvig_fov
    dev = torch.device('cuda:0')
    B, seed = 12, 31
    w = synth.fov_dsm_weights(seed)
    xo = torch.from_numpy(synth.normalized_images(seed, 1, (B, 3, 128, 512))).to(dev)
    xs = (xo[..., :ws] + 0.3 * torch.from_numpy(synth.normalized_images(seed, 2, (B, 3, 128, ws))).to(dev)).contiguous()
    drops = {t: {i: torch.from_numpy(synth.dropout_scales(seed, 10 * k + i, B, 512)).to(dev) for i in (17, 19, 21)}
             for k, t in enumerate('so')}
    out = {}
    for prec in ('fp32', 'bf16'):
        se, oe = _encoders(prec, w, dev)
        s_emb = se(xs, dropout_scales=drops['s'])
        o_emb = oe(xo, dropout_scales=drops['o'])
        assert s_emb.dtype == torch.float32 and s_emb.shape == (B, 16, 4, ws // 8)
        ori, dist = cvig_fov.match(o_emb, s_emb)
        loss = cvig_fov.triplet_loss(dist)
        loss.backward()
        grads = {('s.' + n): p.grad for n, p in se.named_parameters() if p.grad is not None}
        grads.update({('o.' + n): p.grad for n, p in oe.named_parameters() if p.grad is not None})
        out[prec] = (loss.item(), s_emb.detach(), o_emb.detach(), grads)
    l32, s32, o32, g32 = out['fp32']
    l16, s16, o16, g16 = out['bf16']
    assert abs(l16 - l32) <= 2e-2 * abs(l32), (l16, l32)
    for a, b in ((s16, s32), (o16, o32)):
        assert float((a - b).norm() / b.norm()) < 5e-2
    assert set(g16) == set(g32) and len(g16) == 24
    worst = 0.0
    for name, g in g32.items():
        assert g16[name].dtype == torch.float32 and g16[name].shape == g.shape
        rel = float((g16[name] - g).norm() / (g.norm() + 1e-30))
        worst = max(worst, rel)
        assert rel < 8e-2, (name, rel)
    print('bf16 vs fp32 training step (ws=%d): loss %.6f vs %.6f, worst gradient deviation %.3e of its norm' % (ws, l16, l32, worst))


def test_bf16_training_reduces_the_loss():
    from witw_amd import cvig_fov
    dev = torch.device('cuda:0')
    w = synth.fov_dsm_weights(13)
    se, oe = _encoders('bf16', w, dev)
    opt = cvig_fov.Adam(list(se.parameters()) + list(oe.parameters()), lr=1e-4)
    xo = torch.from_numpy(synth.normalized_images(14, 0, (16, 3, 128, 512))).to(dev)
    xs = (xo + 0.3 * torch.from_numpy(synth.normalized_images(14, 1, (16, 3, 128, 512))).to(dev)).contiguous()
    drops = {i: torch.full((16, 512), 1.0, device=dev) for i in (17, 19, 21)}      # dropout off: deterministic
    losses = []
    for _ in range(8):
        _, d = cvig_fov.match(oe(xo, dropout_scales=drops), se(xs, dropout_scales=drops))
        loss = cvig_fov.triplet_loss(d)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0] - 1e-3, losses
    # master weights stay fp32 and the packed bf16 filters follow them
    assert all(p.dtype == torch.float32 for p in se.parameters())


def test_pool_codes_and_maxpool_backward_bf16():
    """The fused max-pool of the bf16 kernel records torch's arg-max (first maximum in scan order) and
    witw_maxpool2x2_bwd_bf16 routes the gradient there."""
    from witw_amd import ops
    B, H, W, Cin, Cout = 2, 8, 64, 16, 64
    x = _rand_bf16(11, (B, Cin, H, W))
    w = _rand_bf16(12, (Cout, Cin, 3, 3), 0.1)
    b = _rand_bf16(13, (Cout,), 0.1)
    pre = torch.relu(O.conv3x3(x, w, b, 1, True)).requires_grad_(True)
    pooled = torch.nn.functional.max_pool2d(pre, 2, 2)
    gy = _rand_bf16(14, tuple(pooled.shape))
    pooled.backward(gy)
    dev = torch.device('cuda:0')
    pk = ops.PackedConvBf16(w.to(dev), b.to(dev))
    y, code = ops.conv3x3_bf16_fwd(_nhwc_bf16(x, dev), pk, circular=True, relu=True, pool=True, want_pool_code=True)
    assert code.dtype == torch.uint8 and code.shape == y.shape == (B, H // 2, W // 2, Cout)
    dx = ops.maxpool2x2_bwd_bf16(_nhwc_bf16(gy, dev), code, (H, W))
    got = dx.float().cpu().permute(0, 3, 1, 2)
    # where the four pre-activations are distinct the routing must equal torch's; exact ties (all four clipped to 0
    # by the ReLU) may differ in position but carry the same gradient, so compare the 2x2 block sums too
    blk = lambda t: torch.nn.functional.avg_pool2d(t, 2, 2) * 4
    np.testing.assert_array_equal(blk(got).numpy(), blk(pre.grad).numpy())
    distinct = (torch.nn.functional.max_pool2d(pre.detach(), 2, 2) > 0)
    mask = distinct.repeat_interleave(2, 2).repeat_interleave(2, 3)
    np.testing.assert_array_equal((got * mask).numpy(), (pre.grad * mask).numpy())


def test_semantic_bf16_training_step_vs_fp32_path():
    """cvig_semantic (layer 0 trainable): the bf16 backward walks all 13 layers (dgrad through the frozen VGG filters,
    arg-max scatter behind the three fused max-pools, layer-0 wgrad with 5 of 16 padded input channels)."""
    from witw_amd import cvig_fov, cvig_semantic
    dev = torch.device('cuda:0')
    B, seed = 8, 41
    w = synth.fov_dsm_weights(seed, in_channels=5)
    xo = torch.from_numpy(synth.normalized_images(seed, 1, (B, 5, 128, 512))).to(dev)
    xs = (xo + 0.3 * torch.from_numpy(synth.normalized_images(seed, 2, (B, 5, 128, 512))).to(dev)).contiguous()
    drops = {t: {i: torch.from_numpy(synth.dropout_scales(seed, 10 * k + i, B, 512)).to(dev) for i in (17, 19, 21)}
             for k, t in enumerate('so')}
    out = {}
    for prec in ('fp32', 'bf16'):
        se = cvig_semantic.FOV_DSM(False, weights=w).to(dev).train()
        oe = cvig_semantic.FOV_DSM(True, weights=w).to(dev).train()
        se.precision = oe.precision = prec
        ori, dist = cvig_fov.match(oe(xo, dropout_scales=drops['o']), se(xs, dropout_scales=drops['s']))
        loss = cvig_fov.triplet_loss(dist)
        loss.backward()
        grads = {('s.' + n): p.grad for n, p in se.named_parameters() if p.grad is not None}
        grads.update({('o.' + n): p.grad for n, p in oe.named_parameters() if p.grad is not None})
        out[prec] = (loss.item(), grads)
    (l32, g32), (l16, g16) = out['fp32'], out['bf16']
    assert abs(l16 - l32) <= 2e-2 * abs(l32), (l16, l32)
    assert set(g16) == set(g32) and len(g16) == 28           # 7 trainable convs x (weight, bias) x 2 encoders
    worst = ('', 0.0)
    for name, g in g32.items():
        assert g16[name].shape == g.shape
        rel = float((g16[name] - g).norm() / (g.norm() + 1e-30))
        if rel > worst[1]:
            worst = (name, rel)
        # layer 0 sits behind 12 bf16-rounded layers and three arg-max routings (near-ties route differently)
        assert rel < (0.3 if 'features.0.' in name else 8e-2), (name, rel)
    print('semantic bf16 vs fp32 step: loss %.6f vs %.6f, worst gradient deviation %.3e (%s)' % (l16, l32, worst[1], worst[0]))


def test_bf16_training_forward_with_the_fused_first_two_layers_is_bitwise_the_unfused_one():
    """round 4: the bf16 TRAINING forward of cvig_fov (frozen trunk: nothing below layer 17 is kept for the backward) runs layers
    0 and 2 on conv_first2_bf16_kernel, as the inference forward does -- the same bits and the same gradients as the two launches;
    cvig_semantic (layer 0 trainable, model/cvig_semantic.py:301-309) keeps the unfused forward"""
    from witw_amd import cvig_fov, cvig_semantic, ops
    dev = torch.device('cuda:0')
    w = synth.fov_dsm_weights(31)
    x = torch.from_numpy(synth.normalized_images(32, 0, (4, 3, 128, 512))).to(dev)
    drops = {i: torch.full((4, 512), 1.25, device=dev) for i in (17, 19, 21)}
    outs, grads = [], []
    for fuse in (None, False):
        enc = cvig_fov.FOV_DSM(circ_padding=True, weights=w).to(dev).train()
        enc.precision = 'bf16'
        enc.fuse_first2 = fuse
        ran = []
        real = ops.conv_first2_bf16

        def recording(*a, **k):
            ran.append('first2')
            return real(*a, **k)
        ops.conv_first2_bf16 = recording
        try:
            e = enc(x, dropout_scales=drops)
        finally:
            ops.conv_first2_bf16 = real
        assert (ran == ['first2']) == (fuse is None), (fuse, ran)
        e.square().sum().backward()
        outs.append(e.detach().clone())
        grads.append([p.grad.clone() for p in enc.parameters() if p.grad is not None])
    assert torch.equal(outs[0], outs[1])
    assert len(grads[0]) == len(grads[1]) == 12 and all(torch.equal(a, b) for a, b in zip(*grads))
    w5 = synth.fov_dsm_weights(31, in_channels=5)
    enc5 = cvig_semantic.FOV_DSM(circ_padding=True, weights=w5).to(dev).train()
    enc5.precision = 'bf16'
    called = []
    real = ops.conv_first2_bf16
    ops.conv_first2_bf16 = lambda *a, **k: called.append(1) or real(*a, **k)
    try:
        enc5(torch.from_numpy(synth.normalized_images(33, 5, (2, 5, 128, 512))).to(dev))
    finally:
        ops.conv_first2_bf16 = real
    assert not called


def test_semantic_training_step_with_the_fused_first_two_layers_is_bitwise_the_unfused_one():
    """round 6: cvig_semantic's bf16 training step (layer 0 trains, model/cvig_semantic.py:301-309) runs layers 0 and 2 forward on the
    TRAINING form of conv_first2_bf16_kernel from 16 images on (the batch at which layer 2's data gradient runs on the kernel that
    reads one-bit gates): arg-max codes + one gate bit per layer-0 output instead of the two activations. Same embedding bits and
    the same bits in every gradient (layer 0's included) as the unfused step; smaller batches keep the two launches."""
    from witw_amd import cvig_semantic, ops
    dev = torch.device('cuda:0')
    w5 = synth.fov_dsm_weights(31, in_channels=5)
    for B, expect_fused in ((16, True), (2, False)):
        x = torch.from_numpy(synth.normalized_images(34, 5, (B, 5, 128, 512))).to(dev)
        drops = {i: torch.full((B, 512), 1.25, device=dev) for i in (17, 19, 21)}
        outs, grads = [], []
        for fuse in (None, False):
            enc = cvig_semantic.FOV_DSM(circ_padding=True, weights=w5).to(dev).train()
            enc.precision = 'bf16'
            enc.fuse_first2 = fuse
            ran = []
            real_f, real_g = ops.conv_first2_bf16_train, ops.conv3x3_bf16_dgrad_gatebits
            ops.conv_first2_bf16_train = lambda *a, **k: ran.append('first2_train') or real_f(*a, **k)
            ops.conv3x3_bf16_dgrad_gatebits = lambda *a, **k: ran.append('gatebits') or real_g(*a, **k)
            try:
                e = enc(x, dropout_scales=drops)
                e.square().sum().backward()
            finally:
                ops.conv_first2_bf16_train, ops.conv3x3_bf16_dgrad_gatebits = real_f, real_g
            assert (ran == ['first2_train', 'gatebits']) == (fuse is None and expect_fused), (B, fuse, ran)
            outs.append(e.detach().clone())
            grads.append({n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None})
        assert torch.equal(outs[0], outs[1])
        assert len(grads[0]) == len(grads[1]) == 14 and any('features.0.' in n for n in grads[0])
        for n in grads[0]:
            assert torch.equal(grads[0][n], grads[1][n]), (B, n, float((grads[0][n] - grads[1][n]).abs().max()))


def test_batched_filter_packing_equals_one_at_a_time():
    """ops.PackedConvBf16.batch (one launch for all of an encoder's stale filter images + bias copies: what the bf16 training step
    calls after every Adam update) writes the same bits as the constructor image by image, forward and dgrad form, with and
    without a buffer to reuse, across the 16-entries-per-launch boundary."""
    from witw_amd import ops
    dev = torch.device('cuda:0')
    g = np.random.Generator(np.random.Philox(key=[77, 1]))
    shapes = [(512, 256), (512, 512), (256, 512), (64, 256), (16, 64), (64, 5), (128, 64), (24, 40)]
    items, singles = [], []
    for n, (co, ci) in enumerate(shapes * 3):           # 24 entries: two launches
        w = torch.from_numpy(g.standard_normal((co, ci, 3, 3), dtype=np.float32)).to(dev)
        tf = n % 2 == 1
        b = None if tf else torch.from_numpy(g.standard_normal((co,), dtype=np.float32)).to(dev)
        items.append((w, b, tf, None))
        singles.append(ops.PackedConvBf16(w, b, transpose_flip=tf))
    batch = ops.PackedConvBf16.batch(items)
    for one, many in zip(singles, batch):
        assert (one.cout, one.cin, one.cin_pad) == (many.cout, many.cin, many.cin_pad)
        assert torch.equal(one.wpk.view(torch.int16), many.wpk.view(torch.int16)) and torch.equal(one.bias, many.bias)
    # re-packing into the same buffers after the weights changed (the training step's case)
    items2 = []
    for (w, b, tf, _r), pk in zip(items, batch):
        w2 = w * 1.5 + 0.25
        items2.append((w2, None if b is None else b - 1.0, tf, pk))
    again = ops.PackedConvBf16.batch(items2)
    for (w2, b2, tf, _r), pk, old in zip(items2, again, batch):
        assert pk.wpk.data_ptr() == old.wpk.data_ptr() and pk.bias.data_ptr() == old.bias.data_ptr()
        ref = ops.PackedConvBf16(w2, b2, transpose_flip=tf)
        assert torch.equal(ref.wpk.view(torch.int16), pk.wpk.view(torch.int16)) and torch.equal(ref.bias, pk.bias)
