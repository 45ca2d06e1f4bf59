"""bf16 MFMA inference path (BASELINE config 'cvig_semantic ... bf16 MFMA'): parity with a CPU emulation of
'fp32 algorithm + bf16 storage' and the accuracy cost against the fp32 reference goldens.

Tolerances (stated, as bf16 cannot meet the fp32 1e-4): against the bf16-storage emulation the embeddings agree
to 1e-2 of their norm (fp32 summation order moves a few intermediate activations across a bf16 rounding
boundary); against the reference's fp32 embeddings the bf16 path is within 5e-2 of the norm."""
import os

import numpy as np
import pytest
import torch

from oracle import cvig_fov_oracle as O
from witw_amd import synth

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.mark.parametrize('case', [(2, 8, 64, 16, 64, 1, True, True, False), (1, 16, 64, 64, 128, 1, False, True, True),
                                  (2, 16, 24, 32, 256, 2, True, True, False), (1, 12, 99, 16, 64, 1, True, True, True),
                                  (2, 4, 64, 64, 16, 1, True, False, False)])
def test_conv3x3_bf16_vs_emulation(case):
    from witw_amd import ops
    B, H, W, Cin, Cout, sh, circ, relu, pool = case
    g = np.random.Generator(np.random.Philox(key=[7, Cin + Cout]))
    x = torch.from_numpy(g.standard_normal((B, Cin, H, W), dtype=np.float32)).bfloat16().float()
    w = torch.from_numpy((g.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) * (2.0 / (9 * Cin)) ** 0.5)).float()
    b = torch.from_numpy(g.standard_normal((Cout,), dtype=np.float32) * 0.1)
    ref = O.conv3x3(x, w.bfloat16().float(), b, sh, circ)
    if relu:
        ref = torch.relu(ref)
    if pool:
        ref = torch.nn.functional.max_pool2d(ref, 2, 2)
    dev = torch.device('cuda:0')
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().bfloat16()
    pk = ops.PackedConvBf16(w.to(dev), b.to(dev))
    last = Cout % 16 != 0 or Cout == 16
    y = ops.conv3x3_bf16_fwd(xd, pk, stride_h=sh, circular=circ, relu=relu, pool=pool, out_nchw_f32=last)
    if last:
        np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-5)           # fp32 out: exact products
    else:
        got = y.float().cpu().permute(0, 3, 1, 2)
        np.testing.assert_allclose(got.numpy(), ref.bfloat16().float().numpy(), rtol=2 ** -7, atol=1e-6)  # <= 1 bf16 ulp


@pytest.mark.parametrize('case', [(2, 3, 16, 64, True), (1, 3, 13, 99, False), (2, 4, 8, 130, True), (1, 1, 9, 70, False)])
def test_first_layer_bf16_vs_emulation(case):
    """C<=4 -> 64 first layer on the bf16 MFMA (two taps of a pixel per operand): bf16-rounded image and filter,
    fp32 accumulate, bf16 NHWC out -> within one bf16 ulp of the CPU emulation."""
    from witw_amd import ops
    B, C, H, W, circ = case
    g = np.random.Generator(np.random.Philox(key=[9, C * 1000 + W]))
    x = torch.from_numpy(g.standard_normal((B, C, H, W), dtype=np.float32))
    w = torch.from_numpy(g.standard_normal((64, C, 3, 3), dtype=np.float32) * (2.0 / (9 * C)) ** 0.5)
    b = torch.from_numpy(g.standard_normal((64,), dtype=np.float32) * 0.1)
    ref = torch.relu(O.conv3x3(x.bfloat16().float(), w.bfloat16().float(), b, 1, circ))
    dev = torch.device('cuda:0')
    pk = ops.PackedFirstConv(w.to(dev), b.to(dev), bf16=True)
    y = ops.conv3x3_first_fwd(x.to(dev), pk, circular=circ, relu=True)
    assert y.dtype == torch.bfloat16 and y.shape == (B, H, W, 64)
    got = y.float().cpu().permute(0, 3, 1, 2)
    np.testing.assert_allclose(got.numpy(), ref.bfloat16().float().numpy(), rtol=2 ** -7, atol=1e-6)


def test_encoder_bf16_vs_emulation_and_fp32_goldens(golden_dir):
    from witw_amd import cvig_fov, cvig_semantic
    g = np.load(os.path.join(golden_dir, 'encoder.npz'))
    seed = int(g['seed'])
    w = synth.fov_dsm_weights(seed)
    wt = {k: (torch.from_numpy(a), torch.from_numpy(b)) for k, (a, b) in w.items()}
    x360 = torch.from_numpy(synth.normalized_images(seed, 10, (2, 3, 128, 512)))
    for circ in (False, True):
        enc = cvig_fov.FOV_DSM(circ_padding=circ, weights=w).cuda().eval()
        e = enc.forward_bf16(x360.cuda()).cpu().numpy()
        assert e.shape == (2, 16, 4, 64) and e.dtype == np.float32
        with torch.no_grad():
            emu = O.fov_dsm_forward_bf16_emulated(x360, wt, circ).numpy()
        assert _rel(e, emu) < 1e-2, _rel(e, emu)
        assert _rel(e, g['embed360_circ%d' % circ]) < 5e-2
    # 5-channel semantic variant on the same kernels
    gs = np.load(os.path.join(golden_dir, 'encoder_semantic.npz'))
    w5 = synth.fov_dsm_weights(seed, in_channels=5)
    x5 = torch.from_numpy(synth.normalized_images(seed, 12, (1, 5, 128, 512)))
    e5 = cvig_semantic.FOV_DSM(circ_padding=True, weights=w5).cuda().eval().forward_bf16(x5.cuda()).cpu().numpy()
    assert _rel(e5, gs['embed5_circ1']) < 5e-2
    with pytest.raises(Exception):
        enc.train().forward_bf16(x360.cuda())


@pytest.mark.parametrize('shape', [(2, 3, 128, 512), (3, 5, 128, 99), (2, 3, 20, 70), (1, 5, 9, 33), (2, 4, 64, 32), (1, 8, 16, 130)])
def test_fused_first_two_layers_equal_the_separate_launches(shape):
    """csrc/conv_first2_bf16.hip (layers 0 and 2 in one persistent kernel, layer 0 recomputed on each tile's halo, the
    64-channel map between them never written) against conv3x3_first_fwd + conv3x3_bf16_fwd(pool): the same bits, both
    padding modes, 3- / 5-channel inputs, sizes that are not multiples of the tile or of the pooling window."""
    from witw_amd import ops
    B, C, H, W = shape
    g = np.random.Generator(np.random.Philox(key=[81, H * W + C]))
    x = torch.from_numpy(g.standard_normal((B, C, H, W), dtype=np.float32)).cuda()
    w0 = torch.from_numpy((g.standard_normal((64, C, 3, 3), dtype=np.float32) * 0.3).astype(np.float32)).cuda()
    b0 = torch.from_numpy((g.standard_normal((64,), dtype=np.float32) * 0.2).astype(np.float32)).cuda()
    w2 = torch.from_numpy((g.standard_normal((64, 64, 3, 3), dtype=np.float32) * 0.06).astype(np.float32)).cuda()
    b2 = torch.from_numpy((g.standard_normal((64,), dtype=np.float32) * 0.2).astype(np.float32)).cuda()
    pf, p2 = ops.PackedFirstConv(w0, b0, bf16=True), ops.PackedConvBf16(w2, b2)
    for circ in (False, True):
        mid = ops.conv3x3_first_fwd(x, pf, circular=circ, relu=True)
        ref = ops.conv3x3_bf16_fwd(mid, p2, circular=circ, relu=True, pool=True)
        got = ops.conv_first2_bf16(x, pf, p2, circular=circ)
        assert got.shape == ref.shape == (B, H // 2, W // 2, 64)
        assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), (shape, circ, float((got.float() - ref.float()).abs().max()))


@pytest.mark.parametrize('shape', [(2, 5, 128, 512), (3, 5, 128, 98), (2, 3, 20, 70), (1, 5, 10, 34), (2, 4, 64, 32), (1, 8, 16, 130)])
def test_fused_first_two_layers_training_form(shape):
    """The training form of csrc/conv_first2_bf16.hip (round 6: cvig_semantic trains layer 0, model/cvig_semantic.py:301-309, so the
    backward crosses both layers): the same y bits as the inference form, the max-pool arg-max codes of the two-launch path
    (conv3x3_bf16_fwd(pool, want_pool_code)) and layer 0's ReLU gate as one bit per output = (layer-0 activation > 0), both padding
    modes, sizes that are not multiples of the tile."""
    from witw_amd import _lib, ops
    B, C, H, W = shape
    g = np.random.Generator(np.random.Philox(key=[82, H * W + C]))
    x = torch.from_numpy(g.standard_normal((B, C, H, W), dtype=np.float32)).cuda()
    w0 = torch.from_numpy((g.standard_normal((64, C, 3, 3), dtype=np.float32) * 0.3).astype(np.float32)).cuda()
    b0 = torch.from_numpy((g.standard_normal((64,), dtype=np.float32) * 0.2).astype(np.float32)).cuda()
    w2 = torch.from_numpy((g.standard_normal((64, 64, 3, 3), dtype=np.float32) * 0.06).astype(np.float32)).cuda()
    b2 = torch.from_numpy((g.standard_normal((64,), dtype=np.float32) * 0.2).astype(np.float32)).cuda()
    pf, p2 = ops.PackedFirstConv(w0, b0, bf16=True), ops.PackedConvBf16(w2, b2)
    for circ in (False, True):
        mid = ops.conv3x3_first_fwd(x, pf, circular=circ, relu=True)                      # [B,H,W,64] bf16
        ref, ref_code = ops.conv3x3_bf16_fwd(mid, p2, circular=circ, relu=True, pool=True, want_pool_code=True)
        y, code, bits = ops.conv_first2_bf16_train(x, pf, p2, circular=circ)
        assert ops.last_kernel_variant() == 'conv_first2_bf16_kernel<%d,train>' % (4 if C <= 4 else 8)
        assert torch.equal(y.view(torch.int16), ref.view(torch.int16)), (shape, circ)
        assert torch.equal(y.view(torch.int16), ops.conv_first2_bf16(x, pf, p2, circular=circ).view(torch.int16))
        assert torch.equal(code, ref_code), (shape, circ, int((code != ref_code).sum()))
        want = (mid.float() > 0).reshape(B, H, W, 8, 8)                                    # channel c = 8 * byte + bit
        want = (want.to(torch.int32) << torch.arange(8, device=x.device, dtype=torch.int32)).sum(-1).to(torch.uint8)
        assert bits.shape == (B, H, W, 8) and torch.equal(bits, want), (shape, circ, int((bits != want).sum()))
    with pytest.raises(_lib.WitwError):
        ops.conv_first2_bf16_train(x[:, :, :H - 1], pf, p2)                                # odd height


@pytest.mark.parametrize('circ', [False, True])
def test_gate_bits_dgrad_equals_the_tensor_gate(circ):
    """conv3x3_bf16_wres_kernel<gate_bits> (round 6: the data gradient of cvig_semantic's layer 2 with layer 0's ReLU gate as one bit
    per output) against the same launch with the bf16 activation as the gate: the same bits, both on the weight-resident kernel."""
    from witw_amd import _lib, ops
    B, H, W = 16, 128, 512
    assert ops.gatebits_dgrad_ok(B, H, W, 64, 64) and not ops.gatebits_dgrad_ok(2, H, W, 64, 64)
    g = np.random.Generator(np.random.Philox(key=[83, int(circ)]))
    dev = torch.device('cuda:0')
    dz = torch.from_numpy(g.standard_normal((B, H, W, 64), dtype=np.float32)).to(dev).bfloat16()
    act = torch.from_numpy(g.standard_normal((B, H, W, 64), dtype=np.float32)).to(dev).clamp_min(0).bfloat16()      # a post-ReLU map: half zeros
    w2 = torch.from_numpy((g.standard_normal((64, 64, 3, 3), dtype=np.float32) * 0.06).astype(np.float32)).to(dev)
    pt = ops.PackedConvBf16(w2, None, transpose_flip=True)
    ref = ops.conv3x3_bf16_fwd(dz, pt, circular=circ, relu=False, gate=act)
    assert ops.last_kernel_variant() == 'conv3x3_bf16_wres_kernel<gate>'
    bits = ((act.float() > 0).reshape(B, H, W, 8, 8).to(torch.int32) << torch.arange(8, device=dev, dtype=torch.int32)).sum(-1).to(torch.uint8)
    got = ops.conv3x3_bf16_dgrad_gatebits(dz, pt, bits.contiguous(), circular=circ)
    assert ops.last_kernel_variant() == 'conv3x3_bf16_wres_kernel<gate_bits>'
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), float((got.float() - ref.float()).abs().max())
    assert float(ref.float().abs().max()) > 0 and float((ref == 0).float().mean()) > 0.4
    with pytest.raises(_lib.WitwError):
        ops.conv3x3_bf16_dgrad_gatebits(dz[:2].contiguous(), pt, bits[:2].contiguous(), circular=circ)      # too small for the weight-resident kernel


def test_encoder_bf16_with_and_without_the_fused_launch():
    from witw_amd import cvig_fov, cvig_semantic
    for mod, c in ((cvig_fov, 3), (cvig_semantic, 5)):
        w = synth.fov_dsm_weights(5, in_channels=c)
        x = torch.from_numpy(synth.normalized_images(5, c, (2, c, 128, 512))).cuda()
        enc = mod.FOV_DSM(circ_padding=True, weights=w).cuda().eval()
        a = enc.forward_bf16(x)
        enc.fuse_first2 = False
        b = enc.forward_bf16(x)
        assert torch.equal(a, b)


@pytest.mark.parametrize('case', [(32, 32, 128, 128, 256, False, False), (32, 32, 128, 128, 256, True, True), (64, 16, 64, 512, 512, False, True),
                                  (40, 64, 200, 64, 128, True, False), (24, 40, 70, 256, 384, False, False), (32, 32, 128, 96, 256, False, True),
                                  (32, 32, 128, 160, 128, True, False), (33, 24, 130, 256, 128, False, False),
                                  (32, 32, 128, 48, 256, False, False)])
def test_bf16_mfma16_kernel_matches_the_32x32_kernel(case):
    """The 16x16x32 form of the bf16 inference forward (two taps of a 16-channel chunk per MFMA, the ninth tap of an even chunk
    sharing its MFMA with the ninth tap of the next chunk) against the 32x32x16 kernel on layers large enough for the 8-wave tile:
    the same products summed in fp32 in a different order, so outputs agree to one bf16 unit in the last place and almost all
    are identical; ragged widths, circular and zero padding, the fused pool, odd and even numbers of chunk pairs. The last two cases do not qualify (297 workgroups =
    1.16 rounds of the 256 CUs; Cin % 32 != 0) and must be bit-identical whatever the switch says."""
    import torch
    from witw_amd import ops
    B, H, W, cin, cout, pool, circ = case
    g = torch.Generator(device='cuda')
    g.manual_seed(cin + cout + W)
    x = torch.randn((B, H, W, cin), generator=g, device='cuda').bfloat16()
    w = torch.randn((cout, cin, 3, 3), generator=g, device='cuda') * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn((cout,), generator=g, device='cuda') * 0.1
    pk = ops.PackedConvBf16(w, b)
    prev = ops.bf16_mfma16(False)
    try:
        y32 = ops.conv3x3_bf16_fwd(x, pk, circular=circ, relu=True, pool=pool)
        assert ops.bf16_mfma16(True) is False
        y16 = ops.conv3x3_bf16_fwd(x, pk, circular=circ, relu=True, pool=pool)
    finally:
        ops.bf16_mfma16(prev)
    # the launcher's rule (csrc/api.hip witw_fills_rounds): 8-wave tiles from two rounds of one workgroup per CU on, or when the last
    # round is at least 90 % full (e.g. exactly 256 workgroups: the 16 x 64 maps at the reference's default batch of 32)
    cu = torch.cuda.get_device_properties(0).multi_processor_count
    big = ((cout + 127) // 128) * B * ((W + 63) // 64) * (H // 8)
    fills = big >= 2 * cu or 10 * big >= 9 * cu * ((big + cu - 1) // cu)
    qualifies = cin % 32 == 0 and cout >= 128 and H % 8 == 0 and fills
    a, c = y32.float(), y16.float()
    if not qualifies:
        assert torch.equal(a, c)
        return
    same = float((a == c).float().mean())
    assert same > 0.999, same
    # one unit in the last place of a bf16 number: 2^-7 relative (2^-8 mantissa step of the larger neighbour, doubled at a binade edge);
    # next to zero (a pre-activation within fp32 summation rounding of the ReLU kink) the difference is that rounding itself
    scale = float(a.abs().max())
    assert bool(((a - c).abs() <= 2.0 ** -7 * torch.maximum(a.abs(), c.abs()) + 1e-5 * scale).all())
    assert not torch.equal(a, c) or cin <= 64          # it really is the other kernel (a different summation order shows somewhere)


@pytest.mark.parametrize('shape', [(3, 5, 33, 70, 16), (2, 3, 128, 512, 16), (4, 16, 9, 17, 16), (2, 21, 8, 24, 32), (1, 1, 5, 3, 16)])
def test_nchw_f32_to_nhwc_bf16_is_exact(shape):
    """witw_nchw_f32_to_nhwc_bf16 (cvig_semantic's training step converts its 5-channel inputs with it; round 5: one thread per
    pixel and 8-channel group): round-to-nearest-even bf16 of every value in its NHWC place, zeros in the padding channels."""
    from witw_amd import ops
    B, C, H, W, cp = shape
    g = np.random.Generator(np.random.Philox(key=[55, B * C * H * W]))
    x = torch.from_numpy(g.standard_normal((B, C, H, W), dtype=np.float32) * 3.0)
    y = ops.nchw_to_nhwc_bf16(x.cuda(), cp).cpu()
    assert y.shape == (B, H, W, cp) and y.dtype == torch.bfloat16
    assert torch.equal(y[..., :C], x.permute(0, 2, 3, 1).bfloat16())
    assert float(y[..., C:].float().abs().max() if cp > C else 0.) == 0.
