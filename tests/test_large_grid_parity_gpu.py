"""Large-grid kernel variants against the ORACLE (not against another HIP kernel).

The conv launchers pick a different instantiation from 512 workgroups on, and (round 4, csrc/api.hip witw_fills_rounds) whenever the
last round of one workgroup per CU is at least 90 % full, e.g. exactly 256 workgroups: the 16 x 64 maps at the reference's default
batch of 32 (model/cvig_semantic.py:416) (8-wave tiles; for the bf16 inference forward the 16x16x32-MFMA kernel
conv3x3_bf16_s16_kernel). Batches of 1-12 images never get there, so these cases use the batch sizes the
benchmarks run (BASELINE configs[1] / configs[3]: 128 images, layers of model/cvig_fov.py:256-272 and
model/cvig_semantic.py:275-325), compare a spread of images of the batch with the CPU oracle (images are independent, the
kernel still sees the whole batch) and ASSERT WHICH KERNEL RAN through witw_last_kernel_variant(), so a change of the launcher's
thresholds cannot silently move a case back onto the small-grid kernel.

Tolerances: bf16 outputs within one bf16 unit in the last place of the oracle's conv on bf16-rounded operands (fp32
accumulation, different summation order); fp32 kernels 3e-5 of the output scale; fp16x3 5e-6 against fp64."""
import numpy as np
import pytest
import torch

from oracle import cvig_fov_oracle as O
from witw_amd import synth

pytestmark = pytest.mark.gpu


def _layer(seed, B, H, W, cin, cout):
    g = np.random.Generator(np.random.Philox(key=[seed, cin * 1000 + cout]))
    x = torch.from_numpy(g.standard_normal((B, cin, H, W), dtype=np.float32))
    w = torch.from_numpy(g.standard_normal((cout, cin, 3, 3), dtype=np.float32) * (2.0 / (9 * cin)) ** 0.5)
    b = torch.from_numpy(g.standard_normal((cout,), dtype=np.float32) * 0.1)
    return x, w, b


def _pick(B):
    """images of the batch the oracle is run on: both ends, the middle, one odd index (different XCDs / workgroup ranges)"""
    return sorted({0, B // 3, B // 2 + 1, B - 1})


def _ref(x, w, b, sh, circ, relu, pool):
    r = O.conv3x3(x, w, b, sh, circ)
    if relu:
        r = torch.relu(r)
    if pool:
        r = torch.nn.functional.max_pool2d(r, 2, 2)
    return r


def _assert_bf16_ulp(got, ref):
    r16 = ref.bfloat16().float()
    # one bf16 unit in the last place, and next to zero the rounding of the fp32 sum itself (thousands of products of size ~1e-2:
    # the CPU's fp32 conv is 2e-6 from the fp64 sum on these layers, tools/debug/bf16_s2.py), which is many units of a 3e-5 output
    bad = (got - r16).abs() > 2.0 ** -7 * torch.maximum(r16.abs(), got.abs()) + 1e-5 * float(r16.abs().max())
    # an fp32 sum that lands within rounding of a bf16 tie may round to the neighbour: still one unit in the last place
    assert not bool(bad.any()), (int(bad.sum()), float((got - r16).abs().max()))
    assert float((got == r16).float().mean()) > 0.97


# (B, H, W, Cin, Cout, stride_h, circular, pool, mfma16 switch, expected kernel)
BF16_CASES = [
    # the config-4 dominant kernel on the layer shapes the bench runs (layers 10/12, 17/19/21, 5, 7 of the encoder)
    (32, 32, 128, 128, 256, 1, False, False, True, 'conv3x3_bf16_s16_kernel<false,false>'),
    (32, 32, 128, 256, 256, 1, True, True, True, 'conv3x3_bf16_s16_kernel<true,false>'),
    (64, 16, 64, 512, 512, 1, True, False, True, 'conv3x3_bf16_s16_kernel<false,false>'),
    (64, 16, 64, 256, 512, 1, False, False, True, 'conv3x3_bf16_s16_kernel<false,false>'),
    (16, 64, 256, 64, 128, 1, True, False, True, 'conv3x3_bf16_s16_kernel<false,false>'),
    (16, 64, 256, 128, 128, 1, False, True, True, 'conv3x3_bf16_s16_kernel<true,false>'),
    (44, 24, 100, 96, 144, 1, True, False, True, 'conv3x3_bf16_s16_kernel<false,false>'),      # ragged width / channels, odd chunk pairs
    # round 4: exactly ONE round of workgroups (4 channel tiles x 32 images x 2 row tiles = 256 = one per CU): layers 17-21 at the
    # reference's default batch of 32, and layer 23 (stride (2,1)) at the bench batch
    (32, 16, 64, 512, 512, 1, True, False, True, 'conv3x3_bf16_s16_kernel<false,false>'),
    (32, 16, 64, 256, 512, 1, False, False, True, 'conv3x3_bf16_s16_kernel<false,false>'),
    (128, 16, 64, 512, 256, 2, True, False, True, 'conv3x3_nhwc_bf16_kernel<128,2,false,8>'),
    (30, 16, 64, 512, 512, 1, True, False, True, 'conv3x3_bf16_s16_kernel<false,false>'),        # 240 workgroups: 94 % of a round
    (27, 16, 64, 512, 512, 1, True, False, True, 'conv3x3_nhwc_bf16_kernel<128,1,false,4>'),    # 216: 84 % -> the 4-wave tiles
    # the 8-wave 32x32x16 kernel: with the switch off, and where the 16x16x32 kernel does not apply
    (32, 32, 128, 128, 256, 1, True, False, False, 'conv3x3_nhwc_bf16_kernel<128,1,false,8>'),
    (32, 32, 128, 128, 256, 1, False, True, False, 'conv3x3_nhwc_bf16_kernel<128,1,true,8>'),
    (32, 32, 128, 48, 256, 1, False, False, True, 'conv3x3_nhwc_bf16_kernel<128,1,false,8>'),     # Cin % 32 != 0
    (128, 32, 64, 512, 256, 2, True, False, True, 'conv3x3_nhwc_bf16_kernel<128,2,false,8>'),     # stride (2,1) as layer 23
    (16, 64, 256, 64, 64, 1, False, True, True, 'conv3x3_nhwc_bf16_kernel<64,1,true,8>'),         # layer 2 shape at half size
    (16, 64, 256, 64, 64, 1, True, False, True, 'conv3x3_nhwc_bf16_kernel<64,1,false,8>'),
]


@pytest.mark.parametrize('case', BF16_CASES)
def test_bf16_large_grid_kernels_vs_oracle(case):
    from witw_amd import ops
    B, H, W, cin, cout, sh, circ, pool, s16, want = case
    x, w, b = _layer(31, B, H, W, cin, cout)
    x = x.bfloat16().float()
    dev = torch.device('cuda:0')
    pk = ops.PackedConvBf16(w.to(dev), b.to(dev))
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().bfloat16()
    prev = ops.bf16_mfma16(s16)
    try:
        y = ops.conv3x3_bf16_fwd(xd, pk, stride_h=sh, circular=circ, relu=True, pool=pool)
        assert ops.last_kernel_variant() == want, ops.last_kernel_variant()
    finally:
        ops.bf16_mfma16(prev)
    sel = _pick(B)
    ref = _ref(x[sel], w.bfloat16().float(), b, sh, circ, True, pool)
    _assert_bf16_ulp(y[sel].float().cpu().permute(0, 3, 1, 2), ref)


# (B, H, W, Cout, circular, relu): 64 input channels, >= 4096 (tile, channel block) units -> the weight-resident kernel
WRES_CASES = [(64, 64, 256, 128, True, True), (32, 64, 256, 128, False, True), (64, 64, 256, 64, True, False), (22, 64, 256, 192, True, True),
              (128, 32, 128, 128, False, True),
              # round 4 (two teams, LDS-DMA input): ONE 8 x 16 tile per image -- top and bottom rows padded, the left and right halo
              # columns both wrapping onto the image itself -- and three tiles per row with an odd number of tile pairs per walker
              (4096, 8, 16, 128, True, True), (4099, 8, 16, 128, False, True), (1400, 16, 48, 64, True, False)]


@pytest.mark.parametrize('case', [(8200, 8, 16, 64, True), (4099, 8, 16, 128, False), (2800, 16, 48, 64, True), (128, 128, 512, 64, True)])
def test_bf16_weight_resident_kernel_gated_form_bitwise_vs_tiled_kernel(case):
    """The dgrad form of conv3x3_bf16_wres_kernel (round 5: the ReLU gate of the previous layer applied to the 16-byte output octets;
    cvig_semantic's layer-2 data gradient at 128 x 512, where layer 0 trains): BITWISE the tiled kernel's gated launch on the whole
    batch -- same products, same accumulation order, and a gate is open exactly where the bf16 gate value is > 0 (zeros, negative
    values and -0 close it)."""
    from witw_amd import ops
    B, H, W, cout, circ = case
    if B * H * W * max(64, cout) * 2 > 3 * 2 ** 30:
        pytest.skip('too large')
    x, w, b = _layer(41, B if H < 100 else 8, H, W, 64, cout)
    if H >= 100:      # the cvig_semantic shape: 8 distinct images repeated to the bench batch
        x = x.repeat(B // 8, 1, 1, 1)
    dev = torch.device('cuda:0')
    pk = ops.PackedConvBf16(w.to(dev), torch.zeros_like(b).to(dev))
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().bfloat16()
    gen = torch.Generator(device='cuda')
    gen.manual_seed(7)
    gate = torch.randn((xd.shape[0], H, W, cout), generator=gen, device=dev).clamp_(min=0).bfloat16()      # a ReLU output: half zeros
    gate[0, 0, 0, :8] = torch.tensor([0.0, -0.0, -1.0, 1e-38, 1.0, float('inf'), -float('inf'), 3.0]).bfloat16().to(dev)
    assert ops.bf16_wres() is True
    y = ops.conv3x3_bf16_fwd(xd, pk, circular=circ, relu=False, gate=gate)
    assert ops.last_kernel_variant() == 'conv3x3_bf16_wres_kernel<gate>', ops.last_kernel_variant()
    prev_w, prev_s = ops.bf16_wres(False), ops.bf16_mfma16(False)      # the 32x32x16 tiled kernel: the same accumulation order
    try:
        tiled = ops.conv3x3_bf16_fwd(xd, pk, circular=circ, relu=False, gate=gate)
        assert ops.last_kernel_variant().startswith('conv3x3_nhwc_bf16_kernel<'), ops.last_kernel_variant()
    finally:
        ops.bf16_wres(prev_w)
        ops.bf16_mfma16(prev_s)
    assert torch.equal(y.view(torch.int16), tiled.view(torch.int16))
    closed = ~(gate.float() > 0)
    assert bool((y.float()[closed] == 0).all()) and float(y.float().abs().max()) > 0


@pytest.mark.parametrize('case', WRES_CASES)
def test_bf16_weight_resident_kernel_vs_oracle_and_tiled_kernel(case):
    """conv3x3_bf16_wres_kernel (layer 5 of the trunk at bench batch sizes: the filter block stays in LDS, persistent workgroups)
    against the oracle at one bf16 unit in the last place on a spread of images, and BITWISE against the 32x32x16 tiled kernel on
    the whole batch (same products, same fp32 accumulation order)."""
    from witw_amd import ops
    B, H, W, cout, circ, relu = case
    x, w, b = _layer(35, B, H, W, 64, cout)
    x = x.bfloat16().float()
    dev = torch.device('cuda:0')
    pk = ops.PackedConvBf16(w.to(dev), b.to(dev))
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().bfloat16()
    assert ops.bf16_wres() is True
    y = ops.conv3x3_bf16_fwd(xd, pk, circular=circ, relu=relu)
    assert ops.last_kernel_variant() == 'conv3x3_bf16_wres_kernel', ops.last_kernel_variant()
    prev_w, prev_s = ops.bf16_wres(False), ops.bf16_mfma16(False)
    try:
        tiled = ops.conv3x3_bf16_fwd(xd, pk, circular=circ, relu=relu)
        assert ops.last_kernel_variant().startswith('conv3x3_nhwc_bf16_kernel<'), ops.last_kernel_variant()
    finally:
        ops.bf16_wres(prev_w)
        ops.bf16_mfma16(prev_s)
    assert torch.equal(y.view(torch.int16), tiled.view(torch.int16))
    sel = _pick(B)
    ref = _ref(x[sel], w.bfloat16().float(), b, 1, circ, relu, False)
    _assert_bf16_ulp(y[sel].float().cpu().permute(0, 3, 1, 2), ref)


def test_bf16_large_grid_fp32_nchw_output_vs_oracle():
    """the embedding layer's form (fp32 NCHW out, no ReLU) on an 8-wave grid: exact products, fp32 sums"""
    from witw_amd import ops
    B, H, W, cin, cout = 512, 8, 64, 64, 128
    x, w, b = _layer(33, B, H, W, cin, cout)
    x = x.bfloat16().float()
    dev = torch.device('cuda:0')
    y = ops.conv3x3_bf16_fwd(x.to(dev).permute(0, 2, 3, 1).contiguous().bfloat16(), ops.PackedConvBf16(w.to(dev), b.to(dev)),
                             circular=True, relu=False, out_nchw_f32=True)
    assert ops.last_kernel_variant() == 'conv3x3_nhwc_bf16_kernel<128,1,false,8>', ops.last_kernel_variant()
    sel = _pick(B)
    ref = _ref(x[sel], w.bfloat16().float(), b, 1, True, False, False)
    np.testing.assert_allclose(y[sel].cpu().numpy(), ref.numpy(), rtol=0, atol=2e-5 * float(ref.abs().max()))


F32_CASES = [
    (32, 32, 128, 128, 256, 1, True, False, 'conv3x3_nhwc_f32_kernel<128,1,false,8,0,9>'),      # the headline's dominant kernel
    (64, 16, 64, 256, 512, 1, False, False, 'conv3x3_nhwc_f32_kernel<128,1,false,8,0,9>'),
    (16, 64, 256, 64, 128, 1, False, True, 'conv3x3_nhwc_f32_kernel<128,1,true,8,0,9>'),
    (16, 64, 256, 64, 64, 1, True, True, 'conv3x3_nhwc_f32_kernel<64,1,true,8,0,9>'),
    (128, 32, 64, 128, 256, 2, True, False, 'conv3x3_nhwc_f32_kernel<128,2,false,8,0,9>'),
    # round 4: one full round of 8-wave workgroups (the 16 x 64 maps at a batch of 32), and a grid just under it
    (32, 16, 64, 256, 512, 1, True, False, 'conv3x3_nhwc_f32_kernel<128,1,false,8,0,9>'),
    (128, 16, 64, 256, 256, 2, False, False, 'conv3x3_nhwc_f32_kernel<128,2,false,8,0,9>'),
    (27, 16, 64, 256, 512, 1, True, False, 'conv3x3_nhwc_f32_kernel<128,1,false,4,0,9>'),
]


@pytest.mark.parametrize('case', F32_CASES)
def test_f32_large_grid_kernels_vs_oracle(case):
    from witw_amd import ops
    B, H, W, cin, cout, sh, circ, pool, want = case
    x, w, b = _layer(35, B, H, W, cin, cout)
    dev = torch.device('cuda:0')
    y = ops.conv3x3_fwd(x.to(dev).permute(0, 2, 3, 1).contiguous(), ops.PackedConv(w.to(dev), b.to(dev)), stride_h=sh,
                        circular=circ, relu=True, pool=pool)
    assert ops.last_kernel_variant() == want, ops.last_kernel_variant()
    sel = _pick(B)
    ref = _ref(x[sel], w, b, sh, circ, True, pool)
    np.testing.assert_allclose(y[sel].cpu().permute(0, 3, 1, 2).numpy(), ref.numpy(), rtol=0, atol=3e-5 * float(ref.abs().max()))


F16X3_CASES = [
    (32, 32, 128, 128, 256, 1, True, False, 'conv3x3_nhwc_f16x3_kernel<128,1,false,8>'),
    (64, 16, 64, 256, 512, 1, False, False, 'conv3x3_nhwc_f16x3_kernel<128,1,false,8>'),
    (16, 64, 256, 64, 128, 1, False, True, 'conv3x3_nhwc_f16x3_kernel<128,1,true,8>'),
    (16, 64, 256, 64, 64, 1, True, True, 'conv3x3_nhwc_f16x3_kernel<64,1,true,8>'),
]


@pytest.mark.parametrize('case', F16X3_CASES)
def test_f16x3_large_grid_kernels_vs_fp64_oracle(case):
    from witw_amd import ops
    B, H, W, cin, cout, sh, circ, pool, want = case
    x, w, b = _layer(37, B, H, W, cin, cout)
    dev = torch.device('cuda:0')
    y = ops.conv3x3_f16x3_fwd(ops.nchw_to_split_f16(x.to(dev), cin), ops.PackedConvF16x3(w.to(dev), b.to(dev)), stride_h=sh,
                              circular=circ, relu=True, pool=pool)
    assert ops.last_kernel_variant() == want, ops.last_kernel_variant()
    sel = _pick(B)
    ref = _ref(x[sel].double(), w.double(), b.double(), sh, circ, True, pool).float()
    got = ops.split_f16_to_f32(y[sel].contiguous()).cpu().permute(0, 3, 1, 2)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=5e-6 * max(1.0, float(ref.abs().max())))
    assert not ops.f16x3_overflowed()


def test_taps4_large_grid_kernel_vs_oracle():
    """cvig_baseline's Conv2d(4,2,0) (model/cvig_baseline.py:236-252) as the 4-tap conv over space-to-depth channels, 8-wave
    tile: against torch's conv2d with the 3x3 filter whose first row / column are zero."""
    from witw_amd import ops
    B, H, W, cin, cout = 128, 32, 64, 64, 128
    x, w, b = _layer(39, B, H, W, cin, cout)
    w[:, :, 0, :] = 0
    w[:, :, :, 0] = 0
    dev = torch.device('cuda:0')
    y = ops.conv3x3_fwd(x.to(dev).permute(0, 2, 3, 1).contiguous(), ops.PackedConv(w.to(dev), b.to(dev), taps4=True), relu=True)
    assert ops.last_kernel_variant() == 'conv3x3_nhwc_f32_kernel<128,1,false,8,0,4>', ops.last_kernel_variant()
    sel = _pick(B)
    ref = _ref(x[sel], w, b, 1, False, True, False)
    np.testing.assert_allclose(y[sel].cpu().permute(0, 3, 1, 2).numpy(), ref.numpy(), rtol=0, atol=3e-5 * float(ref.abs().max()))


@pytest.mark.parametrize('variant', ['semantic', 'fov'])
def test_bf16_encoder_at_bench_batch_vs_emulation(variant, monkeypatch):
    """The bf16 encoder at the batch size of BASELINE configs[3] (128: every layer from 5 to 21 on the large-grid kernels, as
    in bench.py's config4_semantic_bf16 block) against the CPU emulation of 'fp32 algorithm + bf16 storage' on three images of
    that batch, at the 1e-2 of tests/test_bf16_gpu.py. The kernels that ran are recorded per layer and asserted."""
    from witw_amd import cvig_fov, cvig_semantic, ops
    mod, c = (cvig_semantic, 5) if variant == 'semantic' else (cvig_fov, 3)
    B, seed = 128, 21
    w = synth.fov_dsm_weights(seed, in_channels=c)
    wt = {k: (torch.from_numpy(a), torch.from_numpy(b)) for k, (a, b) in w.items()}
    x = torch.from_numpy(synth.normalized_images(seed, 40 + c, (B, c, 128, 512)))
    ran = []
    real = ops.conv3x3_bf16_fwd

    def recording(*a, **k):
        out = real(*a, **k)
        ran.append(ops.last_kernel_variant())
        return out
    monkeypatch.setattr(ops, 'conv3x3_bf16_fwd', recording)
    for circ in (True, False):
        del ran[:]
        enc = mod.FOV_DSM(circ_padding=circ, weights=w).cuda().eval()
        e = enc.forward_bf16(x.cuda()).cpu()
        # layers 5,7 | 10,12,14 | 17,19,21 | 23 | 25, 27 (0 and 2 are the fused first-two-layers kernel; 5 = 64 input channels: the
        # weight-resident kernel; 23 = 256 workgroups of 8 waves, exactly one per CU: csrc/api.hip witw_fills_rounds)
        assert ran == ['conv3x3_bf16_wres_kernel', 'conv3x3_bf16_s16_kernel<true,false>',
                       'conv3x3_bf16_s16_kernel<false,false>', 'conv3x3_bf16_s16_kernel<false,false>', 'conv3x3_bf16_s16_kernel<true,false>',
                       'conv3x3_bf16_s16_kernel<false,false>', 'conv3x3_bf16_s16_kernel<false,false>', 'conv3x3_bf16_s16_kernel<false,false>',
                       'conv3x3_nhwc_bf16_kernel<128,2,false,8>', 'conv3x3_nhwc_bf16_kernel<64,2,false,4>',
                       'conv3x3_nhwc_bf16_kernel<64,1,false,4>'], ran
        sel = [0, 77, 127]
        with torch.no_grad():
            emu = O.fov_dsm_forward_bf16_emulated(x[sel], wt, circ)
        for i, s in enumerate(sel):
            rel = float((e[s] - emu[i]).norm() / emu[i].norm())
            assert rel < 1e-2, (variant, circ, s, rel)


@pytest.mark.parametrize('circ', [False, True])
def test_f32_encoder_at_bench_batch_vs_oracle(circ, monkeypatch):
    """The HEADLINE path itself (BASELINE configs[1]: cvig_fov, bs = 128, fp32; FOV_DSM of model/cvig_fov.py:248-294, zero and
    circular padding) at the bench batch, directly against the oracle's FOV_DSM forward on images {0, 42, 65, 127} of that batch
    at north_star's 1e-4 -- the twin of the bf16 test above (VERDICT r04 weak #1a: until round 5 the fp32 bench batch was tied to
    the oracle only through B = 128 == B = 1 bitwise + the B = 2 goldens). The 13 kernel instantiations that ran are asserted."""
    from witw_amd import cvig_fov, ops
    B, seed = 128, 23
    w = synth.fov_dsm_weights(seed)
    wt = {k: (torch.from_numpy(a), torch.from_numpy(b)) for k, (a, b) in w.items()}
    x = torch.from_numpy(synth.normalized_images(seed, 50 + int(circ), (B, 3, 128, 512)))
    ran = []
    real, real_first = ops.conv3x3_fwd, ops.conv3x3_first_fwd

    def recording(*a, **k):
        out = real(*a, **k)
        ran.append(ops.last_kernel_variant())
        return out

    def recording_first(*a, **k):
        out = real_first(*a, **k)
        ran.append(ops.last_kernel_variant())
        return out
    monkeypatch.setattr(ops, 'conv3x3_fwd', recording)
    monkeypatch.setattr(ops, 'conv3x3_first_fwd', recording_first)
    enc = cvig_fov.FOV_DSM(circ_padding=circ, weights=w).cuda().eval()
    with torch.no_grad():
        e = enc(x.cuda()).cpu()
    # layers 0 | 2 | 5,7 | 10,12,14 | 17,19,21 | 23 | 25 | 27: TN, stride, fused pool, waves, geometry, taps
    assert ran == ['conv3x3_first_persist_kernel', 'conv3x3_nhwc_f32_kernel<64,1,true,8,0,9>',
                   'conv3x3_nhwc_f32_kernel<128,1,false,8,0,9>', 'conv3x3_nhwc_f32_kernel<128,1,true,8,0,9>',
                   'conv3x3_nhwc_f32_kernel<128,1,false,8,0,9>', 'conv3x3_nhwc_f32_kernel<128,1,false,8,0,9>',
                   'conv3x3_nhwc_f32_kernel<128,1,true,8,0,9>',
                   'conv3x3_nhwc_f32_kernel<128,1,false,8,0,9>', 'conv3x3_nhwc_f32_kernel<128,1,false,8,0,9>',
                   'conv3x3_nhwc_f32_kernel<128,1,false,8,0,9>',
                   'conv3x3_nhwc_f32_kernel<128,2,false,8,0,9>', 'conv3x3_nhwc_f32_kernel<64,2,false,4,0,9>',
                   'conv3x3_nhwc_f32_kernel<64,1,false,4,0,9>'], ran
    sel = [0, 42, 65, 127]
    with torch.no_grad():
        ref = O.fov_dsm_forward(x[sel], wt, circ)
    assert tuple(e.shape) == (B, 16, 4, 64)
    np.testing.assert_allclose(e[sel].numpy(), ref.numpy(), rtol=0, atol=1e-4)
    assert float(ref.abs().max()) > 0.05           # the tolerance bites: embeddings are O(0.1 - 1)


def test_bf16_s16_training_forms_vs_oracle():
    """The forms a bf16 training step adds to the forward (model/cvig_fov.py:444-461 with Dropout2d :234-245 and the stride-(2,1)
    layers :263-272), on the 16x16x32 kernel's TRAIN instantiation at a qualifying grid, each against the oracle: Dropout2d scale
    before the ReLU; the ReLU gate of a dgrad launch (transposed filter, outputs zeroed where the forward's activation was <= 0);
    zero-interleaved input rows (dgrad of a stride-(2,1) layer); the arg-max codes of the fused max-pool."""
    from witw_amd import ops
    dev = torch.device('cuda:0')
    B, H, W, cin, cout = 64, 16, 64, 256, 512
    x, w, b = _layer(41, B, H, W, cin, cout)
    x = x.bfloat16().float()
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().bfloat16()
    g = np.random.Generator(np.random.Philox(key=[41, 7]))
    sel = _pick(B)
    # (1) Dropout2d scale
    scale = torch.from_numpy((g.random((B, cout)) > 0.2).astype(np.float32) * 1.25)
    pk = ops.PackedConvBf16(w.to(dev), b.to(dev))
    y = ops.conv3x3_bf16_fwd(xd, pk, circular=True, relu=True, drop_scale=scale.to(dev))
    assert ops.last_kernel_variant() == 'conv3x3_bf16_s16_kernel<false,true>', ops.last_kernel_variant()
    ref = torch.relu(O.conv3x3(x[sel], w.bfloat16().float(), b, 1, True) * scale[sel][:, :, None, None])
    _assert_bf16_ulp(y[sel].float().cpu().permute(0, 3, 1, 2), ref)
    # (2) dgrad form: transposed + rotated filter, gate (a 512 -> 512 layer: 4 channel tiles x 64 images x 2 row tiles = 512 workgroups)
    cin = 512
    _x, w, _b = _layer(42, 1, 1, 1, cin, cout)
    dz = torch.from_numpy(g.standard_normal((B, cout, H, W), dtype=np.float32)).bfloat16().float()
    gate = torch.from_numpy(g.standard_normal((B, cin, H, W), dtype=np.float32)).bfloat16()
    pkt = ops.PackedConvBf16(w.to(dev), None, transpose_flip=True)
    dx = ops.conv3x3_bf16_fwd(dz.to(dev).permute(0, 2, 3, 1).contiguous().bfloat16(), pkt, circular=False, relu=False,
                              gate=gate.to(dev).permute(0, 2, 3, 1).contiguous())
    assert ops.last_kernel_variant() == 'conv3x3_bf16_s16_kernel<false,true>', ops.last_kernel_variant()
    wt = w.bfloat16().float().flip(2, 3).transpose(0, 1).contiguous()          # dgrad filter: [cin, cout, 3, 3], taps rotated
    ref = O.conv3x3(dz[sel], wt, torch.zeros(cin), 1, False) * (gate[sel].float() > 0)
    _assert_bf16_ulp(dx[sel].float().cpu().permute(0, 3, 1, 2), ref)
    # (3) zero-interleaved rows: 8 physical rows stand for 16 logical ones (dgrad of a stride-(2,1) conv, Ho = 8 -> H = 16)
    dz2 = dz[:, :, :8].contiguous()
    dx2 = ops.conv3x3_bf16_fwd(dz2.to(dev).permute(0, 2, 3, 1).contiguous().bfloat16(), pkt, circular=True, relu=False, dilate_h=True, out_h=16)
    assert ops.last_kernel_variant() == 'conv3x3_bf16_s16_kernel<false,true>', ops.last_kernel_variant()
    up = torch.zeros((len(sel), cout, 16, W))
    up[:, :, 0::2] = dz2[sel]
    ref = O.conv3x3(up, wt, torch.zeros(cin), 1, True)
    _assert_bf16_ulp(dx2[sel].float().cpu().permute(0, 3, 1, 2), ref)
    # (4) fused max-pool with arg-max codes (cvig_semantic trains through the pooled layers, model/cvig_semantic.py:301-309)
    B2, H2, W2, ci2, co2 = 32, 32, 128, 128, 256
    x2, w2, b2 = _layer(43, B2, H2, W2, ci2, co2)
    x2 = x2.bfloat16().float()
    y2, code = ops.conv3x3_bf16_fwd(x2.to(dev).permute(0, 2, 3, 1).contiguous().bfloat16(), ops.PackedConvBf16(w2.to(dev), b2.to(dev)),
                                    circular=True, relu=True, pool=True, want_pool_code=True)
    assert ops.last_kernel_variant() == 'conv3x3_bf16_s16_kernel<true,true>', ops.last_kernel_variant()
    sel2 = _pick(B2)
    pre = O.conv3x3(x2[sel2], w2.bfloat16().float(), b2, 1, True)
    ref2, idx = torch.nn.functional.max_pool2d(torch.relu(pre), 2, 2, return_indices=True)
    _assert_bf16_ulp(y2[sel2].float().cpu().permute(0, 3, 1, 2), ref2)
    # the recorded position must attain the window's maximum (ties and fp32 summation order may pick another of equal value)
    win = pre.unfold(2, 2, 2).unfold(3, 2, 2).reshape(len(sel2), co2, H2 // 2, W2 // 2, 4)
    picked = torch.gather(win, 4, code[sel2].cpu().permute(0, 3, 1, 2).long().unsqueeze(-1)).squeeze(-1)
    assert int(code.max()) <= 3
    assert bool(((win.max(-1).values - picked) <= 2e-5 * float(pre.abs().max())).all())
