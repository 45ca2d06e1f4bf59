"""GPU parity of the HIP conv3x3 / encoder against the CPU oracle and the reference goldens."""
import os

import numpy as np
import pytest
import torch

from oracle import cvig_fov_oracle as O
from witw_amd import synth

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star: embeddings within 1e-4 fp32


def _rand(seed, shape, scale=1.0):
    g = np.random.Generator(np.random.Philox(key=[seed, 7]))
    return (g.standard_normal(shape, dtype=np.float32) * np.float32(scale)).astype(np.float32)


CASES = [
    # B, H, W, Cin, Cout, stride_h, circ, relu, pool, nchw, drop
    (2, 8, 64, 8, 64, 1, False, True, False, False, False),
    (2, 8, 64, 8, 64, 1, True, True, True, False, False),
    (1, 12, 99, 16, 64, 1, False, True, True, False, False),     # ragged width, floor pooling
    (1, 12, 99, 16, 64, 1, True, False, False, False, False),
    (2, 16, 64, 64, 128, 1, True, True, False, False, True),     # TN=128, dropout scale
    (2, 16, 64, 32, 128, 1, False, True, True, False, False),    # TN=128 pooled
    (1, 16, 24, 64, 256, 2, True, True, False, False, False),    # stride (2,1), narrow
    (2, 8, 64, 64, 64, 2, False, True, False, False, False),
    (2, 4, 64, 64, 16, 1, True, False, False, True, False),      # last layer: Cout=16, NCHW out
    (1, 5, 130, 8, 64, 1, True, True, False, False, False),      # odd sizes, two column tiles
    (1, 4, 12, 24, 200, 1, True, True, False, False, False),     # Cout not a tile multiple
]


@pytest.mark.parametrize('case', CASES)
def test_conv3x3_matches_oracle(case):
    from witw_amd import ops
    B, H, W, Cin, Cout, sh, circ, relu, pool, nchw, drop = case
    x = _rand(1, (B, Cin, H, W))
    w = _rand(2, (Cout, Cin, 3, 3), scale=(2.0 / (9 * Cin)) ** 0.5)
    b = _rand(3, (Cout,), scale=0.1)
    scale = synth.dropout_scales(5, 0, B, Cout) if drop else None
    ref = O.conv3x3(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), sh, circ)
    if drop:
        ref = ref * torch.from_numpy(scale)[:, :, None, None]
    if relu:
        ref = torch.relu(ref)
    if pool:
        ref = torch.nn.functional.max_pool2d(ref, 2, 2)
    dev = torch.device('cuda:0')
    xd = torch.from_numpy(x).to(dev).permute(0, 2, 3, 1).contiguous()
    pk = ops.PackedConv(torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev))
    y = ops.conv3x3_fwd(xd, pk, stride_h=sh, circular=circ, relu=relu, pool=pool, out_nchw=nchw,
                        drop_scale=None if scale is None else torch.from_numpy(scale).to(dev))
    y = y.cpu() if nchw else y.cpu().permute(0, 3, 1, 2)
    assert y.shape == ref.shape
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-5, atol=2e-5)


def test_nchw_to_nhwc8():
    from witw_amd import ops
    x = torch.from_numpy(_rand(4, (2, 5, 7, 13))).cuda()
    y = ops.nchw_to_nhwc8(x).cpu()
    assert y.shape == (2, 7, 13, 8)
    np.testing.assert_array_equal(y[..., :5].numpy(), x.cpu().permute(0, 2, 3, 1).numpy())
    assert torch.all(y[..., 5:] == 0)


def test_encoder_matches_reference_goldens(golden_dir):
    from witw_amd import cvig_fov
    g = np.load(os.path.join(golden_dir, 'encoder.npz'))
    seed = int(g['seed'])
    w = synth.fov_dsm_weights(seed)
    x360 = torch.from_numpy(synth.normalized_images(seed, 10, (2, 3, 128, 512))).cuda()
    x70 = torch.from_numpy(synth.normalized_images(seed, 11, (2, 3, 128, 99))).cuda()
    for circ in (False, True):
        enc = cvig_fov.FOV_DSM(circ_padding=circ, weights=w).cuda().eval()
        with torch.no_grad():
            e360 = enc(x360).cpu().numpy()
            e70 = enc(x70).cpu().numpy()
        assert e360.shape == (2, 16, 4, 64) and e70.shape == (2, 16, 4, 12)
        np.testing.assert_allclose(e360, g['embed360_circ%d' % circ], rtol=0, atol=TOL)
        np.testing.assert_allclose(e70, g['embed70_circ%d' % circ], rtol=0, atol=TOL)
    # train mode with the reference's captured Dropout2d masks injected
    enc.train()
    scales = {i: torch.from_numpy(g['drop_scale_%d' % i]).cuda() for i in (17, 19, 21)}
    with torch.no_grad():
        e = enc(x360, dropout_scales=scales).cpu().numpy()
    np.testing.assert_allclose(e, g['embed360_circ1_train'], rtol=0, atol=TOL)


def test_encoder_from_vgg16_state_dict_matches_reference_goldens(golden_dir):
    """The reference builds its encoders from a torchvision VGG16 (model/cvig_fov.py:256-261; the goldens: the same through a
    VGG16-shaped module carrying synth's weights, tests/golden/gen_golden.py). The same weights handed over as a
    torchvision-format state_dict (`features.{i}.*`) through FOV_DSM.from_vgg16_state_dict give the same embeddings."""
    from witw_amd import cvig_fov
    g = np.load(os.path.join(golden_dir, 'encoder.npz'))
    seed = int(g['seed'])
    w = synth.fov_dsm_weights(seed)
    sd = {}
    for idx in cvig_fov.VGG16_CONVS:
        sd['features.%d.weight' % idx] = torch.from_numpy(w[idx][0].copy())
        sd['features.%d.bias' % idx] = torch.from_numpy(w[idx][1].copy())
    x360 = torch.from_numpy(synth.normalized_images(seed, 10, (2, 3, 128, 512))).cuda()
    for circ in (False, True):
        enc = cvig_fov.FOV_DSM.from_vgg16_state_dict(sd, circ_padding=circ)
        with torch.no_grad():
            for idx in (23, 25, 27):            # the goldens pin the randomly initialised head to synth's values
                conv = cvig_fov._conv_of(enc.model.features[idx])
                conv.weight.copy_(torch.from_numpy(w[idx][0]))
                conv.bias.copy_(torch.from_numpy(w[idx][1]))
        enc = enc.cuda().eval()
        with torch.no_grad():
            np.testing.assert_allclose(enc(x360).cpu().numpy(), g['embed360_circ%d' % circ], rtol=0, atol=TOL)


def test_encoder_state_dict_keys_match_reference(golden_dir):
    from witw_amd import cvig_fov
    g = np.load(os.path.join(golden_dir, 'encoder.npz'))
    for circ, name in ((False, 'keys_surface'), (True, 'keys_overhead')):
        enc = cvig_fov.FOV_DSM(circ_padding=circ)
        ref_keys = {k for k in g[name] if not k.startswith('model.classifier')}
        assert set(enc.state_dict().keys()) == ref_keys
    enc = cvig_fov.FOV_DSM(circ_padding=False)
    ref_train = [k for k in g['trainable_surface'] if not k.startswith('model.classifier')]
    assert sorted(n for n, p in enc.named_parameters() if p.requires_grad) == ref_train


def test_cpu_tensor_is_refused():
    from witw_amd import _lib, cvig_fov
    enc = cvig_fov.FOV_DSM()
    with pytest.raises(_lib.WitwError):
        enc(torch.zeros(1, 3, 128, 512))


def test_semantic_encoder_matches_reference_golden(golden_dir):
    from witw_amd import cvig_semantic
    g = np.load(os.path.join(golden_dir, 'encoder_semantic.npz'))
    seed = int(g['seed'])
    w5 = synth.fov_dsm_weights(seed, in_channels=5)
    x5 = torch.from_numpy(synth.normalized_images(seed, 12, (1, 5, 128, 512))).cuda()
    enc = cvig_semantic.FOV_DSM(circ_padding=True, weights=w5).cuda().eval()
    with torch.no_grad():
        e = enc(x5).cpu().numpy()
    np.testing.assert_allclose(e, g['embed5_circ1'], rtol=0, atol=TOL)
    ref_train = sorted(k for k in g['trainable'] if not k.startswith('model.classifier'))
    assert sorted(n for n, p in enc.named_parameters() if p.requires_grad) == ref_train


@pytest.mark.parametrize('shape', [(2, 3, 16, 64, True), (1, 3, 13, 99, False), (2, 4, 8, 130, True), (1, 1, 9, 40, False)])
def test_first_layer_fast_path_matches_oracle(shape):
    from witw_amd import ops
    B, C, H, W, circ = shape
    x = _rand(11, (B, C, H, W))
    w = _rand(12, (64, C, 3, 3), scale=(2.0 / (9 * C)) ** 0.5)
    b = _rand(13, (64,), scale=0.1)
    ref = torch.relu(O.conv3x3(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), 1, circ))
    dev = torch.device('cuda:0')
    pk = ops.PackedFirstConv(torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev))
    y = ops.conv3x3_first_fwd(torch.from_numpy(x).to(dev), pk, circular=circ).cpu().permute(0, 3, 1, 2)
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-5, atol=2e-5)
    pkb = ops.PackedFirstConv(torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev), bf16=True)
    yb = ops.conv3x3_first_fwd(torch.from_numpy(x).to(dev), pkb, circular=circ)
    assert yb.dtype == torch.bfloat16
    refb = torch.relu(O.conv3x3(torch.from_numpy(x).bfloat16().float(), torch.from_numpy(w).bfloat16().float(),
                                torch.from_numpy(b), 1, circ)).bfloat16().float()
    np.testing.assert_allclose(yb.float().cpu().permute(0, 3, 1, 2).numpy(), refb.numpy(), rtol=2 ** -7, atol=1e-6)


@pytest.mark.parametrize('case', [(24, 3, 64, 512, True), (32, 3, 72, 200, False), (18, 4, 64, 520, True), (40, 1, 40, 330, False)])
def test_first_layer_persistent_kernel_equals_the_tile_per_workgroup_kernel(case):
    """conv3x3_first_persist_kernel (round 4: two persistent workgroups per CU, the next tile's pixels requested before the current
    tile's MFMAs; taken from 4 x CUs tiles on, W >= 66) against conv3x3_first_kernel, which the same images take when they are
    launched one at a time (fewer tiles than the threshold): the same products in the same order, so the same BITS; and against
    the oracle on a spread of images. Ragged widths / heights, both paddings, 1 / 3 / 4 channels."""
    from witw_amd import ops
    B, C, H, W, circ = case
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    assert B * ((W + 63) // 64) * ((H + 7) // 8) >= 4 * n_cu and ((W + 63) // 64) * ((H + 7) // 8) < 4 * n_cu
    g = np.random.Generator(np.random.Philox(key=[7, H * W + C]))
    x = g.standard_normal((B, C, H, W), dtype=np.float32)
    w = (g.standard_normal((64, C, 3, 3), dtype=np.float32) * 0.3).astype(np.float32)
    b = (g.standard_normal((64,), dtype=np.float32) * 0.2).astype(np.float32)
    dev = torch.device('cuda:0')
    pk = ops.PackedFirstConv(torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev))
    xd = torch.from_numpy(x).to(dev)
    y = ops.conv3x3_first_fwd(xd, pk, circular=circ)
    one = torch.cat([ops.conv3x3_first_fwd(xd[i:i + 1].contiguous(), pk, circular=circ) for i in range(B)])
    assert torch.equal(y, one), float((y - one).abs().max())
    sel = [0, B // 2, B - 1]
    ref = O.conv3x3(torch.from_numpy(x[sel]), torch.from_numpy(w), torch.from_numpy(b), 1, circ).clamp_min(0)
    np.testing.assert_allclose(y[sel].cpu().permute(0, 3, 1, 2).numpy(), ref.numpy(), rtol=0, atol=2e-5 * float(ref.abs().max()))


@pytest.mark.parametrize('C', [1, 2, 3])
def test_first_layer_missing_planes_read_zeros_not_the_next_image(C):
    """the persistent first-layer kernels load plane ch of a pixel with the plane offset in the buffer load's
    soffset, which the hardware range check does not see; for C < 4 (C < CW in the fused bf16 kernel) the planes ch >= C of image b
    therefore ALIASED planes of image b + 1 -- finite data times zero weights, invisible to every bitwise test, NaN as soon as the
    neighbour holds a NaN / Inf. Poison every image but the first: image 0's output must stay finite and equal, bit for bit, what it
    gives when launched alone (the tile-per-workgroup kernel, which guards ch < C)."""
    from witw_amd import ops
    B, H, W = 16, 64, 512
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    assert B * (W // 64) * (H // 8) >= 4 * n_cu
    g = np.random.Generator(np.random.Philox(key=[11, C]))
    x = g.standard_normal((B, C, H, W), dtype=np.float32)
    x[1:] = np.nan
    w = (g.standard_normal((64, C, 3, 3), dtype=np.float32) * 0.3).astype(np.float32)
    b = (g.standard_normal((64,), dtype=np.float32) * 0.2).astype(np.float32)
    dev = torch.device('cuda:0')
    xd = torch.from_numpy(x).to(dev)
    pk = ops.PackedFirstConv(torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev))
    y = ops.conv3x3_first_fwd(xd, pk, circular=True)
    assert 'persist' in ops.last_kernel_variant(), ops.last_kernel_variant()
    alone = ops.conv3x3_first_fwd(xd[:1].contiguous(), pk, circular=True)
    assert bool(torch.isfinite(y[0]).all()) and torch.equal(y[0], alone[0])
    # the fused layers 0 + 2 of the bf16 path (conv_first2_bf16_kernel, CW = 4 planes per pixel for C <= 4)
    w2 = (g.standard_normal((64, 64, 3, 3), dtype=np.float32) * 0.05).astype(np.float32)
    b2 = (g.standard_normal((64,), dtype=np.float32) * 0.1).astype(np.float32)
    pkb = ops.PackedFirstConv(torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev), bf16=True)
    pk2 = ops.PackedConvBf16(torch.from_numpy(w2).to(dev), torch.from_numpy(b2).to(dev))
    yf = ops.conv_first2_bf16(xd, pkb, pk2, circular=True)
    yf_alone = ops.conv_first2_bf16(xd[:1].contiguous(), pkb, pk2, circular=True)
    assert bool(torch.isfinite(yf[0].float()).all()) and torch.equal(yf[0], yf_alone[0])


@pytest.mark.parametrize('circ', [False, True])
def test_dilated_dgrad_launch_skips_the_zero_rows_bitwise(circ):
    """The data gradient of a stride-(2,1) conv (layers 23 / 25, model/cvig_fov.py:263-272 through autograd) runs the forward kernel on a
    zero-interleaved input (dilate_h). On the 8-wave 128-channel tile the launch (round 5: GEO = 2) does not issue the MFMAs whose input
    rows are the interleaved zeros -- half of them. Same bits as the launch that multiplies them (witw_conv3x3_dil_skip(0)), and the
    oracle's transposed convolution on a spread of images, gate and Dropout2d scale in the epilogue included."""
    from witw_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device('cuda:0')
    B, Hp, W, cin, cout = 32, 8, 64, 32, 512          # dz [B,8,64,32] stands for 16 rows; 4 x 32 x 2 = 256 workgroups of 8 waves
    g = np.random.Generator(np.random.Philox(key=[23, int(circ)]))
    dz = torch.from_numpy(g.standard_normal((B, cin, Hp, W), dtype=np.float32))
    w = torch.from_numpy(g.standard_normal((cin, cout, 3, 3), dtype=np.float32) * 0.1)      # the layer's filter [its cout = cin here][its cin = cout here]
    gate = torch.from_numpy(g.standard_normal((B, cout, 16, W), dtype=np.float32))
    scale = torch.from_numpy((g.random((B, cout)) > 0.2).astype(np.float32) * 1.25)
    pkt = ops.PackedConv(w.to(dev), None, transpose_flip=True)
    xd = dz.to(dev).permute(0, 2, 3, 1).contiguous()
    gd = gate.to(dev).permute(0, 2, 3, 1).contiguous()

    def run():
        return ops.conv3x3_fwd(xd, pkt, stride_h=1, circular=circ, relu=False, drop_scale=scale.to(dev), gate=gd, dilate_h=True, out_h=16)
    prev = lib.witw_conv3x3_dil_skip(1)
    try:
        y1 = run()
        assert ops.last_kernel_variant() == 'conv3x3_nhwc_f32_kernel<128,1,false,8,2,9>', ops.last_kernel_variant()
        lib.witw_conv3x3_dil_skip(0)
        y0 = run()
        assert ops.last_kernel_variant() == 'conv3x3_nhwc_f32_kernel<128,1,false,8,0,9>', ops.last_kernel_variant()
    finally:
        lib.witw_conv3x3_dil_skip(prev)
    assert torch.equal(y1, y0)
    sel = [0, 13, 31]
    up = torch.zeros((len(sel), cin, 16, W))
    up[:, :, 0::2] = dz[sel]
    wt = w.flip(2, 3).transpose(0, 1).contiguous()            # [cout, cin, 3, 3], taps rotated
    ref = O.conv3x3(up, wt, torch.zeros(cout), 1, circ) * scale[sel][:, :, None, None] * (gate[sel] > 0)
    np.testing.assert_allclose(y1[sel].cpu().permute(0, 3, 1, 2).numpy(), ref.numpy(), rtol=0, atol=3e-5 * float(ref.abs().max()))
