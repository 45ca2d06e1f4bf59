"""GPU parity of matching / loss / ranking / data-path kernels against reference goldens and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import cvig_fov_oracle as O
from witw_amd import synth

pytestmark = pytest.mark.gpu


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_match_matches_reference_goldens(golden_dir):
    from witw_amd import cvig_fov
    g = _g(golden_dir, 'matching.npz')
    seed = int(g['seed'])
    for tag in 'abcde':
        bo, bs, we = (int(v) for v in g['%s_shape' % tag])
        ov = torch.from_numpy(synth.embeddings(seed, 100 + ord(tag), (bo, 16, 4, 64))).cuda()
        su = torch.from_numpy(synth.embeddings(seed, 200 + ord(tag), (bs, 16, 4, we))).cuda()
        ori, dist = cvig_fov.match(ov, su)
        assert ori.dtype == torch.int64 and ori.shape == (bo, bs)
        np.testing.assert_array_equal(ori.cpu().numpy(), g['%s_orientation' % tag])      # bit-exact indices
        np.testing.assert_allclose(dist.cpu().numpy(), g['%s_distance' % tag], rtol=0, atol=1e-5)
        # the reference's three-call form through the compatibility entry points
        ori2 = cvig_fov.correlation(ov, su)
        crop = cvig_fov.crop_overhead(ov, ori2, we)
        d2 = cvig_fov.l2_distance(crop, su)
        np.testing.assert_array_equal(ori2.cpu().numpy(), g['%s_orientation' % tag])
        np.testing.assert_allclose(d2.cpu().numpy(), g['%s_distance' % tag], rtol=0, atol=1e-5)
        if tag == 'c':
            np.testing.assert_array_equal(crop.cpu().numpy(), g['c_crop'])
        if bo == bs:
            loss = cvig_fov.triplet_loss(dist)
            np.testing.assert_allclose(loss.item(), float(g['%s_loss' % tag]), rtol=1e-5)


def test_match_tie_break_first_index():
    """torch.argmax returns the first maximal index; periodic overhead rows tie exactly."""
    from witw_amd import cvig_fov
    base = synth.embeddings(3, 1, (3, 16, 4, 16))
    ov = np.tile(base, (1, 1, 1, 4))                      # period 16 -> shifts s, s+16, s+32, s+48 tie
    su = np.ascontiguousarray(np.roll(ov, -5, axis=3)[:, :, :, :32])
    ori = cvig_fov.correlation(torch.from_numpy(ov).cuda(), torch.from_numpy(su).cuda()).cpu()
    ref = O.correlation(torch.from_numpy(ov), torch.from_numpy(su))
    assert torch.equal(torch.diagonal(ori), torch.full((3,), 5, dtype=torch.int64))
    assert torch.equal(torch.diagonal(ref), torch.full((3,), 5, dtype=torch.int64))


# (128, 128, *), (100, 77, 64), (130, 257, 64): one minibatch -> match_kernel_nsplit<64 | 32 | 16> (a workgroup per overhead, the waves split
# the shifts); the rest: the generic and the pipelined kernels
@pytest.mark.parametrize('shape', [(1, 1, 64), (130, 257, 64), (3, 200, 1), (70, 5, 63), (1101, 130, 64), (1030, 129, 63), (128, 128, 64),
                                   (128, 128, 12), (128, 128, 33), (128, 128, 17), (100, 77, 64), (128, 128, 1)])
def test_match_ragged_shapes_vs_oracle(shape):
    from witw_amd import cvig_fov
    bo, bs, we = shape
    ov = torch.from_numpy(synth.embeddings(11, bo, (bo, 16, 4, 64)))
    su = torch.from_numpy(synth.embeddings(12, bs, (bs, 16, 4, we)))
    ori_r, dist_r = O.match_fused(ov, su)
    sc = O.correlation_scores(ov.double(), su.double())
    top2 = sc.topk(2, -1).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-3            # compare indices away from argmax near-ties
    ori, dist = cvig_fov.match(ov.cuda(), su.cuda())
    ori, dist = ori.cpu(), dist.cpu()
    assert torch.equal(ori[safe], ori_r[safe])
    np.testing.assert_allclose(dist[safe].numpy(), dist_r[safe].numpy(), rtol=0, atol=1e-5)
    assert safe.float().mean() > (0.95 if we > 1 else 0.5)


def test_triplet_loss_fwd_bwd(golden_dir):
    from witw_amd import cvig_fov
    g = _g(golden_dir, 'matching.npz')
    dm = torch.from_numpy(g['loss_in'])
    assert abs(cvig_fov.triplet_loss(dm.cuda()).item() - float(g['loss_a10'])) < 1e-6
    assert abs(cvig_fov.triplet_loss(dm.cuda(), alpha=3.).item() - float(g['loss_a3'])) < 1e-6
    for B in (2, 7, 300):
        d = torch.from_numpy(synth.embeddings(5, B, (B, B))).abs().clamp(max=3.9)
        dr = d.clone().requires_grad_(True)
        lr = O.triplet_loss(dr)
        lr.backward()
        dg = d.cuda().requires_grad_(True)
        lg = cvig_fov.triplet_loss(dg)
        (lg * 1.0).backward()
        np.testing.assert_allclose(lg.item(), lr.item(), rtol=2e-6)
        np.testing.assert_allclose(dg.grad.cpu().numpy(), dr.grad.numpy(), rtol=1e-4, atol=1e-8)
    with pytest.raises(Exception):
        cvig_fov.triplet_loss(torch.zeros(1, 1).cuda())


def test_ranking_matches_reference_goldens(golden_dir):
    from witw_amd import cvig_fov
    g = _g(golden_dir, 'ranking.npz')
    seed = int(g['seed'])
    for tag in ('r360', 'r70'):
        n, we = (int(v) for v in g['%s_n_we' % tag])
        ov = torch.from_numpy(synth.embeddings(seed, 400 + we, (n, 16, 4, 64)))
        noise = torch.from_numpy(synth.embeddings(seed, 500 + we, (n, 16, 4, we)))
        shifts = g['%s_shifts' % tag]
        su = torch.stack([torch.roll(ov[i], -int(shifts[i]), dims=2)[:, :, :we] for i in range(n)]) \
            + float(g['%s_noise' % tag]) * noise
        r = cvig_fov.ranks(ov.cuda(), su.contiguous().cuda())
        np.testing.assert_array_equal(r, g['%s_ranks' % tag])                         # bit-exact ranks
        ori, dist = cvig_fov.match(ov.cuda(), su.contiguous().cuda())
        np.testing.assert_array_equal(ori.cpu().numpy(), g['%s_ori' % tag])
        np.testing.assert_allclose(dist.cpu().numpy(), g['%s_dist' % tag], rtol=0, atol=1e-5)
        t = cvig_fov.recall_table(r)
        np.testing.assert_allclose([t['top_1'], t['top_5'], t['top_10'], t['top_1pct'], t['mean'], t['median']],
                                   g['%s_table' % tag])


def test_polar_transform_bit_exact(golden_dir):
    from witw_amd import cvig_fov
    g = _g(golden_dir, 'polar.npz')
    img = torch.from_numpy(synth.normalized_images(int(g['seed']), int(g['stream']), (3, 256, 256)))
    out = cvig_fov.PolarTransform()({'overhead': img.cuda()})['polar'].cpu()
    assert out.shape == (3, 128, 512)
    np.testing.assert_array_equal(out[:, ::8, :].numpy(), g['polar_rows'])
    assert out.double().sum().item() == float(g['polar_sum'])
    assert out.double().abs().sum().item() == float(g['polar_abs_sum'])
    batched = cvig_fov.PolarTransform()({'overhead': torch.stack([img, img * 2]).cuda()})['polar'].cpu()
    np.testing.assert_array_equal(batched[0].numpy(), out.numpy())
    np.testing.assert_array_equal(batched[1].numpy(), O.polar_transform(img * 2).numpy())


def test_fused_resize_normalize_polar_equals_the_three_launches(golden_dir):
    """witw_polar_from_raw (Resize -> ImageNormalization -> PolarTransform of the overhead side in one launch,
    model/cvig_fov.py:117-209) against the separate launches: the same BITS for uniform fp32 batches of BASELINE's raw size, odd
    sizes, 1 / 3 / 5 channels with cvig_semantic's normalisation, and no normalisation; and the reference golden of the polar
    transform itself through the fused kernel (a 256 x 256 source resizes to itself exactly)."""
    from witw_amd import ops
    g = _g(golden_dir, 'polar.npz')
    img = torch.from_numpy(synth.normalized_images(int(g['seed']), int(g['stream']), (3, 256, 256))).cuda()
    out = ops.polar_from_raw(img.unsqueeze(0))[0].cpu()
    np.testing.assert_array_equal(out[:, ::8, :].numpy(), g['polar_rows'])
    assert out.double().sum().item() == float(g['polar_sum'])
    mean, std = list(O.SEM_MEAN), list(O.SEM_STD)
    for (B, C, Hi, Wi) in ((7, 3, 512, 512), (2, 5, 512, 512), (3, 3, 500, 470), (2, 1, 300, 777), (1, 3, 97, 64), (13, 3, 256, 256)):
        x = torch.from_numpy(synth.images_u8(70 + C, Hi, (B, C, Hi, Wi))).cuda()
        if C > 3:
            x[:, 3:] /= 255.0
        for norm in (True, False):
            m, s_, nd = (mean[:C], std[:C], min(C, 3)) if norm else (None, None, None)
            ref = ops.polar_transform(ops.resize_bilinear(x, (256, 256), m, s_, nd))
            got = ops.polar_from_raw(x, mean=m, std=s_, n_div255=nd)
            assert torch.equal(ref, got), (B, C, Hi, Wi, norm, float((ref - got).abs().max()))
    # a polar geometry other than the model's (heat-map tiles): its own tile table
    x = torch.from_numpy(synth.images_u8(75, 1, (2, 3, 200, 200))).cuda()
    ref = ops.polar_transform(ops.resize_bilinear(x, (128, 128), mean[:3], std[:3]), 64, 256)
    got = ops.polar_from_raw(x, mean=mean[:3], std=std[:3], size=128, h_s=64, w_s=256)
    assert torch.equal(ref, got)


@pytest.mark.parametrize('B,C,Hi,Wi', [(3, 3, 512, 512), (2, 5, 512, 512), (2, 3, 500, 470), (1, 3, 256, 256)])
def test_fused_resize_normalize_polar_against_the_oracle_directly(B, C, Hi, Wi):
    """ops.polar_from_raw against the ORACLE's own composition O.polar_transform(normalise(O.resize_pair(...)[1])) -- not against
    another HIP path -- at BASELINE's 512 -> 256 raw size (3 bands, and cvig_semantic's 5 bands with its normalisation,
    model/cvig_semantic.py:167-176), an odd size on the general path and the identity resize. Tolerance: the resize restatement's
    2e-4 on the 0..255 scale (tests above) carried through /255 and /std (<= 2e-4 / 255 / 0.22 = 3.6e-6) and the polar taps'
    convex combination, plus fp32 rounding of the normalised values (|x| <= 2.7): 2e-5 absolute."""
    from witw_amd import ops
    x = torch.from_numpy(synth.images_u8(90 + C, Hi, (B, C, Hi, Wi)))
    if C > 3:
        x[:, 3:] /= 255.0
    mean, std = (list(O.SEM_MEAN), list(O.SEM_STD)) if C == 5 else (list(O.IMG_MEAN), list(O.IMG_STD))
    norm = O.image_normalization_semantic if C == 5 else O.image_normalization
    got = ops.polar_from_raw(x.cuda(), mean=mean, std=std, n_div255=3).cpu()
    assert got.shape == (B, C, 128, 512)
    for i in range(B):
        _s, ov = O.resize_pair(torch.zeros(C, 224, 224), x[i])
        ref = O.polar_transform(norm(ov))
        np.testing.assert_allclose(got[i].numpy(), ref.numpy(), rtol=0, atol=2e-5, err_msg=str((B, C, Hi, Wi, i)))
    # the data path's call form (descriptor table of individually sized uint8 HWC images, GpuPreprocess) against the same oracle
    if C == 3:
        from witw_amd import cvig_fov
        u8 = [x[i].permute(1, 2, 0).to(torch.uint8).numpy() for i in range(B)]
        g8 = [np.zeros((224, 224, 3), dtype=np.uint8) for _ in range(B)]
        prep = cvig_fov.GpuPreprocess('witw', 360, random_orientation=False, device=torch.device('cuda:0'))
        out = prep(cvig_fov.collate_packed([{'surface': g8[i], 'overhead': u8[i]} for i in range(B)]))
        for i in range(B):
            ref = O.polar_transform(O.image_normalization(O.resize_pair(torch.zeros(3, 224, 224), x[i])[1]))
            np.testing.assert_allclose(out['polar'][i].cpu().numpy(), ref.numpy(), rtol=0, atol=2e-5)


def test_resize_normalisation_division_is_the_ieee_quotient():
    """The resize kernels divide by 255 and by std through a reciprocal and two fma (csrc/preprocess.hip: div_exact); the result
    must be the correctly rounded quotient, i.e. equal witw_normalize (true divisions) applied to the un-normalised resize --
    the reference's own order of operations (Resize, then ImageNormalization: model/cvig_fov.py:117-149)."""
    from witw_amd import ops
    for C, mean, std, nd in ((3, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], 3), (5, list(O.SEM_MEAN), list(O.SEM_STD), 3),
                             (2, [0.1, -3.0], [1.9999999, 7.0], 1)):        # 1.9999999 = all-ones significand: the true division
        x = torch.from_numpy(synth.images_u8(80, C, (4, C, 224, 224))).cuda()
        x = x + torch.rand_like(x)            # arbitrary significands, not only integers
        a = ops.resize_bilinear(x, (128, 512), mean, std, nd)
        b = ops.normalize(ops.resize_bilinear(x, (128, 512)), mean, std, nd)
        assert torch.equal(a, b), (C, float((a - b).abs().max()))


def test_normalization_bit_exact(golden_dir):
    from witw_amd import cvig_fov
    g = _g(golden_dir, 'normalize.npz')
    raw5 = torch.from_numpy(g['raw5'])
    d = cvig_fov.ImageNormalization()({'surface': raw5[:3].cuda(), 'overhead': raw5[:3].cuda()})
    np.testing.assert_array_equal(d['surface'].cpu().numpy(), g['norm3'])
    d5 = cvig_fov.ImageNormalization(O.SEM_MEAN, O.SEM_STD, n_div255=3)({'surface': raw5.cuda(), 'overhead': raw5.cuda()})
    np.testing.assert_array_equal(d5['overhead'].cpu().numpy(), g['norm5'])


@pytest.mark.parametrize('fov,panorama', [(360, False), (70, False), (70, True), (360, True)])
def test_resize_vs_oracle(fov, panorama):
    from witw_amd import cvig_fov
    s = torch.from_numpy(synth.images_u8(7, 1, (3, 224, 224)))
    o = torch.from_numpy(synth.images_u8(7, 2, (3, 512, 512)))
    rs, ro = O.resize_pair(s, o, fov=fov, panorama=panorama, start=470)
    t = cvig_fov.Resize('cvusa' if panorama else 'witw', fov)
    d = t({'surface': s.cuda(), 'overhead': o.cuda()}, start=470)
    assert d['surface'].shape == rs.shape and d['overhead'].shape == (3, 256, 256)
    np.testing.assert_allclose(d['surface'].cpu().numpy(), rs.numpy(), rtol=0, atol=2e-4)   # values up to 255
    np.testing.assert_allclose(d['overhead'].cpu().numpy(), ro.numpy(), rtol=0, atol=2e-4)


def test_global_batch_slab_evaluation_equals_full():
    """Column-slab evaluation (what each rank does at N>1) summed over slabs == the single-GPU result."""
    from witw_amd import cvig_fov, ops
    B, we = 24, 12
    ov = torch.from_numpy(synth.embeddings(71, 1, (B, 16, 4, 64))).cuda()
    su = torch.from_numpy(synth.embeddings(71, 2, (B, 16, 4, we))).cuda()
    ori, dist = cvig_fov.match(ov, su)
    loss = cvig_fov.triplet_loss(dist)
    ranks = ops.rank_count(dist, 0)
    diag = dist.diagonal().contiguous()
    total = 0.0
    for col0 in (0, 8, 16):
        slab = dist[:, col0:col0 + 8].contiguous()
        total += ops.triplet_loss_slab_fwd(slab, diag, col0).item()
        np.testing.assert_array_equal(ops.rank_count(slab, col0).cpu().numpy(), ranks[col0:col0 + 8].cpu().numpy())
    np.testing.assert_allclose(total / (2. * B * (B - 1)), loss.item(), rtol=1e-5)
    l1, r1, o1, d1 = cvig_fov.evaluate_global_batch(ov, su, 0)          # world size 1: slab == everything
    np.testing.assert_allclose(l1.item(), loss.item(), rtol=1e-5)
    np.testing.assert_array_equal(r1.cpu().numpy(), ranks.cpu().numpy())


@pytest.mark.parametrize('shape', [(1000, 70, 5), (37, 300, 10), (5, 3, 8), (4096, 130, 32), (300, 64, 1), (20011, 200, 10),
                                   (513, 64, 16), (70000, 65, 3)])
def test_topk_smallest_bit_exact(shape):
    from witw_amd import ops
    bo, bs, k = shape
    g = np.random.Generator(np.random.Philox(key=[bo, bs]))
    d = g.integers(0, 50, size=(bo, bs)).astype(np.float32) / 8          # many exact ties
    vals, idx = ops.topk_smallest(torch.from_numpy(d).cuda(), k)
    order = np.lexsort((np.broadcast_to(np.arange(bo)[:, None], d.shape), d), axis=0)   # (distance, index) ascending
    kk = min(k, bo)
    np.testing.assert_array_equal(idx.cpu().numpy()[:, :kk], order[:kk].T)
    np.testing.assert_array_equal(vals.cpu().numpy()[:, :kk], np.take_along_axis(d, order[:kk], 0).T)
    if bo < k:
        assert np.all(idx.cpu().numpy()[:, bo:] == -1)
    # the single-pass entry (no workspace) and the row-split entry agree bit for bit, with a row offset and NaNs as well
    from witw_amd import _lib
    lib = _lib.load()
    d[::7, ::3] = np.nan
    dc = torch.from_numpy(d).cuda()
    v0 = torch.empty((bs, k), dtype=torch.float32, device='cuda')
    i0 = torch.empty((bs, k), dtype=torch.int64, device='cuda')
    _lib.check(lib.witw_topk_smallest(dc.data_ptr(), v0.data_ptr(), i0.data_ptr(), bo, bs, k, 1000, None), 'witw_topk_smallest')
    v1, i1 = ops.topk_smallest(dc, k, row_offset=1000)
    assert torch.equal(i0, i1) and torch.equal(v0, v1)


def test_retrieve_topk_matches_oracle_and_shards():
    from witw_amd import cvig_fov, ops
    n, we, k = 60, 12, 5
    ov = torch.from_numpy(synth.embeddings(81, 1, (n, 16, 4, 64)))
    su = torch.stack([torch.roll(ov[i], -int(i % 64), dims=2)[:, :, :we] for i in range(n)]) \
        + 5.0 * torch.from_numpy(synth.embeddings(81, 2, (n, 16, 4, we)))
    _, d_ref = O.match_fused(ov, su)
    v, i = cvig_fov.retrieve_topk(ov.cuda(), su.contiguous().cuda(), k=k, query_chunk=25)
    ref_i = torch.argsort(d_ref, dim=0, stable=True)[:k].t()
    gaps = torch.sort(d_ref, dim=0).values
    safe = (gaps[1:k + 1] - gaps[:k]).min(0).values > 1e-5           # queries whose top-(k+1) is free of near-ties
    assert safe.float().mean() > 0.9
    assert torch.equal(i.cpu()[safe], ref_i[safe])
    np.testing.assert_allclose(v.cpu().numpy(), torch.sort(d_ref, dim=0).values[:k].t().numpy(), atol=1e-5)
    # two shards merged by hand == unsharded
    _, dist = cvig_fov.match(ov.cuda(), su.contiguous().cuda())
    v1, i1 = ops.topk_smallest(dist[:23].contiguous(), k, 0)
    v2, i2 = ops.topk_smallest(dist[23:].contiguous(), k, 23)
    cv = torch.cat((v1.t(), v2.t())).contiguous()
    ci = torch.cat((i1.t(), i2.t()))
    order = torch.argsort(ci, dim=0, stable=True)
    vm, pm = ops.topk_smallest(torch.gather(cv, 0, order).contiguous(), k)
    im = torch.gather(torch.gather(ci, 0, order).t(), 1, pm)
    assert torch.equal(im, i) and torch.equal(vm, v)


@pytest.mark.parametrize('we', [1, 7, 12, 16, 17, 24, 32, 33, 48, 62])
def test_narrow_surface_pipelined_kernel_is_bit_identical(we, monkeypatch):
    """Retrieval-sized problems with surfaces narrower than 63 columns take match_kernel_rows (several embedding
    rows per LDS stage, zero-padded columns). Zero columns add exact zeros, so scores, orientations and distances
    must equal the generic kernel's bit for bit; a sample of queries is also checked against the oracle."""
    from witw_amd import ops
    bo, bs = 130, 1030                                   # ragged on both sides of the 4 x 128 workgroup tile
    ov = torch.from_numpy(synth.embeddings(80 + we, 1, (bo, 16, 4, 64))).cuda()
    su = torch.from_numpy(synth.embeddings(80 + we, 2, (bs, 16, 4, we))).cuda()
    ori, dist, score = ops.match_fwd(ov, su, want_score=True)
    monkeypatch.setenv('WITW_MATCH_GENERIC', '1')
    ori_g, dist_g, score_g = ops.match_fwd(ov, su, want_score=True)
    monkeypatch.delenv('WITW_MATCH_GENERIC')
    assert torch.equal(score, score_g) and torch.equal(ori, ori_g) and torch.equal(dist, dist_g)
    sel = [0, 1, 127, 128, 1029]
    o_ref, d_ref = O.match(ov.cpu(), su[sel].cpu())
    np.testing.assert_array_equal(ori[:, sel].cpu().numpy(), o_ref.numpy())
    np.testing.assert_allclose(dist[:, sel].cpu().numpy(), d_ref.numpy(), atol=2e-5)


def test_sharded_loss_slab_kernels_equal_full_matrix_gradient():
    """witw_triplet_loss_slab_fwd/_sig/_bwd on the two column slabs of a [B,B] distance matrix (the all-reduces done by
    hand) reproduce witw_triplet_loss_fwd/bwd on the whole matrix: same loss, same dL/dD, column block by column block."""
    from witw_amd import ops
    B, b = 70, 35
    g = np.random.Generator(np.random.Philox(key=[90, 1]))
    D = torch.from_numpy((g.random((B, B), dtype=np.float32) * 1.5 + 0.2)).cuda()
    loss_full, ws = ops.triplet_loss_fwd(D, 10.)
    one = torch.ones((1,), device='cuda')
    g_full = ops.triplet_loss_bwd(D, ws, one, 10.)
    diag = D.diagonal().contiguous()
    slabs = [D[:, r * b:(r + 1) * b].contiguous() for r in range(2)]
    parts = [ops.triplet_loss_slab_fwd(slabs[r], diag, r * b, 10.) for r in range(2)]
    loss = (parts[0] + parts[1]) / (2. * B * (B - 1))
    np.testing.assert_allclose(loss.item(), loss_full.item(), rtol=2e-6)
    sig = [ops.triplet_loss_slab_sig(slabs[r], diag, r * b, 10.) for r in range(2)]
    rowsig = sig[0][0] + sig[1][0]                          # the all-reduce
    for r in range(2):
        g_slab = ops.triplet_loss_slab_bwd(slabs[r], diag, rowsig, sig[r][1], one, r * b, 10.)
        np.testing.assert_allclose(g_slab.cpu().numpy(), g_full[:, r * b:(r + 1) * b].cpu().numpy(), rtol=2e-5, atol=1e-9)
    # and against CPU autograd of the reference formula
    Dc = D.cpu().double().requires_grad_(True)
    O.triplet_loss(Dc).backward()
    np.testing.assert_allclose(g_full.cpu().numpy(), Dc.grad.float().numpy(), rtol=1e-4, atol=1e-8)
