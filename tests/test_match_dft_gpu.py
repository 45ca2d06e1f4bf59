"""GPU parity of the spectral match (witw_match_fwd_dft) against the reference goldens, the fp64 direct sum and the direct kernel.

The spectral form computes the same 64 scores per pair through 64-point row DFTs; scores agree with the direct sum to fp32
rounding, so distances are held to the 1e-5 of the direct kernel's tests and orientations must be equal wherever the two best
scores of the pair differ by more than that rounding (on the goldens: everywhere)."""
import os

import numpy as np
import pytest
import torch

from witw_amd import synth

pytestmark = pytest.mark.gpu


def _scores64(ov, su):
    """fp64 score[o,s,shift] of the direct definition (model/cvig_fov.py:297-315), numpy."""
    bo, bs, we = ov.shape[0], su.shape[0], su.shape[3]
    o = ov.reshape(bo, 64, 64).astype(np.float64)
    s = np.zeros((bs, 64, 64))
    s[:, :, :we] = su.reshape(bs, 64, we)
    fo, fs = np.fft.fft(o, axis=2), np.fft.fft(s, axis=2)
    return np.fft.ifft(np.einsum('ocf,scf->osf', fo, np.conj(fs)), axis=2).real      # exact to 1e-13: the oracle of this test


def _check(ov, su, atol=1e-5):
    from witw_amd import ops
    ovc, suc = torch.from_numpy(ov).cuda(), torch.from_numpy(su).cuda()
    ori, dist, score = ops.match_fwd_dft(ovc, suc, want_score=True)
    ori_d, dist_d, score_d = ops.match_fwd(ovc, suc, want_score=True)
    sc = _scores64(ov, su)
    best = sc.max(axis=2)
    scale = np.linalg.norm(ov.reshape(len(ov), -1), axis=1)[:, None] * np.linalg.norm(su.reshape(len(su), -1), axis=1)[None, :]
    # scores of both kernels against fp64, relative to |ov||su| (the normalisation of the distance)
    e_dft = np.abs(score.cpu().numpy() - best) / scale
    e_dir = np.abs(score_d.cpu().numpy() - best) / scale
    assert e_dft.max() < 2e-6, e_dft.max()
    assert e_dft.max() < 4 * max(e_dir.max(), 2e-7)                # no worse than the direct fp32 sum by more than rounding noise
    # distances: 1e-5 wherever both kernels chose the same shift (for We < 64 the window norm, hence the distance, changes
    # with the shift, so a pair whose two best scores tie to rounding may legitimately land on either side)
    same_shift = (ori == ori_d).cpu().numpy()
    np.testing.assert_allclose(dist.cpu().numpy()[same_shift], dist_d.cpu().numpy()[same_shift], rtol=0, atol=atol)
    assert same_shift.mean() > 0.9999
    # orientation: equal to the fp64 arg-max wherever the runner-up is further away than the fp32 rounding of a score
    srt = np.sort(sc, axis=2)
    clear = (srt[:, :, -1] - srt[:, :, -2]) > 8e-6 * scale
    got = ori.cpu().numpy()
    assert np.array_equal(got[clear], sc.argmax(axis=2)[clear])
    # and everywhere: the chosen shift's true score is the maximum to rounding
    chosen = np.take_along_axis(sc, got[:, :, None], axis=2)[:, :, 0]
    assert ((best - chosen) <= 8e-6 * scale).all()
    return float(clear.mean()), float((got == ori_d.cpu().numpy()).mean())


def test_dft_match_matches_reference_goldens(golden_dir):
    from witw_amd import ops
    g = np.load(os.path.join(golden_dir, 'matching.npz'))
    seed = int(g['seed'])
    for tag in 'abcde':
        bo, bs, we = (int(v) for v in g['%s_shape' % tag])
        ov = torch.from_numpy(synth.embeddings(seed, 100 + ord(tag), (bo, 16, 4, 64))).cuda()
        su = torch.from_numpy(synth.embeddings(seed, 200 + ord(tag), (bs, 16, 4, we))).cuda()
        ori, dist = ops.match_fwd_dft(ov, su)
        np.testing.assert_array_equal(ori.cpu().numpy(), g['%s_orientation' % tag])      # the reference's indices
        np.testing.assert_allclose(dist.cpu().numpy(), g['%s_distance' % tag], rtol=0, atol=1e-5)


@pytest.mark.parametrize('shape', [(1, 1, 64), (130, 257, 64), (3, 200, 1), (70, 5, 63), (33, 31, 12), (1101, 130, 64), (517, 529, 40)])
def test_dft_match_ragged_shapes(shape):
    bo, bs, we = shape
    ov = synth.embeddings(11, bo, (bo, 16, 4, 64))
    su = synth.embeddings(12, bs, (bs, 16, 4, we))
    clear, same = _check(ov, su)
    assert clear > 0.99 and same > 0.99


def test_dft_match_true_pairs_and_ties():
    """Queries cut out of gallery rows at a known shift (+ noise): the true shift is found; periodic rows: first index wins."""
    from witw_amd import ops
    rng = np.random.default_rng(5)
    ov = rng.standard_normal((96, 16, 4, 64)).astype(np.float32)
    shifts = rng.integers(0, 64, 96)
    su = np.stack([np.roll(ov[i], -int(shifts[i]), axis=2)[:, :, :48] for i in range(96)]) \
        + 0.5 * rng.standard_normal((96, 16, 4, 48)).astype(np.float32)
    su = su.astype(np.float32)
    _check(ov, su)
    ori, dist = ops.match_fwd_dft(torch.from_numpy(ov).cuda(), torch.from_numpy(su).cuda())
    assert np.array_equal(np.diagonal(ori.cpu().numpy()), shifts)
    assert (dist.argmin(dim=0).cpu().numpy() == np.arange(96)).all()
    # exact ties: small integers, period 16 -> every quantity is exact in fp32 and in the fp32 spectra up to the final sum
    base = rng.integers(-3, 4, (4, 16, 4, 16)).astype(np.float32)
    ovp = np.tile(base, (1, 1, 1, 4))
    sup = np.ascontiguousarray(np.roll(ovp, -5, axis=3)[:, :, :, :32])
    ori = ops.match_fwd_dft(torch.from_numpy(ovp).cuda(), torch.from_numpy(sup).cuda())[0].cpu().numpy()
    assert (np.diagonal(ori) % 16 == 5).all()


def test_dft_match_degenerate_inputs():
    """All-zero embeddings (every shift ties at score 0: torch.argmax's first index, 0, as the direct kernel gives), a zero
    surface against random overheads, and NaN embeddings (no finite maximum: index 0, nothing out of range is read)."""
    from witw_amd import ops
    rng = np.random.default_rng(11)
    z_ov, z_su = torch.zeros((40, 16, 4, 64), device='cuda'), torch.zeros((70, 16, 4, 64), device='cuda')
    ori = ops.match_fwd_dft(z_ov, z_su)[0]
    assert torch.equal(ori, torch.zeros_like(ori)) and torch.equal(ori, ops.match_fwd(z_ov, z_su)[0])
    ov = torch.from_numpy(rng.standard_normal((40, 16, 4, 64)).astype(np.float32)).cuda()
    ori = ops.match_fwd_dft(ov, z_su)[0]
    assert torch.equal(ori, torch.zeros_like(ori))
    nan_su = torch.full((70, 16, 4, 64), float('nan'), device='cuda')
    ori, dist = ops.match_fwd_dft(ov, nan_su)
    torch.cuda.synchronize()
    assert int(ori.min()) >= 0 and int(ori.max()) <= 63
    # half of the surfaces NaN: the finite ones are untouched by their neighbours in the tile
    su = torch.from_numpy(rng.standard_normal((70, 16, 4, 64)).astype(np.float32)).cuda()
    mixed = su.clone()
    mixed[::2] = float('nan')
    o_ref, d_ref = ops.match_fwd_dft(ov, su)
    o_mix, d_mix = ops.match_fwd_dft(ov, mixed)
    assert torch.equal(o_mix[:, 1::2], o_ref[:, 1::2]) and torch.equal(d_mix[:, 1::2], d_ref[:, 1::2])


@pytest.mark.parametrize('shape', [(7, 5, 64), (130, 257, 64), (1100, 260, 64), (1100, 260, 63), (1100, 300, 12), (1100, 300, 31),
                                   (1100, 300, 40), (40, 9, 1), (2100, 70, 16), (128, 128, 64), (128, 128, 12), (128, 128, 30), (100, 77, 64)])
def test_match_pairs_bit_identical_to_match_fwd(shape):
    """witw_match_pairs re-scores single pairs with the arithmetic of the all-pairs kernels: orientation, score and distance
    carry the same BITS as witw_match_fwd's entries, whichever kernel variant the all-pairs launch dispatched to (generic,
    pipelined full-width, 4 / 2 / 1 rows per stage)."""
    from witw_amd import ops
    bo, bs, we = shape
    ov = torch.from_numpy(synth.embeddings(61, bo, (bo, 16, 4, 64))).cuda()
    su = torch.from_numpy(synth.embeddings(62, bs, (bs, 16, 4, we))).cuda()
    ori, dist, score, ws = ops.match_fwd(ov, su, want_score=True, want_workspace=True)
    g = np.random.Generator(np.random.Philox(key=[63, bo * bs]))
    n = min(bo * bs, 3000)
    flat = torch.from_numpy(g.choice(bo * bs, size=n, replace=False)).cuda()
    po, ps = (flat // bs).to(torch.int32).contiguous(), (flat % bs).to(torch.int32).contiguous()
    o2, d2 = ops.match_pairs(ov, su, ws[:bo * 64], ws[bo * 64:bo * 64 + bs], po, ps)
    assert torch.equal(o2, ori.reshape(-1)[flat])
    assert torch.equal(d2, dist.reshape(-1)[flat])                  # bits, not a tolerance
    # and with the norms of a spectral-pass workspace (the re-scoring path of retrieve(method='dft'))
    _, _, wsd = ops.match_fwd_dft(ov, su, want_workspace=True)
    assert torch.equal(wsd[:bo * 64 + bs], ws[:bo * 64 + bs])
    assert torch.equal(ops.match_pairs(ov, su, wsd[:bo * 64], wsd[bo * 64:bo * 64 + bs], po, ps)[1], d2)


def test_rank_count_band_and_gap():
    from witw_amd import ops
    g = np.random.Generator(np.random.Philox(key=[64, 1]))
    d = torch.from_numpy(g.random((1500, 70), dtype=np.float32) * 2).cuda()
    thr = torch.from_numpy(g.random(70, dtype=np.float32) * 2).cuda()
    d[3, 5] = thr[5]
    d[4, 5] = float('nan')
    eps = 3e-3
    c, po, ps = ops.rank_count_band(d, thr, eps)
    lo, hi = thr - eps, thr + eps
    assert torch.equal(c.long(), (d < lo[None, :]).sum(0))
    band = torch.nonzero((d >= lo[None, :]) & (d <= hi[None, :]))
    got = torch.stack((po.long(), ps.long()), 1)
    assert sorted(map(tuple, got.tolist())) == sorted(map(tuple, band.tolist())) and len(got) > 50
    # gap = best - runner-up of the spectral scores
    ov = synth.embeddings(65, 1, (90, 16, 4, 64))
    su = synth.embeddings(66, 1, (50, 16, 4, 24))
    _, _, gap, _ws = ops.match_fwd_dft(torch.from_numpy(ov).cuda(), torch.from_numpy(su).cuda(), want_gap=True)
    sc = np.sort(_scores64(ov, su), axis=2)
    scale = np.linalg.norm(ov.reshape(90, -1), axis=1)[:, None] * np.linalg.norm(su.reshape(50, -1), axis=1)[None, :]
    assert (np.abs(gap.cpu().numpy() - (sc[:, :, -1] - sc[:, :, -2])) <= 4e-6 * scale).all()
    assert (gap >= 0).all()


def _planted(G, Q, we, noise, seed, near_ties=True):
    """Gallery + queries cut out of gallery rows, plus (near_ties) rows that differ from other rows by one last-place unit in
    a handful of entries and exact duplicates: distances that tie to well inside fp32 rounding or exactly."""
    gen = torch.Generator(device='cuda')
    gen.manual_seed(seed)
    gallery = torch.randn((G, 16, 4, 64), generator=gen, device='cuda')
    shifts = torch.randint(0, 64, (Q,), generator=gen, device='cuda')
    col = (torch.arange(we, device='cuda')[None, :] + shifts[:, None]) % 64
    queries = torch.gather(gallery[:Q], 3, col[:, None, None, :].expand(-1, 16, 4, -1)) \
        + noise * torch.randn((Q, 16, 4, we), generator=gen, device='cuda')
    if near_ties:
        n = min(Q, G // 4)
        src = torch.arange(n, device='cuda')
        dst = G - 1 - src
        gallery[dst] = gallery[src]                                  # duplicates of true matches far away in the gallery
        bump = gallery[dst[::2]].clone()
        bump[:, 0, 0, :3] = torch.nextafter(bump[:, 0, 0, :3], torch.full_like(bump[:, 0, 0, :3], 10.0))
        gallery[dst[::2]] = bump                                     # every other one: three entries one ulp up
    return gallery.contiguous(), queries.contiguous()


@pytest.mark.parametrize('case', [(20000, 2000, 64, 10.0), (20000, 2000, 64, 4.0), (6000, 700, 40, 3.0), (6000, 700, 12, 1.5)])
def test_dft_retrieve_equals_direct_retrieve(case):
    """cvig_fov.retrieve(method='dft') against the direct pass: ranks and top-k INDICES are equal (torch.equal), on random
    galleries with planted matches and on planted near-ties / exact duplicates; listed distances agree to the rounding bound.
    Also measures the actual largest |d_dft - d_direct| against that bound."""
    from witw_amd import cvig_fov, ops
    G, Q, we, noise = case
    gallery, queries = _planted(G, Q, we, noise, seed=7 + we)
    r0, v0, i0 = cvig_fov.retrieve(gallery, queries, k=10, query_chunk=1024)
    r1, v1, i1 = cvig_fov.retrieve(gallery, queries, k=10, query_chunk=1024, method='dft')
    st = dict(cvig_fov.retrieve.last_stats)
    np.testing.assert_array_equal(r1, r0)
    assert torch.equal(i1, i0)
    assert float((v1 - v0).abs().max()) <= st['eps']
    assert (i0[:min(Q, G // 4), :2] >= 0).all() and st['rescored_topk'] > 0 and st['rescored_rank'] > 0
    frac = (st['rescored_rank'] + st['rescored_topk'] + st['rescored_true'] + st['rescored_orientation']) / st['pairs']
    print('G=%d Q=%d We=%d: re-scored %.1f pairs per million, %d fallback queries, eps %.2e' % (G, Q, we, frac * 1e6, st['fallback_queries'], st['eps']))
    assert frac < (2e-3 if G >= 20000 else 1e-2) and st['fallback_queries'] <= Q // 50      # every query has a planted tie here
    # the bound itself: spectral vs direct distances over a block of the matrix (same shift chosen)
    ov, su = gallery[:4000], queries[:512]
    o_d, d_d = ops.match_fwd(ov, su)
    o_s, d_s = ops.match_fwd_dft(ov, su)
    same = o_d == o_s
    worst = float((d_d - d_s)[same].abs().max())
    print('largest |d_dft - d_direct| over %d pairs: %.2e (bound %.2e)' % (int(same.sum()), worst, st['eps']))
    assert worst <= 0.5 * st['eps']
    assert float(same.float().mean()) > 0.999


def test_spectral_retrieve_plain_random_gallery_is_cheap():
    """No planted ties: a handful of pairs per million need re-scoring and no query falls back to the direct pass."""
    from witw_amd import cvig_fov
    gallery, queries = _planted(30000, 1500, 64, 10.0, seed=99, near_ties=False)
    r0, v0, i0 = cvig_fov.retrieve(gallery, queries, k=10)
    r1, v1, i1 = cvig_fov.retrieve(gallery, queries, k=10, method='dft')
    st = cvig_fov.retrieve.last_stats
    np.testing.assert_array_equal(r1, r0)
    assert torch.equal(i1, i0) and st['fallback_queries'] == 0
    frac = (st['rescored_rank'] + st['rescored_topk'] + st['rescored_true']) / st['pairs']
    print('random gallery 30000 x 1500: re-scored %.1f pairs per million' % (frac * 1e6))
    assert frac < 5e-4


def test_evaluation_ranks_through_the_spectral_match():
    """test()'s ranking (model/cvig_fov.py:543-552) through the spectral pass, narrow surfaces (fov < 355): the ranks equal the
    direct kernel's exactly -- pairs whose two best shifts tie to rounding are re-scored before anything is counted."""
    from witw_amd import cvig_fov
    gen = torch.Generator(device='cuda')
    gen.manual_seed(11)
    n = 700
    ov = torch.randn((n, 16, 4, 64), generator=gen, device='cuda')
    shifts = torch.randint(0, 64, (n,), generator=gen, device='cuda')
    col = (torch.arange(40, device='cuda')[None, :] + shifts[:, None]) % 64
    su = torch.gather(ov, 3, col[:, None, None, :].expand(-1, 16, 4, -1)) + 3.0 * torch.randn((n, 16, 4, 40), generator=gen, device='cuda')
    r_direct = cvig_fov.ranks(ov, su)
    r_dft = cvig_fov.retrieve(ov, su, k=5, method='dft')[0]
    np.testing.assert_array_equal(r_dft, r_direct)


@pytest.mark.parametrize('we', [64, 24])
def test_evaluation_ranks_auto_switches_to_the_spectral_pass(we, monkeypatch):
    """cvig_fov.evaluation_ranks (what test() tabulates): 'auto' takes the direct kernel below SPECTRAL_FROM pairs and the
    index-exact spectral pass from there on; all three settings give the same ranks, near-ties and duplicates included."""
    from witw_amd import cvig_fov
    n = 2500
    gallery, queries = _planted(n, n, we, 6.0, 23)
    r_direct = cvig_fov.evaluation_ranks(gallery, queries, method='direct')
    np.testing.assert_array_equal(r_direct, cvig_fov.ranks(gallery, queries))
    r_dft = cvig_fov.evaluation_ranks(gallery, queries, method='dft')
    np.testing.assert_array_equal(r_dft, r_direct)
    assert cvig_fov.retrieve.last_stats['method'] == 'dft'
    cvig_fov.retrieve.last_stats = {}
    np.testing.assert_array_equal(cvig_fov.evaluation_ranks(gallery, queries, method='auto'), r_direct)
    assert cvig_fov.retrieve.last_stats == {}                     # below the switch: the direct kernel, retrieve() not involved
    monkeypatch.setattr(cvig_fov, 'SPECTRAL_FROM', 2000)
    np.testing.assert_array_equal(cvig_fov.evaluation_ranks(gallery, queries, method='auto'), r_direct)
    assert cvig_fov.retrieve.last_stats.get('method') == 'dft'
    with pytest.raises(Exception):
        cvig_fov.evaluation_ranks(gallery, queries, method='fft')


@pytest.mark.parametrize('k', [26, 27, 32])
def test_dft_retrieve_large_k_equals_direct(k):
    """k + DFT_MARGIN <= 32 keeps the candidate margin inside the 32-wide device list; beyond that (k = 27..32) the top-k comes
    from the direct pass and only the rank counts from the spectral one. Planted near-ties around every place, so a query whose
    place k+1 sits within rounding of place k cannot be decided from the list alone."""
    from witw_amd import cvig_fov
    gallery, queries = _planted(6000, 300, 64, 4.0, seed=300 + k)
    # 40 one-ulp neighbours of each of the first 8 queries' true matches: near-ties at every place up to 40
    for q in range(8):
        rows = torch.arange(100 + 40 * q, 140 + 40 * q, device='cuda')
        gallery[rows] = gallery[q].clone()
        gallery[rows[::2], 1, 1, :2] = torch.nextafter(gallery[rows[::2], 1, 1, :2], torch.full((20, 2), 10.0, device='cuda'))
    r0, v0, i0 = cvig_fov.retrieve(gallery, queries, k=k, query_chunk=128)
    r1, v1, i1 = cvig_fov.retrieve(gallery, queries, k=k, query_chunk=128, method='dft')
    np.testing.assert_array_equal(r1, r0)
    assert torch.equal(i1, i0)
    assert float((v1 - v0).abs().max()) <= 1e-5


def test_spectra_refuse_the_wrong_side_and_misaligned_slices():
    """ops.Spectra records what the stored layout depends on (csrc/match_dft.hip:331: the side, and for overheads bit 4 of the
    row index): a fov-360 surface embedding is 64 columns wide like an overhead and must not pass as one; slices of overhead
    spectra are defined at multiples of 32 rows only."""
    from witw_amd import _lib, ops
    ov = torch.from_numpy(synth.embeddings(71, 1, (96, 16, 4, 64))).cuda()
    su = torch.from_numpy(synth.embeddings(72, 1, (40, 16, 4, 64))).cuda()
    s_ov, s_su = ops.match_spectrum(ov, overhead=True), ops.match_spectrum(su, overhead=False)
    assert isinstance(s_ov, ops.Spectra) and s_ov.overhead and not s_su.overhead
    ori, d = ops.match_fwd_dft(ov, su, spec_ov=s_ov, spec_su=s_su)
    with pytest.raises(TypeError):
        ops.match_spectrum(ov)                                     # the side is not guessed from the width
    with pytest.raises(_lib.WitwError):
        ops.match_fwd_dft(ov, su, spec_ov=s_ov.data, spec_su=s_su)          # a bare tensor carries no side
    with pytest.raises(_lib.WitwError):
        ops.match_fwd_dft(ov[:40], su, spec_ov=ops.match_spectrum(su, overhead=False), spec_su=s_su)      # surface spectra as gallery
    with pytest.raises(_lib.WitwError):
        ops.match_fwd_dft(su, ov, spec_ov=ops.match_spectrum(su, overhead=True), spec_su=s_ov)          # overhead spectra as queries
    with pytest.raises(_lib.WitwError):
        s_ov.rows(16, 32)
    with pytest.raises(_lib.WitwError):
        ops.Spectra.cat([s_ov.rows(0, 48), s_ov.rows(64, 32)])     # a 48-row block in front shifts the next rows' index & 16
    # aligned slices and concatenations are the spectra of the same rows
    part = s_ov.rows(32, 64)
    o2, d2 = ops.match_fwd_dft(ov[32:96].contiguous(), su, spec_ov=part, spec_su=s_su)
    assert torch.equal(o2, ori[32:96]) and torch.equal(d2, d[32:96])
    both = ops.Spectra.cat([s_ov.rows(64, 32), s_ov.rows(0, 32)])
    o3, d3 = ops.match_fwd_dft(torch.cat((ov[64:96], ov[0:32])).contiguous(), su, spec_ov=both, spec_su=s_su)
    assert torch.equal(o3, torch.cat((ori[64:96], ori[0:32]))) and torch.equal(d3, torch.cat((d[64:96], d[0:32])))
    # surface spectra have one layout whatever the row
    o4, d4 = ops.match_fwd_dft(ov, su[7:30].contiguous(), spec_ov=s_ov, spec_su=s_su.rows(7, 23))
    assert torch.equal(o4, ori[:, 7:30]) and torch.equal(d4, d[:, 7:30])
