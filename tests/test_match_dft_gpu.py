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


def test_dft_retrieve_equals_direct_retrieve():
    """cvig_fov.retrieve(method='dft') against the direct pass on structured queries: identical ranks and top-k indices."""
    from witw_amd import cvig_fov
    gen = torch.Generator(device='cuda')
    gen.manual_seed(7)
    G, Q = 3000, 500
    gallery = torch.randn((G, 16, 4, 64), generator=gen, device='cuda')
    shifts = torch.randint(0, 64, (Q,), generator=gen, device='cuda')
    col = (torch.arange(64, device='cuda')[None, :] + shifts[:, None]) % 64
    queries = torch.gather(gallery[:Q], 3, col[:, None, None, :].expand(-1, 16, 4, -1)) \
        + 4.0 * torch.randn((Q, 16, 4, 64), generator=gen, device='cuda')
    r0, v0, i0 = cvig_fov.retrieve(gallery, queries, k=10, query_chunk=200)
    r1, v1, i1 = cvig_fov.retrieve(gallery, queries, k=10, query_chunk=200, method='dft')
    np.testing.assert_allclose(v1.cpu().numpy(), v0.cpu().numpy(), rtol=0, atol=1e-5)
    # indices may swap only between candidates whose distances differ by less than the fp32 rounding of a distance
    i0n, i1n, v0n = i0.cpu().numpy(), i1.cpu().numpy(), v0.cpu().numpy()
    diff = i0n != i1n
    if diff.any():
        gaps = np.abs(np.diff(v0n, axis=1))
        near = np.zeros_like(diff)
        near[:, :-1] |= gaps < 2e-6
        near[:, 1:] |= gaps < 2e-6
        assert (near | ~diff).all()
    assert diff.mean() < 0.01
    assert (np.abs(r0 - r1) <= 1).all() and (r0 != r1).mean() < 0.01


def test_evaluation_ranks_through_the_spectral_match():
    """cvig_fov.sharded_ranks takes the matching pass as a callable: with ops.match_fwd_dft the evaluation ranks of test()
    (model/cvig_fov.py:543-552) equal those of the direct kernel on structured pairs."""
    from witw_amd import cvig_fov, ops
    gen = torch.Generator(device='cuda')
    gen.manual_seed(11)
    n = 700
    ov = torch.randn((n, 16, 4, 64), generator=gen, device='cuda')
    shifts = torch.randint(0, 64, (n,), generator=gen, device='cuda')
    col = (torch.arange(40, device='cuda')[None, :] + shifts[:, None]) % 64
    su = torch.gather(ov, 3, col[:, None, None, :].expand(-1, 16, 4, -1)) + 3.0 * torch.randn((n, 16, 4, 40), generator=gen, device='cuda')
    r_direct = cvig_fov.ranks(ov, su)
    r_dft = cvig_fov.sharded_ranks(ov, su, 0, _match=ops.match_fwd_dft)
    assert (np.abs(r_direct - r_dft) <= 1).all() and (r_direct != r_dft).mean() < 0.01
