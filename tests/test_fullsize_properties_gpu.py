"""Size-independent properties at BASELINE.json's full sizes (B = 128 pairs, fov 360; the oracle is too slow
there): equivariances and invariances the reference's algorithm guarantees exactly or to rounding."""
import numpy as np
import pytest
import torch

from witw_amd import synth

pytestmark = pytest.mark.gpu
B = 128


@pytest.fixture(scope='module')
def encoders():
    from witw_amd import cvig_fov
    w = synth.fov_dsm_weights(11)
    dev = torch.device('cuda:0')
    return (cvig_fov.FOV_DSM(False, weights=w).to(dev).eval(), cvig_fov.FOV_DSM(True, weights=w).to(dev).eval())


@pytest.fixture(scope='module')
def batch():
    x = torch.from_numpy(synth.normalized_images(12, 0, (B, 3, 128, 512)))
    return x.cuda()


def test_circular_encoder_is_shift_equivariant(encoders, batch):
    """HorizCircPadding (model/cvig_fov.py:212-231) + three 2x2 pools: rolling the polar image by 8k columns
    rolls the overhead embedding by k columns — exactly (same products, same order)."""
    _s, ov_enc = encoders
    with torch.no_grad():
        e0 = ov_enc(batch)
        e1 = ov_enc(torch.roll(batch, shifts=8 * 5, dims=3).contiguous())
    assert e0.shape == (B, 16, 4, 64)
    assert torch.equal(torch.roll(e0, shifts=5, dims=3), e1)


def test_batch_composition_does_not_change_an_embedding(encoders, batch):
    """No BatchNorm in FOV_DSM: sample i's embedding is the same in a batch of 128 (8-wave workgroups) and
    alone (4-wave workgroups); the K order per output is identical, so the match is exact."""
    s_enc, ov_enc = encoders
    with torch.no_grad():
        for enc in (s_enc, ov_enc):
            full = enc(batch)
            for i in (0, 77, 127):
                assert torch.equal(enc(batch[i:i + 1].contiguous())[0], full[i])


def test_match_orientation_tracks_a_rolled_gallery(encoders, batch):
    """correlation (:297-315): rolling an overhead embedding by -s columns moves every arg-max by -s (mod 64)
    and leaves the chord distance unchanged (same window, same products)."""
    from witw_amd import ops
    s_enc, ov_enc = encoders
    with torch.no_grad():
        su, ov = s_enc(batch), ov_enc(batch)
        ori0, d0, s0 = ops.match_fwd(ov, su, want_score=True)
        ori1, d1, s1 = ops.match_fwd(torch.roll(ov, shifts=-9, dims=3).contiguous(), su, want_score=True)
    assert ori0.shape == (B, B)
    assert torch.equal(s0, s1)                                   # the maximal correlation is bit-identical
    same = ((ori0 - 9) % 64) == ori1
    # the only admissible difference: two shifts tie EXACTLY and the first-index rule picks the other one
    assert same.float().mean() > 0.999
    np.testing.assert_allclose(d1[same].cpu().numpy(), d0[same].cpu().numpy(), rtol=0, atol=2e-6)


def test_ranks_invariants_full_batch(encoders, batch):
    from witw_amd import cvig_fov, ops
    s_enc, ov_enc = encoders
    with torch.no_grad():
        su, ov = s_enc(batch), ov_enc(batch)
        _, d = cvig_fov.match(ov, su)
        r = ops.rank_count(d, 0).cpu().numpy()
        loss, r2, _o, _d = cvig_fov.evaluate_global_batch(ov, su, 0)
    assert r.min() >= 1 and r.max() <= B                      # the true match always counts itself
    dn = d.cpu().numpy()
    np.testing.assert_array_equal(r, (dn <= np.diag(dn)[None, :]).sum(0))
    np.testing.assert_array_equal(r2.cpu().numpy(), r)
    v, i = ops.topk_smallest(d, 5)
    assert torch.all(v[:, 1:] >= v[:, :-1])                   # sorted
    np.testing.assert_array_equal(i[:, 0].cpu().numpy(), dn.argmin(0))
    assert abs(loss.item() - cvig_fov.triplet_loss(d).item()) < 1e-6


def test_training_reduces_the_loss():
    """A few Adam steps of the full train loop body (model/cvig_fov.py:444-461) on one fixed batch lower the loss."""
    from witw_amd import cvig_fov
    dev = torch.device('cuda:0')
    w = synth.fov_dsm_weights(13)
    se = cvig_fov.FOV_DSM(False, weights=w).to(dev).train()
    oe = cvig_fov.FOV_DSM(True, weights=w).to(dev).train()
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


@pytest.mark.parametrize('bf16', [False, True])
def test_step_replayed_as_a_hip_graph_equals_the_eager_step(bf16):
    """parallel.CapturedStep: the whole inference step (transforms, both encoders, fused match, loss, rank counts) captured
    in one hipGraph; replays on NEW inputs are bit-identical to the eager launches."""
    from witw_amd import cvig_fov, ops, parallel
    dev = torch.device('cuda:0')
    w = synth.fov_dsm_weights(21)
    se = cvig_fov.FOV_DSM(False, weights=w).to(dev).eval()
    oe = cvig_fov.FOV_DSM(True, weights=w).to(dev).eval()
    mean, std = cvig_fov.Globals.img_mean, cvig_fov.Globals.img_std

    def body(g_raw, o_raw):
        with torch.no_grad():
            s = ops.resize_bilinear(g_raw, (128, 512), mean, std)
            p = ops.polar_transform(ops.resize_bilinear(o_raw, (256, 256), mean, std))
            su = se.forward_bf16(s) if bf16 else se(s)
            ov = oe.forward_bf16(p) if bf16 else oe(p)
            loss, ranks, ori, d = cvig_fov.evaluate_global_batch(ov, su, 0)
            return loss, ranks, ori, d

    def inputs(seed):
        return (torch.from_numpy(synth.images_u8(seed, 1, (4, 3, 224, 224))).to(dev),
                torch.from_numpy(synth.images_u8(seed, 2, (4, 3, 512, 512))).to(dev))
    step = parallel.CapturedStep(body, inputs(5))
    for seed in (6, 7):
        x = inputs(seed)
        got = [t.clone() for t in step(*x)]
        ref = body(*x)
        for a, b in zip(got, ref):
            assert torch.equal(a, b)


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_pair_embedder_equals_the_plain_calls_bitwise(precision):
    """cvig_fov.PairEmbedder (what test() and the validation phase of train() call: model/cvig_fov.py:524-527, :447-449) at the
    reference's small default batches: the two encoders on two streams, the bf16 pair replayed as one hipGraph from the second
    batch of a shape on -- always the same bits as calling the encoders one after the other; a weight update drops the stale
    graph; a batch above the thresholds, a training-mode encoder and a gradient-recording call take the plain path."""
    from witw_amd import cvig_fov
    dev = torch.device('cuda:0')
    w = synth.fov_dsm_weights(33)
    se = cvig_fov.FOV_DSM(False, weights=w).to(dev).eval()
    oe = cvig_fov.FOV_DSM(True, weights=w).to(dev).eval()
    se.precision = oe.precision = precision
    emb = cvig_fov.PairEmbedder(se, oe)

    def batch(seed, B=16, ws=512):
        return (torch.from_numpy(synth.normalized_images(seed, 1, (B, 3, 128, ws))).to(dev),
                torch.from_numpy(synth.normalized_images(seed, 2, (B, 3, 128, 512))).to(dev))
    with torch.no_grad():
        for n, seed in enumerate((1, 2, 3, 4)):
            s, p = batch(seed)
            su, ov = emb(s, p)
            ref_su, ref_ov = se(s), oe(p)
            assert torch.equal(su, ref_su) and torch.equal(ov, ref_ov), (precision, n)
        if precision == 'bf16':
            assert emb.stats['captures'] == 1 and emb.stats['graph_replay'] == 3 and emb.stats['dual_stream'] == 1, emb.stats
        else:
            assert emb.stats['captures'] == 0 and emb.stats['dual_stream'] == 4, emb.stats
        # another shape (the last, ragged batch of an epoch; a narrower field of view) has its own key
        s, p = batch(5, B=5, ws=96)
        su, ov = emb(s, p)
        assert torch.equal(su, se(s)) and torch.equal(ov, oe(p))
        # a weight update (validation after a training epoch): the stale graph must not be replayed
        with torch.no_grad():
            for enc in (se, oe):
                c = cvig_fov._conv_of(enc.model.features[19])
                c.weight.mul_(1.01)
        before = dict(emb.stats)
        for seed in (6, 7, 8):
            s, p = batch(seed)
            su, ov = emb(s, p)
            assert torch.equal(su, se(s)) and torch.equal(ov, oe(p))
        if precision == 'bf16':
            assert emb.stats['captures'] == before['captures'] + 1 and emb.stats['graph_replay'] == before['graph_replay'] + 2
        # above the thresholds: the plain path
        s, p = batch(9, B=72)
        n_eager = emb.stats['eager']
        su, ov = emb(s, p)
        assert emb.stats['eager'] == n_eager + 1 and torch.equal(su, se(s)) and torch.equal(ov, oe(p))
    # gradient-recording call on training-mode encoders: plain path, autograd intact
    se.train()
    oe.train()
    s, p = batch(10, B=4)
    su, ov = emb(s, p)
    assert su.requires_grad and ov.requires_grad
    (su.sum() + ov.sum()).backward()
    assert cvig_fov._conv_of(se.model.features[27]).weight.grad is not None
