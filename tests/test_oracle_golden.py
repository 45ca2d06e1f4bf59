"""Pin the CPU oracle (oracle/cvig_fov_oracle.py) to outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/gen_golden.py from /root/reference)."""
import os

import numpy as np
import torch

from oracle import cvig_fov_oracle as O
from witw_amd import synth


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _tw(weights):
    return {k: (torch.from_numpy(w), torch.from_numpy(b)) for k, (w, b) in weights.items()}


def test_polar_transform_bit_exact(golden_dir):
    g = _load(golden_dir, 'polar.npz')
    img = torch.from_numpy(synth.normalized_images(int(g['seed']), int(g['stream']), (3, 256, 256)))
    out = O.polar_transform(img)
    assert out.shape == (3, 128, 512) and out.dtype == torch.float32
    np.testing.assert_array_equal(out[:, ::8, :].numpy(), g['polar_rows'])
    assert out.double().sum().item() == float(g['polar_sum'])
    assert out.double().abs().sum().item() == float(g['polar_abs_sum'])
    # the two samples whose clipped taps have all-zero weights (reference behaviour, kept)
    for (y, x), v in zip(g['zero_taps'], g['zero_vals']):
        np.testing.assert_array_equal(out[:, y, x].numpy(), v)
        assert np.all(v == 0)


def test_normalization_bit_exact(golden_dir):
    g = _load(golden_dir, 'normalize.npz')
    raw5 = torch.from_numpy(g['raw5'])
    np.testing.assert_array_equal(O.image_normalization(raw5[:3]).numpy(), g['norm3'])
    np.testing.assert_array_equal(O.image_normalization_semantic(raw5).numpy(), g['norm5'])


def test_encoder_eval_matches_reference(golden_dir):
    g = _load(golden_dir, 'encoder.npz')
    seed = int(g['seed'])
    w = _tw(synth.fov_dsm_weights(seed))
    x360 = torch.from_numpy(synth.normalized_images(seed, 10, (2, 3, 128, 512)))
    x70 = torch.from_numpy(synth.normalized_images(seed, 11, (2, 3, 128, 99)))
    with torch.no_grad():
        for circ in (False, True):
            e360 = O.fov_dsm_forward(x360, w, circ).numpy()
            e70 = O.fov_dsm_forward(x70, w, circ).numpy()
            assert e360.shape == (2, 16, 4, 64) and e70.shape == (2, 16, 4, 12)
            np.testing.assert_allclose(e360, g['embed360_circ%d' % circ], rtol=0, atol=1e-6)
            np.testing.assert_allclose(e70, g['embed70_circ%d' % circ], rtol=0, atol=1e-6)


def test_encoder_train_mode_injected_dropout(golden_dir):
    g = _load(golden_dir, 'encoder.npz')
    seed = int(g['seed'])
    w = _tw(synth.fov_dsm_weights(seed))
    x360 = torch.from_numpy(synth.normalized_images(seed, 10, (2, 3, 128, 512)))
    scales = {i: torch.from_numpy(g['drop_scale_%d' % i]) for i in (17, 19, 21)}
    assert all(0 < (s == 0).float().mean() < 0.5 for s in scales.values())
    with torch.no_grad():
        e = O.fov_dsm_forward(x360, w, True, dropout_scales=scales).numpy()
    np.testing.assert_allclose(e, g['embed360_circ1_train'], rtol=0, atol=1e-6)


def test_encoder_semantic(golden_dir):
    g = _load(golden_dir, 'encoder_semantic.npz')
    seed = int(g['seed'])
    w = _tw(synth.fov_dsm_weights(seed, in_channels=5))
    x5 = torch.from_numpy(synth.normalized_images(seed, 12, (1, 5, 128, 512)))
    with torch.no_grad():
        e = O.fov_dsm_forward(x5, w, True).numpy()
    np.testing.assert_allclose(e, g['embed5_circ1'], rtol=0, atol=1e-6)
    assert 'model.features.0.layer.weight' in set(g['trainable'])  # layer 0 unfrozen, cvig_semantic.py:308


def test_matching_and_loss(golden_dir):
    g = _load(golden_dir, 'matching.npz')
    seed = int(g['seed'])
    for tag in 'abcde':
        bo, bs, we = (int(v) for v in g['%s_shape' % tag])
        ov = torch.from_numpy(synth.embeddings(seed, 100 + ord(tag), (bo, 16, 4, 64))).requires_grad_(True)
        su = torch.from_numpy(synth.embeddings(seed, 200 + ord(tag), (bs, 16, 4, we))).requires_grad_(True)
        ori, dist = O.match(ov, su)
        assert ori.dtype == torch.int64
        np.testing.assert_array_equal(ori.numpy(), g['%s_orientation' % tag])
        np.testing.assert_allclose(dist.detach().numpy(), g['%s_distance' % tag], rtol=0, atol=2e-6)
        ori_f, dist_f = O.match_fused(ov.detach(), su.detach())
        np.testing.assert_array_equal(ori_f.numpy(), g['%s_orientation' % tag])
        np.testing.assert_allclose(dist_f.numpy(), g['%s_distance' % tag], rtol=0, atol=2e-6)
        if tag == 'c':
            np.testing.assert_array_equal(O.crop_overhead(ov.detach(), ori, we).numpy(), g['c_crop'])
        if bo == bs:
            loss = O.triplet_loss(dist)
            loss.backward()
            np.testing.assert_allclose(loss.item(), float(g['%s_loss' % tag]), rtol=1e-6)
            np.testing.assert_allclose(ov.grad.numpy(), g['%s_grad_ov' % tag], rtol=0, atol=1e-6)
            np.testing.assert_allclose(su.grad.numpy(), g['%s_grad_su' % tag], rtol=0, atol=1e-6)
    dm = torch.from_numpy(g['loss_in'])
    assert O.triplet_loss(dm).item() == float(g['loss_a10'])
    assert O.triplet_loss(dm, alpha=3.).item() == float(g['loss_a3'])


def test_ranking(golden_dir):
    g = _load(golden_dir, 'ranking.npz')
    seed = int(g['seed'])
    for tag in ('r360', 'r70'):
        n, we = (int(v) for v in g['%s_n_we' % tag])
        ov = torch.from_numpy(synth.embeddings(seed, 400 + we, (n, 16, 4, 64)))
        noise = torch.from_numpy(synth.embeddings(seed, 500 + we, (n, 16, 4, we)))
        shifts = g['%s_shifts' % tag]
        su = torch.stack([torch.roll(ov[i], -int(shifts[i]), dims=2)[:, :, :we] for i in range(n)]) + float(g['%s_noise' % tag]) * noise
        r = O.ranks(ov, su)
        np.testing.assert_array_equal(r, g['%s_ranks' % tag])
        t = O.recall_table(r)
        np.testing.assert_allclose([t['top_1'], t['top_5'], t['top_10'], t['top_1pct'], t['mean'], t['median']],
                                   g['%s_table' % tag])


def _baseline_params(seed):
    return [{k: torch.from_numpy(v) for k, v in q.items()} for q in synth.baseline_params(seed)]


def test_baseline_oracle_matches_reference(golden_dir):
    from oracle import cvig_baseline_oracle as OB
    g = _load(golden_dir, 'baseline.npz')
    seed = int(g['seed'])
    for tag, hw, stream, off in (('surface', 500, 30, 0), ('overhead', 512, 31, 1)):
        x = torch.from_numpy(synth.images_u8(seed, stream, (2, 3, hw, hw)))
        with torch.no_grad():
            e = OB.encoder_forward(x, _baseline_params(seed + off)).numpy()
        assert e.shape == (2, 1536)
        np.testing.assert_allclose(e, g['embed_' + tag], rtol=0, atol=1e-6)
    e1 = torch.from_numpy(synth.embeddings(seed, 600, (5, 1536))) * 0.018
    e2 = e1 + torch.from_numpy(synth.embeddings(seed, 601, (5, 1536))) * 0.02
    assert OB.exhaustive_minibatch_triplet_loss(e1, e2).item() == float(g['loss_hard'])
    assert OB.exhaustive_minibatch_triplet_loss(e1, e2, soft_margin=True).item() == float(g['loss_soft'])
    assert OB.exhaustive_minibatch_triplet_loss(e1 * 0.55, e2 * 0.55, margin=0.3).item() == float(g['loss_hard_m03'])
    assert OB.exhaustive_minibatch_triplet_loss(e1, e2, soft_margin=True, alpha=2.).item() == float(g['loss_soft_a2'])
    ov = torch.from_numpy(synth.embeddings(seed, 602, (14, 1536)))
    su = ov + 14.0 * torch.from_numpy(synth.embeddings(seed, 603, (14, 1536)))
    np.testing.assert_array_equal(OB.ranks(ov, su), g['ranks'])


def _baseline_train_case(golden_dir):
    g = _load(golden_dir, 'baseline_train.npz')
    seed, B = int(g['seed']), int(g['B'])
    xs = torch.from_numpy(synth.images_u8(seed, 40, (B, 3, 400, 400)))
    xo = torch.from_numpy(synth.images_u8(seed, 41, (B, 3, 416, 416)))
    return g, xs, xo, seed


def test_baseline_oracle_train_step_matches_reference(golden_dir):
    """Oracle train-mode forward + torch autograd + Adam against the reference's loop body (BatchNorm batch statistics)."""
    from oracle import cvig_baseline_oracle as OB
    g, xs, xo, seed = _baseline_train_case(golden_dir)
    ps, po = _baseline_params(seed + 10), _baseline_params(seed + 11)
    leaves = []
    for prm in (ps, po):
        for q in prm:
            for k in ('w', 'b', 'gamma', 'beta'):
                q[k].requires_grad_(True)
                leaves.append(q[k])
    es = OB.encoder_forward(xs, ps, train=True)
    eo = OB.encoder_forward(xo, po, train=True)
    loss = OB.exhaustive_minibatch_triplet_loss(es, eo)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(g['loss']), rtol=1e-6)
    np.testing.assert_allclose(es.detach().numpy(), g['embed_surface'], atol=1e-6)
    key = {'w': 'conv%d.weight', 'b': 'conv%d.bias', 'gamma': 'bn%d.weight', 'beta': 'bn%d.bias'}
    for tag, prm in (('surface', ps), ('overhead', po)):
        for i, q in enumerate(prm, 1):
            for k, pat in key.items():
                name = '%s.%s' % (tag, pat % i)
                gr = q[k].grad.reshape(-1)
                np.testing.assert_allclose(gr.double().norm().item(), float(g['gnorm:' + name]), rtol=1e-4)
            np.testing.assert_allclose(q['mean'].numpy(), g['buf:%s.bn%d.running_mean' % (tag, i)], atol=1e-6)
            np.testing.assert_allclose(q['var'].numpy(), g['buf:%s.bn%d.running_var' % (tag, i)], rtol=1e-5, atol=1e-7)
