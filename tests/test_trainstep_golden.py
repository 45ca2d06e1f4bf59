"""The oracle's training step against the reference's own loop body (tests/golden/trainstep.npz)."""
import os

import numpy as np
import torch

from oracle import cvig_fov_oracle as O
from witw_amd import synth


def ref_key(tag, idx, circ, kind):
    """state-dict style name used by the reference for features[idx]."""
    mid = '.layer' if idx in (17, 19, 21) else ''
    if circ:
        mid += '.layer'
    return '%s.model.features.%d%s.%s' % (tag, idx, mid, kind)


def load_case(golden_dir):
    g = np.load(os.path.join(golden_dir, 'trainstep.npz'))
    seed, B, ws = int(g['seed']), int(g['B']), int(g['ws'])
    xs = torch.from_numpy(synth.normalized_images(seed, 20, (B, 3, 128, ws)))
    xo = torch.from_numpy(synth.normalized_images(seed, 21, (B, 3, 128, 512)))
    w = synth.fov_dsm_weights(seed + 1)
    drops = {t: {i: torch.from_numpy(g['drop_%s_%d' % (t, i)]) for i in (17, 19, 21)} for t in 'so'}
    return g, xs, xo, w, drops


def sample(t):
    t = t.reshape(-1)
    return t[::max(1, t.numel() // 257)]


def test_oracle_train_step_matches_reference(golden_dir):
    g, xs, xo, w, drops = load_case(golden_dir)
    ws_ = {k: (torch.from_numpy(a.copy()), torch.from_numpy(b.copy())) for k, (a, b) in w.items()}
    wo_ = {k: (torch.from_numpy(a.copy()), torch.from_numpy(b.copy())) for k, (a, b) in w.items()}
    loss, ori, dist, grads = O.train_step(xs, xo, ws_, wo_, drops['s'], drops['o'], lr=1.E-5)
    np.testing.assert_allclose(loss.item(), float(g['loss']), rtol=1e-6)
    np.testing.assert_array_equal(ori.numpy(), g['orientation'])
    np.testing.assert_allclose(dist.numpy(), g['distance'], atol=2e-6)
    assert len(g['names']) == 24
    for tag, wd in (('s', ws_), ('o', wo_)):
        for idx in O.TRAINABLE:
            for k, kind in ((0, 'weight'), (1, 'bias')):
                name = ref_key(tag, idx, tag == 'o', kind)
                gr = grads[(tag, idx)][k]
                np.testing.assert_allclose(gr.double().norm().item(), float(g['gnorm:' + name]), rtol=1e-5)
                np.testing.assert_allclose(sample(gr).numpy(), g['gsamp:' + name], rtol=1e-4, atol=1e-9)
                np.testing.assert_allclose(sample(wd[idx][k]).numpy(), g['psamp:' + name], rtol=0, atol=1e-7)
