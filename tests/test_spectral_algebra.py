"""CPU check of the algebra behind witw_match_fwd_dft (33 frequency slots, even/odd split, coefficient table): the numpy fp64
restatement in oracle/ against the reference's own conv2d scores on the golden matching cases."""
import os

import numpy as np
import pytest
import torch

from oracle import cvig_fov_oracle as O
from witw_amd import synth


@pytest.mark.parametrize('tag', list('abcde'))
def test_spectral_scores_equal_the_direct_correlation(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, 'matching.npz'))
    seed = int(g['seed'])
    bo, bs, we = (int(v) for v in g['%s_shape' % tag])
    ov = torch.from_numpy(synth.embeddings(seed, 100 + ord(tag), (bo, 16, 4, 64)))
    su = torch.from_numpy(synth.embeddings(seed, 200 + ord(tag), (bs, 16, 4, we)))
    direct = O.correlation_scores(ov.double(), su.double()).numpy()
    spec = O.spectral_scores(ov, su)
    scale = np.linalg.norm(ov.reshape(bo, -1), axis=1)[:, None, None] * np.linalg.norm(su.reshape(bs, -1), axis=1)[None, :, None]
    assert np.abs(spec - direct).max() / scale.max() < 5e-7          # fp32 rounding of the spectra only
    np.testing.assert_array_equal(spec.argmax(axis=2), g['%s_orientation' % tag])   # the reference's orientations
