"""cvig_baseline's host-side augmentation helpers against the reference's outputs (tests/golden/augment.npz, written by
gen_golden.py --augment from model/cvig_baseline.py:97-128): every unit alias, half-to-even rounding, negative and
beyond-a-turn shifts, the error for an unknown unit, quarter turns for factors -2..6 on a non-square image."""
import os

import numpy as np
import pytest
import torch

from witw_amd import cvig_baseline as cb


def test_horizontal_shift_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'augment.npz'))
    img = torch.from_numpy(g['img'])
    for i, (u, s) in enumerate(zip(g['units'], g['shifts'])):
        s = float(s)
        for shift in ((int(s), s) if s == int(s) else (s,)):          # integers as the reference's callers pass them, and as floats
            got = cb.horizontal_shift(img, shift, unit=str(u))
            np.testing.assert_array_equal(got.numpy(), g['shift_%d' % i], err_msg='%s %r' % (u, shift))
    with pytest.raises(Exception) as e:
        cb.horizontal_shift(img, 1, unit='turns')
    assert str(e.value) == str(g['bad_unit_message'])
    for bad in ('', 'px', 'deg', 'pixelss'):
        with pytest.raises(Exception):
            cb.horizontal_shift(img, 1, unit=bad)


def test_quantized_rotation_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'augment.npz'))
    rect = torch.from_numpy(g['rect'])
    for f in range(-2, 7):
        got = cb.quantized_rotation(rect, f)
        assert tuple(got.shape) == g['rot_%d' % f].shape
        np.testing.assert_array_equal(got.numpy(), g['rot_%d' % f])
    batch = rect.unsqueeze(0).repeat(2, 1, 1, 1)
    np.testing.assert_array_equal(cb.quantized_rotation(batch, 3)[1].numpy(), g['rot_3'])
