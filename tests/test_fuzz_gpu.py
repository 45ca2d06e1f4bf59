"""A bounded, fixed-seed run of tools/fuzz_parity.py inside the suite: every kind of the randomised sweep (fp32 / bf16 / fp16x3 conv
forward, dgrad, wgrad, the 4-tap forms, first layer, resize / polar, fused match direct and spectral, loss, rank counts, the bf16
16x16x32 kernel) drawn ROUNDS times from one seed, each case compared with the CPU oracle through the C ABI."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

ROUNDS = 12


def test_fuzz_parity_fixed_seed_all_kinds():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'fuzz_parity.py')
    spec = importlib.util.spec_from_file_location('fuzz_parity', path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    ran, fails = fz.run(seed=20261004, rounds=ROUNDS)
    assert sorted(ran) == list(range(fz.N_KINDS)) and all(v == ROUNDS for v in ran.values()), ran
    assert not fails, fails[:10]
