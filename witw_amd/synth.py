"""Seeded synthetic weights and inputs (no torch RNG, no network).

The reference starts from torch-hub VGG16 weights (model/cvig_fov.py:256) which cannot be
fetched here; every parity / perf run instead regenerates the same arrays on both boxes from
numpy's counter-based Philox generator. Shapes follow the reference network exactly
(SURVEY.md Appendix A).
"""
import math

import numpy as np

# (features index, cin, cout, stride_h, relu, pool_after, dropout) for FOV_DSM
# reference: model/cvig_fov.py:256-288 (VGG16 features[:23] + 3 extra convs).
FOV_LAYERS = [
    (0, 3, 64, 1, True, False, False),
    (2, 64, 64, 1, True, True, False),
    (5, 64, 128, 1, True, False, False),
    (7, 128, 128, 1, True, True, False),
    (10, 128, 256, 1, True, False, False),
    (12, 256, 256, 1, True, False, False),
    (14, 256, 256, 1, True, True, False),
    (17, 256, 512, 1, True, False, True),
    (19, 512, 512, 1, True, False, True),
    (21, 512, 512, 1, True, False, True),
    (23, 512, 256, 2, True, False, False),
    (25, 256, 64, 2, True, False, False),
    (27, 64, 16, 1, False, False, False),
]
TRAINABLE_FROM = 17  # model/cvig_fov.py:275-278


def _rng(seed, stream):
    return np.random.Generator(np.random.Philox(key=[int(seed), int(stream)]))


def fov_dsm_weights(seed, in_channels=3):
    """dict {features_index: (weight[cout,cin,3,3] f32, bias[cout] f32)}.

    VGG-derived layers (<23): He-normal weights, small uniform bias (stand-in for the
    pretrained values). Extra layers: xavier_uniform weights / zero bias, the reference's
    own init (model/cvig_fov.py:263-272). in_channels=5 gives the semantic variant's first
    layer (model/cvig_semantic.py:301-304).
    """
    out = {}
    for (idx, cin, cout, _sh, _relu, _pool, _drop) in FOV_LAYERS:
        if idx == 0:
            cin = in_channels
        g = _rng(seed, idx)
        if idx < 23:
            std = math.sqrt(2.0 / (9 * cin))
            w = g.standard_normal((cout, cin, 3, 3), dtype=np.float32) * np.float32(std)
            b = (g.random((cout,), dtype=np.float32) - np.float32(0.5)) * np.float32(0.1)
        else:
            bound = math.sqrt(6.0 / (9 * cin + 9 * cout))
            w = (g.random((cout, cin, 3, 3), dtype=np.float32) * np.float32(2.0) - np.float32(1.0)) * np.float32(bound)
            b = np.zeros((cout,), dtype=np.float32)
        out[idx] = (np.ascontiguousarray(w, dtype=np.float32), b.astype(np.float32))
    return out


def images_u8(seed, stream, shape):
    """uint8-valued uniform [0,255] image batch as float32 (what io.imread + astype gives,
    model/cvig_fov.py:88-91)."""
    g = _rng(seed, 1000 + stream)
    return g.integers(0, 256, size=shape, dtype=np.uint8).astype(np.float32)


def normalized_images(seed, stream, shape):
    """Roughly ImageNet-normalised inputs (what ImageNormalization emits), uniform in
    [-2.1, 2.6]."""
    g = _rng(seed, 2000 + stream)
    return (g.random(shape, dtype=np.float32) * np.float32(4.7) - np.float32(2.1)).astype(np.float32)


def embeddings(seed, stream, shape):
    g = _rng(seed, 3000 + stream)
    return g.standard_normal(shape, dtype=np.float32)


def dropout_scales(seed, stream, batch, channels, p=0.2):
    """Dropout2d channel scales (0 or 1/(1-p)) as an injectable [B,C] mask
    (model/cvig_fov.py:241,288: whole-channel dropout, p=0.2)."""
    g = _rng(seed, 4000 + stream)
    keep = g.random((batch, channels)) >= p
    return (keep.astype(np.float32) / np.float32(1.0 - p)).astype(np.float32)


BASELINE_WIDTHS = [3, 64, 128, 256, 512, 512, 512, 512]   # model/cvig_baseline.py:240-253


def baseline_params(seed, bands=3):
    """7 x {w [co,ci,4,4], b, gamma, beta, mean, var} for SurfaceEncoder/OverheadEncoder: the reference's
    init distributions (model/cvig_baseline.py:255-262) plus non-trivial BatchNorm running statistics."""
    out = []
    widths = [bands] + BASELINE_WIDTHS[1:]
    for i in range(7):
        g = _rng(seed, 5000 + i)
        ci, co = widths[i], widths[i + 1]
        out.append({
            'w': (g.standard_normal((co, ci, 4, 4), dtype=np.float32) * np.float32(0.02)).astype(np.float32),
            'b': (g.standard_normal((co,), dtype=np.float32) * np.float32(0.02)).astype(np.float32),
            'gamma': (np.float32(1.0) + g.standard_normal((co,), dtype=np.float32) * np.float32(0.02)).astype(np.float32),
            'beta': (g.standard_normal((co,), dtype=np.float32) * np.float32(0.02)).astype(np.float32),
            'mean': (g.standard_normal((co,), dtype=np.float32) * np.float32(0.05)).astype(np.float32),
            'var': (np.float32(0.05) + g.random((co,), dtype=np.float32) * np.float32(0.2)).astype(np.float32),
        })
    return out
