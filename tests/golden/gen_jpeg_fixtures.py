#!/usr/bin/env python
"""Writes tests/golden/jpeg/*.jpg (small JPEG files: data) and expected.npz = what THIS container's Pillow (libjpeg-turbo, the
decoder behind the reference's skimage.io.imread, model/cvig_fov.py:88-89) decodes them to. The files cover 4:2:0 / 4:2:2 /
4:4:4 / grayscale, odd sizes (partial MCUs, one-column chroma), qualities 10-100, optimised Huffman tables, restart intervals,
and one progressive file (left to the host decoder).   python tests/golden/gen_jpeg_fixtures.py"""
import io
import os

import numpy as np
from PIL import Image

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'jpeg')

CASES = [  # name, H, W, subsampling (0 4:4:4, 1 4:2:2, 2 4:2:0, None gray), quality, extra save options
    ('s420_q90', 64, 96, 2, 90, {}), ('s420_odd', 37, 51, 2, 75, {}), ('s420_tiny', 1, 1, 2, 90, {}), ('s420_col', 40, 1, 2, 85, {}),
    ('s422_q50', 33, 70, 1, 50, {}), ('s444_q100', 24, 40, 0, 100, {}), ('s444_q10', 50, 50, 0, 10, {}),
    ('s420_opt', 72, 56, 2, 80, {'optimize': True}), ('s420_rst', 80, 80, 2, 90, {'restart_marker_blocks': 3}),
    ('s422_rst', 20, 130, 1, 95, {'restart_marker_rows': 1}), ('gray_q85', 45, 67, None, 85, {}),
    ('prog_q85', 48, 48, 2, 85, {'progressive': True}),
]


def picture(g, h, w, gray):
    small = g.integers(0, 256, size=(h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
    img = np.asarray(Image.fromarray(small).resize((w + 16, h + 16), Image.BICUBIC))[8:8 + h, 8:8 + w]
    a = np.clip(img.astype(np.int16) + g.integers(-20, 21, size=(h, w, 3)), 0, 255).astype(np.uint8)
    return a[:, :, 1] if gray else a


def main():
    os.makedirs(HERE, exist_ok=True)
    g = np.random.Generator(np.random.Philox(key=[2026, 10]))
    expected = {}
    for name, h, w, sub, q, kw in CASES:
        bio = io.BytesIO()
        opts = dict(quality=q, **kw)
        if sub is not None:
            opts['subsampling'] = sub
        Image.fromarray(picture(g, h, w, sub is None)).save(bio, 'JPEG', **opts)
        with open(os.path.join(HERE, name + '.jpg'), 'wb') as f:
            f.write(bio.getvalue())
        expected[name] = np.asarray(Image.open(io.BytesIO(bio.getvalue())))
    np.savez_compressed(os.path.join(HERE, 'expected.npz'), **expected)
    print('wrote', len(CASES), 'files,', sum(os.path.getsize(os.path.join(HERE, n)) for n in os.listdir(HERE)), 'bytes')


if __name__ == '__main__':
    main()
