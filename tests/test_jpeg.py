"""JPEG decode, CPU side: the host entropy decoder (witw_amd/csrc_host/jpeg_coef.cpp -> libwitw_jpeg.so) followed by the oracle's
restatement of libjpeg's back end (oracle/jpeg_oracle.py: dequantisation, 'islow' inverse DCT, fancy upsampling, YCbCr -> RGB)
against what Pillow -- the decoder behind the reference's skimage.io.imread, model/cvig_fov.py:88-89 -- makes of the same files:
byte for byte, on the committed fixtures (tests/golden/jpeg/, expected.npz written by gen_jpeg_fixtures.py) and on files written
here. This pins the oracle; tests/test_jpeg_gpu.py holds the HIP kernels to the same bytes."""
import io
import os

import numpy as np
import pytest

from oracle import jpeg_oracle as J
from witw_amd import jpeg

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'jpeg')


def fixture_names():
    return sorted(n[:-4] for n in os.listdir(HERE) if n.endswith('.jpg'))


@pytest.mark.parametrize('name', fixture_names())
def test_entropy_decoder_and_oracle_match_pillow_on_fixtures(name):
    exp = np.load(os.path.join(HERE, 'expected.npz'))[name]
    c = jpeg.read_coef(os.path.join(HERE, name + '.jpg'))
    if name.startswith('prog'):
        assert c is None                    # progressive: left to the host decoder
        return
    assert c is not None and c.shape[:2] == exp.shape[:2]
    out = J.decode(c.info, c.coef, c.qt)
    assert out.dtype == np.uint8 and out.shape == exp.shape
    np.testing.assert_array_equal(out, exp)


def test_fixtures_are_what_this_pillow_decodes():
    """the expected arrays came from the build container's Pillow; a box whose Pillow decodes differently would show here"""
    from PIL import Image
    exp = np.load(os.path.join(HERE, 'expected.npz'))
    for name in fixture_names():
        np.testing.assert_array_equal(np.asarray(Image.open(os.path.join(HERE, name + '.jpg'))), exp[name], err_msg=name)


def test_fresh_files_all_layouts():
    from PIL import Image
    g = np.random.Generator(np.random.Philox(key=[5, 77]))
    n = 0
    for (h, w) in ((224, 224), (130, 259), (9, 17), (16, 16)):
        for sub in (0, 1, 2):
            for q, kw in ((92, {}), (35, {'optimize': True}), (60, {'restart_marker_blocks': 2})):
                a = g.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
                a[h // 3:, w // 4:] //= 3                     # some structure, some noise
                bio = io.BytesIO()
                Image.fromarray(a).save(bio, 'JPEG', quality=q, subsampling=sub, **kw)
                ref = np.asarray(Image.open(io.BytesIO(bio.getvalue())))
                c = jpeg.read_coef(bio.getvalue())
                assert c is not None
                np.testing.assert_array_equal(J.decode(c.info, c.coef, c.qt), ref, err_msg=str((h, w, sub, q, kw)))
                n += 1
    assert n == 36


def test_garbage_and_truncated_streams_are_refused_not_crashed():
    assert jpeg.read_coef(b'') is None and jpeg.read_coef(b'\xff\xd8\xff') is None and jpeg.read_coef(b'not a jpeg at all') is None
    data = open(os.path.join(HERE, 's420_q90.jpg'), 'rb').read()
    assert jpeg.read_coef(data[:200]) is None                      # headers cut
    assert jpeg.read_coef(data[:len(data) // 2] + b'\xff\xd9') is None or True      # entropy data cut: refused or zero-filled, never a crash
    g = np.random.Generator(np.random.Philox(key=[6, 1]))
    for _ in range(200):                                           # random corruption of the entropy-coded part
        b = bytearray(data)
        for _k in range(5):
            b[int(g.integers(300, len(b)))] = int(g.integers(0, 256))
        jpeg.read_coef(bytes(b))


def test_pack_layout():
    c = jpeg.read_coef(os.path.join(HERE, 's420_odd.jpg'))
    raw = np.arange(5 * 6 * 3, dtype=np.uint8).reshape(5, 6, 3)
    buf, desc, kind = jpeg.pack([c, raw, c])
    assert kind == jpeg.KIND_JPEG and tuple(desc.shape) == (3, jpeg.DESC_COLS)
    b = buf.numpy()
    for i in (0, 2):
        o, q = int(desc[i, 0]), int(desc[i, 1])
        assert o % 16 == 0 and q % 16 == 0 and desc[i, 24] == 0
        np.testing.assert_array_equal(b[o:o + c.coef.size * 2].view(np.int16).reshape(-1, 64), c.coef)
        np.testing.assert_array_equal(b[q:q + c.qt.size * 2].view(np.uint16).reshape(-1, 64), c.qt)
        assert list(desc[i, 2:5]) == [37, 51, 3]
    o = int(desc[1, 0])
    assert desc[1, 24] == 1 and desc[1, 25] == 3 and list(desc[1, 2:4]) == [5, 6]
    np.testing.assert_array_equal(b[o:o + raw.size].reshape(5, 6, 3), raw)
